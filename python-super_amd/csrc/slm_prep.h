// slm_prep.h -- host interface of the once-per-frame data-term preparation (slm_prep.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "super_lm.h"

struct PrepBuffers;   // scratch shared by all slots of a solver (grow-only)

// Per-slot, grow-only device buffers that the per-iteration kernels read.
struct V1Plan {
  float* s_pts = nullptr;
  int32_t* s_idx = nullptr;
  float* s_w = nullptr;
  int32_t* grp_run = nullptr;
  int32_t* run_nodes = nullptr;
  double* slab = nullptr;
  int32_t* blk_key = nullptr;
  int32_t* blk_start = nullptr;
  int32_t* blk_entry = nullptr;
  int32_t* run_chunk = nullptr;   // (n_runs) 64-position chunk of each run
  int32_t* wg_first = nullptr;
  int32_t* wg_last = nullptr;
  uint8_t* run_lidx = nullptr;
  double* wgslab = nullptr;
  int32_t* blk2_start = nullptr;
  int32_t* blk2_entry = nullptr;
  int32_t nt_hint = 0;            // distinct KNN tuples of the frame this plan was last built for (0: none yet)
  bool legacy = false;            // a bin of this plan's frames did not fit the LDS sort of the binned preparation: rocPRIM pipeline from then on
  size_t cap_rchunk = 0, cap_wg = 0, cap_lidx = 0, cap_wgslab = 0, cap_b2start = 0, cap_b2entry = 0;
  size_t cap_pts = 0, cap_idx = 0, cap_w = 0, cap_grp = 0, cap_runs = 0, cap_slab = 0, cap_bkey = 0,
         cap_bstart = 0, cap_bentry = 0;
};

struct V1Sizes {
  int n_tuples, n_pos, n_runs, n_blocks;
  int n_wblk, max_wblk_per_wg;   // v2 records; v2 is usable when max_wblk_per_wg <= SLM_LB_MAX
  // hashes of the coupling graph, computed on the device: (J, K_ED, node KNN table) and the same continued over the
  // coupled-pair keys -- what the cached symbolic plan of a slot is compared with
  uint64_t knn_hash = 0, graph_hash = 0;
  bool bad_knn = false;          // a surfel KNN index outside [0, J) was seen: the frame must be refused
};

// ---- K-generic pair plan (any opt.num_neighbors in 1..8; reference super/loss.py:213-220 is K-generic) ----------------
// What the multifrontal solver needs from a frame whose surfels have K != 4 neighbours (the tuple-sorted MFMA assembly is a
// K = 4 structure): the sorted list of coupled node pairs (key a*J + b, a >= b -- the same list prep_v1 produces as blk_key)
// and, per surfel, the position in that list of each of its K(K+1)/2 node pairs in CANONICAL slot order (the surfel's node
// ids ascending c[0] < ... < c[K-1]; slot ra(ra+1)/2 + rb, rb <= ra, is the pair (c[ra], c[rb])), so that the per-iteration
// kernel (k_data_grad_pairs, slm_data.hip) adds a surfel's 7 x 7 blocks into compact per-pair records without searching;
// sf_perm lists the surfels ordered by their smallest ids -- surfels with the same neighbour set are adjacent there, and a
// wave that walks the list accumulates their common blocks in registers before it touches memory.
struct PairPlan {
  int32_t* blk_key = nullptr;    // (n_blocks) sorted pair keys
  int32_t* sf_pidx = nullptr;    // (N, K(K+1)/2) pair index of every (surfel, canonical slot)
  int32_t* sf_perm = nullptr;    // (N) surfel ids in neighbour-set order
  size_t cap_key = 0, cap_pidx = 0, cap_perm = 0;
};
struct PairSizes {
  int n_blocks = 0;
  uint64_t knn_hash = 0, graph_hash = 0;   // as V1Sizes
  bool bad_knn = false;
};
// Stream-synchronising (one read-back).  f.K in 1..8, f.J < 65536.
hipError_t prep_pairs(PrepBuffers*, const slm_frame& f, PairPlan& plan, PairSizes* out, hipStream_t st);
void pairplan_free(PairPlan& plan);

PrepBuffers* prep_create();
void prep_destroy(PrepBuffers*);
// Builds the plan for frame f (stream-synchronising: one small read-back, two for a plan's first frame).
hipError_t prep_v1(PrepBuffers*, const slm_frame& f, V1Plan& plan, V1Sizes* out, hipStream_t st);
// The range test of the surfel KNN table on its own (frames that do not take the tuple-sorted path): *bad = an index
// outside [0, J) exists.  Stream-synchronising (one 4-byte read-back).
hipError_t prep_check_knn(PrepBuffers*, const slm_frame& f, bool* bad, hipStream_t st);
void plan_free(V1Plan& plan);
