// slm_common.h -- shared device-side types and f64 math for libsuper_lm (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "super_lm.h"
#include "slm_nd.h"

#define SLM_NB 64            // scalar tile edge of the banded normal matrix
#define SLM_MAX_KED 8
#define SLM_K 4              // surfel->node neighbours (28-wide Jacobian rows)

// ---------------------------------------------------------------------------------
// A device pointer stored in a descriptor that lives in HBM (FrameDev).  Device code sees the field as a GLOBAL-address-
// space pointer: a plain `T*` loaded from memory is a generic pointer, every access through it becomes a FLAT instruction,
// and a pending flat access makes the compiler's wait-count pass wait for ALL loads in flight (s_waitcnt vmcnt(0)
// lgkmcnt(0)) at each use of a loaded value -- no prefetch, no software pipeline survives that.  With the address space in
// the field's type the accesses are global_load / global_store and the waits count.  (A cast at the use does not do it:
// generic -> global -> generic folds away before the address spaces are inferred.)  Same size and layout as T*; the host
// sees a plain pointer.
template <class T>
struct GP {
#if defined(__HIP_DEVICE_COMPILE__)
  __attribute__((address_space(1))) T* p;
  GP() = default;
  __host__ __device__ GP(T* q) : p((__attribute__((address_space(1))) T*)q) {}
  __host__ __device__ operator T*() const { return (T*)p; }
#else
  T* p;
  GP() = default;
  GP(T* q) : p(q) {}
  operator T*() const { return p; }
#endif
  __host__ __device__ T* operator->() const { return (T*)*this; }
  __host__ __device__ T* get() const { return (T*)*this; }
};

// ---------------------------------------------------------------------------------
// Per-slot LM state, resident in HBM; every kernel reads it, k_accept advances it.
struct LMState {
  double u;              // current damping
  double v;              // damping factor
  double minimal_loss;   // best loss so far
  int32_t iter;          // iterations finished
  int32_t stopped;       // 1 after a solver failure: later kernels become no-ops
  int32_t m_grad;        // matched-surfel counter of the current Jacobian pass
  int32_t m_loss;        // matched-surfel counter of the current loss pass
  int32_t chol_fail;     // set by the factorisation of the current iteration
  int32_t m_grad_local;  // surfel-sharded frames: this rank's own share of m_grad (k_pair_scatter puts the all-reduced count into
                         // m_grad; a Jacobian pass reused after a rejected step sends the share again, not the global count)
  // ---- evaluation buffer of the tuple-sorted data path (k_data_eval, slm_data_v1.hip) ----
  int32_t eval_valid;    // 1: fd.ev_rc holds the evaluation (r, c per position) at the slot's CURRENT beta
  int32_t m_eval;        // matched surfels (of this rank's share) of the evaluation in fd.ev_rc
  unsigned long long eval_acc;   // k_data_eval: blocks done << 32 | sum of their matched counts, ONE relaxed atomic per block
                                 // (no fence: a release / acquire at agent scope writes back / invalidates the XCD's whole L2)
};

// Per-slot descriptor, resident in HBM as an array indexed by blockIdx.y.
struct FrameDev {
  slm_frame f;           // caller's device pointers + sizes
  int32_t P;             // 7*J
  int32_t nt;            // tile columns = ceil(P / NB)
  int32_t wb;            // sub-diagonal tiles per tile column (tile half-bandwidth)
  int32_t bound;         // 1 once slm_bind_frame ran
  GP<double> beta;          // (J,7) current (== best in test phase)
  GP<double> delta;         // (nt*NB) solution of the damped system
  GP<double> rhs;           // (nt*NB) jtl, overwritten by the forward substitution
  GP<double> band;          // nt*(wb+1) tiles of NB*NB doubles, column-major inside a tile
  GP<double> linv;          // nt tiles: inverse of each diagonal Cholesky block
  GP<double> loss_part;     // per-block partial sums of the loss passes
  int32_t n_loss_part;
  int32_t pad2;
  GP<LMState> st;
  GP<slm_iter_record> rec;  // (num_iterations)
  // packed gather tables of the per-surfel evaluation (16-byte loads instead of scalar ones)
  GP<double> node_pk;       // (J,10): beta[0..6], g.x, g.y, g.z (float64: exact for either state dtype)
  GP<double> node_pk_try;   // same at the trial point beta + delta (loss pass of the LM loop)
  GP<float4> tgt_pn;        // (T,2): target point xyz0, target normal xyz0
  // (H*W,2) the same PER PIXEL (round 6): {point xyz, w = 1 where the pixel is mapped (index_map >= 0) else 0}, {normal xyz,
  // w = 1 where tgt_valid}.  A bilinear tap of the evaluation pass (k_data_eval) is then ONE 32-byte gather instead of
  // index_map -> row -> point/normal (two dependent sector reads per tap) and the validity of the rounded pixel -- which
  // is one of the four taps -- needs no load of its own; the two taps of an image row are adjacent in memory.
  GP<float4> tgt_px;
  GP<unsigned long long> dbg;  // diagnostic builds only (-DSLM_STAMPS): in-kernel s_memtime stamps
  // ---- tuple-sorted data-term assembly (slm_prep.hip / slm_data_v1.hip) ----
  int32_t v1_ready;      // 1 when the structures below are valid for this frame
  int32_t n_tuples;      // distinct canonical KNN 4-tuples
  int32_t n_pos;         // padded surfel positions (multiple of 64)
  int32_t n_runs;        // (tuple, 64-chunk) runs = slab entries
  int32_t n_blocks;      // distinct coupled node pairs (a >= b) of the data term
  int32_t pad3;
  GP<void> s_pts;           // (n_pos,3) surfel xyz in tuple-sorted, 4-padded order (dtype of the state: f.state_f64)
  GP<int32_t> s_idx;        // (n_pos,4) KNN ids (original order), -1 for padding positions
  GP<void> s_w;             // (n_pos,4) KNN weights (dtype of the state)
  GP<int32_t> grp_run;      // (n_pos/4) run id of each group of 4 positions, -1 for padding
  GP<int32_t> run_nodes;    // (n_runs,4) ascending node ids of each run's tuple
  GP<double> slab;          // (n_runs, 768) per-run Gram tiles 00,10,11 (16x16 row-major each)
  GP<int32_t> blk_key;      // (n_blocks) a*J + b
  GP<int32_t> blk_start;    // (n_blocks+1) CSR offsets into blk_entry
  GP<int32_t> blk_entry;    // run*16 + pa*4 + pb
  // ---- workgroup-merged Gram blocks (v2): the 4 waves of a workgroup add their runs' 7x7
  //      blocks into LDS accumulators keyed by node pair; one 56-double record per
  //      (workgroup, node pair) goes to HBM instead of one 768-double Gram per run ----
  int32_t v2_ready;
  int32_t n_wblk;          // (workgroup, pair) records
  GP<const int32_t> wg_first; // (n_wg) first record of each workgroup
  GP<const int32_t> wg_last;  // (n_wg) last record (first-1 if none)
  GP<const uint8_t> run_lidx; // (n_runs,10) local record index of the run's 10 node pairs
  GP<double> wgslab;          // (n_wblk, 56): 49 block entries (row-major ca,cb) + 7 entries of -J^T r
  GP<const int32_t> blk2_start;  // (n_blocks+1) CSR over blk2_entry, same pair order as blk_key
  GP<const int32_t> blk2_entry;  // record ids
  // ---- evaluation buffer: per tuple-sorted position {r, c.x, c.y, c.z} (r = lambda n.(T(p) - o), c = dr/dT(p)/lambda; zeros
  //      where unmatched / padding).  Written by every evaluation pass (k_data_eval: the loss pass of the LM loop, at the
  //      trial point), read by the Jacobian pass of the NEXT iteration when the step was accepted: the target-side half of
  //      the per-surfel work (projection, match test, 8 bilinear taps) is done once per iteration, not twice ----
  GP<double> ev_rc;
  // ---- one frame sharded over several GPUs (slm_set_shard): this rank evaluates the workgroups
  //      [wg_lo, wg_hi) of the Jacobian pass and the surfels [sf_lo, sf_hi) of the loss pass; the
  //      per-pair sums travel through pairbuf (n_blocks x 56 doubles + matched count) ----
  int32_t wg_lo, wg_hi;
  int32_t sf_lo, sf_hi;
  GP<double> pairbuf;
  // ---- K-generic pair path (num_neighbors != 4 on the multifrontal solver; slm_prep.hip prep_pairs): the data term's
  //      7 x 7 blocks are added into pairbuf (n_blocks x 56 doubles + matched count) by k_data_grad_pairs -- per surfel
  //      the record of each of its K(K+1)/2 canonical pair slots is known (sf_pidx), surfels are walked in neighbour-set
  //      order (sf_perm) -- and placed into the fronts by k_pair_scatter, exactly like an all-reduced sharded frame ----
  int32_t vk_ready;
  int32_t pad7;
  GP<const int32_t> sf_pidx;
  GP<const int32_t> sf_perm;
  // ---- nested-dissection multifrontal solver (slm_nd_host.hip / slm_front.hip) ----
  int32_t nd_ready;      // 1 when the plan below is valid for this frame
  int32_t n_fronts;
  int32_t n_levels;
  int32_t pad4;
  GP<const int32_t> level_start;  // (n_levels+1) fronts of level l: [level_start[l], level_start[l+1])
  GP<const NDFront> fronts;       // processing order (deepest level first)
  GP<const int32_t> nd_nodes;     // per front: pivot node ids then boundary node ids
  GP<const int32_t> nd_eamap;     // per front: boundary index -> local node position in the parent
  GP<const int32_t> node_front;   // (J) front eliminating the node
  GP<const int32_t> node_pos;     // (J) local pivot position
  GP<const NDDest> block_dest;    // (n_blocks) destination of every data-term block
  GP<const NDDest> pair_dest;     // (J*K_ED) destination of every ARAP pair block
  GP<const NDTileItem> tile_items;  // work lists of k_fschur / k_fpull (slm_nd.h)
  GP<const int32_t> item_off;       // (2*n_levels + 1): per level [Schur items | pull items]
  GP<const int32_t> in_start;     // (J+1) ARAP edges grouped by target node (reverse KNN graph)
  GP<const int32_t> in_edge;
  GP<double> ftiles;              // front tile storage
  GP<double> fvec;                // front vectors (rhs -> y -> x)
  GP<double> flinv;               // inverses of the diagonal Cholesky blocks of the fronts
  GP<double> fmail;               // task graph: per pivot tile column a mailbox of SLM_MAIL_DOUBLES (slm_tile.h) for the streamed hand-off
  long long zero_tile_doubles;    // leading part of ftiles / all of fvec that k_iter_begin_nd zeroes before an assembly
  // Pivot-column tiles by what reaches them (round 5; per slot, from the plan's destinations -- slm_api.hip):
  //   tile_kind[tile number] = 1: PURE FILL -- no assembled block lands in the tile (data term, ARAP, Rot, node diagonal
  //   blocks) and some child maps into it.  Such a tile is never zeroed; its first toucher -- the pull that gathers the
  //   children's updates (k_fpull, or the tile's own task of the task graph) -- starts from zero and STORES the whole tile.
  //   0: everything else (zeroed before the assembly, added into).  zero_tiles lists the offsets (doubles) of the kind-0
  //   pivot-column tiles: what k_iter_begin_nd zeroes.
  GP<const uint8_t> tile_kind;
  GP<const long long> zero_tiles;
  int32_t n_zero_tiles;
  int32_t pad6;
  long long zero_vec_doubles;
  // ---- persistent task-graph solver (slm_dag.hip): task list of the plan + per-iteration flags ----
  GP<const int32_t> dag_tasks;    // (n_dag_tasks, 2) task words (slm_nd.h), in a topological order
  GP<const int32_t> front_kids;   // (n_fronts, 2) children with a boundary (front index or -1)
  GP<const int32_t> pull_off;     // (n_fronts) offset of the front's pull map in pullmap (-1: none)
  GP<const int32_t> pullmap;      // per child: parent scalar index -> boundary scalar index of the child, -1
  GP<const int32_t> prng_off;     // (n_fronts) offset into prng
  GP<const int32_t> prng;         // per front, tile row, child: child boundary tile rows lo | hi << 8 it gathers from, -1 none
  GP<int32_t> dag_flags;          // [0] ticket, [1] abort, [8..] per tile done, per pivot column {b, x, y, 4 x 16 pivots out}
  int32_t n_dag_tasks;
  GP<const int32_t> dag_top_tasks;   // the tasks of the fronts of depth <= dag_cut_depth, same order (hybrid solve)
  int32_t n_dag_top_tasks, dag_cut_depth;
  int32_t dag_n_tiles;         // tiles of all fronts
  int32_t dag_n_pcols;         // pivot tile columns of all fronts
  int32_t dag_n_flags;         // ints in dag_flags
  GP<long long> dag_trace;        // diagnostics (slm_debug_dag_trace): per task {start, ready, end} in 10 ns ticks + workgroup; else null
};
// slm_frame (include/super_lm.h) as DEVICE code reads it: the same bytes, the caller's pointers typed GP<> so that the
// accesses through them are global, not flat (see GP above).  Host code keeps using FrameDev::f.
struct FrameIn {
  int32_t N, J, T, H, W, K, K_ED;
  float fx, fy, cx, cy;
  GP<const void> sf_points;
  GP<const int32_t> sf_knn_idx;
  GP<const void> sf_knn_w;
  GP<const void> ed_points;
  GP<const int32_t> ed_knn_idx;
  GP<const float> tgt_points;
  GP<const float> tgt_norms;
  GP<const int32_t> index_map;
  GP<const uint8_t> tgt_valid;
  int32_t state_f64;
  int32_t pad;
};
static_assert(sizeof(FrameIn) == sizeof(slm_frame), "FrameIn mirrors slm_frame");
static_assert(offsetof(FrameIn, fx) == offsetof(slm_frame, fx) && offsetof(FrameIn, sf_points) == offsetof(slm_frame, sf_points) &&
              offsetof(FrameIn, ed_knn_idx) == offsetof(slm_frame, ed_knn_idx) && offsetof(FrameIn, tgt_valid) == offsetof(slm_frame, tgt_valid) &&
              offsetof(FrameIn, state_f64) == offsetof(slm_frame, state_f64), "FrameIn mirrors slm_frame");
__device__ __forceinline__ const FrameIn& frame_in(const FrameDev& fd) { return reinterpret_cast<const FrameIn&>(fd.f); }

#define SLM_SLAB_STRIDE 768
#define SLM_WREC 56          // doubles per (workgroup, pair) record
#define SLM_LB_MAX 96        // records a workgroup can hold in LDS
#define SLM_VK_TAIL 64        // K-generic pair path: the matched count behind the pair records, spread over this many doubles (one
                             // atomic per wave onto ONE address costs 12 ns each, serialised: tools/micro/atomic_same_mb.hip)

// In-kernel stamps exist only in the diagnostic build; the shipped library executes none.
#ifdef SLM_STAMPS
#define SLM_STAMP(fd, cond, idx)                                                       \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    if ((cond) && (fd).dbg) {                                                          \
      unsigned long long t_;                                                           \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
      if (threadIdx.x == 0) (fd).dbg[idx] = t_;                                        \
    }                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
#else
#define SLM_STAMP(fd, cond, idx) do { } while (0)
#endif

struct d3 {
  double x, y, z;
};
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator*(double s, d3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 cross(d3 a, d3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// R(q)x = x + 2w(v x x) + 2 v x (v x x), un-normalised q (reference super/utils.py:49-54).
__device__ __forceinline__ d3 quat_apply(double w, d3 v, d3 x) {
  d3 c = cross(v, x);
  d3 c2 = cross(v, c);
  return {x.x + 2.0 * w * c.x + 2.0 * c2.x, x.y + 2.0 * w * c.y + 2.0 * c2.y,
          x.z + 2.0 * w * c.z + 2.0 * c2.z};
}

// c^T d(R(q)x)/dq for a row vector c: out[0] = d/dw, out[1..3] = d/dv
// (reference super/utils.py:59-69 contracted with c):
//   c.dw = 2 c.(v x x);  c.dv = 2[(v.x)c + (c.v)x - 2(c.x)v - w (c x x)]
__device__ __forceinline__ void quat_jac_row(double w, d3 v, d3 x, d3 c, double out[4]) {
  d3 vx = cross(v, x);
  out[0] = 2.0 * dot(c, vx);
  double vdx = dot(v, x), cdv = dot(c, v), cdx = dot(c, x);
  d3 cxx = cross(c, x);
  out[1] = 2.0 * (vdx * c.x + cdv * x.x - 2.0 * cdx * v.x - w * cxx.x);
  out[2] = 2.0 * (vdx * c.y + cdv * x.y - 2.0 * cdx * v.y - w * cxx.y);
  out[3] = 2.0 * (vdx * c.z + cdv * x.z - 2.0 * cdx * v.z - w * cxx.z);
}

// Full 3x4 Jacobian d(R(q)x)/dq, Jq[i][j] (used by the ARAP rows).
__device__ __forceinline__ void quat_jac(double w, d3 v, d3 x, double Jq[3][4]) {
  d3 c = cross(v, x);
  Jq[0][0] = 2.0 * c.x;
  Jq[1][0] = 2.0 * c.y;
  Jq[2][0] = 2.0 * c.z;
  double vdx = dot(v, x);
  double vv[3] = {v.x, v.y, v.z}, xx[3] = {x.x, x.y, x.z};
  // skew(x)[i][j]: [x]x
  double S[3][3] = {{0.0, -x.z, x.y}, {x.z, 0.0, -x.x}, {-x.y, x.x, 0.0}};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      Jq[i][j + 1] = 2.0 * ((i == j ? vdx : 0.0) + vv[i] * xx[j] - 2.0 * xx[i] * vv[j] - w * S[i][j]);
}

// address of entry (i,j), i >= j, of the lower band, tiles column-major, NB x NB
__device__ __forceinline__ double* band_entry(const FrameDev& fd, int i, int j) {
  int tr = i / SLM_NB, tc = j / SLM_NB;
  size_t tile = (size_t)tc * (fd.wb + 1) + (tr - tc);
  return fd.band + tile * (SLM_NB * SLM_NB) + (i - tr * SLM_NB) + (size_t)(j - tc * SLM_NB) * SLM_NB;
}

// ---- model-state loads: float32 or float64 arrays (slm_frame.state_f64, uniform per slot) ----
__device__ __forceinline__ d3 ld_state3(const void* p, size_t i, int f64) {
  if (f64) {
    const double* q = static_cast<const double*>(p) + 3 * i;
    return {q[0], q[1], q[2]};
  }
  const float* q = static_cast<const float*>(p) + 3 * i;
  return {(double)q[0], (double)q[1], (double)q[2]};
}
__device__ __forceinline__ void ld_state4(const void* p, size_t i, int f64, double w[4]) {
  if (f64) {
    const double2* q = reinterpret_cast<const double2*>(static_cast<const double*>(p) + 4 * i);
    const double2 a = q[0], b = q[1];
    w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y;
  } else {
    const float4 v = *reinterpret_cast<const float4*>(static_cast<const float*>(p) + 4 * i);
    w[0] = (double)v.x; w[1] = (double)v.y; w[2] = (double)v.z; w[3] = (double)v.w;
  }
}
__device__ __forceinline__ double ld_state1(const void* p, size_t i, int f64) {
  return f64 ? static_cast<const double*>(p)[i] : (double)static_cast<const float*>(p)[i];
}

#define SLM_NPK 10   // doubles per node in node_pk

__device__ __forceinline__ void pack_node(double* dst, const double bb[7], d3 g) {
#pragma unroll
  for (int c = 0; c < 7; ++c) dst[c] = bb[c];
  dst[7] = g.x;
  dst[8] = g.y;
  dst[9] = g.z;
}
// node position from the packed table (what the per-iteration kernels read instead of f.ed_points)
__device__ __forceinline__ d3 node_pos_pk(const double* npk, int j) {
  const double* q = npk + (size_t)SLM_NPK * j + 7;
  return {q[0], q[1], q[2]};
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
  // lowers to one global_atomic_add_f64 on gfx950 (no compare-and-swap loop)
  unsafeAtomicAdd(p, v);
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return x;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in thread 0
__device__ __forceinline__ double block_sum(double x, double* smem /* >= 16 doubles */) {
  x = wave_sum(x);
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) smem[wv] = x;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += smem[i];
  }
  __syncthreads();
  return r;
}
