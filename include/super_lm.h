/*
 * super_lm.h -- C ABI of libsuper_lm.so: the MI355X (gfx950) implementation of
 * SuPer's per-frame embedded-deformation Levenberg-Marquardt step.
 *
 * Drop-in boundary.  Each entry point names the reference
 * interface it replaces (paths relative to the reference repository root):
 *
 *   slm_create / slm_destroy   <- LM_Solver.__init__            super/LM.py:11-34
 *   slm_bind_frame             <- {Data,ARAP,Rot}Loss.prepare   super/loss.py:212-220,408-426,480-485
 *   slm_prepare_model          <- its model-side half, ahead of the frame (see there)
 *   slm_run                    <- LM_Solver.LM                  super/LM.py:81-122
 *   slm_assemble               <- LM_Solver.prepareCostTerm(grad=True)   super/LM.py:54-68
 *   slm_loss                   <- LM_Solver.prepareCostTerm(grad=False)  super/LM.py:70-78
 *   slm_solve / slm_solve_dense <- LM_Solver.Solver             super/LM.py:37-51
 *   slm_data_residuals         <- DataLoss.forward internals    super/loss.py:222-255
 *   slm_apply_update           <- Surfels.update                super/nodes.py:193-223
 *   slm_knn                    <- Surfels.update_sfed_knn / update_ed / find_knn
 *                                                               super/nodes.py:154-191, utils/utils.py:212-221
 *
 * Conventions
 *   - Every pointer marked "device" is a HIP device pointer owned by the caller;
 *     the library never frees or reallocates caller memory.
 *   - Indices are int32.  The MODEL STATE (surfel positions, skinning weights, node positions --
 *     float64 tensors in the reference, true float64 values from the first Surfels.update on) is
 *     read as float64 when slm_frame.state_f64 != 0 (what the Python mirror passes: the caller's
 *     tensors are used in place, nothing is rounded) or as float32 (state_f64 == 0: the compact
 *     layout of BASELINE.json's fp32 configs).  The per-frame target tables are float32: the
 *     reference widens float32 back-projections to float64 (utils/data_loader.py:453-462), so
 *     float32 holds them exactly.  All arithmetic, the normal equations and the solve are float64
 *     (the reference computes in float64).
 *   - beta is (J,7) float64 row-major [qw,qx,qy,qz,bx,by,bz] (super/LM.py:85-88).
 *   - Every call returns an int status (SLM_OK == 0), never throws, and takes the
 *     hipStream_t (as void*) it enqueues on.  Only slm_bind_frame (one 4-byte
 *     read-back to size the band), slm_get_records and slm_status synchronise.
 *   - A solver owns `max_frames` independent slots; slm_run advances the LM
 *     problems of slots [0, n_frames) together in the same launches (the
 *     many-frame / many-hypothesis batch dimension).
 */
#ifndef SUPER_LM_H
#define SUPER_LM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  SLM_OK = 0,
  SLM_ERR_INVALID = 1,      /* bad argument */
  SLM_ERR_HIP = 2,          /* a HIP runtime call failed (slm_last_error has the text) */
  SLM_ERR_NO_DEVICE = 3,    /* no gfx950 device / kernels cannot run */
  SLM_ERR_UNBOUND = 4,      /* slot used before slm_bind_frame */
  SLM_ERR_UNSUPPORTED = 5   /* e.g. num_neighbors outside 1..8, frames of one batch with different num_neighbors */
};

/* per-iteration status in slm_iter_record.status */
enum {
  SLM_ITER_OK = 0,
  SLM_ITER_SOLVER_FAILED = 1, /* Cholesky pivot <= 0: "Solver failed: Ill-posed system!" (super/LM.py:99-103) */
  SLM_ITER_NOT_RUN = 2,       /* iteration skipped because an earlier one stopped the loop */
  SLM_ITER_SOLVER_TIMEOUT = 3 /* the task-graph solve gave up waiting (a scheduling time-out of the persistent launch, not
                                 a property of the matrix); the loop stops like after a failed factorisation */
};

/* ABI handshake.  The structs of this header are passed by pointer and have grown between rounds; a caller built
 * against another revision of the header must get an error, not out-of-bounds reads.  slm_abi_version() returns
 * SLM_ABI_VERSION of the header the library was built from; slm_abi_check() also compares the caller's sizeof of
 * the five structs that carry pointers or were extended (returns SLM_OK or SLM_ERR_INVALID with the mismatch in
 * slm_last_error()).  Bindings call it once after loading the library (super_amd/_lib.py does). */
#define SLM_ABI_VERSION 3
#define SLM_PLAN_INFO_DOUBLES 14
int slm_abi_version(void);
int slm_abi_check(int32_t abi_version, int32_t sizeof_slm_config, int32_t sizeof_slm_frame, int32_t sizeof_slm_gf_config,
                  int32_t sizeof_slm_gf_frame, int32_t sizeof_slm_iter_record);

typedef struct slm_solver slm_solver; /* opaque */

/* Flags read from `opt` by LM_Solver.__init__/LM (super/LM.py:18-31,81-82). */
typedef struct slm_config {
  int32_t num_iterations;   /* opt.num_optimize_iterations (options.py:42), default 10 */
  int32_t phase_test;       /* opt.phase == "test": accept/reject; 0 = "train": always accept */
  int32_t use_data;         /* opt.sf_point_plane */
  int32_t use_arap;         /* opt.mesh_arap */
  int32_t use_rot;          /* opt.mesh_rot */
  int32_t max_frames;       /* number of slots (>= 1).  Performance note (results do not depend on it): the symbolic plan of
                             * a slot is built for the solver form this slot count implies -- a solver whose max_frames x J
                             * fits ONE task-graph launch (<= 8 000 frames x nodes, or <= 2 slots) dissects down to 50-node
                             * leaves, any other to 18-node leaves -- and cached plans keep the leaf size they were built
                             * with, whatever n_frames a later slm_run passes.  Create the solver with the slot count it
                             * will be run with. */
  int32_t data_path;        /* 0 = tuple-sorted MFMA assembly, node-pair blocks merged per workgroup in
                               LDS (default); 1 = per-entry f64 atomics (simple cross-check path, also
                               used when J >= 65536); 2 = MFMA assembly with one Gram per run in HBM */
  int32_t solver_path;      /* 0..4, anything else is rejected by slm_create.  0 = nested-dissection multifrontal Cholesky (default, needs data_path 0); its numeric
                               phase runs as ONE persistent launch over a static task graph (per-tile flags instead
                               of launch boundaries: the latency form) for small launches (one or two frames, or frames x nodes
                               <= 8 000: three or four frames of 2 000 nodes, eight of 512), and as one
                               launch per level / tile column / phase for larger batches (the throughput form),
                               with the top of the tree (the root front and its children: a chain of dependent tile
                               columns) as a task graph over all frames when the frames' trees have the same depth
                               (the hybrid form); 2 / 3 / 4 force the task-graph / the per-level / the hybrid form
                               (4 falls back to 3 when the trees differ in depth); 1 = block-banded Cholesky */
  double w_data;            /* opt.sf_point_plane_weight (1.0) */
  double w_arap;            /* opt.mesh_arap_weight (10.0) */
  double w_rot;             /* opt.mesh_rot_weight (1.0) */
  double u0;                /* LM(..., u=10)            super/LM.py:81 */
  double v;                 /* LM(..., v=7.5)                          */
  double minimal_loss0;     /* LM(..., minimal_loss=1e10)              */
} slm_config;

/* What LM reads from sf / inputs / new_data, as device pointers. */
typedef struct slm_frame {
  int32_t N;                /* surfels */
  int32_t J;                /* ED nodes (sf.ED_nodes.num) */
  int32_t T;                /* rows of the target tables */
  int32_t H, W;             /* image size (inputs[("color",0)] shape) */
  int32_t K;                /* surfel->node neighbours (opt.num_neighbors, README.md:175 / options.py:49): 1..8.  4, the reference's
                             * default, takes the tuple-sorted MFMA assembly; any other value the K-generic pair path (one sort of
                             * the coupled-pair keys at the bind, per-pair records filled by run-length-accumulated atomics) -- both
                             * on the SAME nested-dissection multifrontal solver in every form, and both accepted by the
                             * surfel-sharded mode (data_path = 1 / solver_path = 1 / J >= 65536: per-entry atomics into the band +
                             * block-banded solve); the frames of one batch share their K */
  int32_t K_ED;             /* node->node neighbours (opt.num_ED_neighbors), 1..8 */
  float fx, fy, cx, cy;     /* inputs["K"][0] entries [0,0],[1,1],[0,2],[1,2] (float32 like the reference) */
  const void* sf_points;      /* device (N,3)      sf.points            float32, or float64 with state_f64 */
  const int32_t* sf_knn_idx;  /* device (N,K)      sf.knn_indices */
  const void* sf_knn_w;       /* device (N,K)      sf.knn_w             float32 / float64 */
  const void* ed_points;      /* device (J,3)      sf.ED_nodes.points   float32 / float64 */
  const int32_t* ed_knn_idx;  /* device (J,K_ED)   sf.ED_nodes.knn_indices */
  const float* tgt_points;    /* device (T,3)      new_data.points */
  const float* tgt_norms;     /* device (T,3)      new_data.norms */
  const int32_t* index_map;   /* device (H,W)      new_data.index_map, -1 = invalid */
  const uint8_t* tgt_valid;   /* device (H*W)      new_data.valid */
  int32_t state_f64;          /* 0: sf_points / sf_knn_w / ed_points are float32; 1: float64 (the reference's
                                 own dtype: no rounding of the model state between frames) */
  int32_t pad;
} slm_frame;

typedef struct slm_iter_record {
  double loss;        /* sum of squared residuals at the trial beta (super/LM.py:107) */
  double u;           /* damping used for this iteration's solve */
  int32_t accepted;   /* 1 = step accepted */
  int32_t status;     /* SLM_ITER_* */
  int32_t M_grad;     /* matched surfels in the Jacobian pass */
  int32_t M_loss;     /* matched surfels in the loss pass (fresh match set) */
} slm_iter_record;

/* -- lifetime ------------------------------------------------------------------ */
int slm_create(const slm_config* cfg, slm_solver** out);
int slm_destroy(slm_solver* s);
const char* slm_last_error(void);
/* number of visible HIP devices, or 0 (never initialises a context) */
int slm_device_count(void);

/* -- per-frame binding (loss_term.prepare) ------------------------------------- */
/* Binds device pointers to `slot`, builds the frame's assembly plan (tuple-sorted surfel copies, coupled node pairs)
 * and the symbolic plan of the solver (kept while the coupling graph is unchanged), (re)sizes the slot's workspace and
 * resets beta to identity.  Stream-synchronising (one small device->host read when the slot's plan still applies).
 * A surfel KNN index outside [0, J) -- an IndexError in the reference, super/loss.py:189-197 -- is found on the device
 * and refused on every data path: SLM_ERR_INVALID, nothing was read out of bounds.  A bind that fails leaves the slot UNBOUND (slm_run
 * and friends return SLM_ERR_UNBOUND for it) until a later bind succeeds. */
int slm_bind_frame(slm_solver* s, int32_t slot, const slm_frame* frame, void* stream);
/* The same for the n_frames frames of a batch, slots [first_slot, first_slot + n_frames), `frames` an array in
 * host memory.  The preparation of a frame is a chain of small launches and size read-backs (latency, not
 * throughput), so the frames are bound concurrently: one host thread, stream and set of scratch buffers per frame
 * inside the library (at most 8 at a time), forked from `stream` (they see the work enqueued on it so far) and
 * joined back into it before the call returns.  Results are those of n_frames slm_bind_frame calls. */
int slm_bind_frames(slm_solver* s, int32_t first_slot, int32_t n_frames, const slm_frame* frames, void* stream);

/* The MODEL-side half of slm_bind_frame, ahead of time and asynchronously.  What a bind derives from the surfel model
 * alone -- the tuple-sorted copies and the pair index of the data term (from sf.points, sf.knn_indices, sf.knn_w), the
 * hash of the coupling graph and, when it changed, the symbolic plan of the solver (a host analysis) -- is known as soon
 * as the PREVIOUS frame has finished with the model (after Surfels.update / fuseInputData / the swap: super/super.py:66-73,
 * super/nodes.py:170-191), long before the next frame's target exists.  `model` carries N, J, K, K_ED, state_f64 and the
 * five model pointers (sf_points, sf_knn_idx, sf_knn_w, ed_points, ed_knn_idx); the target fields are ignored.  The call
 * records an event on `stream` (the preparation sees everything enqueued on it so far), queues the work on the solver's
 * own worker thread and stream, and RETURNS AT ONCE; the slot is unbound until the next slm_bind_frame.  That bind joins
 * the preparation; if its model fields are the ones prepared it only binds the target side (no sort, no read-back of
 * sizes, no analysis inside the frame's critical path), otherwise it discards the preparation and binds in full.  The
 * caller must leave the model arrays unchanged (and alive) between the two calls.  An error of the preparation is
 * returned by the bind that consumes it.  Replaces the model-side part of loss_term.prepare, super/loss.py:212-220,
 * 408-426, moved to where its inputs become final. */
int slm_prepare_model(slm_solver* s, int32_t slot, const slm_frame* model, void* stream);
/* Drops the slot's queued / finished preparation: the next slm_bind_frame binds in full whatever its pointers are.  The
 * library can only compare SIZES and POINTERS of the model arrays; a caller that knows the arrays were rewritten in place
 * after slm_prepare_model (same buffers, new values -- the host mirror sees it in the tensors' version counters) calls
 * this before the bind, otherwise the bind would consume a plan sorted from the old values.  Joins the worker; the bind
 * that follows is ordered behind the preparation's launches either way. */
int slm_discard_prepared(slm_solver* s, int32_t slot);

/* -- the LM loop ------------------------------------------------------------------ */
/* Enqueues num_iterations damped accept/reject iterations for slots [0,n_frames),
 * entirely on the device (no host synchronisation).  Running again without binding continues the loop from the slot's
 * current state; the library may then reuse what it derived from the bound input buffers (the Jacobian records of a
 * rejected step), so a caller that changes those buffers IN PLACE must bind the slot again before the next run. */
int slm_run(slm_solver* s, int32_t n_frames, void* stream);
/* Copies the slot's current beta (J*7 doubles) into caller device memory. */
int slm_get_beta(slm_solver* s, int32_t slot, double* beta_out_device, void* stream);
/* Overwrites the slot's beta from caller device memory (hypotheses / warm start / tests). */
int slm_set_beta(slm_solver* s, int32_t slot, const double* beta_in_device, void* stream);
/* Synchronises `stream` and copies the per-iteration records to HOST memory. */
int slm_get_records(slm_solver* s, int32_t slot, slm_iter_record* host_out, int32_t max_records,
                    void* stream);

/* Host-side facts about the slot's per-frame plan (after slm_bind_frame).  info_out has room for `capacity` doubles;
 * the first min(capacity, SLM_PLAN_INFO_DOUBLES) entries are written (a caller built for fewer entries is never
 * overrun, one that asks for more gets zeros):
 * [0] solver in use (0 nested dissection, 1 band), [1] fronts, [2] tree levels,
 * [3] FLOPs of one factorisation (padded dense fronts, or P*w^2 for the band),
 * [4] factor storage bytes, [5] distinct KNN tuples, [6] Gram runs, [7] coupled node pairs,
 * [8] workgroup-merged (workgroup, pair) records (0: one Gram per run in HBM), [9] padded positions,
 * [10] FLOPs of one factorisation without the padding of the fronts to 64, [11] tasks of the task-graph solver,
 * [12] pivot-column tiles of the fronts, [13] of them PURE FILL: no assembled block reaches them, some child maps into
 *      them -- neither zeroed per iteration nor read by their first toucher (0 with SLM_PURE_FILL=0). */
int slm_get_plan_info(slm_solver* s, int32_t slot, double* info_out, int32_t capacity);

/* -- one LARGE frame sharded over the GPUs of a node ----------------
 * After slm_set_shard(rank, world) (before slm_bind_frame; every rank binds the WHOLE frame so that
 * all ranks build the same plan; world == 1 is allowed and runs the same exchange protocol on one rank) a context evaluates only its share of the surfels: workgroups
 * [n_wg*rank/world, ...) of the Jacobian pass and surfels [N*rank/world, ...) of the loss pass.
 * The library holds no communicator; per LM iteration the caller runs on every rank
 *     slm_lm_grad_local   zero, data-term Gram records of the share, per-pair partial sums
 *       -> all-reduce(sum) of SLM_X_PAIR_BLOCKS   (coupled pairs x 56 doubles + matched count;
 *                                                   7 MB at C2 = the block-sparse J^T J and J^T r)
 *     slm_lm_solve        reduced blocks -> fronts, ARAP / Rot rows, factor + solve
 *       -> broadcast of SLM_X_DELTA from rank 0   (keeps every rank on bit-identical parameters)
 *     slm_lm_loss_local   trial point, data loss of the share, regularisation loss
 *       -> all-reduce(sum) of SLM_X_DATA_LOSS     (1024 doubles)
 *     slm_lm_accept       accept / reject, record
 * with slm_lm_exchange_get / _set moving the buffers device to device to and from the caller's
 * tensor (torch.distributed = RCCL over xGMI on the GPU box). */
#define SLM_X_PAIR_BLOCKS 0
#define SLM_X_DELTA 1
#define SLM_X_DATA_LOSS 2
int slm_set_shard(slm_solver* s, int32_t rank, int32_t world);
int slm_lm_grad_local(slm_solver* s, int32_t n_frames, void* stream);
int slm_lm_solve(slm_solver* s, int32_t n_frames, void* stream);
int slm_lm_loss_local(slm_solver* s, int32_t n_frames, void* stream);
int slm_lm_accept(slm_solver* s, int32_t n_frames, void* stream);
int slm_lm_exchange_size(slm_solver* s, int32_t slot, int32_t what, int64_t* n_doubles);
/* The library's own exchange buffer (device pointer, n_doubles long): a caller whose collective can run on foreign
 * device memory reduces IN PLACE on it (no get / set copies).  Valid until the slot is bound again. */
int slm_lm_exchange_ptr(slm_solver* s, int32_t slot, int32_t what, double** device_ptr_out, int64_t* n_doubles);
int slm_lm_exchange_get(slm_solver* s, int32_t slot, int32_t what, double* out_device, void* stream);
int slm_lm_exchange_set(slm_solver* s, int32_t slot, int32_t what, const double* in_device, void* stream);

/* -- phase timing (bench.py roofline leg) ----------------------------------------- */
/* With profiling on, slm_run brackets each phase of every LM iteration with HIP events
 * recorded on the launch stream.  Phases: */
enum {
  SLM_PH_ZERO = 0,       /* band / rhs zeroing */
  SLM_PH_DATA_GRAD = 1,  /* fused data-term Jacobian pass: exactly ONE kernel launch per iteration */
  SLM_PH_REG_GRAD = 2,   /* 7x7 block gather into the band + ARAP / Rot Jacobian pass */
  SLM_PH_SOLVE = 3,      /* banded Cholesky factor + substitutions (many launches) */
  SLM_PH_DATA_LOSS = 4,  /* fused data-term loss pass: exactly ONE kernel launch per iteration */
  SLM_PH_ACCEPT = 5,     /* regulariser loss + accept/reject */
  SLM_PH_COUNT = 6
};
int slm_profile_enable(slm_solver* s, int32_t on);
/* Synchronises the recorded events; adds per-phase elapsed milliseconds and interval
 * counts into ms_out[SLM_PH_COUNT] / count_out[SLM_PH_COUNT] (host arrays, overwritten),
 * then clears the accumulated intervals. */
int slm_profile_read(slm_solver* s, double* ms_out, int64_t* count_out);

/* -- parity / building-block entry points -------------------------------------- */
/* JtJ and jtl = -Jt r at the slot's current beta.  jtj_dense_device is (P,P)
 * float64 row-major (symmetric, full) or NULL; jtl_device is (P) or NULL. P = 7J. */
int slm_assemble(slm_solver* s, int32_t slot, double* jtj_dense_device, double* jtl_device,
                 void* stream);
/* Sum of squared residuals per term at the slot's current beta:
 * out_device[0..2] = data, arap, rot; out_device[3] = matched count (as double). */
int slm_loss(slm_solver* s, int32_t slot, double* out_device, void* stream);
/* Assemble at the current beta, add u to the diagonal, Cholesky-solve; writes delta (P).
 * status_device (int32) receives SLM_ITER_OK or SLM_ITER_SOLVER_FAILED. */
int slm_solve(slm_solver* s, int32_t slot, double u, double* delta_device,
              int32_t* status_device, void* stream);
/* LM_Solver.Solver(A, b, "cholesky") on a caller-supplied dense system: A_device is
 * (P,P) float64 row-major symmetric positive definite, b_device (P); writes x (P) and
 * status (SLM_ITER_OK / SLM_ITER_SOLVER_FAILED).  Compatibility entry point (allocates
 * and frees its workspace, synchronises `stream`); the LM loop never uses it. */
int slm_solve_dense(int32_t P, const double* A_device, const double* b_device, double* x_device,
                    int32_t* status_device, void* stream);
/* Per-surfel data-term internals at the current beta: r_device (N) float64
 * (lambda * n.(T(p)-o), 0 where unmatched), match_device (N) uint8,
 * taps_device (N,4) int32 target rows of the four bilinear taps (-1 invalid). */
int slm_data_residuals(slm_solver* s, int32_t slot, double* r_device, uint8_t* match_device,
                       int32_t* taps_device, void* stream);

/* -- Surfels.update (LM variant, no global row) ---------------------------------- */
/* In place on float32 device arrays: skin points, blend+normalise normals, move
 * nodes, rotate node normals.  beta_device is (J,7) float64. */
int slm_apply_update(int32_t N, int32_t J, int32_t K, float* sf_points, float* sf_norms,
                     const int32_t* sf_knn_idx, const float* sf_knn_w, float* ed_points,
                     float* ed_norms, const double* beta_device, void* stream);
/* The same in place on float64 device arrays (the reference's dtype; nothing is rounded). */
int slm_apply_update_f64(int32_t N, int32_t J, int32_t K, double* sf_points, double* sf_norms,
                         const int32_t* sf_knn_idx, const double* sf_knn_w, double* ed_points,
                         double* ed_norms, const double* beta_device, void* stream);

/* -- KNN feeder --------------------------------------------------------------------- */
/* K nearest `nodes` for every query point: squared L2 in float64, ascending, ties ->
 * lowest index.  skip_self != 0 drops the first (self) hit (update_ed).  Outputs:
 * idx (Nq,K) int32, dist (Nq,K) float32 = sqrt(d2). */
int slm_knn(int32_t Nq, int32_t Nn, int32_t K, int32_t skip_self, const float* query_points,
            const float* node_points, int32_t* idx_out, float* dist_out, void* stream);
/* softmax(exp(-dist/radius)) weights (super/nodes.py:166,191) and the stability
 * test any(dist <= radius) (super/nodes.py:182).  radius_mode 0: radius of each
 * neighbour node (surfels); 1: radius of the query node itself (node-node).
 * stable_io (Nq) uint8 is AND-ed in place, may be NULL. */
int slm_knn_weights(int32_t Nq, int32_t K, int32_t radius_mode, const int32_t* idx,
                    const float* dist, const float* node_radii, float* w_out,
                    uint8_t* stable_io, void* stream);

/* float64 feeder with the Semantic-SuPer branches (the reference's tensors are float64):
 * q_seg / node_seg (both or neither): a query only sees the nodes of its own class -- find_knn with
 * num_classes, utils/utils.py:223-242, used by update_ed / update_sfed_knn under hard_seg
 * (super/nodes.py:157-160,172-175); a class with fewer nodes than asked for is an error like the
 * reference's assert (the call synchronises `stream` in that mode).  dist (Nq,K) float64. */
int slm_knn_f64(int32_t Nq, int32_t Nn, int32_t K, int32_t skip_self, const double* query_points,
                const double* node_points, const int32_t* q_seg, const int32_t* node_seg, int32_t* idx_out,
                double* dist_out, void* stream);
/* As slm_knn_weights in float64; num_classes > 0 with q_seg_conf (Nq,C) and node_seg_conf (Nn,C):
 * softmax(exp(-JSD(node, query))^(1/2) * exp(-dist/radius)^(1/2)), super/nodes.py:183-189. */
int slm_knn_weights_f64(int32_t Nq, int32_t K, int32_t radius_mode, const int32_t* idx, const double* dist,
                        const double* node_radii, int32_t num_classes, const double* q_seg_conf,
                        const double* node_seg_conf, double* w_out, uint8_t* stable_io, void* stream);

/* =====================================================================================
 * The reference's DEFAULT per-frame optimiser (no --use_derived_gradient): GraphFit,
 * autograd + SGD(momentum 0.9) / Adam over (J+1,7) rows, last row = global transform T_g.
 *   slm_gf_create / destroy   <- GraphFit.__init__                  super/deform_mesh.py:11-17
 *   slm_gf_bind_frame         <- arguments of GraphFit.forward      super/deform_mesh.py:232-247
 *   slm_gf_run                <- GraphFit.deform_superedg           super/deform_mesh.py:251-379
 *   slm_gf_loss_grad          <- deform_source + get_losses + backward (one evaluation)
 *                                                                    super/deform_mesh.py:25-230
 *   slm_apply_update_gf       <- Surfels.update, autograd variant   super/nodes.py:193-223
 * The gradients that the reference obtains by autograd are hand-derived here (2 J^T r per
 * term, chained through the global row); results are checked against the autograd oracle.
 * ===================================================================================== */
typedef struct slm_gf slm_gf; /* opaque */

typedef struct slm_gf_config {
  int32_t num_iterations;  /* opt.num_optimize_iterations (10) */
  int32_t optimizer;       /* 0 = "SGD" (momentum 0.9, the reference default), 1 = "Adam" */
  int32_t use_data;        /* opt.sf_point_plane */
  int32_t use_arap;        /* opt.mesh_arap (weighted by the node KNN weights on this path) */
  int32_t use_rot;         /* opt.mesh_rot (all J+1 rows) */
  int32_t use_face;        /* opt.mesh_face */
  int32_t max_frames;
  int32_t seg_mode;        /* point-plane semantic weight: 0 none, 1 opt.sf_hard_seg_point_plane,
                              2 opt.sf_soft_seg_point_plane (either implies the point-plane term,
                              super/deform_mesh.py:76-92; needs slm_gf_bind_semantic) */
  int32_t use_bn_morph;    /* opt.sf_bn_morph (needs slm_gf_bind_semantic) */
  int32_t corr_mode;       /* opt.sf_corr: 0 off, 1 opt.sf_corr_loss_type 'point-point', 2 'point-plane'
                              (super/deform_mesh.py:100-109; needs slm_gf_bind_flow) */
  double w_data, w_arap, w_rot, w_face; /* opt.*_weight */
  double lr;               /* opt.learning_rate (5e-5) */
  double w_bn_morph;       /* opt.sf_bn_morph_weight (0.1) */
  double pp_max;           /* > 0: drop squared point-plane residuals >= pp_max (2e-5 when
                              opt.depth_model == "raft_stereo", super/deform_mesh.py:97;
                              ignored with seg_mode != 0 like the reference) */
  double w_corr;           /* opt.sf_corr_weight (0.001) */
} slm_gf_config;
#define SLM_GF_NTERMS 10   /* doubles in the loss-term block of slm_gf_loss_grad / slm_gf_get_partial */

typedef struct slm_gf_frame {
  slm_frame base;                 /* same fields as the LM path (tgt_valid is not read here) */
  const uint8_t* sf_stable;       /* device (N) sf.isStable, or NULL = all stable */
  const void* ed_knn_w;           /* device (J,K_ED) sf.ED_nodes.knn_w   float32, or float64 with base.state_f64 */
  const int32_t* ed_triangles;    /* device (3,Tr) sf.ED_nodes.triangles, or NULL */
  const void* ed_triangle_areas;  /* device (Tr)   sf.ED_nodes.triangles_areas   float32 / float64 */
  int32_t n_triangles;
  int32_t pad;
} slm_gf_frame;

#define SLM_MAX_CLASSES 4
/* Semantic-SuPer inputs of one frame (super/deform_mesh.py:76-92,126-194, super/loss.py:346-399). */
typedef struct slm_gf_semantic {
  int32_t num_classes;        /* opt.num_classes, 1..SLM_MAX_CLASSES */
  int32_t pad;
  const int32_t* sf_seg;      /* device (N)     src.seg        (indexed like sf_points) */
  const float* sf_seg_conf;   /* device (N,C)   src.seg_conf */
  const float* tgt_seg_conf;  /* device (T,C)   trg.seg_conf */
  const float* img_seg_conf;  /* device (C,H,W) inputs[("seg_conf",0)][0] */
  const int32_t* img_seg;     /* device (H,W)   inputs[("seg",0)][0,0] */
} slm_gf_semantic;

int slm_gf_create(const slm_gf_config* cfg, slm_gf** out);
int slm_gf_destroy(slm_gf* g);
/* Binds device pointers to `slot` and resets deform_verts to identity, optimiser state to 0. */
int slm_gf_bind_frame(slm_gf* g, int32_t slot, const slm_gf_frame* frame, void* stream);
/* After slm_gf_bind_frame: binds the semantic inputs of the slot and extracts, per class, the
 * class-boundary pixels of img_seg in row-major order (find_edge_region with kernel 3 +
 * margin test, utils/utils.py:276-301, super/deform_mesh.py:149-165).  edge_counts_host, if not
 * NULL, receives num_classes counts (synchronises `stream`). */
int slm_gf_bind_semantic(slm_gf* g, int32_t slot, const slm_gf_semantic* sem, int32_t* edge_counts_host,
                         void* stream);
/* After slm_gf_bind_frame: binds the optical flow of the frame for the surfel-correspondence term (corr_mode):
 * flow_device is (2,H,W) float32, channel 0 the x (u) and channel 1 the y (v) displacement in pixels -- the output
 * of the reference's models.optical_flow(src.rgb, inputs[("color",0)]) (super/deform_mesh.py:286-320), which stays
 * with the caller.  The term samples it at each surfel's unrounded projection exactly like
 * F.grid_sample(flow, grid.float()) (bilinear, zero padding, align_corners=False; super/loss.py:318-323), moves
 * the projection by it and evaluates the 4-tap residual there; the gradient includes d(flow)/d(u,v). */
int slm_gf_bind_flow(slm_gf* g, int32_t slot, const float* flow_device, void* stream);
/* Copies the slot's boundary pixels of `class_id` ((x,y) float pairs, row-major pixel order) to
 * device memory `xy_out_device` (capacity max_points pairs). */
int slm_gf_get_edge_points(slm_gf* g, int32_t slot, int32_t class_id, float* xy_out_device,
                           int32_t max_points, void* stream);
/* num_iterations optimiser steps for slots [0,n_frames), entirely on the device. */
int slm_gf_run(slm_gf* g, int32_t n_frames, void* stream);

/* -- one large frame sharded over the GPUs of a node (BASELINE configs[4]) --
 * After slm_gf_set_shard(rank, world) (before slm_gf_bind_frame) this context evaluates the
 * surfels [N*rank/world, N*(rank+1)/world) of every slot; rank 0 also evaluates the node terms
 * (ARAP / Rot / face).  The library holds no communicator: per optimiser iteration the caller runs
 *     slm_gf_eval_morph   -> all-reduce(sum) of the partial state   (only with use_bn_morph: the
 *                            global kept count scales the back-propagation)
 *     slm_gf_eval_losses  -> all-reduce(sum) of the partial state   (gradient + loss terms)
 *     slm_gf_step
 * on every rank; the partial state is [(J+1)*7 gradient | SLM_GF_NTERMS terms] doubles, moved with
 * slm_gf_get_partial / slm_gf_set_partial (device to device) so that the collective runs on the
 * caller's own buffer (torch.distributed all_reduce = RCCL over xGMI). */
int slm_gf_set_shard(slm_gf* g, int32_t rank, int32_t world);
int slm_gf_eval_morph(slm_gf* g, int32_t n_frames, void* stream);
int slm_gf_eval_losses(slm_gf* g, int32_t n_frames, void* stream);
int slm_gf_step(slm_gf* g, int32_t n_frames, void* stream);
int slm_gf_get_partial(slm_gf* g, int32_t slot, double* out_device, void* stream);
int slm_gf_set_partial(slm_gf* g, int32_t slot, const double* in_device, void* stream);
/* Copies deform_verts ((J+1)*7 doubles) of the slot into caller device memory. */
int slm_gf_get_deform(slm_gf* g, int32_t slot, double* out_device, void* stream);
/* One loss + gradient evaluation at dv_device ((J+1)*7): terms_device[SLM_GF_NTERMS]: [0..3] = face, arap,
 * rot, point_plane losses (already weighted), [4] = point-plane residuals kept, [5] = boundary
 * morphing loss (weighted; NaN when candidates exist but none passes the > 15 test, like the
 * reference's mean over an empty tensor), [6] = surfels kept by the morphing term,
 * [7] = 1 when some class contributed to it (the loss key exists in the reference),
 * [8] = flow-correspondence loss (weighted), [9] = its residuals kept;
 * grad_device ((J+1)*7) = d(sum)/d(dv) with the global row divided by J like the reference. */
int slm_gf_loss_grad(slm_gf* g, int32_t slot, const double* dv_device, double* terms_device,
                     double* grad_device, void* stream);
/* Surfels.update for this path: deform_device is (J+1,7); the global row's translation is
 * added to points / nodes and its rotation applied to the normals (super/nodes.py:204-222). */
int slm_apply_update_gf(int32_t N, int32_t J, int32_t K, float* sf_points, float* sf_norms,
                        const int32_t* sf_knn_idx, const float* sf_knn_w, float* ed_points,
                        float* ed_norms, const double* deform_device, void* stream);
int slm_apply_update_gf_f64(int32_t N, int32_t J, int32_t K, double* sf_points, double* sf_norms,
                            const int32_t* sf_knn_idx, const double* sf_knn_w, double* ed_points,
                            double* ed_norms, const double* deform_device, void* stream);

/* ===================================================================================
 * "Next" row f2: depth map -> per-frame target `new_data`
 *   slm_depth_*  <- depth_preprocessing  utils/data_loader.py:333-523 (+ getN :532-584,
 *                   BackprojectDepth depth/monodepth2/layers.py:139-167, torch_dilate /
 *                   find_edge_region utils/utils.py:152-157,276-301)
 * Produces exactly what slm_bind_frame consumes (tgt_points, tgt_norms, index_map, tgt_valid)
 * plus the other fields of the reference's Data object.  The optional SSIM confidence
 * (opt.disable_ssim_conf == False, needs the stereo pair) is not built.
 * =================================================================================== */
typedef struct slm_depth slm_depth; /* opaque: scratch buffers for one image size */

typedef struct slm_depth_config {
  int32_t H, W;                  /* opt.height, opt.width */
  int32_t data_mode;             /* 0 = opt.data "superv1", 1 = "superv2" */
  int32_t raft_stereo;           /* opt.depth_model == "raft_stereo" (superv1 rules) */
  int32_t dilate_invalid_kernel; /* opt.dilate_invalid_kernel (superv1) */
  int32_t load_depth;            /* opt.load_depth (superv2) */
  int32_t normal_model;          /* 0 = "naive", 1 = "8neighbors" */
  int32_t num_classes;           /* opt.num_classes (with segmentation inputs), <= SLM_MAX_CLASSES */
  int32_t n_del_classes;         /* len(opt.del_seg_classes) <= 3 */
  int32_t del_classes[3];
  float depth_width_range[2];    /* opt.depth_width_range (superv2 without load_depth) */
  float inv_K[9];                /* inputs["inv_K"][0,:3,:3], row-major */
  float fx, fy, cx, cy;          /* inputs["K"][0] */
  double divterm;                /* inputs["divterm"] */
  int32_t use_ssim_conf;         /* hasattr(opt, "disable_ssim_conf") and not opt.disable_ssim_conf (the CLI default):
                                    confs = 0.5 * confs + 0.5 * sigmoid(stereo SSIM confidence), data_loader.py:359-373,477-479 */
  float stereo_P[12];            /* torch.matmul(inputs["K"], inputs["stereo_T"])[0,:3,:] row-major (float32) */
} slm_depth_config;

typedef struct slm_depth_inputs {   /* device pointers */
  const float* depth;            /* (H,W)   inputs[("depth",0)][0,0] */
  const float* color;            /* (3,H,W) inputs[("color",0)][0] */
  const uint8_t* valid_mask;     /* (H,W) or NULL: the opt.load_valid_mask image (superv1) */
  const int32_t* seg;            /* (H,W) or NULL: inputs[("seg",0)][0,0] */
  const float* seg_conf;         /* (C,H,W) or NULL: inputs[("seg_conf",0)][0] */
} slm_depth_inputs;

typedef struct slm_depth_outputs {  /* device pointers; row capacity H*W; NULL = not wanted */
  float* points;                 /* (T,3) data.points (the reference widens these float32 values to float64) */
  float* norms;                  /* (T,3) data.norms */
  float* colors;                 /* (T,3) data.colors */
  double* radii;                 /* (T)   data.radii */
  float* confs;                  /* (T)   data.confs */
  int32_t* index_map;            /* (H,W) data.index_map, -1 invalid */
  uint8_t* valid;                /* (H*W) data.valid */
  int32_t* seg;                  /* (T)   data.seg */
  double* seg_conf;              /* (T,C) data.seg_conf (per-pixel softmax) */
  double* dist2edge;             /* (T)   data.dist2edge */
  uint8_t* inval;                /* (H*W) the invalid-pixel map of step 1 (the reference NaNs depth / disp / pcd there) */
  float* disp_conf;              /* (H,W) inputs[("disp_conf",0)] with use_ssim_conf: SSIM between the image and its
                                    warp through the stereo transform, mean over the channels */
} slm_depth_outputs;

int slm_depth_create(int32_t H, int32_t W, slm_depth** out);
int slm_depth_destroy(slm_depth* d);
/* Runs the whole step on `stream`; *n_valid_host receives T (synchronises the stream). */
int slm_depth_preprocess(slm_depth* d, const slm_depth_config* cfg, const slm_depth_inputs* in,
                         const slm_depth_outputs* out, int32_t* n_valid_host, void* stream);

/* ===================================================================================
 * "Next" row f1: surfel fusion, the step right after the solve every frame
 *   slm_fuse_input_data   <- Surfels.fuseInputData                     super/nodes.py:268-541
 *   slm_fuse_swap_stable  <- Surfels.prepareStableIndexNSwapAllModel   super/nodes.py:543-585
 * for opt.method == "super" and "semantic-super" (slm_fuse_bind_semantic).  Geometry is float64 like the
 * reference's tensors, colours / confidences / time stamps float32.  Unpinned by the reference:
 * the order of surfels with EQUAL confidence on one pixel (torch.sort(descending=True) is not
 * stable); here the lower index comes first.
 * =================================================================================== */
typedef struct slm_fuse slm_fuse; /* opaque: sort / layer-map / compaction scratch */

typedef struct slm_fuse_config {
  int32_t H, W;                   /* opt.height, opt.width */
  int32_t merge_new;              /* !opt.disable_merging_new_surfels */
  int32_t merge_exist;            /* !opt.disable_merging_exist_surfels */
  int32_t add_new;                /* !opt.disable_adding_new_surfels */
  int32_t remove_unstable;        /* !opt.disable_removing_unstable_surfels */
  int32_t phase_test;             /* opt.phase == "test": merged surfels get the frame's time stamp */
  int32_t th_time_steps;          /* opt.th_time_steps (30) */
  double th_dist;                 /* opt.th_dist (0.1) */
  double th_cosine_ang;           /* opt.th_cosine_ang (0.4) */
  float fx, fy, cx, cy;           /* inputs["K"][0] */
} slm_fuse_config;

typedef struct slm_surfel_model {  /* device pointers with room for `cap` rows; n rows in use */
  int32_t n, cap;
  double* points;                 /* (cap,3) sf.points */
  double* norms;                  /* (cap,3) sf.norms */
  float* colors;                  /* (cap,3) sf.colors */
  double* radii;                  /* (cap)   sf.radii */
  float* confs;                   /* (cap)   sf.confs */
  float* time_stamp;              /* (cap)   sf.time_stamp */
  uint8_t* is_stable;             /* (cap)   sf.isStable */
  int32_t* knn_idx;               /* (cap,K) sf.knn_indices */
  double* knn_w;                  /* (cap,K) sf.knn_w */
  float* projdata;                /* (cap,2) sf.projdata */
  int32_t J;
  int32_t K;                      /* opt.num_neighbors, 1..8; 0 is read as 4 (the field was zero padding before round 6: same struct size,
                                     same SLM_ABI_VERSION -- a caller that leaves it 0 gets the former behaviour) */
  const double* ed_points;        /* (J,3) sf.ED_nodes.points */
  const double* ed_radii;         /* (J)   sf.ED_nodes.radii */
  int32_t* merged_into;           /* (cap) or NULL.  slm_fuse_input_data: the surfel that absorbed row i
                                     (-1: none) -- what the reference uses to re-point its tracked
                                     evaluation ids, super/nodes.py:440-445 */
} slm_surfel_model;

typedef struct slm_new_frame {     /* sfdata from depth_preprocessing, device pointers */
  int32_t T, time;                /* rows, sfdata.time */
  const double* points;           /* (T,3) */
  const double* norms;            /* (T,3) */
  const float* colors;            /* (T,3) */
  const double* radii;            /* (T) */
  const float* confs;             /* (T) */
  const uint8_t* valid;           /* (H*W) */
  const int32_t* index_map;       /* (H,W), -1 invalid */
} slm_new_frame;

/* Segmentation fields of Semantic-SuPer (sf.seg / sf.seg_conf / sf.dist2edge exist iff the frames carry
 * them, super/nodes.py:60-75): fused in merge_data (nodes.py:348-353), appended with the new surfels
 * (nodes.py:524-525), compacted by the swap (nodes.py:572-575).  All device pointers. */
typedef struct slm_fuse_semantic {
  int32_t num_classes;            /* C, 1..SLM_MAX_CLASSES */
  int32_t soft_weights;           /* opt.method == "semantic-super": skinning weights
                                     softmax(exp(-JSD(node, surfel))^(1/2) * exp(-d/r)^(1/2)), nodes.py:467-477;
                                     for the NEW surfels only without hard_seg (nodes.py:503-509) */
  int32_t hard_seg;               /* sf.hard_seg: new surfels take their 4 nodes among the nodes of their own
                                     class (find_knn with num_classes, utils/utils.py:223-242); a class with
                                     fewer than 4 nodes is an error like the reference's assert */
  int32_t merge_same_class;       /* hard_seg or opt.data == "superv1": only equal classes merge (nodes.py:314-316) */
  int32_t* seg;                   /* (cap)   sf.seg */
  double* seg_conf;               /* (cap,C) sf.seg_conf */
  double* dist2edge;              /* (cap)   sf.dist2edge */
  const int32_t* ed_seg;          /* (J)     sf.ED_nodes.seg      (hard_seg) */
  const double* ed_seg_conf;      /* (J,C)   sf.ED_nodes.seg_conf (soft_weights) */
  const int32_t* new_seg;         /* (T)     sfdata.seg */
  const double* new_seg_conf;     /* (T,C)   sfdata.seg_conf */
  const double* new_dist2edge;    /* (T)     sfdata.dist2edge */
} slm_fuse_semantic;

int slm_fuse_create(int32_t H, int32_t W, int32_t max_surfels, slm_fuse** out);
/* Binds (sem != NULL) or drops (NULL) the segmentation fields used by the next slm_fuse_input_data /
 * slm_fuse_swap_stable calls on `f`; the struct is copied.  The swap only needs seg / seg_conf / dist2edge. */
int slm_fuse_bind_semantic(slm_fuse* f, const slm_fuse_semantic* sem);
int slm_fuse_destroy(slm_fuse* f);
/* Fuses the frame into the model in place; model->n (host struct) becomes the new row count.
 * Synchronises `stream` (three count read-backs). */
int slm_fuse_input_data(slm_fuse* f, const slm_fuse_config* cfg, slm_surfel_model* model,
                        const slm_new_frame* frame, void* stream);
/* Drops unstable / stale surfels (time = inputs["time"]); model->n becomes the new row count.
 * keep_ids (device, n_keep entries, may be NULL): rows kept regardless (the tracked evaluation
 * points, super/nodes.py:559-560); new_index (device, old row count entries, may be NULL): new row
 * of every old row or -1 (the reference's id_map, super/nodes.py:578-581). */
int slm_fuse_swap_stable(slm_fuse* f, const slm_fuse_config* cfg, slm_surfel_model* model, int32_t time,
                         const int32_t* keep_ids, int32_t n_keep, int32_t* new_index, void* stream);

/* ===================================================================================
 * "Next" row f3: ED-graph construction at frame 0
 *   slm_graph_init  <- init_graph + DirectDeformGraph (grid_mesh)   super/graph_encoder.py:11-67,128-195
 * Anchors on a pixel grid of step opt.mesh_step_size (row-major numbering), four edges and two
 * triangles per grid cell (kept when all their vertices are valid anchors), node radius = mean
 * length of the incident edges (isolated nodes get the mean of the others), triangle rest areas.
 * Capacities: nodes <= ceil((H-1)/step) * ceil((W-1)/step), edges <= 4 * nodes, triangles <= 2 * nodes;
 * edge_index / triangles are written row by row with row stride `cap_nodes * 4` / `cap_nodes * 2`.
 * =================================================================================== */
typedef struct slm_graph_outputs {  /* device pointers */
  int32_t cap_nodes, pad;
  double* points;                 /* (cap_nodes,3) graph.points */
  double* norms;                  /* (cap_nodes,3) graph.norms */
  double* radii;                  /* (cap_nodes)   graph.radii */
  int32_t* edge_index;            /* (2, 4*cap_nodes) graph.edge_index */
  double* edges_lens;             /* (4*cap_nodes)    graph.edges_lens */
  int32_t* triangles;             /* (3, 2*cap_nodes) graph.triangles */
  double* triangles_areas;        /* (2*cap_nodes)    graph.triangles_areas */
} slm_graph_outputs;

/* valid (H*W) = data.valid, index_map (H,W) = data.index_map, points / norms (T,3) = data.points /
 * data.norms.  counts_host[3] = {nodes, edges, triangles}.  Synchronises `stream`. */
int slm_graph_init(int32_t H, int32_t W, int32_t step, const uint8_t* valid, const int32_t* index_map,
                   const double* points, const double* norms, const slm_graph_outputs* out,
                   int32_t* counts_host, void* stream);

/* Semantic-SuPer variant (super/graph_encoder.py:134-151,190-192): seg_conf (T,C) = data.seg_conf; every
 * node also gets node_seg_conf (cap_nodes,C) = seg_conf of its row and node_seg (cap_nodes) = first maximum;
 * prune_class_edges (opt.hard_seg and opt.mesh_face): edges / triangles whose vertices differ in class are
 * dropped before lengths, radii and areas are computed. */
int slm_graph_init_semantic(int32_t H, int32_t W, int32_t step, const uint8_t* valid, const int32_t* index_map,
                            const double* points, const double* norms, int32_t num_classes, const double* seg_conf,
                            int32_t prune_class_edges, const slm_graph_outputs* out, int32_t* node_seg,
                            double* node_seg_conf, int32_t* counts_host, void* stream);

/* Diagnostics: {device buffer reallocations in the LM solver, bytes asked for, symbolic analyses,
 * plan reuses with a changed pair list} since the library was loaded. */
/* Form of the numeric phase the last slm_run / slm_lm_solve / slm_solve of this solver enqueued: 0 = per-level
 * launches, 1 = task graph (one persistent launch), 2 = hybrid (per-level launches below, task graph for the top of
 * the tree), -1 = none yet (block-banded path or nothing run). */
int slm_debug_last_solver_form(slm_solver* s);
/* Mode word of the last task-graph launch (k_fdag) this solver enqueued: bit 0 = XCD-affine ticket streams (a multiple of
 * 8 frames per launch on a device whose launches land on the XCD ids 0..7); -1 = the last solve ran no task graph. */
int slm_debug_last_dag_mode(slm_solver* s);
int slm_debug_counters(int64_t out[4]);
/* Diagnostics: copies a solver work buffer of the slot to HOST memory (synchronises `stream`): what = 0 front
 * tiles, 1 front vectors, 2 inverses of the diagonal factor blocks, 3 delta.  *n_doubles receives the buffer's
 * length; at most max_doubles are copied. */
/* Diagnostics of the task-graph solver (solver_path 2): with on != 0 every task of the slot records
 * {start, last dependency ready, end} (10 ns ticks) and its workgroup into a trace that slm_debug_read returns
 * as what = 4 (24 int64 per task, reinterpret the doubles) and what = 5 (the task words, 2 int32 per task; what = 6: the
 * task words of the top-of-tree list the hybrid form runs, which the trace is indexed by after a hybrid solve). */
int slm_debug_dag_trace(slm_solver* s, int32_t slot, int32_t on, void* stream);
/* Diagnostics / tests: the deadline of a wait inside the task-graph solver, in ticks of the 100 MHz wall clock, for every
 * solver of the process on the current device (0 restores the default of 3 s).  A wait that exceeds it aborts the launch;
 * the slots whose solve did not finish record SLM_ITER_SOLVER_TIMEOUT and stop like after a failed factorisation. */
int slm_debug_dag_timeout(int64_t ticks);
/* Tests: leaves slots [0, n_frames) in the state an ABORTED task-graph launch leaves behind (flags reset, abort flag up,
 * no front has published its solution), runs the check every launch ends with and the accept step: every slot must
 * record SLM_ITER_SOLVER_TIMEOUT for its current iteration and stop (also the slots past the first 64 of a batch). */
int slm_debug_dag_abort(slm_solver* s, int32_t n_frames, void* stream);
int slm_debug_read(slm_solver* s, int32_t slot, int32_t what, double* host_out, int64_t max_doubles,
                   int64_t* n_doubles, void* stream);
/* Diagnostics / tests: copies one array of the slot's per-frame assembly plan (the output of the preparation, slm_prep.hip) to
 * HOST memory as raw bytes (synchronises `stream`): what = 0 s_idx, 1 grp_run, 2 run_nodes, 3 s_w, 4 blk_key, 5 blk_start,
 * 6 blk_entry, 7 wg_first, 8 wg_last, 9 run_lidx, 10 blk2_start, 11 blk2_entry, 12 s_pts.  *n_bytes receives the array's
 * length; at most max_bytes are copied.  (tests/test_gpu_prepare_binned.py: the binned preparation against the rocPRIM
 * pipeline, array by array.) */
int slm_debug_read_plan(slm_solver* s, int32_t slot, int32_t what, void* host_out, int64_t max_bytes, int64_t* n_bytes,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SUPER_LM_H */
