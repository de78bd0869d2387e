// slm_fuse.hip -- "next" row f1: surfel fusion after the solve (reference Surfels.fuseInputData and
// prepareStableIndexNSwapAllModel, super/nodes.py:268-585, opt.method == "super" / "semantic-super").
//
//   k_fu_keys        project every surfel (pcd2depth, rounded), key = (pixel << 32) | ~confidence
//   rocPRIM sort     stable radix sort by key: per pixel the surfels in descending confidence
//   k_fu_layers      rank of a surfel inside its pixel (look-back <= 16) -> 16 layer maps
//   k_fu_merge_new   one thread per pixel: the frame's point is fused into the first layer that passes
//                    the distance / normal test, otherwise it becomes a candidate new surfel
//   k_fu_merge_exist one thread per pixel: surfels sharing the pixel are fused pairwise in the
//                    reference's (i, j) order, the absorbed ones are dropped
//   k_fu_weights     skinning weights of every surfel at its fused position
//   k_fu_candidates  4 nearest ED nodes + stability test of the candidate pixels
//   rocPRIM scan     positions of the accepted candidates (row-major pixel order = sfdata order)
//   k_fu_append      new rows;  k_fu_proj  float pixel coordinates of every surfel
//   k_fu_keep / scan / k_fu_compact    drop unstable and stale surfels (swap)
// Everything per pixel / per surfel is independent, so the reference's vectorised tensor
// statements and these per-thread loops produce the same values (float32 confidence / colour
// arithmetic and float64 geometry are kept operation by operation).
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <string>

#include "slm_common.h"

void slm_set_error_text(const char* msg);   // slm_api.hip

#define FU_LAYERS 16

struct slm_fuse {
  int H = 0, W = 0, cap = 0;
  unsigned long long *keys = nullptr, *skeys = nullptr;
  int32_t *ids = nullptr, *sids = nullptr;
  int32_t* layers = nullptr;    // (FU_LAYERS, H*W) surfel id or -1
  int32_t* flag = nullptr;      // per pixel / per surfel flags for the scans
  int32_t* pos = nullptr;       // exclusive scan of flag
  int32_t* cand_idx = nullptr;  // (H*W,4) nearest nodes of the candidate pixels
  double* cand_w = nullptr;     // (H*W,4)
  uint8_t* dead = nullptr;      // (cap) surfels to drop after the pairwise merges
  int32_t* counters = nullptr;  // [0] layers in use, [1] candidates without 4 nodes of their class, [2] set flags (k_fu_total)
  int32_t* h_counters = nullptr;   // pinned host copy of counters: one small D2H read-back per call
  double* boxes = nullptr;         // (ceil(J / FU_RUN), 6) bounding boxes of the node runs (candidate search)
  size_t cap_boxes = 0;
  slm_fuse_semantic sem{};      // segmentation fields (num_classes == 0: none)
  int32_t* s_seg = nullptr;     // compaction scratch of the segmentation fields
  double *s_sc = nullptr, *s_d2e = nullptr;
  void* tmp = nullptr;
  size_t cap_tmp = 0;
  // compaction scratch (swap)
  double *s_d3 = nullptr, *s_d1 = nullptr, *s_d4 = nullptr;
  float *s_f3 = nullptr, *s_f1 = nullptr, *s_f2 = nullptr;
  int32_t* s_i4 = nullptr;
};

namespace {

#define FCHK(expr)                                                        \
  do {                                                                    \
    hipError_t e_ = (expr);                                               \
    if (e_ != hipSuccess) {                                               \
      slm_set_error_text((std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
      return SLM_ERR_HIP;                                                 \
    }                                                                     \
  } while (0)

int ffail(int code, const char* msg) {
  slm_set_error_text(msg);
  return code;
}

__device__ __forceinline__ void fu_project(const slm_fuse_config& c, const double* p, double& u_, double& v_) {
  const double Z = p[2] + 1e-8;
  u_ = p[0] * (double)c.fx / Z + (double)c.cx;
  v_ = p[1] * (double)c.fy / Z + (double)c.cy;
}

// key of surfel i: (pixel << 32) | (0xFFFFFFFF - orderable(confidence)); unprojectable / unstable: ~0
__global__ void __launch_bounds__(256) k_fu_keys(slm_fuse_config c, slm_surfel_model m, unsigned long long* __restrict__ keys,
                                                  int32_t* __restrict__ ids) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.n) return;
  double u_, v_;
  fu_project(c, m.points + 3 * (size_t)i, u_, v_);
  const double ur = rint(u_), vr = rint(v_);
  const bool ok = m.is_stable[i] && vr >= 0.0 && vr < (double)(c.H - 1) && ur >= 0.0 && ur < (double)(c.W - 1);
  unsigned long long key = ~0ull;
  if (ok) {
    const unsigned pix = (unsigned)((int)vr * c.W + (int)ur);
    unsigned b = __float_as_uint(m.confs[i]);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);       // monotone map float -> uint
    key = ((unsigned long long)pix << 32) | (unsigned long long)(0xFFFFFFFFu - b);
  }
  keys[i] = key;
  ids[i] = i;
}

// sorted element e: its rank inside the pixel (number of equal-pixel predecessors, <= 16 looked at)
__global__ void __launch_bounds__(256) k_fu_layers(int n, int HW, const unsigned long long* __restrict__ skeys,
                                                    const int32_t* __restrict__ sids, int32_t* __restrict__ layers,
                                                    uint8_t* __restrict__ dead, int32_t* __restrict__ counters) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  int used = 0;   // layer maps this element needs
  if (e < n) {
    const unsigned long long k = skeys[e];
    if (k != ~0ull) {
      const unsigned pix = (unsigned)(k >> 32);
      int rank = 0;
      while (rank < FU_LAYERS && e - rank - 1 >= 0 && (unsigned)(skeys[e - rank - 1] >> 32) == pix) ++rank;
      if (rank < FU_LAYERS) {
        layers[(size_t)rank * HW + pix] = sids[e];
        used = rank + 1;
      } else {
        dead[sids[e]] = 1;    // beyond the 16 maps: dropped when surfels are merged (nodes.py:390-391)
      }
    }
  }
  // one atomic per wavefront instead of one per surfel
  for (int off = 32; off >= 1; off >>= 1) used = max(used, __shfl_xor(used, off));
  if ((threadIdx.x & 63) == 0 && used > 0) atomicMax(&counters[0], used);
}

struct FuRow {
  double p[3], n[3], r;
  float c[3], w;
  int seg;
  double sc[SLM_MAX_CLASSES];
};

__device__ __forceinline__ FuRow fu_load(const slm_surfel_model& m, const slm_fuse_semantic& sm, int i) {
  FuRow x;
  x.seg = 0;
  if (sm.num_classes > 0) {
    x.seg = sm.seg[i];
    for (int k = 0; k < sm.num_classes; ++k) x.sc[k] = sm.seg_conf[(size_t)sm.num_classes * i + k];
  }
  for (int k = 0; k < 3; ++k) {
    x.p[k] = m.points[3 * (size_t)i + k];
    x.n[k] = m.norms[3 * (size_t)i + k];
    x.c[k] = m.colors[3 * (size_t)i + k];
  }
  x.r = m.radii[i];
  x.w = m.confs[i];
  return x;
}

// merge_data (nodes.py:296-357) for one pair: `a` is the surfel that stays (row ia of the model)
__device__ __forceinline__ bool fu_merge(const slm_fuse_config& c, const slm_surfel_model& m,
                                         const slm_fuse_semantic& sm, int ia, const FuRow& a, const FuRow& b,
                                         bool add_new, int time) {
#pragma clang fp contract(off)
  const double dx = a.p[0] - b.p[0], dy = a.p[1] - b.p[1], dz = a.p[2] - b.p[2];
  const double dist = sqrt(dx * dx + dy * dy + dz * dz);
  const double cosang = a.n[0] * b.n[0] + a.n[1] * b.n[1] + a.n[2] * b.n[2];
  if (!(dist < c.th_dist && cosang > c.th_cosine_ang)) return false;
  if (sm.num_classes > 0 && sm.merge_same_class && a.seg != b.seg) return false;
  const float wu = a.w + b.w;
  const float w = a.w / wu, w2 = b.w / wu;
  const double wd = (double)w, w2d = (double)w2;
  m.radii[ia] = wd * a.r + w2d * b.r;
  m.confs[ia] = wu;
  double nn[3], s = 0.0;
  for (int k = 0; k < 3; ++k) {
    m.points[3 * (size_t)ia + k] = wd * a.p[k] + w2d * b.p[k];
    nn[k] = wd * a.n[k] + w2d * b.n[k];
    s += nn[k] * nn[k];
  }
  const double den = fmax(sqrt(s), 1e-12);
  for (int k = 0; k < 3; ++k) m.norms[3 * (size_t)ia + k] = nn[k] / den;
  if (add_new) {
    const float wn = w2 * 3.0f, ws = w + wn;
    const float f1 = w / ws, f2 = wn / ws;
    for (int k = 0; k < 3; ++k) m.colors[3 * (size_t)ia + k] = f1 * a.c[k] + f2 * b.c[k];
  } else {
    for (int k = 0; k < 3; ++k) m.colors[3 * (size_t)ia + k] = w * a.c[k] + w2 * b.c[k];
  }
  if (c.phase_test) m.time_stamp[ia] = (float)time;
  if (sm.num_classes > 0) {
    // fused class confidences, renormalised; class = first maximum (nodes.py:348-353)
    double q[SLM_MAX_CLASSES], tot = 0.0;
    for (int k = 0; k < sm.num_classes; ++k) {
      q[k] = wd * a.sc[k] + w2d * b.sc[k];
      tot += q[k];
    }
    int best = 0;
    for (int k = 0; k < sm.num_classes; ++k) {
      q[k] /= tot;
      sm.seg_conf[(size_t)sm.num_classes * ia + k] = q[k];
      if (q[k] > q[best]) best = k;
    }
    sm.seg[ia] = best;
  }
  return true;
}

// one thread per pixel: new point -> first matching layer, else candidate (flag = 1)
__global__ void __launch_bounds__(256) k_fu_merge_new(slm_fuse_config c, slm_surfel_model m, slm_fuse_semantic sm,
                                                       slm_new_frame fr, const int32_t* __restrict__ layers,
                                                       const int32_t* __restrict__ n_layers_dev, int32_t* __restrict__ flag) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const int HW = c.H * c.W;
  if (pix >= HW) return;
  const int n_layers = *n_layers_dev;   // number of layer maps in use (k_fu_layers, earlier in the stream)
  int out = 0;
  if (fr.valid[pix]) {
    out = 1;
    if (c.merge_new && n_layers > 0) {
      const int t = fr.index_map[pix];
      FuRow b;
      for (int k = 0; k < 3; ++k) {
        b.p[k] = fr.points[3 * (size_t)t + k];
        b.n[k] = fr.norms[3 * (size_t)t + k];
        b.c[k] = fr.colors[3 * (size_t)t + k];
      }
      b.r = fr.radii[t];
      b.w = fr.confs[t];
      b.seg = 0;
      if (sm.num_classes > 0) {
        b.seg = sm.new_seg[t];
        for (int k = 0; k < sm.num_classes; ++k) b.sc[k] = sm.new_seg_conf[(size_t)sm.num_classes * t + k];
      }
      for (int l = 0; l < n_layers; ++l) {
        const int s = layers[(size_t)l * HW + pix];
        if (s < 0) break;
        const FuRow a = fu_load(m, sm, s);
        if (fu_merge(c, m, sm, s, a, b, true, fr.time)) {
          out = 0;
          break;
        }
      }
    } else {
      out = 0;      // merging disabled or no surfel projects anywhere: add_valid stays None (nodes.py:404,471)
    }
  }
  flag[pix] = out;
}

// one thread per pixel: pairwise fusion of the surfels that share the pixel (nodes.py:424-447)
__global__ void __launch_bounds__(256) k_fu_merge_exist(slm_fuse_config c, slm_surfel_model m, slm_fuse_semantic sm, int time,
                                                         const int32_t* __restrict__ layers,
                                                         const int32_t* __restrict__ n_layers_dev, uint8_t* __restrict__ dead) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const int HW = c.H * c.W;
  if (pix >= HW) return;
  const int n_layers = min(*n_layers_dev, FU_LAYERS);
  if (n_layers < 2) return;
  int id[FU_LAYERS];
  unsigned present = 0;
  for (int l = 0; l < n_layers; ++l) {
    id[l] = layers[(size_t)l * HW + pix];
    if (id[l] >= 0) present |= 1u << l;
  }
  if (!(present & 2u)) return;   // fewer than two surfels here
  for (int i = 0; i < n_layers; ++i) {
    bool alive = (present >> i) & 1u;          // val_maps[i], then ANDed with every val_maps[j] in turn
    for (int j = i + 1; j < n_layers && alive; ++j) {
      alive = alive && ((present >> j) & 1u);
      if (!alive) break;
      const FuRow a = fu_load(m, sm, id[i]), b = fu_load(m, sm, id[j]);
      if (fu_merge(c, m, sm, id[i], a, b, false, time)) {
        present &= ~(1u << j);                  // val_maps[j] loses the pixel for the later i loops
        dead[id[j]] = 1;
        if (m.merged_into) m.merged_into[id[j]] = id[i];
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_fu_apply_dead(int n, const uint8_t* __restrict__ dead, uint8_t* __restrict__ stable) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && dead[i]) stable[i] = 0;
}

// K = opt.num_neighbors (slm_surfel_model::K; the reference's find_knn / weights are K-generic: super/nodes.py:170-191,466-509)
#define FU_KMAX 8
__device__ __forceinline__ int fu_K(const slm_surfel_model& m) { return m.K > 0 ? m.K : 4; }
template <int KK>
__device__ __forceinline__ void fu_softmaxk(const double d[KK], const double r[KK], double w[KK]) {
  double e[KK], mx = -1e300, s = 0.0;
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    e[k] = exp(-d[k] / r[k]);
    mx = fmax(mx, e[k]);
  }
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    w[k] = exp(e[k] - mx);
    s += w[k];
  }
#pragma unroll
  for (int k = 0; k < KK; ++k) w[k] /= s;
}

// Jensen-Shannon divergence of two class distributions (utils/utils.py:244-254)
__device__ __forceinline__ double fu_jsd(const double* P, const double* Q, int C) {
  double a = 0.0, b = 0.0;
  for (int k = 0; k < C; ++k) {
    const double M = 0.5 * (P[k] + Q[k]);
    a += P[k] * log(P[k] / (M + 1e-13) + 1e-13);
    b += Q[k] * log(Q[k] / (M + 1e-13) + 1e-13);
  }
  return 0.5 * (a + b);
}

// softmax(exp(-JSD)^(1/2) * exp(-d/r)^(1/2)) (nodes.py:472-477, power_arg = (1/2, 1/2))
template <int KK>
__device__ __forceinline__ void fu_softmaxk_sem(const double d[KK], const double r[KK], const double js[KK], double w[KK]) {
  double e[KK], mx = -1e300, s = 0.0;
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    e[k] = sqrt(exp(-js[k])) * sqrt(exp(-d[k] / r[k]));
    mx = fmax(mx, e[k]);
  }
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    w[k] = exp(e[k] - mx);
    s += w[k];
  }
#pragma unroll
  for (int k = 0; k < KK; ++k) w[k] /= s;
}

// knn_w = softmax(exp(-dist / radius)) at the current positions (nodes.py:466-481)
template <int KK>
__global__ void __launch_bounds__(256) k_fu_weights(slm_surfel_model m, slm_fuse_semantic sm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.n) return;
  double d[KK], r[KK], w[KK];
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    const int j = m.knn_idx[KK * (size_t)i + k];
    double s = 0.0;
    for (int a = 0; a < 3; ++a) {
      const double t = m.points[3 * (size_t)i + a] - m.ed_points[3 * (size_t)j + a];
      s += t * t;
    }
    d[k] = sqrt(s);
    r[k] = m.ed_radii[j];
  }
  if (sm.num_classes > 0 && sm.soft_weights) {
    const int C = sm.num_classes;
    double js[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k)
      js[k] = fu_jsd(sm.ed_seg_conf + (size_t)C * m.knn_idx[KK * (size_t)i + k], sm.seg_conf + (size_t)C * i, C);
    fu_softmaxk_sem<KK>(d, r, js, w);
  } else {
    fu_softmaxk<KK>(d, r, w);
  }
#pragma unroll
  for (int k = 0; k < KK; ++k) m.knn_w[KK * (size_t)i + k] = w[k];
}

// Bounding boxes of the ED nodes in runs of FU_RUN consecutive indices (the node graph is a mesh grid in
// row-major order, so a run is a compact row segment; any other order only prunes less).  One wave per run.
#define FU_RUN 64
__global__ void __launch_bounds__(64) k_fu_node_boxes(int J, const double* __restrict__ ed_points, double* __restrict__ boxes) {
  const int b = blockIdx.x;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int j = b * FU_RUN + threadIdx.x; j < min((b + 1) * FU_RUN, J); j += 64)
    for (int k = 0; k < 3; ++k) {
      const double v = ed_points[3 * (size_t)j + k];
      lo[k] = fmin(lo[k], v);
      hi[k] = fmax(hi[k], v);
    }
  for (int off = 32; off >= 1; off >>= 1)
    for (int k = 0; k < 3; ++k) {
      lo[k] = fmin(lo[k], __shfl_xor(lo[k], off));
      hi[k] = fmax(hi[k], __shfl_xor(hi[k], off));
    }
  if (threadIdx.x < 3) boxes[6 * (size_t)b + threadIdx.x] = lo[threadIdx.x];
  else if (threadIdx.x < 6) boxes[6 * (size_t)b + threadIdx.x] = hi[threadIdx.x - 3];
}

// squared distance from p to the box (0 inside)
__device__ __forceinline__ double fu_box_d2(const double* __restrict__ bx, double px, double py, double pz) {
  const double dx = fmax(fmax(bx[0] - px, px - bx[3]), 0.0);
  const double dy = fmax(fmax(bx[1] - py, py - bx[4]), 0.0);
  const double dz = fmax(fmax(bx[2] - pz, pz - bx[5]), 0.0);
  return dx * dx + dy * dy + dz * dz;
}

// candidate pixels: 4 nearest nodes (squared L2 ascending, lowest index first), stability test.
// Exact search with pruning instead of a J-wide scan per pixel: every lane finds the run of nodes whose box is
// nearest to its point; those runs are scanned first (a tight 4th-best bound), then only the runs whose box is not
// farther than some lane's bound.  The top-4 is kept in lexicographic (distance, index) order, so the result is
// the one of the ascending brute-force scan whatever the visiting order.  The control flow is WAVE-UNIFORM -- a run
// is scanned by the whole wavefront when any of its lanes needs it (neighbouring pixels need the same runs; a
// lane that did not is not harmed by the extra insert tests) -- so node and box coordinates come through the
// scalar cache (s_load) instead of 64 identical vector loads per step.
// (branch-free shift for the generic K: the list lives in registers, a run-time index into it would not)
template <int KK>
__device__ __forceinline__ void fu_insert(double d2, int j, double bd[KK], int bi[KK]) {
  if constexpr (KK == 4) {
    if (d2 < bd[3] || (d2 == bd[3] && j < bi[3])) {
      int k = 3;
      while (k > 0 && (d2 < bd[k - 1] || (d2 == bd[k - 1] && j < bi[k - 1]))) {
        bd[k] = bd[k - 1];
        bi[k] = bi[k - 1];
        --k;
      }
      bd[k] = d2;
      bi[k] = j;
    }
  } else {
    if (d2 < bd[KK - 1] || (d2 == bd[KK - 1] && j < bi[KK - 1])) {
#pragma unroll
      for (int k = KK - 1; k >= 0; --k) {
        // entry k takes the new item when it sorts before the old entry k and not before entry k - 1; the old entry k - 1
        // when the new item sorts before that too
        const bool before_k = d2 < bd[k] || (d2 == bd[k] && j < bi[k]);
        const bool before_km1 = k > 0 && (d2 < bd[k - 1] || (d2 == bd[k - 1] && j < bi[k - 1]));
        if (before_km1) {
          bd[k] = bd[k - 1];
          bi[k] = bi[k - 1];
        } else if (before_k) {
          bd[k] = d2;
          bi[k] = j;
        }
      }
    }
  }
}

template <int KK>
__device__ __forceinline__ void fu_scan_run(const slm_surfel_model& m, const slm_fuse_semantic& sm, bool by_class, int cls,
                                            int b, double px, double py, double pz, double bd[KK], int bi[KK]) {
  const int j0 = b * FU_RUN, j1 = min(j0 + FU_RUN, m.J);
  int j = j0;
  for (; j + 4 <= j1; j += 4) {          // four nodes per step: their 12 coordinates are requested together
    const double* g = m.ed_points + 3 * (size_t)j;
    double q[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) q[e] = g[e];
    int cl[4] = {cls, cls, cls, cls};
    if (by_class) {
#pragma unroll
      for (int e = 0; e < 4; ++e) cl[e] = sm.ed_seg[j + e];
    }
    double d2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double dx = px - q[3 * e], dy = py - q[3 * e + 1], dz = pz - q[3 * e + 2];
      d2[e] = dx * dx + dy * dy + dz * dz;
      if (cl[e] != cls) d2[e] = 2e300;     // other class: never a neighbour (2e300 > every list entry)
    }
    if (fmin(fmin(d2[0], d2[1]), fmin(d2[2], d2[3])) <= bd[KK - 1]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) fu_insert<KK>(d2[e], j + e, bd, bi);
    }
  }
  for (; j < j1; ++j) {
    const double* g = m.ed_points + 3 * (size_t)j;
    const double dx = px - g[0], dy = py - g[1], dz = pz - g[2];
    const double d2 = dx * dx + dy * dy + dz * dz;
    if (!(by_class && sm.ed_seg[j] != cls)) fu_insert<KK>(d2, j, bd, bi);
  }
}

template <int KK>
__global__ void __launch_bounds__(256) k_fu_candidates(slm_fuse_config c, slm_surfel_model m, slm_fuse_semantic sm,
                                                        slm_new_frame fr, int32_t* __restrict__ flag,
                                                        int32_t* __restrict__ cand_idx, double* __restrict__ cand_w,
                                                        int32_t* __restrict__ counters, const double* __restrict__ boxes) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = pix < c.H * c.W && flag[pix] != 0;
  if (__ballot(act) == 0ull) return;   // no candidate pixel in this wavefront (the common case once the model covers the scene)
  const bool by_class = sm.num_classes > 0 && sm.hard_seg;   // neighbours among the nodes of the point's class
  double px = 0.0, py = 0.0, pz = 0.0;
  int t = 0, cls = -1;
  if (act) {
    t = fr.index_map[pix];
    px = fr.points[3 * (size_t)t];
    py = fr.points[3 * (size_t)t + 1];
    pz = fr.points[3 * (size_t)t + 2];
    if (by_class) cls = sm.new_seg[t];
  }
  // lanes without a candidate follow the wave with a list nothing can enter
  const double init = act ? 1e300 : -1.0;
  double bd[KK];
  int bi[KK];
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    bd[k] = init;
    bi[k] = 0x7fffffff;
  }
  const int n_runs = (m.J + FU_RUN - 1) / FU_RUN;
  int seed = -1;
  double seed_d2 = 1e300;
#pragma unroll 4
  for (int b = 0; b < n_runs; ++b) {
    const double d2 = fu_box_d2(boxes + 6 * (size_t)b, px, py, pz);
    if (act && d2 < seed_d2) {
      seed_d2 = d2;
      seed = b;
    }
  }
  unsigned long long pend = __ballot(act && seed >= 0);
  while (pend) {
    const int sb = __builtin_amdgcn_readfirstlane(__shfl(seed, __ffsll((long long)pend) - 1));
    fu_scan_run<KK>(m, sm, by_class, cls, sb, px, py, pz, bd, bi);
    pend &= ~__ballot(seed == sb);
  }
  for (int b = 0; b < n_runs; ++b) {
    const bool need = act && fu_box_d2(boxes + 6 * (size_t)b, px, py, pz) <= bd[KK - 1];   // equal: a tie with a lower index may be inside
    if (__ballot(need) == 0ull) continue;
    if (__ballot(act && seed == b) != 0ull) continue;     // scanned above as some lane's nearest run
    fu_scan_run<KK>(m, sm, by_class, cls, b, px, py, pz, bd, bi);
  }
  if (!act) return;
  if (bd[KK - 1] >= 1e300) bi[KK - 1] = -1;      // fewer than K nodes (of this class)
  if (bi[KK - 1] < 0) {       // fewer than 4 nodes of this class: the reference asserts (utils/utils.py:237)
    atomicAdd(&counters[1], 1);
    flag[pix] = 0;
    return;
  }
  double d[KK], r[KK], w[KK];
  bool stable = false;
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    d[k] = sqrt(bd[k]);
    r[k] = m.ed_radii[bi[k]];
    stable = stable || d[k] <= r[k];
  }
  if (!stable) {
    flag[pix] = 0;       // too far from its nodes: not added (nodes.py:500)
    return;
  }
  if (sm.num_classes > 0 && sm.soft_weights && !sm.hard_seg) {
    const int C = sm.num_classes;
    double js[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) js[k] = fu_jsd(sm.ed_seg_conf + (size_t)C * bi[k], sm.new_seg_conf + (size_t)C * t, C);
    fu_softmaxk_sem<KK>(d, r, js, w);
  } else {
    fu_softmaxk<KK>(d, r, w);
  }
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    cand_idx[KK * (size_t)pix + k] = bi[k];
    cand_w[KK * (size_t)pix + k] = w[k];
  }
}

__global__ void __launch_bounds__(256) k_fu_append(slm_fuse_config c, slm_surfel_model m, slm_fuse_semantic sm, slm_new_frame fr,
                                                    const int32_t* __restrict__ flag, const int32_t* __restrict__ pos,
                                                    const int32_t* __restrict__ cand_idx, const double* __restrict__ cand_w) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= c.H * c.W || !flag[pix]) return;
  const size_t o = (size_t)m.n + pos[pix];
  if (o >= (size_t)m.cap) return;
  const int t = fr.index_map[pix];
  for (int k = 0; k < 3; ++k) {
    m.points[3 * o + k] = fr.points[3 * (size_t)t + k];
    m.norms[3 * o + k] = fr.norms[3 * (size_t)t + k];
    m.colors[3 * o + k] = fr.colors[3 * (size_t)t + k];
  }
  m.radii[o] = fr.radii[t];
  m.confs[o] = fr.confs[t];
  m.time_stamp[o] = (float)fr.time;
  m.is_stable[o] = 1;
  const int K = fu_K(m);
  for (int k = 0; k < K; ++k) {
    m.knn_idx[K * o + k] = cand_idx[K * (size_t)pix + k];
    m.knn_w[K * o + k] = cand_w[K * (size_t)pix + k];
  }
  if (sm.num_classes > 0) {
    sm.seg[o] = sm.new_seg[t];
    for (int k = 0; k < sm.num_classes; ++k)
      sm.seg_conf[(size_t)sm.num_classes * o + k] = sm.new_seg_conf[(size_t)sm.num_classes * t + k];
    sm.dist2edge[o] = sm.new_dist2edge[t];
  }
}

__global__ void __launch_bounds__(256) k_fu_proj(slm_fuse_config c, slm_surfel_model m, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double u_, v_;
  fu_project(c, m.points + 3 * (size_t)i, u_, v_);
  m.projdata[2 * (size_t)i] = (float)u_;
  m.projdata[2 * (size_t)i + 1] = (float)v_;
}

__global__ void __launch_bounds__(256) k_fu_keep(slm_fuse_config c, slm_surfel_model m, int time, int32_t* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.n) return;
  const float age = (float)time - m.time_stamp[i];
  flag[i] = (m.is_stable[i] && age < (float)c.th_time_steps) ? 1 : 0;
}

__global__ void __launch_bounds__(64) k_fu_force_keep(int n, int n_keep, const int32_t* __restrict__ keep_ids, int32_t* __restrict__ flag) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_keep && keep_ids[k] >= 0 && keep_ids[k] < n) flag[keep_ids[k]] = 1;
}

__global__ void __launch_bounds__(256) k_fu_new_index(int n, const int32_t* __restrict__ flag, const int32_t* __restrict__ pos,
                                                       int32_t* __restrict__ new_index) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) new_index[i] = flag[i] ? pos[i] : -1;
}

__global__ void __launch_bounds__(256) k_fu_compact(slm_surfel_model m, slm_fuse s, const int32_t* __restrict__ flag,
                                                     const int32_t* __restrict__ pos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.n || !flag[i]) return;
  const size_t o = pos[i];
  for (int k = 0; k < 3; ++k) {
    s.s_d3[3 * o + k] = m.points[3 * (size_t)i + k];
    s.s_d3[3 * ((size_t)m.cap + o) + k] = m.norms[3 * (size_t)i + k];
    s.s_f3[3 * o + k] = m.colors[3 * (size_t)i + k];
  }
  s.s_d1[o] = m.radii[i];
  s.s_f1[o] = m.confs[i];
  s.s_f1[(size_t)m.cap + o] = m.time_stamp[i];
  const int K = fu_K(m);
  for (int k = 0; k < K; ++k) {
    s.s_i4[K * o + k] = m.knn_idx[K * (size_t)i + k];
    s.s_d4[K * o + k] = m.knn_w[K * (size_t)i + k];
  }
  s.s_f2[2 * o] = m.projdata[2 * (size_t)i];
  s.s_f2[2 * o + 1] = m.projdata[2 * (size_t)i + 1];
  if (s.sem.num_classes > 0) {
    const int C = s.sem.num_classes;
    s.s_seg[o] = s.sem.seg[i];
    for (int k = 0; k < C; ++k) s.s_sc[(size_t)C * o + k] = s.sem.seg_conf[(size_t)C * i + k];
    s.s_d2e[o] = s.sem.dist2edge[i];
  }
}

#define FU_K_DISPATCH(K, ...)                                          \
  switch (K) {                                                         \
    case 1: { constexpr int KK = 1; __VA_ARGS__; break; }              \
    case 2: { constexpr int KK = 2; __VA_ARGS__; break; }              \
    case 3: { constexpr int KK = 3; __VA_ARGS__; break; }              \
    case 4: { constexpr int KK = 4; __VA_ARGS__; break; }              \
    case 5: { constexpr int KK = 5; __VA_ARGS__; break; }              \
    case 6: { constexpr int KK = 6; __VA_ARGS__; break; }              \
    case 7: { constexpr int KK = 7; __VA_ARGS__; break; }              \
    case 8: { constexpr int KK = 8; __VA_ARGS__; break; }              \
    default: break;                                                    \
  }
static inline size_t fu_Kh(const slm_surfel_model& m) { return (size_t)(m.K > 0 ? m.K : 4); }

template <typename T>
hipError_t falloc(T*& p, size_t n) {
  return hipMalloc((void**)&p, n * sizeof(T));
}

hipError_t scan_flags(slm_fuse* f, int n, hipStream_t st) {
  size_t bytes = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, bytes, f->flag, f->pos, 0, (size_t)n, rocprim::plus<int32_t>(), st);
  if (e != hipSuccess) return e;
  if (bytes > f->cap_tmp) {
    if (f->tmp) (void)hipFree(f->tmp);
    f->tmp = nullptr;
    f->cap_tmp = 0;
    e = hipMalloc(&f->tmp, bytes);
    if (e != hipSuccess) return e;
    f->cap_tmp = bytes;
  }
  return rocprim::exclusive_scan(f->tmp, bytes, f->flag, f->pos, 0, (size_t)n, rocprim::plus<int32_t>(), st);
}

__global__ void k_fu_total(const int32_t* __restrict__ pos, const int32_t* __restrict__ flag, int n, int32_t* __restrict__ counters) {
  counters[2] = n > 0 ? pos[n - 1] + flag[n - 1] : 0;
}

// number of set flags among the first n entries (after scan_flags) in *out, counters[1] in *aux: one read-back
// of the counter block into pinned memory; synchronises the stream
hipError_t count_flags(slm_fuse* f, int n, hipStream_t st, int* out, int* aux = nullptr) {
  hipLaunchKernelGGL(k_fu_total, dim3(1), dim3(1), 0, st, f->pos, f->flag, n, f->counters);
  hipError_t e = hipMemcpyAsync(f->h_counters, f->counters, sizeof(int32_t) * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  *out = f->h_counters[2];
  if (aux) *aux = f->h_counters[1];
  return e;
}

}  // namespace

extern "C" {

int slm_fuse_create(int32_t H, int32_t W, int32_t max_surfels, slm_fuse** out) {
  if (!out || H < 8 || W < 8 || max_surfels < 1) return ffail(SLM_ERR_INVALID, "slm_fuse_create: bad argument");
  if (slm_device_count() < 1) return ffail(SLM_ERR_NO_DEVICE, "slm_fuse_create: no HIP device visible");
  slm_fuse* f = new slm_fuse();
  f->H = H;
  f->W = W;
  f->cap = max_surfels;
  const size_t HW = (size_t)H * W, cap = (size_t)max_surfels, nmax = HW > cap ? HW : cap;
  hipError_t e = falloc(f->keys, cap);
  if (e == hipSuccess) e = falloc(f->skeys, cap);
  if (e == hipSuccess) e = falloc(f->ids, cap);
  if (e == hipSuccess) e = falloc(f->sids, cap);
  if (e == hipSuccess) e = falloc(f->layers, FU_LAYERS * HW);
  if (e == hipSuccess) e = falloc(f->flag, nmax);
  if (e == hipSuccess) e = falloc(f->pos, nmax);
  if (e == hipSuccess) e = falloc(f->cand_idx, FU_KMAX * HW);
  if (e == hipSuccess) e = falloc(f->cand_w, FU_KMAX * HW);
  if (e == hipSuccess) e = falloc(f->dead, cap);
  if (e == hipSuccess) e = falloc(f->counters, 4);
  if (e == hipSuccess) e = hipHostMalloc((void**)&f->h_counters, sizeof(int32_t) * 4, hipHostMallocDefault);
  if (e == hipSuccess) e = falloc(f->s_d3, 6 * cap);
  if (e == hipSuccess) e = falloc(f->s_d1, cap);
  if (e == hipSuccess) e = falloc(f->s_d4, FU_KMAX * cap);
  if (e == hipSuccess) e = falloc(f->s_f3, 3 * cap);
  if (e == hipSuccess) e = falloc(f->s_f1, 2 * cap);
  if (e == hipSuccess) e = falloc(f->s_f2, 2 * cap);
  if (e == hipSuccess) e = falloc(f->s_i4, FU_KMAX * cap);
  if (e == hipSuccess) e = falloc(f->s_seg, cap);
  if (e == hipSuccess) e = falloc(f->s_sc, SLM_MAX_CLASSES * cap);
  if (e == hipSuccess) e = falloc(f->s_d2e, cap);
  if (e != hipSuccess) {
    slm_set_error_text((std::string("slm_fuse_create: ") + hipGetErrorString(e)).c_str());
    slm_fuse_destroy(f);
    return SLM_ERR_HIP;
  }
  *out = f;
  return SLM_OK;
}

int slm_fuse_destroy(slm_fuse* f) {
  if (!f) return SLM_OK;
  void* ptrs[] = {f->keys, f->skeys, f->ids, f->sids, f->layers, f->flag, f->pos, f->cand_idx, f->cand_w, f->dead,
                  f->counters, f->boxes, f->tmp, f->s_d3, f->s_d1, f->s_d4, f->s_f3, f->s_f1, f->s_f2, f->s_i4, f->s_seg, f->s_sc,
                  f->s_d2e};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (f->h_counters) (void)hipHostFree(f->h_counters);
  delete f;
  return SLM_OK;
}

int slm_fuse_bind_semantic(slm_fuse* f, const slm_fuse_semantic* sem) {
  if (!f) return ffail(SLM_ERR_INVALID, "slm_fuse_bind_semantic: null handle");
  if (!sem) {
    f->sem = slm_fuse_semantic{};
    return SLM_OK;
  }
  if (sem->num_classes < 1 || sem->num_classes > SLM_MAX_CLASSES)
    return ffail(SLM_ERR_UNSUPPORTED, "slm_fuse_bind_semantic: num_classes must be in 1..4");
  if (!sem->seg || !sem->seg_conf || !sem->dist2edge)
    return ffail(SLM_ERR_INVALID, "slm_fuse_bind_semantic: seg / seg_conf / dist2edge must be given");
  if ((sem->soft_weights && !sem->ed_seg_conf) || (sem->hard_seg && !sem->ed_seg))
    return ffail(SLM_ERR_INVALID, "slm_fuse_bind_semantic: the ED nodes' seg_conf (soft weights) / seg (hard_seg) is missing");
  f->sem = *sem;
  return SLM_OK;
}

static int fuse_check(slm_fuse* f, const slm_fuse_config* c, const slm_surfel_model* m) {
  if (!f || !c || !m) return ffail(SLM_ERR_INVALID, "slm_fuse: null argument");
  if (c->H != f->H || c->W != f->W) return ffail(SLM_ERR_INVALID, "slm_fuse: image size differs from slm_fuse_create");
  if (m->n < 0 || m->cap > f->cap || m->n > m->cap) return ffail(SLM_ERR_INVALID, "slm_fuse: model rows exceed the capacity");
  if (!m->points || !m->norms || !m->colors || !m->radii || !m->confs || !m->time_stamp || !m->is_stable ||
      !m->knn_idx || !m->knn_w || !m->projdata || !m->ed_points || !m->ed_radii || m->J < (int)fu_Kh(*m))
    return ffail(SLM_ERR_INVALID, "slm_fuse: null device pointer (or fewer than num_neighbors ED nodes)");
  if (m->K < 0 || m->K > FU_KMAX) return ffail(SLM_ERR_UNSUPPORTED, "slm_fuse: num_neighbors must be in 1..8 (0 = 4)");
  return SLM_OK;
}

int slm_fuse_input_data(slm_fuse* f, const slm_fuse_config* cfg, slm_surfel_model* model, const slm_new_frame* frame,
                        void* stream) {
  int rc = fuse_check(f, cfg, model);
  if (rc) return rc;
  if (!frame || !frame->valid || !frame->index_map ||
      (frame->T > 0 && (!frame->points || !frame->norms || !frame->colors || !frame->radii || !frame->confs)))
    return ffail(SLM_ERR_INVALID, "slm_fuse_input_data: null frame pointer");
  const slm_fuse_semantic sm = f->sem;
  if (sm.num_classes > 0 && frame->T > 0 && (!sm.new_seg || !sm.new_seg_conf || !sm.new_dist2edge))
    return ffail(SLM_ERR_INVALID, "slm_fuse_input_data: the frame's seg / seg_conf / dist2edge are not bound");
  hipStream_t st = (hipStream_t)stream;
  const slm_fuse_config c = *cfg;
  slm_surfel_model m = *model;
  const int HW = c.H * c.W, n = m.n;
  const dim3 blk(256), gp((HW + 255) / 256), gs((n + 255) / 256);
  // 1. per-pixel confidence-ordered layers
  if (m.merged_into && n > 0) FCHK(hipMemsetAsync(m.merged_into, 0xFF, sizeof(int32_t) * (size_t)n, st));
  FCHK(hipMemsetAsync(f->layers, 0xFF, sizeof(int32_t) * FU_LAYERS * (size_t)HW, st));
  FCHK(hipMemsetAsync(f->dead, 0, (size_t)f->cap, st));
  FCHK(hipMemsetAsync(f->counters, 0, sizeof(int32_t) * 4, st));
  // (the number of layer maps in use stays on the device, counters[0]: no read-back between the stages)
  if (n > 0) {
    hipLaunchKernelGGL(k_fu_keys, gs, blk, 0, st, c, m, f->keys, f->ids);
    size_t bytes = 0;
    // keys are (pixel << 32) | ~confidence, or ~0 for surfels that do not project: the significant bits are the 32
    // confidence bits and the bits of H*W (one more so that the all-ones key still sorts last)
    unsigned end_bit = 33;
    while (end_bit < 64 && (1ull << (end_bit - 32)) <= (unsigned long long)HW) ++end_bit;
    FCHK(rocprim::radix_sort_pairs(nullptr, bytes, f->keys, f->skeys, f->ids, f->sids, (size_t)n, 0, end_bit, st));
    if (bytes > f->cap_tmp) {
      if (f->tmp) FCHK(hipFree(f->tmp));
      f->tmp = nullptr;
      f->cap_tmp = 0;
      FCHK(hipMalloc(&f->tmp, bytes));
      f->cap_tmp = bytes;
    }
    FCHK(rocprim::radix_sort_pairs(f->tmp, bytes, f->keys, f->skeys, f->ids, f->sids, (size_t)n, 0, end_bit, st));
    hipLaunchKernelGGL(k_fu_layers, gs, blk, 0, st, n, HW, f->skeys, f->sids, f->layers, f->dead, f->counters);
  }
  // 2. the frame's points into the layers; flag = candidate new surfel
  hipLaunchKernelGGL(k_fu_merge_new, gp, blk, 0, st, c, m, sm, *frame, f->layers, f->counters, f->flag);
  // 3. surfels that share a pixel; drop the absorbed ones and those beyond the 16 maps
  if (c.merge_exist && n > 0) {
    hipLaunchKernelGGL(k_fu_merge_exist, gp, blk, 0, st, c, m, sm, frame->time, f->layers, f->counters, f->dead);
    hipLaunchKernelGGL(k_fu_apply_dead, gs, blk, 0, st, n, f->dead, m.is_stable);
  }
  // 4. skinning weights at the fused positions
  const int K = (int)fu_Kh(m);
  if (n > 0) FU_K_DISPATCH(K, hipLaunchKernelGGL(k_fu_weights<KK>, gs, blk, 0, st, m, sm));
  // 5. unmatched points with a nearby node become new surfels, in row-major pixel (= sfdata) order
  int n_new = 0;
  if (c.add_new && c.merge_new && n > 0) {   // (no surfel projects anywhere -> every flag is 0, nothing is added)
    const int n_runs = (m.J + FU_RUN - 1) / FU_RUN;
    if ((size_t)n_runs > f->cap_boxes) {
      if (f->boxes) FCHK(hipFree(f->boxes));
      f->boxes = nullptr;
      f->cap_boxes = 0;
      FCHK(hipMalloc((void**)&f->boxes, sizeof(double) * 6 * (size_t)n_runs));
      f->cap_boxes = (size_t)n_runs;
    }
    hipLaunchKernelGGL(k_fu_node_boxes, dim3(n_runs), dim3(64), 0, st, m.J, m.ed_points, f->boxes);
    FU_K_DISPATCH(K, hipLaunchKernelGGL(k_fu_candidates<KK>, gp, blk, 0, st, c, m, sm, *frame, f->flag, f->cand_idx, f->cand_w,
                                        f->counters, f->boxes));
    FCHK(scan_flags(f, HW, st));
    int n_short = 0;
    FCHK(count_flags(f, HW, st, &n_new, &n_short));
    if (n_short > 0)
      return ffail(SLM_ERR_INVALID, "slm_fuse_input_data: hard_seg needs at least num_neighbors ED nodes of every class that has new points");
    if (n + n_new > m.cap) return ffail(SLM_ERR_INVALID, "slm_fuse_input_data: model capacity too small for the new surfels");
    if (n_new > 0)
      hipLaunchKernelGGL(k_fu_append, gp, blk, 0, st, c, m, sm, *frame, f->flag, f->pos, f->cand_idx, f->cand_w);
  }
  m.n = n + n_new;
  if (m.n > 0) hipLaunchKernelGGL(k_fu_proj, dim3((m.n + 255) / 256), blk, 0, st, c, m, m.n);
  FCHK(hipGetLastError());
  model->n = m.n;
  return SLM_OK;
}

int slm_fuse_swap_stable(slm_fuse* f, const slm_fuse_config* cfg, slm_surfel_model* model, int32_t time,
                         const int32_t* keep_ids, int32_t n_keep, int32_t* new_index, void* stream) {
  int rc = fuse_check(f, cfg, model);
  if (rc) return rc;
  if (n_keep < 0 || (n_keep > 0 && !keep_ids)) return ffail(SLM_ERR_INVALID, "slm_fuse_swap_stable: bad keep_ids");
  if (!cfg->remove_unstable || model->n == 0) return SLM_OK;
  hipStream_t st = (hipStream_t)stream;
  slm_surfel_model m = *model;
  const int n = m.n;
  const dim3 blk(256), gs((n + 255) / 256);
  hipLaunchKernelGGL(k_fu_keep, gs, blk, 0, st, *cfg, m, time, f->flag);
  if (n_keep > 0)
    hipLaunchKernelGGL(k_fu_force_keep, dim3((n_keep + 63) / 64), dim3(64), 0, st, n, n_keep, keep_ids, f->flag);
  FCHK(scan_flags(f, n, st));
  if (new_index) hipLaunchKernelGGL(k_fu_new_index, gs, blk, 0, st, n, f->flag, f->pos, new_index);
  slm_fuse scratch = *f;
  hipLaunchKernelGGL(k_fu_compact, gs, blk, 0, st, m, scratch, f->flag, f->pos);   // before the read-back: runs under it
  int kept = 0;
  FCHK(count_flags(f, n, st, &kept));
  const size_t k = (size_t)kept, cap = (size_t)m.cap;
  if (kept > 0) {
    FCHK(hipMemcpyAsync(m.points, f->s_d3, sizeof(double) * 3 * k, hipMemcpyDeviceToDevice, st));
    FCHK(hipMemcpyAsync(m.norms, f->s_d3 + 3 * cap, sizeof(double) * 3 * k, hipMemcpyDeviceToDevice, st));
    FCHK(hipMemcpyAsync(m.colors, f->s_f3, sizeof(float) * 3 * k, hipMemcpyDeviceToDevice, st));
    FCHK(hipMemcpyAsync(m.radii, f->s_d1, sizeof(double) * k, hipMemcpyDeviceToDevice, st));
    FCHK(hipMemcpyAsync(m.confs, f->s_f1, sizeof(float) * k, hipMemcpyDeviceToDevice, st));
    FCHK(hipMemcpyAsync(m.time_stamp, f->s_f1 + cap, sizeof(float) * k, hipMemcpyDeviceToDevice, st));
    FCHK(hipMemcpyAsync(m.knn_idx, f->s_i4, sizeof(int32_t) * fu_Kh(m) * k, hipMemcpyDeviceToDevice, st));
    FCHK(hipMemcpyAsync(m.knn_w, f->s_d4, sizeof(double) * fu_Kh(m) * k, hipMemcpyDeviceToDevice, st));
    FCHK(hipMemcpyAsync(m.projdata, f->s_f2, sizeof(float) * 2 * k, hipMemcpyDeviceToDevice, st));
    FCHK(hipMemsetAsync(m.is_stable, 1, k, st));
    if (f->sem.num_classes > 0) {
      FCHK(hipMemcpyAsync(f->sem.seg, f->s_seg, sizeof(int32_t) * k, hipMemcpyDeviceToDevice, st));
      FCHK(hipMemcpyAsync(f->sem.seg_conf, f->s_sc, sizeof(double) * f->sem.num_classes * k, hipMemcpyDeviceToDevice, st));
      FCHK(hipMemcpyAsync(f->sem.dist2edge, f->s_d2e, sizeof(double) * k, hipMemcpyDeviceToDevice, st));
    }
  }
  FCHK(hipGetLastError());
  model->n = kept;
  return SLM_OK;
}

}  // extern "C"
