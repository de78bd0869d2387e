// slm_gf.h -- state shared by the GraphFit kernels (slm_gf.hip) and the Semantic-SuPer
// kernels (slm_sem.hip).
#pragma once
#include "slm_data.h"

// Block partials of the per-slot scalars (the global row's gradient, the loss terms, the counts) go to GF_NCOPY spread copies
// behind `terms` -- copy (blockIdx.x % GF_NCOPY), entry a: terms[SLM_GF_NTERMS + 16 copy + a] -- and k_gf_fold sums the copies
// in a fixed order into grad[7J..] / terms[].  One global f64 atomic per block and scalar onto ONE address serialises in the
// L2: 782 blocks x ~0.1 us = the whole 82 us of k_gf_data at C2 (round 6); spread over 64 addresses it is 12 per address.
//   a = 0..6 global row (k_gf_data, k_gf_reg) | 7, 8 point-plane loss, kept | 9, 10 correspondence loss, kept |
//   11, 12, 13 face, arap, rot | 14, 15 morphing sum, kept
#define GF_NCOPY 64
#define GF_PART_DOUBLES (16 * GF_NCOPY)

struct GfSlot {
  slm_gf_frame f;
  int32_t bound;
  int32_t step;          // optimiser steps done
  int32_t shard_lo, shard_hi;   // surfels [lo,hi) are evaluated by this rank (slm_gf_set_shard)
  double* dv;            // (J+1,7)
  double* grad;          // (J+1,7)
  double* m1;            // momentum buffer / Adam exp_avg
  double* m2;            // Adam exp_avg_sq
  double* terms;         // [0..3] face, arap, rot, point-plane; [4] residuals kept;
                         // [5] morphing loss sum (weighted mean after k_gf_finish), [6] kept, [7] candidates;
                         // [8] flow-correspondence loss, [9] its residuals kept   (SLM_GF_NTERMS)
  const float* flow;     // (2,H,W) optical flow of the frame (slm_gf_bind_flow) or null
  // ---- Semantic-SuPer (slm_gf_bind_semantic) ----
  slm_gf_semantic sem;
  int32_t sem_bound;
  int32_t edge_off[SLM_MAX_CLASSES + 1];   // class c owns edge_xy[edge_off[c] .. edge_off[c+1])
  float2* edge_xy;       // boundary pixels (x,y), per class, row-major pixel order
  double2* morph_g;      // (N) d(loss_i)/d(x,y) of the morphing term, 0 when not kept
};

// GfSlot as DEVICE code reads it: the same bytes, every pointer typed GP<> (global address space, slm_common.h) so that
// the kernels issue global_load / global_store instead of FLAT accesses.  Host code keeps using GfSlot.
struct GfFrameIn {
  FrameIn base;
  GP<const uint8_t> sf_stable;
  GP<const void> ed_knn_w;
  GP<const int32_t> ed_triangles;
  GP<const void> ed_triangle_areas;
  int32_t n_triangles;
  int32_t pad;
};
struct GfSemIn {
  int32_t num_classes;
  int32_t pad;
  GP<const int32_t> sf_seg;
  GP<const float> sf_seg_conf;
  GP<const float> tgt_seg_conf;
  GP<const float> img_seg_conf;
  GP<const int32_t> img_seg;
};
struct GfSlotDev {
  GfFrameIn f;
  int32_t bound;
  int32_t step;
  int32_t shard_lo, shard_hi;
  GP<double> dv;
  GP<double> grad;
  GP<double> m1;
  GP<double> m2;
  GP<double> terms;
  GP<const float> flow;
  GfSemIn sem;
  int32_t sem_bound;
  int32_t edge_off[SLM_MAX_CLASSES + 1];
  GP<float2> edge_xy;
  GP<double2> morph_g;
};
static_assert(sizeof(GfFrameIn) == sizeof(slm_gf_frame) && sizeof(GfSemIn) == sizeof(slm_gf_semantic) &&
              sizeof(GfSlotDev) == sizeof(GfSlot), "GfSlotDev mirrors GfSlot");
static_assert(offsetof(GfSlotDev, dv) == offsetof(GfSlot, dv) && offsetof(GfSlotDev, flow) == offsetof(GfSlot, flow) &&
              offsetof(GfSlotDev, sem) == offsetof(GfSlot, sem) && offsetof(GfSlotDev, edge_xy) == offsetof(GfSlot, edge_xy) &&
              offsetof(GfSlotDev, morph_g) == offsetof(GfSlot, morph_g) && offsetof(GfFrameIn, ed_triangle_areas) == offsetof(slm_gf_frame, ed_triangle_areas) &&
              offsetof(GfSemIn, img_seg) == offsetof(slm_gf_semantic, img_seg), "GfSlotDev mirrors GfSlot");
__device__ __forceinline__ GfSlotDev* gf_dev(GfSlot* slots) { return reinterpret_cast<GfSlotDev*>(slots); }

// R(q)^T c for an un-normalised quaternion = R(conj q) c
__device__ __forceinline__ d3 quat_apply_t(double w, d3 v, d3 c) {
  return quat_apply(w, {-v.x, -v.y, -v.z}, c);
}

// deformed surfel i: T(p) = sum_k w_k [R(q_k)(p-g_k) + b_k + g_k], P = R(q_g) T + b_g
// (deform_source, super/deform_mesh.py:198-221; K-generic like the reference: KK = opt.num_neighbors, 1..8)
template <int KK>
struct GfSkinT {
  int id[KK];
  double w[KK], qw[KK];
  d3 qv[KK], dk[KK], T, P;
  double gw;
  d3 gv;
};
typedef GfSkinT<SLM_K> GfSkin;

template <int KK>
__device__ __forceinline__ void gf_skin(const GfSlotDev& s, int i, GfSkinT<KK>& k) {
  const FrameIn& f = s.f.base;
  const d3 p = ld_state3(f.sf_points, (size_t)i, f.state_f64);
  if constexpr (KK == SLM_K) {
    const int4 ids = *reinterpret_cast<const int4*>(f.sf_knn_idx + 4 * (size_t)i);
    k.id[0] = ids.x; k.id[1] = ids.y; k.id[2] = ids.z; k.id[3] = ids.w;
    ld_state4(f.sf_knn_w, (size_t)i, f.state_f64, k.w);
  } else {
#pragma unroll
    for (int a = 0; a < KK; ++a) {
      k.id[a] = f.sf_knn_idx[(size_t)KK * i + a];
      k.w[a] = ld_state1(f.sf_knn_w, (size_t)KK * i + a, f.state_f64);
    }
  }
  k.T = {0, 0, 0};
#pragma unroll
  for (int a = 0; a < KK; ++a) {
    const double* b = s.dv + 7 * k.id[a];
    const d3 g = ld_state3(f.ed_points, (size_t)k.id[a], f.state_f64);
    k.qw[a] = b[0];
    k.qv[a] = {b[1], b[2], b[3]};
    k.dk[a] = p - g;
    d3 t = quat_apply(k.qw[a], k.qv[a], k.dk[a]);
    t = {t.x + b[4] + g.x, t.y + b[5] + g.y, t.z + b[6] + g.z};
    k.T = {k.T.x + k.w[a] * t.x, k.T.y + k.w[a] * t.y, k.T.z + k.w[a] * t.z};
  }
  const double* bgl = s.dv + 7 * f.J;
  k.gw = bgl[0];
  k.gv = {bgl[1], bgl[2], bgl[3]};
  k.P = quat_apply(k.gw, k.gv, k.T);
  k.P = {k.P.x + bgl[4], k.P.y + bgl[5], k.P.z + bgl[6]};
}

// The same without the per-neighbour state: ids, weights, p, T, P and the global row only.  k_gf_data keeps THIS across its
// sampling phase (26 + 3 K registers instead of 30 + 17 K) and re-reads the nodes -- cache hits -- when it back-propagates:
// with GfSkinT held live the kernel needed more than 256 VGPRs and ran at ONE wave per SIMD (round 6).
template <int KK>
struct GfSkinLight {
  int id[KK];
  double w[KK];
  d3 p, T, P;
  double gw;
  d3 gv;
};
template <int KK>
__device__ __forceinline__ void gf_skin_light(const GfSlotDev& s, int i, GfSkinLight<KK>& k) {
  const FrameIn& f = s.f.base;
  k.p = ld_state3(f.sf_points, (size_t)i, f.state_f64);
  if constexpr (KK == SLM_K) {
    const int4 ids = *reinterpret_cast<const int4*>(f.sf_knn_idx + 4 * (size_t)i);
    k.id[0] = ids.x; k.id[1] = ids.y; k.id[2] = ids.z; k.id[3] = ids.w;
    ld_state4(f.sf_knn_w, (size_t)i, f.state_f64, k.w);
  } else {
#pragma unroll
    for (int a = 0; a < KK; ++a) {
      k.id[a] = f.sf_knn_idx[(size_t)KK * i + a];
      k.w[a] = ld_state1(f.sf_knn_w, (size_t)KK * i + a, f.state_f64);
    }
  }
  k.T = {0, 0, 0};
#pragma unroll
  for (int a = 0; a < KK; ++a) {
    const double* b = s.dv + 7 * k.id[a];
    const d3 g = ld_state3(f.ed_points, (size_t)k.id[a], f.state_f64);
    d3 t = quat_apply(b[0], {b[1], b[2], b[3]}, k.p - g);
    t = {t.x + b[4] + g.x, t.y + b[5] + g.y, t.z + b[6] + g.z};
    k.T = {k.T.x + k.w[a] * t.x, k.T.y + k.w[a] * t.y, k.T.z + k.w[a] * t.z};
  }
  const double* bgl = s.dv + 7 * f.J;
  k.gw = bgl[0];
  k.gv = {bgl[1], bgl[2], bgl[3]};
  k.P = quat_apply(k.gw, k.gv, k.T);
  k.P = {k.P.x + bgl[4], k.P.y + bgl[5], k.P.z + bgl[6]};
}

// the deformed position alone, K at run time (the morphing term's pass: it needs P only) -- the same sums in the same order
__device__ __forceinline__ d3 gf_skin_pos(const GfSlotDev& s, int i) {
  const FrameIn& f = s.f.base;
  const int K = f.K;
  const d3 p = ld_state3(f.sf_points, (size_t)i, f.state_f64);
  d3 T = {0, 0, 0};
  for (int a = 0; a < K; ++a) {
    const int id = f.sf_knn_idx[(size_t)K * i + a];
    const double w = ld_state1(f.sf_knn_w, (size_t)K * i + a, f.state_f64);
    const double* b = s.dv + 7 * id;
    const d3 g = ld_state3(f.ed_points, (size_t)id, f.state_f64);
    d3 t = quat_apply(b[0], {b[1], b[2], b[3]}, p - g);
    t = {t.x + b[4] + g.x, t.y + b[5] + g.y, t.z + b[6] + g.z};
    T = {T.x + w * t.x, T.y + w * t.y, T.z + w * t.z};
  }
  const double* bgl = s.dv + 7 * f.J;
  d3 P = quat_apply(bgl[0], {bgl[1], bgl[2], bgl[3]}, T);
  return {P.x + bgl[4], P.y + bgl[5], P.z + bgl[6]};
}
