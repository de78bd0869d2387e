// slm_data.h -- per-surfel evaluation of the point-to-plane data term (device, f64).
//
// Restates SURVEY.md Appendix A.2-A.5 (reference super/utils.py:17-71,
// utils/utils.py:161-184, super/loss.py:106-173,222-290) as one fused per-surfel
// function: skin -> project -> validity -> bilinear gather -> residual -> the
// 28 Jacobian-row entries.  Nothing is materialised in HBM.
#pragma once
#include "slm_common.h"

// KK = surfel -> node neighbours (opt.num_neighbors).  4 (SLM_K) is the reference's default and the only value the
// tuple-sorted MFMA path handles; the per-entry-atomics path (data_path 1: k_data_grad / k_data_loss / k_data_resid) is
// instantiated for 1..SLM_KMAX (reference: super/loss.py:213-220 and super/utils.py:30-36 are K-generic).
#define SLM_KMAX 8
template <int KK>
struct SurfelEvalT {
  bool match;
  double r;            // lambda * n.(T(p) - o)
  int id[KK];          // node ids of the neighbours
  double row[KK * 7];  // lambda * [w_k c.Jq_k | w_k c]  (MODE 1 only)
  double c[3];         // c = d r / d T(p) / lambda (MODE >= 1): everything of the row that depends on the TARGET
  int taps[4];         // target rows of the four bilinear taps (-1 invalid)
};
typedef SurfelEvalT<SLM_K> SurfelEval;

// The 28 Jacobian-row entries of a surfel from c (reference super/loss.py:258-288): lambda * [w_k c^T dR(q_k)(p - g_k)/dq_k | w_k c]
// for its four nodes.  Needs the surfel and its nodes only -- no projection, no target table.
__device__ __forceinline__ void rows_from_c(const d3 p, const int id[4], const double w[4], double lam,
                                            const double* __restrict__ npk, const d3 c, double row[SLM_K * 7]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double2* nq = reinterpret_cast<const double2*>(npk + (size_t)SLM_NPK * id[k]);
    const double2 n0 = nq[0], n1 = nq[1], n3 = nq[3], n4 = nq[4];
    const d3 g = {n3.y, n4.x, n4.y};
    double jq[4];
    quat_jac_row(n0.x, {n0.y, n1.x, n1.y}, p - g, c, jq);
    const double lw = lam * w[k];
    row[7 * k + 0] = lw * jq[0];
    row[7 * k + 1] = lw * jq[1];
    row[7 * k + 2] = lw * jq[2];
    row[7 * k + 3] = lw * jq[3];
    row[7 * k + 4] = lw * c.x;
    row[7 * k + 5] = lw * c.y;
    row[7 * k + 6] = lw * c.z;
  }
}

// npk: packed node table (fd.node_pk at beta, fd.node_pk_try at the trial point beta + delta).
// sf_pts / sf_idx / sf_w: the surfel streams to read (caller order or tuple-sorted copies).
// The evaluation is a chain of dependent gathers (surfel -> its 4 nodes -> projected pixel ->
// 4 target rows); the loads of each stage are issued together, and callers that can fetch the
// surfel stream entries early pass them in (eval_surfel_core).
// MODE 0: residual only (loss pass); 1: residual + the 28 row entries; 2: residual + c (the evaluation pass that feeds
// the tuple-sorted Jacobian pass, slm_data_v1.hip)
// PX: the target taps come from the per-pixel table (FrameDev::tgt_px): one 32-byte gather per tap, the validity of the
// rounded pixel from the tap that IS that pixel; out.taps is not filled (the evaluation pass of the LM loop does not
// need the row numbers).  Same values, same sums: the match set and the residuals are those of the row-table form.
template <int MODE, int KK, bool PX = false>
__device__ __forceinline__ void eval_surfel_coreT(const FrameDev& fd, const d3 p, const int id[KK], const double w[KK],
                                                  double lam, const double* __restrict__ npk, SurfelEvalT<KK>& out) {
  const FrameIn& f = frame_in(fd);

  double qw[KK];
  d3 qv[KK], dk[KK];
  d3 T = {0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    out.id[k] = id[k];
    const double2* nq = reinterpret_cast<const double2*>(npk + (size_t)SLM_NPK * id[k]);
    const double2 n0 = nq[0], n1 = nq[1], n2 = nq[2], n3 = nq[3], n4 = nq[4];
    const double bb[7] = {n0.x, n0.y, n1.x, n1.y, n2.x, n2.y, n3.x};
    const d3 g = {n3.y, n4.x, n4.y};
    qw[k] = bb[0];
    qv[k] = {bb[1], bb[2], bb[3]};
    dk[k] = p - g;
    d3 t = quat_apply(qw[k], qv[k], dk[k]);
    t = {t.x + bb[4] + g.x, t.y + bb[5] + g.y, t.z + bb[6] + g.z};
    T = {T.x + w[k] * t.x, T.y + w[k] * t.y, T.z + w[k] * t.z};
  }

  out.match = false;
  out.r = 0.0;
  out.taps[0] = out.taps[1] = out.taps[2] = out.taps[3] = -1;

  // ---- projection + validity on ROUNDED coordinates (utils/utils.py:171-181) ----
  const double fx = (double)f.fx, fy = (double)f.fy, cx = (double)f.cx, cy = (double)f.cy;
  const double Ze = T.z + 1e-8;
  const double u_ = T.x * fx / Ze + cx;
  const double v_ = T.y * fy / Ze + cy;
  const double ur = rint(u_), vr = rint(v_);   // torch.round: half to even
  const int H = f.H, W = f.W;
  // proj_valid (false for NaN): 0 <= v < H-1, 0 <= u < W-1
  const bool pv = vr >= 0.0 && vr < (double)(H - 1) && ur >= 0.0 && ur < (double)(W - 1);
  // valid_pair (loss.py:229-234) and the four bilinear taps (loss.py:107-129): one round trip
  const double fv = floor(v_), cv = ceil(v_), fu = floor(u_), cu = ceil(u_);
  const double nn[4] = {fv, fv, cv, cv};
  const double mm[4] = {fu, cu, fu, cu};
  d3 o = {0, 0, 0}, n = {0, 0, 0};
  d3 dou = {0, 0, 0}, dov = {0, 0, 0}, dnu = {0, 0, 0}, dnv = {0, 0, 0};
  float4 tpv[4], tnv[4];
  if constexpr (PX) {
    bool all_ok = pv;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ni = (int)nn[t], mi = (int)mm[t];
      const bool inside = pv && (ni >= 0) && (ni < H) && (mi >= 0) && (mi < W);
      const float4* q = fd.tgt_px.get() + 2 * (size_t)(inside ? ni * W + mi : 0);
      tpv[t] = q[0];
      tnv[t] = q[1];
      all_ok = all_ok && inside && tpv[t].w != 0.f;
    }
    // the rounded pixel is one of the taps: (vr, ur) in {fv, cv} x {fu, cu}
    const int tr = ((vr != fv) ? 2 : 0) + ((ur != fu) ? 1 : 0);
    const float tvf = tr == 0 ? tnv[0].w : (tr == 1 ? tnv[1].w : (tr == 2 ? tnv[2].w : tnv[3].w));
    if (!pv || tvf == 0.f) return;
    if (!all_ok) return;   // NaN fill -> surfel dropped (loss.py:241)
  } else {
    const int coords = pv ? (int)vr * W + (int)ur : 0;
    const uint8_t tv = pv ? f.tgt_valid[coords] : (uint8_t)0;
    int rows[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ni = (int)nn[t], mi = (int)mm[t];
      const bool inside = pv && (ni >= 0) && (ni < H) && (mi >= 0) && (mi < W);
      rows[t] = inside ? f.index_map[ni * W + mi] : -1;
    }
    if (!tv) return;
    bool all_ok = true;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      out.taps[t] = rows[t];
      all_ok = all_ok && (rows[t] >= 0);
    }
    if (!all_ok) return;   // NaN fill -> surfel dropped (loss.py:241)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      tpv[t] = fd.tgt_pn[2 * (size_t)rows[t]];
      tnv[t] = fd.tgt_pn[2 * (size_t)rows[t] + 1];
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const double dn = nn[t] - v_, dm = mm[t] - u_;
    const double an = fmax(1.0 - fabs(dn), 0.0), am = fmax(1.0 - fabs(dm), 0.0);
    const float4 tp = tpv[t], tn = tnv[t];
    const d3 P = {(double)tp.x, (double)tp.y, (double)tp.z};
    const d3 Nn = {(double)tn.x, (double)tn.y, (double)tn.z};
    const double wv = an * am;
    o = {o.x + P.x * wv, o.y + P.y * wv, o.z + P.z * wv};
    n = {n.x + Nn.x * wv, n.y + Nn.y * wv, n.z + Nn.z * wv};
    if (MODE) {
      const double sn = dn >= 0.0 ? 1.0 : -1.0, sm = dm >= 0.0 ? 1.0 : -1.0;
      const double gu = an * sm, gv = am * sn;   // d/du, d/dv weights
      dou = {dou.x + P.x * gu, dou.y + P.y * gu, dou.z + P.z * gu};
      dov = {dov.x + P.x * gv, dov.y + P.y * gv, dov.z + P.z * gv};
      dnu = {dnu.x + Nn.x * gu, dnu.y + Nn.y * gu, dnu.z + Nn.z * gu};
      dnv = {dnv.x + Nn.x * gv, dnv.y + Nn.y * gv, dnv.z + Nn.z * gv};
    }
  }
  // NaN in the tables behaves like the reference's isnan filter
  if (!(o.x == o.x && o.y == o.y && o.z == o.z && n.x == n.x && n.y == n.y && n.z == n.z)) return;

  const d3 e = T - o;
  out.match = true;
  out.r = lam * dot(n, e);
  if (!MODE) return;

  // ---- c = n^T (I - A) + e^T B, A = do/d(u,v) Pi, B = dn/d(u,v) Pi (loss.py:257-281) ----
  const double Z = T.z;   // no epsilon in dPi (loss.py:161-173)
  const d3 Pi0 = {fx / Z, 0.0, -fx * T.x / (Z * Z)};
  const d3 Pi1 = {0.0, fy / Z, -fy * T.y / (Z * Z)};
  const double s0 = dot(e, dnu) - dot(n, dou);
  const double s1 = dot(e, dnv) - dot(n, dov);
  const d3 c = {n.x + s0 * Pi0.x + s1 * Pi1.x, n.y + s0 * Pi0.y + s1 * Pi1.y,
                n.z + s0 * Pi0.z + s1 * Pi1.z};
  out.c[0] = c.x;
  out.c[1] = c.y;
  out.c[2] = c.z;
  if (MODE == 2) return;
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    double jq[4];
    quat_jac_row(qw[k], qv[k], dk[k], c, jq);
    const double lw = lam * w[k];
    out.row[7 * k + 0] = lw * jq[0];
    out.row[7 * k + 1] = lw * jq[1];
    out.row[7 * k + 2] = lw * jq[2];
    out.row[7 * k + 3] = lw * jq[3];
    out.row[7 * k + 4] = lw * c.x;
    out.row[7 * k + 5] = lw * c.y;
    out.row[7 * k + 6] = lw * c.z;
  }
}

// the K = 4 form (one 16-byte load of the ids, what every caller of the tuple-sorted path uses)
template <int MODE, bool PX = false>
__device__ __forceinline__ void eval_surfel_core(const FrameDev& fd, const d3 p, int4 ids, const double w[4],
                                                 double lam, const double* __restrict__ npk, SurfelEval& out) {
  const int id[4] = {ids.x, ids.y, ids.z, ids.w};
  eval_surfel_coreT<MODE, SLM_K, PX>(fd, p, id, w, lam, npk, out);
}

template <int MODE>
__device__ __forceinline__ void eval_surfel_at(const FrameDev& fd, const void* __restrict__ sf_pts,
                                               const int* __restrict__ sf_idx,
                                               const void* __restrict__ sf_w, double lam,
                                               const double* __restrict__ npk, int i,
                                               SurfelEval& out) {
  double w[4];
  ld_state4(sf_w, (size_t)i, fd.f.state_f64, w);
  eval_surfel_core<MODE>(fd, ld_state3(sf_pts, (size_t)i, fd.f.state_f64),
                         *reinterpret_cast<const int4*>(sf_idx + 4 * (size_t)i), w, lam, npk, out);
}

// surfel i of the caller's own arrays with KK neighbours per surfel (rows of KK ids / weights)
template <int MODE, int KK>
__device__ __forceinline__ void eval_surfel(const FrameDev& fd, double lam, const double* npk, int i, SurfelEvalT<KK>& out) {
  const FrameIn& f = frame_in(fd);
  if constexpr (KK == SLM_K) {
    eval_surfel_at<MODE>(fd, f.sf_points, f.sf_knn_idx, f.sf_knn_w, lam, npk, i, out);
  } else {
    double wk[KK];
    int idk[KK];
    const void* wp = f.sf_knn_w;
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      idk[k] = f.sf_knn_idx[(size_t)KK * i + k];
      wk[k] = fd.f.state_f64 ? static_cast<const double*>(wp)[(size_t)KK * i + k] : (double)static_cast<const float*>(wp)[(size_t)KK * i + k];
    }
    eval_surfel_coreT<MODE, KK>(fd, ld_state3(f.sf_points, (size_t)i, fd.f.state_f64), idk, wk, lam, npk, out);
  }
}
