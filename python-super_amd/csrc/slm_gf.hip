// slm_gf.hip -- the reference's default per-frame optimiser (GraphFit, autograd + SGD/Adam,
// super/deform_mesh.py:198-379) with hand-derived gradients, entirely on the device.
//
// Parameters: dv (J+1,7), rows 0..J-1 local warps, row J the global transform T_g = (q_g,b_g).
//   surfel:  T(p) = sum_k w_k [R(q_k)(p-g_k) + b_k + g_k],   P = R(q_g) T(p) + b_g
//   node:    V_j  = R(q_g)(g_j + b_j) + b_g
// Losses (super/deform_mesh.py:25-196, super/loss.py:293-401,458-473,502-505):
//   point-plane  w_d sum (n.(P-o))^2   margin-1 validity on rounded coords, all 4 taps mapped
//   ARAP         w_a sum_jk w^ED_jk |R(q_k)d + b_k - f32(d) - b_j|^2   (f32-rounded d: loss.py:468)
//   Rot          w_r sum_{rows 0..J} (1 - |q|^2)^2
//   face         w_f sum_tri (area(V) - area0)^2,  area = 1/2 sqrt(|e1 x e2|^2 + 1e-13)
// Gradient of the global row is divided by J before the step (deform_mesh.py:326).
// One f64 atomic per gradient entry and surfel; the global row is reduced per block first.
#include <string>
#include <vector>

#include "slm_data.h"

struct GfSlot {
  slm_gf_frame f;
  int32_t bound;
  int32_t step;          // optimiser steps done
  double* dv;            // (J+1,7)
  double* grad;          // (J+1,7)
  double* m1;            // momentum buffer / Adam exp_avg
  double* m2;            // Adam exp_avg_sq
  double* terms;         // [0..3] face, arap, rot, point-plane; [4] matched
};

// R(q)^T c for an un-normalised quaternion = R(conj q) c
__device__ __forceinline__ d3 quat_apply_t(double w, d3 v, d3 c) {
  return quat_apply(w, {-v.x, -v.y, -v.z}, c);
}

__global__ void __launch_bounds__(256) k_gf_zero(GfSlot* __restrict__ slots) {
  GfSlot& s = slots[blockIdx.y];
  if (!s.bound) return;
  const int n = (s.f.base.J + 1) * 7;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) s.grad[e] = 0.0;
  if (blockIdx.x == 0 && threadIdx.x < 5) s.terms[threadIdx.x] = 0.0;
}

// grid = (ceil(maxN/256), n_frames)
__global__ void __launch_bounds__(256) k_gf_data(GfSlot* __restrict__ slots, double lam) {
  __shared__ double sm[16];
  GfSlot& s = slots[blockIdx.y];
  if (!s.bound) return;
  const slm_frame& f = s.f.base;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int J = f.J;
  double gq[4] = {0, 0, 0, 0}, gb[3] = {0, 0, 0}, loss = 0.0, cnt = 0.0;
  if (i < f.N && (!s.f.sf_stable || s.f.sf_stable[i])) {
    const d3 p = {(double)f.sf_points[3 * i], (double)f.sf_points[3 * i + 1], (double)f.sf_points[3 * i + 2]};
    const int4 ids = *reinterpret_cast<const int4*>(f.sf_knn_idx + 4 * i);
    const float4 wf = *reinterpret_cast<const float4*>(f.sf_knn_w + 4 * i);
    const int id[4] = {ids.x, ids.y, ids.z, ids.w};
    const double w[4] = {(double)wf.x, (double)wf.y, (double)wf.z, (double)wf.w};
    double qw[4];
    d3 qv[4], dk[4], T = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double* b = s.dv + 7 * id[k];
      const d3 g = {(double)f.ed_points[3 * id[k]], (double)f.ed_points[3 * id[k] + 1],
                    (double)f.ed_points[3 * id[k] + 2]};
      qw[k] = b[0];
      qv[k] = {b[1], b[2], b[3]};
      dk[k] = p - g;
      d3 t = quat_apply(qw[k], qv[k], dk[k]);
      t = {t.x + b[4] + g.x, t.y + b[5] + g.y, t.z + b[6] + g.z};
      T = {T.x + w[k] * t.x, T.y + w[k] * t.y, T.z + w[k] * t.z};
    }
    const double* bgl = s.dv + 7 * J;
    const double gw = bgl[0];
    const d3 gv = {bgl[1], bgl[2], bgl[3]};
    d3 P = quat_apply(gw, gv, T);
    P = {P.x + bgl[4], P.y + bgl[5], P.z + bgl[6]};

    const double fx = (double)f.fx, fy = (double)f.fy, cx = (double)f.cx, cy = (double)f.cy;
    const double Ze = P.z + 1e-8;
    const double u_ = P.x * fx / Ze + cx, v_ = P.y * fy / Ze + cy;
    const double ur = rint(u_), vr = rint(v_);
    const int H = f.H, W = f.W;
    // valid_margin = 1 (loss.py:306-309)
    if (vr >= 1.0 && vr < (double)(H - 2) && ur >= 1.0 && ur < (double)(W - 2)) {
      const double fv = floor(v_), cv = ceil(v_), fu = floor(u_), cu = ceil(u_);
      const double nn[4] = {fv, fv, cv, cv}, mm[4] = {fu, cu, fu, cu};
      int rows[4];
      bool all_ok = true;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        rows[t] = f.index_map[(int)nn[t] * W + (int)mm[t]];
        all_ok = all_ok && rows[t] >= 0;
      }
      if (all_ok) {
        d3 o = {0, 0, 0}, n = {0, 0, 0}, dou = {0, 0, 0}, dov = {0, 0, 0}, dnu = {0, 0, 0}, dnv = {0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double dn = nn[t] - v_, dm = mm[t] - u_;
          const double an = fmax(1.0 - fabs(dn), 0.0), am = fmax(1.0 - fabs(dm), 0.0);
          const float* tp = f.tgt_points + 3 * (size_t)rows[t];
          const float* tn = f.tgt_norms + 3 * (size_t)rows[t];
          const d3 Pt = {(double)tp[0], (double)tp[1], (double)tp[2]};
          const d3 Nt = {(double)tn[0], (double)tn[1], (double)tn[2]};
          const double wv = an * am;
          // autograd through |.|: d|x|/dx = sign(x) with sign(0) = 0
          const double sn = dn > 0.0 ? 1.0 : (dn < 0.0 ? -1.0 : 0.0);
          const double smm = dm > 0.0 ? 1.0 : (dm < 0.0 ? -1.0 : 0.0);
          const double gu = an * smm, gvv = am * sn;
          o = {o.x + Pt.x * wv, o.y + Pt.y * wv, o.z + Pt.z * wv};
          n = {n.x + Nt.x * wv, n.y + Nt.y * wv, n.z + Nt.z * wv};
          dou = {dou.x + Pt.x * gu, dou.y + Pt.y * gu, dou.z + Pt.z * gu};
          dov = {dov.x + Pt.x * gvv, dov.y + Pt.y * gvv, dov.z + Pt.z * gvv};
          dnu = {dnu.x + Nt.x * gu, dnu.y + Nt.y * gu, dnu.z + Nt.z * gu};
          dnv = {dnv.x + Nt.x * gvv, dnv.y + Nt.y * gvv, dnv.z + Nt.z * gvv};
        }
        const d3 e = P - o;
        const double r = dot(n, e);
        loss = lam * r * r;
        cnt = 1.0;
        // c = dr/dP ; the forward divides by Z + 1e-8, and so does its derivative
        const d3 Pi0 = {fx / Ze, 0.0, -fx * P.x / (Ze * Ze)};
        const d3 Pi1 = {0.0, fy / Ze, -fy * P.y / (Ze * Ze)};
        const double s0 = dot(e, dnu) - dot(n, dou), s1 = dot(e, dnv) - dot(n, dov);
        const d3 c = {n.x + s0 * Pi0.x + s1 * Pi1.x, n.y + s0 * Pi0.y + s1 * Pi1.y,
                      n.z + s0 * Pi0.z + s1 * Pi1.z};
        const double G = 2.0 * lam * r;
        double jq[4];
        quat_jac_row(gw, gv, T, c, jq);
#pragma unroll
        for (int a = 0; a < 4; ++a) gq[a] = G * jq[a];
        gb[0] = G * c.x;
        gb[1] = G * c.y;
        gb[2] = G * c.z;
        const d3 cl = quat_apply_t(gw, gv, c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          quat_jac_row(qw[k], qv[k], dk[k], cl, jq);
          const double Gw = G * w[k];
          double* gr = s.grad + 7 * id[k];
          atomic_add_f64(gr + 0, Gw * jq[0]);
          atomic_add_f64(gr + 1, Gw * jq[1]);
          atomic_add_f64(gr + 2, Gw * jq[2]);
          atomic_add_f64(gr + 3, Gw * jq[3]);
          atomic_add_f64(gr + 4, Gw * cl.x);
          atomic_add_f64(gr + 5, Gw * cl.y);
          atomic_add_f64(gr + 6, Gw * cl.z);
        }
      }
    }
  }
  // global row, loss and count: block reduction, then one atomic each
  double vals[9] = {gq[0], gq[1], gq[2], gq[3], gb[0], gb[1], gb[2], loss, cnt};
#pragma unroll
  for (int a = 0; a < 9; ++a) {
    const double t = block_sum(vals[a], sm);
    if (threadIdx.x == 0 && t != 0.0) {
      if (a < 7) atomic_add_f64(s.grad + 7 * J + a, t);
      else if (a == 7) atomic_add_f64(s.terms + 3, t);
      else atomic_add_f64(s.terms + 4, t);
    }
  }
}

// ARAP: one thread per (node, slot); Rot: one thread per row (J+1); face: one per triangle.
// grid = (ceil(max(J*K_ED, J+1, Tr)/256), n_frames)
__global__ void __launch_bounds__(256) k_gf_reg(GfSlot* __restrict__ slots, int use_arap, double lam_a,
                                                 int use_rot, double lam_r, int use_face, double lam_f) {
  __shared__ double sm[16];
  GfSlot& s = slots[blockIdx.y];
  if (!s.bound) return;
  const slm_frame& f = s.f.base;
  const int J = f.J, Ke = f.K_ED;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double la = 0.0, lr = 0.0, lf = 0.0;
  double gq[4] = {0, 0, 0, 0}, gb[3] = {0, 0, 0};   // global-row contributions of this thread
  if (use_arap && t < J * Ke) {
    const int j = t / Ke, k = f.ed_knn_idx[t];
    const float* g = f.ed_points;
    const d3 d = {(double)g[3 * j] - (double)g[3 * k], (double)g[3 * j + 1] - (double)g[3 * k + 1],
                  (double)g[3 * j + 2] - (double)g[3 * k + 2]};
    const d3 d32 = {(double)(float)d.x, (double)(float)d.y, (double)(float)d.z};
    const double* bk = s.dv + 7 * k;
    const double* bj = s.dv + 7 * j;
    const d3 qv = {bk[1], bk[2], bk[3]};
    const d3 tr = quat_apply(bk[0], qv, d);
    const d3 r = {tr.x + bk[4] - d32.x - bj[4], tr.y + bk[5] - d32.y - bj[5], tr.z + bk[6] - d32.z - bj[6]};
    const double wjk = (double)s.f.ed_knn_w[t];
    la = lam_a * wjk * dot(r, r);
    const double G = 2.0 * lam_a * wjk;
    double jq[4];
    quat_jac_row(bk[0], qv, d, r, jq);
    double* gk = s.grad + 7 * k;
    double* gj = s.grad + 7 * j;
    atomic_add_f64(gk + 0, G * jq[0]);
    atomic_add_f64(gk + 1, G * jq[1]);
    atomic_add_f64(gk + 2, G * jq[2]);
    atomic_add_f64(gk + 3, G * jq[3]);
    atomic_add_f64(gk + 4, G * r.x);
    atomic_add_f64(gk + 5, G * r.y);
    atomic_add_f64(gk + 6, G * r.z);
    atomic_add_f64(gj + 4, -G * r.x);
    atomic_add_f64(gj + 5, -G * r.y);
    atomic_add_f64(gj + 6, -G * r.z);
  }
  if (use_rot && t <= J) {
    const double* q = s.dv + 7 * t;
    const double sres = 1.0 - (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    lr = lam_r * sres * sres;
    const double G = -4.0 * lam_r * sres;
    double* gr = s.grad + 7 * t;
    atomic_add_f64(gr + 0, G * q[0]);
    atomic_add_f64(gr + 1, G * q[1]);
    atomic_add_f64(gr + 2, G * q[2]);
    atomic_add_f64(gr + 3, G * q[3]);
  }
  if (use_face && s.f.ed_triangles && t < s.f.n_triangles) {
    const int Tr = s.f.n_triangles;
    const int iv[3] = {s.f.ed_triangles[t], s.f.ed_triangles[Tr + t], s.f.ed_triangles[2 * Tr + t]};
    const double* bgl = s.dv + 7 * J;
    const double gw = bgl[0];
    const d3 gv = {bgl[1], bgl[2], bgl[3]};
    d3 loc[3], V[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double* b = s.dv + 7 * iv[a];
      loc[a] = {(double)f.ed_points[3 * iv[a]] + b[4], (double)f.ed_points[3 * iv[a] + 1] + b[5],
                (double)f.ed_points[3 * iv[a] + 2] + b[6]};
      V[a] = quat_apply(gw, gv, loc[a]);   // + b_g cancels in the edge vectors
    }
    const d3 e1 = V[1] - V[0], e2 = V[2] - V[0];
    const d3 cr = cross(e1, e2);
    const double area = 0.5 * sqrt(dot(cr, cr) + 1e-13);
    const double da = area - (double)s.f.ed_triangle_areas[t];
    lf = lam_f * da * da;
    const double coef = 2.0 * lam_f * da / (4.0 * area);
    const d3 y = {coef * cr.x, coef * cr.y, coef * cr.z};     // dL/d(cr)
    d3 gV[3];
    gV[1] = cross(e2, y);
    gV[2] = cross(y, e1);
    gV[0] = {-gV[1].x - gV[2].x, -gV[1].y - gV[2].y, -gV[1].z - gV[2].z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double jq[4];
      quat_jac_row(gw, gv, loc[a], gV[a], jq);
#pragma unroll
      for (int c = 0; c < 4; ++c) gq[c] += jq[c];
      gb[0] += gV[a].x;
      gb[1] += gV[a].y;
      gb[2] += gV[a].z;
      const d3 gl = quat_apply_t(gw, gv, gV[a]);
      double* gr = s.grad + 7 * iv[a];
      atomic_add_f64(gr + 4, gl.x);
      atomic_add_f64(gr + 5, gl.y);
      atomic_add_f64(gr + 6, gl.z);
    }
  }
  double vals[10] = {gq[0], gq[1], gq[2], gq[3], gb[0], gb[1], gb[2], lf, la, lr};
#pragma unroll
  for (int a = 0; a < 10; ++a) {
    const double tt = block_sum(vals[a], sm);
    if (threadIdx.x == 0 && tt != 0.0) {
      if (a < 7) atomic_add_f64(s.grad + 7 * J + a, tt);
      else atomic_add_f64(s.terms + (a - 7), tt);
    }
  }
}

// grad[J] /= J, then torch.optim.SGD(momentum=0.9) or torch.optim.Adam step (float64).
__global__ void __launch_bounds__(256) k_gf_step(GfSlot* __restrict__ slots, int optimizer, double lr,
                                                  int apply) {
  GfSlot& s = slots[blockIdx.y];
  if (!s.bound) return;
  const int J = s.f.base.J, n = (J + 1) * 7;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double g = s.grad[e];
  if (e >= 7 * J) {
    g /= (double)J;
    s.grad[e] = g;
  }
  if (!apply) return;
  const int t = s.step + 1;
  if (optimizer == 0) {
    const double buf = (t == 1) ? g : 0.9 * s.m1[e] + g;
    s.m1[e] = buf;
    s.dv[e] -= lr * buf;
  } else {
    const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
    const double m = b1 * s.m1[e] + (1.0 - b1) * g;
    const double v = b2 * s.m2[e] + (1.0 - b2) * g * g;
    s.m1[e] = m;
    s.m2[e] = v;
    const double bc1 = 1.0 - pow(b1, (double)t), bc2 = 1.0 - pow(b2, (double)t);
    const double denom = sqrt(v) / sqrt(bc2) + eps;
    s.dv[e] -= (lr / bc1) * m / denom;
  }
}

__global__ void k_gf_advance(GfSlot* __restrict__ slots) {
  GfSlot& s = slots[blockIdx.x];
  if (s.bound && threadIdx.x == 0) s.step += 1;
}

__global__ void __launch_bounds__(256) k_gf_init(GfSlot* __restrict__ slots, int slot) {
  GfSlot& s = slots[slot];
  const int n = (s.f.base.J + 1) * 7;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) {
    s.dv[e] = (e % 7 == 0) ? 1.0 : 0.0;
    s.grad[e] = 0.0;
    s.m1[e] = 0.0;
    s.m2[e] = 0.0;
  }
  if (e == 0) s.step = 0;
}

// Surfels.update, autograd variant (super/nodes.py:193-223): T(p) + b_g (the global ROTATION is
// applied to the normals only, exactly as the reference does), nodes += b_j + b_g.
__global__ void __launch_bounds__(256) k_gf_update_surfels(int N, int J, float* __restrict__ pts,
                                                            float* __restrict__ nrm,
                                                            const int* __restrict__ knn_idx,
                                                            const float* __restrict__ knn_w,
                                                            const float* __restrict__ ed_pts,
                                                            const double* __restrict__ dv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const d3 p = {(double)pts[3 * i], (double)pts[3 * i + 1], (double)pts[3 * i + 2]};
  const d3 n0 = {(double)nrm[3 * i], (double)nrm[3 * i + 1], (double)nrm[3 * i + 2]};
  const int4 ids = *reinterpret_cast<const int4*>(knn_idx + 4 * i);
  const float4 wf = *reinterpret_cast<const float4*>(knn_w + 4 * i);
  const int id[4] = {ids.x, ids.y, ids.z, ids.w};
  const double w[4] = {(double)wf.x, (double)wf.y, (double)wf.z, (double)wf.w};
  d3 T = {0, 0, 0}, Nn = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double* b = dv + 7 * id[k];
    const d3 g = {(double)ed_pts[3 * id[k]], (double)ed_pts[3 * id[k] + 1], (double)ed_pts[3 * id[k] + 2]};
    const d3 qv = {b[1], b[2], b[3]};
    d3 t = quat_apply(b[0], qv, p - g);
    t = {t.x + b[4] + g.x, t.y + b[5] + g.y, t.z + b[6] + g.z};
    T = {T.x + w[k] * t.x, T.y + w[k] * t.y, T.z + w[k] * t.z};
    d3 rn = quat_apply(b[0], qv, n0);
    rn = {rn.x + b[4], rn.y + b[5], rn.z + b[6]};   // 7-wide beta: b is added (nodes.py:207-209)
    Nn = {Nn.x + w[k] * rn.x, Nn.y + w[k] * rn.y, Nn.z + w[k] * rn.z};
  }
  const double* bgl = dv + 7 * J;
  Nn = quat_apply(bgl[0], {bgl[1], bgl[2], bgl[3]}, Nn);
  const double nl = fmax(sqrt(dot(Nn, Nn)), 1e-12);
  pts[3 * i] = (float)(T.x + bgl[4]);
  pts[3 * i + 1] = (float)(T.y + bgl[5]);
  pts[3 * i + 2] = (float)(T.z + bgl[6]);
  nrm[3 * i] = (float)(Nn.x / nl);
  nrm[3 * i + 1] = (float)(Nn.y / nl);
  nrm[3 * i + 2] = (float)(Nn.z / nl);
}

__global__ void __launch_bounds__(256) k_gf_update_nodes(int J, float* __restrict__ ed_pts,
                                                          float* __restrict__ ed_nrm,
                                                          const double* __restrict__ dv) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= J) return;
  const double* b = dv + 7 * j;
  const double* bgl = dv + 7 * J;
  const d3 n0 = {(double)ed_nrm[3 * j], (double)ed_nrm[3 * j + 1], (double)ed_nrm[3 * j + 2]};
  d3 rn = quat_apply(b[0], {b[1], b[2], b[3]}, n0);
  rn = quat_apply(bgl[0], {bgl[1], bgl[2], bgl[3]}, rn);
  const double nl = fmax(sqrt(dot(rn, rn)), 1e-12);
  ed_pts[3 * j] = (float)((double)ed_pts[3 * j] + b[4] + bgl[4]);
  ed_pts[3 * j + 1] = (float)((double)ed_pts[3 * j + 1] + b[5] + bgl[5]);
  ed_pts[3 * j + 2] = (float)((double)ed_pts[3 * j + 2] + b[6] + bgl[6]);
  ed_nrm[3 * j] = (float)(rn.x / nl);
  ed_nrm[3 * j + 1] = (float)(rn.y / nl);
  ed_nrm[3 * j + 2] = (float)(rn.z / nl);
}

// ---------------------------------------------------------------------------------------
void slm_set_error_text(const char* msg);   // slm_api.hip

struct slm_gf {
  slm_gf_config cfg{};
  std::vector<GfSlot> host;
  std::vector<size_t> cap;
  GfSlot* dev = nullptr;
};

#define GFCHK(expr)                                                       \
  do {                                                                    \
    hipError_t e_ = (expr);                                               \
    if (e_ != hipSuccess) {                                               \
      slm_set_error_text((std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
      return SLM_ERR_HIP;                                                 \
    }                                                                     \
  } while (0)

static int gf_fail(int code, const char* msg) {
  slm_set_error_text(msg);
  return code;
}

static void gf_enqueue_eval(slm_gf* g, GfSlot* slots, int n, int maxN, int maxReg, hipStream_t st) {
  const slm_gf_config& c = g->cfg;
  hipLaunchKernelGGL(k_gf_zero, dim3(32, n), dim3(256), 0, st, slots);
  if (c.use_data && maxN > 0)
    hipLaunchKernelGGL(k_gf_data, dim3((maxN + 255) / 256, n), dim3(256), 0, st, slots, c.w_data);
  if ((c.use_arap || c.use_rot || c.use_face) && maxReg > 0)
    hipLaunchKernelGGL(k_gf_reg, dim3((maxReg + 255) / 256, n), dim3(256), 0, st, slots, c.use_arap, c.w_arap,
                       c.use_rot, c.w_rot, c.use_face, c.w_face);
}

extern "C" {

int slm_gf_create(const slm_gf_config* cfg, slm_gf** out) {
  if (!cfg || !out || cfg->max_frames < 1 || cfg->num_iterations < 0 || (cfg->optimizer != 0 && cfg->optimizer != 1))
    return gf_fail(SLM_ERR_INVALID, "slm_gf_create: bad argument");
  if (slm_device_count() < 1) return gf_fail(SLM_ERR_NO_DEVICE, "slm_gf_create: no HIP device visible");
  slm_gf* g = new slm_gf();
  g->cfg = *cfg;
  g->host.assign(cfg->max_frames, GfSlot{});
  g->cap.assign(cfg->max_frames, 0);
  hipError_t e = hipMalloc((void**)&g->dev, sizeof(GfSlot) * cfg->max_frames);
  if (e == hipSuccess) e = hipMemset(g->dev, 0, sizeof(GfSlot) * cfg->max_frames);
  if (e != hipSuccess) {
    slm_set_error_text((std::string("slm_gf_create: ") + hipGetErrorString(e)).c_str());
    delete g;
    return SLM_ERR_HIP;
  }
  *out = g;
  return SLM_OK;
}

int slm_gf_destroy(slm_gf* g) {
  if (!g) return SLM_OK;
  for (GfSlot& s : g->host) {
    if (s.dv) (void)hipFree(s.dv);   // dv | grad | m1 | m2 | terms are one allocation
  }
  if (g->dev) (void)hipFree(g->dev);
  delete g;
  return SLM_OK;
}

int slm_gf_bind_frame(slm_gf* g, int32_t slot, const slm_gf_frame* fr, void* stream) {
  if (!g || !fr) return gf_fail(SLM_ERR_INVALID, "slm_gf_bind_frame: null argument");
  if (slot < 0 || slot >= (int)g->host.size()) return gf_fail(SLM_ERR_INVALID, "slm_gf_bind_frame: bad slot");
  const slm_frame& f = fr->base;
  if (f.K != SLM_K) return gf_fail(SLM_ERR_UNSUPPORTED, "slm_gf_bind_frame: num_neighbors must be 4");
  if (f.K_ED < 1 || f.K_ED > SLM_MAX_KED || f.N < 0 || f.J < 1 || f.H < 4 || f.W < 4)
    return gf_fail(SLM_ERR_INVALID, "slm_gf_bind_frame: bad sizes");
  if (!f.sf_points || !f.sf_knn_idx || !f.sf_knn_w || !f.ed_points || !f.ed_knn_idx || !f.tgt_points ||
      !f.tgt_norms || !f.index_map || (g->cfg.use_arap && !fr->ed_knn_w) ||
      (g->cfg.use_face && (!fr->ed_triangles || !fr->ed_triangle_areas)))
    return gf_fail(SLM_ERR_INVALID, "slm_gf_bind_frame: null device pointer");
  hipStream_t st = (hipStream_t)stream;
  GfSlot& s = g->host[slot];
  const size_t n = (size_t)(f.J + 1) * 7;
  if (n > g->cap[slot]) {
    if (s.dv) GFCHK(hipFree(s.dv));
    s.dv = nullptr;
    GFCHK(hipMalloc((void**)&s.dv, sizeof(double) * (4 * n + 8)));
    g->cap[slot] = n;
  }
  s.grad = s.dv + n;
  s.m1 = s.dv + 2 * n;
  s.m2 = s.dv + 3 * n;
  s.terms = s.dv + 4 * n;
  s.f = *fr;
  s.bound = 1;
  s.step = 0;
  GFCHK(hipMemcpyAsync(g->dev + slot, &s, sizeof(GfSlot), hipMemcpyHostToDevice, st));
  GFCHK(hipStreamSynchronize(st));
  hipLaunchKernelGGL(k_gf_init, dim3((n + 255) / 256), dim3(256), 0, st, g->dev, slot);
  GFCHK(hipGetLastError());
  return SLM_OK;
}

static int gf_dims(slm_gf* g, int first, int n, int* maxN, int* maxReg, int* maxP) {
  if (!g || first < 0 || n < 1 || first + n > (int)g->host.size())
    return gf_fail(SLM_ERR_INVALID, "slm_gf: slot range out of bounds");
  *maxN = *maxReg = *maxP = 0;
  for (int i = first; i < first + n; ++i) {
    const GfSlot& s = g->host[i];
    if (!s.bound) return gf_fail(SLM_ERR_UNBOUND, "slm_gf: slot used before slm_gf_bind_frame");
    *maxN = std::max(*maxN, s.f.base.N);
    int reg = std::max(s.f.base.J * s.f.base.K_ED, s.f.base.J + 1);
    if (g->cfg.use_face) reg = std::max(reg, s.f.n_triangles);
    *maxReg = std::max(*maxReg, reg);
    *maxP = std::max(*maxP, (s.f.base.J + 1) * 7);
  }
  return SLM_OK;
}

int slm_gf_run(slm_gf* g, int32_t n_frames, void* stream) {
  int maxN, maxReg, maxP;
  int rc = gf_dims(g, 0, n_frames, &maxN, &maxReg, &maxP);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  for (int it = 0; it < g->cfg.num_iterations; ++it) {
    gf_enqueue_eval(g, g->dev, n_frames, maxN, maxReg, st);
    hipLaunchKernelGGL(k_gf_step, dim3((maxP + 255) / 256, n_frames), dim3(256), 0, st, g->dev,
                       g->cfg.optimizer, g->cfg.lr, 1);
    hipLaunchKernelGGL(k_gf_advance, dim3(n_frames), dim3(64), 0, st, g->dev);
  }
  GFCHK(hipGetLastError());
  return SLM_OK;
}

int slm_gf_get_deform(slm_gf* g, int32_t slot, double* out, void* stream) {
  int maxN, maxReg, maxP;
  int rc = gf_dims(g, slot, 1, &maxN, &maxReg, &maxP);
  if (rc) return rc;
  if (!out) return gf_fail(SLM_ERR_INVALID, "slm_gf_get_deform: null output");
  GFCHK(hipMemcpyAsync(out, g->host[slot].dv, sizeof(double) * maxP, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return SLM_OK;
}

int slm_gf_loss_grad(slm_gf* g, int32_t slot, const double* dv, double* terms, double* grad, void* stream) {
  int maxN, maxReg, maxP;
  int rc = gf_dims(g, slot, 1, &maxN, &maxReg, &maxP);
  if (rc) return rc;
  if (!dv) return gf_fail(SLM_ERR_INVALID, "slm_gf_loss_grad: null dv");
  hipStream_t st = (hipStream_t)stream;
  const GfSlot& s = g->host[slot];
  GFCHK(hipMemcpyAsync(s.dv, dv, sizeof(double) * maxP, hipMemcpyDeviceToDevice, st));
  gf_enqueue_eval(g, g->dev + slot, 1, maxN, maxReg, st);
  hipLaunchKernelGGL(k_gf_step, dim3((maxP + 255) / 256, 1), dim3(256), 0, st, g->dev + slot, g->cfg.optimizer,
                     g->cfg.lr, 0);
  if (terms) GFCHK(hipMemcpyAsync(terms, s.terms, sizeof(double) * 5, hipMemcpyDeviceToDevice, st));
  if (grad) GFCHK(hipMemcpyAsync(grad, s.grad, sizeof(double) * maxP, hipMemcpyDeviceToDevice, st));
  GFCHK(hipGetLastError());
  return SLM_OK;
}

int slm_apply_update_gf(int32_t N, int32_t J, int32_t K, float* sf_points, float* sf_norms,
                        const int32_t* sf_knn_idx, const float* sf_knn_w, float* ed_points, float* ed_norms,
                        const double* deform, void* stream) {
  if (K != SLM_K) return gf_fail(SLM_ERR_UNSUPPORTED, "slm_apply_update_gf: num_neighbors must be 4");
  if (N < 0 || J < 1 || !ed_points || !ed_norms || !deform ||
      (N > 0 && (!sf_points || !sf_norms || !sf_knn_idx || !sf_knn_w)))
    return gf_fail(SLM_ERR_INVALID, "slm_apply_update_gf: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (N > 0)
    hipLaunchKernelGGL(k_gf_update_surfels, dim3((N + 255) / 256), dim3(256), 0, st, N, J, sf_points, sf_norms,
                       sf_knn_idx, sf_knn_w, ed_points, deform);
  hipLaunchKernelGGL(k_gf_update_nodes, dim3((J + 255) / 256), dim3(256), 0, st, J, ed_points, ed_norms, deform);
  GFCHK(hipGetLastError());
  return SLM_OK;
}

}  // extern "C"
