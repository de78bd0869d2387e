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
// Local rows: summed per workgroup in an LDS table, then one f64 atomic per entry and touched node; the global row is reduced per block first.
#include <string>
#include <vector>

#include "slm_sem.h"
#include "slm_lane.h"

__global__ void __launch_bounds__(256) k_gf_zero(GfSlot* __restrict__ slots) {
  GfSlotDev& s = gf_dev(slots)[blockIdx.y];
  if (!s.bound) return;
  const int n = (s.f.base.J + 1) * 7;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) s.grad[e] = 0.0;
  if (blockIdx.x == 0 && threadIdx.x < SLM_GF_NTERMS) s.terms[threadIdx.x] = 0.0;
  if (blockIdx.x == 1 % gridDim.x)
    for (int e = threadIdx.x; e < GF_PART_DOUBLES; e += blockDim.x) s.terms[SLM_GF_NTERMS + e] = 0.0;
}

// the spread block partials (slm_gf.h) -> grad[7J + 0..6] and terms[], copies summed in a fixed order, then cleared.
// which: bit0 = the entries of k_gf_data / k_gf_reg (0..13), bit1 = those of k_gf_morph (14, 15).  grid = (1, n_frames), 64 threads
__global__ void __launch_bounds__(64) k_gf_fold(GfSlot* __restrict__ slots, int which) {
  GfSlotDev& s = gf_dev(slots)[blockIdx.y];
  if (!s.bound) return;
  const int a = threadIdx.x;
  if (a >= 16) return;
  const bool mine = a < 14 ? (which & 1) != 0 : (which & 2) != 0;
  if (!mine) return;
  double* part = s.terms.get() + SLM_GF_NTERMS;
  double t = 0.0;
  for (int c = 0; c < GF_NCOPY; ++c) {
    t += part[16 * c + a];
    part[16 * c + a] = 0.0;
  }
  if (t == 0.0) return;
  const int J = s.f.base.J;
  if (a < 7) s.grad[7 * J + a] += t;
  else {
    // 7, 8 -> terms[3], [4]; 9, 10 -> [8], [9]; 11, 12, 13 -> [0], [1], [2]; 14, 15 -> [5], [6]
    const int map[9] = {3, 4, 8, 9, 0, 1, 2, 5, 6};
    s.terms[map[a - 7]] += t;
  }
}

// 4-tap gather of the target maps at the float pixel (u_, v_) (bilinear_sample, loss.py:9-80, zero fill):
// false when a tap is unmapped.  o / n = interpolated point / normal, d*u / d*v their derivatives along u / v
// (autograd through clamp(1 - |tap - x|): d|x|/dx = sign(x) with sign(0) = 0).
struct GfSample {
  d3 o, n, dou, dov, dnu, dnv;
  int rows[4];
  double wv[4];
};

__device__ __forceinline__ bool gf_sample(const FrameIn& f, double u_, double v_, GfSample& q) {
  const double fv = floor(v_), cv = ceil(v_), fu = floor(u_), cu = ceil(u_);
  const double nn[4] = {fv, fv, cv, cv}, mm[4] = {fu, cu, fu, cu};
  bool all_ok = true;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    q.rows[t] = f.index_map[(int)nn[t] * f.W + (int)mm[t]];
    all_ok = all_ok && q.rows[t] >= 0;
  }
  if (!all_ok) return false;
  d3 o = {0, 0, 0}, n = {0, 0, 0}, dou = {0, 0, 0}, dov = {0, 0, 0}, dnu = {0, 0, 0}, dnv = {0, 0, 0};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const double dn = nn[t] - v_, dm = mm[t] - u_;
    const double an = fmax(1.0 - fabs(dn), 0.0), am = fmax(1.0 - fabs(dm), 0.0);
    const float* tp = f.tgt_points + 3 * (size_t)q.rows[t];
    const float* tn = f.tgt_norms + 3 * (size_t)q.rows[t];
    const d3 Pt = {(double)tp[0], (double)tp[1], (double)tp[2]};
    const d3 Nt = {(double)tn[0], (double)tn[1], (double)tn[2]};
    const double wv = an * am;
    const double sn = dn > 0.0 ? 1.0 : (dn < 0.0 ? -1.0 : 0.0);
    const double smm = dm > 0.0 ? 1.0 : (dm < 0.0 ? -1.0 : 0.0);
    const double gu = an * smm, gvv = am * sn;
    q.wv[t] = wv;
    o = {o.x + Pt.x * wv, o.y + Pt.y * wv, o.z + Pt.z * wv};
    n = {n.x + Nt.x * wv, n.y + Nt.y * wv, n.z + Nt.z * wv};
    dou = {dou.x + Pt.x * gu, dou.y + Pt.y * gu, dou.z + Pt.z * gu};
    dov = {dov.x + Pt.x * gvv, dov.y + Pt.y * gvv, dov.z + Pt.z * gvv};
    dnu = {dnu.x + Nt.x * gu, dnu.y + Nt.y * gu, dnu.z + Nt.z * gu};
    dnv = {dnv.x + Nt.x * gvv, dnv.y + Nt.y * gvv, dnv.z + Nt.z * gvv};
  }
  q.o = o; q.n = n; q.dou = dou; q.dov = dov; q.dnu = dnu; q.dnv = dnv;
  return true;
}

// The optical flow (2,H,W float32: x then y displacement) at the float pixel (u, v), sampled the way
// F.grid_sample(flow, grid) does at deform_mesh.py / loss.py:318-323: the grid is float32, bilinear, zero padding,
// align_corners=False, i.e. position ((g + 1) * size - 1) / 2 in float32; fl = (flow_x, flow_y) and
// D = [[dfx/du, dfx/dv], [dfy/du, dfy/dv]] (the grid gradient of the same cell, what autograd returns).
__device__ __forceinline__ void gf_flow_sample(const float* __restrict__ flow, int H, int W, double u, double v,
                                               double fl[2], double D[4]) {
  const float gx = (float)(u * 2.0 / (double)W - 1.0), gy = (float)(v * 2.0 / (double)H - 1.0);
  const float ix = __fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), 0.5f * (float)W), 0.5f);
  const float iy = __fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), 0.5f * (float)H), 0.5f);
  const float xw = floorf(ix), yn = floorf(iy);
  const float w = __fsub_rn(ix, xw), e = __fsub_rn(1.f, w), n = __fsub_rn(iy, yn), sth = __fsub_rn(1.f, n);
  const int x0 = (int)xw, y0 = (int)yn;
  const bool okx0 = x0 >= 0 && x0 < W, okx1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool oky0 = y0 >= 0 && y0 < H, oky1 = y0 + 1 >= 0 && y0 + 1 < H;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float* fc = flow + (size_t)c * H * W;
    const float nw = (okx0 && oky0) ? fc[(size_t)y0 * W + x0] : 0.f;
    const float ne = (okx1 && oky0) ? fc[(size_t)y0 * W + x0 + 1] : 0.f;
    const float sw = (okx0 && oky1) ? fc[(size_t)(y0 + 1) * W + x0] : 0.f;
    const float se = (okx1 && oky1) ? fc[(size_t)(y0 + 1) * W + x0 + 1] : 0.f;
    float acc = __fmul_rn(nw, __fmul_rn(e, sth));
    acc = __fadd_rn(acc, __fmul_rn(ne, __fmul_rn(w, sth)));
    acc = __fadd_rn(acc, __fmul_rn(sw, __fmul_rn(e, n)));
    acc = __fadd_rn(acc, __fmul_rn(se, __fmul_rn(w, n)));
    fl[c] = (double)acc;
    D[2 * c + 0] = ((double)ne - (double)nw) * (double)sth + ((double)se - (double)sw) * (double)n;
    D[2 * c + 1] = ((double)sw - (double)nw) * (double)e + ((double)se - (double)ne) * (double)w;
  }
}

struct GfRegArgs {
  int use_arap, use_rot, use_face, pad;
  double lam_a, lam_r, lam_f;
};
__device__ __forceinline__ void gf_reg_body(GfSlotDev& s, const int bx, const GfRegArgs ra, double* sm);

// grid = (ceil(maxN/256) [+ the regulariser's blocks], n_frames)
// n_data_blocks: blocks [0, n_data_blocks) of a slot evaluate surfels; the blocks behind them, if any, run the node terms
// (gf_reg_body: independent work that only meets this kernel's in the gradient's atomics -- slm_gf_run's loop saves a launch)
// seg_mode: 0 none, 1 hard, 2 soft semantic weight on the squared residual (loss.py:379-399);
// pp_max > 0 (and no seg_mode): squared residuals >= pp_max are dropped (loss.py:369-370);
// use_morph: adds the back-propagation of the morphing term prepared by k_gf_morph (2: the kept count is still in the spread partials).
// KK = opt.num_neighbors of the launch's slots (deform_source is K-generic, super/deform_mesh.py:198-221)
#define GF_TAB 128   // LDS gradient table: slots per workgroup (power of two)
// EXTRA = false: the plain point-plane term only (no segmentation weight, clip, morphing or correspondence term) -- the
// instantiation the default options run: those code paths, and the registers they hold, are compiled out (round 6: the
// kernel ran at ONE wave per SIMD with everything in one body).
template <int KK, bool EXTRA>
__global__ void __launch_bounds__(256, (EXTRA || KK > 4) ? 3 : 4) k_gf_data(GfSlot* __restrict__ slots, int use_pp, double lam, int seg_mode_,
                                                     double pp_max_, int use_morph_, double w_morph, int corr_mode_,
                                                     double lam_c, int n_data_blocks, GfRegArgs ra) {
  const int seg_mode = EXTRA ? seg_mode_ : 0, use_morph = EXTRA ? use_morph_ : 0, corr_mode = EXTRA ? corr_mode_ : 0;
  const double pp_max = EXTRA ? pp_max_ : 0.0;
  __shared__ double sm[16];
  if ((int)blockIdx.x >= n_data_blocks) {
    GfSlotDev& sr = gf_dev(slots)[blockIdx.y];
    if (sr.bound) gf_reg_body(sr, blockIdx.x - n_data_blocks, ra, sm);
    return;
  }
  // The 256 surfels of a workgroup are neighbours on the image and share a few dozen ED nodes: their
  // gradient rows are summed in an LDS table keyed by node (ds_add_f64) and flushed with one global
  // atomic per entry and touched node -- about 20x fewer memory-side f64 atomics than one per surfel
  // and entry.  A slot taken by another node (direct-mapped, node & 127) falls back to global atomics.
  __shared__ int tkey[GF_TAB];
  __shared__ double tval[GF_TAB * 7];
  __shared__ double s_gval[4][16][7 * KK];   // gradient rows of 16 surfels of each wave, canonical slot order
  __shared__ int s_gid[4][16][KK];
  __shared__ double s_part[64];              // the four waves' sums of the 16 per-thread scalars
  __shared__ double s_kept;                  // the morphing term's kept count (use_morph == 2)
  GfSlotDev& s = gf_dev(slots)[blockIdx.y];
  if (!s.bound || s.f.base.K != KK) return;
  for (int t = threadIdx.x; t < GF_TAB; t += blockDim.x) tkey[t] = -1;
  for (int t = threadIdx.x; t < GF_TAB * 7; t += blockDim.x) tval[t] = 0.0;
  if (use_morph == 2 && threadIdx.x < 64) {
    const double v = wave_sum((double)s.terms[SLM_GF_NTERMS + 16 * threadIdx.x + 15]);
    if (threadIdx.x == 0) s_kept = v;
  }
  __syncthreads();
  const FrameIn& f = s.f.base;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int J = f.J;
  double gq[4] = {0, 0, 0, 0}, gb[3] = {0, 0, 0}, loss = 0.0, cnt = 0.0, lossc = 0.0, cntc = 0.0;
  bool has_grad = false;
  d3 cl = {0, 0, 0};          // dL/dT(p) of this surfel (the global rotation undone): what its neighbours' rows are formed from
  GfSkinLight<KK> k;
#pragma unroll
  for (int a = 0; a < KK; ++a) k.id[a] = 0;
  if (i >= s.shard_lo && i < s.shard_hi && (!s.f.sf_stable || s.f.sf_stable[i])) {
    gf_skin_light<KK>(s, i, k);
    const d3 P = k.P;
    const double fx = (double)f.fx, fy = (double)f.fy, cx = (double)f.cx, cy = (double)f.cy;
    const double Ze = P.z + 1e-8;
    // the forward divides by Z + 1e-8, and so does its derivative
    const d3 Pi0 = {fx / Ze, 0.0, -fx * P.x / (Ze * Ze)};
    const d3 Pi1 = {0.0, fy / Ze, -fy * P.y / (Ze * Ze)};
    d3 gP = {0, 0, 0};   // dL/dP of this surfel
    bool any = false;
    const double u_ = P.x * fx / Ze + cx, v_ = P.y * fy / Ze + cy;
    const double ur = rint(u_), vr = rint(v_);
    const int H = f.H, W = f.W;
    // valid_margin = 1 (loss.py:306-309)
    if (use_pp && vr >= 1.0 && vr < (double)(H - 2) && ur >= 1.0 && ur < (double)(W - 2)) {
      GfSample q;
      if (gf_sample(f, u_, v_, q)) {
        const d3 o = q.o, n = q.n, dou = q.dou, dov = q.dov, dnu = q.dnu, dnv = q.dnv;
        double conf[SLM_MAX_CLASSES] = {0, 0, 0, 0};
        const int C = (seg_mode && s.sem_bound) ? s.sem.num_classes : 0;
        for (int t = 0; t < 4; ++t)
          for (int c = 0; c < C; ++c) conf[c] += (double)s.sem.tgt_seg_conf[(size_t)q.rows[t] * C + c] * q.wv[t];
        const d3 e = P - o;
        const double r = dot(n, e);
        double wgt = 1.0;
        bool keep = true;
        if (C > 0) {
          // sampled trg.seg_conf is softmaxed again (loss.py:357); weights are detached
          double mx = conf[0];
          int am = 0;
          for (int c = 1; c < C; ++c)
            if (conf[c] > mx) {
              mx = conf[c];
              am = c;
            }
          if (seg_mode == 1) {
            wgt = (s.sem.sf_seg[i] == am) ? 1.0 : 0.0;
          } else {
            double q[SLM_MAX_CLASSES], den = 0.0;
            for (int c = 0; c < C; ++c) {
              q[c] = exp(conf[c] - mx);
              den += q[c];
            }
            // JSD(P, Q) = (KL(P|M) + KL(Q|M)) / 2, KL(P|Q) = sum P log(P / (Q + eps) + eps)  (utils.py:244-254)
            const double eps = 1e-13;
            double k1 = 0.0, k2 = 0.0;
            for (int c = 0; c < C; ++c) {
              const double pc = (double)s.sem.sf_seg_conf[(size_t)i * C + c], qc = q[c] / den;
              const double m = 0.5 * (pc + qc);
              k1 += pc * log(pc / (m + eps) + eps);
              k2 += qc * log(qc / (m + eps) + eps);
            }
            wgt = exp(-0.1 * (0.5 * (k1 + k2)));
          }
        } else if (pp_max > 0.0) {
          keep = (r * r) < pp_max;
        }
        if (keep) {
          loss = lam * wgt * r * r;
          cnt = 1.0;
          // c = dr/dP
          const double s0 = dot(e, dnu) - dot(n, dou), s1 = dot(e, dnv) - dot(n, dov);
          const double G = 2.0 * lam * wgt * r;
          gP = {G * (n.x + s0 * Pi0.x + s1 * Pi1.x), G * (n.y + s0 * Pi0.y + s1 * Pi1.y),
                G * (n.z + s0 * Pi0.z + s1 * Pi1.z)};
          any = true;
        }
      }
    }
    if (corr_mode && s.flow) {
      // flow-correspondence term (opt.sf_corr, deform_mesh.py:100-109 -> loss.py:293-345 with flow): the UNROUNDED
      // projection is moved by the flow sampled at it, validity is margin 1 on the moved float coordinates
      double fl[2], D[4];
      gf_flow_sample(s.flow, H, W, u_, v_, fl, D);
      const double uc = u_ + fl[0], vc = v_ + fl[1];
      if (vc >= 1.0 && vc < (double)(H - 2) && uc >= 1.0 && uc < (double)(W - 2)) {
        GfSample q;
        if (gf_sample(f, uc, vc, q)) {
          // d(u', v')/dP = (I + dflow/d(u,v)) Pi
          const d3 Au = {(1.0 + D[0]) * Pi0.x + D[1] * Pi1.x, (1.0 + D[0]) * Pi0.y + D[1] * Pi1.y,
                         (1.0 + D[0]) * Pi0.z + D[1] * Pi1.z};
          const d3 Av = {D[2] * Pi0.x + (1.0 + D[3]) * Pi1.x, D[2] * Pi0.y + (1.0 + D[3]) * Pi1.y,
                         D[2] * Pi0.z + (1.0 + D[3]) * Pi1.z};
          const d3 e = P - q.o;
          if (corr_mode == 1) {          // 'point-point': |P - o|^2
            lossc = lam_c * dot(e, e);
            const double s0 = -dot(e, q.dou), s1 = -dot(e, q.dov);
            const double G = 2.0 * lam_c;
            gP = {gP.x + G * (e.x + s0 * Au.x + s1 * Av.x), gP.y + G * (e.y + s0 * Au.y + s1 * Av.y),
                  gP.z + G * (e.z + s0 * Au.z + s1 * Av.z)};
          } else {                       // 'point-plane': (n.(P - o))^2
            const double r = dot(q.n, e);
            lossc = lam_c * r * r;
            const double s0 = dot(e, q.dnu) - dot(q.n, q.dou), s1 = dot(e, q.dnv) - dot(q.n, q.dov);
            const double G = 2.0 * lam_c * r;
            gP = {gP.x + G * (q.n.x + s0 * Au.x + s1 * Av.x), gP.y + G * (q.n.y + s0 * Au.y + s1 * Av.y),
                  gP.z + G * (q.n.z + s0 * Au.z + s1 * Av.z)};
          }
          cntc = 1.0;
          any = true;
        }
      }
    }
    if (use_morph && s.sem_bound) {
      // mean over the kept surfels (count from k_gf_morph, earlier in the stream: folded into terms[6] by k_gf_fold, or --
      // use_morph == 2, slm_gf_run's loop -- still in the 64 spread copies of entry 15: summed once per block, s_kept; a count,
      // exact in any order)
      const double2 mg = s.morph_g[i];
      const double kept = use_morph == 2 ? s_kept : (double)s.terms[6];
      if (kept > 0.0 && (mg.x != 0.0 || mg.y != 0.0)) {
        const double sc = w_morph / kept;
        gP = {gP.x + sc * (mg.x * Pi0.x + mg.y * Pi1.x), gP.y + sc * (mg.x * Pi0.y + mg.y * Pi1.y),
              gP.z + sc * (mg.x * Pi0.z + mg.y * Pi1.z)};
        any = true;
      }
    }
    if (any) {
      double jq[4];
      quat_jac_row(k.gw, k.gv, k.T, gP, jq);
#pragma unroll
      for (int a = 0; a < 4; ++a) gq[a] = jq[a];
      gb[0] = gP.x;
      gb[1] = gP.y;
      gb[2] = gP.z;
      cl = quat_apply_t(k.gw, k.gv, gP);
      has_grad = true;
    }
  }
  // ---- the local rows: the WAVE turns round (round 6).  One LDS atomic per surfel, neighbour and entry -- 28 per surfel,
  // most of them onto the few addresses the wave's surfels share -- serialised inside every instruction (k_gf_data was 8x
  // slower per surfel than the LM path's evaluation pass).  Now the gradient rows of 16 surfels at a time go to LDS in
  // canonical slot order and lane e < 7 KK owns ENTRY (slot e / 7, component e % 7): it walks the surfels, accumulates in
  // a register while the slot's node stays the same and adds to the workgroup's table when it changes -- one conflict-free
  // add per run of surfels with a common node instead of one conflicting add per surfel.
  {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long am = __ballot(has_grad);
    // WG entry-lane groups of 7 KK lanes share a round's 16 surfels (K = 4: lanes 0..27 walk surfels 0..7, lanes 28..55
    // surfels 8..15): half the steps per round for one more flush per lane and round
    constexpr int WG = (7 * KK <= 16) ? 4 : ((7 * KK <= 32) ? 2 : 1), WS = 16 / WG;
    const int grp = l / (7 * KK), el = l - 7 * KK * grp;   // group, entry inside the group
    const int slot = el / 7;
    double acc = 0.0;
    int prev = -1;
    auto flush = [&]() {
      if (prev < 0) return;
      const int ts = prev & (GF_TAB - 1);
      const int old = atomicCAS(&tkey[ts], -1, prev);
      if (old == -1 || old == prev) unsafeAtomicAdd(&tval[7 * ts + (el - 7 * slot)], acc);
      else atomic_add_f64(s.grad + 7 * prev + (el - 7 * slot), acc);
    };
    // K <= 4: every lane forms the quaternion parts of its K rows NOW, all 64 lanes at once (the node rows / positions are
    // read again -- cache hits -- and held: 8 K registers); only the staging goes 16 surfels at a time.  (Formed inside the
    // rounds by the 16 lanes of the round, the same instructions issued four times: 40 of the launch's 200 us at 8 C2 frames.)
    constexpr bool EAGER = KK <= 4;
    double jqa[EAGER ? KK : 1][4];
    if (EAGER && has_grad) {
      const FrameIn& f = s.f.base;
#pragma unroll
      for (int a = 0; a < (EAGER ? KK : 0); ++a) {
        const double* b = s.dv + 7 * k.id[a];
        const d3 g = ld_state3(f.ed_points, (size_t)k.id[a], f.state_f64);
        quat_jac_row(b[0], {b[1], b[2], b[3]}, k.p - g, cl, jqa[a]);
      }
    }
    if (am) {
#pragma unroll 1
      for (int sb = 0; sb < 4; ++sb) {
        const unsigned mask16 = (unsigned)((am >> (16 * sb)) & 0xFFFFull);
        if (mask16 == 0u) continue;                       // (uniform)
        if ((l >> 4) == sb && has_grad) {
          // this sub-batch's surfels form their rows now and write them straight to LDS (nothing per neighbour was held
          // across the sampling phase: the node rows / positions are read again -- cache hits)
          const FrameIn& f = s.f.base;
#pragma unroll
          for (int a = 0; a < KK; ++a) {
            double jq[4];
            if constexpr (EAGER) {
#pragma unroll
              for (int c = 0; c < 4; ++c) jq[c] = jqa[a][c];
            } else {
              const double* b = s.dv + 7 * k.id[a];
              const d3 g = ld_state3(f.ed_points, (size_t)k.id[a], f.state_f64);
              quat_jac_row(b[0], {b[1], b[2], b[3]}, k.p - g, cl, jq);
            }
            const double wk = k.w[a];
            // canonical slot of neighbour a: its rank among the surfel's node ids (two surfels with the same neighbour SET
            // have the same node in every slot, whatever the distance order of their KNN lists)
            int rank = 0;
#pragma unroll
            for (int b2 = 0; b2 < KK; ++b2) rank += (k.id[b2] < k.id[a]) ? 1 : 0;
            s_gid[w][l & 15][rank] = k.id[a];
            double* dst = &s_gval[w][l & 15][7 * rank];
            dst[0] = wk * jq[0]; dst[1] = wk * jq[1]; dst[2] = wk * jq[2]; dst[3] = wk * jq[3];
            dst[4] = wk * cl.x;  dst[5] = wk * cl.y;  dst[6] = wk * cl.z;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (grp < WG) {
          for (unsigned mm = mask16 & (((1u << WS) - 1u) << (WS * grp)); mm; mm &= mm - 1) {
            const int si = __builtin_ctz(mm);
            const int id = s_gid[w][si][slot];
            const double v = s_gval[w][si][el];
            if (id != prev) {
              flush();
              prev = id;
              acc = v;
            } else {
              acc += v;
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (grp < WG) flush();
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < GF_TAB * 7; t += blockDim.x) {
    const int node = tkey[t / 7];
    const double v = tval[t];
    if (node >= 0 && v != 0.0) atomic_add_f64(s.grad + 7 * node + t % 7, v);
  }
  // global row, loss and count: one butterfly over the 16 values of a thread per wave (col_reduce16, slm_lane.h: lane 4 a holds
  // the wave's sum of value a), the four waves' sums through LDS, then one atomic each (9 block_sums with two barriers each
  // before: 19 of the launch's 215 us at 8 C2 frames, tools/diag/gf_ablate_time.py)
  {
    double vals[16] = {gq[0], gq[1], gq[2], gq[3], gb[0], gb[1], gb[2], loss, cnt, lossc, cntc, 0.0, 0.0, 0.0, 0.0, 0.0};
    const double r = col_reduce16(vals);
    const int l = threadIdx.x & 63;
    if ((l & 3) == 0) s_part[(threadIdx.x >> 6) * 16 + (l >> 2)] = r;
    __syncthreads();
    const int a = threadIdx.x;
    if (a < (corr_mode ? 11 : 9)) {
      const double t = s_part[a] + s_part[16 + a] + s_part[32 + a] + s_part[48 + a];
      // (spread block partials, slm_gf.h: entries 0..6 global row, 7 / 8 point-plane loss / kept, 9 / 10 correspondence loss / kept)
      if (t != 0.0) atomic_add_f64(s.terms.get() + SLM_GF_NTERMS + 16 * (blockIdx.x % GF_NCOPY) + a, t);
    }
  }
}

// ARAP: one thread per (node, slot); Rot: one thread per row (J+1); face: one per triangle.
// grid = (ceil(max(J*K_ED, J+1, Tr)/256), n_frames)
// bx: the block's index among the regulariser's blocks (its own launch, or the tail blocks of k_gf_data's)
__device__ __forceinline__ void gf_reg_body(GfSlotDev& s, const int bx, const GfRegArgs ra, double* sm) {
  const int use_arap = ra.use_arap, use_rot = ra.use_rot, use_face = ra.use_face;
  const double lam_a = ra.lam_a, lam_r = ra.lam_r, lam_f = ra.lam_f;
  const FrameIn& f = s.f.base;
  const int J = f.J, Ke = f.K_ED;
  const int t = bx * blockDim.x + threadIdx.x;
  double la = 0.0, lr = 0.0, lf = 0.0;
  double gq[4] = {0, 0, 0, 0}, gb[3] = {0, 0, 0};   // global-row contributions of this thread
  if (use_arap && t < J * Ke) {
    const int j = t / Ke, k = f.ed_knn_idx[t];
    const d3 d = ld_state3(f.ed_points, (size_t)j, f.state_f64) - ld_state3(f.ed_points, (size_t)k, f.state_f64);
    const d3 d32 = {(double)(float)d.x, (double)(float)d.y, (double)(float)d.z};
    const double* bk = s.dv + 7 * k;
    const double* bj = s.dv + 7 * j;
    const d3 qv = {bk[1], bk[2], bk[3]};
    const d3 tr = quat_apply(bk[0], qv, d);
    const d3 r = {tr.x + bk[4] - d32.x - bj[4], tr.y + bk[5] - d32.y - bj[5], tr.z + bk[6] - d32.z - bj[6]};
    const double wjk = ld_state1(s.f.ed_knn_w, (size_t)t, f.state_f64);
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
      const d3 gn = ld_state3(f.ed_points, (size_t)iv[a], f.state_f64);
      loc[a] = {gn.x + b[4], gn.y + b[5], gn.z + b[6]};
      V[a] = quat_apply(gw, gv, loc[a]);   // + b_g cancels in the edge vectors
    }
    const d3 e1 = V[1] - V[0], e2 = V[2] - V[0];
    const d3 cr = cross(e1, e2);
    const double area = 0.5 * sqrt(dot(cr, cr) + 1e-13);
    const double da = area - ld_state1(s.f.ed_triangle_areas, (size_t)t, f.state_f64);
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
    // (spread block partials: 0..6 global row, 11 / 12 / 13 face / arap / rot)
    if (threadIdx.x == 0 && tt != 0.0)
      atomic_add_f64(s.terms.get() + SLM_GF_NTERMS + 16 * (bx % GF_NCOPY) + (a < 7 ? a : a + 4), tt);
  }
}
__global__ void __launch_bounds__(256) k_gf_reg(GfSlot* __restrict__ slots, GfRegArgs ra) {
  __shared__ double sm[16];
  GfSlotDev& s = gf_dev(slots)[blockIdx.y];
  if (!s.bound) return;
  gf_reg_body(s, blockIdx.x, ra, sm);
}

// grad[J] /= J, then torch.optim.SGD(momentum=0.9) or torch.optim.Adam step (float64).
// Also turns the morphing term's sum into the reference's weighted mean (NaN over an empty set).
// fold: bit 0 -- this launch also sums the spread block partials of k_gf_data / k_gf_reg (what k_gf_fold(which = 1) does as a
// launch of its own: slm_gf_run's loop saves that launch; the thread of a global-row entry sums its own 64 copies, threads
// 0..6 the loss terms');  bit 1 -- the launch OWNS the partials: it clears what it summed and ASSIGNS the loss terms (nothing
// else wrote them since the last k_gf_zero);  bit 2 -- it leaves the gradient zeroed for the next iteration (bits 1 + 2:
// slm_gf_run's loop needs no k_gf_zero between two iterations);  bit 3 -- the launch also owns the morphing term's partials
// (entries 14 / 15: k_gf_fold(which = 2) as a launch of its own otherwise): it assigns terms[5] / [6] from them, clears them and,
// with bit 2, resets the candidates flag terms[7] for the next iteration's k_gf_morph.
// step_off: optimiser steps of this run that s.step does not count yet (k_gf_advance adds them at the end of the run)
__global__ void __launch_bounds__(256) k_gf_step(GfSlot* __restrict__ slots, int optimizer, double lr,
                                                  int apply, int use_morph, double w_morph, int fold, int step_off) {
  GfSlotDev& s = gf_dev(slots)[blockIdx.y];
  if (!s.bound) return;
  const int J = s.f.base.J, n = (J + 1) * 7;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0 && use_morph) {
    double li = s.terms[5], kept = s.terms[6];
    if (fold & 8) {
      double* part = s.terms.get() + SLM_GF_NTERMS;
      li = 0.0;
      kept = 0.0;
      for (int c = 0; c < GF_NCOPY; ++c) {
        li += part[16 * c + 14];
        kept += part[16 * c + 15];
        part[16 * c + 14] = 0.0;
        part[16 * c + 15] = 0.0;
      }
      s.terms[6] = kept;
    }
    s.terms[5] = s.terms[7] != 0.0 ? (kept > 0.0 ? w_morph * li / kept : nan("")) : 0.0;
    if ((fold & 12) == 12) s.terms[7] = 0.0;
  }
  if (e >= n) return;
  double g = s.grad[e];
  const bool own = (fold & 2) != 0;
  if (fold & 1) {
    double* part = s.terms.get() + SLM_GF_NTERMS;
    if (e < 7) {   // entries 7..13 of the partials: point-plane loss / kept, correspondence loss / kept, face, arap, rot
      const int map[7] = {3, 4, 8, 9, 0, 1, 2};
      double t = 0.0;
      for (int c = 0; c < GF_NCOPY; ++c) {
        t += part[16 * c + 7 + e];
        if (own) part[16 * c + 7 + e] = 0.0;
      }
      if (own) s.terms[map[e]] = t;
      else if (t != 0.0) s.terms[map[e]] += t;
    }
    if (e >= 7 * J) {
      double t = 0.0;
      for (int c = 0; c < GF_NCOPY; ++c) {
        t += part[16 * c + (e - 7 * J)];
        if (own) part[16 * c + (e - 7 * J)] = 0.0;
      }
      g += t;
    }
  }
  if (e >= 7 * J) {
    g /= (double)J;
    s.grad[e] = g;
  }
  if (fold & 4) s.grad[e] = 0.0;
  if (!apply) return;
  const int t = s.step + 1 + step_off;
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

__global__ void k_gf_advance(GfSlot* __restrict__ slots, int inc) {
  GfSlotDev& s = gf_dev(slots)[blockIdx.x];
  if (s.bound && threadIdx.x == 0) s.step += inc;
}

__global__ void __launch_bounds__(256) k_gf_init(GfSlot* __restrict__ slots, int slot) {
  GfSlotDev& s = gf_dev(slots)[slot];
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
template <typename RT, int KK>
__global__ void __launch_bounds__(256) k_gf_update_surfels(int N, int J, RT* __restrict__ pts_,
                                                            RT* __restrict__ nrm_,
                                                            const int* __restrict__ knn_idx,
                                                            const RT* __restrict__ knn_w,
                                                            const RT* __restrict__ ed_pts,
                                                            const double* __restrict__ dv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  RT* pts = pts_ + 3 * (size_t)i - 3 * i;   // 64-bit row offsets below go through these bases
  RT* nrm = nrm_ + 3 * (size_t)i - 3 * i;
  const d3 p = {(double)pts[3 * i], (double)pts[3 * i + 1], (double)pts[3 * i + 2]};
  const d3 n0 = {(double)nrm[3 * i], (double)nrm[3 * i + 1], (double)nrm[3 * i + 2]};
  int id[KK];
  double w[KK];
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    id[k] = knn_idx[(size_t)KK * i + k];
    w[k] = (double)knn_w[(size_t)KK * i + k];
  }
  d3 T = {0, 0, 0}, Nn = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < KK; ++k) {
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
  pts[3 * i] = (RT)(T.x + bgl[4]);
  pts[3 * i + 1] = (RT)(T.y + bgl[5]);
  pts[3 * i + 2] = (RT)(T.z + bgl[6]);
  nrm[3 * i] = (RT)(Nn.x / nl);
  nrm[3 * i + 1] = (RT)(Nn.y / nl);
  nrm[3 * i + 2] = (RT)(Nn.z / nl);
}

template <typename RT>
__global__ void __launch_bounds__(256) k_gf_update_nodes(int J, RT* __restrict__ ed_pts,
                                                          RT* __restrict__ ed_nrm,
                                                          const double* __restrict__ dv) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= J) return;
  const double* b = dv + 7 * j;
  const double* bgl = dv + 7 * J;
  const d3 n0 = {(double)ed_nrm[3 * j], (double)ed_nrm[3 * j + 1], (double)ed_nrm[3 * j + 2]};
  d3 rn = quat_apply(b[0], {b[1], b[2], b[3]}, n0);
  rn = quat_apply(bgl[0], {bgl[1], bgl[2], bgl[3]}, rn);
  const double nl = fmax(sqrt(dot(rn, rn)), 1e-12);
  ed_pts[3 * j] = (RT)((double)ed_pts[3 * j] + b[4] + bgl[4]);
  ed_pts[3 * j + 1] = (RT)((double)ed_pts[3 * j + 1] + b[5] + bgl[5]);
  ed_pts[3 * j + 2] = (RT)((double)ed_pts[3 * j + 2] + b[6] + bgl[6]);
  ed_nrm[3 * j] = (RT)(rn.x / nl);
  ed_nrm[3 * j + 1] = (RT)(rn.y / nl);
  ed_nrm[3 * j + 2] = (RT)(rn.z / nl);
}

// ---------------------------------------------------------------------------------------
void slm_set_error_text(const char* msg);   // slm_api.hip

// one instantiation of the per-surfel kernels per opt.num_neighbors (1..8)
#define GF_K_DISPATCH(K, ...)                                          \
  switch (K) {                                                         \
    case 1: { constexpr int KK = 1; __VA_ARGS__; break; }                     \
    case 2: { constexpr int KK = 2; __VA_ARGS__; break; }                     \
    case 3: { constexpr int KK = 3; __VA_ARGS__; break; }                     \
    case 4: { constexpr int KK = 4; __VA_ARGS__; break; }                     \
    case 5: { constexpr int KK = 5; __VA_ARGS__; break; }                     \
    case 6: { constexpr int KK = 6; __VA_ARGS__; break; }                     \
    case 7: { constexpr int KK = 7; __VA_ARGS__; break; }                     \
    case 8: { constexpr int KK = 8; __VA_ARGS__; break; }                     \
    default: break;                                                    \
  }

struct slm_gf {
  int batch_K = SLM_K;       // num_neighbors of the slots of the launch being enqueued (gf_dims: they must agree)
  slm_gf_config cfg{};
  std::vector<GfSlot> host;
  std::vector<size_t> cap;
  std::vector<SemScratch> sem;
  GfSlot* dev = nullptr;
  int rank = 0, world = 1;   // surfel sharding of every slot (slm_gf_set_shard)
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

// pass 1: zero the gradient / terms, then the morphing term's per-surfel pass (sum, count)
static void gf_enqueue_morph(slm_gf* g, GfSlot* slots, int n, int maxN, hipStream_t st) {
  hipLaunchKernelGGL(k_gf_zero, dim3(32, n), dim3(256), 0, st, slots);
  if (g->cfg.use_bn_morph) {
    launch_gf_morph(slots, n, maxN, st);
    hipLaunchKernelGGL(k_gf_fold, dim3(1, n), dim3(64), 0, st, slots, 2);   // terms[5], [6]: what the back-propagation divides by
  }
}

// pass 2: point-plane (+ morphing back-propagation, needs the GLOBAL kept count in terms[6]) and
// the node terms (on rank 0 only when the surfels are sharded: the caller sums the partials)
static void gf_enqueue_losses(slm_gf* g, GfSlot* slots, int n, int maxN, int maxReg, hipStream_t st, bool fold = true, bool morph_in_partials = false) {
  const slm_gf_config& c = g->cfg;
  const int use_pp = (c.use_data || c.seg_mode) ? 1 : 0;   // either flag enables the term (deform_mesh.py:81)
  const bool data = (use_pp || c.use_bn_morph || c.corr_mode) && maxN > 0;
  const bool reg = g->rank == 0 && (c.use_arap || c.use_rot || c.use_face) && maxReg > 0;
  const GfRegArgs ra = {c.use_arap, c.use_rot, c.use_face, 0, c.w_arap, c.w_rot, c.w_face};
  const int nd = (maxN + 255) / 256, nr = reg ? (maxReg + 255) / 256 : 0;
  if (data) {
    // the node terms ride on this launch as its tail blocks
    const bool extra = c.seg_mode || c.use_bn_morph || c.corr_mode || c.pp_max > 0.0;
    if (extra) {
      GF_K_DISPATCH(g->batch_K, hipLaunchKernelGGL((k_gf_data<KK, true>), dim3(nd + nr, n), dim3(256), 0, st, slots, use_pp, c.w_data,
                                                   c.seg_mode, c.seg_mode ? 0.0 : c.pp_max, c.use_bn_morph ? (morph_in_partials ? 2 : 1) : 0, c.w_bn_morph, c.corr_mode, c.w_corr,
                                                   nd, ra));
    } else {
      GF_K_DISPATCH(g->batch_K, hipLaunchKernelGGL((k_gf_data<KK, false>), dim3(nd + nr, n), dim3(256), 0, st, slots, use_pp, c.w_data,
                                                   0, 0.0, 0, 0.0, 0, 0.0, nd, ra));
    }
  } else if (reg) {
    hipLaunchKernelGGL(k_gf_reg, dim3(nr, n), dim3(256), 0, st, slots, ra);
  }
  if (fold) hipLaunchKernelGGL(k_gf_fold, dim3(1, n), dim3(64), 0, st, slots, 1);   // (else the caller's k_gf_step folds)
}

static void gf_enqueue_eval(slm_gf* g, GfSlot* slots, int n, int maxN, int maxReg, hipStream_t st, bool fold = true) {
  gf_enqueue_morph(g, slots, n, maxN, st);
  gf_enqueue_losses(g, slots, n, maxN, maxReg, st, fold);
}

template <typename RT>
static int apply_update_gf_t(int32_t N, int32_t J, int32_t K, RT* sf_points, RT* sf_norms,
                             const int32_t* sf_knn_idx, const RT* sf_knn_w, RT* ed_points, RT* ed_norms,
                             const double* deform, void* stream) {
  if (K < 1 || K > 8) return gf_fail(SLM_ERR_UNSUPPORTED, "slm_apply_update_gf: num_neighbors must be in 1..8");
  if (N < 0 || J < 1 || !ed_points || !ed_norms || !deform ||
      (N > 0 && (!sf_points || !sf_norms || !sf_knn_idx || !sf_knn_w)))
    return gf_fail(SLM_ERR_INVALID, "slm_apply_update_gf: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (N > 0)
    GF_K_DISPATCH(K, hipLaunchKernelGGL((k_gf_update_surfels<RT, KK>), dim3((N + 255) / 256), dim3(256), 0, st, N, J, sf_points, sf_norms,
                                        sf_knn_idx, sf_knn_w, (const RT*)ed_points, deform));
  hipLaunchKernelGGL(k_gf_update_nodes<RT>, dim3((J + 255) / 256), dim3(256), 0, st, J, ed_points, ed_norms, deform);
  GFCHK(hipGetLastError());
  return SLM_OK;
}

extern "C" {

int slm_gf_create(const slm_gf_config* cfg, slm_gf** out) {
  if (!cfg || !out || cfg->max_frames < 1 || cfg->num_iterations < 0 || (cfg->optimizer != 0 && cfg->optimizer != 1) ||
      cfg->seg_mode < 0 || cfg->seg_mode > 2 || cfg->corr_mode < 0 || cfg->corr_mode > 2)
    return gf_fail(SLM_ERR_INVALID, "slm_gf_create: bad argument");
  if (slm_device_count() < 1) return gf_fail(SLM_ERR_NO_DEVICE, "slm_gf_create: no HIP device visible");
  slm_gf* g = new slm_gf();
  g->cfg = *cfg;
  g->host.assign(cfg->max_frames, GfSlot{});
  g->cap.assign(cfg->max_frames, 0);
  g->sem.assign(cfg->max_frames, SemScratch());
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
  for (SemScratch& sc : g->sem) sem_free(sc);
  if (g->dev) (void)hipFree(g->dev);
  delete g;
  return SLM_OK;
}

int slm_gf_bind_frame(slm_gf* g, int32_t slot, const slm_gf_frame* fr, void* stream) {
  if (!g || !fr) return gf_fail(SLM_ERR_INVALID, "slm_gf_bind_frame: null argument");
  if (slot < 0 || slot >= (int)g->host.size()) return gf_fail(SLM_ERR_INVALID, "slm_gf_bind_frame: bad slot");
  const slm_frame& f = fr->base;
  if (f.K < 1 || f.K > 8) return gf_fail(SLM_ERR_UNSUPPORTED, "slm_gf_bind_frame: num_neighbors must be in 1..8");
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
    GFCHK(hipMalloc((void**)&s.dv, sizeof(double) * (4 * n + SLM_GF_NTERMS + GF_PART_DOUBLES)));   // ... | terms | spread block partials
    g->cap[slot] = n;
  }
  s.grad = s.dv + n;
  s.m1 = s.dv + 2 * n;
  s.m2 = s.dv + 3 * n;
  s.terms = s.dv + 4 * n;
  s.f = *fr;
  s.bound = 1;
  s.step = 0;
  s.sem_bound = 0;   // semantic inputs and the flow belong to the frame: bind them again
  s.flow = nullptr;
  s.shard_lo = (int32_t)((int64_t)f.N * g->rank / g->world);
  s.shard_hi = (int32_t)((int64_t)f.N * (g->rank + 1) / g->world);
  GFCHK(hipMemcpyAsync(g->dev + slot, &s, sizeof(GfSlot), hipMemcpyHostToDevice, st));
  GFCHK(hipStreamSynchronize(st));
  hipLaunchKernelGGL(k_gf_init, dim3((n + 255) / 256), dim3(256), 0, st, g->dev, slot);
  GFCHK(hipGetLastError());
  return SLM_OK;
}

int slm_gf_bind_semantic(slm_gf* g, int32_t slot, const slm_gf_semantic* sem, int32_t* edge_counts_host,
                         void* stream) {
  if (!g || !sem) return gf_fail(SLM_ERR_INVALID, "slm_gf_bind_semantic: null argument");
  if (slot < 0 || slot >= (int)g->host.size()) return gf_fail(SLM_ERR_INVALID, "slm_gf_bind_semantic: bad slot");
  GfSlot& s = g->host[slot];
  if (!s.bound) return gf_fail(SLM_ERR_UNBOUND, "slm_gf_bind_semantic: slm_gf_bind_frame first");
  if (sem->num_classes < 1 || sem->num_classes > SLM_MAX_CLASSES)
    return gf_fail(SLM_ERR_UNSUPPORTED, "slm_gf_bind_semantic: num_classes must be 1..4");
  const slm_frame& f = s.f.base;
  const bool need_pp = g->cfg.seg_mode != 0, need_morph = g->cfg.use_bn_morph != 0;
  if ((f.N > 0 && !sem->sf_seg) || (need_pp && ((f.N > 0 && !sem->sf_seg_conf) || (f.T > 0 && !sem->tgt_seg_conf))) ||
      (need_morph && (!sem->img_seg_conf || !sem->img_seg)))
    return gf_fail(SLM_ERR_INVALID, "slm_gf_bind_semantic: null device pointer");
  hipStream_t st = (hipStream_t)stream;
  SemScratch& sc = g->sem[slot];
  s.sem = *sem;
  for (int c = 0; c <= SLM_MAX_CLASSES; ++c) s.edge_off[c] = 0;
  if (need_morph) {
    GFCHK(sem_extract_edges(sc, *sem, f.H, f.W, s.edge_off, st));
    GFCHK(sem_ensure_morph(sc, f.N));
  }
  s.edge_xy = sc.edge_xy;
  s.morph_g = sc.morph_g;
  s.sem_bound = 1;
  if (edge_counts_host)
    for (int c = 0; c < sem->num_classes; ++c) edge_counts_host[c] = s.edge_off[c + 1] - s.edge_off[c];
  GFCHK(hipMemcpyAsync(g->dev + slot, &s, sizeof(GfSlot), hipMemcpyHostToDevice, st));
  GFCHK(hipStreamSynchronize(st));
  return SLM_OK;
}

int slm_gf_bind_flow(slm_gf* g, int32_t slot, const float* flow, void* stream) {
  if (!g || !flow) return gf_fail(SLM_ERR_INVALID, "slm_gf_bind_flow: null argument");
  if (slot < 0 || slot >= (int)g->host.size()) return gf_fail(SLM_ERR_INVALID, "slm_gf_bind_flow: bad slot");
  GfSlot& s = g->host[slot];
  if (!s.bound) return gf_fail(SLM_ERR_UNBOUND, "slm_gf_bind_flow: slm_gf_bind_frame first");
  hipStream_t st = (hipStream_t)stream;
  s.flow = flow;
  GFCHK(hipMemcpyAsync(g->dev + slot, &s, sizeof(GfSlot), hipMemcpyHostToDevice, st));
  GFCHK(hipStreamSynchronize(st));
  return SLM_OK;
}

int slm_gf_get_edge_points(slm_gf* g, int32_t slot, int32_t class_id, float* xy_out, int32_t max_points,
                           void* stream) {
  if (!g || slot < 0 || slot >= (int)g->host.size() || !xy_out)
    return gf_fail(SLM_ERR_INVALID, "slm_gf_get_edge_points: bad argument");
  const GfSlot& s = g->host[slot];
  if (!s.bound || !s.sem_bound) return gf_fail(SLM_ERR_UNBOUND, "slm_gf_get_edge_points: no semantic inputs bound");
  if (class_id < 0 || class_id >= s.sem.num_classes)
    return gf_fail(SLM_ERR_INVALID, "slm_gf_get_edge_points: bad class");
  const int n = s.edge_off[class_id + 1] - s.edge_off[class_id];
  if (n > max_points) return gf_fail(SLM_ERR_INVALID, "slm_gf_get_edge_points: output too small");
  if (n > 0)
    GFCHK(hipMemcpyAsync(xy_out, s.edge_xy + s.edge_off[class_id], sizeof(float2) * n, hipMemcpyDeviceToDevice,
                         (hipStream_t)stream));
  return SLM_OK;
}

static int gf_dims(slm_gf* g, int first, int n, int* maxN, int* maxReg, int* maxP) {
  if (!g || first < 0 || n < 1 || first + n > (int)g->host.size())
    return gf_fail(SLM_ERR_INVALID, "slm_gf: slot range out of bounds");
  *maxN = *maxReg = *maxP = 0;
  for (int i = first; i < first + n; ++i) {
    const GfSlot& s = g->host[i];
    if (!s.bound) return gf_fail(SLM_ERR_UNBOUND, "slm_gf: slot used before slm_gf_bind_frame");
    if ((g->cfg.seg_mode || g->cfg.use_bn_morph) && !s.sem_bound)
      return gf_fail(SLM_ERR_UNBOUND, "slm_gf: semantic terms enabled but slm_gf_bind_semantic was not called");
    if (g->cfg.corr_mode && !s.flow)
      return gf_fail(SLM_ERR_UNBOUND, "slm_gf: corr_mode set but slm_gf_bind_flow was not called");
    if (s.f.base.K != g->host[first].f.base.K)
      return gf_fail(SLM_ERR_UNSUPPORTED, "slm_gf: the frames of one batch must have the same num_neighbors");
    *maxN = std::max(*maxN, s.f.base.N);
    int reg = std::max(s.f.base.J * s.f.base.K_ED, s.f.base.J + 1);
    if (g->cfg.use_face) reg = std::max(reg, s.f.n_triangles);
    *maxReg = std::max(*maxReg, reg);
    *maxP = std::max(*maxP, (s.f.base.J + 1) * 7);
  }
  g->batch_K = g->host[first].f.base.K;
  return SLM_OK;
}

int slm_gf_set_shard(slm_gf* g, int32_t rank, int32_t world) {
  if (!g || world < 1 || rank < 0 || rank >= world) return gf_fail(SLM_ERR_INVALID, "slm_gf_set_shard: bad rank/world");
  g->rank = rank;
  g->world = world;
  for (GfSlot& s : g->host) s.bound = 0;   // shard bounds are fixed at bind time
  hipError_t e = hipMemset(g->dev, 0, sizeof(GfSlot) * g->host.size());
  if (e != hipSuccess) return gf_fail(SLM_ERR_HIP, hipGetErrorString(e));
  return SLM_OK;
}

int slm_gf_eval_morph(slm_gf* g, int32_t n_frames, void* stream) {
  int maxN, maxReg, maxP;
  int rc = gf_dims(g, 0, n_frames, &maxN, &maxReg, &maxP);
  if (rc) return rc;
  gf_enqueue_morph(g, g->dev, n_frames, maxN, (hipStream_t)stream);
  GFCHK(hipGetLastError());
  return SLM_OK;
}

int slm_gf_eval_losses(slm_gf* g, int32_t n_frames, void* stream) {
  int maxN, maxReg, maxP;
  int rc = gf_dims(g, 0, n_frames, &maxN, &maxReg, &maxP);
  if (rc) return rc;
  gf_enqueue_losses(g, g->dev, n_frames, maxN, maxReg, (hipStream_t)stream);
  GFCHK(hipGetLastError());
  return SLM_OK;
}

int slm_gf_step(slm_gf* g, int32_t n_frames, void* stream) {
  int maxN, maxReg, maxP;
  int rc = gf_dims(g, 0, n_frames, &maxN, &maxReg, &maxP);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_gf_step, dim3((maxP + 255) / 256, n_frames), dim3(256), 0, st, g->dev, g->cfg.optimizer,
                     g->cfg.lr, 1, g->cfg.use_bn_morph, g->cfg.w_bn_morph, 0, 0);
  hipLaunchKernelGGL(k_gf_advance, dim3(n_frames), dim3(64), 0, st, g->dev, 1);
  GFCHK(hipGetLastError());
  return SLM_OK;
}

int slm_gf_get_partial(slm_gf* g, int32_t slot, double* out, void* stream) {
  int maxN, maxReg, maxP;
  int rc = gf_dims(g, slot, 1, &maxN, &maxReg, &maxP);
  if (rc) return rc;
  if (!out) return gf_fail(SLM_ERR_INVALID, "slm_gf_get_partial: null output");
  const GfSlot& s = g->host[slot];
  hipStream_t st = (hipStream_t)stream;
  GFCHK(hipMemcpyAsync(out, s.grad, sizeof(double) * maxP, hipMemcpyDeviceToDevice, st));
  GFCHK(hipMemcpyAsync(out + maxP, s.terms, sizeof(double) * SLM_GF_NTERMS, hipMemcpyDeviceToDevice, st));
  return SLM_OK;
}

int slm_gf_set_partial(slm_gf* g, int32_t slot, const double* in, void* stream) {
  int maxN, maxReg, maxP;
  int rc = gf_dims(g, slot, 1, &maxN, &maxReg, &maxP);
  if (rc) return rc;
  if (!in) return gf_fail(SLM_ERR_INVALID, "slm_gf_set_partial: null input");
  const GfSlot& s = g->host[slot];
  hipStream_t st = (hipStream_t)stream;
  GFCHK(hipMemcpyAsync(s.grad, in, sizeof(double) * maxP, hipMemcpyDeviceToDevice, st));
  GFCHK(hipMemcpyAsync(s.terms, in + maxP, sizeof(double) * SLM_GF_NTERMS, hipMemcpyDeviceToDevice, st));
  return SLM_OK;
}

int slm_gf_run(slm_gf* g, int32_t n_frames, void* stream) {
  int maxN, maxReg, maxP;
  int rc = gf_dims(g, 0, n_frames, &maxN, &maxReg, &maxP);
  if (rc) return rc;
  if (g->world > 1)
    return gf_fail(SLM_ERR_UNSUPPORTED,
                   "slm_gf_run: surfels are sharded; drive slm_gf_eval_morph / eval_losses / step with an "
                   "all-reduce of slm_gf_get_partial between them");
  hipStream_t st = (hipStream_t)stream;
  // An iteration is TWO launches: the losses (k_gf_data with the node terms as its tail blocks) and the step, which folds the
  // block partials, assigns the loss terms and leaves gradient and partials zeroed for the next iteration; k_gf_zero only
  // runs in front of the first.  With the morphing term THREE: k_gf_morph in front -- its kept count stays in the spread
  // partials, k_gf_data sums the 64 copies itself and the step folds them with the rest (k_gf_zero + k_gf_morph + k_gf_fold +
  // k_gf_data + k_gf_step before: five launches of 5-85 us at configs[4]'s size).  The step counter advances once, behind the loop.
  const bool morph = g->cfg.use_bn_morph != 0;
  const int n_it = g->cfg.num_iterations;
  for (int it = 0; it < n_it; ++it) {
    if (it == 0) hipLaunchKernelGGL(k_gf_zero, dim3(32, n_frames), dim3(256), 0, st, g->dev);
    if (morph) launch_gf_morph(g->dev, n_frames, maxN, st);
    gf_enqueue_losses(g, g->dev, n_frames, maxN, maxReg, st, false, morph);   // (the step folds)
    const int fold = (it + 1 < n_it ? 7 : 3) | (morph ? 8 : 0);
    hipLaunchKernelGGL(k_gf_step, dim3((maxP + 255) / 256, n_frames), dim3(256), 0, st, g->dev,
                       g->cfg.optimizer, g->cfg.lr, 1, g->cfg.use_bn_morph, g->cfg.w_bn_morph, fold, it);
  }
  if (n_it > 0) hipLaunchKernelGGL(k_gf_advance, dim3(n_frames), dim3(64), 0, st, g->dev, n_it);
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
                     g->cfg.lr, 0, g->cfg.use_bn_morph, g->cfg.w_bn_morph, 0, 0);
  if (terms) GFCHK(hipMemcpyAsync(terms, s.terms, sizeof(double) * SLM_GF_NTERMS, hipMemcpyDeviceToDevice, st));
  if (grad) GFCHK(hipMemcpyAsync(grad, s.grad, sizeof(double) * maxP, hipMemcpyDeviceToDevice, st));
  GFCHK(hipGetLastError());
  return SLM_OK;
}

int slm_apply_update_gf(int32_t N, int32_t J, int32_t K, float* sf_points, float* sf_norms,
                        const int32_t* sf_knn_idx, const float* sf_knn_w, float* ed_points, float* ed_norms,
                        const double* deform, void* stream) {
  return apply_update_gf_t<float>(N, J, K, sf_points, sf_norms, sf_knn_idx, sf_knn_w, ed_points, ed_norms, deform, stream);
}

int slm_apply_update_gf_f64(int32_t N, int32_t J, int32_t K, double* sf_points, double* sf_norms,
                            const int32_t* sf_knn_idx, const double* sf_knn_w, double* ed_points, double* ed_norms,
                            const double* deform, void* stream) {
  return apply_update_gf_t<double>(N, J, K, sf_points, sf_norms, sf_knn_idx, sf_knn_w, ed_points, ed_norms, deform, stream);
}

}  // extern "C"
