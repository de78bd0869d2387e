// slm_reg.hip -- ARAP and Rot regularisers (fixed sparsity, tiny): Jacobian pass and loss pass.
//
// ARAP (reference super/loss.py:408-455): r_{jk} = lam [R(q_k) d + b_k - d - b_j],
//   d = g_j - g_k, k in KNN_ED(j), un-weighted; row c has 6 non-zeros:
//   cols 7k+0..3 <- lam dR(q_k)d/dq [c,:], col 7k+4+c <- +lam, col 7j+4+c <- -lam.
// Rot (reference super/loss.py:480-499): r_j = lam (1 - |q_j|^2) evaluated in FLOAT32,
//   cols 7j+0..3 <- -2 lam q_j; its JtJ / jtl products are float32 too.
#include "slm_common.h"

__device__ __forceinline__ void load_beta(const double* beta, const double* delta, int j,
                                          double bb[7]) {
#pragma unroll
  for (int c = 0; c < 7; ++c) bb[c] = beta[7 * j + c];
  if (delta) {
#pragma unroll
    for (int c = 0; c < 7; ++c) bb[c] += delta[7 * j + c];
  }
}

__device__ __forceinline__ void arap_residual(const FrameDev& fd, const double* beta,
                                              const double* delta, int j, int k, double lam,
                                              double r[3], double bk[7], d3& d) {
  d = node_pos_pk(fd.node_pk, j) - node_pos_pk(fd.node_pk, k);   // g_j - g_k (node_pk carries g in float64)
  double bj[7];
  load_beta(beta, delta, k, bk);
  load_beta(beta, delta, j, bj);
  d3 t = quat_apply(bk[0], {bk[1], bk[2], bk[3]}, d);
  r[0] = lam * (t.x + bk[4] - d.x - bj[4]);
  r[1] = lam * (t.y + bk[5] - d.y - bj[5]);
  r[2] = lam * (t.z + bk[6] - d.z - bj[6]);
}

// float32 Rot residual, mirroring the reference's dtype (sequential f32 sum, no contraction)
__device__ __forceinline__ float rot_residual32(const double bb[7], float lam32, float q[4]) {
#pragma clang fp contract(off)   // keep mul and add separately rounded, like the reference's f32 ops
#pragma unroll
  for (int c = 0; c < 4; ++c) q[c] = (float)bb[c];
  float s = q[0] * q[0];
  s = s + q[1] * q[1];
  s = s + q[2] * q[2];
  s = s + q[3] * q[3];
  return lam32 * (1.0f - s);
}

// float32 products of the Rot Jacobian row (-2 lam q) with itself and with r
__device__ __forceinline__ void rot_products32(const float q[4], float lam32, float r, float jtj[4][4],
                                               float jtr[4]) {
#pragma clang fp contract(off)
  float jv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) jv[c] = (-lam32 * 2.0f) * q[c];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    jtr[a] = -(jv[a] * r);
#pragma unroll
    for (int b = 0; b < 4; ++b) jtj[a][b] = jv[a] * jv[b];
  }
}

__device__ __forceinline__ float sq32(float r) {
#pragma clang fp contract(off)
  return r * r;
}

// grid = (ceil(maxJ*K_ED / 256), n_frames): one thread per (node, neighbour slot);
// threads with slot 0 also do the node's Rot row.
__global__ void __launch_bounds__(256) k_reg_grad(const FrameDev* __restrict__ frames, int use_arap,
                                                   double lam_a, int use_rot, double lam_r) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || fd.st->stopped) return;
  const int Ke = fd.f.K_ED;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = t / Ke, slot = t % Ke;
  if (j >= fd.f.J) return;

  if (use_arap) {
    const int k = frame_in(fd).ed_knn_idx[j * Ke + slot];
    double r[3], bk[7];
    d3 d;
    arap_residual(fd, fd.beta, nullptr, j, k, lam_a, r, bk, d);
    double Jq[3][4];
    quat_jac(bk[0], {bk[1], bk[2], bk[3]}, d, Jq);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // the 6 non-zeros of residual row c
      int col[6] = {7 * k, 7 * k + 1, 7 * k + 2, 7 * k + 3, 7 * k + 4 + c, 7 * j + 4 + c};
      double val[6] = {lam_a * Jq[c][0], lam_a * Jq[c][1], lam_a * Jq[c][2], lam_a * Jq[c][3],
                       lam_a, -lam_a};
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        atomic_add_f64(fd.rhs + col[a], -val[a] * r[c]);
#pragma unroll
        for (int b = 0; b < 6; ++b)
          if (col[a] >= col[b]) atomic_add_f64(band_entry(fd, col[a], col[b]), val[a] * val[b]);
      }
    }
  }
  if (use_rot && slot == 0) {
    double bb[7];
    load_beta(fd.beta, nullptr, j, bb);
    float q[4];
    const float lam32 = (float)lam_r;
    const float r = rot_residual32(bb, lam32, q);
    float jtj[4][4], jtr[4];
    rot_products32(q, lam32, r, jtj, jtr);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      atomic_add_f64(fd.rhs + 7 * j + a, (double)jtr[a]);
#pragma unroll
      for (int b = 0; b <= a; ++b)
        atomic_add_f64(band_entry(fd, 7 * j + a, 7 * j + b), (double)jtj[a][b]);
    }
  }
}

// grid = (n_reg_blocks, n_frames); partial sums (arap, rot) per block after the data partials
__global__ void __launch_bounds__(256) k_reg_loss(const FrameDev* __restrict__ frames, int use_arap,
                                                   double lam_a, int use_rot, double lam_r,
                                                   int use_delta) {
  __shared__ double sm[16];
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || fd.st->stopped) return;
  const double* delta = use_delta ? fd.delta : nullptr;
  const int Ke = fd.f.K_ED;
  double sa = 0.0, sr = 0.0;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < fd.f.J * Ke; t += gridDim.x * blockDim.x) {
    const int j = t / Ke, slot = t % Ke;
    if (use_arap) {
      const int k = frame_in(fd).ed_knn_idx[j * Ke + slot];
      double r[3], bk[7];
      d3 d;
      arap_residual(fd, fd.beta, delta, j, k, lam_a, r, bk, d);
      sa += r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    }
    if (use_rot && slot == 0) {
      double bb[7];
      load_beta(fd.beta, delta, j, bb);
      float q[4];
      const float r = rot_residual32(bb, (float)lam_r, q);
      sr += (double)sq32(r);
    }
  }
  double a = block_sum(sa, sm);
  double b = block_sum(sr, sm);
  if (threadIdx.x == 0) {
    double* out = fd.loss_part + 2 * (size_t)fd.n_loss_part;
    out[2 * blockIdx.x] = a;
    out[2 * blockIdx.x + 1] = b;
  }
}

// What follows the solve of an LM iteration, in ONE launch (round 5; k_make_trial, k_reg_loss and k_dag_check were three
// launches of 4-5 us each, mostly launch latency): blocks [0, n_trial) pack the trial point beta + delta (node_pk_try: what the
// data term's loss pass reads), blocks [n_trial, n_trial + n_reg) evaluate the regularisers' loss there (they read beta and
// delta themselves), and block 0 of every slot first settles a timed-out task-graph launch (dag_check: the abort flag, kept
// with the ticket in slot 0's flags, marks the slots whose solve did not finish -- k_dag_check, slm_dag.hip).
// grid = (n_trial + n_reg, n_frames)
__global__ void __launch_bounds__(256) k_after_solve(const FrameDev* __restrict__ frames, int n_trial, int n_reg, int use_arap,
                                                      double lam_a, int use_rot, double lam_r, int dag_check) {
  __shared__ double sm[16];
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound) return;
  if (dag_check && blockIdx.x == 0 && threadIdx.x < 64) {
    const FrameDev& fd0 = frames[0];
    // (dag_check 2: an XCD-affine task-graph launch -- checked whether or not the abort flag is up, launch_front_solve_dag)
    if (fd0.bound && fd0.nd_ready && fd0.dag_flags && (dag_check == 2 || fd0.dag_flags[1] != 0) && fd.nd_ready && fd.dag_flags) {
      const int* px = fd.dag_flags.get() + 8 + fd.dag_n_tiles + fd.dag_n_pcols;
      bool done = true;
      for (int fi = threadIdx.x; fi < fd.n_fronts; fi += 64) {
        const NDFront& f = fd.fronts[fi];
        if (f.npt > 0 && px[(int)(f.linv_off / (SLM_NB * SLM_NB))] == 0) done = false;
      }
      if (!__all(done) && threadIdx.x == 0 && fd.st->chol_fail == 0) fd.st->chol_fail = 2;
    }
  }
  if (fd.st->stopped) return;
  if ((int)blockIdx.x < n_trial) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= fd.f.J) return;
    double bb[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) bb[c] = fd.beta[7 * j + c] + fd.delta[7 * j + c];
    pack_node(fd.node_pk_try + (size_t)SLM_NPK * j, bb, ld_state3(frame_in(fd).ed_points, (size_t)j, fd.f.state_f64));
    return;
  }
  const int b = blockIdx.x - n_trial;
  const int Ke = fd.f.K_ED;
  double sa = 0.0, sr = 0.0;
  for (int t = b * blockDim.x + threadIdx.x; t < fd.f.J * Ke; t += n_reg * blockDim.x) {
    const int j = t / Ke, slot = t % Ke;
    if (use_arap) {
      const int k = frame_in(fd).ed_knn_idx[j * Ke + slot];
      double r[3], bk[7];
      d3 d;
      arap_residual(fd, fd.beta, fd.delta, j, k, lam_a, r, bk, d);
      sa += r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    }
    if (use_rot && slot == 0) {
      double bb[7];
      load_beta(fd.beta, fd.delta, j, bb);
      float q[4];
      const float r = rot_residual32(bb, (float)lam_r, q);
      sr += (double)sq32(r);
    }
  }
  double a = block_sum(sa, sm);
  double bsum = block_sum(sr, sm);
  if (threadIdx.x == 0) {
    double* out = fd.loss_part + 2 * (size_t)fd.n_loss_part;
    out[2 * b] = a;
    out[2 * b + 1] = bsum;
  }
}

void launch_after_solve(const FrameDev* frames_dev, int n_frames, int maxJ, int n_reg, int use_arap, double lam_a, int use_rot,
                        double lam_r, int dag_check, hipStream_t st) {
  const int n_trial = maxJ > 0 ? (maxJ + 255) / 256 : 0;
  if (n_trial + n_reg <= 0) return;
  hipLaunchKernelGGL(k_after_solve, dim3(n_trial + n_reg, n_frames), dim3(256), 0, st, frames_dev, n_trial, n_reg, use_arap, lam_a,
                     use_rot, lam_r, dag_check);
}

void launch_reg_grad(const FrameDev* frames_dev, int n_frames, int maxJKe, int use_arap, double lam_a,
                     int use_rot, double lam_r, hipStream_t st) {
  if (maxJKe <= 0 || (!use_arap && !use_rot)) return;
  dim3 grid((maxJKe + 255) / 256, n_frames);
  hipLaunchKernelGGL(k_reg_grad, grid, dim3(256), 0, st, frames_dev, use_arap, lam_a, use_rot, lam_r);
}

void launch_reg_loss(const FrameDev* frames_dev, int n_frames, int n_blocks, int use_arap,
                     double lam_a, int use_rot, double lam_r, int use_delta, hipStream_t st) {
  dim3 grid(n_blocks, n_frames);
  hipLaunchKernelGGL(k_reg_loss, grid, dim3(256), 0, st, frames_dev, use_arap, lam_a, use_rot, lam_r,
                     use_delta);
}
