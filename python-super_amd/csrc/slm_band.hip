// slm_band.hip -- damped normal equations (JtJ + uI) delta = jtl as a block-banded
// float64 Cholesky (replaces the reference's dense torch.linalg.cholesky +
// cholesky_solve on a (7J)^2 matrix, super/LM.py:37-51,97-100).
//
// Storage: lower band in NB x NB tiles, column-major inside a tile; tile (r,c),
// c <= r <= c+wb, at band[(c*(wb+1) + (r-c)) * NB*NB].  The ED graph couples only
// nearby nodes, so wb << nt (half-bandwidth ~156 node blocks at J=2k).
//
// Per tile column c (right-looking):
//   k_panel(c)  block d: factor A(c,c)+uI = L L^T and form L^-1 in LDS (every block,
//               redundantly); d=0 stores L^-1 and forward-substitutes y_c = L^-1 b_c;
//               d>=1: L(c+d,c) = A(c+d,c) L^-T on the f64 MFMA (v_mfma_f64_16x16x4_f64).
//   k_trail(c)  A(r,s) -= L(r,c) L(s,c)^T for the wb x wb window (MFMA); b_s -= L(s,c) y_c.
// Back substitution k_backsub(c), c = nt-1..0: x_c = L_cc^-T y_c, y_(c-d) -= L(c,c-d)^T x_c.
// A non-positive pivot sets st->chol_fail (reference: RuntimeError -> "Solver failed").
#include "slm_tile.h"

// ---------------------------------------------------------------------------------
// Tile half-bandwidth from the KNN tables: max over coupled node pairs (a >= b) of
// tile(7a+6) - tile(7b).  Surfel tuples couple all pairs among their K nodes
// (loss.py:277-288); ARAP couples (j, k) (loss.py:414-426).
__global__ void __launch_bounds__(256) k_bandwidth(slm_frame f, int* __restrict__ out) {
  int wmax = 0;
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < f.N; i += stride) {
    int lo = f.sf_knn_idx[(size_t)f.K * i], hi = lo;
    for (int k = 1; k < f.K; ++k) {   // (K = num_neighbors, any value)
      const int id = f.sf_knn_idx[(size_t)f.K * i + k];
      lo = min(lo, id);
      hi = max(hi, id);
    }
    wmax = max(wmax, (7 * hi + 6) / NB - (7 * lo) / NB);
  }
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < f.J * f.K_ED; t += stride) {
    const int j = t / f.K_ED, k = f.ed_knn_idx[t];
    int lo = min(j, k), hi = max(j, k);
    wmax = max(wmax, (7 * hi + 6) / NB - (7 * lo) / NB);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_down(wmax, o, 64));
  if ((threadIdx.x & 63) == 0 && wmax > 0) atomicMax(out, wmax);
}

// Load the lower triangle of diagonal tile c into LDS with damping u and unit padding rows.
// All 16 global loads of a thread are issued before the first LDS store (one memory
// round trip instead of sixteen).
__device__ __forceinline__ void load_diag_tile(const FrameDev& fd, int c, double u, double* S) {
  const double* src = fd.band + (size_t)c * (fd.wb + 1) * TILE;
  double v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) v[t] = src[threadIdx.x + 256 * t];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int e = threadIdx.x + 256 * t;
    const int i = e % NB, k = e / NB;
    double x = (i >= k) ? v[t] : 0.0;
    if (i == k) {
      const int gi = c * NB + i;
      x = (gi < fd.P) ? x + u : 1.0;
    }
    S[i + k * LD] = x;
  }
}

// grid = (wb_cap + 1, n_frames), 256 threads
__global__ void __launch_bounds__(256) k_panel(const FrameDev* __restrict__ frames, int c,
                                                double u_override) {
  extern __shared__ double lds[];
  double* S = lds;
  double* M = lds + TILE;
  double* dinv = lds + 2 * TILE;
  double* wt = dinv + 4 * 256;   // 3 scratch blocks (inverse_assemble64 runs on at most 3 waves)
  double* vec = wt + 3 * 256;
  int* s_ok = reinterpret_cast<int*>(vec + NB);
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || fd.st->stopped || c >= fd.nt) return;
  const int d = blockIdx.x;
  if (d > fd.wb || c + d >= fd.nt) return;
  const double u = (u_override >= 0.0) ? u_override : fd.st->u;
  const bool stamp = (c == 8 && blockIdx.y == 0 && d == 1);
  SLM_STAMP(fd, stamp, 0);

  // issue this block's own global loads first so they overlap the factorisation
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
  double* At = fd.band + ((size_t)c * (fd.wb + 1) + d) * TILE;
  double4_t a[4];
  if (d > 0) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) a[kb][r] = At[(16 * w + lr) + (size_t)(16 * kb + lk + 4 * r) * NB];
  } else if (threadIdx.x < NB) {
    vec[threadIdx.x] = fd.rhs[(size_t)c * NB + threadIdx.x];
  }

  load_diag_tile(fd, c, u, S);
  __syncthreads();
  SLM_STAMP(fd, stamp, 1);
  const bool ok = potrf64(S, dinv, wt, s_ok, fd, stamp);
  SLM_STAMP(fd, stamp, 14);

  if (d == 0) {
    if (!ok && threadIdx.x == 0) fd.st->chol_fail = 1;
    // full inverse of the diagonal block: used by the substitutions (one parallel matvec each)
    inverse_assemble64(S, M, dinv, wt);
    double* linv = fd.linv + (size_t)c * TILE;
    for (int e = threadIdx.x; e < TILE; e += blockDim.x) linv[e] = M[e];
    // forward substitution of this block row: y_c = L^-1 b_c
    if (threadIdx.x < NB) {
      const int i = threadIdx.x;
      double acc = 0.0;
      for (int k = 0; k <= i; ++k) acc += M[i + k * LD] * vec[k];
      fd.rhs[(size_t)c * NB + i] = acc;
    }
  } else {
    // L(c+d, c) = A(c+d, c) L^-T
    trsm_rows16(S, dinv, a);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) At[(16 * w + lr) + (size_t)(16 * kb + lk + 4 * r) * NB] = a[kb][r];
  }
  SLM_STAMP(fd, stamp, 15);
}

// grid = (wb_cap*(wb_cap+1)/2 + wb_cap, n_frames)
__global__ void __launch_bounds__(256) k_trail(const FrameDev* __restrict__ frames, int c,
                                                int wb_cap) {
  __shared__ double Bl[TILE];
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || fd.st->stopped || c >= fd.nt) return;
  const int ntri = wb_cap * (wb_cap + 1) / 2;
  int t = blockIdx.x;
  if (t < ntri) {
    // (a,b), 1 <= b <= a <= wb_cap, row-major over the lower triangle
    int a = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= t) ++a;
    while (a * (a + 1) / 2 > t) --a;
    const int b = t - a * (a + 1) / 2;
    const int da = a + 1, db = b + 1;
    if (da > fd.wb || c + da >= fd.nt) return;
    const size_t col = (size_t)c * (fd.wb + 1);
    const double* Lr = fd.band + (col + da) * TILE;
    const double* Ls = fd.band + (col + db) * TILE;
    double* Ct = fd.band + ((size_t)(c + db) * (fd.wb + 1) + (da - db)) * TILE;
    // all global loads up front: B tile -> LDS (16 doubles per thread), A fragments and C -> registers
    double breg[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) breg[e] = Ls[threadIdx.x + 256 * e];
    double areg[16];
    load_a_frags(Lr, areg);
    double4_t acc[4];
    load_c_frags(Ct, acc);
#pragma unroll
    for (int e = 0; e < 16; ++e) Bl[threadIdx.x + 256 * e] = breg[e];
    __syncthreads();
    tile_ABt_regs<true>(areg, Bl, acc);
    store_c_frags(Ct, acc);
  } else {
    // rhs: b_s -= L(s,c) y_c, s = c + db
    const int db = t - ntri + 1;
    if (db > fd.wb || c + db >= fd.nt) return;
    __shared__ double y[NB];
    __shared__ double part[4][NB];
    const double* Ls = fd.band + ((size_t)c * (fd.wb + 1) + db) * TILE;
    if (threadIdx.x < NB) y[threadIdx.x] = fd.rhs[(size_t)c * NB + threadIdx.x];
    __syncthreads();
    const int i = threadIdx.x & 63, q = threadIdx.x >> 6;
    double acc = 0.0;
#pragma unroll
    for (int k = 16 * q; k < 16 * q + 16; ++k) acc += Ls[i + k * NB] * y[k];
    part[q][i] = acc;
    __syncthreads();
    if (threadIdx.x < NB)
      fd.rhs[(size_t)(c + db) * NB + i] -= part[0][i] + part[1][i] + part[2][i] + part[3][i];
  }
}

// grid = (wb_cap + 1, n_frames)
__global__ void __launch_bounds__(256) k_backsub(const FrameDev* __restrict__ frames, int c_from_end) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || fd.st->stopped) return;
  const int c = fd.nt - 1 - c_from_end;
  if (c < 0) return;
  const int d = blockIdx.x;
  if (d > fd.wb || c - d < 0) return;
  __shared__ double y[NB];
  __shared__ double x[NB];
  __shared__ double part[4][NB];
  const double* linv = fd.linv + (size_t)c * TILE;
  if (threadIdx.x < NB) y[threadIdx.x] = fd.rhs[(size_t)c * NB + threadIdx.x];
  __syncthreads();
  {
    // x_c = L^-T y_c : x[k] = sum_{i>=k} Linv[i][k] y[i]
    const int k = threadIdx.x & 63, q = threadIdx.x >> 6;
    double acc = 0.0;
    for (int i = 16 * q; i < 16 * q + 16; ++i) acc += linv[i + k * NB] * y[i];
    part[q][k] = acc;
    __syncthreads();
    if (threadIdx.x < NB) x[k] = part[0][k] + part[1][k] + part[2][k] + part[3][k];
    __syncthreads();
  }
  if (d == 0) {
    if (threadIdx.x < NB) fd.delta[(size_t)c * NB + threadIdx.x] = x[threadIdx.x];
  } else {
    // y_(c-d) -= L(c, c-d)^T x_c ; tile (c, c-d) is at column c-d, offset d
    const double* Lt = fd.band + ((size_t)(c - d) * (fd.wb + 1) + d) * TILE;
    const int n = threadIdx.x >> 2, q = threadIdx.x & 3;
    double acc = 0.0;
    for (int mrow = 16 * q; mrow < 16 * q + 16; ++mrow) acc += Lt[mrow + n * NB] * x[mrow];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (q == 0) fd.rhs[(size_t)(c - d) * NB + n] -= acc;
  }
}

// Expand the assembled lower band to a dense symmetric (P,P) row-major matrix (parity tests).
__global__ void __launch_bounds__(256) k_band_to_dense(const FrameDev* __restrict__ frames, int slot,
                                                        double* __restrict__ out) {
  const FrameDev& fd = frames[slot];
  const size_t P = fd.P;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < P * P;
       e += (size_t)gridDim.x * blockDim.x) {
    int i = (int)(e / P), j = (int)(e % P);
    int hi = max(i, j), lo = min(i, j);
    double v = 0.0;
    if (hi / NB - lo / NB <= fd.wb) v = *band_entry(fd, hi, lo);
    out[e] = v;
  }
}

// Pack a dense symmetric (P,P) row-major matrix into the (full-width) band (slm_solve_dense).
__global__ void __launch_bounds__(256) k_dense_to_band(const FrameDev* __restrict__ frames,
                                                        const double* __restrict__ A,
                                                        const double* __restrict__ b) {
  const FrameDev& fd = frames[0];
  const size_t P = fd.P, Ppad = (size_t)fd.nt * NB;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < Ppad * Ppad;
       e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / Ppad), j = (int)(e % Ppad);
    if (i < j) continue;
    *band_entry(fd, i, j) = (i < (int)P && j < (int)P) ? A[(size_t)i * P + j] : 0.0;
  }
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < Ppad;
       e += (size_t)gridDim.x * blockDim.x)
    fd.rhs[e] = e < P ? b[e] : 0.0;
}

// ---- host launchers --------------------------------------------------------------
void launch_dense_to_band(const FrameDev* frames_dev, const double* A, const double* b,
                          hipStream_t st) {
  hipLaunchKernelGGL(k_dense_to_band, dim3(1024), dim3(256), 0, st, frames_dev, A, b);
}

void launch_bandwidth(const slm_frame& f, int* out_dev, hipStream_t st) {
  (void)hipMemsetAsync(out_dev, 0, sizeof(int), st);
  hipLaunchKernelGGL(k_bandwidth, dim3(256), dim3(256), 0, st, f, out_dev);
}

// Factor + forward substitution + back substitution for all frames; nt_max / wb_cap are
// the maxima over the batch (blocks beyond a frame's own nt / wb exit immediately).
void launch_band_solve(const FrameDev* frames_dev, int n_frames, int nt_max, int wb_cap,
                       double u_override, hipStream_t st) {
  const size_t lds = PANEL_LDS_DOUBLES * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int ntrail = wb_cap * (wb_cap + 1) / 2 + wb_cap;
  for (int c = 0; c < nt_max; ++c) {
    hipLaunchKernelGGL(k_panel, dim3(wb_cap + 1, n_frames), dim3(256), lds, st, frames_dev, c,
                       u_override);
    if (ntrail > 0)
      hipLaunchKernelGGL(k_trail, dim3(ntrail, n_frames), dim3(256), 0, st, frames_dev, c, wb_cap);
  }
  for (int e = 0; e < nt_max; ++e)
    hipLaunchKernelGGL(k_backsub, dim3(wb_cap + 1, n_frames), dim3(256), 0, st, frames_dev, e);
}

void launch_band_to_dense(const FrameDev* frames_dev, int slot, double* out, hipStream_t st) {
  hipLaunchKernelGGL(k_band_to_dense, dim3(1024), dim3(256), 0, st, frames_dev, slot, out);
}
