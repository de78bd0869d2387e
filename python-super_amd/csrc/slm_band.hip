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
#include "slm_common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

#define NB SLM_NB
#define TILE (NB * NB)

// ---------------------------------------------------------------------------------
// Tile half-bandwidth from the KNN tables: max over coupled node pairs (a >= b) of
// tile(7a+6) - tile(7b).  Surfel tuples couple all pairs among their K nodes
// (loss.py:277-288); ARAP couples (j, k) (loss.py:414-426).
__global__ void __launch_bounds__(256) k_bandwidth(slm_frame f, int* __restrict__ out) {
  int wmax = 0;
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < f.N; i += stride) {
    const int4 id = *reinterpret_cast<const int4*>(f.sf_knn_idx + 4 * i);
    int lo = min(min(id.x, id.y), min(id.z, id.w));
    int hi = max(max(id.x, id.y), max(id.z, id.w));
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

// ---------------------------------------------------------------------------------
// MFMA helpers (v_mfma_f64_16x16x4_f64).  Products are formed "transposed" so that lane&15
// runs along ROWS of the column-major result (contiguous addresses):
//   acc reg r of lane l  <->  C[m = l&15][n = (l>>4) + 4r]
//   first operand  (lane l, k-step ks) = Y[p][n = l&15],  p = (l>>4) + 4ks
//   second operand (lane l, k-step ks) = X[m = l&15][p]
// computes C += X Y  (X is m x p, Y is p x n).
#define LD NB   // leading dimension of 64x64 column-major LDS tiles

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one 16x16 block product over p = 0..15
template <bool NEG>
__device__ __forceinline__ double4_t blk_mma(double4_t acc, const double* X, int xs_m, int xs_p,
                                             const double* Y, int ys_p, int ys_n) {
  const int l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int p = 4 * ks + lk;
    const double y = Y[p * ys_p + lr * ys_n];
    double x = X[lr * xs_m + p * xs_p];
    if (NEG) x = -x;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, acc, 0, 0, 0);
  }
  return acc;
}

__device__ __forceinline__ double4_t blk_load(const double* C, int cs_m, int cs_n) {
  const int l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
  double4_t v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = C[lr * cs_m + (lk + 4 * r) * cs_n];
  return v;
}

__device__ __forceinline__ void blk_store(double* C, int cs_m, int cs_n, double4_t v) {
  const int l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) C[lr * cs_m + (lk + 4 * r) * cs_n] = v[r];
}

// ---------------------------------------------------------------------------------
// ~1 ulp reciprocal / reciprocal square root from the hardware estimates + Newton steps
// (an IEEE f64 division costs ~30 dependent instructions and sits on the pivot chain).
__device__ __forceinline__ double rcp_nr(double p) {
  double r = __builtin_amdgcn_rcp(p);
  double e = fma(-p, r, 1.0);
  r = fma(r, e, r);
  e = fma(-p, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ double rsq_nr(double p) {
  double r = __builtin_amdgcn_rsq(p);
  double h = 0.5 * p;
  r = r * fma(-h * r, r, 1.5);
  return r * fma(-h * r, r, 1.5);
}
__device__ __forceinline__ double readlane_d(double x, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}

// 16x16 diagonal block on ONE wave, register resident, rank-1 updates on the f64 MFMA.
// The block S (full symmetric) and the running inverse M (starts as I) live in MFMA
// accumulator layout: reg r of lane l <-> [row (l>>4)+4r][col l&15].  Pivot step j
// (q = j&3, r = j>>2): row j of S already sits in register r of the 16 lanes of quarter
// q, indexed by column -- exactly the k = q slot of both MFMA operands -- so
//   S -= (v/p) v^T  and  M -= (v/p) M[j,:]
// are one MFMA each with no cross-lane traffic; only the pivot p travels (v_readlane).
// Row j is excluded from the update (its A-operand entry is zeroed), so on exit S holds
// U = diag(p) L~^T (upper) and M holds L~^-1 (unit lower):
//   L = U^T diag(p)^-1/2,  L^-1 = diag(p)^-1/2 L~^-1.
// Writes L (lower, zeros above) to Sd (LDS, ld LD) and L^-1 to Dinv (16x16, ld 16).
__device__ __forceinline__ bool diag16(double* Sd, double* Dinv) {
  const int l = threadIdx.x & 63, lc = l & 15, lq = l >> 4;
  double4_t S, M;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lq + 4 * r;
    S[r] = (row >= lc) ? Sd[row + lc * LD] : Sd[lc + row * LD];   // symmetric from the lower part
    M[r] = (row == lc) ? 1.0 : 0.0;
  }
  bool ok = true;
  double pv = 1.0;   // lane with (l & 15) == j keeps pivot j
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int q = j & 3, r = j >> 2;
    const double v = S[r];
    const double p = readlane_d(v, 16 * q + j);
    ok = ok && (p > 0.0);
    const double rinv = rcp_nr(p);
    pv = (lc == j) ? p : pv;
    const bool mine = (lq == q);
    const double a = (mine && lc != j) ? -v * rinv : 0.0;
    const double bs = mine ? v : 0.0;
    const double bm = mine ? M[r] : 0.0;
    S = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bs, S, 0, 0, 0);
    M = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bm, M, 0, 0, 0);
  }
  const double rsv = rsq_nr(pv);   // lane l: 1/sqrt(pivot (l & 15))
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lq + 4 * r;
    const double rsr = __shfl(rsv, row, 64);
    // U[row][lc] -> L[lc][row]; zero the strict upper part of L explicitly
    if (lc >= row) Sd[lc + row * LD] = S[r] * rsr;
    else Sd[lc + row * LD] = 0.0;
    Dinv[row + 16 * lc] = (lc <= row) ? M[r] * rsr : 0.0;
  }
  wave_sync();
  return ok;
}

// ---------------------------------------------------------------------------------
// Factor a 64x64 SPD tile held in LDS (column-major, ld LD; only the lower triangle is
// read) as L L^T and form L^-1, blocked by 16: the four diagonal blocks run on wave 0
// (diag16), the panel / trailing / inverse-assembly products on the f64 MFMA across the
// four waves.  On exit S = L (lower, zero above), dinv[kb] = inverse of diagonal block kb
// (4 x 256 doubles).  256 threads.  Returns false (in every thread) when a pivot is <= 0 / NaN.
__device__ __forceinline__ bool potrf64(double* S, double* dinv, int* s_ok, const FrameDev& fd,
                                        bool stamp) {
  const int w = threadIdx.x >> 6;
  if (threadIdx.x == 0) *s_ok = 1;
  __syncthreads();
  for (int kb = 0; kb < 4; ++kb) {
    if (w == 0) {
      const bool ok = diag16(S + kb * 16 * (LD + 1), dinv + kb * 256);
      if (!ok && (threadIdx.x & 63) == 0) *s_ok = 0;
    }
    SLM_STAMP(fd, stamp, 2 + 3 * kb);
    __syncthreads();
    // panel: S[ib,kb] = S[ib,kb] Dinv^T   (Y[p][n] = Dinv[n][p])
    if (w < 3 - kb) {
      const int ib = kb + 1 + w;
      double* Xb = S + ib * 16 + kb * 16 * LD;
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
      acc = blk_mma<false>(acc, Xb, 1, LD, dinv + kb * 256, 16, 1);
      blk_store(Xb, 1, LD, acc);
    }
    __syncthreads();
    SLM_STAMP(fd, stamp, 3 + 3 * kb);
    // trailing: S[ib,jb] -= S[ib,kb] S[jb,kb]^T, kb < jb <= ib <= 3
    {
      int t = 0;
      for (int ib = kb + 1; ib < 4; ++ib)
        for (int jb = kb + 1; jb <= ib; ++jb, ++t) {
          if ((t & 3) != w) continue;
          double* Cb = S + ib * 16 + jb * 16 * LD;
          double4_t acc = blk_load(Cb, 1, LD);
          acc = blk_mma<true>(acc, S + ib * 16 + kb * 16 * LD, 1, LD, S + jb * 16 + kb * 16 * LD, LD, 1);
          blk_store(Cb, 1, LD, acc);
        }
    }
    __syncthreads();
    SLM_STAMP(fd, stamp, 4 + 3 * kb);
  }
  return *s_ok != 0;
}

// L^-1 (64x64, lower) from L (in S) and the four diagonal-block inverses, on the MFMA.
__device__ __forceinline__ void inverse_assemble64(const double* S, double* M, const double* dinv,
                                                   double* wt) {
  const int w = threadIdx.x >> 6;
  // inverse assembly: M[ib,ib] = Dinv_ib; M[ib,jb] = -Dinv_ib sum_{t=jb}^{ib-1} L[ib,t] M[t,jb]
  for (int e = threadIdx.x; e < TILE; e += blockDim.x) {
    const int i = e % NB, k = e / NB;
    M[i + k * LD] = ((i >> 4) == (k >> 4)) ? dinv[(i >> 4) * 256 + (i & 15) + 16 * (k & 15)] : 0.0;
  }
  __syncthreads();
  for (int dl = 1; dl < 4; ++dl) {
    if (w < 4 - dl) {
      const int jb = w, ib = jb + dl;
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
      for (int t = jb; t < ib; ++t)
        acc = blk_mma<false>(acc, S + ib * 16 + t * 16 * LD, 1, LD, M + t * 16 + jb * 16 * LD, 1, LD);
      double* W = wt + w * 256;
      blk_store(W, 1, 16, acc);
      wave_sync();
      double4_t m2 = {0.0, 0.0, 0.0, 0.0};
      m2 = blk_mma<true>(m2, dinv + ib * 256, 1, 16, W, 1, 16);
      blk_store(M + ib * 16 + jb * 16 * LD, 1, LD, m2);
    }
    __syncthreads();
  }
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

// C(64x64) = Cinit + sign * A B^T with B staged in LDS (Bl, ld LD) and the A fragments /
// C tile of this wave's 16 rows already in registers (loaded by the caller so that the
// global loads overlap whatever precedes).  Wave w owns rows [16w, 16w+16).
//   areg[ks] = A[16w + (l&15)][4ks + (l>>4)],  acc[ni][r] = C[16w + (l&15)][16ni + (l>>4) + 4r]
__device__ __forceinline__ void load_a_frags(const double* __restrict__ A, double areg[16]) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) areg[ks] = A[(16 * w + lr) + (size_t)(4 * ks + lk) * NB];
}

__device__ __forceinline__ void load_c_frags(const double* __restrict__ Cg, double4_t acc[4]) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[ni][r] = Cg[(16 * w + lr) + (size_t)(16 * ni + lk + 4 * r) * NB];
}

template <bool NEGATE>
__device__ __forceinline__ void tile_ABt_regs(const double areg[16], const double* Bl,
                                              double4_t acc[4]) {
  const int l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const double a = NEGATE ? -areg[ks] : areg[ks];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const double b = Bl[(16 * ni + lr) + (4 * ks + lk) * LD];
      acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc[ni], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void store_c_frags(double* __restrict__ Cg, const double4_t acc[4]) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) Cg[(16 * w + lr) + (size_t)(16 * ni + lk + 4 * r) * NB] = acc[ni][r];
}

#define PANEL_LDS_DOUBLES (2 * TILE + 8 * 256 + NB + 8)

// Rows [16w,16w+16) of X = A L^-T for one 64x64 tile, blockwise forward substitution on
// the MFMA with everything in registers: x[kb] / a[kb] are 16x16 blocks in accumulator
// layout (reg r of lane l <-> [row l&15][col (l>>4)+4r]), which is ALSO the layout of the
// second MFMA operand (X[m = l&15][p = (l>>4)+4ks] = reg ks), so products chain with no
// layout conversion:  X_kb = (A_kb - sum_{t<kb} X_t L[kb,t]^T) Dinv_kb^T.
__device__ __forceinline__ void trsm_rows16(const double* S, const double* dinv, double4_t a[4]) {
  const int l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    double4_t acc = a[kb];
#pragma unroll
    for (int t = 0; t < kb; ++t) {
      // acc -= X_t L[kb,t]^T : first operand Y[p][n] = L[16kb+n][16t+p]
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const double y = S[(16 * kb + lr) + (16 * t + 4 * ks + lk) * LD];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, -a[t][ks], acc, 0, 0, 0);
      }
    }
    double4_t x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const double y = dinv[kb * 256 + lr + 16 * (4 * ks + lk)];   // Y[p][n] = Dinv[n][p]
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(y, acc[ks], x, 0, 0, 0);
    }
    a[kb] = x;
  }
}

// grid = (wb_cap + 1, n_frames), 256 threads
__global__ void __launch_bounds__(256) k_panel(const FrameDev* __restrict__ frames, int c,
                                                double u_override) {
  extern __shared__ double lds[];
  double* S = lds;
  double* M = lds + TILE;
  double* dinv = lds + 2 * TILE;
  double* wt = dinv + 4 * 256;
  double* vec = wt + 4 * 256;
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
  const bool ok = potrf64(S, dinv, s_ok, fd, stamp);
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
