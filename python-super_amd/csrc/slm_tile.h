// slm_tile.h -- 64x64 float64 tile kernels shared by the band solver (slm_band.hip) and the
// multifrontal solver (slm_front.hip): blocked Cholesky of a tile in LDS on the f64 MFMA
// (v_mfma_f64_16x16x4_f64), triangular solve of a tile against it, tile products.
#pragma once
#include "slm_common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

#define NB SLM_NB
#define TILE (NB * NB)

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
  // The reciprocal of pivot j+1 is computed on the VALU while the rank-1 MFMAs of step j are
  // still in flight: p_{j+1} = S[j+1][j+1] - S[j][j+1]^2 / p_j, formed exactly like the
  // MFMA forms it (fma of the rounded A entry), from registers that step j-1 has completed.
  double p = readlane_d(S[0], 0);
  double rinv = rcp_nr(p);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int q = j & 3, r = j >> 2;
    const double v = S[r];
    ok = ok && (p > 0.0);
    pv = (lc == j) ? p : pv;
    // Only the A operand is masked: k-slots other than q (and row j itself) carry a zero A
    // entry, so whatever (finite) B value sits in those slots contributes nothing.
    const double a = (l != 16 * q + j && lq == q) ? -v * rinv : 0.0;
    const double bs = v;
    const double bm = M[r];
    double p_next = 1.0, rinv_next = 1.0;
    if (j + 1 < 16) {
      const int q1 = (j + 1) & 3, r1 = (j + 1) >> 2;
      const double x = readlane_d(v, 16 * q + (j + 1));           // S[j][j+1]
      const double d = readlane_d(S[r1], 16 * q1 + (j + 1));      // S[j+1][j+1] before this step
      p_next = fma(-x * rinv, x, d);
      rinv_next = rcp_nr(p_next);
    }
    S = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bs, S, 0, 0, 0);
    M = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bm, M, 0, 0, 0);
    p = p_next;
    rinv = rinv_next;
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

