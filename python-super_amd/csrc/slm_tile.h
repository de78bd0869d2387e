// slm_tile.h -- 64x64 float64 tile kernels shared by the band solver (slm_band.hip) and the
// multifrontal solver (slm_front.hip): blocked Cholesky of a tile in LDS on the f64 MFMA
// (v_mfma_f64_16x16x4_f64), triangular solve of a tile against it, tile products.
#pragma once
#include "slm_common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

// A time stamp of the (optional) task trace.  The store goes through a GLOBAL-address-space pointer on purpose: a store
// through a generic pointer is a FLAT instruction, and one pending flat access makes the compiler's wait-count pass
// drop to "wait for every load in flight" (s_waitcnt vmcnt(0)) at each later use of a loaded register -- in a loop, at
// every iteration, because the loop header inherits the state of the code before it.  A single trace stamp at the top
// of a task function serialised every software-pipelined tile stream of that function on the memory latency this way.
__device__ __forceinline__ void trace_put(long long* trc, int k, long long v) {
  typedef __attribute__((address_space(1))) long long gll;
  *((gll*)(trc + k)) = v;
}

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

// ~1 ulp reciprocal square root: v_rsq_f64 is good to ~1e-9, one Newton step squares that.
__device__ __forceinline__ double rsq1(double p) {
  const double r = __builtin_amdgcn_rsq(p);
  return r * fma(-(0.5 * p) * r, r, 1.5);
}

// 16x16 diagonal block on ONE wave, register resident: blocked right-looking Cholesky with 4x4
// pivot blocks, so that all four k-slots of v_mfma_f64_16x16x4_f64 carry a rank-1 term (on gfx950
// an f64 MFMA issues in 70 cycles and does not overlap f64 VALU work: one MFMA per pivot, as in a
// rank-1 formulation, costs 370 cycles per pivot; this costs 260, tools/micro/diag16_mb.hip).
// The block S (full symmetric) and the running inverse M (starts as I) live in MFMA accumulator
// layout: reg r of lane l <-> [row (l>>4)+4r][col l&15].  Block step b (rows 4b..4b+3 = register
// b of the four lane quarters):
//   * rows 4b..4b+3 of S and M go through LDS once (xch, 128 doubles owned by this wave): every
//     lane reads the 4x4 pivot block P (broadcast) and the four entries S[i][4b..4b+3] of its
//     row i = l&15 (resp. M[4b..4b+3][n] of its column);
//   * every lane factors P = Lp Lp^T (10 entries, identical in all lanes) and forward-substitutes
//     W[i][:] = S[i][blk] Lp^-T, Bm[:][n] = Lp^-1 M[blk][n]; lane (q,i) keeps W[i][q], Bm[q][i];
//   * L[i][4b+q] = W and L^-1[4b+q][n] = Bm are final and go to Sd / Dinv;
//   * S -= W W^T and M -= W Bm for the rows below the block: lane 16q+i holds A[i][k=q] = -W[i][q]
//     and B[k=q][n=i] = W[i][q] (resp. Bm[q][i]) -- ONE MFMA each, no further data movement.
// Writes L (lower, zeros above) to Sd (LDS, ld LD) and L^-1 to Dinv (16x16, ld 16).
// (LDD: leading dimension of the block at Sd -- LD inside a tile, 16 for a block staged on its own; a run-time value so
//  that a caller with both kinds of block keeps ONE copy of this code in its loop)
__device__ __forceinline__ bool diag16(double* Sd, double* Dinv, double* xch, const int LDD = LD) {
  const int l = threadIdx.x & 63, lc = l & 15, lq = l >> 4;
  double4_t S, M;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lq + 4 * r;
    S[r] = (row >= lc) ? Sd[row + lc * LDD] : Sd[lc + row * LDD];   // symmetric from the lower part
    M[r] = (row == lc) ? 1.0 : 0.0;
  }
  bool ok = true;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    xch[l] = S[b];
    xch[64 + l] = M[b];
    wave_sync();
    const double* ps = xch + 4 * b;     // P[q][q'] = xch[16q + 4b + q']
    const double p00 = ps[0];
    const double p10 = ps[16], p11 = ps[17];
    const double p20 = ps[32], p21 = ps[33], p22 = ps[34];
    const double p30 = ps[48], p31 = ps[49], p32 = ps[50], p33 = ps[51];
    const double s0 = xch[lc], s1 = xch[16 + lc], s2 = xch[32 + lc], s3 = xch[48 + lc];
    const double m0 = xch[64 + lc], m1 = xch[80 + lc], m2 = xch[96 + lc], m3 = xch[112 + lc];
    // Cholesky of the pivot block
    const double r0 = rsq1(p00);
    const double l10 = p10 * r0, l20 = p20 * r0, l30 = p30 * r0;
    const double d1 = fma(-l10, l10, p11);
    const double r1 = rsq1(d1);
    const double l21 = fma(-l20, l10, p21) * r1, l31 = fma(-l30, l10, p31) * r1;
    const double d2 = fma(-l21, l21, fma(-l20, l20, p22));
    const double r2 = rsq1(d2);
    const double l32 = fma(-l31, l21, fma(-l30, l20, p32)) * r2;
    const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, p33)));
    const double r3 = rsq1(d3);
    ok = ok && (p00 > 0.0) && (d1 > 0.0) && (d2 > 0.0) && (d3 > 0.0);
    // forward substitutions (every lane solves the four unknowns of its row / column)
    const double w0 = s0 * r0, w1 = fma(-l10, w0, s1) * r1, w2 = fma(-l21, w1, fma(-l20, w0, s2)) * r2,
                 w3 = fma(-l32, w2, fma(-l31, w1, fma(-l30, w0, s3))) * r3;
    const double b0 = m0 * r0, b1 = fma(-l10, b0, m1) * r1, b2 = fma(-l21, b1, fma(-l20, b0, m2)) * r2,
                 b3 = fma(-l32, b2, fma(-l31, b1, fma(-l30, b0, m3))) * r3;
    const double W = lq == 0 ? w0 : (lq == 1 ? w1 : (lq == 2 ? w2 : w3));
    const double Bm = lq == 0 ? b0 : (lq == 1 ? b1 : (lq == 2 ? b2 : b3));
    const int col = 4 * b + lq;
    Sd[lc + col * LDD] = (lc >= col) ? W : 0.0;       // L[i][4b+q]; zero the strict upper part
    Dinv[col + 16 * lc] = (lc <= col) ? Bm : 0.0;     // L^-1[4b+q][n]
    if (b < 3) {
      const double a = (lc > 4 * b + 3) ? -W : 0.0;   // rows of finished blocks stay as they are
      S = __builtin_amdgcn_mfma_f64_16x16x4f64(a, W, S, 0, 0, 0);
      M = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bm, M, 0, 0, 0);
      wave_sync();                                    // xch is rewritten by the next block
    }
  }
  wave_sync();
  return ok;
}

// ---------------------------------------------------------------------------------
// Factor a 64x64 SPD tile held in LDS (column-major, ld LD; only the lower triangle is
// read) as L L^T and form L^-1, blocked by 16: the four diagonal blocks run on wave 0
// (diag16), the panel / trailing / inverse-assembly products on the f64 MFMA across the
// four waves.  On exit S = L (lower, zero above), dinv[kb] = inverse of diagonal block kb
// (4 x 256 doubles); xch = 128 doubles of LDS scratch (may alias the scratch of
// inverse_assemble64, which runs afterwards).  256 threads.  Returns false (in every thread) when
// a pivot is <= 0 / NaN.
__device__ __forceinline__ bool potrf64(double* S, double* dinv, double* xch, int* s_ok, const FrameDev& fd,
                                        bool stamp) {
  const int w = threadIdx.x >> 6;
  if (threadIdx.x == 0) *s_ok = 1;
  __syncthreads();
  for (int kb = 0; kb < 4; ++kb) {
    if (w == 0) {
      const bool ok = diag16(S + kb * 16 * (LD + 1), dinv + kb * 256, xch);
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

// ---------------------------------------------------------------------------------
// Pipelined form of potrf64 + inverse_assemble64 for the latency-critical callers (slm_dag.hip): wave 0 runs
// the pivot chain -- diag16(kb), its own panel block (kb+1,kb), the update of the NEXT diagonal block, diag16(kb+1)
// -- with one workgroup barrier per 16 pivots and nothing else on its path; waves 1..3 trail it: the other panel
// blocks, the other trailing updates (one round behind, while wave 0 is inside the next diag16) and the blocks of
// L^-1 as soon as their operands exist.  After the last diagonal block only one 16x16 product per wave remains.
// The waves hand the panel blocks of a round to each other through LDS flags (pf, 16 ints), not barriers.
//   S   in: SPD tile, lower triangle (+ damping) ; out: L (lower, zeros above inside the diagonal blocks)
//   M   out: L^-1 (full 64x64, zeros above)
//   dinv 4 x 256 (inverses of the diagonal blocks), wt 3 x 256 (one scratch block per trailing wave),
//   xch 128 doubles (wave 0's exchange buffer for diag16)
// (the flags are addressed as LDS, address space 3: through a generic pointer the polls are FLAT loads -- a longer round
// trip, and a pending flat access makes the compiler wait for every global load in flight at the next use of one)
typedef __attribute__((address_space(3))) volatile int lds_vint;
__device__ __forceinline__ void lds_signal(int* f) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & 63) == 0) *((lds_vint*)f) = 1;
}
__device__ __forceinline__ void lds_wait_all(int* f, int n) {
  for (int i = 0; i < n; ++i)
    while (((lds_vint*)f)[i] == 0) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// W = A B (16x16 blocks); A = L block of S (row stride 1, column stride LD); B = a diagonal-block inverse
// (b_dinv: 16x16 column-major ld 16) or a block of M (column-major ld LD); accumulates into acc
__device__ __forceinline__ double4_t blk_LM(double4_t acc, const double* Lblk, const double* B, bool b_dinv) {
  return blk_mma<false>(acc, Lblk, 1, LD, B, 1, b_dinv ? 16 : LD);
}
// M(ib,jb) = -Dinv_ib W, with W taken from the wave's scratch block (16x16 column-major ld 16)
__device__ __forceinline__ void blk_neg_dinv_store(double* Mblk, const double* dinv_ib, double* Wscr, double4_t w) {
  blk_store(Wscr, 1, 16, w);
  wave_sync();
  double4_t m2 = {0.0, 0.0, 0.0, 0.0};
  m2 = blk_mma<true>(m2, dinv_ib, 1, 16, Wscr, 1, 16);
  blk_store(Mblk, 1, LD, m2);
  wave_sync();
}
__device__ __forceinline__ void blk_trail(double* S, int ib, int jb, int kb) {
  double* Cb = S + ib * 16 + jb * 16 * LD;
  double4_t acc = blk_load(Cb, 1, LD);
  acc = blk_mma<true>(acc, S + ib * 16 + kb * 16 * LD, 1, LD, S + jb * 16 + kb * 16 * LD, LD, 1);
  blk_store(Cb, 1, LD, acc);
}

// Streaming (optional, g_mail != nullptr): as soon as 16 more pivots are done, a trailing wave publishes what a
// consumer needs to run ITS row solve against this tile block column by block column (slm_dag.hip): the inverse of
// diagonal block kb and the finished row block of L (blocks (kb, t < kb)), with agent-scope (sc1) stores.
// g_early: a flag for stores that ALL waves of the caller issued just before the call (slm_dag.hip: the tile left of
// the diagonal one): every wave drains them before the first barrier -- wave 0 after its first 16 pivots, when they
// have long landed -- and one lane publishes the flag after it.
// The streamed hand-off of a tile factorisation (slm_dag.hip): per pivot tile column a MAILBOX of 10 blocks of 16 x 16
// doubles -- blocks 0..3: the inverses of the diagonal blocks; block 4 + kb (kb - 1) / 2 + t: L block (kb, t), t < kb --
// element (i, k) of a block at i + 16 k.  The mailbox is filled with SLM_MAIL_EMPTY before the launch (k_dag_reset) and
// the consumers poll the DATA: a value that is not the sentinel has arrived.  No drain, no flag: one memory round trip
// per hand-off instead of store - drain - flag, poll, payload load.  The sentinel is a NaN with a payload that no
// arithmetic produces (operations return the canonical quiet NaN).
#define SLM_MAIL_DOUBLES 2560
#define SLM_MAIL_EMPTY 0x7FF8DEADBEEF0001ll
__device__ __forceinline__ void publish_blocks16(const double* S, const double* dinv, int kb, double* g_mail) {
  typedef __attribute__((address_space(1))) double gdbl;
  const int l = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = l + 64 * j;
    __hip_atomic_store((gdbl*)(g_mail + kb * 256 + e), dinv[kb * 256 + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int t = 0; t < kb; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = l + 64 * j, i = e & 15, k = e >> 4;
      __hip_atomic_store((gdbl*)(g_mail + (4 + kb * (kb - 1) / 2 + t) * 256 + e), S[(16 * kb + i) + (16 * t + k) * LD],
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// nblk (1..4): 16-pivot blocks of the tile that hold real pivots.  The blocks beyond are the identity padding of a front's
// last pivot tile column (unit diagonal, zero rows): their "factorisation" is known -- the pivot chain skips diag16 for
// them (1.7 us each on the critical path) and only writes the identity inverse; everything else runs unchanged.
// LOOK-AHEAD entry (la_blk != nullptr; slm_dag.hip, POTRF(s > 0)): the caller's wave 0 has staged the first 16 x 16 diagonal
// block on its own (la_blk, ld 16, lower triangle + damping) as soon as ITS rows of the tile were updated, and enters here
// while the other waves are still writing rows 16..63 of S: there is no barrier in front of the first 16 pivots -- the one
// behind them (which every wave reaches with its part of S in place) is the first.  Row block 0 of S is never read.
// la_part (4 x 64 doubles, may alias wt): row partials the caller's waves stored before entering; their sums are
// returned in *la_sum for the threads of wave 0 -- read right behind the first barrier, before wt is used as scratch.
__device__ __forceinline__ bool factor_inverse64p(double* S, double* M, double* dinv, double* wt, double* xch,
                                                  int* s_ok, int* pf, double* g_mail = nullptr, int* g_early = nullptr,
                                                  long long* trc = nullptr, int nblk = 4, double* la_blk = nullptr,
                                                  const double* la_part = nullptr, double* la_sum = nullptr) {
#define FTRC(k) do { if (trc && threadIdx.x == 0) trace_put(trc, (k), wall_clock64()); } while (0)
  const int w = threadIdx.x >> 6;
  const double4_t z4 = {0.0, 0.0, 0.0, 0.0};
  if (threadIdx.x < 16) pf[threadIdx.x] = 0;
  if (threadIdx.x == 0) *s_ok = 1;
  if (!la_blk) __syncthreads();
  double* Wscr = wt + (w > 0 ? w - 1 : 0) * 256;
  int* vpf = pf;
#pragma unroll 1   // one copy of diag16: the unrolled form needs ~250 VGPRs and spills on the critical path
  for (int kb = 0; kb < 4; ++kb) {
    FTRC(3 * kb);
    if (w == 0) {
      if (kb < nblk) {
        const bool staged = kb == 0 && la_blk;
        const bool ok = diag16(staged ? la_blk : S + kb * 16 * (LD + 1), dinv + kb * 256, xch, staged ? 16 : LD);
        if (!ok && (threadIdx.x & 63) == 0) *s_ok = 0;
      } else {   // identity block: L = I is in place, its inverse is I
        const int l = threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = l + 64 * j;
          dinv[kb * 256 + e] = ((e & 15) == (e >> 4)) ? 1.0 : 0.0;
        }
        wave_sync();
      }
      FTRC(3 * kb + 1);
    } else if (kb == 0) {
      // the six blocks above the diagonal stay zero (everything else is written below)
      for (int e = threadIdx.x - 64; e < 6 * 256; e += 192) {
        const int b = e >> 8, ib = b < 3 ? 0 : (b < 5 ? 1 : 2), jb = b < 3 ? b + 1 : (b < 5 ? b - 1 : 3);
        M[(16 * ib + (e & 15)) + (16 * jb + ((e >> 4) & 15)) * LD] = 0.0;
      }
    } else if (kb == 1) {
      lds_wait_all(vpf + 0, 3);                    // panel blocks (1,0) (2,0) (3,0)
      if (w == 1) { blk_trail(S, 2, 1, 0); blk_trail(S, 3, 2, 0); }
      else if (w == 2) { blk_trail(S, 2, 2, 0); blk_trail(S, 3, 3, 0); }
      else blk_trail(S, 3, 1, 0);
    } else if (kb == 2) {
      lds_wait_all(vpf + 4, 2);                    // panel blocks (2,1) (3,1)
      if (w == 1) blk_trail(S, 3, 2, 1);
      else if (w == 2) blk_trail(S, 3, 3, 1);
      else {                                       // M10 = -Dinv_1 (L10 Dinv_0)
        const double4_t t = blk_LM(z4, S + 16, dinv, true);
        blk_neg_dinv_store(M + 16, dinv + 256, Wscr, t);
      }
    } else {
      lds_wait_all(vpf + 8, 1);                    // panel block (3,2)
      if (w == 3) {                                // M21 = -Dinv_2 (L21 Dinv_1);  T1 = L31 Dinv_1 + L32 M21
        double4_t t = blk_LM(z4, S + 32 + 16 * LD, dinv + 256, true);
        blk_neg_dinv_store(M + 32 + 16 * LD, dinv + 512, Wscr, t);
        t = blk_LM(z4, S + 48 + 16 * LD, dinv + 256, true);
        t = blk_LM(t, S + 48 + 32 * LD, M + 32 + 16 * LD, false);
        blk_store(Wscr, 1, 16, t);
      } else if (w == 1) {                         // M20 = -Dinv_2 (L20 Dinv_0 + L21 M10);  T0 = L30 Dinv_0 + L31 M10 + L32 M20
        double4_t t = blk_LM(z4, S + 32, dinv, true);
        t = blk_LM(t, S + 32 + 16 * LD, M + 16, false);
        blk_neg_dinv_store(M + 32, dinv + 512, Wscr, t);
        t = blk_LM(z4, S + 48, dinv, true);
        t = blk_LM(t, S + 48 + 16 * LD, M + 16, false);
        t = blk_LM(t, S + 48 + 32 * LD, M + 32, false);
        blk_store(Wscr, 1, 16, t);
      } else {                                     // T2 = L32 Dinv_2
        const double4_t t = blk_LM(z4, S + 48 + 32 * LD, dinv + 512, true);
        blk_store(Wscr, 1, 16, t);
      }
      wave_sync();
    }
    if (kb == 0 && g_early) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the caller's earlier stores (see g_early)
    __syncthreads();   // Dinv_kb and L(kb,kb) visible; the trailing work of the previous round is complete
    FTRC(3 * kb + 2);
    if (kb == 0 && la_part && threadIdx.x < NB)
      *la_sum = la_part[threadIdx.x] + la_part[NB + threadIdx.x] + la_part[2 * NB + threadIdx.x] + la_part[3 * NB + threadIdx.x];
    if (kb == 3) break;
    // the wave with the least trailing work in the coming round publishes
    if (g_mail && w == (kb == 0 ? 3 : 2)) publish_blocks16(S, dinv, kb, g_mail);
    if (kb == 0 && g_early && threadIdx.x == 128)
      __hip_atomic_store((__attribute__((address_space(1))) int*)g_early, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w < 3 - kb) {
      // panel: S[ib,kb] = S[ib,kb] Dinv_kb^T
      const int ib = kb + 1 + w;
      double* Xb = S + ib * 16 + kb * 16 * LD;
      double4_t acc = blk_mma<false>(z4, Xb, 1, LD, dinv + kb * 256, 16, 1);
      blk_store(Xb, 1, LD, acc);
      wave_sync();
      lds_signal(vpf + 4 * kb + w);
    }
    if (w == 0) {      // the next diagonal block, from this wave's own panel block
      blk_trail(S, kb + 1, kb + 1, kb);
      wave_sync();
    }
  }
  // row 3 of the inverse needs Dinv_3: one product per trailing wave; wave 0 copies the diagonal blocks
  if (w >= 1) {
    const int j = (w == 1) ? 0 : (w == 2 ? 2 : 1);
    double4_t m2 = blk_mma<true>(z4, dinv + 768, 1, 16, Wscr, 1, 16);
    blk_store(M + 48 + j * 16 * LD, 1, LD, m2);
  } else {
    if (g_mail) publish_blocks16(S, dinv, 3, g_mail);
    for (int e = threadIdx.x; e < 1024; e += 64) {
      const int b = e >> 8, i = e & 15, k = (e >> 4) & 15;
      M[(16 * b + i) + (16 * b + k) * LD] = dinv[b * 256 + i + 16 * k];
    }
  }
  __syncthreads();
  FTRC(12);
#undef FTRC
  return *s_ok != 0;
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

// LOWER_B: B is lower triangular by 16 x 16 blocks (the inverse of a tile factor: B[n][p] = 0 for p-block > n-block) -- the
// block products above the diagonal are not issued (40 MFMAs per wave instead of 64; the sums are the same, minus additions of 0)
template <bool NEGATE, bool LOWER_B = false>
__device__ __forceinline__ void tile_ABt_regs(const double areg[16], const double* Bl,
                                              double4_t acc[4]) {
  const int l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const double a = NEGATE ? -areg[ks] : areg[ks];
#pragma unroll
    for (int ni = LOWER_B ? (ks >> 2) : 0; ni < 4; ++ni) {
      const double b = Bl[(16 * ni + lr) + (4 * ks + lk) * LD];
      acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc[ni], 0, 0, 0);
    }
    // Keep the scheduler from hoisting all 64 LDS operand reads of the product in front of its first MFMA (128 VGPRs
    // of operands in flight: k_fL11 needed 366 registers -- one workgroup per CU -- because of it; an f64 MFMA issues
    // in ~70 cycles, the reads of the next four k-steps are covered many times over)
    if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
}

// The same product restricted to the first `kblk` blocks of 16 inner columns and the first `nblk` 16-column blocks of
// the result (uniform bounds, 1..4): what lies beyond is known to be zero -- the padding of a front's last pivot tile
// column.  The column-block count is a template parameter behind a switch and the inner blocks are whole branches: with
// per-instruction predicates the compiler hoists every operand read of the product (250 VGPRs).
template <bool NEGATE, int NBLK, bool LOWER_B>
__device__ __forceinline__ void tile_ABt_regs_nblk(const double areg[16], const double* Bl, double4_t acc[4], int kblk) {
  const int l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    if (kb < kblk) {
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const int ks = 4 * kb + k4;
        const double a = NEGATE ? -areg[ks] : areg[ks];
#pragma unroll
        for (int ni = LOWER_B ? kb : 0; ni < NBLK; ++ni) {
          const double b = Bl[(16 * ni + lr) + (4 * ks + lk) * LD];
          acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc[ni], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <bool NEGATE, bool LOWER_B = false>
__device__ __forceinline__ void tile_ABt_regs_trim(const double areg[16], const double* Bl, double4_t acc[4], int kblk,
                                                   int nblk) {
  switch (nblk) {
    case 1: tile_ABt_regs_nblk<NEGATE, 1, LOWER_B>(areg, Bl, acc, kblk); break;
    case 2: tile_ABt_regs_nblk<NEGATE, 2, LOWER_B>(areg, Bl, acc, kblk); break;
    case 3: tile_ABt_regs_nblk<NEGATE, 3, LOWER_B>(areg, Bl, acc, kblk); break;
    default: tile_ABt_regs_nblk<NEGATE, 4, LOWER_B>(areg, Bl, acc, kblk); break;
  }
}

__device__ __forceinline__ void store_c_frags(double* __restrict__ Cg, const double4_t acc[4]) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) Cg[(16 * w + lr) + (size_t)(16 * ni + lk + 4 * r) * NB] = acc[ni][r];
}

// S, M, 4 diagonal-block inverses, 3 scratch blocks, one vector: 80 448 B -> two workgroups per CU
#define PANEL_LDS_DOUBLES (2 * TILE + 7 * 256 + 3 * NB + 16)   // ... + vec, the factorisation's exchange buffer (2 NB), 32 ints: 81 536 B

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

