// Diagnostic micro-benchmark (not part of the library): cycles per pivot step of variants of the
// 16x16 diagonal-block factorisation used by potrf64 (slm_tile.h::diag16).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I python-super_amd/csrc tools/micro/diag16_mb.hip -o python-super_amd/build/mb.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "slm_tile.h"

template <int VAR>
__device__ __forceinline__ bool diag16_var(double* Sd, double* Dinv) {
  const int l = threadIdx.x & 63, lc = l & 15, lq = l >> 4;
  double4_t S, M;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lq + 4 * r;
    S[r] = (row >= lc) ? Sd[row + lc * LD] : Sd[lc + row * LD];
    M[r] = (row == lc) ? 1.0 : 0.0;
  }
  bool ok = true;
  double pv = 1.0;
  double p = readlane_d(S[0], 0);
  double rinv = rcp_nr(p);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int q = j & 3, r = j >> 2;
    const double v = S[r];
    ok = ok && (p > 0.0);
    pv = (lc == j) ? p : pv;
    const double a = (l != 16 * q + j && lq == q) ? -v * rinv : 0.0;
    const double bs = v;
    const double bm = M[r];
    double p_next = 1.0, rinv_next = 1.0;
    if (j + 1 < 16) {
      const int q1 = (j + 1) & 3, r1 = (j + 1) >> 2;
      const double x = readlane_d(v, 16 * q + (j + 1));
      const double d = readlane_d(S[r1], 16 * q1 + (j + 1));
      p_next = fma(-x * rinv, x, d);
      if (VAR & 2) rinv_next = __builtin_amdgcn_rcp(p_next);   // no Newton steps
      else rinv_next = rcp_nr(p_next);
    }
    S = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bs, S, 0, 0, 0);
    if (!(VAR & 1)) M = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bm, M, 0, 0, 0);
    p = p_next;
    rinv = rinv_next;
  }
  const double rsv = rsq_nr(pv);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lq + 4 * r;
    const double rsr = __shfl(rsv, row, 64);
    if (lc >= row) Sd[lc + row * LD] = S[r] * rsr;
    else Sd[lc + row * LD] = 0.0;
    Dinv[row + 16 * lc] = (lc <= row) ? M[r] * rsr : 0.0;
  }
  wave_sync();
  return ok;
}

// VAR 4: right-looking on the VALU: lane l < 16 owns ROW l of the 16x16 block in 16 registers;
// the pivot row travels by v_readlane (16 per step), no MFMA at all.
__device__ __forceinline__ bool diag16_valu(double* Sd) {
  const int l = threadIdx.x & 63;
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] = (l < 16) ? ((l >= c) ? Sd[l + c * LD] : Sd[c + l * LD]) : 0.0;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double p = readlane_d(a[j], j);
    ok = ok && p > 0.0;
    const double rs = rsq_nr(p);
    const double lij = a[j] * rs;            // L[l][j] for l >= j
    a[j] = lij;
#pragma unroll
    for (int c = j + 1; c < 16; ++c) {
      const double ljc = readlane_d(lij, c);   // L[c][j]
      a[c] = fma(-lij, ljc, a[c]);
    }
  }
  if (l < 16) {
#pragma unroll
    for (int c = 0; c < 16; ++c) Sd[l + c * LD] = (l >= c) ? a[c] : 0.0;
  }
  wave_sync();
  return ok;
}

// VAR 5: blocked right-looking Cholesky with 4x4 pivot blocks: all four k-slots of the MFMA carry a
// rank-1 term, so a 16x16 block needs 4 (+4 for the inverse) MFMAs instead of 16 (+16).
__device__ __forceinline__ double rsq1(double p) {
  const double r = __builtin_amdgcn_rsq(p);
  return r * fma(-(0.5 * p) * r, r, 1.5);
}
// xch: 128 doubles of LDS scratch owned by the calling wave (row b of S and of M, one entry per lane)
__device__ __forceinline__ bool diag16_b4(double* Sd, double* Dinv, double* xch) {
  const int l = threadIdx.x & 63, lc = l & 15, lq = l >> 4;
  double4_t S, M;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lq + 4 * r;
    S[r] = (row >= lc) ? Sd[row + lc * LD] : Sd[lc + row * LD];
    M[r] = (row == lc) ? 1.0 : 0.0;
  }
  bool ok = true;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    // rows 4b..4b+3 of S and M (register b of the four quarters) go through LDS once: the pivot
    // block is read back as broadcasts, the cross-quarter gathers as plain reads
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
    // Cholesky of the 4x4 pivot block (same in every lane) and the inverse of its factor
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
    // W[i = lc][:] = S[i][blk] Lp^-T and Bm[:][n = lc] = Lp^-1 M[blk][n] by forward substitution
    // (every lane solves all four unknowns of its row / column and keeps the one of its quarter)
    const double w0 = s0 * r0, w1 = fma(-l10, w0, s1) * r1, w2 = fma(-l21, w1, fma(-l20, w0, s2)) * r2,
                 w3 = fma(-l32, w2, fma(-l31, w1, fma(-l30, w0, s3))) * r3;
    const double b0 = m0 * r0, b1 = fma(-l10, b0, m1) * r1, b2 = fma(-l21, b1, fma(-l20, b0, m2)) * r2,
                 b3 = fma(-l32, b2, fma(-l31, b1, fma(-l30, b0, m3))) * r3;
    const double W = lq == 0 ? w0 : (lq == 1 ? w1 : (lq == 2 ? w2 : w3));
    const double Bm = lq == 0 ? b0 : (lq == 1 ? b1 : (lq == 2 ? b2 : b3));
    const int col = 4 * b + lq;
    Sd[lc + col * LD] = (lc >= col) ? W : 0.0;        // L[i][4b+q]
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

template <int VAR>
__global__ void __launch_bounds__(64) k_mb(const double* A, double* out, unsigned long long* cyc, int reps) {
  __shared__ double S[TILE];
  __shared__ double D[256];
  __shared__ double X[128];
  for (int e = threadIdx.x; e < 16 * 16; e += 64) {
    const int i = e & 15, k = e >> 4;
    S[i + k * LD] = A[e];
  }
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  bool ok = true;
  for (int it = 0; it < reps; ++it) {
    // restore the block (cheap relative to the factorisation; identical across variants)
    for (int e = threadIdx.x; e < 256; e += 64) S[(e & 15) + (e >> 4) * LD] = A[e];
    wave_sync();
    if (VAR == 5) ok = diag16_b4(S, D, X) && ok;
    else if (VAR == 4) ok = diag16_valu(S) && ok;
    else ok = diag16_var<VAR>(S, D) && ok;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; out[256] = ok ? 1.0 : 0.0; }
  for (int e = threadIdx.x; e < 256; e += 64) out[e] = S[(e & 15) + (e >> 4) * LD];
  for (int e = threadIdx.x; e < 256; e += 64) out[257 + e] = D[e];
}

int main() {
  std::vector<double> h(256);
  for (int i = 0; i < 16; ++i)
    for (int k = 0; k < 16; ++k) h[i + 16 * k] = (i == k) ? 20.0 + i : 1.0 / (1.0 + abs(i - k));
  double *A, *out; unsigned long long* cyc;
  (void)hipMalloc(&A, 256 * 8); (void)hipMalloc(&out, 513 * 8); (void)hipMalloc(&cyc, 8);
  (void)hipMemcpy(A, h.data(), 256 * 8, hipMemcpyHostToDevice);
  const int reps = 200;
  auto run = [&](auto kern, const char* name) {
    for (int w = 0; w < 3; ++w) {
      hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, A, out, cyc, reps);
      (void)hipDeviceSynchronize();
    }
    unsigned long long c; std::vector<double> o(513);
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(o.data(), out, 513 * 8, hipMemcpyDeviceToHost);
    // residuals against the input: |L L^T - A|_max and |Dinv L - I|_max
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 16; ++i)
      for (int k = 0; k < 16; ++k) {
        double a = 0, d = 0;
        for (int t = 0; t < 16; ++t) { a += o[i + 16 * t] * o[k + 16 * t]; d += o[257 + i + 16 * t] * o[t + 16 * k]; }
        e1 = fmax(e1, fabs(a - h[i + 16 * k])); e2 = fmax(e2, fabs(d - (i == k ? 1.0 : 0.0)));
      }
    printf("%-30s %8.1f cyc/block %6.1f /pivot ok=%g |LLt-A|=%.2e |Dinv L-I|=%.2e L[15][15]=%.12f\n", name,
           (double)c / reps, (double)c / reps / 16, o[256], e1, e2, o[15 + 16 * 15]);
  };
  run(k_mb<0>, "mfma S+M, rcp+2NR (current)");
  run(k_mb<1>, "mfma S only, rcp+2NR");
  run(k_mb<2>, "mfma S+M, rcp only");
  run(k_mb<3>, "mfma S only, rcp only");
  run(k_mb<4>, "VALU rows + readlane");
  run(k_mb<5>, "4x4 block pivots (rank-4 MFMA)");
  return 0;
}
