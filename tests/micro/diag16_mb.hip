// Diagnostic micro-benchmark (not part of the library): cycles per pivot step of variants of the
// 16x16 diagonal-block factorisation used by potrf64 (slm_tile.h::diag16).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I python-super_amd/csrc tests/micro/diag16_mb.hip -o python-super_amd/build/mb.bin
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

template <int VAR>
__global__ void __launch_bounds__(64) k_mb(const double* A, double* out, unsigned long long* cyc, int reps) {
  __shared__ double S[TILE];
  __shared__ double D[256];
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
    if (VAR == 4) ok = diag16_valu(S) && ok;
    else ok = diag16_var<VAR>(S, D) && ok;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; out[256] = ok ? 1.0 : 0.0; }
  for (int e = threadIdx.x; e < 256; e += 64) out[e] = S[(e & 15) + (e >> 4) * LD];
}

int main() {
  std::vector<double> h(256);
  for (int i = 0; i < 16; ++i)
    for (int k = 0; k < 16; ++k) h[i + 16 * k] = (i == k) ? 20.0 + i : 1.0 / (1.0 + abs(i - k));
  double *A, *out; unsigned long long* cyc;
  hipMalloc(&A, 256 * 8); hipMalloc(&out, 257 * 8); hipMalloc(&cyc, 8);
  hipMemcpy(A, h.data(), 256 * 8, hipMemcpyHostToDevice);
  const int reps = 200;
  auto run = [&](auto kern, const char* name) {
    for (int w = 0; w < 3; ++w) {
      hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, A, out, cyc, reps);
      hipDeviceSynchronize();
    }
    unsigned long long c; std::vector<double> o(257);
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(o.data(), out, 257 * 8, hipMemcpyDeviceToHost);
    printf("%-34s %8.1f cycles / block  %6.1f / pivot  ok=%g L[0][0]=%.6f L[15][14]=%.9f L[15][15]=%.9f\n", name,
           (double)c / reps, (double)c / reps / 16, o[256], o[0], o[15 + 16 * 14], o[15 + 16 * 15]);
  };
  run(k_mb<0>, "mfma S+M, rcp+2NR (current)");
  run(k_mb<1>, "mfma S only, rcp+2NR");
  run(k_mb<2>, "mfma S+M, rcp only");
  run(k_mb<3>, "mfma S only, rcp only");
  run(k_mb<4>, "VALU rows + readlane");
  return 0;
}
