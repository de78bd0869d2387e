// Micro-benchmark: what does a second workgroup on the CU cost a latency-bound pivot chain (slm_tile.h::diag16 on ONE wave)?
// k_fL11 takes 44 us at one workgroup per CU and 80 us at two (both resident, both start at t = 0: docs/LAB_NOTEBOOK.md).
//   mode 0: 64-thread workgroups, every wave runs the chain
//   mode 1: 256-thread workgroups, wave 0 runs the chain, waves 1-3 wait at the closing barrier
//   mode 2: 256-thread workgroups, wave 0 runs the chain, waves 1-3 poll an LDS flag (s_sleep 1 between polls) as the trailing
//           waves of factor_inverse64p do
// each with 256 workgroups (one per CU) and 512 (two per CU; 80 KB of dynamic LDS each keeps a third one away).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I python-super_amd/csrc tools/micro/chain_pair_mb.hip -o tools/micro/bin/chain_pair_mb
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include "slm_tile.h"

extern __shared__ double dyn[];

template <int MODE>
__global__ void __launch_bounds__(256) k_chain(const double* A, double* out, long long* stamps, int reps) {
  double* S = dyn;              // 16 x 16 block, ld LD
  double* D = dyn + TILE;       // 256
  double* X = D + 256;          // 128
  int* flag = reinterpret_cast<int*>(X + 128);
  const int w = threadIdx.x >> 6;
  if (threadIdx.x == 0) *flag = 0;
  __syncthreads();
  const long long t0 = wall_clock64();
  bool ok = true;
  if (w == 0) {
    for (int it = 0; it < reps; ++it) {
      for (int e = threadIdx.x; e < 256; e += 64) S[(e & 15) + (e >> 4) * LD] = A[e];
      wave_sync();
      ok = diag16(S, D, X) && ok;
    }
    if (MODE == 2) lds_signal(flag);
  } else if (MODE == 2) {
    lds_wait_all(flag, 1);
  }
  if (MODE != 0) __syncthreads();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = wall_clock64() - t0;
    stamps[2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_getreg(63492);   // HW_ID: simd in bits 5:4, cu 11:8
    out[blockIdx.x] = ok ? S[0] : -1.0;
  }
}

// mode 3: the whole pipelined tile factorisation (factor_inverse64p, 256 threads) on an LDS-resident tile, `reps` times
// mode 4: the memory side of k_fL11 only: `reps` dependent (global tile -> LDS -> global tile) round trips of 32 KB, no arithmetic
template <int MODE>
__global__ void __launch_bounds__(256, 2) k_tile(const double* A, double* G, double* out, long long* stamps, int reps) {
  double* S = dyn;
  double* M = dyn + TILE;
  double* dinv = dyn + 2 * TILE;
  double* wt = dinv + 4 * 256;
  double* xch = wt + 3 * 256;
  int* s_ok = reinterpret_cast<int*>(xch + 2 * NB);
  int* pf = s_ok + 8;
  const long long t0 = wall_clock64();
  bool ok = true;
  double* mine = G + (size_t)blockIdx.x * 2 * TILE;
  for (int it = 0; it < reps; ++it) {
    if (MODE == 3) {
      for (int e = threadIdx.x; e < TILE; e += 256) {
        const int i = e & 63, k = e >> 6;
        S[e] = (i == k) ? 100.0 + i : ((i > k) ? 1.0 / (1.0 + i - k) : 0.0);
      }
      __syncthreads();
      ok = factor_inverse64p(S, M, dinv, wt, xch, s_ok, pf) && ok;
    } else {
      double v[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = mine[(it & 1) * TILE + threadIdx.x + 256 * t];
#pragma unroll
      for (int t = 0; t < 16; ++t) S[threadIdx.x + 256 * t] = v[t] + 1.0;
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 16; ++t) mine[((it + 1) & 1) * TILE + threadIdx.x + 256 * t] = S[(threadIdx.x + 256 * t + 64) & (TILE - 1)];
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = wall_clock64() - t0;
    stamps[2 * blockIdx.x + 1] = 0;
    out[blockIdx.x] = ok ? M[5] : -1.0;
  }
}

template <int MODE>
static void run_tile(const double* A, double* G, double* out, long long* stamps, int n_wg) {
  const int lds = 81024, reps = 10;
  hipFuncSetAttribute((const void*)k_tile<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  std::vector<long long> h(2 * n_wg);
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL(k_tile<MODE>, dim3(n_wg), dim3(256), lds, 0, A, G, out, stamps, reps);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * n_wg, hipMemcpyDeviceToHost);
  double mean = 0, mx = 0;
  for (int i = 0; i < n_wg; ++i) { mean += h[2 * i] / 100.0 / n_wg; mx = std::max(mx, h[2 * i] / 100.0); }
  printf("mode %d  %4d workgroups: per workgroup %.2f us mean / %.2f max for %d steps (%.2f us each)\n", MODE, n_wg, mean, mx, reps, mean / reps);
}

template <int MODE>
static void run(const double* A, double* out, long long* stamps, int n_wg) {
  const int lds = 80 * 1024, reps = 20;
  hipFuncSetAttribute((const void*)k_chain<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  std::vector<long long> h(2 * n_wg);
  double best = 1e9;
  for (int r = 0; r < 4; ++r) {
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_chain<MODE>, dim3(n_wg), dim3(MODE == 0 ? 64 : 256), lds, 0, A, out, stamps, reps);
    hipDeviceSynchronize();
    best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * n_wg, hipMemcpyDeviceToHost);
  double mean = 0, mx = 0;
  int simd[4] = {0, 0, 0, 0};
  for (int i = 0; i < n_wg; ++i) {
    mean += h[2 * i] / 100.0 / n_wg;
    mx = std::max(mx, h[2 * i] / 100.0);
    simd[(h[2 * i + 1] >> 4) & 3]++;
  }
  printf("mode %d  %4d workgroups: per workgroup %.2f us mean / %.2f max for %d chained diag16 (%.2f us each); chain waves on SIMD 0..3: %d %d %d %d; launch %.1f us\n",
         MODE, n_wg, mean, mx, reps, mean / reps, simd[0], simd[1], simd[2], simd[3], best * 1e6);
}

int main() {
  std::vector<double> h(256);
  for (int i = 0; i < 16; ++i)
    for (int k = 0; k < 16; ++k) h[i + 16 * k] = (i == k) ? 20.0 + i : 1.0 / (1.0 + abs(i - k));
  double *A, *out;
  long long* stamps;
  hipMalloc(&A, 256 * 8);
  hipMalloc(&out, 1024 * 8);
  hipMalloc(&stamps, 2048 * 8);
  hipMemcpy(A, h.data(), 256 * 8, hipMemcpyHostToDevice);
  for (int n : {256, 512}) run<0>(A, out, stamps, n);
  for (int n : {256, 512}) run<1>(A, out, stamps, n);
  for (int n : {256, 512}) run<2>(A, out, stamps, n);
  double* G;
  hipMalloc(&G, (size_t)512 * 2 * TILE * 8);
  hipMemset(G, 0, (size_t)512 * 2 * TILE * 8);
  for (int n : {256, 512}) run_tile<3>(A, G, out, stamps, n);
  for (int n : {256, 512}) run_tile<4>(A, G, out, stamps, n);
  return 0;
}
