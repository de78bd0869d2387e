// Diagnostic micro-benchmark: GPU-side cost of a dependent kernel boundary on gfx950 / ROCm 7.2,
// for N tiny dependent kernels enqueued (a) on a stream, (b) as one captured hipGraph, and (c) one
// persistent kernel whose workgroups hand a token round with sc1 flag stores / polls (1 hop = one
// dependent step).   hipcc --offload-arch=gfx950 -O3 launch_gap_mb.hip -o launch_gap_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_step(double* buf, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = buf[i] * 1.0000001 + 1e-9;
}

// chain of hand-offs: workgroup w waits for flag[w-1] (value = round), touches a 32 KB tile written by its
// predecessor (sc1 stores / sc1 loads), publishes flag[w]
__global__ void __launch_bounds__(256) k_chain(double* tiles, int* flags, int hops, int n_wg, long long* spins) {
  const int w = blockIdx.x;
  long long sp = 0;
  for (int h = w; h < hops; h += n_wg) {
    if (h > 0) {
      if (threadIdx.x == 0) {
        while (__hip_atomic_load(&flags[h - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && sp < (1ll << 26)) {
          __builtin_amdgcn_s_sleep(1);
          ++sp;
        }
      }
      __syncthreads();
    }
    double* src = tiles + (size_t)((h + hops - 1) % hops) * 4096;
    double* dst = tiles + (size_t)h * 4096;
    double v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = __hip_atomic_load(src + threadIdx.x + 256 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int e = 0; e < 16; ++e) __hip_atomic_store(dst + threadIdx.x + 256 * e, v[e] + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&flags[h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) spins[w] = sp;
}

int main() {
  const int N = 400, n = 256 * 64;
  double* buf;
  CK(hipMalloc(&buf, n * sizeof(double)));
  CK(hipMemset(buf, 0, n * sizeof(double)));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms;
  for (int grid : {1, 64}) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_step, dim3(grid), dim3(256), 0, st, buf, n);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("stream  grid %3d: %7.2f us per dependent kernel\n", grid, ms * 1e3 / N);
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_step, dim3(grid), dim3(256), 0, st, buf, n);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("graph   grid %3d: %7.2f us per dependent kernel\n", grid, ms * 1e3 / N);
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  // persistent chain of flag hand-offs with a 32 KB payload each
  const int hops = 2048;
  double* tiles;
  int* flags;
  long long* spins;
  CK(hipMalloc(&tiles, (size_t)hops * 4096 * sizeof(double)));
  CK(hipMalloc(&flags, hops * sizeof(int)));
  CK(hipMalloc(&spins, 1024 * sizeof(long long)));
  for (int n_wg : {2, 8, 64, 256}) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemsetAsync(tiles, 0, (size_t)hops * 4096 * sizeof(double), st));
      CK(hipMemsetAsync(flags, 0, hops * sizeof(int), st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(k_chain, dim3(n_wg), dim3(256), 0, st, tiles, flags, hops, n_wg, spins);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      double last;
      CK(hipMemcpy(&last, tiles + (size_t)(hops - 1) * 4096, sizeof(double), hipMemcpyDeviceToHost));
      printf("flags   %3d workgroups: %7.2f us per hop (32 KB sc1 payload), check %.0f (want %d)\n", n_wg, ms * 1e3 / hops, last, hops);
    }
  }
  return 0;
}
