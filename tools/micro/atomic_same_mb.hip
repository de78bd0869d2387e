// Diagnostic micro-benchmark (round 6): what does ONE device-scope atomic per workgroup cost when every workgroup of a launch
// hits the SAME address (a ticket counter, a block-partial sum, a "blocks done" counter), on gfx950 / ROCm 7.2?
// N workgroups of 256 threads; thread 0 issues the atomic after a short body; variants: returning int fetch-add
// (ticket), non-returning f64 add (partial sum), the same spread over 64 addresses, and no atomic at all.
//   hipcc --offload-arch=gfx950 -O3 atomic_same_mb.hip -o atomic_same_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_one(int* ctr, double* acc, int* sink) {
  __shared__ int s;
  int v = 0;
  if (threadIdx.x == 0) {
    if (MODE == 1) v = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 2) unsafeAtomicAdd(acc, 1.0);
    if (MODE == 3) v = __hip_atomic_fetch_add(ctr + 32 * (blockIdx.x & 63), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 4) unsafeAtomicAdd(acc + 16 * (blockIdx.x & 63), 1.0);
    s = v;
  }
  __syncthreads();
  if (s == -12345) sink[threadIdx.x] = s;   // (keeps the returning form's result live)
}

// persistent ticket loop: every workgroup takes tickets until the counter passes `total` (the task graph's pattern)
template <int SPREAD>
__global__ void __launch_bounds__(256) k_tickets(int* ctr, int total, int* sink) {
  __shared__ int s;
  int taken = 0;
  for (;;) {
    if (threadIdx.x == 0) s = __hip_atomic_fetch_add(ctr + (SPREAD ? 32 * (blockIdx.x & 7) : 0), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int t = s;
    __syncthreads();
    if (t >= (SPREAD ? total / 8 : total)) break;
    ++taken;
  }
  if (taken < 0) sink[0] = taken;
}

int main() {
  int *ctr, *sink;
  double* acc;
  CK(hipMalloc((void**)&ctr, 4096 * sizeof(int)));
  CK(hipMalloc((void**)&sink, 4096 * sizeof(int)));
  CK(hipMalloc((void**)&acc, 4096 * sizeof(double)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int sizes[] = {512, 1024, 4096, 16384};
  for (int n : sizes) {
    float ms[5] = {0, 0, 0, 0, 0};
    for (int mode = 0; mode < 5; ++mode) {
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctr, 0, 4096 * sizeof(int)));
        CK(hipMemset(acc, 0, 4096 * sizeof(double)));
        CK(hipEventRecord(e0, 0));
        switch (mode) {
          case 0: hipLaunchKernelGGL(k_one<0>, dim3(n), dim3(256), 0, 0, ctr, acc, sink); break;
          case 1: hipLaunchKernelGGL(k_one<1>, dim3(n), dim3(256), 0, 0, ctr, acc, sink); break;
          case 2: hipLaunchKernelGGL(k_one<2>, dim3(n), dim3(256), 0, 0, ctr, acc, sink); break;
          case 3: hipLaunchKernelGGL(k_one<3>, dim3(n), dim3(256), 0, 0, ctr, acc, sink); break;
          default: hipLaunchKernelGGL(k_one<4>, dim3(n), dim3(256), 0, 0, ctr, acc, sink); break;
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[mode], e0, e1));
      }
    }
    printf("%6d workgroups: none %.1f us | int fetch-add, one address %.1f | f64 add, one address %.1f | fetch-add over 64 addresses %.1f | f64 add over 64 %.1f\n",
           n, ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3, ms[3] * 1e3, ms[4] * 1e3);
  }
  for (int total : {4096, 16384}) {
    float a = 0, b = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(ctr, 0, 4096 * sizeof(int)));
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_tickets<0>, dim3(512), dim3(256), 0, 0, ctr, total, sink);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&a, e0, e1));
      CK(hipMemset(ctr, 0, 4096 * sizeof(int)));
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_tickets<1>, dim3(512), dim3(256), 0, 0, ctr, total, sink);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&b, e0, e1));
    }
    printf("ticket loop, 512 persistent workgroups, %d empty tasks: one counter %.1f us (%.0f ns per ticket) | 8 counters %.1f us\n", total,
           a * 1e3, a * 1e6 / total, b * 1e3);
  }
  return 0;
}
