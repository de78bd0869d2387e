// Micro-benchmark: do kernels from two HIP streams overlap on this GPU / runtime?
//   A: chain of small latency-bound kernels (1..64 workgroups spinning ~20 us) on stream 1
//   B: chain of wide bandwidth-bound kernels (copy of `mb` MB) on stream 2
// prints the time of A alone, B alone and A || B.   hipcc --offload-arch=gfx950 -O3 -o bin/stream_overlap_mb ...
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void k_spin(long long cycles, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 0 && blockIdx.x == 1 << 30) *sink = 1;
}
__global__ void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int nA = 200, nB = 40;
  const size_t mb = argc > 1 ? atol(argv[1]) : 256;
  const int wgA = argc > 2 ? atoi(argv[2]) : 8;
  const size_t n = mb * (1 << 20) / sizeof(double2);
  double2 *a, *b;
  hipMalloc(&a, n * sizeof(double2)); hipMalloc(&b, n * sizeof(double2));
  hipMemset(a, 0, n * sizeof(double2));
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  const long long cyc = 2000;   // wall_clock64 ticks at 100 MHz -> 20 us
  auto runA = [&] { for (int i = 0; i < nA; ++i) hipLaunchKernelGGL(k_spin, dim3(wgA), dim3(256), 0, s1, cyc, (int*)nullptr); };
  auto runB = [&] { for (int i = 0; i < nB; ++i) hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, s2, a, b, n); };
  for (int rep = 0; rep < 2; ++rep) {
    hipDeviceSynchronize();
    double t0 = now(); runA(); hipDeviceSynchronize(); double tA = now() - t0;
    t0 = now(); runB(); hipDeviceSynchronize(); double tB = now() - t0;
    t0 = now(); runA(); runB(); hipDeviceSynchronize(); double tAB = now() - t0;
    t0 = now();
    for (int i = 0; i < nA; ++i) {   // interleaved enqueue
      hipLaunchKernelGGL(k_spin, dim3(wgA), dim3(256), 0, s1, cyc, (int*)nullptr);
      if (i % (nA / nB) == 0) hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, s2, a, b, n);
    }
    hipDeviceSynchronize();
    double tI = now() - t0;
    printf("A alone (%d x %d WG x 20us) %.2f ms | B alone (%d copies of %zu MB) %.2f ms (%.0f GB/s) | A||B %.2f ms | interleaved enqueue %.2f ms\n",
           nA, wgA, tA * 1e3, nB, mb, tB * 1e3, 2.0 * nB * mb / 1024.0 / tB, tAB * 1e3, tI * 1e3);
  }
  return 0;
}
