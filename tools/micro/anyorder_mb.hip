// Micro-benchmark: does hipExtAnyOrderLaunch (hipExtLaunchKernelGGL, flags = 1: no barrier bit on the dispatch packet) let a
// kernel start while the PREVIOUS kernel of the same stream is still running on gfx950 / this runtime?
//   pairs of spinning kernels (1 workgroup, ~100 us each): normal + normal should take ~200 us per pair,
//   normal + any-order ~100 us if the second one overlaps the first.
// Also: a normal kernel AFTER an any-order one must still wait for both (it carries the barrier bit).
//   hipcc --offload-arch=gfx950 -O3 -o bin/anyorder_mb anyorder_mb.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void k_spin(long long ticks, long long* out, int slot) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (out && threadIdx.x == 0 && blockIdx.x == 0) { out[2 * slot] = t0; out[2 * slot + 1] = wall_clock64(); }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  long long* d;
  hipMalloc(&d, 64 * sizeof(long long));
  long long h[64];
  const long long T = 10000;   // 100 us at 100 MHz
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipDeviceSynchronize();
      const double t0 = now();
      for (int i = 0; i < 10; ++i) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, T, d, 0);
        if (mode == 0) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, T, d, 1);
        else hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, T, d, 1);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, (long long)100, d, 2);   // normal: must start after BOTH
      }
      hipDeviceSynchronize();
      const double dt = now() - t0;
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("%s: 10 x (A, B, c) %.3f ms  | last triple: A %lld..%lld  B %lld..%lld  c starts %lld (ticks of 10 ns, relative to A's start)\n",
             mode ? "B any-order" : "all in order", dt * 1e3, 0ll, h[1] - h[0], h[2] - h[0], h[3] - h[0], h[4] - h[0]);
    }
  }
  return 0;
}
