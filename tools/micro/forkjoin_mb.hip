// Micro-benchmark: what does a cross-stream dependency cost on this runtime?  Stream 1: A (100 us) - [record E1] - C; stream 2:
// [wait E1] - B (100 us) - [record E2]; stream 1 waits E2 before D.  Prints, in 10 ns ticks relative to A's start:
// A end, B start (fork latency = B start - A end), C start (same-stream successor), B end, D start (join latency).
// Also the plain same-stream gap for comparison.   hipcc --offload-arch=gfx950 -O3 -o bin/forkjoin_mb forkjoin_mb.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_spin(long long ticks, long long* out, int slot) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (out && threadIdx.x == 0 && blockIdx.x == 0) { out[2 * slot] = t0; out[2 * slot + 1] = wall_clock64(); }
}

int main() {
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t e1, e2;
  hipEventCreateWithFlags(&e1, hipEventDisableTiming);
  hipEventCreateWithFlags(&e2, hipEventDisableTiming);
  long long* d;
  hipMalloc(&d, 64 * sizeof(long long));
  long long h[64];
  for (int rep = 0; rep < 5; ++rep) {
    hipDeviceSynchronize();
    for (int i = 0; i < 5; ++i) {
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 10000ll, d, 0);   // A
      hipEventRecord(e1, s1);
      hipStreamWaitEvent(s2, e1, 0);
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s2, 10000ll, d, 1);   // B (other stream, after A)
      hipEventRecord(e2, s2);
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 2000ll, d, 2);    // C (same stream, after A)
      hipStreamWaitEvent(s1, e2, 0);
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 100ll, d, 3);     // D (after B and C)
    }
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("A end %lld | B start %lld (fork +%lld) | C start %lld (in-order +%lld) | B end %lld | D start %lld (join +%lld)\n", h[1] - h[0],
           h[2] - h[0], h[2] - h[1], h[4] - h[0], h[4] - h[1], h[3] - h[0], h[6] - h[0], h[6] - h[3]);
  }
  return 0;
}
