// col_reduce_mb.hip -- the 16-value lane reduction of the back substitution (slm_dag.hip col_reduce16): ds_bpermute
// butterfly (__shfl_xor) against a VALU-only one (v_permlane32_swap, v_permlane16_swap, DPP).  Checks that both return
// the same sums in the same lanes, and times a dependent chain of reductions.
// build: hipcc --offload-arch=gfx950 -O3 -I python-super_amd/csrc -o tools/micro/bin/col_reduce_mb tools/micro/col_reduce_mb.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "slm_lane.h"

__global__ void k_check(double* out) {
  const int l = threadIdx.x & 63;
  double a[16], b[16];
  for (int e = 0; e < 16; ++e) a[e] = b[e] = 1.0 + 0.001 * l + 0.37 * e + 1e-5 * l * e;
  out[threadIdx.x] = col_reduce16_shfl(a);
  out[256 + threadIdx.x] = col_reduce16(b);
}
// col_reduce8 (what the solver uses since the 16-byte tile layout): lanes with (l & 3) == 0 hold the sum of
// v[(l >> 2) & 7] over the 32 lanes of their half wave
__global__ void k_check8(double* out) {
  const int l = threadIdx.x & 63;
  double a[8], ref = 0.0;
  for (int e = 0; e < 8; ++e) a[e] = 1.0 + 0.001 * l + 0.37 * e + 1e-5 * l * e;
  const int mine = (l >> 2) & 7;
  for (int e = 0; e < 8; ++e) {
    double s = a[e];
    for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);     // sum over the half wave
    if (e == mine) ref = s;
  }
  out[threadIdx.x] = ref;
  out[256 + threadIdx.x] = col_reduce8(a);
}
template <int MODE>
__global__ void k_time(double* out, long long* t, int reps) {
  double v[16];
  for (int e = 0; e < 16; ++e) v[e] = 1.0 + threadIdx.x + e;
  const long long t0 = wall_clock64();
  double r = 0;
  for (int i = 0; i < reps; ++i) {
    double w[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) w[e] = v[e] + r;
    r = MODE ? col_reduce16(w) : col_reduce16_shfl(w);
  }
  const long long t1 = wall_clock64();
  out[threadIdx.x] = r;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main() {
  double* d; long long* t;
  (void)hipMalloc(&d, 512 * 8); (void)hipMalloc(&t, 8);
  hipLaunchKernelGGL(k_check, dim3(1), dim3(256), 0, 0, d);
  double h[512]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0; double worst = 0;
  for (int i = 0; i < 256; ++i) if ((i & 3) == 0) { const double e = fabs(h[i] - h[256 + i]); worst = e > worst ? e : worst; bad += e > 1e-9 * fabs(h[i]); }
  printf("result lanes ((lane & 3) == 0): %s, largest difference %.3g (sum %.6f)\n", bad ? "MISMATCH" : "ok", worst, h[0]);
  hipLaunchKernelGGL(k_check8, dim3(1), dim3(256), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  bad = 0; worst = 0;
  for (int i = 0; i < 256; ++i) if ((i & 3) == 0) { const double e = fabs(h[i] - h[256 + i]); worst = e > worst ? e : worst; bad += e > 1e-9 * fabs(h[i]); }
  printf("col_reduce8 result lanes: %s, largest difference %.3g (sum %.6f)\n", bad ? "MISMATCH" : "ok", worst, h[0]);
  for (int mode = 0; mode < 2; ++mode) {
    const int reps = 2000;
    if (mode) hipLaunchKernelGGL(k_time<1>, dim3(1), dim3(256), 0, 0, d, t, reps); else hipLaunchKernelGGL(k_time<0>, dim3(1), dim3(256), 0, 0, d, t, reps);
    long long c; (void)hipMemcpy(&c, t, 8, hipMemcpyDeviceToHost);
    printf("%s: %.3f us per reduction of 16 values (4 waves)\n", mode ? "permlane swap + DPP  " : "__shfl_xor butterfly ", c / 100.0 / reps);
  }
  return 0;
}
