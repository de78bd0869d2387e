// Micro-test: VALU-only lane exchanges on gfx950 (v_permlane32_swap / v_permlane16_swap, DPP row_ror:8, row_half_mirror,
// quad_perm) against __shfl_xor (ds_bpermute), values and timing.   hipcc --offload-arch=gfx950 -O3 ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

__device__ __forceinline__ double xch32(double x) {
  const int l = threadIdx.x & 63;
  unsigned lo = __double2loint(x), hi = __double2hiint(x);
  auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const unsigned rl = l >= 32 ? a[0] : a[1], rh = l >= 32 ? b[0] : b[1];
  return __hiloint2double(rh, rl);
}
__device__ __forceinline__ double xch16(double x) {
  const int l = threadIdx.x & 63;
  unsigned lo = __double2loint(x), hi = __double2hiint(x);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const bool odd = (l >> 4) & 1;
  const unsigned rl = odd ? a[0] : a[1], rh = odd ? b[0] : b[1];
  return __hiloint2double(rh, rl);
}
template <int CTRL>
__device__ __forceinline__ double dpp(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__global__ void k_check(double* out) {
  const int l = threadIdx.x;
  const double x = 100.0 + l;
  out[0 * 64 + l] = xch32(x);        // expect 100 + (l ^ 32)
  out[1 * 64 + l] = xch16(x);        // l ^ 16
  out[2 * 64 + l] = dpp<0x128>(x);   // row_ror:8  -> l ^ 8
  out[3 * 64 + l] = dpp<0x141>(x);   // row_half_mirror -> l ^ 7
  out[4 * 64 + l] = dpp<0x1B>(x);    // quad_perm [3,2,1,0] -> l ^ 3
  out[5 * 64 + l] = dpp<0xB1>(x);    // quad_perm [1,0,3,2] -> l ^ 1
}
__global__ void k_time(double* out, int reps, int mode) {
  double v = threadIdx.x;
  for (int i = 0; i < reps; ++i) {
    if (mode == 0) v += __shfl_xor(v, 32, 64) + __shfl_xor(v, 16, 64) + __shfl_xor(v, 8, 64) + __shfl_xor(v, 4, 64);
    else v += xch32(v) + xch16(v) + dpp<0x128>(v) + dpp<0x141>(v);
  }
  out[threadIdx.x] = v;
}
int main() {
  double* d; hipMalloc(&d, 6 * 64 * 8);
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d);
  double h[6 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const int masks[6] = {32, 16, 8, 7, 3, 1};
  for (int t = 0; t < 6; ++t) {
    int bad = 0;
    for (int l = 0; l < 64; ++l) bad += h[t * 64 + l] != 100.0 + (l ^ masks[t]);
    printf("xor %2d: %s (lane0 got %g, lane 5 got %g)\n", masks[t], bad ? "MISMATCH" : "ok", h[t * 64] - 100, h[t * 64 + 5] - 100);
  }
  for (int mode = 0; mode < 2; ++mode) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_time, dim3(1), dim3(256), 0, 0, d, 100000, mode);
    hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%s: %.1f ns per group of 4 double exchanges (256 threads)\n", mode ? "permlane/dpp" : "shfl_xor    ", us * 1e3 / 100000);
  }
  return 0;
}
