// Diagnostic micro-benchmark for the mixed-precision study (tools/studies/f32_factor_study.py):
// what would a float32 tile factorisation gain on the CRITICAL PATH of the multifrontal solver?  One wave, dependent
// chains, float32 against float64, of the pieces the 16 x 16 diagonal-block factorisation (slm_tile.h diag16) is made
// of: the MFMA (16x16x4), the fused multiply-add, the reciprocal square root + Newton step, and one emulated 4-pivot
// block step of diag16 (4 rsq, a 10-entry Cholesky of the pivot block, two 4-term forward substitutions, 2 MFMAs).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/f32_vs_f64_mb.hip -o tools/micro/bin/f32_vs_f64_mb && tools/micro/bin/f32_vs_f64_mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <typename T> struct V4;
template <> struct V4<double> { typedef double4_t t; };
template <> struct V4<float> { typedef float4_t t; };
__device__ __forceinline__ double4_t mma(double a, double b, double4_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float4_t mma(float a, float b, float4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double rsq1(double p) { const double r = __builtin_amdgcn_rsq(p); return r * fma(-(0.5 * p) * r, r, 1.5); }
__device__ __forceinline__ float rsq1(float p) { const float r = __builtin_amdgcn_rsqf(p); return r * fmaf(-(0.5f * p) * r, r, 1.5f); }
template <typename T> __device__ __forceinline__ T tfma(T a, T b, T c);
template <> __device__ __forceinline__ double tfma(double a, double b, double c) { return fma(a, b, c); }
template <> __device__ __forceinline__ float tfma(float a, float b, float c) { return fmaf(a, b, c); }

template <typename T, int VAR>
__global__ void __launch_bounds__(64) k(T* out, unsigned long long* cyc, int reps, T seed) {
  typedef typename V4<T>::t v4;
  v4 acc = {seed, seed, seed, seed}, acc2 = {seed, 1, 2, 3};
  T a = seed * (T)1e-3, b = (T)1.0 + seed * (T)1e-3;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (VAR == 0) {            // MFMA, accumulator-dependent chain
        acc = mma(a, b, acc);
      } else if (VAR == 1) {     // 2 independent MFMAs (issue rate)
        acc = mma(a, b, acc);
        acc2 = mma(b, a, acc2);
      } else if (VAR == 2) {     // dependent fma chain
        a = tfma(a, b, (T)1e-3);
      } else if (VAR == 3) {     // dependent rsq + Newton step
        a = rsq1(a) + (T)1.0;
      } else if (VAR == 4) {     // one 4-pivot block step of diag16: pivot-block Cholesky, two forward substitutions, 2 MFMAs
        const T p00 = a + (T)4, p10 = a * (T).1, p11 = b + (T)4, p20 = a * (T).2, p21 = b * (T).1, p22 = a + (T)5,
                p30 = b * (T).2, p31 = a * (T).3, p32 = b * (T).3, p33 = b + (T)5;
        const T r0 = rsq1(p00);
        const T l10 = p10 * r0, l20 = p20 * r0, l30 = p30 * r0;
        const T d1 = tfma(-l10, l10, p11);
        const T r1 = rsq1(d1);
        const T l21 = tfma(-l20, l10, p21) * r1, l31 = tfma(-l30, l10, p31) * r1;
        const T d2 = tfma(-l21, l21, tfma(-l20, l20, p22));
        const T r2 = rsq1(d2);
        const T l32 = tfma(-l31, l21, tfma(-l30, l20, p32)) * r2;
        const T d3 = tfma(-l32, l32, tfma(-l31, l31, tfma(-l30, l30, p33)));
        const T r3 = rsq1(d3);
        const T s0 = acc[0], s1 = acc[1], s2 = acc[2], s3 = acc[3];
        const T w0 = s0 * r0, w1 = tfma(-l10, w0, s1) * r1, w2 = tfma(-l21, w1, tfma(-l20, w0, s2)) * r2,
                w3 = tfma(-l32, w2, tfma(-l31, w1, tfma(-l30, w0, s3))) * r3;
        const T m0 = acc2[0], m1 = acc2[1], m2 = acc2[2], m3 = acc2[3];
        const T b0 = m0 * r0, b1 = tfma(-l10, b0, m1) * r1, b2 = tfma(-l21, b1, tfma(-l20, b0, m2)) * r2,
                b3 = tfma(-l32, b2, tfma(-l31, b1, tfma(-l30, b0, m3))) * r3;
        const int lq = threadIdx.x >> 4;
        const T W = lq == 0 ? w0 : (lq == 1 ? w1 : (lq == 2 ? w2 : w3));
        const T Bm = lq == 0 ? b0 : (lq == 1 ? b1 : (lq == 2 ? b2 : b3));
        acc = mma(-W, W, acc);
        acc2 = mma(-W, Bm, acc2);
        a = acc[u & 3] * (T)1e-3 + (T)1.0;
        b = acc2[u & 3] * (T)1e-3 + (T)1.0;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  out[threadIdx.x] = acc[0] + acc2[1] + a + b;
}

int main() {
  void* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
  const int reps = 400;
  double res[5][2];
  auto run = [&](auto kern, auto seed, int var, int ty) {
    for (int w = 0; w < 3; ++w) { hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, (decltype(seed)*)out, cyc, reps, seed); (void)hipDeviceSynchronize(); }
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    res[var][ty] = (double)c / reps / 16;
  };
  run(k<double, 0>, 1.0, 0, 0); run(k<float, 0>, 1.0f, 0, 1);
  run(k<double, 1>, 1.0, 1, 0); run(k<float, 1>, 1.0f, 1, 1);
  run(k<double, 2>, 1.0, 2, 0); run(k<float, 2>, 1.0f, 2, 1);
  run(k<double, 3>, 1.0, 3, 0); run(k<float, 3>, 1.0f, 3, 1);
  run(k<double, 4>, 1.0, 4, 0); run(k<float, 4>, 1.0f, 4, 1);
  const char* names[5] = {"mfma 16x16x4, accumulator-dependent chain", "2 independent mfma 16x16x4 (issue rate, per pair)", "dependent fma chain",
                          "dependent rsq + Newton step + add", "one 4-pivot block step of diag16 (emulated)"};
  printf("%-52s %10s %10s %8s\n", "cycles per step (one wave)", "float64", "float32", "ratio");
  for (int v = 0; v < 5; ++v) printf("%-52s %10.1f %10.1f %8.2f\n", names[v], res[v][0], res[v][1], res[v][0] / res[v][1]);
  return 0;
}
