// Diagnostic micro-benchmark: dependent-chain latencies of v_mfma_f64_16x16x4_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ void __launch_bounds__(64) k(double* out, unsigned long long* cyc, int reps, double seed) {
  double4_t acc = {seed, seed, seed, seed}, acc2 = {seed, 1, 2, 3};
  double a = seed * 1e-3, b = 1.0 + seed * 1e-3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (VAR == 0) {            // accumulator dependency only
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      } else if (VAR == 1) {     // result -> VALU -> A operand of the next MFMA
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        a = acc[u & 3] * 1e-3;
      } else if (VAR == 2) {     // two independent accumulators, same A (S and M of diag16)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
        a = acc[u & 3] * 1e-3;
      } else if (VAR == 3) {     // independent MFMAs (throughput)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc2, 0, 0, 0);
      } else if (VAR == 4) {     // dependent f64 FMA chain (VALU)
        a = fma(a, b, 1e-3);
      } else if (VAR == 5) {     // v_rcp_f64 dependent chain
        a = __builtin_amdgcn_rcp(a) + 1.0;
      } else if (VAR == 7) {     // 10 x (2 v_readlane) feeding one dependent VALU op (pivot-block broadcast)
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
          int lo = __builtin_amdgcn_readlane(__double2loint(a), (3 * k + u) & 63);
          int hi = __builtin_amdgcn_readlane(__double2hiint(a), (3 * k + u) & 63);
          t += __hiloint2double(hi, lo);
        }
        a = a * 0.5 + t * 1e-3;
      } else if (VAR == 8) {     // 4 f64 shuffles (8 ds_bpermute) feeding one dependent VALU op
        const int l = threadIdx.x & 63;
        double t = __shfl(a, l & 15, 64) + __shfl(a, 16 + (l & 15), 64) + __shfl(a, 32 + (l & 15), 64) + __shfl(a, 48 + (l & 15), 64);
        a = a * 0.5 + t * 1e-3;
      } else if (VAR == 9) {     // dependent v_rsq_f64 + 1 Newton step
        const double r = __builtin_amdgcn_rsq(a);
        a = r * fma(-(0.5 * a) * r, r, 1.5) + 1.0;
      } else if (VAR == 6) {     // readlane -> VALU chain
        int lo = __builtin_amdgcn_readlane(__double2loint(a), u);
        int hi = __builtin_amdgcn_readlane(__double2hiint(a), u);
        a = a * __hiloint2double(hi, lo) + 1e-3;
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  out[threadIdx.x] = acc[0] + acc2[1] + a;
}

int main() {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
  const int reps = 500;
  auto run = [&](auto kern, const char* name, int per) {
    for (int w = 0; w < 3; ++w) { hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, out, cyc, reps, 1.0); (void)hipDeviceSynchronize(); }
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-52s %7.1f cycles per step\n", name, (double)c / reps / 16);
  };
  run(k<0>, "mfma f64 16x16x4: acc-dependent chain", 1);
  run(k<1>, "mfma -> v_mul -> A operand of next mfma", 1);
  run(k<2>, "2 mfma (shared A) -> v_mul -> next (diag16 shape)", 1);
  run(k<3>, "2 independent mfma (throughput)", 1);
  run(k<4>, "dependent v_fma_f64", 1);
  run(k<5>, "dependent v_rcp_f64 + v_add_f64", 1);
  run(k<6>, "2 v_readlane + v_fma_f64 chain", 1);
  run(k<7>, "20 v_readlane (10 doubles) + 10 adds + fma", 1);
  run(k<8>, "4 f64 __shfl (8 ds_bpermute) + adds + fma", 1);
  run(k<9>, "dependent v_rsq_f64 + 1 NR step + add", 1);
  return 0;
}
