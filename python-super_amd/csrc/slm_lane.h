// slm_lane.h -- cross-lane reduction of the back substitution's 16 partial sums per thread.
#pragma once
#include <hip/hip_runtime.h>

// Each thread holds v[0..15]; wanted: s_j = sum over the 64 lanes of v[j], j = 0..15.  A halving butterfly: at the step
// with lane bit O and CNT pairs, lanes with the bit clear keep v[i] and take the partner's v[i], lanes with the bit set
// keep v[i + CNT] and take the partner's v[i + CNT]; after the steps 32, 16, 8, 4 one value is left per thread, summed
// over the last two lane bits.  Result: s_j in the lanes with (lane >> 2) == j' where j' is j's bits reversed in the
// order the steps consumed them -- the callers only use "thread (l & 3) == 0 holds the sum for index ((l >> 2) & 15)"
// in THEIR numbering of v[], which both forms below share.

// the portable form: 17 ds_bpermute exchanges of a double, each behind an LDS round trip (~0.5 us per reduction)
__device__ __forceinline__ double col_reduce16_shfl(double v[16]) {
  const int l = threadIdx.x & 63;
#define CR_STEP(CNT, O)                                                         \
  _Pragma("unroll") for (int i = 0; i < (CNT); ++i) {                           \
    const bool up = (l & (O)) != 0;                                             \
    const double send = up ? v[i] : v[i + (CNT)];                               \
    const double keep = up ? v[i + (CNT)] : v[i];                               \
    v[i] = keep + __shfl_xor(send, (O), 64);                                    \
  }
  CR_STEP(8, 32)
  CR_STEP(4, 16)
  CR_STEP(2, 8)
  CR_STEP(1, 4)
#undef CR_STEP
  double r = v[0];
  r += __shfl_xor(r, 2, 64);
  r += __shfl_xor(r, 1, 64);
  return r;
}

// ---- VALU-only exchanges (gfx950) ----
// v_permlane32_swap a, b: a's lanes 32..63 <-> b's lanes 0..31.  Afterwards a + b is, in lanes 0..31, a[l] + a[l + 32]
// and in lanes 32..63 b[l - 32] + b[l]: exactly one butterfly step on the pair (a, b) with no select.
__device__ __forceinline__ double lane_fold32(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// v_permlane16_swap a, b: a's odd rows (of 16 lanes) <-> b's even rows: the same for lane bit 16
__device__ __forceinline__ double lane_fold16(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
template <int CTRL>
__device__ __forceinline__ double lane_dpp(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// a butterfly step through DPP: the partner is lane ^ 8 (row_ror:8) or lane ^ 7 (row_half_mirror -- bit 2 flips, and
// the low two bits are summed over afterwards anyway)
template <int CTRL, int BIT>
__device__ __forceinline__ double lane_fold_dpp(double a, double b) {
  const bool up = (threadIdx.x & BIT) != 0;
  const double send = up ? a : b, keep = up ? b : a;
  return keep + lane_dpp<CTRL>(send);
}

__device__ __forceinline__ double col_reduce16(double v[16]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = lane_fold32(v[i], v[i + 8]);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = lane_fold16(v[i], v[i + 4]);
#pragma unroll
  for (int i = 0; i < 2; ++i) v[i] = lane_fold_dpp<0x128, 8>(v[i], v[i + 2]);   // row_ror:8
  double r = lane_fold_dpp<0x141, 4>(v[0], v[1]);                                // row_half_mirror
  r += lane_dpp<0x4E>(r);                                                       // quad_perm [2,3,0,1]: lane ^ 2
  r += lane_dpp<0xB1>(r);                                                       // quad_perm [1,0,3,2]: lane ^ 1
  return r;
}

// The same for the 16-byte-per-lane tile layout (load_tile_regs2 in slm_dag.hip): a thread holds 8 partial sums (its two
// inner indices already added), the inner index runs over the 32 lanes l & 31, lane bit 5 belongs to the OUTPUT index.
// Steps over the lane bits 16, 8, 4, then the sum over the last two.  Result, in the lanes with (l & 3) == 0: the sum
// of v[(l >> 2) & 7] over the 32 lanes of the thread's half wave.
__device__ __forceinline__ double col_reduce8(double v[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = lane_fold16(v[i], v[i + 4]);
#pragma unroll
  for (int i = 0; i < 2; ++i) v[i] = lane_fold_dpp<0x128, 8>(v[i], v[i + 2]);   // row_ror:8
  double r = lane_fold_dpp<0x141, 4>(v[0], v[1]);                                // row_half_mirror
  r += lane_dpp<0x4E>(r);
  r += lane_dpp<0xB1>(r);
  return r;
}
