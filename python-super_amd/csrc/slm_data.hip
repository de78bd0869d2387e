// slm_data.hip -- data-term passes (point-to-plane ICP), v0: one thread per surfel.
//
//   k_data_grad   : JtJ (lower band) and jtl = -Jt r accumulation  (loss.py:222-288)
//   k_data_loss   : sum r^2 at beta (+ delta), fresh match set     (loss.py:222-255,290)
//   k_data_resid  : per-surfel r / match / taps for parity tests
#include "slm_data.h"

// grid = (ceil(maxN/256), n_frames); KK = opt.num_neighbors of every slot of the launch (a slot with another K is skipped:
// the launcher only sees the batch's common value)
template <int KK>
__global__ void __launch_bounds__(256) k_data_grad(const FrameDev* __restrict__ frames, double lam) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || fd.st->stopped || fd.f.K != KK) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  SurfelEvalT<KK> ev;
  ev.match = false;
  if (i < fd.f.N) eval_surfel<1, KK>(fd, lam, fd.node_pk, i, ev);

  // matched-surfel count: one atomic per wave
  unsigned long long m = __ballot(ev.match);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&fd.st->m_grad, __popcll(m));
  if (!ev.match) return;

  // scatter row^T row into the lower band and -row^T r into rhs
#pragma unroll 1
  for (int a = 0; a < 7 * KK; ++a) {
    const int ia = 7 * ev.id[a / 7] + a % 7;
    const double ja = ev.row[a];
    atomic_add_f64(fd.rhs + ia, -ja * ev.r);
#pragma unroll 1
    for (int b = 0; b < 7 * KK; ++b) {
      const int ib = 7 * ev.id[b / 7] + b % 7;
      if (ia >= ib) atomic_add_f64(band_entry(fd, ia, ib), ja * ev.row[b]);
    }
  }
}

// grid = (n_loss_blocks, n_frames); grid-stride over surfels; partial sums per block
template <int KK>
__global__ void __launch_bounds__(256) k_data_loss(const FrameDev* __restrict__ frames, double lam,
                                                    int use_delta) {
  __shared__ double sm[16];
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || fd.st->stopped || fd.f.K != KK) return;
  double acc = 0.0;
  int cnt = 0;
  // [sf_lo, sf_hi) = all surfels unless the frame is sharded over several GPUs
  for (int i = fd.sf_lo + blockIdx.x * blockDim.x + threadIdx.x; i < fd.sf_hi; i += gridDim.x * blockDim.x) {
    SurfelEvalT<KK> ev;
    eval_surfel<0, KK>(fd, lam, use_delta ? fd.node_pk_try : fd.node_pk, i, ev);
    if (ev.match) {
      acc += ev.r * ev.r;
      ++cnt;
    }
  }
  double s = block_sum(acc, sm);
  double c = block_sum((double)cnt, sm);
  if (threadIdx.x == 0) {
    fd.loss_part[2 * blockIdx.x] = s;
    fd.loss_part[2 * blockIdx.x + 1] = c;
  }
}

template <int KK>
__global__ void __launch_bounds__(256) k_data_resid(const FrameDev* __restrict__ frames, int slot,
                                                     double lam, double* __restrict__ r_out,
                                                     uint8_t* __restrict__ match_out,
                                                     int32_t* __restrict__ taps_out) {
  const FrameDev& fd = frames[slot];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= fd.f.N) return;
  SurfelEvalT<KK> ev;
  eval_surfel<0, KK>(fd, lam, fd.node_pk, i, ev);
  if (r_out) r_out[i] = ev.match ? ev.r : 0.0;
  if (match_out) match_out[i] = ev.match ? 1 : 0;
  if (taps_out) {
    taps_out[4 * i + 0] = ev.taps[0];
    taps_out[4 * i + 1] = ev.taps[1];
    taps_out[4 * i + 2] = ev.taps[2];
    taps_out[4 * i + 3] = ev.taps[3];
  }
}

// ---- host launchers (called from slm_api.hip) ------------------------------------
// K = the batch's num_neighbors (1..SLM_KMAX): one instantiation per value
#define SLM_K_DISPATCH(K, CALL)                                        \
  switch (K) {                                                         \
    case 1: { constexpr int KK = 1; CALL; break; }                     \
    case 2: { constexpr int KK = 2; CALL; break; }                     \
    case 3: { constexpr int KK = 3; CALL; break; }                     \
    case 4: { constexpr int KK = 4; CALL; break; }                     \
    case 5: { constexpr int KK = 5; CALL; break; }                     \
    case 6: { constexpr int KK = 6; CALL; break; }                     \
    case 7: { constexpr int KK = 7; CALL; break; }                     \
    case 8: { constexpr int KK = 8; CALL; break; }                     \
    default: break;                                                    \
  }

void launch_data_grad(const FrameDev* frames_dev, int n_frames, int maxN, int K, double lam, hipStream_t st) {
  if (maxN <= 0) return;
  dim3 grid((maxN + 255) / 256, n_frames);
  SLM_K_DISPATCH(K, hipLaunchKernelGGL(k_data_grad<KK>, grid, dim3(256), 0, st, frames_dev, lam));
}

void launch_data_loss(const FrameDev* frames_dev, int n_frames, int n_blocks, int K, double lam, int use_delta, hipStream_t st) {
  dim3 grid(n_blocks, n_frames);
  SLM_K_DISPATCH(K, hipLaunchKernelGGL(k_data_loss<KK>, grid, dim3(256), 0, st, frames_dev, lam, use_delta));
}

void launch_data_resid(const FrameDev* frames_dev, int slot, int N, int K, double lam, double* r, uint8_t* match, int32_t* taps,
                       hipStream_t st) {
  if (N <= 0) return;
  SLM_K_DISPATCH(K, hipLaunchKernelGGL(k_data_resid<KK>, dim3((N + 255) / 256), dim3(256), 0, st, frames_dev, slot, lam, r, match, taps));
}
