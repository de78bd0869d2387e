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


// ---- K-generic data term on the multifrontal path (num_neighbors != 4) ------------------------------------------------
// The Jacobian row of a surfel has 7K entries; its outer product lands in the K(K+1)/2 node-pair blocks of the surfel.
// One thread per surfel evaluates (as k_data_grad), then the WAVE turns round: the rows of its 64 surfels go to LDS in the
// surfel's canonical neighbour order (ids ascending: slot (ra, rb <= ra) is the pair (c[ra], c[rb]), never transposed) and
// every lane owns a fixed set of ENTRIES -- (slot, ca, cb) of the blocks and (node, c) of J^T r -- which it forms for the
// wave's surfels one after the other, accumulating in a register while the destination record stays the same and issuing
// one f64 atomic when it changes.  Surfels are walked in neighbour-set order (FrameDev::sf_perm), so the surfels of a set
// are consecutive and their common blocks reach memory once per wave instead of once per surfel: K = 6 at C2 issues
// ~945 atomics per RUN of surfels with one neighbour set instead of per surfel.  The records (pairbuf: 49 block entries
// row-major (ca, cb) of the larger-id node's row index first + 7 entries of J^T r of the diagonal pair, as the wgslab
// records of the tuple-sorted path) are placed into the fronts by k_pair_scatter (slm_front.hip).
// grid = (ceil(max positions / 64), n_frames), ONE wave per workgroup
template <int KK>
__global__ void __launch_bounds__(64) k_data_grad_pairs(const FrameDev* __restrict__ frames, double lam) {
  constexpr int NP = KK * (KK + 1) / 2, NR = 7 * KK, NB49 = NP * 49, NE = NB49 + NR, NPL = (NE + 63) / 64;
  constexpr int LDR = (NR + 1) | 1;   // odd row stride: the lanes of one surfel's row spread over the banks
  __shared__ double s_row[64 * LDR];
  __shared__ int s_pi[64 * NP];
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || fd.st->stopped || fd.f.K != KK || !fd.vk_ready) return;
  const int l = threadIdx.x;
  const int pos = fd.sf_lo + blockIdx.x * 64 + l;   // [sf_lo, sf_hi): all positions unless the frame is sharded over several GPUs
  SurfelEvalT<KK> ev;
  ev.match = false;
  int i = 0;
  if (pos < fd.sf_hi) {
    i = fd.sf_perm[pos];
    eval_surfel<1, KK>(fd, lam, fd.node_pk, i, ev);
  }
  const unsigned long long m = __ballot(ev.match);
  if (!m) return;
  // the matched count rides behind the records, spread over SLM_VK_TAIL doubles (k_pair_scatter sums them into m_grad)
  if (l == 0) atomic_add_f64(fd.pairbuf + (size_t)fd.n_blocks * SLM_WREC + (blockIdx.x % SLM_VK_TAIL), (double)__popcll(m));
  if (ev.match) {
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      int rank = 0;
#pragma unroll
      for (int j = 0; j < KK; ++j) rank += (ev.id[j] < ev.id[k]) ? 1 : 0;
#pragma unroll
      for (int c = 0; c < 7; ++c) s_row[l * LDR + 7 * rank + c] = ev.row[7 * k + c];
    }
    s_row[l * LDR + NR] = ev.r;
#pragma unroll
    for (int sl = 0; sl < NP; ++sl) s_pi[l * NP + sl] = fd.sf_pidx[(size_t)NP * i + sl];
  }
  __syncthreads();
  // this lane's entries: e = l + 64 q
  int desc[NPL];   // slot | rowa << 6 | rowb << 12 | off << 18 | active << 24
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
    const int e = l + 64 * q;
    int d = 0;
    if (e < NB49) {
      const int slot = e / 49, rem = e - 49 * slot, ca = rem / 7, cb = rem - 7 * ca;
      int ra = 0;
      while ((ra + 1) * (ra + 2) / 2 <= slot) ++ra;
      const int rb = slot - ra * (ra + 1) / 2;
      if (!(ra == rb && ca < cb))   // (a diagonal pair's block is symmetric: its lower part is what is placed)
        d = slot | (7 * ra + ca) << 6 | (7 * rb + cb) << 12 | (7 * ca + cb) << 18 | 1 << 24;
    } else if (e < NE) {
      const int k = (e - NB49) / 7, c = (e - NB49) - 7 * k;
      d = (k * (k + 1) / 2 + k) | (7 * k + c) << 6 | NR << 12 | (49 + c) << 18 | 1 << 24;
    }
    desc[q] = d;
  }
  double acc[NPL];
  int prev[NPL];
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
    acc[q] = 0.0;
    prev[q] = -1;
  }
  double* pb = fd.pairbuf;
  for (unsigned long long mm = m; mm; mm &= mm - 1) {
    const int sidx = __builtin_ctzll(mm);   // uniform: the wave's matched surfels in list order
    const double* rw = s_row + sidx * LDR;
    const int* pi = s_pi + sidx * NP;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
      const int d = desc[q];
      if (d >> 24) {
        const int dest = pi[d & 63] * SLM_WREC + ((d >> 18) & 63);
        const double v = rw[(d >> 6) & 63] * rw[(d >> 12) & 63];
        if (dest != prev[q]) {
          if (prev[q] >= 0) atomic_add_f64(pb + prev[q], acc[q]);
          prev[q] = dest;
          acc[q] = v;
        } else {
          acc[q] += v;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NPL; ++q)
    if (prev[q] >= 0) atomic_add_f64(pb + prev[q], acc[q]);
}

// zero the pair records (+ the matched count behind them) of the slots that take the K-generic pair path
__global__ void __launch_bounds__(256) k_zero_pairbuf(const FrameDev* __restrict__ frames) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || !fd.vk_ready || !fd.pairbuf || fd.st->stopped) return;
  const size_t n = (size_t)fd.n_blocks * SLM_WREC + SLM_VK_TAIL;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) fd.pairbuf[e] = 0.0;
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

// the K-generic Jacobian pass of the multifrontal path: pair records zeroed, then filled (max_pos = the batch's largest surfel count)
void launch_data_grad_pairs(const FrameDev* frames_dev, int n_frames, int max_pos, int K, double lam, hipStream_t st) {
  if (max_pos <= 0) return;
  hipLaunchKernelGGL(k_zero_pairbuf, dim3(128, n_frames), dim3(256), 0, st, frames_dev);
  dim3 grid((max_pos + 63) / 64, n_frames);
  SLM_K_DISPATCH(K, hipLaunchKernelGGL(k_data_grad_pairs<KK>, grid, dim3(64), 0, st, frames_dev, lam));
}
