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
// One thread per surfel evaluates (as k_data_grad), then the WAVE forms the Gram matrices of its surfels on the f64 MFMA:
// the rows of its 64 surfels go to LDS in the surfel's canonical neighbour order (ids ascending: slot (ra, rb <= ra) is
// the pair (c[ra], c[rb]), never transposed) with the residual as one more column -- A = [row | r], so that A^T A carries
// J^T r in its last row -- padded to NT 16-column tiles; unmatched surfels are zero rows.  Surfels are walked in
// neighbour-set order (FrameDev::sf_perm), so the surfels of a set are consecutive: per RUN of surfels with one set the
// lower tile pairs of A^T A are accumulated over the run's 4-surfel k-steps (v_mfma_f64_16x16x4_f64; a k-step that
// straddles a run boundary has the other run's surfels masked out of one operand), and every accumulator entry that
// belongs to a block -- (slot, ca, cb) with the larger-id node's row index first, or (node, c) of J^T r -- is added to
// the run's pair records with one f64 atomic.  (Round 6, first form: lane = entry, 7K(7K + 1) / 2 + 7K scalar
// multiply-adds per surfel from LDS -- 0.45 ms per C2 frame at K = 6, bound by that loop, not by the atomics; a
// workgroup-level LDS table of pair records on top of it cut the atomics 4 x and was SLOWER: 0.61 ms,
// tools/studies/r06_patches/pair_table_lds.patch.)
// The records (pairbuf: 49 block entries row-major (ca, cb) + 7 entries of J^T r of the diagonal pair, as the wgslab
// records of the tuple-sorted path) are placed into the fronts by k_pair_scatter (slm_front.hip).
// grid = (ceil(max positions / 64), n_frames), ONE wave per workgroup
template <int KK>
__global__ void __launch_bounds__(64) k_data_grad_pairs(const FrameDev* __restrict__ frames, double lam) {
  typedef double double4_t __attribute__((ext_vector_type(4)));
  constexpr int NP = KK * (KK + 1) / 2, NR = 7 * KK, NC = NR + 1, NT = (NC + 15) / 16, NTP = NT * (NT + 1) / 2;
  constexpr int LDR = 16 * NT + 1;   // odd row stride
  __shared__ double s_row[64 * LDR];
  __shared__ int s_pi[64 * NP];
  __shared__ int s_cid[64 * KK];
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || fd.st->stopped || fd.f.K != KK || !fd.vk_ready) return;
  const int l = threadIdx.x, lr = l & 15, lk = l >> 4;
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
  {
    double* rw = s_row + l * LDR;
#pragma unroll
    for (int c = 0; c < 16 * NT; ++c) rw[c] = 0.0;
    if (ev.match) {
#pragma unroll
      for (int k = 0; k < KK; ++k) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < KK; ++j) rank += (ev.id[j] < ev.id[k]) ? 1 : 0;
#pragma unroll
        for (int c = 0; c < 7; ++c) rw[7 * rank + c] = ev.row[7 * k + c];
        s_cid[l * KK + rank] = ev.id[k];
      }
      rw[NR] = ev.r;
#pragma unroll
      for (int sl = 0; sl < NP; ++sl) s_pi[l * NP + sl] = fd.sf_pidx[(size_t)NP * i + sl];
    }
  }
  __syncthreads();
  // runs: a matched surfel starts one when its neighbour set differs from the previous matched surfel's
  bool start = false;
  if (ev.match) {
    const unsigned long long below = m & ((1ull << l) - 1ull);
    if (!below) start = true;
    else {
      const int pl = 63 - __builtin_clzll(below);
#pragma unroll
      for (int k = 0; k < KK; ++k) start = start || (s_cid[l * KK + k] != s_cid[pl * KK + k]);
    }
  }
  unsigned long long starts = __ballot(start);
  // A run's Gram matrix goes from the accumulators (tile pair tp = (ti, tj <= ti), register r <-> entry (16 ti + lr,
  // 16 tj + lk + 4 r)) through LDS to RECORD order: lane e < 56 owns entry e of every pair record, so one atomic
  // instruction of the wave touches the four cache lines of ONE record instead of a dozen records' (the L2 executes
  // atomics line by line).
  __shared__ double s_g[16 * NT * LDR];
  double* pb = fd.pairbuf;
  const int eca = l / 7, ecb = l - 7 * eca;   // lane -> (ca, cb) of a block entry (l < 49), or component l - 49 of J^T r
  while (starts) {
    const int rs = __builtin_ctzll(starts);   // uniform
    starts &= starts - 1;
    const int re = starts ? __builtin_ctzll(starts) : 64;
    double4_t acc[NTP];
#pragma unroll
    for (int tp = 0; tp < NTP; ++tp) acc[tp] = double4_t{0.0, 0.0, 0.0, 0.0};
    for (int ks = rs >> 2; ks <= (re - 1) >> 2; ++ks) {
      const int p = 4 * ks + lk;
      const bool inrun = p >= rs && p < re;
      double v[NT], vm[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        v[t] = s_row[p * LDR + 16 * t + lr];
        vm[t] = inrun ? v[t] : 0.0;
      }
      int tp = 0;
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj, ++tp) acc[tp] = __builtin_amdgcn_mfma_f64_16x16x4f64(vm[tj], v[ti], acc[tp], 0, 0, 0);
    }
    {
      int tp = 0;
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj, ++tp)
#pragma unroll
          for (int r = 0; r < 4; ++r) s_g[(16 * ti + lr) * LDR + 16 * tj + lk + 4 * r] = acc[tp][r];
    }
    __syncthreads();
    const int* pi = s_pi + rs * NP;
    int sl = 0;
#pragma unroll 1
    for (int ra = 0; ra < KK; ++ra)
#pragma unroll 1
      for (int rb = 0; rb <= ra; ++rb, ++sl) {
        double val = 0.0;
        bool on = false;
        if (l < 49) {
          on = ra > rb || eca >= ecb;   // (a diagonal pair's block is symmetric: its lower part is what is placed)
          val = s_g[(7 * ra + eca) * LDR + 7 * rb + ecb];
        } else if (l < 56 && ra == rb) {
          on = true;
          val = s_g[NR * LDR + 7 * rb + (l - 49)];
        }
        if (on) atomic_add_f64(pb + (size_t)pi[sl] * SLM_WREC + l, val);
      }
    __syncthreads();   // s_g is rewritten by the next run
  }
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
