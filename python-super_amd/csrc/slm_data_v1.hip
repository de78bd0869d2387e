// slm_data_v1.hip -- tuple-sorted data-term Jacobian pass (JtJ / jtl of the point-to-plane
// term, reference super/loss.py:222-288 + 200-205), no per-entry atomics on the matrix:
//
//   k_data_eval     the TARGET-side half of the per-surfel work, one thread per tuple-sorted position: skin ->
//                   project -> match -> 8 bilinear taps -> residual r and c = dr/dT(p); {r, c} go to the evaluation
//                   buffer (32 B per position), sum r^2 / the matched count to the loss partials.  It IS the loss
//                   pass of the LM loop (trial point beta + delta); when the step is accepted its output is what the
//                   next Jacobian pass needs, so the projection and the taps are done once per iteration, not twice.
//   k_data_gram     one wave per 64 tuple-sorted surfel positions: every lane forms the 28 Jacobian entries of
//                   its surfel from {r, c} and its four nodes (no target access), writes the augmented row [J(28) | r | 0 0 0]
//                   (node slots in ascending-id order) to LDS, and the wave contracts
//                   groups of 4 surfels on the f64 MFMA:  G += row^T row  (32 x 32 as the
//                   three 16x16 tiles 00, 10, 11).  G[28][0..27] = J^T r and G[28][28] =
//                   sum r^2 ride along for free.  One G per (tuple, chunk) run goes to
//                   the slab with plain coalesced stores (no atomics at all).
//   k_band_assemble one wave per coupled node pair (a >= b): sums the 7x7 sub-blocks of
//                   the runs that contain both nodes (inverted index built once per frame
//                   by slm_prep.hip) and stores them into the lower band; the diagonal
//                   pair (a,a) also sums row 28 of its runs into jtl.  Every band / jtl
//                   entry is written by exactly one lane: bitwise reproducible.
#include <cstdlib>

#include "slm_data.h"
#include "slm_begin.h"

#define ROW_STRIDE 17   // doubles per surfel half-row in LDS (odd: conflict-free 64-bit writes)

// Evaluation pass over the tuple-sorted positions [256 wg_lo, min(n_pos, 256 wg_hi)) of every slot (the whole frame unless
// it is sharded over several GPUs).  grid = (n_blocks, n_frames), grid-stride; partial sums per block in a fixed order
// (bitwise reproducible).  mode 0: trial point (node_pk_try), always runs -- the loss pass; 1: current beta, only slots whose
// buffer is not valid (first iteration of a frame; after a reject when records are not reused); 2: current beta, always
// (parity entry points).  The LAST block of a slot (ticket) publishes the matched count of the pass (m_eval) and, for the
// passes at the current beta, marks the buffer valid; for the trial pass k_accept decides (valid iff accepted).
__global__ void __launch_bounds__(256) k_data_eval(const FrameDev* __restrict__ frames, double lam, int mode,
                                                    const int* __restrict__ reuse) {
  __shared__ double sm[16];
  __shared__ int s_skip;
  if (mode == 1 && reuse && reuse[blockIdx.y]) return;   // the Jacobian pass that would read the buffer is skipped too
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || !fd.v1_ready || fd.st->stopped) return;
  if (mode == 1) {
    // (uniform per slot for the whole launch: only the last block to FINISH changes the flag)
    if (threadIdx.x == 0) s_skip = __hip_atomic_load(&fd.st->eval_valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1;
    __syncthreads();
    if (s_skip) return;
  }
  const double* npk = mode == 0 ? fd.node_pk_try.get() : fd.node_pk.get();
  const int p0 = 256 * fd.wg_lo, p1 = min(fd.n_pos, 256 * fd.wg_hi);
  double acc = 0.0;
  int cnt = 0;
  for (int pos = p0 + blockIdx.x * blockDim.x + threadIdx.x; pos < p1; pos += gridDim.x * blockDim.x) {
    const int4 ids = *reinterpret_cast<const int4*>(fd.s_idx + 4 * (size_t)pos);
    double2 o0 = make_double2(0.0, 0.0), o1 = make_double2(0.0, 0.0);
    if (ids.x >= 0) {
      double wk[4];
      ld_state4(fd.s_w, (size_t)pos, fd.f.state_f64, wk);
      const d3 pp = ld_state3(fd.s_pts, (size_t)pos, fd.f.state_f64);
      SurfelEval ev;
      eval_surfel_core<2, true>(fd, pp, ids, wk, lam, npk, ev);   // (target taps from the per-pixel table)
      if (ev.match) {
        acc += ev.r * ev.r;
        ++cnt;
        o0 = make_double2(ev.r, ev.c[0]);
        o1 = make_double2(ev.c[1], ev.c[2]);
      }
    }
    double2* out = reinterpret_cast<double2*>(fd.ev_rc.get() + 4 * (size_t)pos);
    out[0] = o0;
    out[1] = o1;
  }
  const double s = block_sum(acc, sm);
  const double c = block_sum((double)cnt, sm);
  if (threadIdx.x == 0) {
    fd.loss_part[2 * blockIdx.x] = s;
    fd.loss_part[2 * blockIdx.x + 1] = c;
    LMState* st = fd.st;
    const unsigned long long old = __hip_atomic_fetch_add(&st->eval_acc, (1ull << 32) | (unsigned long long)(unsigned)(int)c,
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)(old >> 32) == (int)gridDim.x - 1) {            // the last block of the slot to finish
      st->m_eval = (int)(unsigned)(old & 0xFFFFFFFFull) + (int)c;
      __hip_atomic_store(&st->eval_acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (mode != 0) __hip_atomic_store(&st->eval_valid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// grid = (ceil(max n_pos / 256), n_frames), 256 threads = 4 waves, one 64-position chunk each
// dbg (diagnostic build only): bit0 skip slab stores, bit1 skip MFMA, bit2 skip surfel evaluation
// The 32-entry rows pass through LDS in two halves of 16 entries (8.7 KB per wave instead of
// 17 KB: three to four workgroups per CU instead of two); the first half's MFMA operands wait
// in 16 registers per lane.
//
// MERGE (default): the workgroup's 4 waves add the 7x7 node-pair blocks of their runs (and the
// 7-entry J^T r pieces of the diagonal pairs) into LDS records keyed by node pair (ds_add_f64),
// and the workgroup writes one 56-double record per distinct pair -- about 4x fewer HBM bytes
// than one 768-double Gram per run, and the assemble kernels read contiguous records.
// BEGIN (round 6): the launch also carries the iteration's zeroing (iter_begin_nd_body, slm_begin.h) -- blocks
// [n_gram, n_gram + n_begin) of a slot: the Jacobian pass writes records, the zeroing writes fronts / vectors / flags,
// nothing in common; one launch boundary fewer per iteration and the zeroing's store stream (HBM-bound) runs under the
// Gram pass (VALU / LDS-bound).
template <bool MERGE, bool BEGIN = false>
__global__ void __launch_bounds__(256, 2) k_data_gram(const FrameDev* __restrict__ frames, double lam,
                                                       int dbg, const int* __restrict__ reuse, int n_gram = 0, int n_begin = 0,
                                                       int dag_cut = -2) {
  if (BEGIN && (int)blockIdx.x >= n_gram) {
    iter_begin_nd_body(frames[blockIdx.y], (int)blockIdx.x - n_gram, n_begin, reuse && reuse[blockIdx.y], dag_cut,
                       frames[blockIdx.y].v1_ready && frames[blockIdx.y].v2_ready);
    return;
  }
  __shared__ double rows[4][64 * ROW_STRIDE];
  __shared__ double recs[MERGE ? SLM_LB_MAX * SLM_WREC : 1];
  __shared__ uint8_t lidx[4][MERGE ? 160 : 4];   // per wave: record of each of the 10 node pairs of its 16 groups
  // After a REJECTED step beta is rolled back, so this pass would reproduce the records of the previous iteration
  // entry for entry (reference super/LM.py:114-117 then :96 rebuilds an identical JtJ): the slot keeps them and
  // k_front_assemble re-reads them.  (The flag sits in an array of its own, indexed by the slot: a load that does not
  // depend on the descriptor, so it costs the accepted iterations nothing.)
  if (reuse && reuse[blockIdx.y]) return;
  const FrameDev& fd = frames[blockIdx.y];
  // (no test of st->stopped here: a stopped slot only wastes this pass, and the test would put one more
  //  dependent load in front of everything)
  if (!fd.bound || !fd.v1_ready) return;
  if (MERGE != (fd.v2_ready != 0)) return;   // the host launches both variants when slots differ
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int base = (blockIdx.x * 4 + w) * 64;
  if (blockIdx.x * 256 >= fd.n_pos) return;
  // the matched count of this pass is the one of the evaluation it consumes (this rank's share of it when sharded)
  if (blockIdx.x == 0 && threadIdx.x == 0) fd.st->m_grad = fd.st->m_eval;
  if ((int)blockIdx.x < fd.wg_lo || (int)blockIdx.x >= fd.wg_hi) return;   // another rank's share
  const bool wact = base < fd.n_pos;
  const int pos = base + l;
  const int lc = l & 15, lq = l >> 4;
  // ---- stage 0: everything that only depends on the position is requested at once ----
  int rec0 = 0, nrec = 0;
  if (MERGE) {
    rec0 = fd.wg_first[blockIdx.x];
    nrec = fd.wg_last[blockIdx.x] - rec0 + 1;
  }
  int my_run = -1;
  int4 ids = {-1, -1, -1, -1};
  double wk[4] = {0, 0, 0, 0};
  d3 pp = {0, 0, 0};
  double2 e0 = make_double2(0.0, 0.0), e1 = make_double2(0.0, 0.0);   // {r, c.x}, {c.y, c.z} of the position
  if (wact) {
    my_run = fd.grp_run[(base >> 2) + lc];   // lane (l & 15) holds the run of group (l & 15)
    const double2* e2 = reinterpret_cast<const double2*>(fd.ev_rc.get() + 4 * (size_t)pos);
    e0 = e2[0];
    e1 = e2[1];
    ids = *reinterpret_cast<const int4*>(fd.s_idx + 4 * (size_t)pos);
    ld_state4(fd.s_w, (size_t)pos, fd.f.state_f64, wk);
    pp = ld_state3(fd.s_pts, (size_t)pos, fd.f.state_f64);
  }
  // ---- stage 1: record numbers of the wave's runs (needs my_run), in flight with the node gathers ----
  uint8_t lv0 = 0, lv1 = 0, lv2 = 0;
  if (MERGE && wact) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int idx = l + 64 * k;                      // entry idx = group * 10 + pair slot, 160 per wave
      const int g = idx / 10, e = idx - 10 * g;
      const int run = __shfl(my_run, g < 16 ? g : 0, 64);
      const uint8_t v = (idx < 160 && run >= 0) ? fd.run_lidx[10 * (size_t)run + e] : (uint8_t)0;
      if (k == 0) lv0 = v; else if (k == 1) lv1 = v; else lv2 = v;
    }
  }
  if (MERGE)
    for (int i = threadIdx.x; i < nrec * SLM_WREC; i += 256) recs[i] = 0.0;
  double* myrow = &rows[w][l * ROW_STRIDE];

  // ---- the surfel's row from the evaluation buffer: no projection, no target access --------------
  SurfelEval ev;
  ev.match = false;
  ev.r = 0.0;
  ev.id[0] = ev.id[1] = ev.id[2] = ev.id[3] = -1;
  const bool live = wact && ids.x >= 0;
#ifdef SLM_STAMPS
  if (live && !(dbg & 4))
#else
  if (live)
#endif
  {
    ev.id[0] = ids.x; ev.id[1] = ids.y; ev.id[2] = ids.z; ev.id[3] = ids.w;
    // (an unmatched surfel has {r, c} = 0: its row is zero and it adds nothing to the Gram)
    ev.match = e0.x != 0.0 || e0.y != 0.0 || e1.x != 0.0 || e1.y != 0.0;
    ev.r = e0.x;
    if (ev.match) rows_from_c(pp, ev.id, wk, lam, fd.node_pk, {e0.y, e1.x, e1.y}, ev.row);
  }
  if (MERGE && wact) {
    lidx[w][l] = lv0;
    lidx[w][l + 64] = lv1;
    if (l + 128 < 160) lidx[w][l + 128] = lv2;
  }
  if (MERGE) __syncthreads();   // accumulators zeroed, record numbers in place
  if (wact) {

  // canonical slot of neighbour k = number of neighbour ids smaller than id[k]
  int slot[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    slot[k] = 0;
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) slot[k] += (ev.id[k2] < ev.id[k]) ? 1 : 0;
  }
  double a0[16];   // first-half operands of the 16 groups: rows[4g + lq][lc]
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int e = 0; e < 16; ++e) myrow[e] = 0.0;
    if (ev.match) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          const int e = 7 * slot[k] + c - 16 * half;
          if (e >= 0 && e < 16) myrow[e] = ev.row[7 * k + c];
        }
      if (half == 1) myrow[28 - 16] = ev.r;
    }
    // the operands below are read by other lanes of the same wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (half == 0) {
#pragma unroll
      for (int g = 0; g < 16; ++g) a0[g] = rows[w][(4 * g + lq) * ROW_STRIDE + lc];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }

  // ---- Gram accumulation per run ---------------------------------------------------
  double4_t g00 = {0, 0, 0, 0}, g10 = {0, 0, 0, 0}, g11 = {0, 0, 0, 0};
  int cur = -1;

  auto flush = [&](int run, int grp) {
#ifdef SLM_STAMPS
    if (dbg & 1) return;
#endif
    if (MERGE) {
      // accumulator element (tile, r) of lane l is G[i][j], i = 4r + lq (+16), j = lc (+16)
      const uint8_t* lvp = &lidx[w][10 * grp];   // record of each of the 10 node pairs of this run
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 4 * r + lq + (t >= 1 ? 16 : 0), j = lc + (t == 2 ? 16 : 0);
          const double v = t == 0 ? g00[r] : (t == 1 ? g10[r] : g11[r]);
          const int pa = (i * 37) >> 8, pb = (j * 37) >> 8;   // i / 7 for i < 32
          const int ca = i - 7 * pa, cb = j - 7 * pb;
          const bool jrow = (i == 28);
          const bool act = j < 28 && (jrow || (i < 28 && i >= j));
          // row 28 (J^T r) goes to the diagonal record of node slot pb, entries 49..55
          int ps = jrow ? (pb * (pb + 1) / 2 + pb) : (pa * (pa + 1) / 2 + pb);
          ps = act ? ps : 0;
          const int rec = lvp[ps];
          if (act) unsafeAtomicAdd(&recs[rec * SLM_WREC + (jrow ? 49 + cb : 7 * ca + cb)], v);
        }
    } else {
      double* out = fd.slab + (size_t)run * SLM_SLAB_STRIDE;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        out[r * 64 + l] = g00[r];
        out[256 + r * 64 + l] = g10[r];
        out[512 + r * 64 + l] = g11[r];
      }
      // row 28 of G (= J^T r, sum r^2) travels in the slab; k_*_assemble turns it into jtl
    }
  };

#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int run = __builtin_amdgcn_readlane(my_run, g);
    if (run != cur) {
      if (cur >= 0) flush(cur, g - 1);
      g00 = double4_t{0, 0, 0, 0};
      g10 = double4_t{0, 0, 0, 0};
      g11 = double4_t{0, 0, 0, 0};
      cur = run;
    }
    if (run < 0) continue;
    const double a1 = rows[w][(4 * g + lq) * ROW_STRIDE + lc];
#ifdef SLM_STAMPS
    if (dbg & 2) { g00[0] += a0[g]; g11[0] += a1; continue; }
#endif
    g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[g], a0[g], g00, 0, 0, 0);
    g10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a0[g], g10, 0, 0, 0);
    g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, g11, 0, 0, 0);
  }
  if (cur >= 0) flush(cur, 15);
  }   // wave has positions
  if (MERGE) {
    __syncthreads();
    double* out = fd.wgslab + (size_t)rec0 * SLM_WREC;
    for (int i = threadIdx.x; i < nrec * SLM_WREC; i += 256) out[i] = recs[i];
  }
}

// G[i][j], i >= j, from a slab entry (tiles 00, 10, 11; 16x16 row-major each)
__device__ __forceinline__ double gram_at(const double* g, int i, int j) {
  if (i < 16) return g[16 * i + j];
  if (j < 16) return g[256 + 16 * (i - 16) + j];
  return g[512 + 16 * (i - 16) + (j - 16)];
}

// grid = (ceil(max n_blocks / 4), n_frames), 256 threads: one wave per node pair
__global__ void __launch_bounds__(256) k_band_assemble(const FrameDev* __restrict__ frames) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || !fd.v1_ready || fd.st->stopped) return;
  const int bi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bi >= fd.n_blocks) return;
  const int l = threadIdx.x & 63;
  const unsigned key = (unsigned)fd.blk_key[bi];
  const int a = (int)(key / (unsigned)fd.f.J), b = (int)(key % (unsigned)fd.f.J);
  const int ca = l / 7, cb = l % 7;
  const bool act = (l < 49) && (a != b || ca >= cb);   // lower triangle only on diagonal blocks
  const bool jt = (a == b) && l >= 49 && l < 56;       // diagonal block: lanes 49..55 build jtl of node a
  double acc = 0.0;
  if (fd.v2_ready) {
    const int s0 = fd.blk2_start[bi], s1 = fd.blk2_start[bi + 1];
    if (l < SLM_WREC)
      for (int s = s0; s < s1; ++s) acc += fd.wgslab[(size_t)fd.blk2_entry[s] * SLM_WREC + l];
  } else {
    const int s0 = fd.blk_start[bi], s1 = fd.blk_start[bi + 1];
    for (int s = s0; s < s1; ++s) {
      const int pl = fd.blk_entry[s];
      const int run = pl >> 4, pa = (pl >> 2) & 3, pb = pl & 3;
      const double* G = fd.slab + (size_t)run * SLM_SLAB_STRIDE;
      if (act) acc += gram_at(G, 7 * pa + ca, 7 * pb + cb);
      else if (jt) acc += gram_at(G, 28, 7 * pa + (l - 49));     // (J^T r) of node a in this run
    }
  }
  if (act) *band_entry(fd, 7 * a + ca, 7 * b + cb) = acc;
  else if (jt) fd.rhs[7 * a + (l - 49)] = -acc;               // jtl = -J^T r, one writer per entry
}

// variants: bit0 = some slot uses the workgroup-merged records, bit1 = some slot uses the per-run slab
// The Jacobian pass AND the iteration's zeroing in one launch (every slot of the batch on the workgroup-merged records):
// what slm_run's loop enqueues instead of k_iter_begin_nd + k_data_gram.
void launch_begin_and_gram(const FrameDev* frames_dev, int n_frames, int max_pos, double lam, hipStream_t st, const int* reuse,
                           int dag_cut) {
  int dbg = 0;
#ifdef SLM_STAMPS
  if (const char* e = getenv("SLM_DBG")) dbg = atoi(e);
#endif
  const int n_gram = (max_pos + 255) / 256, n_begin = 1024;
  hipLaunchKernelGGL((k_data_gram<true, true>), dim3(n_gram + n_begin, n_frames), dim3(256), 0, st, frames_dev, lam, dbg, reuse, n_gram,
                     n_begin, dag_cut);
}

void launch_data_gram(const FrameDev* frames_dev, int n_frames, int max_pos, double lam, int variants,
                      hipStream_t st, const int* reuse) {
  if (max_pos <= 0) return;
  int dbg = 0;
#ifdef SLM_STAMPS
  if (const char* e = getenv("SLM_DBG")) dbg = atoi(e);
#endif
  const dim3 grid((max_pos + 255) / 256, n_frames);
  if (variants & 1) hipLaunchKernelGGL(k_data_gram<true>, grid, dim3(256), 0, st, frames_dev, lam, dbg, reuse);
  if (variants & 2) hipLaunchKernelGGL(k_data_gram<false>, grid, dim3(256), 0, st, frames_dev, lam, dbg, reuse);
}

void launch_data_eval(const FrameDev* frames_dev, int n_frames, int n_blocks, double lam, int mode, hipStream_t st,
                      const int* reuse) {
  hipLaunchKernelGGL(k_data_eval, dim3(n_blocks, n_frames), dim3(256), 0, st, frames_dev, lam, mode, reuse);
}

void launch_band_assemble(const FrameDev* frames_dev, int n_frames, int max_blocks, hipStream_t st) {
  if (max_blocks <= 0) return;
  hipLaunchKernelGGL(k_band_assemble, dim3((max_blocks + 3) / 4, n_frames), dim3(256), 0, st, frames_dev);
}
