// slm_prep.hip -- once-per-frame preparation of the tuple-sorted data-term assembly
// (the analogue of DataLoss.prepare, reference super/loss.py:212-220, which caches
// per-frame gathers; here the per-frame cache is an ordering + index, not 137 MB of f64).
//
// The KNN tables are fixed during the LM iterations of a frame, so all scatter structure
// is resolved once, on the device:
//   1. canonical key of every surfel's KNN 4-tuple (ascending node ids, 4 x 16 bit)
//   2. radix sort (rocPRIM) -> surfels grouped by tuple; run-length encode -> tuples
//   3. every tuple's segment is padded to a multiple of 4 positions (one MFMA k-group never
//      mixes tuples); positions are cut into 64-wide chunks (one wave each); a
//      (tuple, chunk) pair is a "run" = one Gram-matrix slab entry
//   4. surfel xyz / ids / weights are copied into position order (coalesced streaming)
//   5. inverted index: for every coupled node pair (a >= b) the list of (run, slot pair)
//      contributions, by a second sort + run-length encode
// rocPRIM is used only for these once-per-frame sorts/scans (plumbing); every kernel on
// the per-iteration path is hand-written.
#include <algorithm>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "slm_common.h"
#include "slm_prep.h"

namespace {

__device__ __forceinline__ void sort4(int& a, int& b, int& c, int& d) {
  int t;
#define CSWAP(x, y) if (x > y) { t = x; x = y; y = t; }
  CSWAP(a, b) CSWAP(c, d) CSWAP(a, c) CSWAP(b, d) CSWAP(b, c)
#undef CSWAP
}

// (a node index outside [0, J) -- the reference would raise an IndexError -- is reported through *bad and clamped, so
//  that nothing downstream reads out of bounds before the host has seen the flag)
__global__ void __launch_bounds__(256) k_tuple_keys(int N, int J, const int* __restrict__ knn,
                                                     unsigned long long* __restrict__ keys,
                                                     int* __restrict__ ids, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int4 v = *reinterpret_cast<const int4*>(knn + 4 * i);
  int a = v.x, b = v.y, c = v.z, d = v.w;
  if ((unsigned)a >= (unsigned)J || (unsigned)b >= (unsigned)J || (unsigned)c >= (unsigned)J || (unsigned)d >= (unsigned)J) {
    *bad = 1;
    a = min(max(a, 0), J - 1); b = min(max(b, 0), J - 1); c = min(max(c, 0), J - 1); d = min(max(d, 0), J - 1);
  }
  sort4(a, b, c, d);
  keys[i] = ((unsigned long long)a << 48) | ((unsigned long long)b << 32) |
            ((unsigned long long)c << 16) | (unsigned long long)d;
  ids[i] = i;
}

// the same range test on its own, for the frames that do not take the tuple-sorted path (data_path 1, J >= 65536)
__global__ void __launch_bounds__(256) k_check_knn(int N, int J, const int* __restrict__ knn, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int4 v = *reinterpret_cast<const int4*>(knn + 4 * i);
  if ((unsigned)v.x >= (unsigned)J || (unsigned)v.y >= (unsigned)J || (unsigned)v.z >= (unsigned)J || (unsigned)v.w >= (unsigned)J) *bad = 1;
}

__global__ void __launch_bounds__(256) k_padded_counts(const int* __restrict__ d_nt,
                                                        const int* __restrict__ tcount,
                                                        int* __restrict__ pc) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < *d_nt) pc[t] = (tcount[t] + 3) & ~3;
}

__global__ void __launch_bounds__(256) k_run_counts(const int* __restrict__ d_nt,
                                                     const int* __restrict__ pstart,
                                                     const int* __restrict__ pc, int* __restrict__ nruns) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < *d_nt) {
    const int ps = pstart[t], pe = ps + pc[t];
    nruns[t] = (pe - 1) / 64 - ps / 64 + 1;
  }
}

// A hinted preparation sizes its buffers and grids for `bound` tuples before the true count is known on the host: the
// count every later kernel works with is cut down to the bound (a frame with more tuples is then prepared on a
// truncated tuple list, inside its buffers, and prepared again by the host with the exact count); scal[12] keeps the
// true count for that decision.
__global__ void k_clamp_tuples(int* __restrict__ scal, int bound) {
  const int nt = scal[0];
  scal[12] = nt;
  if (nt > bound) scal[0] = bound;
}

// scal[0] = n_tuples (already there), scal[1] = total padded positions, scal[2] = total runs
__global__ void k_totals(int* __restrict__ scal, const int* __restrict__ pstart,
                         const int* __restrict__ pc, const int* __restrict__ rstart,
                         const int* __restrict__ nruns) {
  const int nt = scal[0];
  scal[1] = nt > 0 ? pstart[nt - 1] + pc[nt - 1] : 0;
  scal[2] = nt > 0 ? rstart[nt - 1] + nruns[nt - 1] : 0;
}

__global__ void __launch_bounds__(256) k_fill_sorted(
    int n_pos_bound, const int* __restrict__ scal, const slm_frame f,
    const unsigned long long* __restrict__ tkeys, const int* __restrict__ tcount,
    const int* __restrict__ tstart, const int* __restrict__ pstart, const int* __restrict__ rstart,
    const int* __restrict__ sids, void* __restrict__ s_pts, int* __restrict__ s_idx,
    void* __restrict__ s_w, int* __restrict__ grp_run, int* __restrict__ run_nodes,
    int* __restrict__ run_chunk) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= n_pos_bound) return;
  const int nt = scal[0], ptot = scal[1];
  int4 idv = make_int4(-1, -1, -1, -1);
  double wv[4] = {0.0, 0.0, 0.0, 0.0};
  d3 pp = {0.0, 0.0, 0.0};
  int run = -1;
  if (pos < ptot) {
    // tuple owning this position: last t with pstart[t] <= pos
    int lo = 0, hi = nt - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (pstart[mid] <= pos) lo = mid; else hi = mid - 1;
    }
    const int t = lo, ps = pstart[t], e = pos - ps;
    run = rstart[t] + (pos / 64 - ps / 64);
    if (e < tcount[t]) {
      const int i = sids[tstart[t] + e];
      idv = *reinterpret_cast<const int4*>(f.sf_knn_idx + 4 * i);
      ld_state4(f.sf_knn_w, (size_t)i, f.state_f64, wv);
      pp = ld_state3(f.sf_points, (size_t)i, f.state_f64);
    }
    if (e == 0 || (pos & 63) == 0) {
      const unsigned long long k = tkeys[t];
      int4 nodes = make_int4((int)(k >> 48) & 0xFFFF, (int)(k >> 32) & 0xFFFF, (int)(k >> 16) & 0xFFFF,
                             (int)k & 0xFFFF);
      *reinterpret_cast<int4*>(run_nodes + 4 * run) = nodes;
      run_chunk[run] = pos >> 6;
    }
  }
  *reinterpret_cast<int4*>(s_idx + 4 * pos) = idv;
  // the sorted copies keep the dtype of the state (values are copied, never rounded)
  if (f.state_f64) {
    double* sw = static_cast<double*>(s_w) + 4 * (size_t)pos;
    double* sp = static_cast<double*>(s_pts) + 3 * (size_t)pos;
    *reinterpret_cast<double2*>(sw) = make_double2(wv[0], wv[1]);
    *reinterpret_cast<double2*>(sw + 2) = make_double2(wv[2], wv[3]);
    sp[0] = pp.x; sp[1] = pp.y; sp[2] = pp.z;
  } else {
    *reinterpret_cast<float4*>(static_cast<float*>(s_w) + 4 * (size_t)pos) =
        make_float4((float)wv[0], (float)wv[1], (float)wv[2], (float)wv[3]);
    float* sp = static_cast<float*>(s_pts) + 3 * (size_t)pos;
    sp[0] = (float)pp.x; sp[1] = (float)pp.y; sp[2] = (float)pp.z;
  }
  if ((pos & 3) == 0) grp_run[pos >> 2] = run;
}

// 10 (a >= b) node pairs per run: key = a*J + b, payload = run*16 + pa*4 + pb
__global__ void __launch_bounds__(256) k_pairs(int n_runs_bound, const int* __restrict__ scal, int J,
                                                const int* __restrict__ run_nodes,
                                                unsigned* __restrict__ pkeys, int* __restrict__ pvals) {
  const int rr = blockIdx.x * blockDim.x + threadIdx.x;
  if (rr >= n_runs_bound) return;
  const bool live = rr < scal[2];
  int n[4] = {0, 0, 0, 0};
  if (live) {
    const int4 v = *reinterpret_cast<const int4*>(run_nodes + 4 * rr);
    n[0] = v.x; n[1] = v.y; n[2] = v.z; n[3] = v.w;
  }
  int e = 0;
#pragma unroll
  for (int pa = 0; pa < 4; ++pa)
#pragma unroll
    for (int pb = 0; pb <= pa; ++pb, ++e) {
      pkeys[10 * rr + e] = live ? (unsigned)(n[pa] * J + n[pb]) : 0xFFFFFFFFu;
      pvals[10 * rr + e] = rr * 16 + pa * 4 + pb;
    }
}

// scal[3] = n_blocks (unique live keys); blk_start[n_blocks] = end of the last live block
__global__ void k_totals2(int* __restrict__ scal, const unsigned* __restrict__ ukeys,
                          int* __restrict__ blk_start, int n_entries) {
  int nb = scal[4];   // unique keys incl. a possible trailing 0xFFFFFFFF run
  if (nb > 0 && ukeys[nb - 1] == 0xFFFFFFFFu) {
    nb -= 1;          // blk_start[nb] already holds the start of the invalid run
  } else {
    blk_start[nb] = n_entries;
  }
  scal[3] = nb;
}

// ---- v2: (workgroup, pair) records ------------------------------------------------------
// key = (workgroup << 32) | (a*J + b), payload = run*16 + pa*4 + pb; workgroup = chunk / 4
__global__ void __launch_bounds__(256) k_pairs2(int n_runs_bound, const int* __restrict__ scal, int J,
                                                 const int* __restrict__ run_nodes,
                                                 const int* __restrict__ run_chunk,
                                                 unsigned long long* __restrict__ keys,
                                                 int* __restrict__ vals) {
  const int rr = blockIdx.x * blockDim.x + threadIdx.x;
  if (rr >= n_runs_bound) return;
  const bool live = rr < scal[2];
  int n[4] = {0, 0, 0, 0};
  unsigned long long wg = 0;
  if (live) {
    const int4 v = *reinterpret_cast<const int4*>(run_nodes + 4 * rr);
    n[0] = v.x; n[1] = v.y; n[2] = v.z; n[3] = v.w;
    wg = (unsigned long long)(run_chunk[rr] >> 2);
  }
  int e = 0;
#pragma unroll
  for (int pa = 0; pa < 4; ++pa)
#pragma unroll
    for (int pb = 0; pb <= pa; ++pb, ++e) {
      keys[10 * rr + e] = live ? ((wg << 32) | (unsigned)(n[pa] * J + n[pb])) : ~0ull;
      vals[10 * rr + e] = rr * 16 + pa * 4 + pb;
    }
}

// per unique (workgroup, pair) record u: first/last record of its workgroup, max records per
// workgroup, and the pair key / record id for the final pair -> records index
__global__ void __launch_bounds__(256) k_wg_bounds(int nwb_bound, int* __restrict__ scal,
                                                    const unsigned long long* __restrict__ wkeys,
                                                    int* __restrict__ wg_first, int* __restrict__ wg_last,
                                                    unsigned* __restrict__ pk, int* __restrict__ pv) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= nwb_bound) return;
  const int nu = scal[5];
  const bool live = u < nu && wkeys[u] != ~0ull;
  pk[u] = live ? (unsigned)(wkeys[u] & 0xFFFFFFFFull) : 0xFFFFFFFFu;
  pv[u] = u;
  if (!live) return;
  const int wg = (int)(wkeys[u] >> 32);
  if (u == 0 || (int)(wkeys[u - 1] >> 32) != wg) wg_first[wg] = u;
  const bool last = (u + 1 >= nu) || wkeys[u + 1] == ~0ull || (int)(wkeys[u + 1] >> 32) != wg;
  if (last) wg_last[wg] = u;
  atomicAdd(&scal[6], 1);   // live records
}

__global__ void __launch_bounds__(256) k_wg_max(int n_wg, const int* __restrict__ wg_first,
                                                 const int* __restrict__ wg_last, int* __restrict__ scal) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_wg) return;
  const int c = wg_last[g] - wg_first[g] + 1;
  if (c > 0) atomicMax(&scal[7], c);
}

// local record index of every (run, pair slot)
__global__ void __launch_bounds__(256) k_run_lidx(int nwb_bound, const int* __restrict__ scal,
                                                   const unsigned long long* __restrict__ wkeys,
                                                   const int* __restrict__ wstart, const int* __restrict__ wcount,
                                                   const int* __restrict__ svals, const int* __restrict__ wg_first,
                                                   uint8_t* __restrict__ run_lidx) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= nwb_bound || u >= scal[5] || wkeys[u] == ~0ull) return;
  const int wg = (int)(wkeys[u] >> 32);
  const int lidx = u - wg_first[wg];
  for (int e = wstart[u]; e < wstart[u] + wcount[u]; ++e) {
    const int pl = svals[e];
    const int run = pl >> 4, pa = (pl >> 2) & 3, pb = pl & 3;
    run_lidx[10 * run + pa * (pa + 1) / 2 + pb] = (uint8_t)(lidx < 255 ? lidx : 255);
  }
}

// Hash of the coupling graph on the DEVICE, so that a bind whose graph is the one the slot's symbolic plan was built
// for needs no read-back of the lists at all: out[0] = hash of (J, K_ED, node KNN table), out[1] = the same continued
// over the coupled-pair keys.  h = sum_i mix(word_i, i) mod 2^64 -- position dependent, and a sum, so the order in
// which the threads add does not matter (integer atomics: bitwise reproducible).
__device__ __forceinline__ unsigned long long plan_mix(unsigned long long w, unsigned long long i) {
  unsigned long long z = (w ^ ((i + 1ull) * 0x9E3779B97F4A7C15ull)) + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) k_plan_hash(int J, int K_ED, const int32_t* __restrict__ ed_knn,
                                                    const int32_t* __restrict__ blk_key, const int* __restrict__ scal,
                                                    unsigned long long* __restrict__ out) {
  const int n_knn = J * K_ED, n_pairs = scal[3];
  unsigned long long h0 = 0, h1 = 0;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
  if (tid == 0) h0 = plan_mix((unsigned long long)(unsigned)J << 32 | (unsigned)K_ED, ~0ull);
  for (int i = tid; i < n_knn; i += nthr) h0 += plan_mix((unsigned)ed_knn[i], (unsigned long long)i);
  for (int i = tid; i < n_pairs; i += nthr) h1 += plan_mix((unsigned)blk_key[i], (1ull << 40) + (unsigned long long)i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    h0 += __shfl_down(h0, o, 64);
    h1 += __shfl_down(h1, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out, h0);
    atomicAdd(out + 1, h0 + h1);
  }
}

// blk2_start[n_blocks] = number of live records (end of the last live pair)
__global__ void k_totals3(const int* __restrict__ scal, int* __restrict__ blk2_start) {
  blk2_start[scal[3]] = scal[6];
}

template <typename T>
hipError_t grow_raw(T*& p, size_t& cap, size_t need) {
  if (need <= cap) return hipSuccess;
  if (p) {
    hipError_t e = hipFree(p);
    if (e != hipSuccess) return e;
    p = nullptr;
    cap = 0;
  }
  const size_t want = need + need / 8 + 64;
  hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
  if (e == hipSuccess) cap = want;
  return e;
}

}  // namespace

struct PrepBuffers {
  // phase A
  unsigned long long *keys = nullptr, *skeys = nullptr, *tkeys = nullptr;
  int *ids = nullptr, *sids = nullptr, *tcount = nullptr;
  size_t cap_n = 0;
  // phase B
  int *tstart = nullptr, *pc = nullptr, *pstart = nullptr, *nruns = nullptr, *rstart = nullptr;
  size_t cap_t = 0;
  unsigned *pkeys = nullptr, *spkeys = nullptr, *ukeys = nullptr;
  int *pvals = nullptr, *bcount = nullptr;
  size_t cap_e = 0;
  unsigned long long *wkeys = nullptr, *swkeys = nullptr, *uwkeys = nullptr;
  int *wvals = nullptr, *swvals = nullptr, *wcount = nullptr, *wstart = nullptr, *pv2 = nullptr, *spv2 = nullptr,
      *b2count = nullptr;
  unsigned *pk2 = nullptr, *spk2 = nullptr, *upk2 = nullptr;
  size_t cap_w = 0;
  void* tmp = nullptr;
  size_t cap_tmp = 0;
  size_t q_n = 0, q_t = 0, q_e = 0;   // sizes the rocPRIM temporary-storage requirement was last queried for
  int* scal = nullptr;        // device: nt, ptot, nruns, nblocks, nunique, ..., [8..11] the two 64-bit graph hashes, [12] unclamped nt, [13] bad surfel KNN index seen
  int* scal_host = nullptr;   // pinned mirror
};

PrepBuffers* prep_create() {
  PrepBuffers* p = new PrepBuffers();
  if (hipMalloc((void**)&p->scal, 16 * sizeof(int)) != hipSuccess ||
      hipHostMalloc((void**)&p->scal_host, 16 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
    prep_destroy(p);
    return nullptr;
  }
  return p;
}

void prep_destroy(PrepBuffers* p) {
  if (!p) return;
  void* ptrs[] = {p->keys, p->skeys, p->tkeys, p->ids, p->sids, p->tcount, p->tstart, p->pc, p->pstart,
                  p->nruns, p->rstart, p->pkeys, p->spkeys, p->ukeys, p->pvals, p->bcount, p->tmp, p->scal,
                  p->wkeys, p->swkeys, p->uwkeys, p->wvals, p->swvals, p->wcount, p->wstart, p->pv2, p->spv2,
                  p->b2count, p->pk2, p->spk2, p->upk2};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  if (p->scal_host) (void)hipHostFree(p->scal_host);
  delete p;
}

#define PCHK(expr)                      \
  do {                                  \
    hipError_t e_ = (expr);             \
    if (e_ != hipSuccess) return e_;    \
  } while (0)

static hipError_t ensure_tmp(PrepBuffers* p, size_t bytes) {
  size_t cap = p->cap_tmp;
  char* q = (char*)p->tmp;
  hipError_t e = grow_raw(q, cap, bytes);
  p->tmp = q;
  p->cap_tmp = cap;
  return e;
}

// rocPRIM's size queries cost tens of microseconds of host time each (device-property look-ups) and
// slm_bind_frame is host-bound, so the requirement is queried once per buffer capacity -- for N, the
// tuple capacity and the entry capacity, which only change when a buffer grows -- and every call then
// gets the whole scratch buffer (rocPRIM accepts more than it needs and reports too little as an error).
static hipError_t ensure_tmp_for(PrepBuffers* p, size_t N, size_t cap_t, size_t cap_e, hipStream_t st) {
  if (p->q_n == N && p->q_t == cap_t && p->q_e == cap_e && p->tmp) return hipSuccess;
  size_t need = 0, b = 0;
  auto upd = [&](hipError_t e) {
    need = b > need ? b : need;
    b = 0;
    return e;
  };
  hipError_t e = hipSuccess;
  unsigned long long* k64 = nullptr;
  unsigned* k32 = nullptr;
  int* v = nullptr;
  if ((e = upd(rocprim::radix_sort_pairs(nullptr, b, k64, k64, v, v, N, 0, 64, st))) != hipSuccess) return e;
  if ((e = upd(rocprim::run_length_encode(nullptr, b, k64, N, k64, v, v, st))) != hipSuccess) return e;
  if ((e = upd(rocprim::exclusive_scan(nullptr, b, v, v, 0, cap_t, rocprim::plus<int>(), st))) != hipSuccess) return e;
  if ((e = upd(rocprim::radix_sort_pairs(nullptr, b, k32, k32, v, v, cap_e, 0, 32, st))) != hipSuccess) return e;
  if ((e = upd(rocprim::run_length_encode(nullptr, b, k32, cap_e, k32, v, v, st))) != hipSuccess) return e;
  if ((e = upd(rocprim::exclusive_scan(nullptr, b, v, v, 0, cap_e, rocprim::plus<int>(), st))) != hipSuccess) return e;
  if ((e = upd(rocprim::radix_sort_pairs(nullptr, b, k64, k64, v, v, cap_e, 0, 64, st))) != hipSuccess) return e;
  if ((e = upd(rocprim::run_length_encode(nullptr, b, k64, cap_e, k64, v, v, st))) != hipSuccess) return e;
  e = ensure_tmp(p, need + need / 4 + (1u << 20));
  if (e != hipSuccess) return e;
  p->q_n = N;
  p->q_t = cap_t;
  p->q_e = cap_e;
  return hipSuccess;
}

hipError_t prep_check_knn(PrepBuffers* p, const slm_frame& f, bool* bad, hipStream_t st) {
  *bad = false;
  if (f.N <= 0) return hipSuccess;
  PCHK(hipMemsetAsync(p->scal + 13, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_check_knn, dim3((f.N + 255) / 256), dim3(256), 0, st, f.N, f.J, f.sf_knn_idx, p->scal + 13);
  PCHK(hipMemcpyAsync(p->scal_host + 13, p->scal + 13, sizeof(int), hipMemcpyDeviceToHost, st));
  PCHK(hipStreamSynchronize(st));
  *bad = p->scal_host[13] != 0;
  return hipSuccess;
}

void plan_free(V1Plan& plan) {
  void* ptrs[] = {plan.s_pts, plan.s_idx, plan.s_w, plan.grp_run, plan.run_nodes, plan.slab, plan.blk_key,
                  plan.blk_start, plan.blk_entry, plan.run_chunk, plan.wg_first, plan.wg_last, plan.run_lidx,
                  plan.wgslab, plan.blk2_start, plan.blk2_entry};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  plan = V1Plan();
}

hipError_t prep_v1(PrepBuffers* p, const slm_frame& f, V1Plan& plan, V1Sizes* out, hipStream_t st) {
  const size_t N = (size_t)f.N;
  out->n_tuples = out->n_pos = out->n_runs = out->n_blocks = 0;
  out->n_wblk = out->max_wblk_per_wg = 0;
  if (N == 0) return hipSuccess;
  // ---- phase A: tuples ----------------------------------------------------------
  if (N > p->cap_n) {
    size_t c;
    c = p->cap_n; PCHK(grow_raw(p->keys, c, N));
    c = p->cap_n; PCHK(grow_raw(p->skeys, c, N));
    c = p->cap_n; PCHK(grow_raw(p->tkeys, c, N));
    c = p->cap_n; PCHK(grow_raw(p->ids, c, N));
    c = p->cap_n; PCHK(grow_raw(p->sids, c, N));
    c = p->cap_n; PCHK(grow_raw(p->tcount, c, N));
    p->cap_n = c;
  }
  PCHK(hipMemsetAsync(p->scal + 13, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_tuple_keys, dim3((N + 255) / 256), dim3(256), 0, st, (int)N, f.J, f.sf_knn_idx,
                     p->keys, p->ids, p->scal + 13);
  PCHK(ensure_tmp_for(p, N, p->cap_t ? p->cap_t : 1, p->cap_e ? p->cap_e : 1, st));
  size_t b1 = p->cap_tmp, b2 = p->cap_tmp;
  PCHK(rocprim::radix_sort_pairs(p->tmp, b1, p->keys, p->skeys, p->ids, p->sids, N, 0, 64, st));
  PCHK(rocprim::run_length_encode(p->tmp, b2, p->skeys, N, p->tkeys, p->tcount, p->scal, st));
  // The tuple count sizes the buffers and grids of everything below.  A plan that has been built before carries the
  // count of its last frame: with 12 % + 64 head-room on that hint as the BOUND nothing has to be read back here (the
  // kernels take the count from the device, cut down to the bound by k_clamp_tuples; the bound sizes buffers, grids and scans) -- the one read-back at the end
  // says whether the bound held; if not (the scene changed abruptly) the preparation runs again with the exact count.
  // A host <-> device round trip costs 0.1 ms when all is well and was seen to take 3-7 ms now and then (stall_hunt.py).
  size_t nt;
  const bool hinted = plan.nt_hint > 0;
  if (hinted) {
    nt = (size_t)plan.nt_hint + (size_t)plan.nt_hint / 8 + 64;
    if (nt > N) nt = N;
    hipLaunchKernelGGL(k_clamp_tuples, dim3(1), dim3(1), 0, st, p->scal, (int)nt);
  } else {
    PCHK(hipMemcpyAsync(p->scal_host, p->scal, sizeof(int), hipMemcpyDeviceToHost, st));
    PCHK(hipStreamSynchronize(st));
    nt = (size_t)p->scal_host[0];
    if (nt == 0) return hipSuccess;
  }

  // ---- phase B: layout, sorted copies, inverted index ------------------------------
  const size_t pos_bound = (N + 3 * nt + 63) / 64 * 64;
  const size_t runs_bound = nt + pos_bound / 64 + 1;
  const size_t n_entries = 10 * runs_bound;
  if (nt > p->cap_t) {
    size_t c;
    c = p->cap_t; PCHK(grow_raw(p->tstart, c, nt));
    c = p->cap_t; PCHK(grow_raw(p->pc, c, nt));
    c = p->cap_t; PCHK(grow_raw(p->pstart, c, nt));
    c = p->cap_t; PCHK(grow_raw(p->nruns, c, nt));
    c = p->cap_t; PCHK(grow_raw(p->rstart, c, nt));
    p->cap_t = c;
  }
  if (n_entries > p->cap_e) {
    size_t c;
    c = p->cap_e; PCHK(grow_raw(p->pkeys, c, n_entries));
    c = p->cap_e; PCHK(grow_raw(p->spkeys, c, n_entries));
    c = p->cap_e; PCHK(grow_raw(p->ukeys, c, n_entries));
    c = p->cap_e; PCHK(grow_raw(p->pvals, c, n_entries));
    c = p->cap_e; PCHK(grow_raw(p->bcount, c, n_entries));
    p->cap_e = c;
  }
  // s_pts / s_w are sized in floats; a float64 state needs twice that
  const size_t esz = f.state_f64 ? 2 : 1;
  PCHK(grow_raw(plan.s_pts, plan.cap_pts, esz * 3 * pos_bound));
  PCHK(grow_raw(plan.s_idx, plan.cap_idx, 4 * pos_bound));
  PCHK(grow_raw(plan.s_w, plan.cap_w, esz * 4 * pos_bound));
  PCHK(grow_raw(plan.grp_run, plan.cap_grp, pos_bound / 4));
  PCHK(grow_raw(plan.run_nodes, plan.cap_runs, 4 * runs_bound));
  PCHK(grow_raw(plan.run_chunk, plan.cap_rchunk, runs_bound));
  PCHK(grow_raw(plan.slab, plan.cap_slab, (size_t)SLM_SLAB_STRIDE * runs_bound));
  PCHK(grow_raw(plan.blk_key, plan.cap_bkey, n_entries));
  PCHK(grow_raw(plan.blk_start, plan.cap_bstart, n_entries + 1));
  PCHK(grow_raw(plan.blk_entry, plan.cap_bentry, n_entries));

  const dim3 gt((nt + 255) / 256), blk(256);
  hipLaunchKernelGGL(k_padded_counts, gt, blk, 0, st, p->scal, p->tcount, p->pc);
  PCHK(ensure_tmp_for(p, N, p->cap_t, p->cap_e, st));
  size_t bs = p->cap_tmp;
  PCHK(rocprim::exclusive_scan(p->tmp, bs, p->tcount, p->tstart, 0, nt, rocprim::plus<int>(), st));
  PCHK(rocprim::exclusive_scan(p->tmp, bs, p->pc, p->pstart, 0, nt, rocprim::plus<int>(), st));
  hipLaunchKernelGGL(k_run_counts, gt, blk, 0, st, p->scal, p->pstart, p->pc, p->nruns);
  PCHK(rocprim::exclusive_scan(p->tmp, bs, p->nruns, p->rstart, 0, nt, rocprim::plus<int>(), st));
  hipLaunchKernelGGL(k_totals, dim3(1), dim3(1), 0, st, p->scal, p->pstart, p->pc, p->rstart, p->nruns);
  hipLaunchKernelGGL(k_fill_sorted, dim3((pos_bound + 255) / 256), blk, 0, st, (int)pos_bound, p->scal, f,
                     p->tkeys, p->tcount, p->tstart, p->pstart, p->rstart, p->sids, plan.s_pts,
                     plan.s_idx, plan.s_w, plan.grp_run, plan.run_nodes, plan.run_chunk);
  hipLaunchKernelGGL(k_pairs, dim3((runs_bound + 255) / 256), blk, 0, st, (int)runs_bound, p->scal, f.J,
                     plan.run_nodes, p->pkeys, p->pvals);
  size_t b3 = p->cap_tmp, b4 = p->cap_tmp, b5 = p->cap_tmp;
  // (only the bits a key can have are sorted: every 8 bits less is one launch less in a chain of ~90 small launches.
  //  A live pair key is < J*J <= 2^pbits - 1, the padding key 0xFFFFFFFF has all of those bits set and still sorts last.)
  unsigned pbits = 1;
  while (pbits < 32 && (1ull << pbits) <= (unsigned long long)f.J * (unsigned long long)f.J) ++pbits;
  PCHK(rocprim::radix_sort_pairs(p->tmp, b3, p->pkeys, p->spkeys, p->pvals, plan.blk_entry, n_entries, 0,
                                 pbits, st));
  PCHK(rocprim::run_length_encode(p->tmp, b4, p->spkeys, n_entries, p->ukeys, p->bcount, p->scal + 4, st));
  // scan / copy over the full bound: entries past the unique count are never read
  PCHK(rocprim::exclusive_scan(p->tmp, b5, p->bcount, plan.blk_start, 0, n_entries, rocprim::plus<int>(), st));
  PCHK(hipMemcpyAsync(plan.blk_key, p->ukeys, sizeof(unsigned) * n_entries, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_totals2, dim3(1), dim3(1), 0, st, p->scal, p->ukeys, plan.blk_start, (int)n_entries);
  // ---- phase C: (workgroup, pair) records for the LDS-merged Gram path -------------------
  const size_t n_wg = pos_bound / 256 + 1;
  if (n_entries > p->cap_w) {
    size_t c;
    c = p->cap_w; PCHK(grow_raw(p->wkeys, c, n_entries));
    c = p->cap_w; PCHK(grow_raw(p->swkeys, c, n_entries));
    c = p->cap_w; PCHK(grow_raw(p->uwkeys, c, n_entries));
    c = p->cap_w; PCHK(grow_raw(p->wvals, c, n_entries));
    c = p->cap_w; PCHK(grow_raw(p->swvals, c, n_entries));
    c = p->cap_w; PCHK(grow_raw(p->wcount, c, n_entries));
    c = p->cap_w; PCHK(grow_raw(p->wstart, c, n_entries));
    c = p->cap_w; PCHK(grow_raw(p->pv2, c, n_entries));
    c = p->cap_w; PCHK(grow_raw(p->spv2, c, n_entries));
    c = p->cap_w; PCHK(grow_raw(p->b2count, c, n_entries));
    c = p->cap_w; PCHK(grow_raw(p->pk2, c, n_entries));
    c = p->cap_w; PCHK(grow_raw(p->spk2, c, n_entries));
    c = p->cap_w; PCHK(grow_raw(p->upk2, c, n_entries));
    p->cap_w = c;
  }
  {
    size_t c1 = plan.cap_wg, c2 = plan.cap_wg;
    PCHK(grow_raw(plan.wg_first, c1, n_wg));
    PCHK(grow_raw(plan.wg_last, c2, n_wg));
    plan.cap_wg = c1 < c2 ? c1 : c2;
  }
  PCHK(grow_raw(plan.run_lidx, plan.cap_lidx, 10 * runs_bound));
  PCHK(grow_raw(plan.blk2_start, plan.cap_b2start, n_entries + 1));
  PCHK(grow_raw(plan.blk2_entry, plan.cap_b2entry, n_entries));
  PCHK(hipMemsetAsync(p->scal + 5, 0, 3 * sizeof(int), st));
  PCHK(hipMemsetAsync(plan.wg_first, 0, n_wg * sizeof(int), st));
  PCHK(hipMemsetAsync(plan.wg_last, 0xFF, n_wg * sizeof(int), st));   // -1
  hipLaunchKernelGGL(k_pairs2, dim3((runs_bound + 255) / 256), blk, 0, st, (int)runs_bound, p->scal, f.J,
                     plan.run_nodes, plan.run_chunk, p->wkeys, p->wvals);
  size_t c1 = p->cap_tmp, c2 = p->cap_tmp, c3 = p->cap_tmp, c4 = p->cap_tmp, c5 = p->cap_tmp;
  unsigned wbits = 1;   // a live key's workgroup is < n_wg <= 2^wbits - 1; the padding key ~0 sorts last
  while (wbits < 32 && (1ull << wbits) <= (unsigned long long)n_wg) ++wbits;
  PCHK(rocprim::radix_sort_pairs(p->tmp, c1, p->wkeys, p->swkeys, p->wvals, p->swvals, n_entries, 0, 32 + wbits, st));
  PCHK(rocprim::run_length_encode(p->tmp, c2, p->swkeys, n_entries, p->uwkeys, p->wcount, p->scal + 5, st));
  PCHK(rocprim::exclusive_scan(p->tmp, c3, p->wcount, p->wstart, 0, n_entries, rocprim::plus<int>(), st));
  const dim3 ge((n_entries + 255) / 256);
  hipLaunchKernelGGL(k_wg_bounds, ge, blk, 0, st, (int)n_entries, p->scal, p->uwkeys, plan.wg_first, plan.wg_last,
                     p->pk2, p->pv2);
  hipLaunchKernelGGL(k_wg_max, dim3((n_wg + 255) / 256), blk, 0, st, (int)n_wg, plan.wg_first, plan.wg_last,
                     p->scal);
  hipLaunchKernelGGL(k_run_lidx, ge, blk, 0, st, (int)n_entries, p->scal, p->uwkeys, p->wstart, p->wcount,
                     p->swvals, plan.wg_first, plan.run_lidx);
  // pair -> records: same pair order as blk_key (both are the ascending unique pair keys)
  PCHK(rocprim::radix_sort_pairs(p->tmp, c4, p->pk2, p->spk2, p->pv2, plan.blk2_entry, n_entries, 0, pbits, st));
  PCHK(rocprim::run_length_encode(p->tmp, c5, p->spk2, n_entries, p->upk2, p->b2count, p->scal + 4, st));
  PCHK(rocprim::exclusive_scan(p->tmp, c3, p->b2count, plan.blk2_start, 0, n_entries, rocprim::plus<int>(), st));
  hipLaunchKernelGGL(k_totals3, dim3(1), dim3(1), 0, st, p->scal, plan.blk2_start);
  // hash of the coupling graph (node KNN table + pair keys): rides along with the sizes in the one read-back
  PCHK(hipMemsetAsync(p->scal + 8, 0, 4 * sizeof(int), st));
  hipLaunchKernelGGL(k_plan_hash, dim3(16), blk, 0, st, f.J, f.K_ED, f.ed_knn_idx, plan.blk_key, p->scal,
                     reinterpret_cast<unsigned long long*>(p->scal + 8));

  PCHK(hipMemcpyAsync(p->scal_host, p->scal, 14 * sizeof(int), hipMemcpyDeviceToHost, st));
  PCHK(hipStreamSynchronize(st));
  out->bad_knn = p->scal_host[13] != 0;
  if (out->bad_knn) {   // the caller refuses the frame; the next preparation starts without a hint
    plan.nt_hint = 0;
    return hipSuccess;
  }
  if (hinted && (size_t)p->scal_host[12] > nt) {   // the hinted bound did not hold: once more with the exact count
    plan.nt_hint = 0;
    return prep_v1(p, f, plan, out, st);
  }
  plan.nt_hint = p->scal_host[0];
  memcpy(&out->knn_hash, p->scal_host + 8, 8);
  memcpy(&out->graph_hash, p->scal_host + 10, 8);
  out->n_tuples = p->scal_host[0];
  out->n_pos = (p->scal_host[1] + 63) / 64 * 64;
  out->n_runs = p->scal_host[2];
  out->n_blocks = p->scal_host[3];
  out->n_wblk = p->scal_host[6];
  out->max_wblk_per_wg = p->scal_host[7];
  PCHK(grow_raw(plan.wgslab, plan.cap_wgslab, (size_t)SLM_WREC * (out->n_wblk + 1)));
  return hipGetLastError();
}
