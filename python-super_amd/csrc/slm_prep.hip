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
// (n = N * K table entries: any num_neighbors)
__global__ void __launch_bounds__(256) k_check_knn(long long n, int J, const int* __restrict__ knn, int* __restrict__ bad) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if ((unsigned)knn[i] >= (unsigned)J) *bad = 1;
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

// ---- K-generic pair plan (prep_pairs) ---------------------------------------------------------------------------------
// keys of the K(K+1)/2 node pairs of every surfel, slot (ka, kb <= ka) at ka(ka+1)/2 + kb: max(id)*J + min(id)
template <int KK>
__global__ void __launch_bounds__(256) k_pair_keys(int N, int J, const int* __restrict__ knn, unsigned* __restrict__ keys,
                                                    int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  constexpr int NP = KK * (KK + 1) / 2;
  int id[KK];
  bool b = false;
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    id[k] = knn[(size_t)KK * i + k];
    if ((unsigned)id[k] >= (unsigned)J) {
      b = true;
      id[k] = min(max(id[k], 0), J - 1);
    }
  }
  if (b) *bad = 1;
#pragma unroll
  for (int ka = 0; ka < KK; ++ka)
#pragma unroll
    for (int kb = 0; kb <= ka; ++kb) {
      const int a = max(id[ka], id[kb]), c = min(id[ka], id[kb]);
      keys[(size_t)NP * i + ka * (ka + 1) / 2 + kb] = (unsigned)a * (unsigned)J + (unsigned)c;
    }
}
// Canonical neighbour order of a surfel: its K node ids ascending (they are distinct).  Slot (ra, rb <= ra) of the surfel is
// the pair (c[ra], c[rb]) -- the larger id first, as in the pair keys -- so two surfels with the same neighbour SET have the
// same pair in every slot whatever the distance order of their KNN lists, and no block is ever transposed.
template <int KK>
__device__ __forceinline__ void canon_ids(const int* __restrict__ knn, int i, int J, int c[KK]) {
#pragma unroll
  for (int k = 0; k < KK; ++k) c[k] = min(max(knn[(size_t)KK * i + k], 0), J - 1);
#pragma unroll
  for (int a = 1; a < KK; ++a)   // insertion sort, fully unrolled compare-exchanges
#pragma unroll
    for (int b = a; b > 0; --b) {
      const int lo = min(c[b - 1], c[b]), hi = max(c[b - 1], c[b]);
      c[b - 1] = lo;
      c[b] = hi;
    }
}
// per surfel: position of each canonical slot's pair in the sorted unique list (binary searches; once per bind) and the
// 64-bit order key of the surfel (its four smallest ids): surfels with the same neighbour set become neighbours in sf_perm
template <int KK>
__global__ void __launch_bounds__(256) k_pair_index(int N, int J, int n_blocks, const int* __restrict__ knn,
                                                     const unsigned* __restrict__ ukeys, int* __restrict__ pidx,
                                                     unsigned long long* __restrict__ okey, unsigned long long* __restrict__ okey2,
                                                     int* __restrict__ oid) {
  constexpr int NP = KK * (KK + 1) / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int c[KK];
  canon_ids<KK>(knn, i, J, c);
#pragma unroll
  for (int ra = 0; ra < KK; ++ra)
#pragma unroll
    for (int rb = 0; rb <= ra; ++rb) {
      const unsigned key = (unsigned)c[ra] * (unsigned)J + (unsigned)c[rb];
      int lo = 0, hi = n_blocks - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ukeys[mid] < key) lo = mid + 1;
        else hi = mid;
      }
      pidx[(size_t)NP * i + ra * (ra + 1) / 2 + rb] = lo;
    }
  // order keys: the canonical tuple, lexicographic -- ids 0..3 in okey, ids 4..7 in okey2 (K > 4: a second, less significant
  // sort key; tuples that share a prefix keep the pairs of that prefix in the same slots)
  unsigned long long k64 = 0, k64b = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) k64 = (k64 << 16) | (unsigned long long)(k < KK ? c[k] : 0);
#pragma unroll
  for (int k = 4; k < 8; ++k) k64b = (k64b << 16) | (unsigned long long)(k < KK ? c[k] : 0);
  okey[i] = k64;
  if (KK > 4) okey2[i] = k64b;
  oid[i] = i;
}
__global__ void __launch_bounds__(256) k_gather_keys(int N, const unsigned long long* __restrict__ keys, const int* __restrict__ order,
                                                      unsigned long long* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) out[i] = keys[order[i]];
}
__global__ void k_pair_count(int* __restrict__ scal, const unsigned* __restrict__ cnt) { scal[3] = (int)cnt[0]; }

// blk2_start[n_blocks] = number of live records (end of the last live pair)
__global__ void k_totals3(const int* __restrict__ scal, int* __restrict__ blk2_start) {
  blk2_start[scal[3]] = scal[6];
}


// =====================================================================================================================
// Binned preparation (round 4).  The rocPRIM pipeline above is four device-wide sorts, each followed by a run-length
// encode and a scan: ~88 small launches per frame, launch latency from end to end (0.43 ms warm at C2 for one frame; 8
// concurrent binds per step are launch-RATE bound: 700 launches in 1.6 ms).  Every one of those sorts groups items by a
// key whose leading field is a NODE id, and a node owns few items (C2: ~100 surfels, ~30 pair entries): so the items are
// binned by that node (count -> one-block scan -> scatter) and every bin is sorted by ONE workgroup in LDS (bitonic,
// <= BIN_CAP items), which also finds the distinct keys of its bin.  Bins are contiguous and ascending, so the result is
// the stable device-wide sort of the rocPRIM pipeline ITEM FOR ITEM -- the plan, and with it every floating-point
// summation order downstream, is unchanged (tests/test_gpu_prepare_binned.py compares the plans array by array).
// 19 launches per frame.  A bin that does not fit (degenerate inputs: thousands of surfels on one node) sends the slot's
// plan to the rocPRIM pipeline for good (V1Plan::legacy).
#define BIN_CAP 1024

__device__ __forceinline__ bool pair_less(unsigned long long ka, unsigned va, unsigned long long kb, unsigned vb) {
  return ka < kb || (ka == kb && va < vb);
}
// ascending bitonic sort of n <= BIN_CAP (key, val) pairs held in LDS, by (key, val); 256 threads, ends on a barrier
__device__ void lds_sort_pairs(unsigned long long* k, unsigned* v, int n) {
  int m = 2;
  while (m < n) m <<= 1;
  for (int i = n + (int)threadIdx.x; i < m; i += 256) {
    k[i] = ~0ull;
    v[i] = ~0u;
  }
  __syncthreads();
  for (int size = 2; size <= m; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < (m >> 1); i += 256) {
        const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long ka = k[lo], kb = k[hi];
        const unsigned va = v[lo], vb = v[hi];
        if (pair_less(kb, vb, ka, va) == up) {
          k[lo] = kb; k[hi] = ka;
          v[lo] = vb; v[hi] = va;
        }
      }
      __syncthreads();
    }
}
// in-place exclusive scan of a[0..n) (LDS or global, one block of NT threads); returns the total; tmp: NT ints of LDS
// (per-thread chunk sums, a shuffle scan inside every wave, the wave totals scanned by the first wave: three barriers)
template <int NT>
__device__ int block_excl_scan(int* a, int n, int* tmp) {
  const int chunk = (n + NT - 1) / NT, b0 = threadIdx.x * chunk, b1 = min(b0 + chunk, n);
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  int s = 0;
  for (int i = b0; i < b1; ++i) s += a[i];
  int inc = s;                                   // inclusive scan over the wave's 64 chunk sums
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int x = __shfl_up(inc, off, 64);
    if (l >= off) inc += x;
  }
  __syncthreads();                               // (tmp may still be read by a previous scan's last step)
  if (l == 63) tmp[w] = inc;
  __syncthreads();
  if (w == 0) {
    const int nw = NT / 64;
    int t = l < nw ? tmp[l] : 0, ti = t;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int x = __shfl_up(ti, off, 64);
      if (l >= off) ti += x;
    }
    if (l < nw) tmp[l] = ti - t;                 // waves before this one
    if (l == nw - 1) tmp[NT / 64] = ti;          // total
  }
  __syncthreads();
  const int total = tmp[NT / 64];
  int run = tmp[w] + inc - s;
  for (int i = b0; i < b1; ++i) {
    const int x = a[i];
    a[i] = run;
    run += x;
  }
  __syncthreads();
  return total;
}

// per-bin value tables in PrepBuffers::binv, (J + 1) ints each
enum { BV_STARTA = 0, BV_NTUP, BV_NPOS, BV_TUPOFF, BV_POSOFF, BV_RUNS, BV_RUNOFF, BV_STARTB, BV_NUQ, BV_UQOFF, BV_STARTC, BV_COUNT };

// Surfels arrive in raster order (the model is built from depth-map pixels row by row and compacted stably), so the lanes of
// a wave fall into a few RUNS of equal bins: one global atomic per run instead of one per lane (device-scope atomics run
// at ~5 G/s on this part, the two per-surfel passes were 39 + 43 us of a 0.32 ms bind at C2).
// Returns, for every lane, the head lane of its run of equal `bin` values among consecutive active lanes, and the run's length.
__device__ __forceinline__ void lane_runs(int bin, bool active, int& head_lane, int& run_len) {
  const int l = threadIdx.x & 63;
  const int prev = __shfl_up(bin, 1, 64);
  const bool prev_active = __shfl_up(active ? 1 : 0, 1, 64) != 0;
  const bool head = active && (l == 0 || !prev_active || prev != bin);
  const unsigned long long heads = __ballot(head), act = __ballot(active);
  // head of my run: the highest head bit at or below my lane
  const unsigned long long below = heads & (l == 63 ? ~0ull : ((2ull << l) - 1ull));
  head_lane = below ? 63 - __clzll((long long)below) : 0;
  // the run ends before the next head or the first inactive lane after its head
  const unsigned long long stops = (heads | ~act) & ~((head_lane == 63) ? ~0ull : ((2ull << head_lane) - 1ull));
  const int end = stops ? __ffsll((long long)stops) - 1 : 64;
  run_len = end - head_lane;
}

// A1: canonical tuple key of every surfel, range test, surfels per smallest node
__global__ void __launch_bounds__(256) kb_keys(int N, int J, const int* __restrict__ knn, unsigned long long* __restrict__ keys,
                                                int* __restrict__ cntA, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < N;
  int a = 0;
  if (active) {
    int4 v = *reinterpret_cast<const int4*>(knn + 4 * (size_t)i);
    int b = v.y, c = v.z, d = v.w;
    a = v.x;
    if ((unsigned)a >= (unsigned)J || (unsigned)b >= (unsigned)J || (unsigned)c >= (unsigned)J || (unsigned)d >= (unsigned)J) {
      *bad = 1;
      a = min(max(a, 0), J - 1); b = min(max(b, 0), J - 1); c = min(max(c, 0), J - 1); d = min(max(d, 0), J - 1);
    }
    sort4(a, b, c, d);
    keys[i] = ((unsigned long long)a << 48) | ((unsigned long long)b << 32) | ((unsigned long long)c << 16) | (unsigned long long)d;
  }
  int head_lane, run_len;
  lane_runs(a, active, head_lane, run_len);
  if (active && (int)(threadIdx.x & 63) == head_lane) atomicAdd(&cntA[a], run_len);
}

// A2 / B2: counts -> bin starts (exclusive scan over nb bins, start[nb] = total), counts zeroed (they become the scatter
// cursors); the largest bin sets the overflow flag.  One block of 1024.
__global__ void __launch_bounds__(1024) kb_scan_bins(int nb, int* __restrict__ cnt, int* __restrict__ start, int* __restrict__ scal, int cap) {
  __shared__ int tmp[1024];
  __shared__ int s_max;
  if (threadIdx.x == 0) s_max = 0;
  __syncthreads();
  int mx = 0;
  for (int i = threadIdx.x; i < nb; i += 1024) {
    const int c = cnt[i];
    start[i] = c;
    mx = max(mx, c);
  }
  atomicMax(&s_max, mx);
  __syncthreads();
  const int total = block_excl_scan<1024>(start, nb, tmp);
  for (int i = threadIdx.x; i < nb; i += 1024) cnt[i] = 0;
  if (threadIdx.x == 0) {
    start[nb] = total;
    if (s_max > scal[14]) scal[14] = s_max;
    if (s_max > cap) scal[15] = 1;
  }
}

// A3: surfels into their bins (the order inside a bin is settled by the sort); one cursor update per run of lanes
__global__ void __launch_bounds__(256) kb_scatter(int N, const unsigned long long* __restrict__ keys, const int* __restrict__ start,
                                                   int* __restrict__ cursor, unsigned long long* __restrict__ bkeys, int* __restrict__ bids,
                                                   const int* __restrict__ scal) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (scal[15]) return;
  const bool active = i < N;
  const unsigned long long k = active ? keys[i] : 0ull;
  const int a = (int)(k >> 48);
  int head_lane, run_len;
  lane_runs(a, active, head_lane, run_len);
  const int l = threadIdx.x & 63;
  int base = 0;
  if (active && l == head_lane) base = start[a] + atomicAdd(&cursor[a], run_len);
  base = __shfl(base, head_lane, 64);
  if (active) {
    const int p = base + (l - head_lane);
    bkeys[p] = k;
    bids[p] = i;
  }
}

// A4: one workgroup per bin: sort by (key, surfel id), the distinct tuples of the bin (their first element and padded
// local start, stored at the bin's own offset), tuple / padded-position counts of the bin
__global__ void __launch_bounds__(256) kb_sort_bins(int J, const int* __restrict__ binv, unsigned long long* __restrict__ bkeys,
                                                     int* __restrict__ bids, int* __restrict__ sp_head, int* __restrict__ sp_pcl,
                                                     int* __restrict__ ntup_out, int* __restrict__ npos_out, const int* __restrict__ scal) {
  __shared__ unsigned long long k[BIN_CAP];
  __shared__ unsigned v[BIN_CAP];
  __shared__ int hs[BIN_CAP], hl[BIN_CAP];
  __shared__ int tmp[256];
  if (scal[15]) return;
  const int b = blockIdx.x;
  const int* startA = binv + (size_t)BV_STARTA * (J + 1);
  const int base = startA[b], n = startA[b + 1] - base;
  if (n <= 0) {
    if (threadIdx.x == 0) { ntup_out[b] = 0; npos_out[b] = 0; }
    return;
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    k[i] = bkeys[base + i];
    v[i] = (unsigned)bids[base + i];
  }
  __syncthreads();
  lds_sort_pairs(k, v, n);
  for (int i = threadIdx.x; i < n; i += 256) {
    bkeys[base + i] = k[i];
    bids[base + i] = (int)v[i];
    hs[i] = (i == 0 || k[i] != k[i - 1]) ? 1 : 0;
  }
  __syncthreads();
  // tuple index of every head
  for (int i = threadIdx.x; i < n; i += 256) hl[i] = hs[i];
  __syncthreads();
  const int ntup = block_excl_scan<256>(hl, n, tmp);      // hl[i] = tuples before element i
  for (int i = threadIdx.x; i < n; i += 256)
    if (hs[i]) sp_head[base + hl[i]] = i;
  __syncthreads();
  // (the heads just written are read back by this block only)
  __threadfence_block();
  for (int t = threadIdx.x; t < ntup; t += 256) {
    const int h0 = sp_head[base + t], h1 = t + 1 < ntup ? sp_head[base + t + 1] : n;
    hs[t] = ((h1 - h0) + 3) & ~3;                          // padded count of tuple t
  }
  __syncthreads();
  const int npos = block_excl_scan<256>(hs, ntup, tmp);
  for (int t = threadIdx.x; t < ntup; t += 256) sp_pcl[base + t] = hs[t];
  if (threadIdx.x == 0) { ntup_out[b] = ntup; npos_out[b] = npos; }
}

// A5: tuples / padded positions before every bin; totals; the plan buffers must hold them
__global__ void __launch_bounds__(1024) kb_scan_layout(int J, int* __restrict__ binv, int* __restrict__ scal, int pos_bound) {
  __shared__ int tmp[1024];
  if (scal[15]) return;
  int* ntup = binv + (size_t)BV_NTUP * (J + 1);
  int* npos = binv + (size_t)BV_NPOS * (J + 1);
  int* tupoff = binv + (size_t)BV_TUPOFF * (J + 1);
  int* posoff = binv + (size_t)BV_POSOFF * (J + 1);
  for (int i = threadIdx.x; i < J; i += 1024) { tupoff[i] = ntup[i]; posoff[i] = npos[i]; }
  __syncthreads();
  const int nt = block_excl_scan<1024>(tupoff, J, tmp);
  const int ptot = block_excl_scan<1024>(posoff, J, tmp);
  if (threadIdx.x == 0) {
    scal[0] = nt;
    scal[12] = nt;
    scal[1] = ptot;
    if (pos_bound > 0 && (ptot + 63) / 64 * 64 > pos_bound) scal[15] = 2;
  }
}

// A6: runs ((tuple, 64-position chunk) pairs) of every tuple of the bin, local run offsets, runs of the bin
__global__ void __launch_bounds__(256) kb_runs(int J, int* __restrict__ binv, const int* __restrict__ sp_head, const int* __restrict__ sp_pcl,
                                                int* __restrict__ sp_rl, const int* __restrict__ scal) {
  __shared__ int nr[BIN_CAP];
  __shared__ int tmp[256];
  if (scal[15]) return;
  const int b = blockIdx.x;
  const int* startA = binv + (size_t)BV_STARTA * (J + 1);
  const int base = startA[b], n = startA[b + 1] - base;
  const int ntup = binv[(size_t)BV_NTUP * (J + 1) + b];
  int* runs = binv + (size_t)BV_RUNS * (J + 1);
  if (ntup <= 0) {
    if (threadIdx.x == 0) runs[b] = 0;
    return;
  }
  const int posoff = binv[(size_t)BV_POSOFF * (J + 1) + b];
  for (int t = threadIdx.x; t < ntup; t += 256) {
    const int h0 = sp_head[base + t], h1 = t + 1 < ntup ? sp_head[base + t + 1] : n;
    const int pc = ((h1 - h0) + 3) & ~3;
    const int ps = posoff + sp_pcl[base + t], pe = ps + pc;
    nr[t] = (pe - 1) / 64 - ps / 64 + 1;
  }
  __syncthreads();
  const int total = block_excl_scan<256>(nr, ntup, tmp);
  for (int t = threadIdx.x; t < ntup; t += 256) sp_rl[base + t] = nr[t];
  if (threadIdx.x == 0) runs[b] = total;
}

// A7: runs before every bin; total; bound check
__global__ void __launch_bounds__(1024) kb_scan_runs(int J, int* __restrict__ binv, int* __restrict__ scal, int runs_bound) {
  __shared__ int tmp[1024];
  if (scal[15]) return;
  int* runs = binv + (size_t)BV_RUNS * (J + 1);
  int* runoff = binv + (size_t)BV_RUNOFF * (J + 1);
  for (int i = threadIdx.x; i < J; i += 1024) runoff[i] = runs[i];
  __syncthreads();
  const int nruns = block_excl_scan<1024>(runoff, J, tmp);
  if (threadIdx.x == 0) {
    scal[2] = nruns;
    if (runs_bound > 0 && nruns > runs_bound) scal[15] = 2;
  }
}

// A8: the tuple-sorted streams of the bin's positions, the runs' node tuples / chunks; and, per run, its 10 pair entries
// counted per larger node (cntB) and the run range of its workgroup (wgr0x = RB - first run, wgr1 = last run + 1, by max).
// Block J fills the tail [ptot, roundup64(ptot)) with padding.
__global__ void __launch_bounds__(256) kb_fill(int J, const slm_frame f, const int* __restrict__ binv, const unsigned long long* __restrict__ bkeys,
                                                const int* __restrict__ bids, const int* __restrict__ sp_head, const int* __restrict__ sp_pcl,
                                                const int* __restrict__ sp_rl, const int* __restrict__ scal, void* __restrict__ s_pts,
                                                int* __restrict__ s_idx, void* __restrict__ s_w, int* __restrict__ grp_run,
                                                int* __restrict__ run_nodes, int* __restrict__ run_chunk, int* __restrict__ cntB,
                                                int* __restrict__ wgr0x, int* __restrict__ wgr1, int RB) {
  __shared__ int pcl[BIN_CAP];
  if (scal[15]) return;
  const int b = blockIdx.x;
  auto put = [&](int pos, int4 idv, const double wv[4], d3 pp, int run) {
    *reinterpret_cast<int4*>(s_idx + 4 * (size_t)pos) = idv;
    if (f.state_f64) {
      double* sw = static_cast<double*>(s_w) + 4 * (size_t)pos;
      double* sp = static_cast<double*>(s_pts) + 3 * (size_t)pos;
      *reinterpret_cast<double2*>(sw) = make_double2(wv[0], wv[1]);
      *reinterpret_cast<double2*>(sw + 2) = make_double2(wv[2], wv[3]);
      sp[0] = pp.x; sp[1] = pp.y; sp[2] = pp.z;
    } else {
      *reinterpret_cast<float4*>(static_cast<float*>(s_w) + 4 * (size_t)pos) = make_float4((float)wv[0], (float)wv[1], (float)wv[2], (float)wv[3]);
      float* sp = static_cast<float*>(s_pts) + 3 * (size_t)pos;
      sp[0] = (float)pp.x; sp[1] = (float)pp.y; sp[2] = (float)pp.z;
    }
    if ((pos & 3) == 0) grp_run[pos >> 2] = run;
  };
  if (b == J) {
    const int ptot = scal[1], pend = (ptot + 63) / 64 * 64;
    const double z[4] = {0.0, 0.0, 0.0, 0.0};
    for (int pos = ptot + threadIdx.x; pos < pend; pos += 256) put(pos, make_int4(-1, -1, -1, -1), z, {0.0, 0.0, 0.0}, -1);
    return;
  }
  const int* startA = binv + (size_t)BV_STARTA * (J + 1);
  const int base = startA[b], n = startA[b + 1] - base;
  const int ntup = binv[(size_t)BV_NTUP * (J + 1) + b];
  if (ntup <= 0) return;
  const int npos = binv[(size_t)BV_NPOS * (J + 1) + b], posoff = binv[(size_t)BV_POSOFF * (J + 1) + b];
  const int runoff = binv[(size_t)BV_RUNOFF * (J + 1) + b];
  for (int t = threadIdx.x; t < ntup; t += 256) pcl[t] = sp_pcl[base + t];
  __syncthreads();
  for (int pl = threadIdx.x; pl < npos; pl += 256) {
    int lo = 0, hi = ntup - 1;                           // last tuple with pcl <= pl
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (pcl[mid] <= pl) lo = mid; else hi = mid - 1;
    }
    const int t = lo, e = pl - pcl[t];
    const int h0 = sp_head[base + t], h1 = t + 1 < ntup ? sp_head[base + t + 1] : n;
    const int ps = posoff + pcl[t], pos = posoff + pl;
    const int run = runoff + sp_rl[base + t] + (pos / 64 - ps / 64);
    int4 idv = make_int4(-1, -1, -1, -1);
    double wv[4] = {0.0, 0.0, 0.0, 0.0};
    d3 pp = {0.0, 0.0, 0.0};
    if (e < h1 - h0) {
      const int i = bids[base + h0 + e];
      idv = *reinterpret_cast<const int4*>(f.sf_knn_idx + 4 * (size_t)i);
      ld_state4(f.sf_knn_w, (size_t)i, f.state_f64, wv);
      pp = ld_state3(f.sf_points, (size_t)i, f.state_f64);
    }
    put(pos, idv, wv, pp, run);
    if (e == 0 || (pos & 63) == 0) {
      const unsigned long long k = bkeys[base + h0];
      const int nd[4] = {(int)(k >> 48) & 0xFFFF, (int)(k >> 32) & 0xFFFF, (int)(k >> 16) & 0xFFFF, (int)k & 0xFFFF};
      *reinterpret_cast<int4*>(run_nodes + 4 * (size_t)run) = make_int4(nd[0], nd[1], nd[2], nd[3]);
      run_chunk[run] = pos >> 6;
#pragma unroll
      for (int pa = 0; pa < 4; ++pa) atomicAdd(&cntB[nd[pa]], pa + 1);   // entries (pa, pb <= pa): larger node nd[pa]
      const int wg = pos >> 8;
      atomicMax(&wgr0x[wg], RB - run);
      atomicMax(&wgr1[wg], run + 1);
    }
  }
}

// B3: the 10 (a >= b) pair entries of every run into the bin of their larger node
__global__ void __launch_bounds__(256) kb_pair_scatter(int runs_bound, int J, const int* __restrict__ scal, const int* __restrict__ run_nodes,
                                                        const int* __restrict__ startB, int* __restrict__ cursor, unsigned* __restrict__ ekey,
                                                        int* __restrict__ eval) {
  const int rr = blockIdx.x * blockDim.x + threadIdx.x;
  if (scal[15] || rr >= runs_bound || rr >= scal[2]) return;
  const int4 v = *reinterpret_cast<const int4*>(run_nodes + 4 * (size_t)rr);
  const int n[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int pa = 0; pa < 4; ++pa) {
    const int p0 = startB[n[pa]] + atomicAdd(&cursor[n[pa]], pa + 1);   // the pa + 1 entries with this larger node
#pragma unroll
    for (int pb = 0; pb <= pa; ++pb) {
      ekey[p0 + pb] = (unsigned)(n[pa] * J + n[pb]);
      eval[p0 + pb] = rr * 16 + pa * 4 + pb;
    }
  }
}

// B4: sort the bin by (pair key, entry): blk_entry in its final place; distinct pairs of the bin
__global__ void __launch_bounds__(256) kb_sort_pairs(int J, const int* __restrict__ startB, unsigned* __restrict__ ekey, const int* __restrict__ eval,
                                                      int* __restrict__ blk_entry, int* __restrict__ sp_uhead, int* __restrict__ nuq,
                                                      const int* __restrict__ scal) {
  __shared__ unsigned long long k[BIN_CAP];
  __shared__ unsigned v[BIN_CAP];
  __shared__ int hs[BIN_CAP];
  __shared__ int tmp[256];
  if (scal[15]) return;
  const int b = blockIdx.x, base = startB[b], n = startB[b + 1] - base;
  if (n <= 0) {
    if (threadIdx.x == 0) nuq[b] = 0;
    return;
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    k[i] = ekey[base + i];
    v[i] = (unsigned)eval[base + i];
  }
  __syncthreads();
  lds_sort_pairs(k, v, n);
  for (int i = threadIdx.x; i < n; i += 256) {
    ekey[base + i] = (unsigned)k[i];
    blk_entry[base + i] = (int)v[i];
    hs[i] = (i == 0 || k[i] != k[i - 1]) ? 1 : 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) v[i] = (unsigned)hs[i];
  __syncthreads();
  const int nu = block_excl_scan<256>(hs, n, tmp);
  for (int i = threadIdx.x; i < n; i += 256)
    if (v[i]) sp_uhead[base + hs[i]] = i;
  if (threadIdx.x == 0) nuq[b] = nu;
}

// C2: one workgroup per 256-position workgroup of the Jacobian pass: its (workgroup, pair) records = the distinct pair
// keys of its runs' entries; local record index of every (run, pair slot); record keys; records per larger node (cntC)
__global__ void __launch_bounds__(256) kb_wg_records(int n_wg, int J, int RB, const int* __restrict__ scal, const int* __restrict__ wgr0x,
                                                      const int* __restrict__ wgr1, const int* __restrict__ run_nodes,
                                                      uint8_t* __restrict__ run_lidx, unsigned* __restrict__ reckey_sp, int* __restrict__ nrec,
                                                      int* __restrict__ cntC) {
  __shared__ unsigned long long k[1024];
  __shared__ unsigned v[1024];
  __shared__ int hs[1024], hv[1024];
  __shared__ int tmp[256];
  if (scal[15]) return;
  const int wg = blockIdx.x;
  const int r1 = wgr1[wg], r0 = r1 > 0 ? RB - wgr0x[wg] : 0;
  const int n = 10 * (r1 - r0);
  if (n <= 0 || n > 1024) {                              // (<= 64 runs per workgroup by construction: 640 entries)
    if (threadIdx.x == 0) {
      nrec[wg] = 0;
      // an invariant of the layout, not of the data: should it ever break, the plan would silently miss records --
      // hand the slot to the rocPRIM pipeline instead (the final read-back sees the flag)
      if (n > 1024) const_cast<int*>(scal)[15] = 1;
    }
    return;
  }
  for (int j = threadIdx.x; j < n; j += 256) {
    const int rr = r0 + j / 10, e = j % 10;
    int pa = 0, rem = e;
    while (rem > pa) { rem -= pa + 1; ++pa; }            // e -> (pa, pb), pb <= pa, row by row
    const int pb = rem;
    const int na = run_nodes[4 * (size_t)rr + pa], nbn = run_nodes[4 * (size_t)rr + pb];
    k[j] = (unsigned)(na * J + nbn);
    v[j] = (unsigned)(rr * 16 + pa * 4 + pb);
  }
  __syncthreads();
  lds_sort_pairs(k, v, n);
  for (int i = threadIdx.x; i < n; i += 256) {
    hs[i] = (i == 0 || k[i] != k[i - 1]) ? 1 : 0;
    hv[i] = hs[i];
  }
  __syncthreads();
  const int nr = block_excl_scan<256>(hs, n, tmp);        // hs[i] = records before entry i
  for (int i = threadIdx.x; i < n; i += 256) {
    const int lidx = hs[i] + hv[i] - 1;
    const int pl = (int)v[i], run = pl >> 4, pa = (pl >> 2) & 3, pb = pl & 3;
    run_lidx[10 * (size_t)run + pa * (pa + 1) / 2 + pb] = (uint8_t)(lidx < 255 ? lidx : 255);
    if (hv[i]) {
      reckey_sp[10 * (size_t)r0 + lidx] = (unsigned)k[i];
      atomicAdd(&cntC[(int)((unsigned)k[i] / (unsigned)J)], 1);
    }
  }
  if (threadIdx.x == 0) nrec[wg] = nr;
}

// S: the three scans that were waiting: distinct pairs before every bin (-> n_blocks), records before every workgroup
// (wg_first / wg_last, n_wblk, most records of a workgroup), records per larger node -> bin starts (cursors zeroed)
__global__ void __launch_bounds__(1024) kb_scan_index(int J, int n_wg, int* __restrict__ binv, int* __restrict__ cntC, const int* __restrict__ nrec,
                                                       int* __restrict__ wg_first, int* __restrict__ wg_last, int* __restrict__ scal) {
  __shared__ int tmp[1024];
  __shared__ int s_max;
  if (scal[15]) return;
  int* nuq = binv + (size_t)BV_NUQ * (J + 1);
  int* uqoff = binv + (size_t)BV_UQOFF * (J + 1);
  int* startC = binv + (size_t)BV_STARTC * (J + 1);
  if (threadIdx.x == 0) s_max = 0;
  for (int i = threadIdx.x; i < J; i += 1024) { uqoff[i] = nuq[i]; startC[i] = cntC[i]; }
  int mx = 0;
  for (int i = threadIdx.x; i < n_wg; i += 1024) {
    wg_first[i] = nrec[i];
    mx = max(mx, nrec[i]);
  }
  __syncthreads();
  atomicMax(&s_max, mx);
  const int nblocks = block_excl_scan<1024>(uqoff, J, tmp);
  const int nrecs = block_excl_scan<1024>(startC, J, tmp);
  const int nwblk = block_excl_scan<1024>(wg_first, n_wg, tmp);
  for (int i = threadIdx.x; i < J; i += 1024) cntC[i] = 0;
  for (int i = threadIdx.x; i < n_wg; i += 1024) wg_last[i] = wg_first[i] + nrec[i] - 1;
  if (threadIdx.x == 0) {
    uqoff[J] = nblocks;
    startC[J] = nrecs;
    scal[3] = nblocks;
    scal[6] = nwblk;
    scal[7] = s_max;
  }
}

// B6: blk_key / blk_start of the bin's distinct pairs at their global place (block J: the end marker)
__global__ void __launch_bounds__(256) kb_pair_fill(int J, const int* __restrict__ binv, const unsigned* __restrict__ ekey,
                                                     const int* __restrict__ sp_uhead, int* __restrict__ blk_key, int* __restrict__ blk_start,
                                                     const int* __restrict__ scal) {
  if (scal[15]) return;
  const int b = blockIdx.x;
  const int* startB = binv + (size_t)BV_STARTB * (J + 1);
  const int* uqoff = binv + (size_t)BV_UQOFF * (J + 1);
  if (b == J) {
    if (threadIdx.x == 0) blk_start[scal[3]] = startB[J];
    return;
  }
  const int base = startB[b], nu = binv[(size_t)BV_NUQ * (J + 1) + b], u0 = uqoff[b];
  for (int u = threadIdx.x; u < nu; u += 256) {
    const int i = sp_uhead[base + u];
    blk_key[u0 + u] = (int)ekey[base + i];
    blk_start[u0 + u] = base + i;
  }
}

// C6: records into the bin of their larger node
__global__ void __launch_bounds__(256) kb_rec_scatter(int n_wg, int J, int RB, const int* __restrict__ scal, const int* __restrict__ wgr0x,
                                                       const int* __restrict__ wgr1, const int* __restrict__ nrec, const int* __restrict__ wg_first,
                                                       const unsigned* __restrict__ reckey_sp, const int* __restrict__ startC,
                                                       int* __restrict__ cursor, unsigned* __restrict__ rkey, int* __restrict__ ru) {
  if (scal[15]) return;
  const int wg = blockIdx.x;
  const int r1 = wgr1[wg], r0 = r1 > 0 ? RB - wgr0x[wg] : 0;
  const int nr = nrec[wg], u0 = wg_first[wg];
  for (int l = threadIdx.x; l < nr; l += 256) {
    const unsigned key = reckey_sp[10 * (size_t)r0 + l];
    const int a = (int)(key / (unsigned)J);
    const int p = startC[a] + atomicAdd(&cursor[a], 1);
    rkey[p] = key;
    ru[p] = u0 + l;
  }
}

// C7: sort the bin by (pair key, record): blk2_entry in its final place, blk2_start aligned with blk_key (same pairs in
// the same order as phase B's); block J: the end marker
__global__ void __launch_bounds__(256) kb_rec_sort(int J, const int* __restrict__ binv, const unsigned* __restrict__ rkey, const int* __restrict__ ru,
                                                    int* __restrict__ blk2_start, int* __restrict__ blk2_entry, const int* __restrict__ scal) {
  __shared__ unsigned long long k[BIN_CAP];
  __shared__ unsigned v[BIN_CAP];
  __shared__ int hs[BIN_CAP];
  __shared__ int tmp[256];
  if (scal[15]) return;
  const int b = blockIdx.x;
  const int* startC = binv + (size_t)BV_STARTC * (J + 1);
  const int* uqoff = binv + (size_t)BV_UQOFF * (J + 1);
  if (b == J) {
    if (threadIdx.x == 0) blk2_start[scal[3]] = scal[6];
    return;
  }
  const int base = startC[b], n = startC[b + 1] - base;
  if (n <= 0) return;
  if (n > BIN_CAP) {                                       // (never: records of a node <= its entries, checked in phase B)
    if (threadIdx.x == 0) const_cast<int*>(scal)[15] = 1;  // ... and if it ever happens: the rocPRIM pipeline, not a plan with holes
    return;
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    k[i] = rkey[base + i];
    v[i] = (unsigned)ru[base + i];
  }
  __syncthreads();
  lds_sort_pairs(k, v, n);
  for (int i = threadIdx.x; i < n; i += 256) {
    blk2_entry[base + i] = (int)v[i];
    hs[i] = (i == 0 || k[i] != k[i - 1]) ? 1 : 0;
  }
  __syncthreads();
  // distinct-pair index of every head, then its start
  for (int i = threadIdx.x; i < n; i += 256) v[i] = (unsigned)hs[i];
  __syncthreads();
  block_excl_scan<256>(hs, n, tmp);
  const int u0 = uqoff[b];
  for (int i = threadIdx.x; i < n; i += 256)
    if (v[i]) blk2_start[u0 + hs[i]] = base + i;
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
                              // binned preparation: [14] largest bin, [15] 1 = a bin exceeds the LDS sort (legacy path), 2 = a plan buffer bound did not hold
  int* scal_host = nullptr;   // pinned mirror
  // ---- binned preparation (prep_v1_binned) ----
  int* bins = nullptr;        // one zeroed region per bind: [cntA | cntB | cntC | wg_r0 | wg_r1]
  size_t cap_bins = 0;
  int* binv = nullptr;        // per-bin values: startA, ntup, npos, tupoff, posoff, runs, runoff, startB, nuq, uqoff, startC (J + 1 each)
  size_t cap_binv = 0;
  int *sp_head = nullptr, *sp_pcl = nullptr, *sp_rl = nullptr;   // per tuple, stored sparsely at its bin's surfel offset (N each)
  size_t cap_sp = 0;
  unsigned *ekey = nullptr, *rkey = nullptr;   // pair key of every (run, slot) entry / of every record, binned
  int *eval = nullptr, *ru = nullptr, *sp_uhead = nullptr, *nrec = nullptr;
  unsigned* reckey_sp = nullptr;
  size_t cap_ent = 0, cap_nwg = 0;
  // ---- K-generic pair plan (prep_pairs) ----
  unsigned *gk = nullptr, *gsk = nullptr;   // N * K(K+1)/2 pair keys, unsorted / sorted (gk again: the distinct ones)
  unsigned* gcnt = nullptr;                 // their count
  size_t cap_g = 0, q_g = 0, q_gN = 0;      // capacity; key / surfel counts the rocPRIM scratch requirement was last queried for
  void* gtmp = nullptr;
  size_t cap_gtmp = 0;
};

PrepBuffers* prep_create() {
  PrepBuffers* p = new PrepBuffers();
  if (hipMalloc((void**)&p->scal, 32 * sizeof(int)) != hipSuccess ||
      hipHostMalloc((void**)&p->scal_host, 32 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
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
                  p->b2count, p->pk2, p->spk2, p->upk2, p->bins, p->binv, p->sp_head, p->sp_pcl, p->sp_rl, p->ekey, p->rkey,
                  p->eval, p->ru, p->sp_uhead, p->nrec, p->reckey_sp, p->gk, p->gsk, p->gcnt, p->gtmp};
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
  const long long n_ent = (long long)f.N * f.K;
  hipLaunchKernelGGL(k_check_knn, dim3((unsigned)((n_ent + 255) / 256)), dim3(256), 0, st, n_ent, f.J, f.sf_knn_idx, p->scal + 13);
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

static hipError_t prep_v1_legacy(PrepBuffers* p, const slm_frame& f, V1Plan& plan, V1Sizes* out, hipStream_t st) {
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
    return prep_v1_legacy(p, f, plan, out, st);
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

// The binned preparation (kernels above).  Same outputs as prep_v1_legacy, array for array.  Hinted (the plan carries the
// tuple count of its last frame): one read-back at the end; first frame of a plan: one more after the layout pass.
static hipError_t prep_v1_binned(PrepBuffers* p, const slm_frame& f, V1Plan& plan, V1Sizes* out, hipStream_t st, bool* use_legacy) {
  const size_t N = (size_t)f.N;
  const int J = f.J;
  *use_legacy = false;
  out->n_tuples = out->n_pos = out->n_runs = out->n_blocks = 0;
  out->n_wblk = out->max_wblk_per_wg = 0;
  if (N == 0) return hipSuccess;
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
  if (N > p->cap_sp) {
    size_t c;
    c = p->cap_sp; PCHK(grow_raw(p->sp_head, c, N));
    c = p->cap_sp; PCHK(grow_raw(p->sp_pcl, c, N));
    c = p->cap_sp; PCHK(grow_raw(p->sp_rl, c, N));
    p->cap_sp = c;
  }
  const size_t nb1 = (size_t)J + 1;
  const size_t n_wg_max = (4 * N + 63) / 256 + 2;          // positions <= N + 3 * tuples <= 4 N
  PCHK(grow_raw(p->bins, p->cap_bins, 3 * nb1 + 2 * n_wg_max));
  PCHK(grow_raw(p->binv, p->cap_binv, (size_t)BV_COUNT * nb1));
  PCHK(grow_raw(p->nrec, p->cap_nwg, n_wg_max));
  int* cntA = p->bins;
  int* cntB = cntA + nb1;
  int* cntC = cntB + nb1;
  int* wgr0x = cntC + nb1;
  int* wgr1 = wgr0x + n_wg_max;
  int* bv = p->binv;
  auto BV = [&](int which) { return bv + (size_t)which * nb1; };
  PCHK(hipMemsetAsync(p->bins, 0, sizeof(int) * (3 * nb1 + 2 * n_wg_max), st));
  PCHK(hipMemsetAsync(p->scal, 0, sizeof(int) * 16, st));
  const dim3 blk(256);
  hipLaunchKernelGGL(kb_keys, dim3((N + 255) / 256), blk, 0, st, (int)N, J, f.sf_knn_idx, p->keys, cntA, p->scal + 13);
  hipLaunchKernelGGL(kb_scan_bins, dim3(1), dim3(1024), 0, st, J, cntA, BV(BV_STARTA), p->scal, BIN_CAP);
  hipLaunchKernelGGL(kb_scatter, dim3((N + 255) / 256), blk, 0, st, (int)N, p->keys, BV(BV_STARTA), cntA, p->skeys, p->sids, p->scal);
  hipLaunchKernelGGL(kb_sort_bins, dim3(J), blk, 0, st, J, bv, p->skeys, p->sids, p->sp_head, p->sp_pcl, BV(BV_NTUP), BV(BV_NPOS), p->scal);

  const bool hinted = plan.nt_hint > 0;
  size_t nt = 0, pos_bound = 0, runs_bound = 0;
  if (hinted) {
    nt = (size_t)plan.nt_hint + (size_t)plan.nt_hint / 8 + 64;
    if (nt > N) nt = N;
    pos_bound = (N + 3 * nt + 63) / 64 * 64;
    runs_bound = nt + pos_bound / 64 + 1;
  }
  hipLaunchKernelGGL(kb_scan_layout, dim3(1), dim3(1024), 0, st, J, bv, p->scal, (int)pos_bound);
  hipLaunchKernelGGL(kb_runs, dim3(J), blk, 0, st, J, bv, p->sp_head, p->sp_pcl, p->sp_rl, p->scal);
  hipLaunchKernelGGL(kb_scan_runs, dim3(1), dim3(1024), 0, st, J, bv, p->scal, (int)runs_bound);
  if (!hinted) {
    PCHK(hipMemcpyAsync(p->scal_host, p->scal, 16 * sizeof(int), hipMemcpyDeviceToHost, st));
    PCHK(hipStreamSynchronize(st));
    if (p->scal_host[15] == 1) { *use_legacy = true; return hipSuccess; }
    if (p->scal_host[13]) { out->bad_knn = true; plan.nt_hint = 0; return hipSuccess; }
    nt = (size_t)p->scal_host[0];
    if (nt == 0) return hipSuccess;
    pos_bound = ((size_t)p->scal_host[1] + 63) / 64 * 64;
    runs_bound = (size_t)p->scal_host[2];
  }
  const size_t n_entries = 10 * runs_bound;
  const size_t n_wg = pos_bound / 256 + 1;
  // ---- plan buffers (the layout of prep_v1_legacy) ----
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
  {
    size_t c1 = plan.cap_wg, c2 = plan.cap_wg;
    PCHK(grow_raw(plan.wg_first, c1, n_wg));
    PCHK(grow_raw(plan.wg_last, c2, n_wg));
    plan.cap_wg = c1 < c2 ? c1 : c2;
  }
  PCHK(grow_raw(plan.run_lidx, plan.cap_lidx, 10 * runs_bound));
  PCHK(grow_raw(plan.blk2_start, plan.cap_b2start, n_entries + 1));
  PCHK(grow_raw(plan.blk2_entry, plan.cap_b2entry, n_entries));
  if (n_entries > p->cap_ent) {
    size_t c;
    c = p->cap_ent; PCHK(grow_raw(p->ekey, c, n_entries));
    c = p->cap_ent; PCHK(grow_raw(p->rkey, c, n_entries));
    c = p->cap_ent; PCHK(grow_raw(p->eval, c, n_entries));
    c = p->cap_ent; PCHK(grow_raw(p->ru, c, n_entries));
    c = p->cap_ent; PCHK(grow_raw(p->sp_uhead, c, n_entries));
    c = p->cap_ent; PCHK(grow_raw(p->reckey_sp, c, n_entries));
    p->cap_ent = c;
  }
  const int RB = (int)runs_bound + 1;
  hipLaunchKernelGGL(kb_fill, dim3(J + 1), blk, 0, st, J, f, bv, p->skeys, p->sids, p->sp_head, p->sp_pcl, p->sp_rl, p->scal, plan.s_pts,
                     plan.s_idx, plan.s_w, plan.grp_run, plan.run_nodes, plan.run_chunk, cntB, wgr0x, wgr1, RB);
  hipLaunchKernelGGL(kb_scan_bins, dim3(1), dim3(1024), 0, st, J, cntB, BV(BV_STARTB), p->scal, BIN_CAP);
  hipLaunchKernelGGL(kb_pair_scatter, dim3((runs_bound + 255) / 256), blk, 0, st, (int)runs_bound, J, p->scal, plan.run_nodes,
                     BV(BV_STARTB), cntB, p->ekey, p->eval);
  hipLaunchKernelGGL(kb_sort_pairs, dim3(J), blk, 0, st, J, BV(BV_STARTB), p->ekey, p->eval, plan.blk_entry, p->sp_uhead, BV(BV_NUQ), p->scal);
  hipLaunchKernelGGL(kb_wg_records, dim3(n_wg), blk, 0, st, (int)n_wg, J, RB, p->scal, wgr0x, wgr1, plan.run_nodes, plan.run_lidx,
                     p->reckey_sp, p->nrec, cntC);
  hipLaunchKernelGGL(kb_scan_index, dim3(1), dim3(1024), 0, st, J, (int)n_wg, bv, cntC, p->nrec, plan.wg_first, plan.wg_last, p->scal);
  hipLaunchKernelGGL(kb_pair_fill, dim3(J + 1), blk, 0, st, J, bv, p->ekey, p->sp_uhead, plan.blk_key, plan.blk_start, p->scal);
  hipLaunchKernelGGL(kb_rec_scatter, dim3(n_wg), blk, 0, st, (int)n_wg, J, RB, p->scal, wgr0x, wgr1, p->nrec, plan.wg_first, p->reckey_sp,
                     BV(BV_STARTC), cntC, p->rkey, p->ru);
  hipLaunchKernelGGL(kb_rec_sort, dim3(J + 1), blk, 0, st, J, bv, p->rkey, p->ru, plan.blk2_start, plan.blk2_entry, p->scal);
  // hash of the coupling graph (node KNN table + pair keys): rides along with the sizes in the one read-back
  hipLaunchKernelGGL(k_plan_hash, dim3(16), blk, 0, st, f.J, f.K_ED, f.ed_knn_idx, plan.blk_key, p->scal,
                     reinterpret_cast<unsigned long long*>(p->scal + 8));
  PCHK(hipMemcpyAsync(p->scal_host, p->scal, 16 * sizeof(int), hipMemcpyDeviceToHost, st));
  PCHK(hipStreamSynchronize(st));
  out->bad_knn = p->scal_host[13] != 0;
  if (out->bad_knn) {
    plan.nt_hint = 0;
    return hipSuccess;
  }
  if (p->scal_host[15] == 1) { *use_legacy = true; return hipSuccess; }
  if (p->scal_host[15] == 2) {            // a hinted bound did not hold: once more, sizes read back
    plan.nt_hint = 0;
    return prep_v1_binned(p, f, plan, out, st, use_legacy);
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

// Binned preparation unless the slot's plan has met a bin that does not fit the LDS sort (sticky) or SLM_PREP_LEGACY=1.
hipError_t prep_v1(PrepBuffers* p, const slm_frame& f, V1Plan& plan, V1Sizes* out, hipStream_t st) {
  static const bool force_legacy = [] {
    const char* e = getenv("SLM_PREP_LEGACY");
    return e && atoi(e) != 0;
  }();
  if (!force_legacy && !plan.legacy) {
    bool use_legacy = false;
    const hipError_t e = prep_v1_binned(p, f, plan, out, st, &use_legacy);
    if (e != hipSuccess || !use_legacy) return e;
    plan.legacy = true;
    plan.nt_hint = 0;
  }
  return prep_v1_legacy(p, f, plan, out, st);
}


// ---- K-generic pair plan ---------------------------------------------------------------------------------------------
void pairplan_free(PairPlan& plan) {
  if (plan.blk_key) (void)hipFree(plan.blk_key);
  if (plan.sf_pidx) (void)hipFree(plan.sf_pidx);
  if (plan.sf_perm) (void)hipFree(plan.sf_perm);
  plan = PairPlan();
}

#define SLM_PREP_K_DISPATCH(K, CALL)                                   \
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

hipError_t prep_pairs(PrepBuffers* p, const slm_frame& f, PairPlan& plan, PairSizes* out, hipStream_t st) {
  *out = PairSizes();
  if (f.N <= 0 || f.K < 1 || f.K > 8 || f.J >= 65536) return hipSuccess;
  const size_t NP = (size_t)f.K * (f.K + 1) / 2, n = (size_t)f.N * NP;
  if (n > p->cap_g) {
    size_t c;
    c = p->cap_g; PCHK(grow_raw(p->gk, c, n));
    c = p->cap_g; PCHK(grow_raw(p->gsk, c, n));
    p->cap_g = c;
  }
  if (!p->gcnt) PCHK(hipMalloc((void**)&p->gcnt, sizeof(unsigned)));
  // key bits: a*J + b < J*J
  int bits = 1;
  while (bits < 32 && (1ull << bits) < (unsigned long long)f.J * (unsigned long long)f.J) ++bits;
  const size_t N = (size_t)f.N;
  if (N > p->cap_n) {   // (the phase-A buffers of the tuple-sorted preparation: order keys / surfel ids, unsorted and sorted)
    size_t c;
    c = p->cap_n; PCHK(grow_raw(p->keys, c, N));
    c = p->cap_n; PCHK(grow_raw(p->skeys, c, N));
    c = p->cap_n; PCHK(grow_raw(p->tkeys, c, N));
    c = p->cap_n; PCHK(grow_raw(p->ids, c, N));
    c = p->cap_n; PCHK(grow_raw(p->sids, c, N));
    c = p->cap_n; PCHK(grow_raw(p->tcount, c, N));
    p->cap_n = c;
  }
  if (p->q_g != n || p->q_gN != N || !p->gtmp) {
    size_t b1 = 0, b2 = 0, b3 = 0;
    PCHK(rocprim::radix_sort_keys(nullptr, b1, p->gk, p->gsk, n, 0, 32, st));
    PCHK(rocprim::unique(nullptr, b2, p->gsk, p->gk, p->gcnt, n, rocprim::equal_to<unsigned>(), st));
    PCHK(rocprim::radix_sort_pairs(nullptr, b3, p->keys, p->skeys, p->ids, p->sids, N, 0, 64, st));
    b1 = b1 > b3 ? b1 : b3;
    size_t need = (b1 > b2 ? b1 : b2), cap = p->cap_gtmp;
    need += need / 4 + (1u << 16);
    char* q = (char*)p->gtmp;
    PCHK(grow_raw(q, cap, need));
    p->gtmp = q;
    p->cap_gtmp = cap;
    p->q_g = n;
    p->q_gN = N;
  }
  PCHK(hipMemsetAsync(p->scal, 0, 16 * sizeof(int), st));
  const dim3 blk(256);
  SLM_PREP_K_DISPATCH(f.K, hipLaunchKernelGGL(k_pair_keys<KK>, dim3((unsigned)((f.N + 255) / 256)), blk, 0, st, f.N, f.J, f.sf_knn_idx,
                                              p->gk, p->scal + 13));
  size_t b1 = p->cap_gtmp, b2 = p->cap_gtmp;
  PCHK(rocprim::radix_sort_keys(p->gtmp, b1, p->gk, p->gsk, n, 0, bits, st));
  PCHK(rocprim::unique(p->gtmp, b2, p->gsk, p->gk, p->gcnt, n, rocprim::equal_to<unsigned>(), st));
  hipLaunchKernelGGL(k_pair_count, dim3(1), dim3(1), 0, st, p->scal, p->gcnt);
  // the hashes the cached symbolic plan is compared with (same function as the tuple-sorted preparation's)
  hipLaunchKernelGGL(k_plan_hash, dim3(16), blk, 0, st, f.J, f.K_ED, f.ed_knn_idx, reinterpret_cast<const int32_t*>(p->gk), p->scal,
                     reinterpret_cast<unsigned long long*>(p->scal + 8));
  PCHK(hipMemcpyAsync(p->scal_host, p->scal, 16 * sizeof(int), hipMemcpyDeviceToHost, st));
  PCHK(hipStreamSynchronize(st));
  out->bad_knn = p->scal_host[13] != 0;
  if (out->bad_knn) return hipSuccess;
  out->n_blocks = p->scal_host[3];
  memcpy(&out->knn_hash, p->scal_host + 8, 8);
  memcpy(&out->graph_hash, p->scal_host + 10, 8);
  if (out->n_blocks <= 0) return hipSuccess;
  PCHK(grow_raw(plan.blk_key, plan.cap_key, (size_t)out->n_blocks));
  PCHK(grow_raw(plan.sf_pidx, plan.cap_pidx, n));
  PCHK(grow_raw(plan.sf_perm, plan.cap_perm, N));
  PCHK(hipMemcpyAsync(plan.blk_key, p->gk, sizeof(unsigned) * (size_t)out->n_blocks, hipMemcpyDeviceToDevice, st));
  SLM_PREP_K_DISPATCH(f.K, hipLaunchKernelGGL(k_pair_index<KK>, dim3((unsigned)((N + 255) / 256)), blk, 0, st, f.N, f.J, out->n_blocks,
                                              f.sf_knn_idx, reinterpret_cast<const unsigned*>(plan.blk_key), plan.sf_pidx,
                                              p->keys, p->tkeys, p->ids));
  size_t b3 = p->cap_gtmp;
  if (f.K <= 4) {
    PCHK(rocprim::radix_sort_pairs(p->gtmp, b3, p->keys, p->skeys, p->ids, plan.sf_perm, N, 0, 64, st));
  } else {
    // lexicographic order of the full canonical tuple: stable sorts, least significant key (ids 4..7) first
    PCHK(rocprim::radix_sort_pairs(p->gtmp, b3, p->tkeys, p->skeys, p->ids, p->sids, N, 0, 64, st));
    hipLaunchKernelGGL(k_gather_keys, dim3((unsigned)((N + 255) / 256)), blk, 0, st, f.N, p->keys, p->sids, p->tkeys);
    b3 = p->cap_gtmp;
    PCHK(rocprim::radix_sort_pairs(p->gtmp, b3, p->tkeys, p->skeys, p->sids, plan.sf_perm, N, 0, 64, st));
  }
  return hipGetLastError();
}
