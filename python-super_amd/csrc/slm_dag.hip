// slm_dag.hip -- the numeric phase of the nested-dissection multifrontal Cholesky (slm_nd.h) as ONE
// persistent launch: factorisation of every front, forward substitution, extend-adds into the parents
// and the back substitution run as tasks of a static task graph.
//
// Why: with one launch per (level, pivot tile column, phase) -- slm_front.hip, ~125 dependent launches
// per LM iteration at C2 -- an iteration at one frame per launch is a chain of launch boundaries (each
// 3-5 us of dispatch + a kernel prologue of dependent descriptor loads) around 33 sequential 64x64 tile
// factorisations.  Here a dependent step costs one flag hand-off between two resident workgroups
// (measured 1.8-1.9 us including a 32 KB tile written and re-read, tools/micro/launch_gap_mb.hip), tasks
// of different fronts / levels / frames overlap freely, and nothing waits for a whole level.
//
// Scheduling.  The plan carries the task list in a topological order (sorted by modelled earliest start,
// slm_nd_host.hip).  Workgroups take tickets (one agent-scope atomic add) and run the task of their ticket;
// a task waits on flags of tasks with SMALLER tickets only, and a ticket is only ever held by a RUNNING
// workgroup, so the earliest unfinished task can always finish: no deadlock whatever the residency.
// Every spin is bounded; on a time-out the abort flag stops all workgroups and the solve reports failure.
//
// Tasks (left-looking: every tile is written by exactly ONE task, then read-only; the extend-add is a GATHER: a task
// of a parent front pulls the entries of its children's update tiles that map into its tile, child 0 before child 1):
//   POTRF(f,s)   owns the diagonal tile (s,s) AND its left neighbour (s,s-1): accumulates the columns c < s-1 as they
//                become available, then follows the factorisation of column s-1 16 pivots at a time (streamed row
//                solve L(s,s-1) = A(s,s-1) L_{s-1,s-1}^-T and update of (s,s)), factors (s,s) + inverse -> flinv
//                (slm_tile.h factor_inverse64p, itself publishing its row blocks per 16 pivots into the column's
//                mailbox for POTRF(f,s+1) and the COL tasks, which poll the DATA: a non-sentinel value has arrived),
//                y_s = L_ss^-1 (b_s - sum_{c<s} L(s,c) y_c)                                  (forward subst.)
//   COL(f,r,s)   r > s+1 (or a boundary row): L(r,s) = (A(r,s) - sum_{c<s} L(r,c) L(s,c)^T) L_ss^-T
//   SCHUR(f,r,s) boundary tile: U = A(r,s) - sum_{c<npt} L(r,c) L(s,c)^T, stored IN PLACE (the parent gathers it);
//                diagonal tiles carry the vector rows  v_r = b_r - sum_c L(r,c) y_c
//   BACKB(f,c)   y_c -= sum_{boundary r} L(r,c)^T x_r        (x of the boundary nodes from the global solution)
//   BACK(f)      ONE task per front: the chain x_c = L_cc^-T (y_c - sum_{c<r<npt} L(r,c)^T x_r), c = npt-1 .. 0,
//                operands streamed three ahead, x scattered into delta
// All sums run in a fixed order: results are bitwise reproducible from run to run.
//
// Top-of-tree mode (cut >= 0, the hybrid solve of a batch, slm_api.hip enqueue_front_solve): only the tasks of the
// fronts of depth <= cut run here (fd.dag_top_tasks); the deeper levels were factored by the per-level launches of
// slm_front.hip, whose k_fschur stored their update matrices in place -- the fronts at depth == cut gather them like any
// child's, without a flag to wait for (task_deps kids_done).  The list ends with the BACKB / BACK tasks of ALL deeper fronts: the back
// substitution of the whole tree runs here, a deeper front's tasks waiting for their parent's solution only (their own
// tiles and y are final before the launch: `prefactored`).
//
// Memory model (MI355X_MICROARCH.md, inter-workgroup visibility): per-XCD L2s are not coherent and a CU's L1
// is never refreshed, so EVERY byte that one task hands to another (tiles, vectors, inverses, the solution) is
// stored with sc1 (write-through, agent scope) and loaded with sc1 (L1 bypass); a producer drains its stores
// (s_waitcnt vmcnt(0) in every wave, then the workgroup barrier) before ONE lane publishes the flag / counter,
// a consumer polls with relaxed agent-scope loads, then the workgroup barrier.  Descriptors (FrameDev, NDFront,
// task list, maps) are written by the host before the launch and read with plain loads.
#include <atomic>
#include <cstdlib>

#include "slm_tile.h"
#include "slm_lane.h"

namespace {

// Deadlock guard: a wait gives up after DAG_TIMEOUT_TICKS of the constant 100 MHz wall clock (not after a number of
// polls: a poll's duration depends on clocks and memory latency -- a counter-serialised profiler pass or two ranks
// sharing a GPU stretch it).  The clock is first read after 256 unsuccessful polls, so a wait that is served quickly
// never touches it.  A time-out sets the abort flag to 2 and is reported as SLM_ITER_SOLVER_TIMEOUT for the slots
// whose solve did not finish (k_dag_check), distinct from a non-positive pivot.
#define DAG_TIMEOUT_TICKS 300000000ll   // 3 s
__device__ long long g_dag_timeout_ticks = DAG_TIMEOUT_TICKS;   // (slm_debug_dag_timeout: tests shorten it to force an abort)
__device__ __forceinline__ bool dag_timed_out(long long& t0) {
  const long long now = (long long)wall_clock64();
  if (t0 == 0) { t0 = now; return false; }
  return now - t0 > g_dag_timeout_ticks;
}

// Every hand-off access is a GLOBAL-segment instruction with sc1 (global_load / global_store ... sc1): pointers read
// from the slot descriptor are generic, and a flat_ access is not a valid hand-off form (MI355X_MICROARCH.md), so
// they are cast to the global address space explicitly.
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) int gint;
__device__ __forceinline__ double ld1(const double* p) {
  return __hip_atomic_load((const gdouble*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st1(double* p, double v) {
  __hip_atomic_store((gdouble*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ldf(const int* p) {
  return __hip_atomic_load((const gint*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stf(int* p, int v) {
  __hip_atomic_store((gint*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int addf(int* p, int v) {
  return __hip_atomic_fetch_add((gint*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Values read through pointers that were themselves read from memory are "divergent" for the compiler even
// when every lane reads the same address: whole descriptors would live in VGPRs (and spill).  The task loop
// therefore takes a scalar snapshot (v_readfirstlane) of everything it uses from the slot and the front.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ long long uni64(long long x) {
  const int lo = __builtin_amdgcn_readfirstlane((int)(unsigned)(x & 0xFFFFFFFFll));
  const int hi = __builtin_amdgcn_readfirstlane((int)(x >> 32));
  return ((long long)hi << 32) | (unsigned)lo;
}
template <typename T>
__device__ __forceinline__ T* unip(T* p) { return reinterpret_cast<T*>(uni64(reinterpret_cast<long long>(p))); }
template <typename T>
__device__ __forceinline__ T* unip(const GP<T>& p) { return unip(p.get()); }

struct FS {   // scalar snapshot of an NDFront
  int nt, npt, nv, nb, n1, n1p, n2p, parent, which_child, nodes_off, eamap_off, is_leaf;
  int tile0, pcol0;      // first tile / first pivot tile column of the front in the slot's flag arrays
  long long tile_off, f22_base, vec_off, linv_off;
};
__device__ __forceinline__ FS front_snapshot(const NDFront& f) {
  FS o;
  o.nt = uni(f.nt); o.npt = uni(f.npt); o.nv = uni(f.nv); o.nb = uni(f.nb); o.n1 = uni(f.n1); o.n1p = uni(f.n1p);
  o.n2p = uni(f.n2p); o.parent = uni(f.parent); o.which_child = uni(f.which_child);
  o.nodes_off = uni(f.nodes_off); o.eamap_off = uni(f.eamap_off);
  o.tile_off = uni64(f.tile_off); o.f22_base = uni64(f.f22_base); o.vec_off = uni64(f.vec_off); o.linv_off = uni64(f.linv_off);
  o.is_leaf = uni(f.is_leaf);
  o.tile0 = uni(f.tile_first);
  o.pcol0 = (int)(o.linv_off / TILE);
  return o;
}
struct SS {   // scalar snapshot of the slot's buffers
  double *ftiles, *fvec, *flinv, *fmail, *delta;
  const int32_t *nd_nodes, *front_kids, *pull_off, *pullmap, *prng_off, *prng;
  const NDFront* fronts;
  int n_fronts;
  const uint8_t* tile_kind;   // FrameDev::tile_kind (only the factor tasks read it)
  bool cached_ops;            // XCD-affine launch: operand tiles through the XCD's L2 (load_tile_regs2)
};

struct DagFlags {
  int* ticket;
  int* abort_;
  int* tile;
  int* pb;
  int* px;
  int* py;
  int* pk;   // 4 per pivot tile column (unused since the streamed hand-off polls the column's mailbox, slm_tile.h)
};
__device__ __forceinline__ DagFlags dag_flags_of(const FrameDev& fd) {
  DagFlags g;
  int* base = unip(fd.dag_flags);
  const int nt = uni(fd.dag_n_tiles), np = uni(fd.dag_n_pcols);
  g.ticket = base;
  g.abort_ = base + 1;
  g.tile = base + 8;
  g.pb = g.tile + nt;
  g.px = g.pb + np;
  g.py = g.px + np;
  g.pk = g.py + np;
  return g;
}

__device__ __forceinline__ int tile_index(const FS& f, int r, int c) {
  return f.tile0 + c * f.nt - c * (c - 1) / 2 + (r - c);
}
__device__ __forceinline__ double* tile_ptr(const SS& fd, const FS& f, int r, int c) {
  const size_t t = (size_t)c * f.nt - (size_t)c * (c - 1) / 2 + (size_t)(r - c);
  if (c < f.npt) return fd.ftiles + f.tile_off + t * TILE;
  // boundary block (NDFront::f22_base).  Written with the block's own start and (t - pivot tiles): the direct form
  // `ftiles + f22_base + t * TILE` makes this compiler merge the two returns into a pointer select and fail in
  // instruction selection ("Operand has incorrect register class") at the address-space casts of ld1 / st1.
  const size_t piv = (size_t)f.npt * f.nt - (size_t)f.npt * (f.npt - 1) / 2;
  return fd.ftiles + (f.f22_base + (long long)(piv * TILE)) + (t - piv) * TILE;
}
__device__ __forceinline__ int dag_base(const FS& f, int p) {
  return p < f.nv ? 7 * p : f.n1p + 7 * (p - f.nv);
}

// Wait until all `n` flags flag_of(i) (i < n) have reached `want`: the lanes of wave 0 poll one flag each
// (n <= 64 per round).  Returns false when the solve was aborted.  Workgroup barrier inside.
template <typename F>
__device__ __forceinline__ bool dag_wait(int n, F flag_of, int want, int* abort_flag, int* s_abort) {
  if (n > 0 && threadIdx.x < 64) {
    int spins = 0;
    long long t0 = 0;
    for (int base = 0; base < n; base += 64) {
      const int i = base + threadIdx.x;
      const int* fl = i < n ? flag_of(i) : nullptr;
      for (;;) {
        const bool ok = fl ? ldf(fl) >= want : true;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 255) == 0) {
          const bool late = dag_timed_out(t0);
          if (ldf(abort_flag) != 0 || late) {
            if (threadIdx.x == 0) {
              if (late) stf(abort_flag, 2);
              *s_abort = 1;
            }
            base = n;
            break;
          }
        }
      }
    }
  }
  __syncthreads();
  return *s_abort == 0;
}

// producer side of a hand-off: every wave drains its stores, barrier, one lane publishes
__device__ __forceinline__ void dag_publish_begin() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
__device__ __forceinline__ void dag_set_flag(int* flag) {
  if (threadIdx.x == 0) stf(flag, 1);
}

// ---- sc1 forms of the fragment loads / stores of slm_tile.h ------------------------------------------
__device__ __forceinline__ void load_a_frags1(const double* __restrict__ A, double areg[16]) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) areg[ks] = ld1(A + (16 * w + lr) + (size_t)(4 * ks + lk) * NB);
}
__device__ __forceinline__ void load_c_frags1(const double* __restrict__ Cg, double4_t acc[4]) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[ni][r] = ld1(Cg + (16 * w + lr) + (size_t)(16 * ni + lk + 4 * r) * NB);
}
__device__ __forceinline__ void store_c_frags1(double* __restrict__ Cg, const double4_t acc[4]) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) st1(Cg + (16 * w + lr) + (size_t)(16 * ni + lk + 4 * r) * NB, acc[ni][r]);
}
__device__ __forceinline__ void load_tile_regs1(const double* __restrict__ T, double breg[16]) {
#pragma unroll
  for (int e = 0; e < 16; ++e) breg[e] = ld1(T + threadIdx.x + 256 * e);
}
// The same tile with 16 bytes per lane: thread t gets T[2t + 512 e] and T[2t + 512 e + 1] in r[2e], r[2e+1], e = 0..7.
// One workgroup streams a 32 KB tile in 0.29 us this way, in 0.43 us with 8-byte loads (tools/micro/tile_stream_mb.hip:
// the address path, not the memory, is the limit).  The language has no 16-byte atomic load, so these are buffer loads
// with the same cache policy as ld1 (sc1: L1 and the non-coherent L2 bypassed); they follow the task's flag wait in
// program order (a workgroup barrier with its fence stands between), which is all a hand-off needs.  T is uniform.
// cached (uniform): an OPERAND tile -- final, written by a task of the SAME frame -- in an XCD-affine launch (k_fdag): every task
// of the frame runs on one XCD, whose L2 saw the tile's write-through stores; a plain load shares the tile among the frame's
// tasks through that L2 instead of fetching it from HBM once per task.
__device__ __forceinline__ void load_tile_regs2(const double* T, double r[16], bool cached = false) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(T), 0, TILE * 8, 0x00020000);
  if (cached) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const v4u q = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * threadIdx.x + 4096 * e, 0, 0);
      r[2 * e] = __hiloint2double(q.y, q.x);
      r[2 * e + 1] = __hiloint2double(q.w, q.z);
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const v4u q = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * threadIdx.x + 4096 * e, 0, 1 << 4 /* sc1 */);
    r[2 * e] = __hiloint2double(q.y, q.x);
    r[2 * e + 1] = __hiloint2double(q.w, q.z);
  }
}
// ... and its tile-linear copy into LDS (16-byte stores)
__device__ __forceinline__ void store_tile_lds2(double* L, const double r[16]) {
  typedef double dvec2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const dvec2 v = {r[2 * e], r[2 * e + 1]};
    *reinterpret_cast<dvec2*>(L + 2 * threadIdx.x + 512 * e) = v;
  }
}

// ---- the dependency list of a task ---------------------------------------------------------------------
// POTRF / COL / SCHUR of tile (r,s):  [update tiles of child 0 it gathers from][... of child 1]
//                                     [per operand column c < kc: two flags][COL only: the diagonal factor (s,s)]
//   kc = s (POTRF, COL) or npt (SCHUR); the two flags of column c are
//   POTRF: L(s,c), y_c    COL: L(r,c), L(s,c)    SCHUR: L(r,c) and L(s,c), or y_c on diagonal tiles.
//   Stage 0 = everything but the last operand column (and the diagonal factor): what the task needs to START.
// BACKB(f,c): [x of the parent's column 0]
// BACK(f,c):  [y_c][boundary part subtracted (fronts with a boundary)][L(r,c), c < r < npt] | chain: x_r, r = npt-1 .. c+1
struct TD {
  int type, r, s, kc, diag;
  int np[2], pr0[2], pc0[2], pnc[2];     // gathered child tiles: count, first tile row / column, columns
  int np2[2], pc2[2], pnc2[2];           // POTRF(s > 0): the same for its second tile (s, s-1) (same tile rows)
  int ctile0[2], cnt[2], cnpt[2];        // the children's tile numbering
  int n0, n, nskip;                      // flags needed to start / all flags / leading flags that need no wait
  int pcol_self, pcol_parent, nb, npt, c;
};
// kids_done: the front's children were factored by the per-level launches BEFORE this launch (hybrid solve, the fronts
// at the cut): their update tiles are final -- gathered like any child's, but there is no flag to wait for (nskip)
__device__ __forceinline__ TD task_deps(const SS& fd, const FS& f, int fi, int type, int r, int s, bool kids_done) {
  TD d;
  d.type = type; d.r = r; d.s = s; d.diag = (r == s); d.npt = f.npt; d.nb = f.nb; d.c = s;
  d.pcol_self = f.pcol0;
  d.pcol_parent = 0;
  d.np[0] = d.np[1] = 0;
  d.kc = 0;
  d.np2[0] = d.np2[1] = 0;
  d.nskip = 0;
  if (type <= ND_T_SCHUR) {
    const int32_t* pr = fd.prng + uni(fd.prng_off[fi]);
    const bool two = type == ND_T_POTRF && s > 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ch = uni(fd.front_kids[2 * fi + k]);
      const int rr = ch >= 0 ? uni(pr[2 * r + k]) : -1, cc = ch >= 0 ? uni(pr[2 * s + k]) : -1;
      const int c2 = (ch >= 0 && two) ? uni(pr[2 * (s - 1) + k]) : -1;
      d.pr0[k] = d.pc0[k] = d.pc2[k] = 0; d.pnc[k] = d.pnc2[k] = 1; d.ctile0[k] = d.cnt[k] = d.cnpt[k] = 0;
      if (rr >= 0 && (cc >= 0 || c2 >= 0)) {
        const NDFront& cf = fd.fronts[ch];
        d.ctile0[k] = uni(cf.tile_first); d.cnt[k] = uni(cf.nt); d.cnpt[k] = uni(cf.npt);
        d.pr0[k] = rr & 255;
        const int nr = (rr >> 8) - (rr & 255) + 1;
        if (cc >= 0) { d.pc0[k] = cc & 255; d.pnc[k] = (cc >> 8) - (cc & 255) + 1; d.np[k] = nr * d.pnc[k]; }
        if (c2 >= 0) { d.pc2[k] = c2 & 255; d.pnc2[k] = (c2 >> 8) - (c2 & 255) + 1; d.np2[k] = nr * d.pnc2[k]; }
      }
    }
    d.kc = type == ND_T_SCHUR ? f.npt : s;
    const int npull = d.np[0] + d.np[1] + d.np2[0] + d.np2[1];
    d.n0 = npull;                 // to start: the children's update tiles; the operand columns are consumed as they come
    if (kids_done) d.nskip = npull;
    if (type == ND_T_POTRF)       // per column c < s-1: L(s,c), y_c, L(s-1,c); then 4 slots for the 16-pivot rounds of (s-1,s-1)
                                  // (not waited for: the rounds are taken from the column's mailbox), then y_{s-1}
      d.n = npull + 3 * (s > 0 ? s - 1 : 0) + (s > 0 ? 5 : 0);
    else
      d.n = npull + 2 * d.kc + (type == ND_T_COL ? 1 : 0);
  } else if (type == ND_T_BACKB) {
    d.pcol_parent = (int)(uni64(fd.fronts[f.parent].linv_off) / TILE);
    d.n0 = d.n = 1;
  } else {
    // BACK(f): y_c (and its boundary part) of every column, and every tile L(r,c), c < r < npt.  The FUSED form (word 1 of
    // the task = 1: the task subtracts the boundary part itself) waits for the parent's solution in place of the npt flags
    // of its BACKB tasks.
    const bool fusedf = s == 1 && f.nb > 0;   // (word 1 of a BACK task: 0 or 1)
    d.r = fusedf ? 1 : 0;
    if (fusedf) d.pcol_parent = (int)(uni64(fd.fronts[f.parent].linv_off) / TILE);
    d.n0 = d.n = f.npt + (f.nb > 0 ? (fusedf ? 1 : f.npt) : 0) + f.npt * (f.npt - 1) / 2;
  }
  return d;
}
__device__ __forceinline__ const int* dep_flag(const TD& d, const FS& f, const DagFlags& g, int i) {
  if (d.type <= ND_T_SCHUR) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (i < d.np[k]) {
        const int cr = d.cnpt[k] + d.pr0[k] + i / d.pnc[k];
        int cc = d.cnpt[k] + d.pc0[k] + i % d.pnc[k];
        if (cc > cr) cc = cr;            // lower triangle only (a duplicate flag in the list is harmless)
        return g.tile + d.ctile0[k] + cc * d.cnt[k] - cc * (cc - 1) / 2 + (cr - cc);
      }
      i -= d.np[k];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (i < d.np2[k]) {
        const int cr = d.cnpt[k] + d.pr0[k] + i / d.pnc2[k];
        int cc = d.cnpt[k] + d.pc2[k] + i % d.pnc2[k];
        if (cc > cr) cc = cr;
        return g.tile + d.ctile0[k] + cc * d.cnt[k] - cc * (cc - 1) / 2 + (cr - cc);
      }
      i -= d.np2[k];
    }
    if (d.type == ND_T_POTRF) {
      const int s = d.s;
      if (i < 3 * (s - 1)) {
        const int c = i / 3, j = i % 3;
        return j == 0 ? g.tile + tile_index(f, s, c) : (j == 1 ? g.py + f.pcol0 + c : g.tile + tile_index(f, s - 1, c));
      }
      i -= 3 * (s - 1);
      return i < 4 ? g.pk + 4 * (f.pcol0 + s - 1) + i : g.py + f.pcol0 + s - 1;
    }
    if (i < 2 * d.kc) {
      const int c = i >> 1, odd = i & 1;
      if (d.type == ND_T_COL) return g.tile + tile_index(f, odd ? d.s : d.r, c);
      if (d.diag) return odd ? g.py + f.pcol0 + c : g.tile + tile_index(f, d.r, c);
      return g.tile + tile_index(f, odd ? d.s : d.r, c);
    }
    return g.tile + tile_index(f, d.s, d.s);
  }
  if (d.type == ND_T_BACKB) return g.px + d.pcol_parent;
  // BACK(f)
  if (i < d.npt) return g.py + f.pcol0 + i;
  i -= d.npt;
  if (d.nb > 0) {
    if (d.r == 1) {                 // fused: the parent's solution
      if (i < 1) return g.px + d.pcol_parent;
      i -= 1;
    } else {
      if (i < d.npt) return g.pb + f.pcol0 + i;
      i -= d.npt;
    }
  }
  // tiles (r, c), c < r < npt, row by row: i = r (r - 1) / 2 + c
  int r = 1;
  while ((r + 1) * r / 2 <= i) ++r;
  return g.tile + tile_index(f, r, i - r * (r - 1) / 2);
}
// blocking wait for the flags [a, b) of the list
__device__ __forceinline__ bool dag_wait_deps(const TD& d, const FS& f, const DagFlags& g, int a, int b, int* abort_flag,
                                              int* s_abort) {
  if (b <= a) return true;
  return dag_wait(b - a, [&](int i) { return dep_flag(d, f, g, a + i); }, 1, abort_flag, s_abort);
}
// How many leading groups (of `per` flags each) of the flags [a, b) are set?  Waits until at least one group is
// (returns 0 only when the solve was aborted); looks at up to 64 flags per call.  The operand columns of a task are
// consumed in order AS THEY BECOME AVAILABLE: old columns cost one look, and only the newest one is ever waited for.
__device__ __forceinline__ int dag_wait_prefix(const TD& d, const FS& f, const DagFlags& g, int a, int b, int per,
                                               int* abort_flag, int* s_abort, int* s_cnt) {
  const int nfl = min(b - a, 64 / per * per);
  if (threadIdx.x < 64) {
    int spins = 0;
    long long t0 = 0;
    for (;;) {
      const bool ok = (int)threadIdx.x < nfl ? ldf(dep_flag(d, f, g, a + threadIdx.x)) >= 1 : true;
      const unsigned long long m = __ballot(ok);
      const int lead = (m == ~0ull) ? 64 : (__ffsll((long long)~m) - 1);
      const int groups = min(lead, nfl) / per;
      if (groups > 0) {
        if (threadIdx.x == 0) *s_cnt = groups;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255) == 0) {
        const bool late = dag_timed_out(t0);
        if (!(ldf(abort_flag) != 0 || late)) continue;
        if (threadIdx.x == 0) {
          if (late) stf(abort_flag, 2);
          *s_abort = 1;
          *s_cnt = 0;
        }
        break;
      }
    }
  }
  __syncthreads();
  return uni(*s_cnt);
}

// Gather (pull) form of the extend-add: acc (tile (r,s) of front fi in accumulator layout) += the entries of the
// children's update tiles that map into it -- child 0 first, then child 1: a fixed order, so the sum does not
// depend on which workgroup ran when.  (The update tiles are complete: their flags are part of the task's stage 0.)
// VEC (diagonal tiles): bvec (threads < NB, row threadIdx.x of tile row r) += the children's vector rows.
// maps: 256 int2 of LDS per tile: [128 k + 0..63] the rows of tile row r, [128 k + 64..127] the columns of tile column s, child k:
// .x = the child's boundary scalar that maps there (-1: none), .y = the BYTE offset of that row (resp. column) inside the
// child's update block -- an entry's address is block + row offset + column offset (one add per gather instead of the
// tile arithmetic: 1 KB of code per pulled tile instead of 5).  Static plan data, written before the task's wait.
__device__ __forceinline__ void dag_pull_maps(const SS& fd, int fi, int r, int s, const int np[2], int2* maps) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (np[0] == 0 && np[1] == 0) return;
  __syncthreads();   // earlier readers of maps are done
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (np[k] == 0) continue;
    const int ch = uni(fd.front_kids[2 * fi + k]);
    const int32_t* pm = fd.pullmap + uni(fd.pull_off[ch]);
    const int cnpt = uni(fd.fronts[ch].npt), cnt = uni(fd.fronts[ch].nt);
    if (threadIdx.x < 128) {
      const int c = pm[64 * (w == 0 ? r : s) + l];
      int off = 0;
      if (c >= 0) {
        const int t = cnpt + (c >> 6);      // tile row / tile column of the child's front
        off = (w == 0) ? (t * TILE + (c & 63)) * 8 : ((t * cnt - t * (t - 1) / 2 - t) * TILE + (c & 63) * NB) * 8;
      }
      maps[128 * k + threadIdx.x] = make_int2(c, off);
    }
  }
  __syncthreads();
}

template <bool VEC>
__device__ __forceinline__ void dag_pull(const SS& fd, int fi, int r, int s, double4_t acc[4], double& bvec, const int np[2],
                                         int2* maps, bool maps_loaded = false) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
  if (np[0] == 0 && np[1] == 0) return;
  // both children's maps first (or already in LDS: dag_pull_maps before the task's wait), then ALL gathers in flight
  // together (scattered 8-byte sc1 loads: one latency instead of two), then the sums in the fixed order child 0, child 1
  if (!maps_loaded) dag_pull_maps(fd, fi, r, s, np, maps);
  double v[2][16], bv[2] = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[k][e] = 0.0;
    if (np[k] == 0) continue;
    const int ch = uni(fd.front_kids[2 * fi + k]);
    const long long cbase = uni64(fd.fronts[ch].f22_base);   // the child's update tiles live in its boundary block
    const char* ct = reinterpret_cast<const char*>(fd.ftiles + cbase);
    const int2* mk = maps + 128 * k;
    const int2 rw = mk[16 * w + lr];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int2 cl = mk[64 + 16 * ni + lk + 4 * rr];
        if (cl.x >= 0 && rw.x >= cl.x) {
          const double* src = reinterpret_cast<const double*>(ct + (unsigned)(rw.y + cl.y));
          // (XCD-affine launch: the child's update tiles were written by tasks of this frame on this XCD, or by a launch before)
          v[k][4 * ni + rr] = fd.cached_ops ? *(const gdouble*)src : ld1(src);
        }
      }
    if (VEC && threadIdx.x < NB) {
      const int ci = mk[threadIdx.x].x;
      if (ci >= 0) {
        const long long cvec = uni64(fd.fronts[ch].vec_off);
        const int cnpt = uni(fd.fronts[ch].npt);
        bv[k] = ld1(fd.fvec + cvec + (size_t)cnpt * NB + ci);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[ni][rr] += v[k][4 * ni + rr];
    if (VEC) bvec += bv[k];
  }
}

// acc -= sum_{c in [c0,c1)} L(ra,c) L(rb,c)^T.  Both operand tiles pass through LDS (Bl <- L(rb,c), Al <- L(ra,c);
// one tile when ra == rb) with tile-linear, fully coalesced sc1 loads, and the next column's tiles are in
// flight (registers) under the MFMAs.  VEC (needs ra == rb): also tsum += (L(ra,c) y_c)[row threadIdx.x & 63] over
// the 16 inner columns of this thread's quarter; y_c is read from the front's vector.
// The front's LAST pivot tile column only counts up to its true pivots (blocks of 16: the padding columns of L are zero).
template <bool VEC>
__device__ __forceinline__ void dag_accumulate(const SS& fd, const FS& f, int ra, int rb, int c0, int c1,
                                               double4_t acc[4], double* Bl, double* Al, double* yv, double& tsum) {
  if (c0 >= c1) return;
  const int kb_last = min(4, (f.n1 - NB * (f.npt - 1) + 15) >> 4);   // 16-column blocks with true pivots in column npt - 1
  const bool two = ra != rb;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
  double breg[16], areg[16];
  load_tile_regs2(tile_ptr(fd, f, rb, c0), breg, fd.cached_ops);
  if (two) load_tile_regs2(tile_ptr(fd, f, ra, c0), areg, fd.cached_ops);
  double ynext = 0.0;
  if (VEC && threadIdx.x < NB) ynext = ld1(fd.fvec + f.vec_off + (size_t)c0 * NB + threadIdx.x);
  const double* Ar = two ? Al : Bl;
  for (int c = c0; c < c1; ++c) {
    __syncthreads();   // earlier readers of Bl / Al / yv are done
    store_tile_lds2(Bl, breg);
    if (two) store_tile_lds2(Al, areg);
    if (VEC && threadIdx.x < NB) yv[threadIdx.x] = ynext;
    if (c + 1 < c1) {
      load_tile_regs2(tile_ptr(fd, f, rb, c + 1), breg, fd.cached_ops);
      if (two) load_tile_regs2(tile_ptr(fd, f, ra, c + 1), areg, fd.cached_ops);
      if (VEC && threadIdx.x < NB) ynext = ld1(fd.fvec + f.vec_off + (size_t)(c + 1) * NB + threadIdx.x);
    }
    __syncthreads();
    const int kblk = (c == f.npt - 1) ? kb_last : 4;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (kb < kblk) {
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          const int ks = 4 * kb + k4;
          const double a = -Ar[(16 * w + lr) + (4 * ks + lk) * LD];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const double bv = Bl[(16 * ni + lr) + (4 * ks + lk) * LD];
            acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, a, acc[ni], 0, 0, 0);
          }
        }
      }
    }
    if (VEC) {
      const int i = threadIdx.x & 63, q = threadIdx.x >> 6;
#pragma unroll
      for (int k = 16 * q; k < 16 * q + 16; ++k) tsum += Bl[i + k * LD] * yv[k];
    }
  }
}

// POTRF(s), s > 0, columns c in [c0, c1): acc_d (tile (s,s)) -= L(s,c) L(s,c)^T, acc_l (tile (s,s-1)) -= L(s,c) L(s-1,c)^T,
// tsum += (L(s,c) y_c)[row]; L(s,c) is staged in Bl, L(s-1,c) in Al.
__device__ __forceinline__ void dag_accumulate2(const SS& fd, const FS& f, int s, int c0, int c1, double4_t acc_d[4],
                                                double4_t acc_l[4], double* Bl, double* Al, double* yv, double& tsum) {
  if (c0 >= c1) return;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
  double breg[16], areg[16];
  load_tile_regs2(tile_ptr(fd, f, s, c0), breg, fd.cached_ops);
  load_tile_regs2(tile_ptr(fd, f, s - 1, c0), areg, fd.cached_ops);
  double ynext = 0.0;
  if (threadIdx.x < NB) ynext = ld1(fd.fvec + f.vec_off + (size_t)c0 * NB + threadIdx.x);
  for (int c = c0; c < c1; ++c) {
    __syncthreads();
    store_tile_lds2(Bl, breg);
    store_tile_lds2(Al, areg);
    if (threadIdx.x < NB) yv[threadIdx.x] = ynext;
    if (c + 1 < c1) {
      load_tile_regs2(tile_ptr(fd, f, s, c + 1), breg, fd.cached_ops);
      load_tile_regs2(tile_ptr(fd, f, s - 1, c + 1), areg, fd.cached_ops);
      if (threadIdx.x < NB) ynext = ld1(fd.fvec + f.vec_off + (size_t)(c + 1) * NB + threadIdx.x);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const double a = -Bl[(16 * w + lr) + (4 * ks + lk) * LD];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const double bd = Bl[(16 * ni + lr) + (4 * ks + lk) * LD];
        const double bl = Al[(16 * ni + lr) + (4 * ks + lk) * LD];
        acc_d[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(bd, a, acc_d[ni], 0, 0, 0);
        acc_l[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(bl, a, acc_l[ni], 0, 0, 0);
      }
    }
    {
      const int i = threadIdx.x & 63, q = threadIdx.x >> 6;
#pragma unroll
      for (int k = 16 * q; k < 16 * q + 16; ++k) tsum += Bl[i + k * LD] * yv[k];
    }
  }
}

// Transposed matrix-vector product with a tile held tile-linear in registers (lv[e] = T[threadIdx.x + 256 e], i.e.
// row m = lane, column n = wave + 4 e; loaded with fully coalesced instructions): the caller forms
// v[e] = lv[e] * x[lane]; the butterfly then sums every column over the 64 lanes (halving the number of live
// values at each of the first four steps).  Returns, in the lanes with (lane & 3) == 0, the sum
// of column n = wave + 4 * ((lane >> 2) & 15).
// (col_reduce16: slm_lane.h -- the VALU-only butterfly; tools/micro/col_reduce_mb.hip compares it with the __shfl_xor one)

// sum of the four quarter partials of dag_accumulate<true>: result for row i in every thread with (threadIdx.x & 63) == i
__device__ __forceinline__ double dag_reduce_rows(double tsum, double* part /* 4 * NB */) {
  const int i = threadIdx.x & 63, q = threadIdx.x >> 6;
  __syncthreads();
  part[q * NB + i] = tsum;
  __syncthreads();
  return part[i] + part[NB + i] + part[2 * NB + i] + part[3 * NB + i];
}

// S, M, the four diagonal-block inverses, three 16x16 scratch blocks, two vectors, a few ints: 80 960 B -> two
// workgroups per CU.  `part` (row partials) shares the scratch blocks and the extend-add maps share the
// diagonal-block inverses: neither is live while a tile is being factored.
#define DAG_LDS_DOUBLES (2 * TILE + 7 * 256 + 2 * NB + 24)

// Tile factorisation + inverse as a real call: inlined into the task loop it raises the register demand of the
// whole kernel beyond 256 VGPRs (every path pays the maximum).  The LDS regions are derived from the dynamic LDS
// base inside the function, so their address space stays known.
extern __shared__ double dag_lds[];
// look_ahead: wave 0 has staged the first diagonal block at dinv + 256 (ld 16) and the waves' row partials lie in wt
// (factor_inverse64p's look-ahead entry); *la_sum receives their sums (threads < NB)
__device__ __forceinline__ bool dag_factor_tile(double* g_mail, int* g_early, long long* trc, int nblk, bool look_ahead = false,
                                                double* la_sum = nullptr) {
  double* S = dag_lds;
  double* M = dag_lds + TILE;
  double* dinv = dag_lds + 2 * TILE;
  double* wt = dinv + 4 * 256;
  double* xch = wt + 3 * 256;                   // vec | yv: not live while a tile is being factored
  int* s_ok = reinterpret_cast<int*>(xch + 2 * NB);
  int* pf = s_ok + 16;                          // 16 hand-off flags of the trailing waves
  // (ONE inlined copy serves both entries: with two, each of them came out ~1.5 us slower per tile)
  return factor_inverse64p(S, M, dinv, wt, xch, s_ok, pf, g_mail, g_early, trc, nblk, look_ahead ? dinv + 256 : nullptr,
                           look_ahead ? wt : nullptr, la_sum);
}

}  // namespace

// ---- the streamed hand-off, consumer side (producer: publish_blocks16 in slm_tile.h) --------------------------------
// Round kb of a streamed row solve needs the inverse of diagonal block kb and the L blocks (kb, t < kb) of the tile being
// factored by another workgroup: 1 + kb values per thread, read from the column's mailbox.  A value that is not the
// "empty" sentinel HAS arrived -- the data is its own flag.  MAIL_ISSUE requests a round's values (registers only: a
// round is requested before the products of the one before it, speculatively -- what has not arrived reads as "empty"
// and is requested again); MAIL_TAKE stores them into LDS (dinv block kb, row block kb of Lst) and leaves when, workgroup-
// wide, all of them were there.  Its barrier doubles as the barrier behind those LDS stores.
// NB: `dinv` doubles as the pull maps of dag_pull.  The first MAIL_TAKE of a task must be preceded by a workgroup barrier
// that every wave passes after its last look at the maps -- without it a fast wave's round-0 block lands in a slow wave's
// gather indices (seen as results that differed from run to run in the third digit).
__device__ __forceinline__ bool mail_here(double v) { return __double_as_longlong(v) != SLM_MAIL_EMPTY; }
#define MAIL_ISSUE(MAIL, PD, PL, KB)                                                                             \
  do {                                                                                                           \
    PD[KB] = ld1((MAIL) + (KB) * 256 + threadIdx.x);                                                             \
    _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_)                                                             \
      if (j_ < (KB)) PL[KB][j_] = ld1((MAIL) + (4 + (KB) * ((KB) - 1) / 2 + j_) * 256 + threadIdx.x);            \
  } while (0)
#define MAIL_TAKE(MAIL, PD, PL, KB, DINV, LST)                                                                   \
  do {                                                                                                           \
    const int pe_i_ = threadIdx.x & 15, pe_k_ = threadIdx.x >> 4;                                                \
    int* votes_ = s_ok + 4;                                                                                      \
    int spins_ = 0;                                                                                              \
    long long t0_ = 0;                                                                                           \
    for (;;) {                                                                                                   \
      bool here_ = mail_here(PD[KB]);                                                                            \
      (DINV)[(KB) * 256 + pe_i_ + 16 * pe_k_] = PD[KB];                                                          \
      _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_)                                                           \
        if (j_ < (KB)) {                                                                                         \
          here_ = here_ && mail_here(PL[KB][j_]);                                                                \
          (LST)[(16 * (KB) + pe_i_) + (16 * j_ + pe_k_) * LD] = PL[KB][j_];                                      \
        }                                                                                                        \
      bool late_ = false;                                                                                        \
      const bool stop_ = (++spins_ & 63) == 0 && (ldf(abort_flag) != 0 || (late_ = dag_timed_out(t0_)));         \
      const int vote_ = (__all(here_) ? 1 : 0) | (__any(stop_) ? 2 : 0) | (__any(late_) ? 4 : 0);                \
      if ((threadIdx.x & 63) == 0) votes_[threadIdx.x >> 6] = vote_;                                             \
      __syncthreads();                                                                                           \
      const int v0_ = votes_[0], v1_ = votes_[1], v2_ = votes_[2], v3_ = votes_[3];                              \
      if ((v0_ & v1_ & v2_ & v3_ & 1) != 0) break;                                                               \
      if (((v0_ | v1_ | v2_ | v3_) & 2) != 0) {                                                                  \
        if (threadIdx.x == 0) { if (((v0_ | v1_ | v2_ | v3_) & 4) != 0) stf(abort_flag, 2); *s_abort = 1; }      \
        __syncthreads();                                                                                         \
        return;                                                                                                  \
      }                                                                                                          \
      __syncthreads();                                                                                           \
      __builtin_amdgcn_s_sleep(1);                                                                               \
      MAIL_ISSUE(MAIL, PD, PL, KB);                                                                              \
    }                                                                                                            \
  } while (0)


// grid = persistent (one workgroup per CU), 256 threads
//
// Scheduling.  One ticket stream over all slots (ticket -> slot = ticket % n_frames, task = ticket / n_frames, the
// frames of a batch advance side by side).  A workgroup takes a ticket, waits until the task can START (stage 0 of
// its dependency list), and inside the task waits again for the last operand column / the diagonal factor ("start
// early, wait late": the earlier columns are accumulated meanwhile).  A task only ever waits for smaller tickets
// and a ticket is only held by a running workgroup, so the smallest unfinished ticket can always finish.
// This is a LATENCY scheduler (one or two frames per launch: the drop-in case): workgroups that hold tickets of
// the narrow top of the tree idle until their turn, and every operand tile is fetched from L2 / MALL by the task
// that needs it.  Larger batches are bandwidth- and occupancy-bound and run the per-level launches of
// slm_front.hip (XCD-aware work lists, operands shared through L2) -- slm_api.hip picks.
// ---- the tasks.  Each kind is a function of its own (a real call): inlined into one loop, every path pays the
// register demand of all of them, and the tile factorisation -- the critical path -- ends up spilling.
// A task re-derives its descriptors from (ticket) itself; LDS regions come from the dynamic LDS base.
// (DIAG: the POTRF tasks and the COL tasks are instantiations of their own -- 85 KB of code as one function, more than the
//  instruction cache two CUs share; the pivot chain runs through the smaller one)
template <bool DIAG>
__device__ __noinline__ void dag_task_factor(const FrameDev* __restrict__ frames, int n_frames, int slot, int ti, double u_override, int cut, int mode) {
  double* lds = dag_lds;
  double* S = lds;                 // tile being factored / B operand staging
  double* M = lds + TILE;          // inverse of the factored tile / second staging tile
  double* dinv = lds + 2 * TILE;   // 4 x 256
  double* wt = dinv + 4 * 256;     // 3 x 256 scratch
  double* vec = wt + 3 * 256;      // NB
  double* yv = vec + NB;           // NB
  double* part = wt;               // 4 * NB row partials (not live during a tile factorisation)
  int* s_ok = reinterpret_cast<int*>(yv + NB);
  int* s_task = s_ok + 1;
  int* s_abort = s_ok + 2;
  int* s_cnt = s_ok + 3;
  int2* maps = reinterpret_cast<int2*>(dinv);   // 256 int2: pull maps of the tile being loaded (not live in a factorisation)
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
  int* abort_flag = unip(frames[0].dag_flags) + 1;
  (void)s_task; (void)S; (void)M; (void)dinv; (void)vec; (void)yv; (void)part; (void)maps; (void)s_cnt; (void)l; (void)w; (void)lr; (void)lk;
    const FrameDev& fdr = frames[slot];
    const int32_t* tasks = unip(cut >= 0 ? fdr.dag_top_tasks : fdr.dag_tasks);
    const int w0 = uni(tasks[2 * ti]), w1 = uni(tasks[2 * ti + 1]);
    constexpr int type = DIAG ? ND_T_POTRF : ND_T_COL;   // (= w0 >> 24: the caller dispatched on it)
    const int fi = w0 & 0xFFFFFF, tr_ = w1 >> 8, ts_ = w1 & 255;
    SS fd;
    fd.cached_ops = (mode & 1) != 0;
    fd.ftiles = unip(fdr.ftiles); fd.fvec = unip(fdr.fvec); fd.flinv = unip(fdr.flinv); fd.fmail = unip(fdr.fmail); fd.delta = unip(fdr.delta);
    fd.nd_nodes = unip(fdr.nd_nodes); fd.front_kids = unip(fdr.front_kids); fd.pull_off = unip(fdr.pull_off);
    fd.pullmap = unip(fdr.pullmap); fd.prng_off = unip(fdr.prng_off); fd.prng = unip(fdr.prng);
    fd.fronts = unip(fdr.fronts); fd.n_fronts = uni(fdr.n_fronts);
    const FS f = front_snapshot(fd.fronts[fi]);
    const DagFlags g = dag_flags_of(fdr);
    const TD d = task_deps(fd, f, fi, type, tr_, ts_, cut >= 0 && uni(fd.fronts[fi].depth) == cut);
    double* vecs = fd.fvec + f.vec_off;
    LMState* lmst = unip(fdr.st);
    long long* trc_base = unip(fdr.dag_trace);
    long long* trc = trc_base ? trc_base + 24 * (size_t)ti : nullptr;
    if (trc && threadIdx.x == 0) {
      trace_put(trc, 0, wall_clock64());
      trace_put(trc, 3, blockIdx.x);
    }
#define DAG_READY() do { if (trc && threadIdx.x == 0) trace_put(trc, 1, wall_clock64()); } while (0)
#define DAG_END() do { if (trc && threadIdx.x == 0) trace_put(trc, 2, wall_clock64()); } while (0)
#define DAG_MARK(k) do { if (trc && threadIdx.x == 0) trace_put(trc, (k), wall_clock64()); } while (0)
    // The task's own tile and vector rows are final before the launch (assembly; in the hybrid form also the pushes of
    // the per-level launches): requested BEFORE the wait for what the task needs to start, not after it
    double4_t acc[4];
    load_c_frags1(tile_ptr(fd, f, tr_, ts_), acc);
    // a PURE-FILL tile (FrameDev::tile_kind: nothing assembled into it, never zeroed) starts from zero: what the load
    // returned is whatever the previous iteration left there
    fd.tile_kind = unip(fdr.tile_kind);
    if (uni((int)fd.tile_kind[tile_index(f, tr_, ts_)]) != 0) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[ni] = double4_t{0.0, 0.0, 0.0, 0.0};
    }
    double bvec = 0.0, tsum = 0.0;
    if (type == ND_T_POTRF && threadIdx.x < NB) bvec = ld1(vecs + (size_t)ts_ * NB + threadIdx.x);
    dag_pull_maps(fd, fi, tr_, ts_, d.np, maps);   // (static plan data: also before the wait)
    // stage 0: what the task needs to start
    if (!dag_wait_deps(d, f, g, d.nskip, d.n0, abort_flag, s_abort)) return;

  {
      // ================= POTRF(f,s) / COL(f,r,s) ==================================================
      const int r = tr_, s = ts_;
      constexpr bool diag = DIAG;
      const double u = (u_override >= 0.0 || !diag) ? u_override : lmst->u;   // requested now, used after the updates
      // (stage 0: the children's update tiles and the operand columns c < s-1)
      // the children's contributions to this tile (and to the vector rows of a diagonal tile)
      if (diag) dag_pull<true>(fd, fi, r, s, acc, bvec, d.np, maps, true);
      else dag_pull<false>(fd, fi, r, s, acc, bvec, d.np, maps, true);
      double4_t xl[4];   // POTRF(s > 0): L(s, s-1), rows of this wave (for its product with y_{s-1} after the factorisation)
      if (diag && s > 0) {
        // POTRF(s) owns the tile (s, s-1) too: L(s,s-1) = (A(s,s-1) - sum_{c<s-1} L(s,c) L(s-1,c)^T) L_{s-1,s-1}^-T is
        // formed here as soon as the factor of column s-1 is out, and goes into the update of (s,s) from LDS
        double4_t accl[4];
        load_c_frags1(tile_ptr(fd, f, s, s - 1), accl);
        if (uni((int)fd.tile_kind[tile_index(f, s, s - 1)]) != 0) {   // (pure fill: see above)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) accl[ni] = double4_t{0.0, 0.0, 0.0, 0.0};
        }
        double dummy = 0.0;
        dag_pull<false>(fd, fi, s, s - 1, accl, dummy, d.np2, maps);
        {
          // columns c < s-2: both tiles at once, as the columns come.  The last column, c = s-2, is split: its L(s-1,c)
          // is the tile POTRF(s-1) itself produces (out only when that task starts factoring), so everything that
          // does not need it -- the update of (s,s) and the vector sum -- is done before, and only one tile load and
          // one product remain between its arrival and the streamed rounds.
          const int clast = s - 2;
          int c = 0, m = 1;
          while (c < clast && m > 0) {
            m = dag_wait_prefix(d, f, g, d.n0 + 3 * c, d.n0 + 3 * clast, 3, abort_flag, s_abort, s_cnt);
            dag_accumulate2(fd, f, s, c, c + m, acc, accl, S, M, yv, tsum);
            c += m;
          }
          if (m == 0) return;
          if (clast >= 0) {
            if (!dag_wait_deps(d, f, g, d.n0 + 3 * clast, d.n0 + 3 * clast + 2, abort_flag, s_abort)) return;   // L(s,c), y_c
            dag_accumulate<true>(fd, f, s, s, clast, clast + 1, acc, S, M, yv, tsum);       // leaves L(s,c) in S
            if (!dag_wait_deps(d, f, g, d.n0 + 3 * clast + 2, d.n0 + 3 * clast + 3, abort_flag, s_abort)) return;   // L(s-1,c)
            double areg[16];
            load_tile_regs2(tile_ptr(fd, f, s - 1, clast), areg, fd.cached_ops);
            store_tile_lds2(M, areg);
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
              const double a = -S[(16 * w + lr) + (4 * ks + lk) * LD];
#pragma unroll
              for (int ni = 0; ni < 4; ++ni)
                accl[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(M[(16 * ni + lr) + (4 * ks + lk) * LD], a, accl[ni], 0, 0, 0);
            }
          }
        }
        // ---- streamed row solve against column s-1 and update of (s,s), 16 pivots at a time: the producer
        // (POTRF(s-1), still inside its factorisation) publishes every finished diagonal-block inverse and row
        // block of L; this task trails it by one block, so that (s,s) is fully updated shortly after the
        // producer's last pivot -- not a whole inverse assembly + publish + reload + 64x64 products later.
        {
          const double* mail = fd.fmail + (size_t)(f.pcol0 + s - 1) * SLM_MAIL_DOUBLES;
          double* Lst = M;     // row block kb of L(s-1,s-1), at its place in a 64 x 64 tile
          double* Xs = S;      // X(:, 16 kb .. 16 kb + 15) of all 64 rows, column-major ld 64
          // Software pipeline over the four rounds (a consumer that has fallen behind the producer catches up instead of
          // ending 4-5 us after the producer's last pivot): see MAIL_ISSUE / MAIL_TAKE
          double pd[4] = {0.0, 0.0, 0.0, 0.0}, pl[4][3];   // per round: Dinv element, L blocks t = 0..kb-1
          MAIL_ISSUE(mail, pd, pl, 0);
          __syncthreads();   // dinv doubles as the pull maps (dag_pull): every wave is done with them before round 0 lands there
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            MAIL_TAKE(mail, pd, pl, kb, dinv, Lst);
            if (kb == 3) DAG_READY();
            if (kb < 3) MAIL_ISSUE(mail, pd, pl, kb + 1);
            double4_t t4 = accl[kb];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              if (t >= kb) break;
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) {
                const double y = Lst[(16 * kb + lr) + (16 * t + 4 * ks + lk) * LD];
                t4 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, -accl[t][ks], t4, 0, 0, 0);
              }
            }
            double4_t x4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const double y = dinv[kb * 256 + lr + 16 * (4 * ks + lk)];   // Y[p][n] = Dinv[n][p]
              x4 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, t4[ks], x4, 0, 0, 0);
            }
            accl[kb] = x4;
            if (kb < 3) {
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) Xs[(16 * w + lr) + (lk + 4 * rr) * LD] = x4[rr];
              __syncthreads();
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) {
                const double a = -Xs[(16 * w + lr) + (4 * ks + lk) * LD];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                  acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[(16 * ni + lr) + (4 * ks + lk) * LD], a, acc[ni], 0, 0, 0);
              }
            } else {
              // LAST round, look-ahead layout: X(:, 48..63) goes to blocks of M that hold nothing any more (the row blocks of
              // L(s-1,s-1) of the earlier rounds: (0,0) (1,0) (2,0) for rows 0..47, (1,1) for rows 48..63) instead of S, so
              // that nothing below waits for the slowest reader of this strip: wave 0 -- the pivot chain -- updates ITS
              // diagonal block only and leaves for the factorisation (dag_factor_tile(..., look_ahead)), the others finish
              // the tile and write rows 16..63 of S behind it.
              double* xrow = (w < 3) ? M + 16 * w : M + 16 + 16 * LD;      // this wave's 16 rows of the strip
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) xrow[lr + (lk + 4 * rr) * LD] = x4[rr];
              __syncthreads();
              if (w == 0) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                  const double x0 = M[lr + (4 * ks + lk) * LD];
                  acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, -x0, acc[0], 0, 0, 0);
                }
              } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                  const double a = -xrow[lr + (4 * ks + lk) * LD];
#pragma unroll
                  for (int ni = 0; ni < 4; ++ni) {
                    const double b = (ni < 3) ? M[(16 * ni + lr) + (4 * ks + lk) * LD] : M[(16 + lr) + (16 + 4 * ks + lk) * LD];
                    acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc[ni], 0, 0, 0);
                  }
                }
              }
            }
          }
        }
        store_c_frags1(tile_ptr(fd, f, s, s - 1), accl);         // L(s, s-1) for the other tasks; published below
        xl[0] = accl[0]; xl[1] = accl[1]; xl[2] = accl[2]; xl[3] = accl[3];
      } else if (!diag) {
        int c = 0, m = 1;
        while (c < s && m > 0) {
          m = dag_wait_prefix(d, f, g, d.n0 + 2 * c, d.n0 + 2 * s, 2, abort_flag, s_abort, s_cnt);
          dag_accumulate<false>(fd, f, r, s, c, c + m, acc, S, M, yv, tsum);
          c += m;
        }
        if (m == 0) return;
      }
      if (diag) {
        DAG_MARK(4);
        const bool la = s > 0;                             // look-ahead entry of the factorisation (see the last streamed round)
        double t;
        if (!la) {
          t = dag_reduce_rows(tsum, part);                 // (sum_{c<s-1} L(s,c) y_c)[row threadIdx.x & 63]
        } else {
          part[(threadIdx.x >> 6) * NB + (threadIdx.x & 63)] = tsum;   // summed behind the first 16 pivots (dag_factor_tile)
          t = 0.0;
        }
        // updated tile -> S: lower triangle, damping on real pivots, identity on the padding rows
        if (!la || w > 0) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int i = 16 * w + lr, k = 16 * ni + lk + 4 * rr;
              double x = (i >= k) ? acc[ni][rr] : 0.0;
              if (i == k) x = (s * NB + i < f.n1) ? x + u : 1.0;
              S[i + k * LD] = x;
            }
        } else {
          // wave 0: its diagonal block on its own (dinv + 256, ld 16: the mailbox copies there are spent)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int i = lr, k = lk + 4 * rr;
            double x = (i >= k) ? acc[0][rr] : 0.0;
            if (i == k) x = (s * NB + i < f.n1) ? x + u : 1.0;
            dinv[256 + i + 16 * k] = x;
          }
          wave_sync();
        }
        if (!la) __syncthreads();
        DAG_MARK(5);
        const bool stream = s + 1 < f.nt;                  // POTRF(s+1) and the column's COL tasks follow the factorisation 16 pivots at a time
        double* linv = fd.flinv + f.linv_off + (size_t)s * TILE;
        const bool ok = dag_factor_tile(stream ? fd.fmail + (size_t)(f.pcol0 + s) * SLM_MAIL_DOUBLES : nullptr,
                                       s > 0 ? g.tile + tile_index(f, s, s - 1) : nullptr,    // L(s, s-1) goes out during the first 16 pivots
                                       trc ? trc + 8 : nullptr,
                                       min(4, (f.n1 - s * NB + 15) >> 4),   // 16-pivot blocks with real pivots (the front's last column: fewer)
                                       la, &t);
        const double bt = bvec - t;                        // threads < NB: right-hand side of row threadIdx.x (minus L(s,s-1) y_{s-1}, below)
        DAG_MARK(6);
        if (!ok && threadIdx.x == 0) lmst->chol_fail = 1;
#pragma unroll
        for (int e = 0; e < 16; ++e) st1(linv + threadIdx.x + 256 * e, M[threadIdx.x + 256 * e]);
        dag_publish_begin();
        dag_set_flag(g.tile + tile_index(f, s, s));        // the whole inverse is out: the column's other row solves can start
        DAG_MARK(7);
        {
          // y_s = L_ss^-1 (b_s - sum_{c<s} L(s,c) y_c), off the factorisation's critical path, under its own flag
          double xy = 0.0;
          if (s > 0) {
            if (!dag_wait_deps(d, f, g, d.n - 1, d.n, abort_flag, s_abort)) return;   // y_{s-1}
            if (threadIdx.x < NB) yv[threadIdx.x] = ld1(vecs + (size_t)(s - 1) * NB + threadIdx.x);
            __syncthreads();
            double p = 0.0;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) p += xl[kb][rr] * yv[16 * kb + lk + 4 * rr];
            p += __shfl_xor(p, 16, 64);
            p += __shfl_xor(p, 32, 64);
            if (lk == 0) part[16 * w + lr] = p;            // (L(s,s-1) y_{s-1})[row 16 w + lr]
            __syncthreads();
            if (threadIdx.x < NB) xy = part[threadIdx.x];
            __syncthreads();
          }
          if (threadIdx.x < NB) vec[threadIdx.x] = bt - xy;
          __syncthreads();
          const int i = threadIdx.x & 63, q = threadIdx.x >> 6;
          double a = 0.0;
#pragma unroll
          for (int k = 16 * q; k < 16 * q + 16; ++k) a += M[i + k * LD] * vec[k];
          part[q * NB + i] = a;
          __syncthreads();
          if (threadIdx.x < NB) st1(vecs + (size_t)s * NB + i, part[i] + part[NB + i] + part[2 * NB + i] + part[3 * NB + i]);
        }
        dag_publish_begin();
        dag_set_flag(g.py + f.pcol0 + s);
        if (s == 0) DAG_READY();
        DAG_END();
      } else {
        // stage 2: X = acc L_ss^-T, 16 pivots at a time behind the factorisation of (s,s) -- the same streamed row solve
        // as POTRF(s+1) runs on its tile (s+1,s): the row is complete ~3 us after the producer's last pivot instead of
        // after the whole inverse has been assembled, published and re-read (the boundary rows of a front's LAST pivot
        // column are what its Schur complement, and with it the parent front, waits for).
        {
          const double* mail = fd.fmail + (size_t)(f.pcol0 + s) * SLM_MAIL_DOUBLES;
          double* Lst = M;     // row block kb of L(s,s), at its place in a 64 x 64 tile
          double pd[4] = {0.0, 0.0, 0.0, 0.0}, pl[4][3];
          MAIL_ISSUE(mail, pd, pl, 0);
          __syncthreads();   // dinv doubles as the pull maps (dag_pull): every wave is done with them before round 0 lands there
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            MAIL_TAKE(mail, pd, pl, kb, dinv, Lst);
            if (kb == 3) DAG_READY();
            if (kb < 3) MAIL_ISSUE(mail, pd, pl, kb + 1);
            double4_t t4 = acc[kb];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              if (t >= kb) break;
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) {
                const double y = Lst[(16 * kb + lr) + (16 * t + 4 * ks + lk) * LD];
                t4 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, -acc[t][ks], t4, 0, 0, 0);
              }
            }
            double4_t x4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const double y = dinv[kb * 256 + lr + 16 * (4 * ks + lk)];   // Y[p][n] = Dinv[n][p]
              x4 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, t4[ks], x4, 0, 0, 0);
            }
            acc[kb] = x4;
            __syncthreads();   // the next round rewrites what this one read
          }
        }
        double4_t* xa = acc;
        store_c_frags1(tile_ptr(fd, f, r, s), xa);
        dag_publish_begin();
        dag_set_flag(g.tile + tile_index(f, r, s));
        DAG_END();
      }
  }
}

__device__ __noinline__ void dag_task_schur(const FrameDev* __restrict__ frames, int n_frames, int slot, int ti, double u_override, int cut, int mode) {
  double* lds = dag_lds;
  double* S = lds;                 // tile being factored / B operand staging
  double* M = lds + TILE;          // inverse of the factored tile / second staging tile
  double* dinv = lds + 2 * TILE;   // 4 x 256
  double* wt = dinv + 4 * 256;     // 3 x 256 scratch
  double* vec = wt + 3 * 256;      // NB
  double* yv = vec + NB;           // NB
  double* part = wt;               // 4 * NB row partials (not live during a tile factorisation)
  int* s_ok = reinterpret_cast<int*>(yv + NB);
  int* s_task = s_ok + 1;
  int* s_abort = s_ok + 2;
  int* s_cnt = s_ok + 3;
  int2* maps = reinterpret_cast<int2*>(dinv);   // 256 int2: pull maps of the tile being loaded (not live in a factorisation)
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
  int* abort_flag = unip(frames[0].dag_flags) + 1;
  (void)s_task; (void)S; (void)M; (void)dinv; (void)vec; (void)yv; (void)part; (void)maps; (void)s_cnt; (void)l; (void)w; (void)lr; (void)lk;
    const FrameDev& fdr = frames[slot];
    const int32_t* tasks = unip(cut >= 0 ? fdr.dag_top_tasks : fdr.dag_tasks);
    const int w0 = uni(tasks[2 * ti]), w1 = uni(tasks[2 * ti + 1]);
    const int type = w0 >> 24, fi = w0 & 0xFFFFFF, tr_ = w1 >> 8, ts_ = w1 & 255;
    SS fd;
    fd.cached_ops = (mode & 1) != 0;
    fd.ftiles = unip(fdr.ftiles); fd.fvec = unip(fdr.fvec); fd.flinv = unip(fdr.flinv); fd.fmail = unip(fdr.fmail); fd.delta = unip(fdr.delta);
    fd.nd_nodes = unip(fdr.nd_nodes); fd.front_kids = unip(fdr.front_kids); fd.pull_off = unip(fdr.pull_off);
    fd.pullmap = unip(fdr.pullmap); fd.prng_off = unip(fdr.prng_off); fd.prng = unip(fdr.prng);
    fd.fronts = unip(fdr.fronts); fd.n_fronts = uni(fdr.n_fronts);
    const FS f = front_snapshot(fd.fronts[fi]);
    const DagFlags g = dag_flags_of(fdr);
    const TD d = task_deps(fd, f, fi, type, tr_, ts_, cut >= 0 && uni(fd.fronts[fi].depth) == cut);
    double* vecs = fd.fvec + f.vec_off;
    LMState* lmst = unip(fdr.st);
    long long* trc_base = unip(fdr.dag_trace);
    long long* trc = trc_base ? trc_base + 24 * (size_t)ti : nullptr;
    if (trc && threadIdx.x == 0) {
      trace_put(trc, 0, wall_clock64());
      trace_put(trc, 3, blockIdx.x);
    }
#define DAG_READY() do { if (trc && threadIdx.x == 0) trace_put(trc, 1, wall_clock64()); } while (0)
#define DAG_END() do { if (trc && threadIdx.x == 0) trace_put(trc, 2, wall_clock64()); } while (0)
#define DAG_MARK(k) do { if (trc && threadIdx.x == 0) trace_put(trc, (k), wall_clock64()); } while (0)
    // own tile, vector rows and pull maps: final before the launch, requested before the wait (see dag_task_factor)
    double4_t acc[4];
    // A boundary block holds nothing before its Schur task writes it (never read first, never zeroed: k_iter_begin_nd)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[ni] = double4_t{0.0, 0.0, 0.0, 0.0};
    double bvec = 0.0, tsum = 0.0;
    if (tr_ == ts_ && threadIdx.x < NB) bvec = ld1(vecs + (size_t)tr_ * NB + threadIdx.x);
    dag_pull_maps(fd, fi, tr_, ts_, d.np, maps);
    // stage 0: what the task needs to start
    if (!dag_wait_deps(d, f, g, d.nskip, d.n0, abort_flag, s_abort)) return;

  {
      // ================= SCHUR(f,r,s): update tile of the boundary block, stored in place =============
      const int r = tr_, sc = ts_;
      const bool dg = r == sc;
      if (dg) dag_pull<true>(fd, fi, r, sc, acc, bvec, d.np, maps, true);
      else dag_pull<false>(fd, fi, r, sc, acc, bvec, d.np, maps, true);
      {
        int c = 0, m = 1;
        while (c < f.npt && m > 0) {
          m = dag_wait_prefix(d, f, g, d.n0 + 2 * c, d.n0 + 2 * f.npt, 2, abort_flag, s_abort, s_cnt);
          if (c + m == f.npt) DAG_READY();
          if (dg) dag_accumulate<true>(fd, f, r, r, c, c + m, acc, S, M, yv, tsum);
          else dag_accumulate<false>(fd, f, r, sc, c, c + m, acc, S, M, yv, tsum);
          c += m;
        }
        if (m == 0) return;
      }
      // the update tile replaces the assembled one (the parent's tasks gather from it); diagonal tiles carry
      // the vector rows v_r = b_r - sum_c L(r,c) y_c
      store_c_frags1(tile_ptr(fd, f, r, sc), acc);
      if (dg) {
        const double tvec = dag_reduce_rows(tsum, part);
        if (threadIdx.x < NB) st1(vecs + (size_t)r * NB + threadIdx.x, bvec - tvec);
      }
      dag_publish_begin();
      dag_set_flag(g.tile + tile_index(f, r, sc));
      DAG_END();
  }
}

__device__ __noinline__ void dag_task_backb(const FrameDev* __restrict__ frames, int n_frames, int slot, int ti, double u_override, int cut, int mode) {
  double* lds = dag_lds;
  double* S = lds;                 // tile being factored / B operand staging
  double* M = lds + TILE;          // inverse of the factored tile / second staging tile
  double* dinv = lds + 2 * TILE;   // 4 x 256
  double* wt = dinv + 4 * 256;     // 3 x 256 scratch
  double* vec = wt + 3 * 256;      // NB
  double* yv = vec + NB;           // NB
  double* part = wt;               // 4 * NB row partials (not live during a tile factorisation)
  int* s_ok = reinterpret_cast<int*>(yv + NB);
  int* s_task = s_ok + 1;
  int* s_abort = s_ok + 2;
  int* s_cnt = s_ok + 3;
  int2* maps = reinterpret_cast<int2*>(dinv);   // 256 int2: pull maps of the tile being loaded (not live in a factorisation)
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
  int* abort_flag = unip(frames[0].dag_flags) + 1;
  (void)s_task; (void)S; (void)M; (void)dinv; (void)vec; (void)yv; (void)part; (void)maps; (void)s_cnt; (void)l; (void)w; (void)lr; (void)lk;
    const FrameDev& fdr = frames[slot];
    const int32_t* tasks = unip(cut >= 0 ? fdr.dag_top_tasks : fdr.dag_tasks);
    const int w0 = uni(tasks[2 * ti]), w1 = uni(tasks[2 * ti + 1]);
    const int type = w0 >> 24, fi = w0 & 0xFFFFFF, tr_ = w1 >> 8, ts_ = w1 & 255;
    SS fd;
    fd.cached_ops = (mode & 1) != 0;
    fd.ftiles = unip(fdr.ftiles); fd.fvec = unip(fdr.fvec); fd.flinv = unip(fdr.flinv); fd.fmail = unip(fdr.fmail); fd.delta = unip(fdr.delta);
    fd.nd_nodes = unip(fdr.nd_nodes); fd.front_kids = unip(fdr.front_kids); fd.pull_off = unip(fdr.pull_off);
    fd.pullmap = unip(fdr.pullmap); fd.prng_off = unip(fdr.prng_off); fd.prng = unip(fdr.prng);
    fd.fronts = unip(fdr.fronts); fd.n_fronts = uni(fdr.n_fronts);
    const FS f = front_snapshot(fd.fronts[fi]);
    const DagFlags g = dag_flags_of(fdr);
    const TD d = task_deps(fd, f, fi, type, tr_, ts_, cut >= 0 && uni(fd.fronts[fi].depth) == cut);
    double* vecs = fd.fvec + f.vec_off;
    LMState* lmst = unip(fdr.st);
    long long* trc_base = unip(fdr.dag_trace);
    long long* trc = trc_base ? trc_base + 24 * (size_t)ti : nullptr;
    if (trc && threadIdx.x == 0) {
      trace_put(trc, 0, wall_clock64());
      trace_put(trc, 3, blockIdx.x);
    }
#define DAG_READY() do { if (trc && threadIdx.x == 0) trace_put(trc, 1, wall_clock64()); } while (0)
#define DAG_END() do { if (trc && threadIdx.x == 0) trace_put(trc, 2, wall_clock64()); } while (0)
#define DAG_MARK(k) do { if (trc && threadIdx.x == 0) trace_put(trc, (k), wall_clock64()); } while (0)
    // The boundary tiles of this column are final once the front is factored -- long before the parent's solution
    // arrives: wait for THEM first and request the first two, so that the load latency is gone when x comes
    const int c_ = ts_;
    // (hybrid solve: a front below the cut was factored by the per-level launches before this kernel -- no flags to wait for)
    const bool prefactored = cut >= 0 && uni(fd.fronts[fi].depth) > cut;
    if (!prefactored &&
        !dag_wait(f.nt - f.npt, [&](int i) { return (const int*)(g.tile + tile_index(f, f.npt + i, c_)); }, 1, abort_flag, s_abort)) return;
    // Tiles in the 16-byte-per-lane layout (load_tile_regs2): thread t holds T[n][m], n = (t >> 5) + 8 e, m = 2 (t & 31) + {0, 1}
    // -- out[n] = sum_m T[n][m] x[m] needs a reduction over the 32 lanes t & 31 only.  Requests are UNCONDITIONAL (rows past
    // the end repeat the last one, multiplied by 0) so that the compiler's wait counts stay exact.
    const int rlast = f.nt - 1;
    auto brow = [&](int r) { return min(r, rlast); };
    double l0[16], l1[16], l2[16], l3[16];
    load_tile_regs2(tile_ptr(fd, f, brow(f.npt), c_), l0);
    load_tile_regs2(tile_ptr(fd, f, brow(f.npt + 1), c_), l1);
    // stage 0: what the task needs to start
    if (!dag_wait_deps(d, f, g, d.nskip, d.n0, abort_flag, s_abort)) return;

  {
      // ================= BACKB(f,c): y_c -= sum over boundary tiles L(r,c)^T x_r ====================
      // (the parent's pivots -- and with them all ancestors' -- are solved: stage 0)
      const int c = ts_;
      DAG_READY();
      double* xb = S;   // n2p doubles
      const int* nodes = fd.nd_nodes + f.nodes_off + f.nv;
      for (int i = threadIdx.x; i < f.n2p; i += blockDim.x) xb[i] = (i < 7 * f.nb) ? ld1(fd.delta + 7 * nodes[i / 7] + i % 7) : 0.0;
      __syncthreads();
      // two tiles being used while the next two are in flight; the partial products of all tiles are summed per thread
      // first, one reduction over the lanes at the end
      double acc8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc8[e] = 0.0;
      const int m2 = 2 * (l & 31);
#define BACKB_USE(LV, R)                                                          \
  do {                                                                            \
    const bool on_ = (R) < f.nt;                                                  \
    const double x0_ = on_ ? xb[(size_t)(brow(R) - f.npt) * NB + m2] : 0.0;       \
    const double x1_ = on_ ? xb[(size_t)(brow(R) - f.npt) * NB + m2 + 1] : 0.0;   \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) acc8[e] = fma(LV[2 * e + 1], x1_, fma(LV[2 * e], x0_, acc8[e])); \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(acc8[e]) :: "memory");   \
  } while (0)
      for (int r = f.npt; r < f.nt; r += 4) {
        load_tile_regs2(tile_ptr(fd, f, brow(r + 2), c), l2);
        load_tile_regs2(tile_ptr(fd, f, brow(r + 3), c), l3);
        BACKB_USE(l0, r);
        BACKB_USE(l1, r + 1);
        load_tile_regs2(tile_ptr(fd, f, brow(r + 4), c), l0);
        load_tile_regs2(tile_ptr(fd, f, brow(r + 5), c), l1);
        BACKB_USE(l2, r + 2);
        BACKB_USE(l3, r + 3);
      }
#undef BACKB_USE
      const double a = col_reduce8(acc8);
      if ((l & 3) == 0) {
        double* pv = vecs + (size_t)c * NB + 2 * w + (l >> 5) + 8 * ((l >> 2) & 7);
        st1(pv, ld1(pv) - a);
      }
      dag_publish_begin();
      dag_set_flag(g.pb + f.pcol0 + c);
      DAG_END();
  }
}

__device__ __noinline__ void dag_task_back(const FrameDev* __restrict__ frames, int n_frames, int slot, int ti, double u_override, int cut, int mode) {
  double* lds = dag_lds;
  double* S = lds;                 // tile being factored / B operand staging
  double* M = lds + TILE;          // inverse of the factored tile / second staging tile
  double* dinv = lds + 2 * TILE;   // 4 x 256
  double* wt = dinv + 4 * 256;     // 3 x 256 scratch
  double* vec = wt + 3 * 256;      // NB
  double* yv = vec + NB;           // NB
  double* part = wt;               // 4 * NB row partials (not live during a tile factorisation)
  int* s_ok = reinterpret_cast<int*>(yv + NB);
  int* s_task = s_ok + 1;
  int* s_abort = s_ok + 2;
  int* s_cnt = s_ok + 3;
  int2* maps = reinterpret_cast<int2*>(dinv);   // 256 int2: pull maps of the tile being loaded (not live in a factorisation)
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
  int* abort_flag = unip(frames[0].dag_flags) + 1;
  (void)s_task; (void)S; (void)M; (void)dinv; (void)vec; (void)yv; (void)part; (void)maps; (void)s_cnt; (void)l; (void)w; (void)lr; (void)lk;
    const FrameDev& fdr = frames[slot];
    const int32_t* tasks = unip(cut >= 0 ? fdr.dag_top_tasks : fdr.dag_tasks);
    const int w0 = uni(tasks[2 * ti]), w1 = uni(tasks[2 * ti + 1]);
    const int type = w0 >> 24, fi = w0 & 0xFFFFFF, tr_ = w1 >> 8, ts_ = w1 & 255;
    SS fd;
    fd.cached_ops = (mode & 1) != 0;
    fd.ftiles = unip(fdr.ftiles); fd.fvec = unip(fdr.fvec); fd.flinv = unip(fdr.flinv); fd.fmail = unip(fdr.fmail); fd.delta = unip(fdr.delta);
    fd.nd_nodes = unip(fdr.nd_nodes); fd.front_kids = unip(fdr.front_kids); fd.pull_off = unip(fdr.pull_off);
    fd.pullmap = unip(fdr.pullmap); fd.prng_off = unip(fdr.prng_off); fd.prng = unip(fdr.prng);
    fd.fronts = unip(fdr.fronts); fd.n_fronts = uni(fdr.n_fronts);
    const FS f = front_snapshot(fd.fronts[fi]);
    const DagFlags g = dag_flags_of(fdr);
    const TD d = task_deps(fd, f, fi, type, tr_, ts_, cut >= 0 && uni(fd.fronts[fi].depth) == cut);
    double* vecs = fd.fvec + f.vec_off;
    LMState* lmst = unip(fdr.st);
    long long* trc_base = unip(fdr.dag_trace);
    long long* trc = trc_base ? trc_base + 24 * (size_t)ti : nullptr;
    if (trc && threadIdx.x == 0) {
      trace_put(trc, 0, wall_clock64());
      trace_put(trc, 3, blockIdx.x);
    }
#define DAG_READY() do { if (trc && threadIdx.x == 0) trace_put(trc, 1, wall_clock64()); } while (0)
#define DAG_END() do { if (trc && threadIdx.x == 0) trace_put(trc, 2, wall_clock64()); } while (0)
#define DAG_MARK(k) do { if (trc && threadIdx.x == 0) trace_put(trc, (k), wall_clock64()); } while (0)
    // stage 0 in two steps: the factorisation's outputs (y of every column, the tiles of the pivot block) are there long
    // before the boundary part of the right-hand side (BACKB, which waits for the parent's solution): wait for the
    // former, request the first three tiles of the chain, and only then wait for the latter
    // FUSED form (word 1 = 1, fronts with few pivot tile columns): no BACKB tasks -- this task waits for the parent's
    // solution itself and subtracts the boundary part before the chain
    const bool fusedf = ts_ == 1 && f.nb > 0;
    const int nb_flags = f.nb > 0 ? (fusedf ? 1 : f.npt) : 0;
    const bool prefactored = cut >= 0 && uni(fd.fronts[fi].depth) > cut;   // (hybrid solve: factored by the per-level launches)
    if (!prefactored) {
      if (!dag_wait_deps(d, f, g, 0, f.npt, abort_flag, s_abort)) return;
      if (!dag_wait_deps(d, f, g, f.npt + nb_flags, d.n0, abort_flag, s_abort)) return;
    }
    // LEFT-looking order, column by column from the last: op k of column c is the tile L(npt-1-j, c), j = 0 .. npt-2-c
    // (its product with the known x_r goes into per-thread partial sums -- no lane reduction per tile), then the
    // inverse of the diagonal factor.  One workgroup runs the whole chain, so the order costs no latency, and a column
    // needs two butterflies instead of one per tile.
    const int nops = f.npt * (f.npt + 1) / 2;
    // (ops past the end repeat the last one's address: every load below is UNCONDITIONAL, so the compiler can count the
    // loads in flight and wait for exactly the tile it needs -- behind a branch it falls back to "wait for all")
    // An op is (c, j); `lc, lj` walk three ops ahead of `oc, oj` (both stop at the last op).
    // (branch-free on purpose: with branches between an op's multiply-adds and the next request, the compiler sinks
    // the multiply-adds below the request, needs a second register set for the new tile and copies it over at the
    // loop end -- behind a wait for ALL loads in flight, which serialises the whole chain on the memory latency)
    auto op_next = [&](int& c, int& j) {
      const bool last = j == f.npt - 1 - c, adv = last && c > 0;
      j = last ? (adv ? 0 : j) : j + 1;
      c = adv ? c - 1 : c;
    };
    auto op_addr = [&](int c, int j) -> const double* {
      const int r = f.npt - 1 - j;
      const size_t t = (size_t)c * f.nt - (size_t)c * (c - 1) / 2 + (size_t)(r - c);   // (tile_ptr, pivot columns)
      const double* tile = fd.ftiles + f.tile_off + t * TILE;
      const double* inv = fd.flinv + f.linv_off + (size_t)c * TILE;
      return j == f.npt - 1 - c ? inv : tile;
    };
    int oc = f.npt - 1, oj = 0, lc = oc, lj = 0;
    double l0[16], l1[16], l2[16];
#define BACK_LOAD(LV)                                                                     \
  do {                                                                                    \
    load_tile_regs2(op_addr(lc, lj), LV);                                                 \
    op_next(lc, lj);                                                                      \
  } while (0)
    if (!fusedf) {
      BACK_LOAD(l0);
      BACK_LOAD(l1);
      BACK_LOAD(l2);
    }
    if (!dag_wait_deps(d, f, g, f.npt, f.npt + nb_flags, abort_flag, s_abort)) return;

  {
      // ================= BACK(f): the chain over the front's pivot columns =========================
      // for c = npt-1 .. 0:  x_c = L_cc^-T (y_c - sum_{r>c} L(r,c)^T x_r).  y / x stay in LDS, the tiles (in the order
      // they are used) stream through registers three ahead: their addresses do not depend on x.
      DAG_READY();
      double* ya = S;                  // npt * NB: y, overwritten by x column by column
      for (int i = threadIdx.x; i < f.npt * NB; i += blockDim.x) ya[i] = ld1(vecs + i);
      // tiles in the 16-byte-per-lane layout (load_tile_regs2): thread t holds T[n][m], n = (t >> 5) + 8 e, m = 2 (t & 31) + {0, 1};
      // out[n] = sum_m T[n][m] v[m]: 8 partial sums per thread, reduced over the 32 lanes t & 31 once per column
      double acc8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc8[e] = 0.0;
      const int m2 = 2 * (l & 31), n_ = 2 * w + (l >> 5) + 8 * ((l >> 2) & 7);
      if (fusedf) {
        // ---- the boundary part (what the BACKB tasks do for the fronts that have them): y_c -= sum_r L(r,c)^T x_r over the
        // boundary tile rows, x of the boundary nodes gathered ONCE from the global solution (all ancestors are solved)
        double* xb = M;   // n2p doubles
        const int* nodes = fd.nd_nodes + f.nodes_off + f.nv;
        for (int i = threadIdx.x; i < f.n2p; i += blockDim.x) xb[i] = (i < 7 * f.nb) ? ld1(fd.delta + 7 * nodes[i / 7] + i % 7) : 0.0;
        __syncthreads();   // xb and ya complete
        // the npt x (nt - npt) boundary tiles in column-major order, two in use while the next two are in flight (requests
        // past the end repeat the last tile and count for nothing: unconditional loads keep the wait counts exact)
        const int nbt = f.nt - f.npt, nbops = f.npt * nbt;
        auto b_addr = [&](int k) -> const double* {
          const int kk = min(k, nbops - 1), c = kk / nbt;
          return tile_ptr(fd, f, f.npt + (kk - c * nbt), c);
        };
        double t0[16], t1[16], t2[16], t3[16];
        load_tile_regs2(b_addr(0), t0);
        load_tile_regs2(b_addr(1), t1);
#define BNDRY_USE(LV, K)                                                                       \
  do {                                                                                         \
    const int k_ = (K);                                                                        \
    const bool on_ = k_ < nbops;                                                               \
    const int kk_ = min(k_, nbops - 1), c_ = kk_ / nbt, rb_ = kk_ - c_ * nbt;                   \
    const double x0_ = on_ ? xb[(size_t)rb_ * NB + m2] : 0.0, x1_ = on_ ? xb[(size_t)rb_ * NB + m2 + 1] : 0.0; \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) acc8[e] = fma(LV[2 * e + 1], x1_, fma(LV[2 * e], x0_, acc8[e])); \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(acc8[e]) :: "memory");   \
    if (on_ && rb_ == nbt - 1) {            /* the column's last tile: finish y_c */              \
      const double a_ = col_reduce8(acc8);                                                     \
      if ((l & 3) == 0) ya[(size_t)c_ * NB + n_] -= a_;                                        \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) acc8[e] = 0.0;                             \
    }                                                                                          \
  } while (0)
        for (int k = 0; k < nbops; k += 4) {
          load_tile_regs2(b_addr(k + 2), t2);
          load_tile_regs2(b_addr(k + 3), t3);
          BNDRY_USE(t0, k);
          BNDRY_USE(t1, k + 1);
          load_tile_regs2(b_addr(k + 4), t0);
          load_tile_regs2(b_addr(k + 5), t1);
          BNDRY_USE(t2, k + 2);
          BNDRY_USE(t3, k + 3);
        }
#undef BNDRY_USE
        BACK_LOAD(l0);
        BACK_LOAD(l1);
        BACK_LOAD(l2);
      }
#define BACK_OP(LV, K)                                                                    \
  do {                                                                                    \
    const int c_ = oc, j_ = oj;                                                           \
    op_next(oc, oj);                                                                      \
    const bool inv_ = (K) < nops && j_ == f.npt - 1 - c_;                                 \
    if (inv_ && c_ < f.npt - 1) {           /* the column's tiles are done: finish y_c */ \
      const double a_ = col_reduce8(acc8);                                                \
      if ((l & 3) == 0) ya[(size_t)c_ * NB + n_] -= a_;                                   \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) acc8[e] = 0.0;                        \
      __syncthreads();                                                                    \
    }                                                                                     \
    /* tile (r, c): x_r; the inverse: y_c; past the end: nothing */                       \
    const size_t xo_ = (size_t)(inv_ ? c_ : f.npt - 1 - j_) * NB + m2;                    \
    const double x0_ = (K) < nops ? ya[xo_] : 0.0, x1_ = (K) < nops ? ya[xo_ + 1] : 0.0;  \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) acc8[e] = fma(LV[2 * e + 1], x1_, fma(LV[2 * e], x0_, acc8[e])); \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(acc8[e]) :: "memory");   /* (done HERE) */ \
    BACK_LOAD(LV);                                                                        \
    if (inv_) {                                                                           \
      const double x_ = col_reduce8(acc8);                                                \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) acc8[e] = 0.0;                        \
      __syncthreads();                      /* everybody has read y_c */                  \
      if ((l & 3) == 0) ya[(size_t)c_ * NB + n_] = x_;                                    \
      __syncthreads();                      /* x_c visible */                             \
    }                                                                                     \
  } while (0)
      __syncthreads();                 // ya complete
      for (int k0 = 0; k0 < nops; k0 += 3) {
        BACK_OP(l0, k0);
        BACK_OP(l1, k0 + 1);
        BACK_OP(l2, k0 + 2);
      }
#undef BACK_OP
#undef BACK_LOAD
      // x -> the front's vector and the global solution
      for (int i = threadIdx.x; i < f.npt * NB; i += blockDim.x) {
        const double x = ya[i];
        st1(vecs + i, x);
        if (i < f.n1) st1(fd.delta + 7 * fd.nd_nodes[f.nodes_off + i / 7] + i % 7, x);
      }
      dag_publish_begin();
      dag_set_flag(g.px + f.pcol0);    // (column 0 stands for the whole front: what the children's BACKB wait for)
      DAG_END();
    }
}

// cut < 0: the whole tree (fd.dag_tasks).  cut >= 0: only the fronts of depth <= cut (fd.dag_top_tasks); the deeper
// levels are factored before and back-substituted after this launch by the per-level kernels (slm_front.hip).
// mode bit 0: XCD-AFFINE launch (n_frames a multiple of 8; the hybrid's top at 8 frames per launch): the workgroups of XCD x
// serve the frames x, x + 8, ... from a ticket stream of their own (the ticket word of frame x), so every task of a frame
// runs on ONE XCD and the frame's final operand tiles are shared through that XCD's L2 (load_tile_regs2(..., cached)) instead
// of being fetched from HBM by every task that multiplies with them.  Frames are independent: a task still waits on
// smaller tickets of its own stream only.
__global__ void __launch_bounds__(256, 2) k_fdag(const FrameDev* __restrict__ frames, int n_frames, int max_tasks,
                                                 double u_override, int cut, int mode) {
  int* s_ok = reinterpret_cast<int*>(dag_lds + 2 * TILE + 7 * 256 + 2 * NB);
  int* s_task = s_ok + 1;
  int* s_abort = s_ok + 2;
  const FrameDev& fd0 = frames[0];
  if (!fd0.bound || !fd0.nd_ready) return;
  const bool affine = (mode & 1) != 0;
  const int xcc = affine ? (__builtin_amdgcn_s_getreg(63508) & 7) : 0;      // hwreg(HW_REG_XCC_ID)
  const int per = affine ? n_frames / 8 : n_frames;                          // frames served by this ticket stream
  int* ticket = unip(affine ? frames[xcc].dag_flags : fd0.dag_flags);
  if (threadIdx.x == 0) *s_abort = 0;
  __syncthreads();
  const int total = per * max_tasks;
  for (;;) {
    if (threadIdx.x == 0) *s_task = addf(ticket, 1);
    __syncthreads();
    const int tk = uni(*s_task);   // provably uniform: descriptors stay in SGPRs
    __syncthreads();
    if (tk >= total || uni(*s_abort)) break;
    const int slot = affine ? xcc + 8 * (tk % per) : tk % per, ti = tk / per;
    const FrameDev& fq = frames[slot];
    if (!uni(fq.bound) || !uni(fq.nd_ready) || ti >= uni(cut >= 0 ? fq.n_dag_top_tasks : fq.n_dag_tasks)) continue;
    const int type = uni(unip(cut >= 0 ? fq.dag_top_tasks : fq.dag_tasks)[2 * ti]) >> 24;
    if (type == ND_T_POTRF) dag_task_factor<true>(frames, n_frames, slot, ti, u_override, cut, mode);
    else if (type == ND_T_COL) dag_task_factor<false>(frames, n_frames, slot, ti, u_override, cut, mode);
    else if (type == ND_T_SCHUR) dag_task_schur(frames, n_frames, slot, ti, u_override, cut, mode);
    else if (type == ND_T_BACKB) dag_task_backb(frames, n_frames, slot, ti, u_override, cut, mode);
    else dag_task_back(frames, n_frames, slot, ti, u_override, cut, mode);
  }
}

// how many XCDs do the workgroups of a launch land on (HW_REG_XCC_ID)?  one bit per id seen
__global__ void k_xcc_probe(unsigned* mask) {
  if (threadIdx.x == 0) atomicOr(mask, 1u << (__builtin_amdgcn_s_getreg(63508) & 31));
}

// zero the flags of slots [0, n_frames) (ticket, abort, counters, tile / column flags) and empty the mailboxes of the
// pivot tile columns this launch factors (all fronts, or those of depth <= cut); grid = (blocks, n_frames)
__global__ void __launch_bounds__(256) k_dag_reset(const FrameDev* __restrict__ frames, int cut) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || !fd.nd_ready || !fd.dag_flags) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < fd.dag_n_flags; i += gridDim.x * blockDim.x) fd.dag_flags[i] = 0;
  typedef __attribute__((address_space(1))) long long gll;
  gll* mail = (gll*)(double*)fd.fmail;
  for (int fi = blockIdx.x; fi < fd.n_fronts; fi += gridDim.x) {
    const NDFront& f = fd.fronts[fi];
    if (cut >= 0 && f.depth > cut) continue;
    const size_t base = (size_t)(f.linv_off / TILE) * SLM_MAIL_DOUBLES, n = (size_t)f.npt * SLM_MAIL_DOUBLES;
    for (size_t e = threadIdx.x; e < n; e += blockDim.x) mail[base + e] = SLM_MAIL_EMPTY;
  }
}

// An aborted launch (time-out of a wait: the abort flag, kept with the ticket in slot 0's flags) stops the LM loop of every
// slot whose solve did not finish -- a front whose BACK task never published its solution -- with its own status
// (chol_fail = 2 -> SLM_ITER_SOLVER_TIMEOUT, not "ill-posed system"); slots that were complete keep their result.
// One workgroup per slot: grid = (n_frames).
// always: an XCD-affine launch (launch_front_solve_dag) -- the check runs whether or not the abort flag is up.
__global__ void __launch_bounds__(64) k_dag_check(const FrameDev* __restrict__ frames, int n_frames, int always) {
  const FrameDev& fd0 = frames[0];
  if (!fd0.bound || !fd0.nd_ready || !fd0.dag_flags) return;
  if (fd0.dag_flags[1] == 0 && !always) return;
  const FrameDev& fd = frames[blockIdx.x];
  if (!fd.bound || !fd.nd_ready || !fd.dag_flags) return;
  const int* px = fd.dag_flags.get() + 8 + fd.dag_n_tiles + fd.dag_n_pcols;
  bool done = true;
  for (int fi = threadIdx.x; fi < fd.n_fronts; fi += blockDim.x) {
    const NDFront& f = fd.fronts[fi];
    if (f.npt > 0 && px[(int)(f.linv_off / TILE)] == 0) done = false;
  }
  if (!__all(done) && threadIdx.x == 0 && fd.st->chol_fail == 0) fd.st->chol_fail = 2;
}

// tests: what an aborted launch leaves behind -- flags reset (no front has published its solution), abort flag up --
// followed by the check that every launch ends with
__global__ void k_dag_raise_abort(const FrameDev* __restrict__ frames) {
  const FrameDev& fd0 = frames[0];
  if (fd0.bound && fd0.nd_ready && fd0.dag_flags) fd0.dag_flags[1] = 2;
}
void launch_dag_abort_check(const FrameDev* fr, int n_frames, hipStream_t st) {
  hipLaunchKernelGGL(k_dag_reset, dim3(64, n_frames), dim3(256), 0, st, fr, -1);
  hipLaunchKernelGGL(k_dag_raise_abort, dim3(1), dim3(1), 0, st, fr);
  hipLaunchKernelGGL(k_dag_check, dim3(n_frames), dim3(64), 0, st, fr, n_frames, 0);
}
hipError_t set_dag_timeout_ticks(long long ticks) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_dag_timeout_ticks), &ticks, sizeof(ticks));
}
// Per-device set-up of the task-graph launch: the dynamic-LDS attribute of k_fdag, the workgroup count (two per CU) and the
// XCD probe -- do the workgroups of a launch land on exactly the XCD ids 0..7?  A partitioned device must not wait for
// XCDs it does not have (k_fdag's XCD-affine mode).  slm_create calls this, so the probe (a hipMalloc, a launch on the null
// stream, a blocking copy, a hipFree) never sits between the launches of an LM iteration; launch_front_solve_dag falls
// back to it for a device it has not seen (a solver created on one device and driven on another).
struct DagDevice {
  std::atomic<int> n_wg{0};    // 0: not set up yet
  std::atomic<int> xcd8{0};    // 0 unknown, 1 launches land on the XCD ids 0..7, 2 no (or SLM_DAG_XCD=0)
};
static DagDevice g_dag_dev[64];
static std::atomic<int> g_last_dag_mode{-1};
int dag_device_setup(int dev, int* xcd8_out) {
  const bool tracked = dev >= 0 && dev < 64;
  int n_wg = tracked ? g_dag_dev[dev].n_wg.load(std::memory_order_acquire) : 0;
  int xcd8 = tracked ? g_dag_dev[dev].xcd8.load(std::memory_order_acquire) : 2;
  if (n_wg == 0) {
    const size_t lds = DAG_LDS_DOUBLES * sizeof(double);
    if (hipFuncSetAttribute((const void*)k_fdag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;   // (the launch would fail: hipGetLastError reports it)
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const char* e = getenv("SLM_DAG_WG_PER_CU");
    const int per_cu = e ? atoi(e) : 2;
    n_wg = cus * (per_cu > 0 ? per_cu : 1);
    if (tracked && xcd8 == 0) {
      static const bool off = [] { const char* e2 = getenv("SLM_DAG_XCD"); return e2 && atoi(e2) == 0; }();
      xcd8 = 2;
      unsigned* d_mask = nullptr;
      unsigned h_mask = 0;
      if (!off && hipMalloc((void**)&d_mask, sizeof(unsigned)) == hipSuccess) {
        if (hipMemset(d_mask, 0, sizeof(unsigned)) == hipSuccess) {
          hipLaunchKernelGGL(k_xcc_probe, dim3(4096), dim3(64), 0, 0, d_mask);
          if (hipMemcpy(&h_mask, d_mask, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess && h_mask == 0xFFu) xcd8 = 1;
        }
        (void)hipFree(d_mask);
      }
      g_dag_dev[dev].xcd8.store(xcd8, std::memory_order_release);
    }
    if (tracked) g_dag_dev[dev].n_wg.store(n_wg, std::memory_order_release);
  }
  if (xcd8_out) *xcd8_out = xcd8;
  return n_wg;
}
int dag_last_mode() { return g_last_dag_mode.load(std::memory_order_relaxed); }

// (TWO workgroups per CU since round 5 -- 248 VGPRs, 2 x 81 088 B of LDS: the bottom of the tree and the Schur tasks of a
//  batch are short of workgroups, not of registers: C2 one frame per launch 0.830 -> 0.804 ms per LM iteration, eight frames
//  2.462 -> 2.445; the build limited to 256 registers is itself 1 % faster at one workgroup per CU -- less scratch in the
//  task functions' prologues.  Tickets make any grid size deadlock-free)
// Returns the launch's mode word (bit 0: XCD-affine ticket streams), -1 when nothing was launched.
int launch_front_solve_dag(const FrameDev* fr, int n_frames, int max_tasks, double u_override, hipStream_t st, int cut, bool reset, bool check) {
  if (max_tasks <= 0) return -1;
  const size_t lds = DAG_LDS_DOUBLES * sizeof(double);
  int dev = 0;
  (void)hipGetDevice(&dev);
  int xcd8 = 2;
  const int n_wg = dag_device_setup(dev, &xcd8);
  if (n_wg <= 0) return -1;
  // reset = false: the flags and mailboxes were reset by this iteration's k_iter_begin_nd (launch_iter_begin_nd(..., dag_cut):
  // slm_run's loop); true: by a launch of their own here (the host-driven sharded loop, slm_solve)
  if (reset) hipLaunchKernelGGL(k_dag_reset, dim3(64, n_frames), dim3(256), 0, st, fr, cut);
  const long total = (long)n_frames * max_tasks;
  const int grid = (int)(total < n_wg ? total : n_wg);
  // XCD-affine ticket streams (k_fdag): the hybrid's top at a multiple of 8 frames per launch, on a device whose launches
  // land on exactly the XCD ids 0..7.  The probe says what the DEVICE does; a stream with a CU mask, or a partition change
  // behind the probe, could still leave an XCD without a workgroup of this launch -- the frames of its ticket stream would
  // never be taken and no wait would time out.  The completion check of an affine launch is therefore UNCONDITIONAL (every
  // front of every slot must have published its solution: k_dag_check / k_after_solve with mode 2), so such a launch ends
  // as SLM_ITER_SOLVER_TIMEOUT, not with a stale delta.
  const int mode = (n_frames >= 8 && n_frames % 8 == 0 && xcd8 == 1 && grid == n_wg) ? 1 : 0;
  hipLaunchKernelGGL(k_fdag, dim3(grid), dim3(256), lds, st, fr, n_frames, max_tasks, u_override, cut, mode);
  g_last_dag_mode.store(mode, std::memory_order_relaxed);
  // (check = false: the caller's next launch settles an aborted launch itself -- k_after_solve, slm_reg.hip)
  if (check) hipLaunchKernelGGL(k_dag_check, dim3(n_frames), dim3(64), 0, st, fr, n_frames, mode & 1);
  return mode;
}
