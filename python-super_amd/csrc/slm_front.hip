// slm_front.hip -- numeric phase of the nested-dissection multifrontal Cholesky (slm_nd.h).
//
// Per LM iteration:  zero fronts -> k_front_assemble (data-term 7x7 blocks through the
// inverted index, one writer per entry) -> k_reg_grad_nd (ARAP / Rot rows, f64 atomics)
// -> k_front_load_rhs -> for every tree level, deepest first:
//        for c < max pivot tile columns: k_fpanel(c), k_ftrail(c)   dense partial Cholesky
//        k_fschur(child 0), k_fschur(child 1)   Schur complements added into the parents (next level)
//    (the forward substitution rides along exactly as in the band solver)
// -> for every level, root first: k_fback_prep, k_fbacksub(step)...   back substitution,
//    each front scatters its pivots' solution into delta.
// Grid convention: blockIdx.y = front within the level, blockIdx.z = frame slot.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "slm_tile.h"
#include "slm_begin.h"

__device__ __forceinline__ int nd_base(const NDFront& f, int p) {
  return p < f.nv ? 7 * p : f.n1p + 7 * (p - f.nv);
}

// Tile (r, c) of a front.  One formula for every tile this file touches: the boundary block of an INTERNAL front
// follows its pivot columns (NDFront::f22_base == tile_off), and the boundary block of a LEAF -- the only tiles that
// live elsewhere -- is never read or written by the per-level form (k_fschur starts a leaf's update from zero and
// adds it straight into the parent, which has children and is therefore internal).
__device__ __forceinline__ double* ftile(const FrameDev& fd, const NDFront& f, int r, int c) {
  const size_t t = (size_t)c * f.nt - (size_t)c * (c - 1) / 2 + (size_t)(r - c);
  return fd.ftiles + f.tile_off + t * TILE;
}

// entry (i,j), i >= j, in front-local scalar coordinates
__device__ __forceinline__ double* front_entry(const FrameDev& fd, const NDFront& f, int i, int j) {
  return ftile(fd, f, i >> 6, j >> 6) + (i & 63) + (size_t)(j & 63) * NB;
}

// element (x,y) of a node-pair block given as (a,b) with a >= b by id -> lower-stored location
__device__ __forceinline__ double* dest_entry(const FrameDev& fd, const NDDest& d, int x, int y) {
  const NDFront& f = fd.fronts[d.front];
  const int rb = nd_base(f, d.prow), cb = nd_base(f, d.pcol);
  return d.transpose ? front_entry(fd, f, rb + y, cb + x) : front_entry(fd, f, rb + x, cb + y);
}


// ---------------------------------------------------------------------------------
// Workgroup -> (unit, front, frame) decoding.  The plain form is a 3-D grid.  The XCD-aware form
// (used when a launch has >= 8 (front, frame) pairs) is a 1-D grid in which all units (tiles) of
// a pair run on ONE XCD, back to back: blocks b and b+8 share an XCD and its 4 MiB L2, so block b
// takes pair (b & 7) * chunk + (b >> 3) / n_units.  A front's operand tiles are then fetched from
// HBM once per launch instead of once per consuming tile.
struct WgMap {
  int n_units, n_fronts, n_frames, xcd;
};
// Level of the separator tree a launch works on.  When every slot of the batch has the same plan
// the host passes the level's first front and front count by value (first >= 0), which takes a
// dependent load (the slot's level table) out of every kernel prologue.
struct LevelRef {
  int level, first, count;
};
__device__ __forceinline__ bool level_front(const FrameDev& fd, const LevelRef& lr, int idx, int& fi) {
  if (lr.first >= 0) {
    fi = lr.first + idx;
    return idx < lr.count;
  }
  if (lr.level >= fd.n_levels) return false;
  fi = fd.level_start[lr.level] + idx;
  return fi < fd.level_start[lr.level + 1];
}
struct WgId {
  int unit, front, frame;
};
__device__ __forceinline__ bool wg_decode(const WgMap& m, WgId& o) {
  if (!m.xcd) {
    o.unit = blockIdx.x;
    o.front = blockIdx.y;
    o.frame = blockIdx.z;
    return true;
  }
  const int n_pairs = m.n_fronts * m.n_frames;
  const int chunk = (n_pairs + 7) >> 3;
  const int q = (blockIdx.x >> 3) / m.n_units;
  const int pair = (blockIdx.x & 7) * chunk + q;
  o.unit = (blockIdx.x >> 3) % m.n_units;
  o.front = pair % m.n_fronts;
  o.frame = pair / m.n_fronts;
  return q < chunk && pair < n_pairs;
}
static inline WgMap make_map(int n_units, int n_fronts, int n_frames) {
  return WgMap{n_units, n_fronts, n_frames, (n_fronts * n_frames >= 8) ? 1 : 0};
}
static inline dim3 map_grid(const WgMap& m) {
  if (!m.xcd) return dim3(m.n_units, m.n_fronts, m.n_frames);
  return dim3((unsigned)((m.n_fronts * m.n_frames + 7) / 8 * 8) * m.n_units);
}

// ---------------------------------------------------------------------------------
// G[i][j], i >= j, from a slab entry (tiles 00, 10, 11; 16x16 row-major each)
__device__ __forceinline__ double gram_at(const double* g, int i, int j) {
  if (i < 16) return g[16 * i + j];
  if (j < 16) return g[256 + 16 * (i - 16) + j];
  return g[512 + 16 * (i - 16) + (j - 16)];
}

// grid = (ceil(max n_blocks / 4), n_frames): one wave per coupled node pair of the data term
__global__ void __launch_bounds__(256) k_front_assemble(const FrameDev* __restrict__ frames) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || !fd.v1_ready || !fd.nd_ready || fd.st->stopped) return;
  const int bi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bi >= fd.n_blocks) return;
  const int l = threadIdx.x & 63;
  const NDDest d = fd.block_dest[bi];
  const int ca = l / 7, cb = l % 7;
  const bool diag = d.prow == d.pcol;
  const bool act = (l < 49) && (!diag || ca >= cb);
  const bool jt = diag && l >= 49 && l < 56;   // diagonal pair: lanes 49..55 build jtl of the node
  double acc = 0.0;
  if (fd.v2_ready) {
    // one 56-double record per (workgroup, pair): 49 block entries + 7 entries of J^T r
    const int s0 = fd.blk2_start[bi], s1 = fd.blk2_start[bi + 1];
    // record numbers in chunks of 8, then the 8 record reads together (two round trips per chunk
    // instead of two per record); the sum keeps the record order
    for (int sb = s0; sb < s1; sb += 8) {
      int ent[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) ent[k] = sb + k < s1 ? fd.blk2_entry[sb + k] : -1;
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (ent[k] >= 0 && l < SLM_WREC) ? fd.wgslab[(size_t)ent[k] * SLM_WREC + l] : 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (ent[k] >= 0) acc += v[k];
    }
  } else {
    const int s0 = fd.blk_start[bi], s1 = fd.blk_start[bi + 1];
    for (int s = s0; s < s1; ++s) {
      const int pl = fd.blk_entry[s];
      const int run = pl >> 4, pa = (pl >> 2) & 3, pb = pl & 3;
      const double* G = fd.slab + (size_t)run * SLM_SLAB_STRIDE;
      if (act) acc += gram_at(G, 7 * pa + ca, 7 * pb + cb);
      else if (jt) acc += gram_at(G, 28, 7 * pa + (l - 49));
    }
  }
  if (act) *dest_entry(fd, d, ca, cb) = acc;
  else if (jt) {
    const unsigned key = (unsigned)fd.blk_key[bi];
    fd.rhs[7 * (int)(key / (unsigned)fd.f.J) + (l - 49)] = -acc;   // jtl = -J^T r
  }
}

// Sharded frames: the pair sums stop in pairbuf (k_pair_reduce), the caller all-reduces it over the
// ranks, and k_pair_scatter places the reduced blocks.  Both are k_front_assemble cut in two.
__global__ void __launch_bounds__(256) k_pair_reduce(const FrameDev* __restrict__ frames) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || !fd.v1_ready || !fd.nd_ready || fd.st->stopped) return;
  const int bi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bi == 0 && (threadIdx.x & 63) == 0) fd.pairbuf[(size_t)fd.n_blocks * SLM_WREC] = (double)fd.st->m_grad;
  if (bi >= fd.n_blocks) return;
  const int l = threadIdx.x & 63;
  if (l >= SLM_WREC) return;
  double acc = 0.0;
  if (fd.v2_ready) {
    for (int s = fd.blk2_start[bi]; s < fd.blk2_start[bi + 1]; ++s)
      acc += fd.wgslab[(size_t)fd.blk2_entry[s] * SLM_WREC + l];
  } else {   // one Gram per run in HBM (a workgroup would need more than SLM_LB_MAX records)
    const int ca = l / 7, cb = l % 7;
    for (int s = fd.blk_start[bi]; s < fd.blk_start[bi + 1]; ++s) {
      const int pl = fd.blk_entry[s];
      const int run = pl >> 4, pa = (pl >> 2) & 3, pb = pl & 3;
      const double* G = fd.slab + (size_t)run * SLM_SLAB_STRIDE;
      // entries 0..48: block (pa,pb) (the diagonal pair only needs ca >= cb, the rest is ignored by the
      // scatter); entries 49..55: row 28 = J^T r of node slot pa (only used for diagonal pairs)
      if (l < 49) acc += (pa != pb || ca >= cb) ? gram_at(G, 7 * pa + ca, 7 * pb + cb) : 0.0;
      else acc += gram_at(G, 28, 7 * pa + (l - 49));
    }
  }
  fd.pairbuf[(size_t)bi * SLM_WREC + l] = acc;
}

// (also the assembly of the K-generic pair path -- FrameDev::vk_ready: k_data_grad_pairs filled pairbuf)
__global__ void __launch_bounds__(256) k_pair_scatter(const FrameDev* __restrict__ frames) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || !(fd.v1_ready || fd.vk_ready) || !fd.nd_ready || fd.st->stopped) return;
  const int bi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bi == 0 && (threadIdx.x & 63) == 0) {
    // the rank's own count is kept apart: after a rejected step the records -- and this count -- are sent again
    // (k_iter_begin_nd restores it), and the all-reduced count must not take its place
    fd.st->m_grad_local = fd.st->m_grad;
    double cnt = fd.pairbuf[(size_t)fd.n_blocks * SLM_WREC];
    if (fd.vk_ready)   // K-generic pair path: the count is spread over SLM_VK_TAIL doubles (k_data_grad_pairs)
      for (int e = 1; e < SLM_VK_TAIL; ++e) cnt += fd.pairbuf[(size_t)fd.n_blocks * SLM_WREC + e];
    fd.st->m_grad = (int)cnt;
  }
  if (bi >= fd.n_blocks) return;
  const int l = threadIdx.x & 63;
  if (l >= SLM_WREC) return;
  const NDDest d = fd.block_dest[bi];
  const int ca = l / 7, cb = l % 7;
  const bool diag = d.prow == d.pcol;
  const double acc = fd.pairbuf[(size_t)bi * SLM_WREC + l];
  if (l < 49) {
    if (!diag || ca >= cb) *dest_entry(fd, d, ca, cb) = acc;
  } else if (diag) {
    const unsigned key = (unsigned)fd.blk_key[bi];
    fd.rhs[7 * (int)(key / (unsigned)fd.f.J) + (l - 49)] = -acc;   // jtl = -J^T r
  }
}

// ---------------------------------------------------------------------------------
// ARAP + Rot Jacobian rows into the fronts (same maths as k_reg_grad, slm_reg.hip; reference
// super/loss.py:408-455, 480-499).  One thread per (node j, neighbour slot).
__device__ __forceinline__ void nd_load_beta(const double* beta, int j, double bb[7]) {
#pragma unroll
  for (int c = 0; c < 7; ++c) bb[c] = beta[7 * j + c];
}

__device__ __forceinline__ float nd_rot_residual32(const double bb[7], float lam32, float q[4]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int c = 0; c < 4; ++c) q[c] = (float)bb[c];
  float s = q[0] * q[0];
  s = s + q[1] * q[1];
  s = s + q[2] * q[2];
  s = s + q[3] * q[3];
  return lam32 * (1.0f - s);
}

__device__ __forceinline__ void nd_rot_products32(const float q[4], float lam32, float r,
                                                  float jtj[4][4], float jtr[4]) {
#pragma clang fp contract(off)
  float jv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) jv[c] = (-lam32 * 2.0f) * q[c];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    jtr[a] = -(jv[a] * r);
#pragma unroll
    for (int b = 0; b < 4; ++b) jtj[a][b] = jv[a] * jv[b];
  }
}

// ARAP residual of edge (j -> k) and the quaternion Jacobian of node k's part (reference
// super/loss.py:408-455): r = lam [R(q_k) d + b_k - d - b_j], d = g_j - g_k
__device__ __forceinline__ void nd_arap_edge(const FrameDev& fd, int j, int k, double lam_a, double r[3], double Jq[3][4]) {
  const d3 d = node_pos_pk(fd.node_pk, j) - node_pos_pk(fd.node_pk, k);
  double bk[7], bj[7];
  nd_load_beta(fd.beta, k, bk);
  nd_load_beta(fd.beta, j, bj);
  const d3 tt = quat_apply(bk[0], {bk[1], bk[2], bk[3]}, d);
  r[0] = lam_a * (tt.x + bk[4] - d.x - bj[4]);
  r[1] = lam_a * (tt.y + bk[5] - d.y - bj[5]);
  r[2] = lam_a * (tt.z + bk[6] - d.z - bj[6]);
  quat_jac(bk[0], {bk[1], bk[2], bk[3]}, d, Jq);
}

// No atomics and a fixed summation order.  Node j is served by a group of RG_LANES = 8 consecutive lanes
// (K_ED <= 8): lane `slot` < K_ED owns the cross block of edge j -> k; for node j's diagonal block and right-hand
// side the lanes SHARE the edges -- lane s takes the in-edges start+s, start+s+8, ... (reverse KNN graph) and the
// out-edge s -- and a butterfly sum over the 8 lanes (same tree in every lane: deterministic) gives the block,
// whose 35 entries are then read-modify-written 4-5 per lane instead of 35 in a row by one thread.
// Every entry is touched by exactly one thread (k_front_assemble stored the data term first).
#define RG_LANES 8
__global__ void __launch_bounds__(256) k_reg_grad_nd(const FrameDev* __restrict__ frames, int use_arap,
                                                      double lam_a, int use_rot, double lam_r) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || !fd.nd_ready || fd.st->stopped) return;
  const int Ke = fd.f.K_ED, J = fd.f.J;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = t / RG_LANES, slot = t % RG_LANES;
  if (j >= J) return;            // (whole groups: RG_LANES divides the wave size)
  const double l2 = lam_a * lam_a;
  if (use_arap && slot < Ke) {
    // ---- cross block between the nodes of edge j -> k ----
    const int k = frame_in(fd).ed_knn_idx[j * Ke + slot];
    if (k >= 0 && k < J && k != j) {
      const NDDest pd = fd.pair_dest[j * Ke + slot];   // block (max(j,k), min(j,k))
      const NDFront fp = fd.fronts[pd.front];
      const int prb = nd_base(fp, pd.prow), pcb = nd_base(fp, pd.pcol);
      const bool k_is_row = ((k > j) != (pd.transpose != 0));   // is node k the ROW node of the stored block?
      double r[3], Jq[3][4];
      nd_arap_edge(fd, j, k, lam_a, r, Jq);
      // the reverse edge k -> j (if it exists) shares the three (translation, translation) entries:
      // the edge with the smaller source node writes them for both
      bool reverse = false;
      for (int s2 = 0; s2 < Ke; ++s2) reverse = reverse || frame_in(fd).ed_knn_idx[k * Ke + s2] == j;
      double* dst[15];
      double add[15];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          add[4 * c + a] = -l2 * Jq[c][a];                       // between (k, a) and (j, 4+c)
          dst[4 * c + a] = k_is_row ? front_entry(fd, fp, prb + a, pcb + 4 + c) : front_entry(fd, fp, prb + 4 + c, pcb + a);
        }
        const bool mine = !reverse || j < k;
        dst[12 + c] = mine ? front_entry(fd, fp, prb + 4 + c, pcb + 4 + c) : nullptr;   // between (k, 4+c) and (j, 4+c)
        add[12 + c] = reverse ? -2.0 * l2 : -l2;
      }
      // all loads before the first store: the destinations are distinct but the compiler cannot know
      double cur[15];
#pragma unroll
      for (int e = 0; e < 15; ++e) cur[e] = dst[e] ? *dst[e] : 0.0;
#pragma unroll
      for (int e = 0; e < 15; ++e)
        if (dst[e]) *dst[e] = cur[e] + add[e];
    }
  }
  // ---- node j: diagonal block (lower 7x7, packed row-major: a(a+1)/2 + b) and right-hand side ----
  double acc[28], rh[7];
#pragma unroll
  for (int e = 0; e < 28; ++e) acc[e] = 0.0;
#pragma unroll
  for (int a = 0; a < 7; ++a) rh[a] = 0.0;
  if (use_arap) {
    for (int ie = fd.in_start[j] + slot; ie < fd.in_start[j + 1]; ie += RG_LANES) {   // edges src -> j: j plays node k
      const int src = fd.in_edge[ie] / Ke;
      if (src == j) continue;
      double r[3], Jq[3][4];
      nd_arap_edge(fd, src, j, lam_a, r, Jq);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        double gr = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) gr += Jq[c][a] * r[c];
        rh[a] += -lam_a * gr;
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          double q = 0.0;
#pragma unroll
          for (int c = 0; c < 3; ++c) q += Jq[c][a] * Jq[c][b];
          acc[a * (a + 1) / 2 + b] += l2 * q;
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        rh[4 + c] += -lam_a * r[c];
        acc[(4 + c) * (5 + c) / 2 + 4 + c] += l2;
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[(4 + c) * (5 + c) / 2 + a] += l2 * Jq[c][a];
      }
    }
    if (slot < Ke) {                                                  // edge j -> k: j plays node j
      const int k = frame_in(fd).ed_knn_idx[j * Ke + slot];
      if (k >= 0 && k < J && k != j) {
        double r[3], Jq[3][4];
        nd_arap_edge(fd, j, k, lam_a, r, Jq);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          rh[4 + c] += lam_a * r[c];
          acc[(4 + c) * (5 + c) / 2 + 4 + c] += l2;
        }
      }
    }
  }
  if (use_rot && slot == 0) {
    double bb[7];
    nd_load_beta(fd.beta, j, bb);
    float q[4];
    const float lam32 = (float)lam_r;
    const float r = nd_rot_residual32(bb, lam32, q);
    float jtj[4][4], jtr[4];
    nd_rot_products32(q, lam32, r, jtj, jtr);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      rh[a] += (double)jtr[a];
#pragma unroll
      for (int b = 0; b <= a; ++b) acc[a * (a + 1) / 2 + b] += (double)jtj[a][b];
    }
  }
  // group sum: every lane ends with the same 35 values
#pragma unroll
  for (int off = 1; off < RG_LANES; off <<= 1) {
#pragma unroll
    for (int e = 0; e < 28; ++e) acc[e] += __shfl_xor(acc[e], off);
#pragma unroll
    for (int a = 0; a < 7; ++a) rh[a] += __shfl_xor(rh[a], off);
  }
  const NDFront fj = fd.fronts[fd.node_front[j]];
  const int bj0 = nd_base(fj, fd.node_pos[j]);
  // entry e = slot, slot + 8, ...: 0..27 the block, 28..34 the right-hand side
  double* dst[5];
  double add[5], cur[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int e = slot + RG_LANES * q;
    dst[q] = nullptr;
    add[q] = 0.0;
    if (e < 28) {
      int a = 0;
      while ((a + 1) * (a + 2) / 2 <= e) ++a;
      const int b = e - a * (a + 1) / 2;
      // select acc[e] without dynamic register indexing
      double v = 0.0;
#pragma unroll
      for (int x = 0; x < 28; ++x) v = (x == e) ? acc[x] : v;
      if (v != 0.0) {
        dst[q] = front_entry(fd, fj, bj0 + a, bj0 + b);
        add[q] = v;
      }
    } else if (e < 35) {
      double v = 0.0;
#pragma unroll
      for (int x = 0; x < 7; ++x) v = (x == e - 28) ? rh[x] : v;
      dst[q] = fd.rhs + 7 * j + (e - 28);
      add[q] = v;
    }
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) cur[q] = dst[q] ? *dst[q] : 0.0;
#pragma unroll
  for (int q = 0; q < 5; ++q)
    if (dst[q]) *dst[q] = cur[q] + add[q];
  // the node's right-hand side is final here (data term + regularisers): it goes to the pivot part of its front's vector in
  // the same pass (k_front_load_rhs was a launch of its own; it still runs when no regulariser is enabled)
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int e = slot + RG_LANES * q;
    if (e >= 28 && e < 35) fd.fvec[fj.vec_off + 7 * fd.node_pos[j] + (e - 28)] = cur[q] + add[q];
  }
}

// global jtl -> the pivot part of each front's vector (boundary parts stay zero)
__global__ void __launch_bounds__(256) k_front_load_rhs(const FrameDev* __restrict__ frames) {
  const FrameDev& fd = frames[blockIdx.y];
  if (!fd.bound || !fd.nd_ready || fd.st->stopped) return;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= fd.P) return;
  const int v = e / 7, c = e % 7;
  const NDFront& f = fd.fronts[fd.node_front[v]];
  fd.fvec[f.vec_off + 7 * fd.node_pos[v] + c] = fd.rhs[e];
}

// ---- zeroing of the fronts (slm_begin.h) --------------------------------------------------------------------------
// Start of an iteration on the multifrontal path as a launch of its own: for ALL slots of the batch (one hipMemsetAsync
// pair per slot cost ~25 us each, back to back).  grid = (blocks, n_frames); contiguous 16 KB pieces per workgroup.
// (The LM loop's tuple-sorted path runs the same body as the tail blocks of its Jacobian pass instead: launch_data_gram.)
__global__ void __launch_bounds__(256) k_iter_begin_nd(const FrameDev* __restrict__ frames, const int* __restrict__ reuse, int dag_cut) {
  iter_begin_nd_body(frames[blockIdx.y], blockIdx.x, gridDim.x, reuse && reuse[blockIdx.y], dag_cut, false);
}

// ---------------------------------------------------------------------------------
// Dense partial Cholesky of the fronts of one level, tile column c.
// grid = (max tiles below + 1, fronts in level, n_frames)
__global__ void __launch_bounds__(256) k_fpanel(const FrameDev* __restrict__ frames, LevelRef lvl,
                                                 int c, double u_override, WgMap map) {
  extern __shared__ double lds[];
  double* S = lds;
  double* M = lds + TILE;
  double* dinv = lds + 2 * TILE;
  double* wt = dinv + 4 * 256;   // 3 scratch blocks (inverse_assemble64 runs on at most 3 waves)
  double* vec = wt + 3 * 256;
  double* xch = vec + NB;        // 2 NB: exchange buffer of the pipelined tile factorisation
  int* s_ok = reinterpret_cast<int*>(xch + 2 * NB);
  int* pf = s_ok + 8;            // its 16 hand-off flags
  WgId wg;
  if (!wg_decode(map, wg)) return;
  const FrameDev& fd = frames[wg.frame];
  if (!fd.bound || !fd.nd_ready) return;   // (a stopped slot only wastes the work: no dependent flag load here)
  int fi;
  if (!level_front(fd, lvl, wg.front, fi)) return;
  const NDFront& f = fd.fronts[fi];
  if (c >= f.npt) return;
  const int d = wg.unit;
  if (c + d >= f.nt) return;
  const double u = (u_override >= 0.0) ? u_override : fd.st->u;

  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
  const bool stamp = (c == 1 && wg.frame == 0 && wg.front == 0 && d == 1 && lvl.level == fd.n_levels - 1);
  SLM_STAMP(fd, stamp, 0);
  double* At = ftile(fd, f, c + d, c);
  double* yv = fd.fvec + f.vec_off + (size_t)c * NB;
  double4_t a[4];
  if (d > 0) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) a[kb][r] = At[(16 * w + lr) + (size_t)(16 * kb + lk + 4 * r) * NB];
  } else if (threadIdx.x < NB) {
    vec[threadIdx.x] = yv[threadIdx.x];
  }
  {
    // diagonal tile -> LDS: damping on real pivots, identity on the padding rows
    const double* src = ftile(fd, f, c, c);
    double v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = src[threadIdx.x + 256 * t];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int e = threadIdx.x + 256 * t;
      const int i = e % NB, k = e / NB;
      double x = (i >= k) ? v[t] : 0.0;
      if (i == k) x = (c * NB + i < f.n1) ? x + u : 1.0;
      S[i + k * LD] = x;
    }
  }
  __syncthreads();
  SLM_STAMP(fd, stamp, 1);
  // (factor + inverse in the pipelined form of the task graph; the row blocks d > 0 only need L and the diagonal-block
  //  inverses, which are complete at the same time as with potrf64)
  const bool ok = factor_inverse64p(S, M, dinv, wt, xch, s_ok, pf, nullptr, nullptr, nullptr, min(4, (f.n1 - c * NB + 15) >> 4));
  SLM_STAMP(fd, stamp, 14);

  if (d == 0) {
    if (!ok && threadIdx.x == 0) fd.st->chol_fail = 1;
    double* linv = fd.flinv + f.linv_off + (size_t)c * TILE;
    for (int e = threadIdx.x; e < TILE; e += blockDim.x) linv[e] = M[e];
    if (threadIdx.x < NB) {
      const int i = threadIdx.x;
      double acc = 0.0;
      for (int k = 0; k <= i; ++k) acc += M[i + k * LD] * vec[k];
      yv[i] = acc;
    }
  } else {
    trsm_rows16(S, dinv, a);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) At[(16 * w + lr) + (size_t)(16 * kb + lk + 4 * r) * NB] = a[kb][r];
  }
  SLM_STAMP(fd, stamp, 15);
}

// Split form of k_fpanel for levels with many fronts x frames (the fused form factors the
// diagonal tile redundantly in every block of the tile column, which is the right trade only
// while the launch is latency-bound):
//   k_fpotrf  one block per front: factor, full inverse -> flinv, forward substitution
//   k_ftrsm   block d >= 1: L(c+d,c) = A(c+d,c) L_cc^-T as one tile product with L_cc^-1
// grid = (1, fronts in level, n_frames)
__global__ void __launch_bounds__(256) k_fpotrf(const FrameDev* __restrict__ frames, LevelRef lvl, int c,
                                                 double u_override) {
  extern __shared__ double lds[];
  double* S = lds;
  double* M = lds + TILE;
  double* dinv = lds + 2 * TILE;
  double* wt = dinv + 4 * 256;   // 3 scratch blocks (inverse_assemble64 runs on at most 3 waves)
  double* vec = wt + 3 * 256;
  double* xch = vec + NB;        // 2 NB: exchange buffer of the pipelined tile factorisation
  int* s_ok = reinterpret_cast<int*>(xch + 2 * NB);
  int* pf = s_ok + 8;            // its 16 hand-off flags
  const FrameDev& fd = frames[blockIdx.z];
  if (!fd.bound || !fd.nd_ready) return;   // (a stopped slot only wastes the work: no dependent flag load here)
  int fi;
  if (!level_front(fd, lvl, blockIdx.y, fi)) return;
  const NDFront& f = fd.fronts[fi];
  if (c >= f.npt) return;
  const double u = (u_override >= 0.0) ? u_override : fd.st->u;
  double* yv = fd.fvec + f.vec_off + (size_t)c * NB;
  if (threadIdx.x < NB) vec[threadIdx.x] = yv[threadIdx.x];
  {
    const double* src = ftile(fd, f, c, c);
    double v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = src[threadIdx.x + 256 * t];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int e = threadIdx.x + 256 * t;
      const int i = e % NB, k = e / NB;
      double x = (i >= k) ? v[t] : 0.0;
      if (i == k) x = (c * NB + i < f.n1) ? x + u : 1.0;
      S[i + k * LD] = x;
    }
  }
  __syncthreads();
  const bool ok = factor_inverse64p(S, M, dinv, wt, xch, s_ok, pf, nullptr, nullptr, nullptr, min(4, (f.n1 - c * NB + 15) >> 4));
  if (!ok && threadIdx.x == 0) fd.st->chol_fail = 1;
  double* linv = fd.flinv + f.linv_off + (size_t)c * TILE;
  for (int e = threadIdx.x; e < TILE; e += blockDim.x) linv[e] = M[e];
  if (threadIdx.x < NB) {
    const int i = threadIdx.x;
    double acc = 0.0;
    for (int k = 0; k <= i; ++k) acc += M[i + k * LD] * vec[k];
    yv[i] = acc;
  }
}

// grid = (max tiles below, fronts in level, n_frames): block d-1 -> tile (c+d, c)
__global__ void __launch_bounds__(256) k_ftrsm(const FrameDev* __restrict__ frames, LevelRef lvl, int c, WgMap map) {
  __shared__ double Bl[TILE];
  WgId wg;
  if (!wg_decode(map, wg)) return;
  const FrameDev& fd = frames[wg.frame];
  if (!fd.bound || !fd.nd_ready) return;   // (a stopped slot only wastes the work: no dependent flag load here)
  int fi;
  if (!level_front(fd, lvl, wg.front, fi)) return;
  const NDFront& f = fd.fronts[fi];
  if (c >= f.npt) return;
  const int d = wg.unit + 1;
  if (c + d >= f.nt) return;
  double* At = ftile(fd, f, c + d, c);
  const double* linv = fd.flinv + f.linv_off + (size_t)c * TILE;
  double breg[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) breg[e] = linv[threadIdx.x + 256 * e];
  double areg[16];
  load_a_frags(At, areg);
#pragma unroll
  for (int e = 0; e < 16; ++e) Bl[threadIdx.x + 256 * e] = breg[e];
  __syncthreads();
  double4_t acc[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) acc[ni] = double4_t{0.0, 0.0, 0.0, 0.0};
  tile_ABt_regs<false, true>(areg, Bl, acc);   // A (L^-1)^T
  store_c_frags(At, acc);
}

// ---- pull-form extend-add (the data flow of the task graph, slm_dag.hip, in the per-level kernels) ---------------
// Every tile has ONE writer.  A front's boundary block F22 holds its UPDATE matrix U = [children's updates mapped into
// it] - L21 L21^T, written once by k_fschur (plain coalesced stores: first touch is a store, nothing is zeroed, nothing
// is read-modify-written) and read once by the parent, which GATHERS the entries that map into its own tiles through
// the plan's pull maps (pullmap: parent scalar index -> boundary scalar index of the child, -1 none): k_fpull for the
// pivot columns (added to the assembled entries before the factorisation), k_fschur for the boundary block.  The sums
// run in a fixed order (own, child 0, child 1): bitwise reproducible.  The boundary rows of the front VECTOR travel
// the same way (v_r = [children's v mapped] - sum_c L(r,c) y_c).
// Both kernels take their work from the plan's exact lists of 128-byte records (NDTileItem, slm_nd.h): no empty
// workgroups, and one load in place of the chain level table -> front -> child list -> pull ranges -> child fronts.
//
// Work item of workgroup blockIdx.x.  XCD-aware order: blocks b and b+8 share an XCD, so XCD x takes the contiguous
// range [x*chunk, (x+1)*chunk) of the (frame, item) space -- the tiles of a front (adjacent items) run on one XCD,
// back to back, and its L21 operands are fetched from HBM once.  `at` < 0: the slots of the batch have different
// plans, the slot's own table says where the level's items are (which: 0 Schur items, 1 pull items).
__device__ __forceinline__ const NDTileItem* tile_item(const FrameDev* __restrict__ frames, int level, int which, int n_items,
                                                       int at, int n_frames, const FrameDev*& fdp) {
  const int total = n_items * n_frames;
  const int chunk = (total + 7) >> 3;
  const int q = blockIdx.x >> 3;
  const int w_id = (blockIdx.x & 7) * chunk + q;
  if (q >= chunk || w_id >= total) return nullptr;
  const int frame = w_id / n_items, item_idx = w_id - frame * n_items;
  const FrameDev& fd = frames[frame];
  if (!fd.bound || !fd.nd_ready) return nullptr;   // (a stopped slot only wastes the work: no dependent flag load here)
  int i0 = at;
  if (at < 0) {
    if (level >= fd.n_levels) return nullptr;
    i0 = fd.item_off[2 * level + which];
    if (item_idx >= fd.item_off[2 * level + which + 1] - i0) return nullptr;
  }
  fdp = &fd;
  return fd.tile_items.get() + i0 + item_idx;
}
// The record as scalars: loaded through a pointer that was itself loaded from memory, the fields are "divergent" for the
// compiler and would live in vector registers (33 VGPRs: k_fschur fell from four to three waves per SIMD); every lane
// reads the same record, so a v_readfirstlane per word puts it into SGPRs.
__device__ __forceinline__ NDTileItem item_snapshot(const NDTileItem* p) {
  NDTileItem o;
  const int* src = reinterpret_cast<const int*>(p);
  int* dst = reinterpret_cast<int*>(&o);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(NDTileItem) / 4); ++i) dst[i] = __builtin_amdgcn_readfirstlane(src[i]);
  return o;
}
// tile (r, c) of a front from its item record (pivot columns at tile_off, boundary block at f22_base: NDFront)
__device__ __forceinline__ double* item_tile(const FrameDev& fd, const NDTileItem& it, int r, int c) {
  const size_t t = (size_t)c * it.nt - (size_t)c * (c - 1) / 2 + (size_t)(r - c);
  return fd.ftiles + (c < it.npt ? it.tile_off : it.f22_base) + t * TILE;
}
// maps -> LDS: [128 k + 0..63] child k's boundary scalar of every row of tile row r, [128 k + 64..127] of tile row c.
// Needs a __syncthreads() before the first use (the callers have one on their way).
// maps: 256 int2 of LDS: [128 k + 0..63] the rows of the item's tile row, [128 k + 64..127] the columns of its tile column, child k:
// .x = the child's boundary scalar that maps there (-1: none), .y = the offset (in doubles) of that row (resp. column) inside the
// child's update block -- an entry's address is block + row offset + column offset: one add per gathered entry instead of
// the tile arithmetic (the same layout as the task graph's pull maps, slm_dag.hip)
__device__ __forceinline__ void pull_maps(const FrameDev& fd, const NDTileItem& it, int2* maps) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (it.kid[k].front >= 0 && threadIdx.x < 128) {
      const int32_t* pm = fd.pullmap + it.kid[k].pull_off;
      const bool row = threadIdx.x < 64;
      const int c = pm[64 * (row ? it.r : it.c) + (threadIdx.x & 63)];
      int off = 0;
      if (c >= 0) {
        const int t = it.kid[k].npt + (c >> 6);      // tile row / tile column of the child's front
        off = row ? t * TILE + (c & 63) : (t * it.kid[k].nt - t * (t - 1) / 2 - t) * TILE + (c & 63) * NB;
      }
      maps[128 * k + threadIdx.x] = make_int2(c, off);
    }
  }
}
// acc (the item's tile in accumulator layout: wave w rows 16w.., see load_c_frags) += the children's entries.  All 16
// gathers of a child are in flight together; the sum order is child 0, then child 1.  (One child at a time, fenced
// for the instruction scheduler: both children's 32 values in flight cost 64 more registers.)
__device__ __forceinline__ void pull_tile(const FrameDev& fd, const NDTileItem& it, const int2* maps, double4_t acc[4]) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (it.kid[k].front < 0) continue;
    __builtin_amdgcn_sched_barrier(0);   // (the 16 gather addresses are formed here, not hoisted above the caller's barrier)
    double v[16];
    const double* ct = fd.ftiles + it.kid[k].f22_base;
    const int2* mk = maps + 128 * k;
    const int2 rw = mk[16 * w + lr];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int2 cl = mk[64 + 16 * ni + lk + 4 * rr];
        // The maps are monotone (a front lists its nodes in elimination order, and so does its parent): an entry of the
        // parent's lower triangle comes from the child's lower triangle.  ci < cj only occurs above the diagonal of a
        // diagonal tile, which nothing reads.
        // (branch-free: an entry nothing maps into reads the child's first word and drops it -- 16 loads back to back
        //  instead of 16 predicated branches with one load each)
        const bool ok = cl.x >= 0 && rw.x >= cl.x;
        const double x = ct[ok ? rw.y + cl.y : 0];
        v[4 * ni + rr] = ok ? x : 0.0;
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[ni][rr] += v[4 * ni + rr];
    __builtin_amdgcn_sched_barrier(0);
  }
}
// the children's boundary vector rows that map into tile row r of the front (threads < NB: row threadIdx.x)
__device__ __forceinline__ double pull_vec(const FrameDev& fd, const NDTileItem& it, const int2* maps) {
  double s = 0.0;
  if (threadIdx.x < NB) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (it.kid[k].front < 0) continue;
      const int ci = maps[128 * k + threadIdx.x].x;
      if (ci >= 0) s += fd.fvec[it.kid[k].vec_boundary + ci];
    }
  }
  return s;
}

// ---------------------------------------------------------------------------------------
// Compact form for levels whose fronts have few pivot tile columns (npt <= 4: the leaf side
// of the tree, where there are many fronts): two launches per level instead of three per
// tile column.
//   k_fL11  one workgroup per front factors the whole pivot block L11 (all npt tile columns:
//           tile Cholesky + inverse, row solves and trailing updates inside the block, forward
//           substitution of the pivot rows), looping over its tiles through LDS / L2;
//   k_fL21  one workgroup per boundary row tile r solves its whole row against L11:
//           X_c = (A(r,c) - sum_{c'<c} X_c' L(c,c')^T) L_cc^-T, results chained in registers
//           (accumulator layout == next A-fragment layout), rhs row updated at the end.
// grid k_fL11 = (1, fronts in level, n_frames); k_fL21 = (max boundary tiles, fronts, frames)
// S (tile being factored; later the B operand of the trailing update), M (its inverse), the four
// diagonal-block inverses, three 16x16 scratch blocks, two vectors: 80 960 B, two workgroups per CU
#define L11_LDS_DOUBLES (2 * TILE + 7 * 256 + 2 * NB + 16)   // S, M, dinv, wt, vec | part (= the factorisation's exchange buffer), 32 ints

__global__ void __launch_bounds__(256, 2) k_fL11(const FrameDev* __restrict__ frames, LevelRef lvl,
                                               double u_override) {
  extern __shared__ double lds[];
  double* S = lds;
  double* M = lds + TILE;
  double* Bl = S;                  // S is dead once its inverse M exists
  double* dinv = lds + 2 * TILE;
  double* wt = dinv + 4 * 256;     // 3 blocks: inverse_assemble64 runs on at most 3 waves; diag16 uses 128 doubles
  double* vec = wt + 3 * 256;      // NB: rhs tile in / y tile out
  double* part = vec + NB;         // NB scratch
  int* s_ok = reinterpret_cast<int*>(part + NB);
  int* pf = s_ok + 8;              // 16 hand-off flags of the pipelined tile factorisation
  const FrameDev& fd = frames[blockIdx.z];
  if (!fd.bound || !fd.nd_ready) return;   // (a stopped slot only wastes the work: no dependent flag load here)
  int fi;
  if (!level_front(fd, lvl, blockIdx.y, fi)) return;
  const NDFront& f = fd.fronts[fi];
  if (f.npt == 0) return;
  const double u = (u_override >= 0.0) ? u_override : fd.st->u;
  double* vecs = fd.fvec + f.vec_off;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;

  for (int c = 0; c < f.npt; ++c) {
    // ---- diagonal tile: factor + inverse + forward substitution of rhs tile c ----
    double rhs_c = 0.0;
    {
      const double* src = ftile(fd, f, c, c);
      double v[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = src[threadIdx.x + 256 * t];
      if (threadIdx.x < NB) rhs_c = vecs[(size_t)c * NB + threadIdx.x];   // (kept in a register: vec | part is the factorisation's exchange buffer)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int e = threadIdx.x + 256 * t;
        const int i = e % NB, k = e / NB;
        double x = (i >= k) ? v[t] : 0.0;
        if (i == k) x = (c * NB + i < f.n1) ? x + u : 1.0;
        S[i + k * LD] = x;
      }
    }
    __syncthreads();
    // factor + inverse in the pipelined form of the task graph (wave 0 runs the pivot chain, waves 1-3 trail with the
    // panel / trailing / inverse blocks): 11.7 us per tile against ~14 for potrf64 + inverse_assemble64
    const bool ok = factor_inverse64p(S, M, dinv, wt, vec, s_ok, pf, nullptr, nullptr, nullptr, min(4, (f.n1 - c * NB + 15) >> 4));
    if (!ok && threadIdx.x == 0) fd.st->chol_fail = 1;
    if (threadIdx.x < NB) vec[threadIdx.x] = rhs_c;
    __syncthreads();
    {
      double* linv = fd.flinv + f.linv_off + (size_t)c * TILE;
      for (int e = threadIdx.x; e < TILE; e += blockDim.x) linv[e] = M[e];
      if (threadIdx.x < NB) {
        const int i = threadIdx.x;
        double acc = 0.0;
        for (int k = 0; k <= i; ++k) acc += M[i + k * LD] * vec[k];
        part[i] = acc;
      }
      __syncthreads();
      if (threadIdx.x < NB) {
        vec[threadIdx.x] = part[threadIdx.x];                       // y_c
        vecs[(size_t)c * NB + threadIdx.x] = part[threadIdx.x];
      }
      __syncthreads();
    }
    // ---- row solves inside the pivot block: L(r,c) = A(r,c) L_cc^-T, rhs_r -= L(r,c) y_c ----
    for (int r = c + 1; r < f.npt; ++r) {
      double* At = ftile(fd, f, r, c);
      double areg[16];
      load_a_frags(At, areg);
      double4_t acc[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[ni] = double4_t{0.0, 0.0, 0.0, 0.0};
      tile_ABt_regs<false, true>(areg, M, acc);
      store_c_frags(At, acc);
      // rhs rows of this wave: sum_col L[row][col] y[col], reduced over the 4 lk lanes
      double sacc = 0.0;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) sacc += acc[ni][rr] * vec[16 * ni + lk + 4 * rr];
      sacc += __shfl_xor(sacc, 16, 64);
      sacc += __shfl_xor(sacc, 32, 64);
      if (lk == 0) vecs[(size_t)r * NB + 16 * w + lr] -= sacc;
    }
    __syncthreads();   // L(r,c) tiles visible to the whole workgroup
    // ---- trailing update inside the pivot block: A(r,s) -= L(r,c) L(s,c)^T ----
    for (int sc = c + 1; sc < f.npt; ++sc) {
      const double* Ls = ftile(fd, f, sc, c);
      double breg[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) breg[e] = Ls[threadIdx.x + 256 * e];
      __syncthreads();   // previous users of Bl are done
#pragma unroll
      for (int e = 0; e < 16; ++e) Bl[threadIdx.x + 256 * e] = breg[e];
      __syncthreads();
      for (int r = sc; r < f.npt; ++r) {
        double* Ct = ftile(fd, f, r, sc);
        double areg[16];
        load_a_frags(ftile(fd, f, r, c), areg);
        double4_t acc[4];
        load_c_frags(Ct, acc);
        // (a diagonal tile is only ever read in its lower triangle: wave w leaves the blocks right of its own alone)
        if (r == sc) tile_ABt_regs_trim<true>(areg, Bl, acc, 4, w + 1);
        else tile_ABt_regs<true>(areg, Bl, acc);
        store_c_frags(Ct, acc);
      }
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256, 3) k_fL21(const FrameDev* __restrict__ frames, LevelRef lvl, WgMap map) {
  __shared__ double Bl[TILE];
  __shared__ double yv[NB];
  WgId wg;
  if (!wg_decode(map, wg)) return;
  const FrameDev& fd = frames[wg.frame];
  if (!fd.bound || !fd.nd_ready) return;   // (a stopped slot only wastes the work: no dependent flag load here)
  int fi;
  if (!level_front(fd, lvl, wg.front, fi)) return;
  const NDFront& f = fd.fronts[fi];
  const int r = f.npt + wg.unit;
  if (f.npt == 0 || r >= f.nt) return;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
  double* vecs = fd.fvec + f.vec_off;
  // known-zero parts are skipped: a wave whose 16 rows lie beyond the true boundary size issues no MFMA and stores
  // nothing (those rows of the pivot columns stay zero), and the front's last pivot tile column only counts up to the
  // true pivot count (blocks of 16)
  const bool wave_on = 16 * w < min(NB, 7 * f.nb - NB * wg.unit);
  // (X_c' of the earlier columns is re-read from the tile this thread itself stored it to -- the accumulator layout is
  //  the A-fragment layout, element for element -- instead of being held in 32 registers per column: two workgroups
  //  per CU instead of one)
  double rhs_acc = 0.0;
  for (int c = 0; c < f.npt; ++c) {
    {
      const int ncol = min(NB, f.n1 - NB * c);            // true pivots of this tile column
      const int nblk = (ncol + 15) >> 4;                  // 16-column blocks with true pivots (also the inner blocks of X_c L_cc^-T)
      double* At = ftile(fd, f, r, c);
      double4_t acc[4];
      if (wave_on) {
        load_c_frags(At, acc);
      } else {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = double4_t{0.0, 0.0, 0.0, 0.0};
      }
      // acc -= X_c' L(c,c')^T for the earlier pivot columns
      for (int cp = 0; cp < c; ++cp) {
        const double* Lt = ftile(fd, f, c, cp);
        double breg[16], xreg[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) breg[e] = Lt[threadIdx.x + 256 * e];
        if (wave_on) load_a_frags(ftile(fd, f, r, cp), xreg);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) Bl[threadIdx.x + 256 * e] = breg[e];
        __syncthreads();
        if (wave_on) tile_ABt_regs_trim<true>(xreg, Bl, acc, 4, nblk);
      }
      // X_c = acc L_cc^-T
      {
        const double* linv = fd.flinv + f.linv_off + (size_t)c * TILE;
        double breg[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) breg[e] = linv[threadIdx.x + 256 * e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) Bl[threadIdx.x + 256 * e] = breg[e];
        if (threadIdx.x < NB) yv[threadIdx.x] = vecs[(size_t)c * NB + threadIdx.x];
        __syncthreads();
        double areg[16];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) areg[4 * ni + rr] = acc[ni][rr];
        double4_t xa[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) xa[ni] = double4_t{0.0, 0.0, 0.0, 0.0};
        if (wave_on) {
          tile_ABt_regs_trim<false, true>(areg, Bl, xa, nblk, nblk);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            if (ni < nblk) {
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) At[(16 * w + lr) + (size_t)(16 * ni + lk + 4 * rr) * NB] = xa[ni][rr];
            }
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) rhs_acc += xa[ni][rr] * yv[16 * ni + lk + 4 * rr];
      }
    }
  }
  rhs_acc += __shfl_xor(rhs_acc, 16, 64);
  rhs_acc += __shfl_xor(rhs_acc, 32, 64);
  if (lk == 0 && wave_on) vecs[(size_t)r * NB + 16 * w + lr] -= rhs_acc;
}

// Trailing update of tile column c, restricted to the PIVOT tile columns that are still to be
// factored (s < npt): A(r,s) -= L(r,c) L(s,c)^T for c < s < npt, s <= r < nt, plus the rhs
// rows b_r -= L(r,c) y_c for all r > c.  The boundary x boundary block is updated once per
// front by k_fschur with the whole pivot block as the inner dimension.
// Block index t: columns b = s-c = 1..bcap, rows a = r-c = b..mcap, then mcap rhs blocks.
// grid = (ntile_cap + mcap, fronts in level, n_frames)
__global__ void __launch_bounds__(256) k_ftrail(const FrameDev* __restrict__ frames, LevelRef lvl,
                                                 int c, int mcap, int bcap, int ntile_cap, WgMap map) {
  __shared__ double Bl[TILE];
  WgId wg;
  if (!wg_decode(map, wg)) return;
  const FrameDev& fd = frames[wg.frame];
  if (!fd.bound || !fd.nd_ready) return;   // (a stopped slot only wastes the work: no dependent flag load here)
  int fi;
  if (!level_front(fd, lvl, wg.front, fi)) return;
  const NDFront& f = fd.fronts[fi];
  if (c >= f.npt) return;
  const int m = f.nt - 1 - c;
  int t = wg.unit;
  if (t < ntile_cap) {
    int db = 1;
    while (db <= bcap && t >= mcap - db + 1) {
      t -= mcap - db + 1;
      ++db;
    }
    const int da = db + t;
    if (da > m || c + db >= f.npt) return;
    const double* Lr = ftile(fd, f, c + da, c);
    const double* Ls = ftile(fd, f, c + db, c);
    double* Ct = ftile(fd, f, c + da, c + db);
    double breg[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) breg[e] = Ls[threadIdx.x + 256 * e];
    double areg[16];
    load_a_frags(Lr, areg);
    double4_t acc[4];
    load_c_frags(Ct, acc);
#pragma unroll
    for (int e = 0; e < 16; ++e) Bl[threadIdx.x + 256 * e] = breg[e];
    __syncthreads();
    tile_ABt_regs<true>(areg, Bl, acc);
    store_c_frags(Ct, acc);
  } else {
    const int db = t - ntile_cap + 1;
    if (db > m) return;
    __shared__ double y[NB];
    __shared__ double part[4][NB];
    const double* Ls = ftile(fd, f, c + db, c);
    double* vecs = fd.fvec + f.vec_off;
    if (threadIdx.x < NB) y[threadIdx.x] = vecs[(size_t)c * NB + threadIdx.x];
    __syncthreads();
    const int i = threadIdx.x & 63, q = threadIdx.x >> 6;
    double acc = 0.0;
#pragma unroll
    for (int k = 16 * q; k < 16 * q + 16; ++k) acc += Ls[i + k * NB] * y[k];
    part[q][i] = acc;
    __syncthreads();
    if (threadIdx.x < NB)
      vecs[(size_t)(c + db) * NB + i] -= part[0][i] + part[1][i] + part[2][i] + part[3][i];
  }
}

// Children's updates into the PIVOT columns of the fronts of a level (before their factorisation): one workgroup per
// pivot-column tile (r, c), c < npt, that some child maps into (the plan's pull items).
// grid = (pull items of the level x frames, rounded to 8)
__global__ void __launch_bounds__(256) k_fpull(const FrameDev* __restrict__ frames, int level, int n_items, int items_at,
                                                int n_frames) {
  __shared__ int2 maps[256];
  const FrameDev* fdp;
  const NDTileItem* itp = tile_item(frames, level, 1, n_items, items_at, n_frames, fdp);
  if (!itp) return;
  const FrameDev& fd = *fdp;
  const NDTileItem it = item_snapshot(itp);
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
  // only the 16 x 16 blocks that hold real scalars move: the padding of the last pivot tile column / of the last tile
  // row of the pivots or of the boundary stays zero (nothing maps into it)
  const int vrow = it.r < it.npt ? min(NB, it.n1 - NB * it.r) : min(NB, it.n2 - NB * (it.r - it.npt));
  const int nblk = (min(NB, it.n1 - NB * it.c) + 15) >> 4;
  const bool wave_on = 16 * w < vrow;
  pull_maps(fd, it, maps);
  double* T = item_tile(fd, it, it.r, it.c);
  // pure-fill tile (FrameDev::tile_kind): nothing was assembled into it and nothing zeroed it -- start from zero and store
  // the WHOLE tile (its padding too: the factor kernels read full tiles)
  const bool pure = __builtin_amdgcn_readfirstlane((int)fd.tile_kind[it.pad0]) != 0;
  double4_t acc[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) acc[ni] = double4_t{0.0, 0.0, 0.0, 0.0};
  if (wave_on && !pure) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      if (ni < nblk) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[ni][rr] = T[(16 * w + lr) + (size_t)(16 * ni + lk + 4 * rr) * NB];
      }
  }
  __syncthreads();
  if (wave_on) pull_tile(fd, it, maps, acc);
  if (pure) {
    store_c_frags(T, acc);
  } else if (wave_on) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      if (ni < nblk) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) T[(16 * w + lr) + (size_t)(16 * ni + lk + 4 * rr) * NB] = acc[ni][rr];
      }
  }
  if (it.r == it.c) {
    const double v = pull_vec(fd, it, maps);
    if (threadIdx.x < NB && v != 0.0) fd.fvec[it.vec_off + (size_t)it.c * NB + threadIdx.x] += v;
  }
}

// Update matrix of a front in one pass: U(r,s) = [children's updates] - sum_{c < npt} L(r,c) L(s,c)^T for the boundary
// tiles npt <= s <= r < nt, stored IN PLACE in the front's boundary block (plain stores; the parent gathers it).
// The next pivot column's operands are fetched while the current one is on the MFMA.  Known-zero parts are skipped:
// the inner dimension stops at the front's true pivot count (the padding columns of L21 are zero), and a wave whose
// 16 rows lie beyond the true boundary size, or a 16-column block beyond it, issues no MFMA (its outputs are padding
// that nothing reads).
// grid = (Schur items of the level x frames, rounded to 8)
__global__ void __launch_bounds__(256, 3) k_fschur(const FrameDev* __restrict__ frames, int level, int n_items,
                                                 int items_at, int n_frames) {
  __shared__ __attribute__((aligned(16))) double Bl[TILE];
  __shared__ int2 maps[256];
  const FrameDev* fdp;
  const NDTileItem* itp = tile_item(frames, level, 0, n_items, items_at, n_frames, fdp);
  if (!itp) return;
  const FrameDev& fd = *fdp;
  // (fields read through the pointer where they are used, like a front descriptor: a local copy of the record -- in
  //  vector or scalar registers -- lengthens live ranges by 40 VGPRs in this kernel and costs the fourth wave per SIMD)
  const NDTileItem& it = *itp;
  const int r = it.r, sc = it.c, tr = r - it.npt, tc = sc - it.npt;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
  // true sizes: rows of this tile row / columns of this tile column that are boundary scalars
  const int vrow = min(NB, it.n2 - NB * tr), vcol = min(NB, it.n2 - NB * tc);
  const bool wave_on = 16 * w < vrow;
  // 16-column blocks with real columns -- and, in a DIAGONAL tile of the (symmetric) update matrix, not beyond this wave's
  // own row block: the blocks strictly above the diagonal are never read (the gathers take row >= column only), so they
  // are neither computed nor stored (6 of a full tile's 16 blocks)
  const int nblk = (tr == tc) ? min((vcol + 15) >> 4, w + 1) : (vcol + 15) >> 4;
  const bool kids = it.kid[0].front >= 0 || it.kid[1].front >= 0;
  pull_maps(fd, it, maps);
  // The B operand (L21 tile of block-row sc) passes through LDS in HALF tiles of 32 inner columns,
  // double-buffered (2 x 16 KB): three workgroups per CU instead of two, and the next half is in
  // flight while the current one is on the MFMA.
  // (B half tiles move with 16 bytes per lane: half the load / LDS-store instructions, and a workgroup streams a tile
  //  1.5x faster that way -- tools/micro/tile_stream_mb.hip)
  typedef double dvec2 __attribute__((ext_vector_type(2)));
  dvec2 breg[4];
  double areg[16], acur[16];
  if (it.npt > 0) {
    const dvec2* Ls = reinterpret_cast<const dvec2*>(item_tile(fd, it, sc, 0));
#pragma unroll
    for (int e = 0; e < 4; ++e) breg[e] = Ls[threadIdx.x + 256 * e];
    if (wave_on) load_a_frags(item_tile(fd, it, r, 0), areg);
  }
  double4_t acc[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) acc[ni] = double4_t{0.0, 0.0, 0.0, 0.0};
  __syncthreads();     // maps visible
  if (kids) pull_tile(fd, it, maps, acc);
  for (int c = 0; c < it.npt; ++c) {
    const dvec2* Ls = reinterpret_cast<const dvec2*>(item_tile(fd, it, sc, c));
    dvec2* Bl2 = reinterpret_cast<dvec2*>(Bl);
    const int ksteps = min(16, (it.n1 - NB * c + 3) >> 2);   // inner steps of 4 with real pivots in this tile column
    // ---- inner columns 0..31 ----
#pragma unroll
    for (int e = 0; e < 4; ++e) Bl2[threadIdx.x + 256 * e] = breg[e];
#pragma unroll
    for (int e = 0; e < 16; ++e) acur[e] = areg[e];
    if (ksteps > 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) breg[e] = Ls[TILE / 4 + threadIdx.x + 256 * e];
    }
    if (c + 1 < it.npt && wave_on) load_a_frags(item_tile(fd, it, r, c + 1), areg);
    __syncthreads();
    if (wave_on) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks < ksteps) {
          const double av = -acur[ks];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            if (ni < nblk)
              acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bl[(16 * ni + lr) + (4 * ks + lk) * LD], av, acc[ni], 0, 0, 0);
        }
      }
    }
    // ---- inner columns 32..63 ----
    if (ksteps > 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) Bl2[TILE / 4 + threadIdx.x + 256 * e] = breg[e];
    }
    if (c + 1 < it.npt) {
      const dvec2* Ln = reinterpret_cast<const dvec2*>(item_tile(fd, it, sc, c + 1));
#pragma unroll
      for (int e = 0; e < 4; ++e) breg[e] = Ln[threadIdx.x + 256 * e];
    }
    __syncthreads();
    if (wave_on) {
#pragma unroll
      for (int ks = 8; ks < 16; ++ks) {
        if (ks < ksteps) {
          const double av = -acur[ks];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            if (ni < nblk)
              acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bl[(16 * ni + lr) + (4 * ks + lk) * LD], av, acc[ni], 0, 0, 0);
        }
      }
    }
  }
  // the update tile, in place (the parent gathers it): only the 16 x 16 blocks that hold boundary scalars -- the padding
  // of the last tile row / column is never read by anything (the pull maps address true scalars only)
  if (wave_on) {
    double* Cg = item_tile(fd, it, r, sc);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      if (ni < nblk) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Cg[(16 * w + lr) + (size_t)(16 * ni + lk + 4 * rr) * NB] = acc[ni][rr];
      }
  }
  // vector rows: diagonal tiles add the children's rows to v_r (k_fL21 / k_ftrail subtracted sum_c L(r,c) y_c before)
  if (tr == tc && kids) {
    const double v = pull_vec(fd, it, maps);
    if (threadIdx.x < NB && v != 0.0) fd.fvec[it.vec_off + (size_t)r * NB + threadIdx.x] += v;
  }
}

// Back substitution, part 1: y_c -= sum over boundary tiles r of L(r,c)^T x_r, where x of
// the boundary nodes is read from the global solution (their fronts are done already).
// grid = (max npt, fronts in level, n_frames)
__global__ void __launch_bounds__(256) k_fback_prep(const FrameDev* __restrict__ frames, LevelRef lvl) {
  extern __shared__ double xb[];   // n2p doubles
  const FrameDev& fd = frames[blockIdx.z];
  if (!fd.bound || !fd.nd_ready) return;   // (a stopped slot only wastes the work: no dependent flag load here)
  int fi;
  if (!level_front(fd, lvl, blockIdx.y, fi)) return;
  const NDFront& f = fd.fronts[fi];
  const int c = blockIdx.x;
  if (c >= f.npt || f.nb == 0) return;
  const int* nodes = fd.nd_nodes + f.nodes_off + f.nv;
  for (int i = threadIdx.x; i < f.n2p; i += blockDim.x)
    xb[i] = (i < 7 * f.nb) ? fd.delta[7 * nodes[i / 7] + i % 7] : 0.0;
  __syncthreads();
  const int n = threadIdx.x >> 2, q = threadIdx.x & 3;
  double acc = 0.0;
  for (int r = f.npt; r < f.nt; ++r) {
    const double* Lt = ftile(fd, f, r, c);
    const double* xr = xb + (size_t)(r - f.npt) * NB;
#pragma unroll
    for (int mrow = 16 * q; mrow < 16 * q + 16; ++mrow) acc += Lt[mrow + n * NB] * xr[mrow];
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  if (q == 0) fd.fvec[f.vec_off + (size_t)c * NB + n] -= acc;
}

// Back substitution, part 2, pivot tile column c = npt-1-step of every front of the level:
// x_c = L_cc^-T y_c, y_(c-d) -= L(c,c-d)^T x_c; block 0 scatters x_c to the global solution.
// grid = (max npt, fronts in level, n_frames)
__global__ void __launch_bounds__(256) k_fbacksub(const FrameDev* __restrict__ frames, LevelRef lvl,
                                                   int step) {
  const FrameDev& fd = frames[blockIdx.z];
  if (!fd.bound || !fd.nd_ready) return;   // (a stopped slot only wastes the work: no dependent flag load here)
  int fi;
  if (!level_front(fd, lvl, blockIdx.y, fi)) return;
  const NDFront& f = fd.fronts[fi];
  const int c = f.npt - 1 - step;
  if (c < 0) return;
  const int d = blockIdx.x;
  if (d > c) return;
  __shared__ double y[NB];
  __shared__ double x[NB];
  __shared__ double part[4][NB];
  double* vecs = fd.fvec + f.vec_off;
  const double* linv = fd.flinv + f.linv_off + (size_t)c * TILE;
  if (threadIdx.x < NB) y[threadIdx.x] = vecs[(size_t)c * NB + threadIdx.x];
  __syncthreads();
  {
    const int k = threadIdx.x & 63, q = threadIdx.x >> 6;
    double acc = 0.0;
#pragma unroll
    for (int i = 16 * q; i < 16 * q + 16; ++i) acc += linv[i + k * NB] * y[i];
    part[q][k] = acc;
    __syncthreads();
    if (threadIdx.x < NB) x[k] = part[0][k] + part[1][k] + part[2][k] + part[3][k];
    __syncthreads();
  }
  if (d == 0) {
    if (threadIdx.x < NB) {
      const int i = c * NB + threadIdx.x;
      if (i < f.n1) {
        const int node = fd.nd_nodes[f.nodes_off + i / 7];
        fd.delta[7 * node + i % 7] = x[threadIdx.x];
      }
    }
  } else {
    const double* Lt = ftile(fd, f, c, c - d);
    const int n = threadIdx.x >> 2, q = threadIdx.x & 3;
    double acc = 0.0;
#pragma unroll
    for (int mrow = 16 * q; mrow < 16 * q + 16; ++mrow) acc += Lt[mrow + n * NB] * x[mrow];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (q == 0) vecs[(size_t)(c - d) * NB + n] -= acc;
  }
}

// ---- host launchers --------------------------------------------------------------------
void launch_pair_reduce(const FrameDev* fr, int n_frames, int max_blocks, hipStream_t st) {
  if (max_blocks <= 0) return;
  hipLaunchKernelGGL(k_pair_reduce, dim3((max_blocks + 3) / 4, n_frames), dim3(256), 0, st, fr);
}

void launch_pair_scatter(const FrameDev* fr, int n_frames, int max_blocks, hipStream_t st) {
  if (max_blocks <= 0) return;
  hipLaunchKernelGGL(k_pair_scatter, dim3((max_blocks + 3) / 4, n_frames), dim3(256), 0, st, fr);
}

void launch_front_assemble(const FrameDev* fr, int n_frames, int max_blocks, hipStream_t st) {
  if (max_blocks <= 0) return;
  hipLaunchKernelGGL(k_front_assemble, dim3((max_blocks + 3) / 4, n_frames), dim3(256), 0, st, fr);
}

void launch_reg_grad_nd(const FrameDev* fr, int n_frames, int maxJ, int use_arap, double lam_a,
                        int use_rot, double lam_r, hipStream_t st) {
  if (maxJ <= 0 || (!use_arap && !use_rot)) return;
  hipLaunchKernelGGL(k_reg_grad_nd, dim3((maxJ * RG_LANES + 255) / 256, n_frames), dim3(256), 0, st, fr, use_arap,
                     lam_a, use_rot, lam_r);
}

// (only when no regulariser is enabled: k_reg_grad_nd loads the front vectors itself)
void launch_front_load_rhs(const FrameDev* fr, int n_frames, int maxP, hipStream_t st) {
  hipLaunchKernelGGL(k_front_load_rhs, dim3((maxP + 255) / 256, n_frames), dim3(256), 0, st, fr);
}

void launch_iter_begin_nd(const FrameDev* fr, int n_frames, hipStream_t st, const int* reuse, int dag_cut) {
  hipLaunchKernelGGL(k_iter_begin_nd, dim3(1024, n_frames), dim3(256), 0, st, fr, reuse, dag_cut);
}

// Level schedule shared by all slots of a batch (they may have different plans: the host
// passes, per level, the maxima over the batch; blocks beyond a front's own size exit).
// Factorisation launches of levels [0, l_factor_end), back-substitution launches of levels [0, l_back_end), the
// deepest level being 0.  The whole solve is (n_levels, 0) followed by (0, n_levels); the hybrid solve of a batch
// runs (l_cut, 0), the task-graph kernel for the levels >= l_cut, then (0, l_cut).
void launch_front_levels(const FrameDev* fr, int n_frames, const NDLevelSched* lv, int n_levels, int l_factor_end,
                         int l_back_end, double u_override, hipStream_t st) {
  const size_t lds = PANEL_LDS_DOUBLES * sizeof(double);
  const size_t lds11 = L11_LDS_DOUBLES * sizeof(double);
  // the dynamic-LDS limit is a property of a kernel ON a device: set once per device id
  // (host threads may drive different solvers on one device: the flags are atomics; a device id beyond the table sets
  //  the attributes on every call instead of aliasing another device's flag)
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  const bool tracked = dev >= 0 && dev < 64;
  if (!tracked || !attr_set[dev].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute((const void*)k_fL11, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds11) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_fpanel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_fpotrf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return;   // (sticky HIP error: the caller's hipGetLastError reports it)
    if (tracked) attr_set[dev].store(true, std::memory_order_release);
  }
  for (int l = 0; l < l_factor_end; ++l) {
    const NDLevelSched& s = lv[l];
    if (s.n_fronts <= 0) continue;
    const LevelRef lr{l, s.first, s.n_fronts};
    // One workgroup per front (k_fL11 + k_fL21) only pays when a level has enough fronts to fill the
    // chip: with few fronts the per-column panel launches, which spread a front over its row tiles, have
    // the shorter critical path (C2: 581 instead of 543 it/s at one frame per launch; the cross-over is
    // around 128 fronts x frames; 64 since round 3, with k_fL11 at two workgroups per CU and the leaner k_fL21: C2 at 8
    // frames 2.133 instead of 2.147 ms per solve).  SLM_COMPACT_MIN / SLM_COMPACT_NPT override the thresholds for
    // experiments (wider pivot blocks than 4 tile columns lose: 2.22 ms with the 5-column level in this form).
    static const long compact_min = [] {
      const char* e = getenv("SLM_COMPACT_MIN");
      return e ? atol(e) : 64L;
    }();
    static const int compact_npt = [] {
      const char* e = getenv("SLM_COMPACT_NPT");
      return e ? atoi(e) : 4;
    }();
    const bool compact = s.max_npt <= compact_npt && (long)s.n_fronts * n_frames >= compact_min;
    // children's update matrices into the pivot columns (levels whose fronts all are leaves have nothing to gather).
    // A launch of its own: gathering inside k_fL11 / k_fL21 (at the first touch of every pivot-column tile) was built
    // and measured -- one launch fewer per level, but the gathers then sit on the critical path of workgroups that run
    // two per CU (C2, 8 frames: k_fL11 44 -> 83 us and k_fL21 56 -> 114 us at the level with 256 fronts, against 44 us
    // for this kernel, which runs thousands of small workgroups at four per SIMD)
    if (s.n_pull > 0) {
      const int total = s.n_pull * n_frames;
      hipLaunchKernelGGL(k_fpull, dim3((unsigned)((total + 7) / 8 * 8)), dim3(256), 0, st, fr, l, s.n_pull, s.pull_at, n_frames);
    }
    if (compact) {
      hipLaunchKernelGGL(k_fL11, dim3(1, s.n_fronts, n_frames), dim3(256), lds11, st, fr, lr, u_override);
      if (s.max_n2p > 0) {
        const WgMap m = make_map(s.max_n2p / 64, s.n_fronts, n_frames);
        hipLaunchKernelGGL(k_fL21, map_grid(m), dim3(256), 0, st, fr, lr, m);
      }
    }
    for (int c = 0; !compact && c < s.max_npt; ++c) {
      const int mcap = s.max_nt - 1 - c;
      // fused panel while the launch is small (latency-bound); split once the redundant
      // factorisations would take more than ~2 blocks per CU
      if ((long)(mcap + 1) * s.n_fronts * n_frames <= 512) {
        const WgMap m = make_map(mcap + 1, s.n_fronts, n_frames);
        hipLaunchKernelGGL(k_fpanel, map_grid(m), dim3(256), lds, st, fr, lr, c, u_override, m);
      } else {
        hipLaunchKernelGGL(k_fpotrf, dim3(1, s.n_fronts, n_frames), dim3(256), lds, st, fr, lr, c,
                           u_override);
        if (mcap > 0) {
          const WgMap m = make_map(mcap, s.n_fronts, n_frames);
          hipLaunchKernelGGL(k_ftrsm, map_grid(m), dim3(256), 0, st, fr, lr, c, m);
        }
      }
      const int bcap = std::min(mcap, s.max_npt - 1 - c);
      int ntile = 0;
      for (int b = 1; b <= bcap; ++b) ntile += mcap - b + 1;
      if (ntile + mcap > 0) {
        const WgMap m = make_map(ntile + mcap, s.n_fronts, n_frames);
        hipLaunchKernelGGL(k_ftrail, map_grid(m), dim3(256), 0, st, fr, lr, c, mcap, bcap, ntile, m);
      }
    }
    // update matrices of this level, stored in place (the parents gather them: k_fpull / k_fschur of the next level,
    // or the task graph's own pulls): one launch, every tile written by exactly one workgroup
    if (s.n_schur > 0) {
      const int total = s.n_schur * n_frames;
      hipLaunchKernelGGL(k_fschur, dim3((unsigned)((total + 7) / 8 * 8)), dim3(256), 0, st, fr, l, s.n_schur, s.schur_at,
                         n_frames);
    }
  }
  for (int l = l_back_end - 1; l >= 0; --l) {
    const NDLevelSched& s = lv[l];
    if (s.n_fronts <= 0 || s.max_npt <= 0) continue;
    const LevelRef lr{l, s.first, s.n_fronts};
    if (s.max_n2p > 0)
      hipLaunchKernelGGL(k_fback_prep, dim3(s.max_npt, s.n_fronts, n_frames), dim3(256),
                         (size_t)s.max_n2p * sizeof(double), st, fr, lr);
    for (int e = 0; e < s.max_npt; ++e)
      hipLaunchKernelGGL(k_fbacksub, dim3(s.max_npt, s.n_fronts, n_frames), dim3(256), 0, st, fr,
                         lr, e);
  }
}

void launch_front_solve(const FrameDev* fr, int n_frames, const NDLevelSched* lv, int n_levels,
                        double u_override, hipStream_t st) {
  launch_front_levels(fr, n_frames, lv, n_levels, n_levels, 0, u_override, st);
  launch_front_levels(fr, n_frames, lv, n_levels, 0, n_levels, u_override, st);
}
