// slm_begin.h -- start of an LM iteration on the multifrontal path: zero the assembled pivot-column tiles of the fronts, the
// front vectors, rhs and the counters; reset the task graph's flags and mailboxes.  A device function (round 6) so that it
// can run as k_iter_begin_nd (slm_front.hip) or as the tail blocks of the Jacobian pass's launch (k_data_gram, slm_data_v1.hip:
// the two touch disjoint memory -- records vs fronts -- and the zeroing's store stream hides under the Gram pass's arithmetic).
#pragma once
#include "slm_tile.h"

// ---- zeroing of the fronts ------------------------------------------------------------------------------------
// A front's tiles: the pivot columns (the assembly adds into them, then the children's updates: k_fpull / the task
// graph's pulls), then the boundary block F22 -- two thirds of the tile storage (C2: 134 MB per frame) -- which holds
// the front's update matrix, written once by k_fschur / the SCHUR tasks (first touch is a store: never zeroed).
typedef double dvec2_t __attribute__((ext_vector_type(2)));
// 16 KB pieces of a contiguous region, piece-strided: pc = first, first + step, ...; non-temporal 16-byte stores
__device__ __forceinline__ void zero_pieces(double* base, size_t n_pieces, size_t first, size_t step) {
  const dvec2_t zz = {0.0, 0.0};
  dvec2_t* b2 = reinterpret_cast<dvec2_t*>(base);
  for (size_t pc = first; pc < n_pieces; pc += step) {
    dvec2_t* q = b2 + pc * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(zz, q + 256 * k);
  }
}
__device__ __forceinline__ size_t front_piv_tiles(const NDFront& f) {
  return (size_t)f.npt * f.nt - (size_t)f.npt * (f.npt - 1) / 2;
}
// the slot's kind-0 pivot-column tiles (FrameDev::zero_tiles: assembled into, or touched by nothing at all), 16 KB pieces
// strided over the launch's workgroups; pure-fill tiles (tile_kind 1) are not zeroed: their first toucher stores them
__device__ __forceinline__ void zero_fronts(const FrameDev& fd, int b, int nb) {
  const dvec2_t zz = {0.0, 0.0};
  const int n = fd.n_zero_tiles;
  for (int pc = b; pc < 2 * n; pc += nb) {
    dvec2_t* q = reinterpret_cast<dvec2_t*>(fd.ftiles.get() + fd.zero_tiles[pc >> 1]) + (size_t)(pc & 1) * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(zz, q + 256 * k);
  }
}
// b / nb: this workgroup's index among the nb workgroups that share the slot's zeroing (256 threads each).
// dag_cut: what the task-graph launch of this iteration's solve needs reset -- its flags (ticket, abort, per-tile and
// per-column flags) and the mailboxes of the pivot tile columns it factors: -1 all fronts (whole-tree task graph),
// >= 0 the fronts of depth <= dag_cut (hybrid form), -2 none (per-level launches only).  A stopped slot is reset too: slot
// 0's flags carry the ticket of the whole batch.
// reused: this slot's Jacobian pass is skipped (records reused after a rejected step).  gram_sets_count: the Jacobian pass
// of THIS launch publishes m_grad itself for the slots it runs for (k_data_gram block 0) -- the zeroing must not race it.
__device__ __forceinline__ void iter_begin_nd_body(const FrameDev& fd, int b, int nb, bool reused, int dag_cut, bool gram_sets_count) {
  if (!fd.bound) return;
  if (dag_cut >= -1 && fd.nd_ready && fd.dag_flags) {
    for (int i = b * blockDim.x + threadIdx.x; i < fd.dag_n_flags; i += nb * blockDim.x) fd.dag_flags[i] = 0;
    typedef __attribute__((address_space(1))) long long gll;
    gll* mail = (gll*)(double*)fd.fmail;
    for (int fi = b; fi < fd.n_fronts; fi += nb) {
      const NDFront& f = fd.fronts[fi];
      if (dag_cut >= 0 && f.depth > dag_cut) continue;
      const size_t base = (size_t)(f.linv_off / TILE) * SLM_MAIL_DOUBLES, n = (size_t)f.npt * SLM_MAIL_DOUBLES;
      for (size_t e = threadIdx.x; e < n; e += blockDim.x) mail[base + e] = SLM_MAIL_EMPTY;
    }
  }
  if (fd.st->stopped) return;
  const size_t tid = (size_t)b * blockDim.x + threadIdx.x, nthr = (size_t)nb * blockDim.x;
  const double2 z = make_double2(0.0, 0.0);
  if (fd.nd_ready) {
    zero_fronts(fd, b, nb);
    double2* v2 = reinterpret_cast<double2*>(fd.fvec.get());
    const size_t nv2 = (size_t)fd.zero_vec_doubles / 2;
    for (size_t e = tid; e < nv2; e += nthr) v2[e] = z;
  }
  const size_t nrhs = (size_t)fd.nt * SLM_NB;
  for (size_t e = tid; e < nrhs; e += nthr) fd.rhs[e] = 0.0;
  if (b == 0 && threadIdx.x == 0) {
    // (a reused Jacobian pass keeps its matched count -- on a surfel-sharded frame the rank's OWN share of it, which
    //  k_pair_scatter set aside before it stored the all-reduced count)
    if (!reused) {
      if (!gram_sets_count) fd.st->m_grad = 0;   // (k_data_gram takes the count of the evaluation it consumes)
    } else if (fd.pairbuf) {
      fd.st->m_grad = fd.st->m_grad_local;
    }
    fd.st->chol_fail = 0;
  }
}
