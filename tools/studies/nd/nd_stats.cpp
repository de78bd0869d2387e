// Host-only harness around the symbolic analysis (python-super_amd/csrc/slm_nd_host.hip): builds the plan of a coupling
// graph handed over by tools/studies/nd_order_study.py and returns its cost figures.  Study tool, not part of the library.
#include <cstdint>
#include <cstring>
#include <vector>
#include "slm_nd.h"

// Is a task list a valid order for the ticket scheduler of slm_dag.hip (a task only waits for EARLIER tasks)?  Checks, per
// front: POTRF(s) behind POTRF(s-1); POTRF(s) behind the producers of L(s,c) and L(s-1,c), c <= s-2; COL(r,s) behind POTRF(s)
// and behind the producers of L(r,c), L(s,c), c < s; SCHUR(r,sc) behind the producers of L(r,c), L(sc,c), c < npt; a front's
// first task behind the SCHUR tasks of its children; BACK / BACKB behind the front's POTRF tasks and its parent's BACK.
// The producer of tile (r,c), c < npt, r > c: POTRF(r) when r == c + 1 < npt (it owns the tile left of its diagonal one),
// else COL(r,c).  Returns 0, or the 1-based index of the first task that precedes something it waits for.
static int check_order(const NDPlanHost& p, const std::vector<int32_t>& list, bool top_only) {
  const int T = (int)p.fronts.size();
  const size_t n = list.size() / 2;
  std::vector<std::vector<long>> potrf(T), col(T), schur(T);
  std::vector<long> back(T, -1), first(T, -1);
  auto key = [](int r, int s, int nt) { return (size_t)r * nt + s; };
  for (int i = 0; i < T; ++i) {
    const NDFront& f = p.fronts[i];
    potrf[i].assign(f.npt > 0 ? f.npt : 1, -1);
    col[i].assign((size_t)f.nt * f.nt + 1, -1);
    schur[i].assign((size_t)f.nt * f.nt + 1, -1);
  }
  for (size_t k = 0; k < n; ++k) {
    const int type = list[2 * k] >> 24, fi = list[2 * k] & 0xFFFFFF, r = list[2 * k + 1] >> 8, s = list[2 * k + 1] & 255;
    if (fi < 0 || fi >= T) return (int)k + 1;
    const NDFront& f = p.fronts[fi];
    if (first[fi] < 0 && type <= ND_T_SCHUR) first[fi] = (long)k;
    if (type == ND_T_POTRF) potrf[fi][s] = (long)k;
    else if (type == ND_T_COL) col[fi][key(r, s, f.nt)] = (long)k;
    else if (type == ND_T_SCHUR) schur[fi][key(r, s, f.nt)] = (long)k;
    else if (type == ND_T_BACK) back[fi] = (long)k;
  }
  auto producer = [&](int fi, int r, int c) -> long {
    const NDFront& f = p.fronts[fi];
    return (r == c + 1 && r < f.npt) ? potrf[fi][r] : col[fi][key(r, c, f.nt)];
  };
  for (size_t k = 0; k < n; ++k) {
    const int type = list[2 * k] >> 24, fi = list[2 * k] & 0xFFFFFF, r = list[2 * k + 1] >> 8, s = list[2 * k + 1] & 255;
    const NDFront& f = p.fronts[fi];
    const bool listed = !top_only || f.depth <= p.dag_cut_depth;    // (a top list holds no factor tasks of the deeper fronts)
    auto before = [&](long dep) { return dep >= 0 && dep < (long)k; };
    if (type == ND_T_POTRF) {
      if (s > 0 && !before(potrf[fi][s - 1])) return (int)k + 1;
      for (int c = 0; c + 1 < s; ++c)
        if (!before(producer(fi, s, c)) || !before(producer(fi, s - 1, c))) return (int)k + 1;
    } else if (type == ND_T_COL) {
      if (!before(potrf[fi][s])) return (int)k + 1;
      for (int c = 0; c < s; ++c)
        if (!before(producer(fi, r, c)) || !before(producer(fi, s, c))) return (int)k + 1;
    } else if (type == ND_T_SCHUR) {
      for (int c = 0; c < f.npt; ++c)
        if (!before(producer(fi, r, c)) || (s != r && !before(producer(fi, s, c)))) return (int)k + 1;
      if (f.npt > 0 && !before(potrf[fi][f.npt - 1])) return (int)k + 1;
    } else if (type == ND_T_BACK || type == ND_T_BACKB) {
      if (listed)
        for (int c = 0; c < f.npt; ++c)
          if (!before(potrf[fi][c])) return (int)k + 1;
      if (f.parent >= 0 && !before(back[f.parent])) return (int)k + 1;
    }
    if (type <= ND_T_SCHUR && (long)k == first[fi])
      for (int kid = 0; kid < 2; ++kid) {
        const int ch = p.front_kids[2 * (size_t)fi + kid];
        if (ch < 0) continue;
        if (top_only && p.fronts[ch].depth > p.dag_cut_depth) continue;   // factored by the per-level launches before the list runs
        for (long q : schur[ch])
          if (q >= (long)k) return (int)k + 1;
        for (long q : potrf[ch])
          if (p.fronts[ch].npt > 0 && !before(q)) return (int)k + 1;
      }
  }
  return 0;
}

// 0: both task lists of the plan are valid ticket orders; else 1000000 * list + the offending task's 1-based index
extern "C" int nd_check_orders(int J, int K_ED, const float* pts, const int32_t* knn, const uint32_t* pairs, int n_pairs) {
  NDPlanHost p;
  if (!nd_build_plan(J, K_ED, pts, knn, pairs, n_pairs, p)) return -1;
  const int a = check_order(p, p.dag_tasks, false);
  if (a) return 1000000 + a;
  const int b = p.dag_top_tasks.empty() ? 0 : check_order(p, p.dag_top_tasks, true);
  return b ? 2000000 + b : 0;
}

extern "C" int nd_stats(int J, int K_ED, const float* pts, const int32_t* knn, const uint32_t* pairs, int n_pairs,
                        double* out, int32_t* fronts, int max_fronts, int32_t* kp_out) {
  NDPlanHost p;
  if (!nd_build_plan(J, K_ED, pts, knn, pairs, n_pairs, p)) return -1;
  out[0] = p.flops;
  out[1] = p.flops_exact;
  out[2] = (double)p.fronts.size();
  out[3] = (double)(p.level_start.size() - 1);
  out[4] = p.dag_critical_us;
  out[5] = (double)p.tile_doubles * 8.0;
  out[6] = (double)p.dag_tasks.size() / 2;
  out[7] = (double)p.tile_items.size();
  int n = 0;
  for (const NDFront& f : p.fronts) {
    if (n >= max_fronts) break;
    fronts[4 * n + 0] = f.depth;
    fronts[4 * n + 1] = f.nv;
    fronts[4 * n + 2] = f.nb;
    fronts[4 * n + 3] = f.parent;
    if (kp_out) {   // boundary nodes of the front that are PIVOTS of its parent (they come first: elimination order)
      int kp = 0;
      if (f.parent >= 0)
        for (int b = 0; b < f.nb; ++b) kp += p.eamap[f.eamap_off + b] < p.fronts[f.parent].nv;
      kp_out[n] = kp;
    }
    ++n;
  }
  return n;
}
