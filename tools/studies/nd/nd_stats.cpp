// Host-only harness around the symbolic analysis (python-super_amd/csrc/slm_nd_host.hip): builds the plan of a coupling
// graph handed over by tools/studies/nd_order_study.py and returns its cost figures.  Study tool, not part of the library.
#include <cstdint>
#include <cstring>
#include "slm_nd.h"

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
