// slm_graph.hip -- "next" row f3: ED-graph construction at frame 0 (reference init_graph +
// DirectDeformGraph grid_mesh, super/graph_encoder.py:11-67,128-195).
//   k_gr_anchor      anchor flags on the pixel grid (valid pixels at multiples of `step`)
//   rocPRIM scans    node numbers (row-major), edge and triangle positions (anchor-major, kind-minor)
//   k_gr_nodes       node positions / normals gathered through data.index_map
//   k_gr_cells       per anchor: 4 edges (right, diagonal, down, anti-diagonal) and 2 triangles,
//                    kept when every vertex is a valid anchor; lengths and rest areas
//   k_gr_radii       radius = mean length of the incident edges, gathered in edge order (deterministic)
//   k_gr_fix_nan     isolated nodes get the mean radius of the others
// Semantic-SuPer (slm_graph_init_semantic): the nodes' class confidences are gathered with their rows
// (class = first maximum); with hard_seg + mesh_face an edge / triangle is only kept when its vertices
// share a class (graph_encoder.py:134-151).
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <string>

#include "slm_common.h"

void slm_set_error_text(const char* msg);   // slm_api.hip

namespace {

#define GCHK(expr)                                                        \
  do {                                                                    \
    hipError_t e_ = (expr);                                               \
    if (e_ != hipSuccess) {                                               \
      slm_set_error_text((std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
      for (void* p_ : owned)                                              \
        if (p_) (void)hipFree(p_);                                        \
      return SLM_ERR_HIP;                                                 \
    }                                                                     \
  } while (0)

struct Grid {
  int H, W, step, gw, gh;   // gw x gh grid points: u = 0, step, ... < W-1; v likewise < H-1
};

struct GrSem {               // segmentation fields (C == 0: none)
  int C, prune;
  const double* seg_conf;    // (T,C) data.seg_conf
  int32_t* node_seg;         // (cap_nodes)
  double* node_seg_conf;     // (cap_nodes,C)
};

// the anchor at grid cell (gx, gy), or -1: valid pixel and inside the grid
__device__ __forceinline__ int anchor_id(const Grid& g, const int32_t* __restrict__ node_of, int gx, int gy) {
  if (gx < 0 || gy < 0 || gx >= g.gw || gy >= g.gh) return -1;
  return node_of[gy * g.gw + gx];
}

__global__ void __launch_bounds__(256) k_gr_anchor(Grid g, const uint8_t* __restrict__ valid, int32_t* __restrict__ flag) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= g.gw * g.gh) return;
  const int gx = c % g.gw, gy = c / g.gw;
  flag[c] = valid[(size_t)(gy * g.step) * g.W + gx * g.step] ? 1 : 0;
}

// node_of[cell] = node number or -1; node rows
__global__ void __launch_bounds__(256) k_gr_nodes(Grid g, const int32_t* __restrict__ flag, const int32_t* __restrict__ pos,
                                                   const int32_t* __restrict__ index_map, const double* __restrict__ points,
                                                   const double* __restrict__ norms, int32_t* __restrict__ node_of,
                                                   slm_graph_outputs o, GrSem sm) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= g.gw * g.gh) return;
  if (!flag[c]) {
    node_of[c] = -1;
    return;
  }
  const int k = pos[c];
  node_of[c] = k;
  const int gx = c % g.gw, gy = c / g.gw;
  const int row = index_map[(size_t)(gy * g.step) * g.W + gx * g.step];
  for (int a = 0; a < 3; ++a) {
    o.points[3 * (size_t)k + a] = points[3 * (size_t)row + a];
    o.norms[3 * (size_t)k + a] = norms[3 * (size_t)row + a];
  }
  if (sm.C > 0) {
    int best = 0;
    for (int a = 0; a < sm.C; ++a) {
      const double v = sm.seg_conf[(size_t)sm.C * row + a];
      sm.node_seg_conf[(size_t)sm.C * k + a] = v;
      if (v > sm.seg_conf[(size_t)sm.C * row + best]) best = a;
    }
    sm.node_seg[k] = best;
  }
}

// the 4 candidate edges of cell (gx,gy): (a, b) as node numbers, -1 when dropped
__device__ __forceinline__ void cell_edges(const Grid& g, const int32_t* __restrict__ node_of, int gx, int gy, int ea[4], int eb[4]) {
  const int s = anchor_id(g, node_of, gx, gy), p1 = anchor_id(g, node_of, gx + 1, gy), p2 = anchor_id(g, node_of, gx + 1, gy + 1),
            p3 = anchor_id(g, node_of, gx, gy + 1);
  ea[0] = s; eb[0] = p1;      // right
  ea[1] = s; eb[1] = p2;      // diagonal
  ea[2] = s; eb[2] = p3;      // down
  ea[3] = p1; eb[3] = p3;     // anti-diagonal (belongs to the cell of s)
  for (int k = 0; k < 4; ++k)
    if (s < 0 || ea[k] < 0 || eb[k] < 0) ea[k] = eb[k] = -1;
}

// per anchor cell: flags of its 4 edges and 2 triangles (positions come from scans over these)
__global__ void __launch_bounds__(256) k_gr_cell_flags(Grid g, const int32_t* __restrict__ node_of, int J,
                                                        const int32_t* __restrict__ cell_of_node, int32_t* __restrict__ eflag,
                                                        int32_t* __restrict__ tflag, GrSem sm) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= J) return;
  const int c = cell_of_node[k], gx = c % g.gw, gy = c / g.gw;
  const bool prune = sm.C > 0 && sm.prune;
  const int32_t* seg = sm.node_seg;
  int ea[4], eb[4];
  cell_edges(g, node_of, gx, gy, ea, eb);
  for (int q = 0; q < 4; ++q) eflag[4 * k + q] = (ea[q] >= 0 && (!prune || seg[ea[q]] == seg[eb[q]])) ? 1 : 0;
  const int p1 = anchor_id(g, node_of, gx + 1, gy), p2 = anchor_id(g, node_of, gx + 1, gy + 1), p3 = anchor_id(g, node_of, gx, gy + 1);
  bool t0 = p1 >= 0 && p2 >= 0, t1 = p2 >= 0 && p3 >= 0;    // (s, pt1, pt2), (s, pt2, pt3)
  if (prune) {
    t0 = t0 && seg[k] == seg[p1] && seg[k] == seg[p2];
    t1 = t1 && seg[k] == seg[p2] && seg[k] == seg[p3];
  }
  tflag[2 * k] = t0 ? 1 : 0;
  tflag[2 * k + 1] = t1 ? 1 : 0;
}

__global__ void __launch_bounds__(256) k_gr_cell_of_node(Grid g, const int32_t* __restrict__ node_of, int32_t* __restrict__ cell_of_node) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= g.gw * g.gh) return;
  if (node_of[c] >= 0) cell_of_node[node_of[c]] = c;
}

__device__ __forceinline__ double dist3(const double* P, int a, int b) {
  const double dx = P[3 * (size_t)a] - P[3 * (size_t)b], dy = P[3 * (size_t)a + 1] - P[3 * (size_t)b + 1],
               dz = P[3 * (size_t)a + 2] - P[3 * (size_t)b + 2];
  return sqrt(dx * dx + dy * dy + dz * dz);
}

__global__ void __launch_bounds__(256) k_gr_cells(Grid g, const int32_t* __restrict__ node_of, int J,
                                                   const int32_t* __restrict__ cell_of_node, const int32_t* __restrict__ eflag,
                                                   const int32_t* __restrict__ epos, const int32_t* __restrict__ tflag,
                                                   const int32_t* __restrict__ tpos, slm_graph_outputs o) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= J) return;
  const int c = cell_of_node[k], gx = c % g.gw, gy = c / g.gw;
  const size_t es = 4 * (size_t)o.cap_nodes, ts = 2 * (size_t)o.cap_nodes;
  int ea[4], eb[4];
  cell_edges(g, node_of, gx, gy, ea, eb);
  for (int q = 0; q < 4; ++q)
    if (eflag[4 * k + q]) {
      const int e = epos[4 * k + q];
      o.edge_index[e] = ea[q];
      o.edge_index[es + e] = eb[q];
      o.edges_lens[e] = dist3(o.points, ea[q], eb[q]);
    }
  const int p1 = anchor_id(g, node_of, gx + 1, gy), p2 = anchor_id(g, node_of, gx + 1, gy + 1), p3 = anchor_id(g, node_of, gx, gy + 1);
  const int tv[2][2] = {{p1, p2}, {p2, p3}};
  for (int q = 0; q < 2; ++q)
    if (tflag[2 * k + q]) {
      const int t = tpos[2 * k + q];
      const int v1 = tv[q][0], v2 = tv[q][1];
      o.triangles[t] = k;
      o.triangles[ts + t] = v1;
      o.triangles[2 * ts + t] = v2;
      const double* P = o.points;
      const double ax = P[3 * (size_t)v1] - P[3 * (size_t)k], ay = P[3 * (size_t)v1 + 1] - P[3 * (size_t)k + 1],
                   az = P[3 * (size_t)v1 + 2] - P[3 * (size_t)k + 2];
      const double bx = P[3 * (size_t)v2] - P[3 * (size_t)k], by = P[3 * (size_t)v2 + 1] - P[3 * (size_t)k + 1],
                   bz = P[3 * (size_t)v2 + 2] - P[3 * (size_t)k + 2];
      const double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
      o.triangles_areas[t] = 0.5 * sqrt(cx * cx + cy * cy + cz * cz + 1e-13);
    }
}

// radius of node k = mean length of its incident edges, visited in ascending edge number:
// cells (gx-1,gy-1): diag; (gx,gy-1): down, anti-diag; (gx-1,gy): right, anti-diag; (gx,gy): right, diag, down
__global__ void __launch_bounds__(256) k_gr_radii(Grid g, const int32_t* __restrict__ node_of, int J,
                                                   const int32_t* __restrict__ cell_of_node, const int32_t* __restrict__ eflag,
                                                   const int32_t* __restrict__ epos, slm_graph_outputs o,
                                                   double* __restrict__ sum_out, int32_t* __restrict__ cnt_out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= J) return;
  const int c = cell_of_node[k], gx = c % g.gw, gy = c / g.gw;
  const int cells[4][2] = {{gx - 1, gy - 1}, {gx, gy - 1}, {gx - 1, gy}, {gx, gy}};
  const unsigned kinds[4] = {1u << 1, (1u << 2) | (1u << 3), (1u << 0) | (1u << 3), (1u << 0) | (1u << 1) | (1u << 2)};
  double s = 0.0;
  int n = 0;
  for (int a = 0; a < 4; ++a) {
    const int nb = anchor_id(g, node_of, cells[a][0], cells[a][1]);
    if (nb < 0) continue;
    for (int q = 0; q < 4; ++q)
      if (((kinds[a] >> q) & 1u) && eflag[4 * nb + q]) {
        const int e = epos[4 * nb + q];
        if (o.edge_index[e] == k || o.edge_index[4 * (size_t)o.cap_nodes + e] == k) {
          s += o.edges_lens[e];
          ++n;
        }
      }
  }
  o.radii[k] = n > 0 ? s / n : nan("");
  if (n > 0) {
    atomicAdd(cnt_out, 1);
    unsafeAtomicAdd(sum_out, s / n);
  }
}

__global__ void __launch_bounds__(256) k_gr_fix_nan(int J, double* __restrict__ radii, const double* __restrict__ sum,
                                                     const int32_t* __restrict__ cnt) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < J && isnan(radii[k])) radii[k] = *sum / (double)*cnt;
}

}  // namespace

static int graph_init_impl(int32_t H, int32_t W, int32_t step, const uint8_t* valid, const int32_t* index_map,
                           const double* points, const double* norms, const slm_graph_outputs* out, GrSem sm,
                           int32_t* counts_host, void* stream) {
  if (H < 2 || W < 2 || step < 1 || !valid || !index_map || !points || !norms || !out || !out->points || !out->norms ||
      !out->radii || !out->edge_index || !out->edges_lens || !out->triangles || !out->triangles_areas) {
    slm_set_error_text("slm_graph_init: bad argument");
    return SLM_ERR_INVALID;
  }
  Grid g{H, W, step, (W - 1 + step - 1) / step, (H - 1 + step - 1) / step};
  const int cells = g.gw * g.gh;
  if (cells < 1 || out->cap_nodes < cells) {
    slm_set_error_text("slm_graph_init: cap_nodes is smaller than the anchor grid");
    return SLM_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  void* owned[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int32_t *flag = nullptr, *pos = nullptr, *node_of = nullptr, *cell_of = nullptr, *eflag = nullptr, *epos = nullptr;
  double* acc = nullptr;
  void* tmp = nullptr;
  const size_t nmax = 4 * (size_t)cells;
  GCHK(hipMalloc((void**)&flag, sizeof(int32_t) * 6 * (size_t)cells)); owned[0] = flag;
  GCHK(hipMalloc((void**)&pos, sizeof(int32_t) * 6 * (size_t)cells)); owned[1] = pos;
  GCHK(hipMalloc((void**)&node_of, sizeof(int32_t) * (size_t)cells)); owned[2] = node_of;
  GCHK(hipMalloc((void**)&cell_of, sizeof(int32_t) * (size_t)cells)); owned[3] = cell_of;
  GCHK(hipMalloc((void**)&eflag, sizeof(int32_t) * 6 * (size_t)cells)); owned[4] = eflag;
  GCHK(hipMalloc((void**)&epos, sizeof(int32_t) * 6 * (size_t)cells)); owned[5] = epos;
  GCHK(hipMalloc((void**)&acc, sizeof(double) * 2)); owned[6] = acc;
  size_t bytes = 0;
  GCHK(rocprim::exclusive_scan(nullptr, bytes, flag, pos, 0, nmax, rocprim::plus<int32_t>(), st));
  GCHK(hipMalloc(&tmp, bytes)); owned[7] = tmp;
  const dim3 blk(256), gc((cells + 255) / 256);
  // anchors -> node numbers
  hipLaunchKernelGGL(k_gr_anchor, gc, blk, 0, st, g, valid, flag);
  size_t b1 = bytes;
  GCHK(rocprim::exclusive_scan(tmp, b1, flag, pos, 0, (size_t)cells, rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(k_gr_nodes, gc, blk, 0, st, g, flag, pos, index_map, points, norms, node_of, *out, sm);
  int32_t last[2];
  GCHK(hipMemcpyAsync(&last[0], pos + cells - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  GCHK(hipMemcpyAsync(&last[1], flag + cells - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  GCHK(hipStreamSynchronize(st));
  const int J = last[0] + last[1];
  int E = 0, F = 0;
  if (J > 0) {
    const dim3 gj((J + 255) / 256);
    hipLaunchKernelGGL(k_gr_cell_of_node, gc, blk, 0, st, g, node_of, cell_of);
    int32_t* tflag = eflag + 4 * (size_t)cells;
    int32_t* tpos = epos + 4 * (size_t)cells;
    hipLaunchKernelGGL(k_gr_cell_flags, gj, blk, 0, st, g, node_of, J, cell_of, eflag, tflag, sm);
    b1 = bytes;
    GCHK(rocprim::exclusive_scan(tmp, b1, eflag, epos, 0, 4 * (size_t)J, rocprim::plus<int32_t>(), st));
    b1 = bytes;
    GCHK(rocprim::exclusive_scan(tmp, b1, tflag, tpos, 0, 2 * (size_t)J, rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(k_gr_cells, gj, blk, 0, st, g, node_of, J, cell_of, eflag, epos, tflag, tpos, *out);
    GCHK(hipMemsetAsync(acc, 0, sizeof(double) * 2, st));
    hipLaunchKernelGGL(k_gr_radii, gj, blk, 0, st, g, node_of, J, cell_of, eflag, epos, *out, acc,
                       reinterpret_cast<int32_t*>(acc + 1));
    hipLaunchKernelGGL(k_gr_fix_nan, gj, blk, 0, st, J, out->radii, acc, reinterpret_cast<const int32_t*>(acc + 1));
    int32_t le[4];
    GCHK(hipMemcpyAsync(&le[0], epos + 4 * (size_t)J - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GCHK(hipMemcpyAsync(&le[1], eflag + 4 * (size_t)J - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GCHK(hipMemcpyAsync(&le[2], tpos + 2 * (size_t)J - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GCHK(hipMemcpyAsync(&le[3], tflag + 2 * (size_t)J - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GCHK(hipStreamSynchronize(st));
    E = le[0] + le[1];
    F = le[2] + le[3];
  }
  GCHK(hipGetLastError());
  if (counts_host) {
    counts_host[0] = J;
    counts_host[1] = E;
    counts_host[2] = F;
  }
  for (void* p : owned)
    if (p) (void)hipFree(p);
  return SLM_OK;
}

extern "C" int slm_graph_init(int32_t H, int32_t W, int32_t step, const uint8_t* valid, const int32_t* index_map,
                              const double* points, const double* norms, const slm_graph_outputs* out,
                              int32_t* counts_host, void* stream) {
  return graph_init_impl(H, W, step, valid, index_map, points, norms, out, GrSem{0, 0, nullptr, nullptr, nullptr},
                         counts_host, stream);
}

extern "C" int slm_graph_init_semantic(int32_t H, int32_t W, int32_t step, const uint8_t* valid, const int32_t* index_map,
                                       const double* points, const double* norms, int32_t num_classes,
                                       const double* seg_conf, int32_t prune_class_edges, const slm_graph_outputs* out,
                                       int32_t* node_seg, double* node_seg_conf, int32_t* counts_host, void* stream) {
  if (num_classes < 1 || num_classes > SLM_MAX_CLASSES || !seg_conf || !node_seg || !node_seg_conf) {
    slm_set_error_text("slm_graph_init_semantic: bad argument (1..4 classes, seg_conf and both node outputs)");
    return SLM_ERR_INVALID;
  }
  return graph_init_impl(H, W, step, valid, index_map, points, norms, out,
                         GrSem{num_classes, prune_class_edges ? 1 : 0, seg_conf, node_seg, node_seg_conf}, counts_host,
                         stream);
}
