// slm_nd_host.hip -- host-side symbolic analysis for the nested-dissection multifrontal
// solver (see slm_nd.h).  Runs once per frame inside slm_bind_frame on a node graph of
// J ~ 10^3 vertices (milliseconds); everything numeric runs on the device.
#include <algorithm>
#include <cstdlib>
#include <cstdint>
#include <functional>
#include <numeric>
#include <vector>

#include "slm_nd.h"

namespace {

struct TreeNode {
  std::vector<int> vars;
  int child[2] = {-1, -1};
  int parent = -1;
  int depth = 0;
};

struct Builder {
  int J;
  const float* pts;
  std::vector<int> adj_start, adj;   // CSR adjacency (duplicates allowed: every use is a membership test)
  std::vector<TreeNode> tree;
  std::vector<char> side;   // scratch: 1 = A, 2 = B
  std::vector<int> idx_scratch;   // scratch: index of a node inside its boundary layer
  int leaf_max = SLM_ND_LEAF;     // stop bisecting at this many nodes
  static bool use_cover() {
    static const bool on = [] {
      const char* e = getenv("SLM_ND_COVER");
      return e ? atoi(e) != 0 : true;
    }();
    return on;
  }

  int dissect(std::vector<int>& nodes, int depth) {
    const int id = (int)tree.size();
    tree.emplace_back();
    tree[id].depth = depth;
    if ((int)nodes.size() <= leaf_max) {
      tree[id].vars = nodes;
      return id;
    }
    // widest coordinate axis, median split (only the partition matters: nth_element, ties by node id)
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (int v : nodes)
      for (int a = 0; a < 3; ++a) {
        lo[a] = std::min(lo[a], pts[3 * v + a]);
        hi[a] = std::max(hi[a], pts[3 * v + a]);
      }
    int ax = 0;
    for (int a = 1; a < 3; ++a)
      if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a;
    const size_t half = nodes.size() / 2;
    std::nth_element(nodes.begin(), nodes.begin() + half, nodes.end(), [&](int x, int y) {
      const float px = pts[3 * x + ax], py = pts[3 * y + ax];
      return px < py || (px == py && x < y);
    });
    for (size_t i = 0; i < nodes.size(); ++i) side[nodes[i]] = i < half ? 1 : 2;
    // vertex separator: the nodes of one side that touch the other side (smaller choice)
    std::vector<int> SA, SB;
    for (size_t i = 0; i < nodes.size(); ++i) {
      const int v = nodes[i];
      const char other = side[v] == 1 ? 2 : 1;
      bool touches = false;
      for (int e = adj_start[v]; e < adj_start[v + 1]; ++e)
        if (side[adj[e]] == other) { touches = true; break; }
      if (touches) (side[v] == 1 ? SA : SB).push_back(v);
    }
    // separator: a minimum vertex cover of the cut edges (Koenig's theorem on the bipartite graph
    // between the two boundary layers) -- never larger than the smaller layer, and usually thinner
    std::vector<int> cover;
    if (use_cover()) {
      const int na = (int)SA.size(), nb2 = (int)SB.size();
      for (int i = 0; i < nb2; ++i) idx_scratch[SB[i]] = i;
      std::vector<std::vector<int>> ed(na);
      for (int i = 0; i < na; ++i)
        for (int e = adj_start[SA[i]]; e < adj_start[SA[i] + 1]; ++e)
          if (side[adj[e]] == 2) ed[i].push_back(idx_scratch[adj[e]]);
      std::vector<int> matchA(na, -1), matchB(nb2, -1);
      std::vector<char> seen;
      std::function<bool(int)> aug = [&](int u) -> bool {
        for (int v : ed[u]) {
          if (seen[v]) continue;
          seen[v] = 1;
          if (matchB[v] < 0 || aug(matchB[v])) {
            matchA[u] = v;
            matchB[v] = u;
            return true;
          }
        }
        return false;
      };
      for (int u = 0; u < na; ++u) {
        seen.assign(nb2, 0);
        aug(u);
      }
      // Z = unmatched A vertices and everything reachable from them by alternating paths;
      // cover = (A \ Z) + (B in Z)
      std::vector<char> za(na, 0), zb(nb2, 0);
      std::vector<int> stack;
      for (int u = 0; u < na; ++u)
        if (matchA[u] < 0) {
          za[u] = 1;
          stack.push_back(u);
        }
      while (!stack.empty()) {
        const int u = stack.back();
        stack.pop_back();
        for (int v : ed[u])
          if (!zb[v]) {
            zb[v] = 1;
            const int u2 = matchB[v];
            if (u2 >= 0 && !za[u2]) {
              za[u2] = 1;
              stack.push_back(u2);
            }
          }
      }
      for (int u = 0; u < na; ++u)
        if (!za[u]) cover.push_back(SA[u]);
      for (int v = 0; v < nb2; ++v)
        if (zb[v]) cover.push_back(SB[v]);
    }
    const std::vector<int>& S0 = SA.size() <= SB.size() ? SA : SB;
    const std::vector<int>& S = (!cover.empty() && cover.size() < S0.size()) ? cover : S0;
    for (int v : S) side[v] = 3;
    std::vector<int> A, B;
    A.reserve(half + 1);
    B.reserve(half + 1);
    for (int v : nodes) {
      if (side[v] == 1) A.push_back(v);
      else if (side[v] == 2) B.push_back(v);
    }
    std::vector<int> Sv = S;
    for (int v : nodes) side[v] = 0;
    if (A.empty() || B.empty()) {   // could not split: dense leaf
      tree[id].vars = nodes;
      return id;
    }
    std::sort(Sv.begin(), Sv.end());
    tree[id].vars = Sv;
    const int ca = dissect(A, depth + 1);
    const int cb = dissect(B, depth + 1);
    tree[id].child[0] = ca;
    tree[id].child[1] = cb;
    tree[ca].parent = id;
    tree[cb].parent = id;
    return id;
  }
};

inline int round64(int x) { return (x + 63) / 64 * 64; }

}  // namespace

bool nd_build_plan(int J, int K_ED, const float* pts, const int32_t* ed_knn, const uint32_t* pairs,
                   int n_pairs, NDPlanHost& out, int leaf_nodes) {
  if (J < 1) return false;
  Builder b;
  static const int leaf_env = [] {
    const char* e = getenv("SLM_ND_LEAF");
    return e ? atoi(e) : 0;
  }();
  b.leaf_max = leaf_env > 0 ? leaf_env : (leaf_nodes > 0 ? leaf_nodes : SLM_ND_LEAF);
  b.J = J;
  b.pts = pts;
  b.side.assign(J, 0);
  b.idx_scratch.assign(J, -1);
  // CSR adjacency in two counting passes over the pair list and the node KNN table
  auto valid_edge = [&](int x, int y) { return x != y && x >= 0 && y >= 0 && x < J && y < J; };
  b.adj_start.assign(J + 1, 0);
  for (int i = 0; i < n_pairs; ++i) {
    const int x = (int)(pairs[i] / (uint32_t)J), y = (int)(pairs[i] % (uint32_t)J);
    if (valid_edge(x, y)) { b.adj_start[x + 1]++; b.adj_start[y + 1]++; }
  }
  for (int j = 0; j < J; ++j)
    for (int k = 0; k < K_ED; ++k) {
      const int y = ed_knn[j * K_ED + k];
      if (valid_edge(j, y)) { b.adj_start[j + 1]++; b.adj_start[y + 1]++; }
    }
  for (int j = 0; j < J; ++j) b.adj_start[j + 1] += b.adj_start[j];
  b.adj.resize(b.adj_start[J]);
  {
    std::vector<int> fill(b.adj_start.begin(), b.adj_start.end() - 1);
    for (int i = 0; i < n_pairs; ++i) {
      const int x = (int)(pairs[i] / (uint32_t)J), y = (int)(pairs[i] % (uint32_t)J);
      if (valid_edge(x, y)) { b.adj[fill[x]++] = y; b.adj[fill[y]++] = x; }
    }
    for (int j = 0; j < J; ++j)
      for (int k = 0; k < K_ED; ++k) {
        const int y = ed_knn[j * K_ED + k];
        if (valid_edge(j, y)) { b.adj[fill[j]++] = y; b.adj[fill[y]++] = j; }
      }
  }
  // the graph may be disconnected: dissect() only needs the node list
  std::vector<int> all(J);
  std::iota(all.begin(), all.end(), 0);
  const int root = b.dissect(all, 0);
  (void)root;
  const int T = (int)b.tree.size();

  // ---- elimination order: post-order over the separator tree ----------------------
  std::vector<int> post;   // tree node ids, children before parents
  {
    std::vector<std::pair<int, int>> st;
    st.push_back({0, 0});
    while (!st.empty()) {
      auto& top = st.back();
      const int id = top.first;
      if (top.second < 2 && b.tree[id].child[top.second] >= 0) {
        const int c = b.tree[id].child[top.second];
        ++top.second;
        st.push_back({c, 0});
      } else if (top.second < 2 && b.tree[id].child[top.second] < 0) {
        ++top.second;
      } else {
        post.push_back(id);
        st.pop_back();
      }
    }
  }
  std::vector<int32_t>& order = out.order;
  std::vector<int32_t>& node_tree = out.node_tree;
  order.assign(J, -1);
  node_tree.assign(J, -1);
  int cnt = 0;
  for (int id : post)
    for (int v : b.tree[id].vars) {
      order[v] = cnt++;
      node_tree[v] = id;
    }
  if (cnt != J) return false;

  // ---- boundaries, bottom-up -----------------------------------------------------------
  std::vector<std::vector<int>> bnd(T);
  std::vector<char> mark(J, 0);
  for (int id : post) {
    std::vector<int> acc;
    auto push = [&](int w) {
      if (!mark[w]) { mark[w] = 1; acc.push_back(w); }
    };
    int my_max = -1;
    for (int v : b.tree[id].vars) my_max = std::max(my_max, order[v]);
    for (int v : b.tree[id].vars)
      for (int e = b.adj_start[v]; e < b.adj_start[v + 1]; ++e)
        if (order[b.adj[e]] > my_max) push(b.adj[e]);
    for (int c = 0; c < 2; ++c)
      if (b.tree[id].child[c] >= 0)
        for (int w : bnd[b.tree[id].child[c]])
          if (node_tree[w] != id) push(w);
    for (int w : acc) mark[w] = 0;
    std::sort(acc.begin(), acc.end(), [&](int x, int y) { return order[x] < order[y]; });
    bnd[id] = acc;
  }

  // ---- fronts in processing order: deepest level first ------------------------------------
  int max_depth = 0;
  for (auto& t : b.tree) max_depth = std::max(max_depth, t.depth);
  std::vector<int> proc;   // tree ids
  out.level_start.clear();
  for (int d = max_depth; d >= 0; --d) {
    out.level_start.push_back((int)proc.size());
    for (int id = 0; id < T; ++id)
      if (b.tree[id].depth == d) proc.push_back(id);
  }
  out.level_start.push_back((int)proc.size());
  std::vector<int32_t>& front_of_tree = out.front_of_tree;
  front_of_tree.assign(T, -1);
  for (int i = 0; i < (int)proc.size(); ++i) front_of_tree[proc[i]] = i;

  out.fronts.assign(T, NDFront());
  out.nodes.clear();
  out.eamap.clear();
  out.node_front.assign(J, -1);
  out.node_pos.assign(J, -1);
  out.tile_doubles = out.vec_doubles = out.linv_doubles = 0;
  out.max_nt = out.max_npt = out.max_level_fronts = 0;
  out.flops = out.flops_exact = 0.0;
  // local position of a node in a front: every node occurs in its own front and in the few descendants'
  // fronts that have it on their boundary -> a short (tree id, position) list per node
  std::vector<int32_t>& occ_start = out.occ_start;
  occ_start.assign(J + 1, 0);
  for (int id = 0; id < T; ++id) {
    for (int v : b.tree[id].vars) occ_start[v + 1]++;
    for (int v : bnd[id]) occ_start[v + 1]++;
  }
  for (int j = 0; j < J; ++j) occ_start[j + 1] += occ_start[j];
  std::vector<std::pair<int32_t, int32_t>>& occ = out.occ;
  occ.assign(occ_start[J], {-1, -1});
  std::vector<int> occ_fill(occ_start.begin(), occ_start.end() - 1);
  auto pos_in = [&](int tid, int node) -> int {
    for (int e = occ_start[node]; e < occ_fill[node]; ++e)
      if (occ[e].first == tid) return occ[e].second;
    return -1;
  };
  int64_t n_tiles_logical = 0;
  for (int i = 0; i < T; ++i) {
    const int id = proc[i];
    TreeNode& t = b.tree[id];
    // pivots in elimination order
    std::sort(t.vars.begin(), t.vars.end(), [&](int x, int y) { return order[x] < order[y]; });
    NDFront& f = out.fronts[i];
    f.nv = (int)t.vars.size();
    f.nb = (int)bnd[id].size();
    f.n1 = 7 * f.nv;
    f.n1p = round64(f.n1);
    f.n2p = round64(7 * f.nb);
    // the substitution tasks stage x of a front's boundary in LDS (slm_dag.hip BACKB: two tiles; slm_front.hip
    // k_fback_prep: n2p doubles of dynamic LDS): a coupling graph whose front has a wider boundary (> 1 170 nodes -- no
    // geometric ED graph comes near) is refused here and takes the block-banded solver
    if (f.n2p > ND_MAX_N2P) return false;
    f.nt = (f.n1p + f.n2p) / 64;
    f.npt = f.n1p / 64;
    f.depth = t.depth;
    f.parent = t.parent >= 0 ? front_of_tree[t.parent] : -1;
    f.which_child = 0;
    if (t.parent >= 0 && b.tree[t.parent].child[1] == id) f.which_child = 1;
    f.nodes_off = (int)out.nodes.size();
    int p = 0;
    for (int v : t.vars) {
      out.nodes.push_back(v);
      out.node_front[v] = i;
      out.node_pos[v] = p;
      occ[occ_fill[v]++] = {id, p++};
    }
    for (int v : bnd[id]) {
      out.nodes.push_back(v);
      occ[occ_fill[v]++] = {id, p++};
    }
    // storage: pivot-column tiles of every front and the boundary blocks of the internal fronts first (zeroed before
    // every assembly), the boundary blocks of the leaves behind them (see NDFront::f22_base)
    const int64_t piv_tiles = (int64_t)f.npt * f.nt - (int64_t)f.npt * (f.npt - 1) / 2;
    const int64_t all_tiles = (int64_t)f.nt * (f.nt + 1) / 2;
    f.is_leaf = (t.child[0] < 0 && t.child[1] < 0) ? 1 : 0;
    f.tile_first = (int32_t)n_tiles_logical;
    n_tiles_logical += all_tiles;
    f.tile_off = out.tile_doubles;
    f.f22_base = f.tile_off;
    out.tile_doubles += (f.is_leaf ? piv_tiles : all_tiles) * 4096;
    f.vec_off = out.vec_doubles;
    out.vec_doubles += (int64_t)f.nt * 64;
    f.linv_off = out.linv_doubles;
    out.linv_doubles += (int64_t)f.npt * 4096;
    out.max_nt = std::max(out.max_nt, f.nt);
    out.max_npt = std::max(out.max_npt, f.npt);
    const double n1 = f.n1p, n2 = f.n2p;
    out.flops += n1 * n1 * n1 / 3.0 + n1 * n1 * n2 + n1 * n2 * n2;
    const double e1 = f.n1, e2 = 7.0 * f.nb;
    out.flops_exact += e1 * e1 * e1 / 3.0 + e1 * e1 * e2 + e1 * e2 * e2;
  }
  out.tile_zero_doubles = out.tile_doubles;
  for (int i = 0; i < T; ++i) {
    NDFront& f = out.fronts[i];
    if (!f.is_leaf) continue;
    const int64_t piv = (int64_t)f.npt * f.nt - (int64_t)f.npt * (f.npt - 1) / 2;
    f.f22_base = out.tile_doubles - piv * 4096;
    out.tile_doubles += ((int64_t)f.nt * (f.nt + 1) / 2 - piv) * 4096;
  }
  out.sched.clear();
  for (size_t l = 0; l + 1 < out.level_start.size(); ++l) {
    NDLevelSched sc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    sc.first = out.level_start[l];
    sc.n_fronts = out.level_start[l + 1] - out.level_start[l];
    for (int i = out.level_start[l]; i < out.level_start[l + 1]; ++i) {
      const NDFront& f = out.fronts[i];
      sc.max_npt = std::max(sc.max_npt, f.npt);
      sc.max_nt = std::max(sc.max_nt, f.nt);
      sc.max_pairs = std::max(sc.max_pairs, f.parent >= 0 ? f.nb * (f.nb + 1) / 2 : 0);
      sc.max_n2p = std::max(sc.max_n2p, f.n2p);
    }
    out.sched.push_back(sc);
    out.max_level_fronts = std::max(out.max_level_fronts, sc.n_fronts);
  }
  // extend-add maps: boundary index of a child -> local node position in the parent
  for (int i = 0; i < T; ++i) {
    const int id = proc[i];
    NDFront& f = out.fronts[i];
    f.eamap_off = (int)out.eamap.size();
    if (b.tree[id].parent < 0) {
      if (f.nb != 0) return false;   // the root cannot have a boundary
      continue;
    }
    for (int v : bnd[id]) {
      const int pp = pos_in(b.tree[id].parent, v);
      if (pp < 0) return false;   // boundary must be covered by the parent front
      out.eamap.push_back(pp);
    }
  }

  // ---- task list of the persistent kernel (slm_dag.hip) -----------------------------------------
  // Pull maps: the children do not scatter their Schur complements into the parent; every task of the parent
  // GATHERS the entries of its own tile from the children's update tiles (fixed order: own, child 0, child 1).
  // pullmap[pull_off[c] + parent scalar index] = boundary scalar index of child c, or -1.
  out.front_kids.assign(2 * (size_t)T, -1);
  out.pull_off.assign((size_t)T, -1);
  out.pullmap.clear();
  for (int i = 0; i < T; ++i) {
    const NDFront& f = out.fronts[i];
    if (f.parent < 0 || f.nb == 0) continue;
    const NDFront& pf = out.fronts[f.parent];
    out.front_kids[2 * (size_t)f.parent + f.which_child] = i;
    out.pull_off[i] = (int32_t)out.pullmap.size();
    out.pullmap.resize(out.pullmap.size() + (size_t)pf.nt * 64, -1);
    int32_t* pm = out.pullmap.data() + out.pull_off[i];
    for (int bq = 0; bq < f.nb; ++bq) {
      const int pp = out.eamap[f.eamap_off + bq];
      const int base = pp < pf.nv ? 7 * pp : pf.n1p + 7 * (pp - pf.nv);
      for (int k = 0; k < 7; ++k) pm[base + k] = 7 * bq + k;
    }
  }
  // per (front, child, tile row of the front): range of the child's boundary tile rows that map into it,
  // lo | hi << 8, or -1: which update tiles of the children a task of the front has to wait for
  out.prng_off.assign((size_t)T, 0);
  out.prng.clear();
  for (int i = 0; i < T; ++i) {
    const NDFront& f = out.fronts[i];
    out.prng_off[i] = (int32_t)out.prng.size();
    for (int t = 0; t < f.nt; ++t)
      for (int k = 0; k < 2; ++k) {
        const int ch = out.front_kids[2 * (size_t)i + k];
        int lo = 1 << 20, hi = -1;
        if (ch >= 0) {
          const int32_t* pm = out.pullmap.data() + out.pull_off[ch];
          for (int e = 0; e < 64; ++e) {
            const int a = pm[64 * t + e];
            if (a >= 0) { lo = std::min(lo, a >> 6); hi = std::max(hi, a >> 6); }
          }
        }
        out.prng.push_back(hi >= 0 ? (lo | (hi << 8)) : -1);
      }
  }
  // ---- exact work lists of the pull-form per-level kernels (slm_front.hip k_fschur, k_fpull) --------------------
  {
    out.tile_items.clear();
    out.item_off.assign(1, 0);
    auto make_item = [&](int i, int r, int c) {
      const NDFront& f = out.fronts[i];
      NDTileItem it{};
      it.front = i; it.r = r; it.c = c; it.nt = f.nt; it.npt = f.npt; it.n1 = f.n1; it.n2 = 7 * f.nb;
      it.tile_off = f.tile_off; it.f22_base = f.f22_base; it.vec_off = f.vec_off;
      it.pad0 = f.tile_first + c * f.nt - c * (c - 1) / 2 + (r - c);   // the tile's number in the slot's per-tile tables (FrameDev::tile_kind)
      const int32_t* pr = out.prng.data() + out.prng_off[i];
      for (int k = 0; k < 2; ++k) {
        NDTileKid& kd = it.kid[k];
        kd.front = -1;
        const int ch = out.front_kids[2 * (size_t)i + k];
        if (ch < 0 || pr[2 * r + k] < 0 || pr[2 * c + k] < 0) continue;
        const NDFront& cf = out.fronts[ch];
        kd.front = ch; kd.nt = cf.nt; kd.npt = cf.npt; kd.pull_off = out.pull_off[ch];
        kd.f22_base = cf.f22_base; kd.vec_boundary = cf.vec_off + (int64_t)cf.npt * 64;
      }
      return it;
    };
    for (size_t l = 0; l + 1 < out.level_start.size(); ++l) {
      // Schur items: every boundary tile pair of every front with a parent
      for (int i = out.level_start[l]; i < out.level_start[l + 1]; ++i) {
        const NDFront& f = out.fronts[i];
        if (f.parent < 0) continue;
        if (f.nt > 255) return false;
        for (int tr = f.npt; tr < f.nt; ++tr)
          for (int tc = f.npt; tc <= tr; ++tc) out.tile_items.push_back(make_item(i, tr, tc));
      }
      out.item_off.push_back((int32_t)out.tile_items.size());
      // pull items: the pivot-column tiles some child maps into
      for (int i = out.level_start[l]; i < out.level_start[l + 1]; ++i) {
        const NDFront& f = out.fronts[i];
        if (f.is_leaf) continue;
        for (int c = 0; c < f.npt; ++c)
          for (int r = c; r < f.nt; ++r) {
            const NDTileItem it = make_item(i, r, c);
            if (it.kid[0].front >= 0 || it.kid[1].front >= 0) out.tile_items.push_back(it);
          }
      }
      out.item_off.push_back((int32_t)out.tile_items.size());
      out.sched[l].schur_at = out.item_off[2 * l];
      out.sched[l].n_schur = out.item_off[2 * l + 1] - out.item_off[2 * l];
      out.sched[l].pull_at = out.item_off[2 * l + 1];
      out.sched[l].n_pull = out.item_off[2 * l + 2] - out.item_off[2 * l + 1];
    }
  }
  // Earliest start times from a duration model (microseconds; only the ORDER matters): tasks sorted by
  // them are in a topological order, and workgroups that take tasks in that order find them ready
  // about when they get to them.  The model runs once per task list: the whole tree, and the top of the tree for the hybrid solve
  // with everything below the cut final at time 0.  (Through the first half of round 6 the top list was the whole-tree list
  // filtered by depth: the fronts of the cut level kept the start times of their SUBTREES, so the two with the smaller subtrees
  // were listed first, their waiting tasks held a frame's 64 workgroups, and the other two -- the longer chains, the ones their
  // parents wait for -- got theirs 50-120 us late: the 4-front level of C2 took 205 us at 8 frames per launch for 5 columns.)
  if (T >= (1 << 24) || out.max_nt > 255) return false;
  {
    static const int top_fronts = [] {
      const char* e = getenv("SLM_DAG_TOP_FRONTS");
      // (2 through round 4; with two workgroups per CU in the task graph the 4-front level pays as tasks too: C2, 8 frames,
      //  ms per LM iteration at 2 / 4 / 8 fronts per level: 2.448 / 2.397 / 2.429)
      return e && atoi(e) > 0 ? atoi(e) : 4;
    }();
    int max_depth = 0;
    for (int i = 0; i < T; ++i) max_depth = std::max(max_depth, (int)out.fronts[i].depth);
    std::vector<int> per_depth(max_depth + 1, 0);
    for (int i = 0; i < T; ++i) ++per_depth[out.fronts[i].depth];
    int cut = -1;
    while (cut + 1 <= max_depth && per_depth[cut + 1] <= top_fronts) ++cut;
    if (cut >= max_depth) cut = max_depth - 1;        // leave at least the deepest level to the launches
    out.dag_cut_depth = cut;
  }
  auto model_list = [&](const int top_cut, std::vector<int32_t>& list) {
    struct Task { double start; int32_t w0, w1; };
    std::vector<Task> tasks;
    const double HOP = 1.0;   // flag + payload hand-off between workgroups
    auto d_potrf = [](int s) { return 15.0 + 0.6 * s; };
    auto d_schur = [](int npt) { return 3.0 + 1.5 * npt; };
    std::vector<std::vector<double>> done_all(T);   // per front, per tile: when its final content is published
    std::vector<double> fact_done(T, 0.0);
    for (int i = 0; i < T; ++i) {                  // processing order: children before parents
      const NDFront& f = out.fronts[i];
      auto tix = [&](int r, int c) { return (size_t)c * f.nt - (size_t)c * (c - 1) / 2 + (size_t)(r - c); };
      std::vector<double>& done = done_all[i];
      done.assign((size_t)f.nt * (f.nt + 1) / 2, 0.0);
      // a list for the top of the tree (top_cut >= 0): the fronts below the cut are factored by the per-level launches BEFORE the
      // list runs -- their tiles are final at time 0 and they have no factor tasks here (only their substitution tasks, below)
      if (top_cut >= 0 && f.depth > top_cut) continue;
      // when are the children's update tiles that tile (r,s) of this front gathers from complete?
      auto pulled = [&](int r, int s) {
        double t = 0.0;
        const int32_t* pr = out.prng.data() + out.prng_off[i];   // (per tile row and child: the child's tile rows that map into it)
        for (int k = 0; k < 2; ++k) {
          const int ch = out.front_kids[2 * (size_t)i + k];
          if (ch < 0) continue;
          const int rr = pr[2 * r + k], cc = pr[2 * s + k];
          if (rr < 0 || cc < 0) continue;
          const NDFront& cf = out.fronts[ch];
          const int r0 = rr & 255, r1 = rr >> 8, c0 = cc & 255, c1 = cc >> 8;
          for (int cr = r0; cr <= r1; ++cr)
            for (int cc2 = c0; cc2 <= std::min(c1, cr); ++cc2) {
              const size_t ti = (size_t)(cf.npt + cc2) * cf.nt - (size_t)(cf.npt + cc2) * (cf.npt + cc2 - 1) / 2 + (size_t)(cr - cc2);
              t = std::max(t, done_all[ch][ti] + HOP);
            }
        }
        return t;
      };
      // Order inside a front: POTRF(0), POTRF(1), COL(., 0), POTRF(2), COL(., 1), ... -- the NEXT column's chain task is listed
      // before this column's row solves (round 5).  It needs nothing from them (column s + 1 reads the columns <= s - 1 and
      // follows the factor of column s through its mailbox), and listed behind them it takes its ticket only when a workgroup
      // comes free: with several frames per launch a column's ~13 COL tasks per front hold the workgroups while they wait
      // for the factor, the next POTRF starts when they are done, and its 15-30 us of own work (gathers, earlier columns) run
      // AFTER the factor it should have followed instead of beside it (C2, 8 frames: 20-40 us per column against 14.4).
      // The keys are made non-decreasing along this order so that the stable sort below keeps it.
      double st_prev = 0.0, last_key = 0.0;
      auto push = [&](double key, int32_t w0, int32_t w1) {   // returns the key the task got: the times derived from it stay consistent
        last_key = std::max(last_key, key);
        tasks.push_back({last_key, w0, w1});
        return last_key;
      };
      auto potrf = [&](int s) {
        // POTRF(s) also owns the tile (s, s-1) left of the diagonal one: its row solve needs nothing but the
        // factor of column s-1, and its result feeds the update of (s,s) without leaving the workgroup
        double st = std::max(pulled(s, s), st_prev);
        if (s > 0) st = std::max(st, pulled(s, s - 1));
        for (int c = 0; c + 1 < s; ++c) st = std::max(st, std::max(done[tix(s, c)], done[tix(s - 1, c)]) + HOP);
        st = push(st, (ND_T_POTRF << 24) | i, (s << 8) | s);
        st_prev = st;
        double ready_at = st + 1.5 + 0.8 * s;
        if (s > 0) {
          ready_at = std::max(ready_at, done[tix(s - 1, s - 1)] + HOP) + 4.0;
          done[tix(s, s - 1)] = ready_at;
        }
        done[tix(s, s)] = ready_at + d_potrf(0);
        return st;
      };
      std::vector<double> st_of(f.npt + 1, 0.0);
      if (f.npt > 0) st_of[0] = potrf(0);
      // ... and the BOUNDARY rows of a column (only the Schur tasks read them) two columns later than its pivot rows (which the
      // following chain tasks read): listed with them they hold ~10 workgroups per front idle until the column's factor
      // is out; later that factor exists and they only work.
      auto cols = [&](int s, int r0, int r1) {
        const double st = st_of[s];
        for (int r = r0; r < r1; ++r) {
          if (r == s + 1 && r < f.npt) continue;             // tile (s+1, s) belongs to POTRF(s+1)
          double sr = std::max(pulled(r, s), st);   // never listed before the POTRF it waits for
          for (int c = 0; c < s; ++c) sr = std::max(sr, std::max(done[tix(r, c)], done[tix(s, c)]) + HOP);
          sr = push(sr, (ND_T_COL << 24) | i, (r << 8) | s);
          const double fin = std::max(sr + 1.5 + 0.8 * s, done[tix(s, s)] + HOP) + 2.5;
          done[tix(r, s)] = fin;
        }
      };
      static const int defer_boundary = [] { const char* e = getenv("SLM_DAG_DEFER_BOUNDARY"); return e ? atoi(e) : 2; }();   // columns of delay (0: none; C2, 8 frames, ms per LM iteration at 0 / 1 / 2 / all: 2.342 / 2.320 / 2.323 vs 2.335 / 2.334)
      for (int s = 0; s < f.npt; ++s) {
        if (s + 1 < f.npt) st_of[s + 1] = potrf(s + 1);
        cols(s, s + 1, defer_boundary > 0 ? f.npt : f.nt);       // pivot rows of column s (all rows without the deferral)
        if (defer_boundary > 0 && s >= defer_boundary) cols(s - defer_boundary, f.npt, f.nt);   // boundary rows of an earlier column
      }
      if (defer_boundary > 0)
        for (int s = std::max(0, f.npt - defer_boundary); s < f.npt; ++s) cols(s, f.npt, f.nt);
      double fd_ = 0.0;
      for (int s = 0; s < f.npt; ++s) fd_ = std::max(fd_, done[tix(s, s)]);
      fact_done[i] = fd_;
      if (f.parent >= 0) {
        for (int r = f.npt; r < f.nt; ++r)
          for (int sc = f.npt; sc <= r; ++sc) {
            double st = pulled(r, sc);
            for (int c = 0; c < f.npt; ++c) st = std::max(st, std::max(done[tix(r, c)], done[tix(sc, c)]) + HOP);
            st = push(st, (ND_T_SCHUR << 24) | i, (r << 8) | sc);
            done[tix(r, sc)] = st + d_schur(f.npt);
          }
      }
    }
    // back substitution, root first (fronts are stored deepest level first)
    std::vector<double> back_done(T, 0.0);
    double t_end = 0.0;
    for (int i = T - 1; i >= 0; --i) {
      const NDFront& f = out.fronts[i];
      double st0 = fact_done[i];
      if (f.parent >= 0) st0 = std::max(st0, back_done[f.parent] + HOP);
      else for (int k = 0; k < T; ++k) st0 = std::max(st0, fact_done[k]);   // the root starts when everything is factored
      // BACKB(f,c) for every pivot column in parallel, then ONE task per front for the chain over its columns.  Fronts with
      // few pivot tile columns (round 5: <= SLM_BACK_FUSE_NPT, default 2 -- the five deepest levels at C2) get
      // ONE task that does both (word 1 = 1): the boundary's x is gathered once instead of once per column, one hand-off
      // per tree level instead of two, and a third as many tasks to take tickets for.
      static const int fuse_npt = [] {
        const char* e = getenv("SLM_BACK_FUSE_NPT");
        return e ? atoi(e) : 2;
      }();
      // (the fused form stages the boundary's x in ONE tile of LDS -- dag_task_back's `xb = M`: n2p <= ND_MAX_N2P_FUSED)
      const bool fused = f.nb > 0 && f.npt <= fuse_npt && f.n2p <= ND_MAX_N2P_FUSED;
      for (int c = f.npt - 1; c >= 0; --c)
        if (f.nb > 0 && !fused) tasks.push_back({st0, (ND_T_BACKB << 24) | i, (c << 8) | c});
      const double stc = st0 + ((f.nb > 0 && !fused) ? 2.0 + 0.5 * (f.nt - f.npt) : 0.0);
      tasks.push_back({stc, (ND_T_BACK << 24) | i, fused ? 1 : 0});
      double prev = stc + 2.0 + 0.25 * f.npt * (f.npt + 1) / 2 + (fused ? 1.0 + 0.45 * (f.nt - f.npt) * f.npt : 0.0);
      back_done[i] = prev;
      t_end = std::max(t_end, prev);
    }
    if (top_cut < 0) out.dag_critical_us = t_end;
    std::stable_sort(tasks.begin(), tasks.end(), [](const Task& a, const Task& b) { return a.start < b.start; });
    list.resize(2 * tasks.size());
    for (size_t k = 0; k < tasks.size(); ++k) {
      list[2 * k] = tasks[k].w0;
      list[2 * k + 1] = tasks[k].w1;
    }
  };
  model_list(-1, out.dag_tasks);
  // top of the tree for the hybrid solve: the factor tasks of the fronts above the cut plus the back substitution of the
  // WHOLE tree (the deeper fronts are factored and forward-substituted by the per-level launches before this list runs, so
  // their BACKB / BACK tasks need nothing but their parent's solution)
  out.dag_top_tasks.clear();
  if (out.dag_cut_depth >= 0) {
    static const bool legacy_top = getenv("SLM_DAG_TOP_LEGACY") != nullptr;   // (A/B: the whole-tree order filtered by depth)
    if (legacy_top) {
      for (size_t k = 0; 2 * k < out.dag_tasks.size(); ++k) {
        const int w0 = out.dag_tasks[2 * k], type = w0 >> 24;
        if (out.fronts[w0 & 0xFFFFFF].depth <= out.dag_cut_depth || type == ND_T_BACKB || type == ND_T_BACK) {
          out.dag_top_tasks.push_back(w0);
          out.dag_top_tasks.push_back(out.dag_tasks[2 * k + 1]);
        }
      }
    } else {
      model_list(out.dag_cut_depth, out.dag_top_tasks);
    }
  }
  // ---- destinations of the assembled blocks -------------------------------------------------
  auto dest_of = [&](int a, int bnode, NDDest& d) -> bool {   // block given as (a,b), a >= b by id
    const int e = order[a] < order[bnode] ? a : bnode;          // earlier eliminated -> column
    const int l = (e == a) ? bnode : a;
    const int tid = node_tree[e];
    const int prow = pos_in(tid, l);
    if (prow < 0) return false;
    d.front = front_of_tree[tid];
    d.pcol = out.node_pos[e];
    d.prow = prow;
    d.transpose = (e == a && a != bnode) ? 1 : 0;
    return true;
  };
  out.block_dest.resize(n_pairs);
  for (int i = 0; i < n_pairs; ++i) {
    const int a = (int)(pairs[i] / (uint32_t)J), bb = (int)(pairs[i] % (uint32_t)J);
    if (!dest_of(a, bb, out.block_dest[i])) return false;
  }
  // reverse adjacency of the node KNN graph: which ARAP edges end in node k (k_reg_grad_nd gathers
  // a node's diagonal block from them instead of scattering with atomics)
  out.in_start.assign(J + 1, 0);
  for (int e = 0; e < J * K_ED; ++e) {
    const int k = ed_knn[e];
    if (k >= 0 && k < J) out.in_start[k + 1]++;
  }
  for (int k = 0; k < J; ++k) out.in_start[k + 1] += out.in_start[k];
  out.in_edge.assign(out.in_start[J], 0);
  {
    std::vector<int32_t> fill(out.in_start.begin(), out.in_start.end() - 1);
    for (int e = 0; e < J * K_ED; ++e) {
      const int k = ed_knn[e];
      if (k >= 0 && k < J) out.in_edge[fill[k]++] = e;
    }
  }
  out.pair_dest.resize((size_t)J * K_ED);
  for (int j = 0; j < J; ++j)
    for (int s = 0; s < K_ED; ++s) {
      const int k = ed_knn[j * K_ED + s];
      if (k < 0 || k >= J) {   // not an edge (the kernels skip such entries): no destination
        out.pair_dest[(size_t)j * K_ED + s] = NDDest{-1, 0, 0, 0};
        continue;
      }
      const int a = std::max(j, k), bb = std::min(j, k);
      if (!dest_of(a, bb, out.pair_dest[(size_t)j * K_ED + s])) return false;
    }
  return true;
}

bool nd_dest_of(const NDPlanHost& p, int J, uint32_t key, NDDest& d) {
  const int a = (int)(key / (uint32_t)J), b = (int)(key % (uint32_t)J);
  if (a >= J || (int)p.order.size() != J) return false;
  const int e = p.order[a] < p.order[b] ? a : b;   // earlier eliminated -> column
  const int l = (e == a) ? b : a;
  const int tid = p.node_tree[e];
  int prow = -1;
  for (int k = p.occ_start[l]; k < p.occ_start[l + 1]; ++k)
    if (p.occ[k].first == tid) { prow = p.occ[k].second; break; }
  if (prow < 0) return false;
  d.front = p.front_of_tree[tid];
  d.pcol = p.node_pos[e];
  d.prow = prow;
  d.transpose = (e == a && a != b) ? 1 : 0;
  return true;
}
