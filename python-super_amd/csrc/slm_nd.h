// slm_nd.h -- nested-dissection multifrontal Cholesky of the damped normal equations
// (replaces the dense torch.linalg.cholesky / cholesky_solve of reference super/LM.py:37-51).
//
// The coupling graph of the ED nodes is geometric (surfel KNN tuples + node KNN), so a
// recursive coordinate bisection of the node positions gives small vertex separators.
// Elimination follows the separator tree bottom-up; every tree node ("front") is a DENSE
// partial factorisation [F11 F21^T; F21 F22]: factor the n1 pivot variables, push the
// Schur complement F22 - L21 L21^T to the parent (extend-add).  All fronts of one tree
// level are independent and are processed by the same launches (blockIdx.y = front,
// blockIdx.z = frame slot), on the same 64x64 f64-MFMA tile kernels as the band solver.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <utility>
#include <vector>

#define SLM_ND_LEAF 18   // stop bisecting at this many nodes (18 x 7 = 126 scalars: two 64-wide tiles)
// ... and for a solver that runs the whole tree as ONE persistent task graph (one or two frames per launch: the drop-in
// case): fewer, larger leaves -- a leaf's six pivot columns chain inside one front at ~14 us each, while every tree level
// saved is a ~13 us transition (last column -> boundary rows -> update tiles -> the parent's gather) off the critical path.
// C2, one frame per launch, ms per LM iteration at 18 / 30 / 42 / 50 / 60 / 72 nodes: 0.888 / 0.859 / 0.846 / 0.840 / 0.851 / 0.861.
#define SLM_ND_LEAF_LATENCY 50
// Padded boundary scalars of a front that the substitution tasks can stage in LDS: the fused BACK task of the task graph
// holds x of the boundary in ONE 64 x 64 tile (4 096 doubles), its BACKB tasks and the per-level k_fback_prep in two.
#define ND_MAX_N2P_FUSED 4096
#define ND_MAX_N2P 8192

// One front, device + host view.  Local node positions: [0,nv) pivots (elimination order),
// [nv, nv+nb) boundary (ancestor separator nodes, elimination order).  Scalar layout:
// pivots at 7*pos, padded with identity rows to n1p (multiple of 64); boundary at
// n1p + 7*(pos-nv), padded to n2p.  Tiles (64x64, column-major inside) of the lower
// triangle, packed by tile column: tile(r,c) = c*nt - c*(c-1)/2 + (r-c).
struct NDFront {
  int32_t nv, nb;        // pivot / boundary node counts
  int32_t n1, n1p;       // 7*nv and its 64-padding
  int32_t n2p;           // padded boundary scalars
  int32_t nt, npt;       // tiles per side, pivot tile columns
  int32_t parent;        // front index of the parent or -1
  int32_t which_child;   // 0/1: index among the parent's children (extend-add pass)
  int32_t depth;
  int32_t nodes_off;     // into nodes[]: nv pivot node ids then nb boundary node ids
  int32_t eamap_off;     // into eamap[]: boundary index -> local node position in the parent
  int32_t tile_first;    // logical number of the front's first tile (flags of the task-graph form)
  int32_t is_leaf;       // no children: nothing is ever added into its boundary block before its own Schur update
  int64_t tile_off;      // into the slot's front tile storage (doubles): the tiles of the PIVOT columns (c < npt)
  int64_t f22_base;      // ... the tiles of the boundary block (c >= npt): tile t of the front's numbering lives at
                         // f22_base + t * 4096.  Internal fronts: == tile_off (right behind the pivot columns).
                         // Leaves: their boundary block sits in a tail region of the storage that is never zeroed
                         // (it is written once, by the leaf's own Schur update, and read once)
  int64_t vec_off;       // into the slot's front vector storage (doubles), length nt*64
  int64_t linv_off;      // into the slot's diagonal-inverse storage (doubles), npt tiles
};

// destination of a 7x7 node-pair block inside a front
struct NDDest {
  int32_t front;   // front index
  int32_t prow;    // local node position of the later-eliminated node (row)
  int32_t pcol;    // local node position of the earlier-eliminated node (column)
  int32_t transpose;  // 1: the block given as (a,b) with a >= b by id lands transposed
};

// Work item of the pull-form kernels (k_fpull, k_fschur): everything a workgroup needs about its tile, the front and the
// two children it gathers from in ONE 128-byte record -- instead of a chain of dependent loads through the level table,
// the front descriptor, the child lists, the pull-range table and the children's descriptors (each a memory round trip
// in front of the first useful load of a workgroup that lives for ~10 us).
struct NDTileKid {
  int32_t front;        // child front, or -1: nothing of this child maps into the tile
  int32_t nt, npt;      // the child's tiles per side / pivot tile columns
  int32_t pull_off;     // offset of the child's pull map in pullmap (parent scalar -> boundary scalar of the child)
  int64_t f22_base;     // NDFront::f22_base of the child (its update matrix)
  int64_t vec_boundary; // offset of the child's boundary vector rows in the front-vector storage
};
struct NDTileItem {
  int32_t front;        // front index
  int32_t r, c;         // tile row / column in the front's numbering (Schur items: both >= npt)
  int32_t nt, npt;      // the front's tiles per side / pivot tile columns
  int32_t n1, n2;       // true pivot / boundary scalars (7 nv, 7 nb)
  int32_t pad0;         // number of tile (r, c) in the slot's per-tile tables: NDFront::tile_first + its index in the front
  int64_t tile_off;     // NDFront::tile_off, f22_base, vec_off of the front
  int64_t f22_base;
  int64_t vec_off;
  int64_t pad1;
  NDTileKid kid[2];
};
static_assert(sizeof(NDTileItem) == 128, "NDTileItem is one 128-byte record");

// per-level launch bounds (maxima over the slots of a batch)
struct NDLevelSched {
  int32_t n_fronts;    // fronts in the level
  int32_t max_npt;     // pivot tile columns
  int32_t max_nt;      // tiles per side
  int32_t max_pairs;   // boundary node pairs nb*(nb+1)/2 (extend-add)
  int32_t max_n2p;     // padded boundary scalars
  int32_t first;       // first front of the level when all slots of the batch agree, else -1
  int32_t n_schur;     // Schur work items (front, boundary tile pair) of the level
  int32_t schur_at;    // their offset in tile_items when all slots of the batch agree, else -1
  int32_t n_pull;      // pull work items (front, pivot-column tile that a child maps into) of the level
  int32_t pull_at;     // their offset in tile_items when all slots of the batch agree, else -1
};

// ---- persistent task-graph form of the numeric phase (slm_dag.hip) ------------------------------
// One launch runs the whole factorisation + substitutions: workgroups take tasks from a list in
// ticket order and wait on per-tile flags.  Task word 0 = type << 24 | front, word 1 = r << 8 | s.
enum {
  ND_T_POTRF = 0,   // (f, s, s): left-looking update of the diagonal tile of pivot column s, factor, inverse, y_s
  ND_T_COL = 1,     // (f, r, s), r > s: L(r,s) = (A(r,s) - sum_{c<s} L(r,c) L(s,c)^T) L_ss^-T
  ND_T_SCHUR = 2,   // (f, r, s) boundary tiles: Schur complement (update) tile, stored in place; the parent gathers it
  ND_T_BACKB = 3,   // (f, c): y_c -= sum over boundary tiles L(r,c)^T x_r
  ND_T_BACK = 4     // (f): the chain over the front's pivot columns, x_c = L_cc^-T (y_c - sum_{c<r<npt} L(r,c)^T x_r), c = npt-1 .. 0
};

struct NDPlanHost {
  std::vector<NDFront> fronts;        // processing order: deepest level first
  std::vector<int32_t> level_start;   // fronts[level_start[l] .. level_start[l+1]) are independent
  std::vector<int32_t> nodes;
  std::vector<int32_t> eamap;
  std::vector<int32_t> node_front;    // (J) front that eliminates the node
  std::vector<int32_t> node_pos;      // (J) its local pivot position
  // exact work lists of the pull-form kernels: per level its Schur items (front-major, boundary tile pairs) then its
  // pull items (pivot-column tiles some child maps into); item_off[2*level] / [2*level+1] / [2*level+2] delimit them
  std::vector<NDTileItem> tile_items;
  std::vector<int32_t> item_off;
  std::vector<int32_t> in_start;      // (J+1) CSR over in_edge
  std::vector<int32_t> in_edge;       // ARAP edges e = j*K_ED + slot grouped by their TARGET node k, ascending e
  std::vector<NDDest> block_dest;     // per data-term block (order of blk_key)
  std::vector<NDDest> pair_dest;      // per (j, slot) ARAP pair, J*K_ED
  // kept for nd_dest_of(): elimination position and tree node of every ED node, front of every tree node,
  // and per node the (tree node, local position) pairs of the fronts it occurs in
  std::vector<int32_t> order, node_tree, front_of_tree, occ_start;
  std::vector<std::pair<int32_t, int32_t>> occ;
  int64_t tile_doubles = 0, vec_doubles = 0, linv_doubles = 0;
  int64_t tile_zero_doubles = 0;      // leading part of the tile storage that is zeroed before every assembly
  int32_t max_nt = 0, max_npt = 0, max_level_fronts = 0;
  std::vector<NDLevelSched> sched;    // one entry per level
  double flops = 0.0;          // factorisation FLOPs of the 64-padded dense fronts (what the kernels execute)
  double flops_exact = 0.0;    // the same with the true pivot / boundary sizes (no padding)
  // task list of the persistent kernel, sorted by earliest possible start (a topological order: every task
  // comes after the tasks it waits for), and per front {tasks that extend-add into it, those of its child 0}
  std::vector<int32_t> dag_tasks;     // 2 words per task
  std::vector<int32_t> front_kids;    // 2 per front: its children with a boundary (front index or -1)
  std::vector<int32_t> pull_off;      // per front: offset of its pull map (as a child) in pullmap, or -1
  std::vector<int32_t> prng_off;      // per front: offset into prng
  std::vector<int32_t> prng;          // per front, tile row, child k: child boundary tile rows lo | hi << 8 gathered from, -1 none
  std::vector<int32_t> pullmap;       // per child front: parent scalar index -> boundary scalar index of the child, -1
  // hybrid solve of a batch: the levels with many fronts run as per-level launches (throughput), the top of the tree
  // -- depth <= dag_cut_depth, the levels with at most SLM_DAG_TOP_FRONTS (2) fronts: the root and its children -- as tasks (latency)
  std::vector<int32_t> dag_top_tasks; // the tasks of those fronts, in the order of dag_tasks
  int32_t dag_cut_depth = -1;
  double dag_critical_us = 0.0;       // modelled critical path (diagnostic)
};

// Host symbolic analysis.  pairs: unique coupled node pairs key = a*J + b (a >= b) of the data
// term; ed_knn: (J,K_ED); pts: (J,3).  Returns false when the graph cannot be handled.
// leaf_nodes: stop bisecting at this many nodes (0: SLM_ND_LEAF); the environment variable SLM_ND_LEAF overrides both (tests).
bool nd_build_plan(int J, int K_ED, const float* pts, const int32_t* ed_knn, const uint32_t* pairs,
                   int n_pairs, NDPlanHost& out, int leaf_nodes = 0);

// Destination of the 7x7 block of node pair key = a*J + b (a >= b) in an existing plan.  Also succeeds for
// pairs the plan was NOT built from when the later-eliminated node lies in the front of the earlier one
// (a fill position of the dense front): such a pair needs no new symbolic analysis.
bool nd_dest_of(const NDPlanHost& plan, int J, uint32_t key, NDDest& d);
