"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy restatement of the reference's ED-graph construction at frame 0, SURVEY.md 8(f) row f3:
``init_graph`` (``super/graph_encoder.py:11-67``) and the ``grid_mesh`` branch of
``DirectDeformGraph.init_ED_nodes`` / ``forward`` (``:128-161,169-195``): anchors on a pixel grid of
step ``opt.mesh_step_size``, four edges and two triangles per grid cell, node radius = mean length of
the incident edges (NaN -> mean of the others), triangle rest areas.  Pinned against the reference by
``tests/golden/make_golden_graph.py`` -> ``tests/golden/gr_*.npz``.  Only ``tests/`` import this.
"""
from __future__ import annotations

import numpy as np


def init_graph(valid, step):
    """Returns (anchor mask (H,W), edges (2,E), faces (3,F)) with anchors numbered in row-major order."""
    valid = np.asarray(valid, bool)
    h, w = valid.shape
    u = np.arange(0, w - 1, step)
    v = np.arange(0, h - 1, step)
    U, V = np.meshgrid(u, v, indexing="xy")            # (len(v), len(u))
    av = valid[V, U]
    U, V = U[av], V[av]
    n = len(U)
    index_map = -np.ones((h, w), np.int64)
    index_map[V, U] = np.arange(n)
    s = np.stack([U, V], 1)                             # (n,2) as (x,y)
    edges = np.tile(s[:, None, None, :], (1, 4, 2, 1))
    edges[:, 0, 1, 0] += step
    edges[:, 1, 1, 0] += step
    edges[:, 1, 1, 1] += step
    edges[:, 2, 1, 1] += step
    edges[:, 3, 0, 0] += step
    edges[:, 3, 1, 1] += step
    faces = np.concatenate([np.tile(s[:, None, None, :], (1, 2, 1, 1)),
                            np.stack([edges[:, 0:2, 1, :], edges[:, 1:3, 1, :]], 2)], 2)   # (n,2,3,2)
    vp = np.pad(valid, ((0, step), (0, step)), constant_values=False)
    pm = np.pad(index_map, ((0, step), (0, step)), constant_values=-1)
    e = edges.reshape(-1, 2, 2)
    e = e[vp[e[..., 1], e[..., 0]].all(1)]
    e = pm[e[..., 1], e[..., 0]]
    e = e[(e >= 0).all(1)]
    f = faces.reshape(-1, 3, 2)
    f = f[vp[f[..., 1], f[..., 0]].all(1)]
    f = pm[f[..., 1], f[..., 0]]
    f = f[(f >= 0).all(1)]
    return index_map >= 0, e.T.copy(), f.T.copy()


def direct_deform_graph(valid, data_index_map, points, norms, step, seg_conf=None, prune_class_edges=False):
    """``DirectDeformGraph.forward`` (grid_mesh): dict with points, norms, radii, edge_index, edges_lens,
    triangles, triangles_areas, num.  ``seg_conf`` (T,C): the nodes' class fields ``seg`` / ``seg_conf``
    (graph_encoder.py:134-139); ``prune_class_edges`` (``opt.hard_seg and opt.mesh_face``): edges and
    triangles whose vertices differ in class are dropped before lengths, radii and areas (:141-151)."""
    H, W = data_index_map.shape
    mask, edge_index, triangles = init_graph(np.asarray(valid, bool).reshape(H, W), step)
    rows = data_index_map[mask]
    P, N = points[rows], norms[rows]
    sem = {}
    if seg_conf is not None:
        sc = np.asarray(seg_conf, np.float64)[rows]
        seg = np.argmax(sc, axis=1)
        sem = dict(seg=seg, seg_conf=sc)
        if prune_class_edges:
            edge_index = edge_index[:, seg[edge_index[0]] == seg[edge_index[1]]]
            triangles = triangles[:, (seg[triangles[0]] == seg[triangles[1]]) & (seg[triangles[0]] == seg[triangles[2]])]
    lens = np.sqrt(((P[edge_index[0]] - P[edge_index[1]]) ** 2).sum(1))
    J = len(P)
    radii = np.full(J, np.nan)
    for k in range(J):
        inc = (edge_index == k).any(0)
        if inc.any():
            radii[k] = lens[inc].mean()
    bad = np.isnan(radii)
    if bad.any():
        radii[bad] = radii[~bad].mean()
    c = np.cross(P[triangles[1]] - P[triangles[0]], P[triangles[2]] - P[triangles[0]])
    areas = 0.5 * np.sqrt((c ** 2).sum(1) + 1e-13)
    return dict(points=P, norms=N, radii=radii, edge_index=edge_index, edges_lens=lens, triangles=triangles,
                triangles_areas=areas, num=J, **sem)
