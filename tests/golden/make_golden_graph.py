"""Golden vectors for SURVEY.md 8(f) row f3: run the REFERENCE's ``DirectDeformGraph`` (grid mesh,
``super/graph_encoder.py:11-195``, unmodified, through ``ref_shim``) on seeded synthetic frames.

    python tests/golden/make_golden_graph.py        ->  tests/golden/gr_60x80.npz
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd"))
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
from super_amd import synth  # noqa: E402

torch.set_num_threads(1)
VARIANTS = {"step6": 6, "step9": 9, "step4": 4}
# Semantic-SuPer: node class fields (graph_encoder.py:134-139,190-192) and, with hard_seg + mesh_face, edges /
# triangles across a class boundary dropped before the radii (graph_encoder.py:141-151)
SEM_VARIANTS = {"sem6": dict(step=6, hard_seg=False), "hard6": dict(step=6, hard_seg=True),
                "hard4": dict(step=4, hard_seg=True)}


def main():
    ref = ref_shim.install()
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    import super.graph_encoder as ge  # noqa: E402
    sc = synth.make_scene(N=1000, J=12, H=60, W=80, seed=13, src_border=4, tgt_border=3, tgt_holes=0.06)
    g = dict(in_H=sc.H, in_W=sc.W, in_valid=sc.valid, in_index_map=sc.index_map, in_points=sc.f64("tgt_points"),
             in_norms=sc.f64("tgt_norms"))
    for tag, step in VARIANTS.items():
        opt = SimpleNamespace(height=sc.H, width=sc.W, mesh_step_size=step, downsample_params=[], ball_piv_radii=[0.08],
                              method="super", mesh_face=True)
        data = ref_shim.Data(points=torch.from_numpy(g["in_points"]), norms=torch.from_numpy(g["in_norms"]),
                             valid=torch.from_numpy(sc.valid), index_map=torch.from_numpy(sc.index_map))
        graph = ge.DirectDeformGraph(opt)(None, data)
        for k in ("points", "norms", "radii", "edge_index", "edges_lens", "triangles", "triangles_areas"):
            g[f"{tag}_{k}"] = getattr(graph, k).cpu().numpy()
        g[f"{tag}_num"] = graph.num
        print(tag, "nodes", graph.num, "edges", graph.edge_index.shape[1], "triangles", graph.triangles.shape[1],
              "nan radii fixed", int(np.isnan(g[f"{tag}_radii"]).sum()))
    # segmentation confidences of the frame's points: three vertical bands with noise
    rs = np.random.default_rng(113)
    T = len(g["in_points"])
    u = (np.nonzero(sc.valid.reshape(-1))[0] % sc.W).astype(np.float64)
    conf = np.stack([np.exp(-((u - c) / 14.0) ** 2) for c in (12.0, 40.0, 68.0)], 1) + rs.uniform(0, 0.25, (T, 3))
    conf /= conf.sum(1, keepdims=True)
    g["in_seg_conf"] = conf
    for tag, kw in SEM_VARIANTS.items():
        opt = SimpleNamespace(height=sc.H, width=sc.W, mesh_step_size=kw["step"], downsample_params=[], ball_piv_radii=[0.08],
                              method="semantic-super", mesh_face=True, hard_seg=kw["hard_seg"])
        data = ref_shim.Data(points=torch.from_numpy(g["in_points"]), norms=torch.from_numpy(g["in_norms"]),
                             valid=torch.from_numpy(sc.valid), index_map=torch.from_numpy(sc.index_map),
                             seg=torch.from_numpy(np.argmax(conf, 1)), seg_conf=torch.from_numpy(conf))
        graph = ge.DirectDeformGraph(opt)(None, data)
        for k in ("points", "norms", "radii", "edge_index", "edges_lens", "triangles", "triangles_areas", "seg", "seg_conf"):
            g[f"{tag}_{k}"] = getattr(graph, k).cpu().numpy()
        g[f"{tag}_num"] = graph.num
        print(tag, "nodes", graph.num, "edges", graph.edge_index.shape[1], "triangles", graph.triangles.shape[1],
              "classes", np.bincount(g[f"{tag}_seg"]))
    path = os.path.join(HERE, "gr_60x80.npz")
    np.savez_compressed(path, **g)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
