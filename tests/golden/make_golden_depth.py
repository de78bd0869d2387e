"""Golden vectors for SURVEY.md 8(f) row f2: run the REFERENCE's ``depth_preprocessing``
(``utils/data_loader.py:333-523``, unmodified, through ``ref_shim``) on small seeded synthetic
depth / colour / segmentation images and record inputs + outputs per option variant.

    python tests/golden/make_golden_depth.py        ->  tests/golden/dp_60x80.npz
"""
from __future__ import annotations

import os
import sys
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

# variant tag -> reference option overrides
VARIANTS = {
    "v1": dict(data="superv1"),
    "v1dil": dict(data="superv1", dilate_invalid_kernel=2, use_seg=True, del_seg_classes=[1]),
    "v1raft": dict(data="superv1", depth_model="raft_stereo", dilate_invalid_kernel=3),
    "v1n8": dict(data="superv1", normal_model="8neighbors"),
    "v1seg": dict(data="superv1", use_seg=True, del_seg_classes=[2]),
    "v2": dict(data="superv2", load_depth=True),
    "v2range": dict(data="superv2", load_depth=False, depth_width_range=(0.1, 0.8)),
    # stereo (SSIM) confidence blended in: confs = 0.5 * confs + 0.5 * sigmoid(inputs[("disp_conf", 0)])
    "v2ssim": dict(data="superv2", load_depth=True, disable_ssim_conf=False),
}


def make_inputs(H=60, W=80, seed=7):
    rng = np.random.default_rng(seed)
    K = synth._scaled_intrinsics(H, W)
    inv_K = np.linalg.pinv(K)                                  # data_loader.py:125
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    depth = (0.2 * synth._surface(uu, vv, H, W, 0.3)).astype(np.float32)
    depth[rng.uniform(size=(H, W)) < 0.02] = 0.0               # holes
    depth[5:9, 30:34] = 2.0                                    # beyond the 1.5 threshold
    color = rng.uniform(0, 255, (3, H, W)).astype(np.float32)
    logits = synth._f32(synth._box_mean(synth._class_logits(uu, vv, H, W, 3, 0.0), 5))
    # stereo pair: right camera 5.5 mm to the right (units of the depth map), slight rotation; image in [0, 1]
    r2 = np.random.default_rng(seed + 100)
    img01 = (0.5 + 0.25 * np.sin(uu / 3.0)[None] * np.cos(vv / 4.0)[None] + r2.uniform(-0.2, 0.2, (3, H, W))).astype(np.float32)
    a = 0.01
    stereo_T = np.array([[np.cos(a), 0, np.sin(a), -0.0055], [0, 1, 0, 0.0002], [-np.sin(a), 0, np.cos(a), 0.0001],
                         [0, 0, 0, 1]], np.float32)
    return dict(H=H, W=W, K=K, inv_K=inv_K, depth=depth, color=color, divterm=1.0 / (2.0 * 0.6 * 0.6),
                seg_conf=logits, seg=np.argmax(logits, 0).astype(np.int64), color01=img01, stereo_T=stereo_T)


def run_reference(ref, base, okw):
    kw = dict(okw)
    use_seg = kw.pop("use_seg", False)
    opt = SimpleNamespace(height=base["H"], width=base["W"], load_valid_mask=False, depth_model="monodepth2",
                          dilate_invalid_kernel=0, normal_model="naive", phase="test", load_depth=True,
                          depth_width_range=(0.02, 0.98), num_classes=3)
    for k, v in kw.items():
        setattr(opt, k, v)
    inputs = {("depth", 0): torch.from_numpy(base["depth"].copy())[None, None],
              ("disp", 0): torch.zeros(1, 1, base["H"], base["W"]),
              "inv_K": torch.from_numpy(base["inv_K"])[None], "K": torch.from_numpy(base["K"])[None],
              ("color", 0): torch.from_numpy(base["color"].copy())[None], "divterm": base["divterm"],
              "filename": ["000001"]}
    if hasattr(opt, "disable_ssim_conf"):
        inputs["stereo_T"] = torch.from_numpy(base["stereo_T"].copy())[None]
        inputs[("color", 0)] = torch.from_numpy(base["color01"].copy())[None]
    if use_seg:
        inputs[("seg", 0)] = torch.from_numpy(base["seg"])[None, None]
        inputs[("seg_conf", 0)] = torch.from_numpy(base["seg_conf"].astype(np.float64))[None]
    data, _, not_inval = ref.data_loader.depth_preprocessing(opt, None, inputs, return_valid_map=True)
    out = dict(points=data.points, norms=data.norms, colors=data.colors, radii=data.radii, confs=data.confs,
               valid=data.valid, index_map=data.index_map, valid_map=data.valid_map, inval=~not_inval[0, 0] if not_inval.dim() == 4 else ~not_inval[0])
    if use_seg:
        out.update(seg=data.seg, seg_conf=data.seg_conf, dist2edge=data.dist2edge)
    if ("disp_conf", 0) in inputs:
        out.update(disp_conf=inputs[("disp_conf", 0)])
    return {k: v.detach().cpu().numpy() for k, v in out.items()}


def main():
    ref = ref_shim.install_data_loader()
    base = make_inputs()
    g = {"in_" + k: v for k, v in base.items()}
    for tag, okw in VARIANTS.items():
        out = run_reference(ref, base, okw)
        for k, v in out.items():
            g[f"{tag}_{k}"] = v
        print(tag, "valid", int(out["valid"].sum()), "of", base["H"] * base["W"])
    path = os.path.join(HERE, "dp_60x80.npz")
    np.savez_compressed(path, **g)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
