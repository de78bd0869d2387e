"""HIP depth_preprocessing vs the reference's goldens and the NumPy oracle,
through the C ABI.  Needs an MI355X (-m gpu)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import depth_oracle as dpo
from test_depth_oracle import GOLD, VARIANTS

pytestmark = pytest.mark.gpu


def _run_gpu(base, okw):
    import torch
    from super_amd.data_loader import depth_preprocessing
    kw = dict(okw)
    use_seg = kw.pop("use_seg", False)
    H, W = int(base["H"]), int(base["W"])
    opt = SimpleNamespace(height=H, width=W, load_valid_mask=False, depth_model="monodepth2", dilate_invalid_kernel=0,
                          normal_model="naive", phase="test", load_depth=True, depth_width_range=(0.02, 0.98),
                          num_classes=3)
    for k, v in kw.items():
        setattr(opt, k, v)
    inputs = {("depth", 0): torch.from_numpy(base["depth"].copy())[None, None].cuda(),
              ("disp", 0): torch.zeros(1, 1, H, W).cuda(),
              "inv_K": torch.from_numpy(base["inv_K"])[None], "K": torch.from_numpy(base["K"])[None],
              ("color", 0): torch.from_numpy(base["color"].copy())[None].cuda(), "divterm": float(base["divterm"]),
              "filename": ["000001"]}
    if use_seg:
        inputs[("seg", 0)] = torch.from_numpy(base["seg"])[None, None].cuda()
        inputs[("seg_conf", 0)] = torch.from_numpy(base["seg_conf"].astype(np.float64))[None].cuda()
    if "disable_ssim_conf" in kw:
        inputs["stereo_T"] = torch.from_numpy(base["stereo_T"].copy())[None]
        inputs[("color", 0)] = torch.from_numpy(base["color01"].copy())[None].cuda()
    data, inputs, not_inval = depth_preprocessing(opt, None, inputs, return_valid_map=True)
    out = {k: v.cpu().numpy() for k, v in vars(data).items() if hasattr(v, "cpu")}
    if ("disp_conf", 0) in inputs:
        out["disp_conf"] = inputs[("disp_conf", 0)].cpu().numpy()
    out["inval"] = ~not_inval[0, 0].cpu().numpy()
    out["depth_after"] = inputs[("depth", 0)][0, 0].cpu().numpy()
    return out


def _check(out, ref, tag, exact_points=True):
    np.testing.assert_array_equal(out["inval"], ref[f"{tag}inval"])
    np.testing.assert_array_equal(out["valid"], ref[f"{tag}valid"])
    np.testing.assert_array_equal(out["index_map"], ref[f"{tag}index_map"])
    np.testing.assert_array_equal(out["points"], ref[f"{tag}points"])          # bit-exact float32 back-projection
    np.testing.assert_array_equal(out["colors"], ref[f"{tag}colors"])
    np.testing.assert_allclose(out["norms"], ref[f"{tag}norms"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["radii"], ref[f"{tag}radii"], rtol=1e-5)
    np.testing.assert_allclose(out["confs"], ref[f"{tag}confs"], rtol=1e-5 if f"{tag}disp_conf" in ref else 1e-6)
    if f"{tag}disp_conf" in ref:
        # warp + blend pinned by the reference; the SSIM kernel by the restated skimage algorithm
        np.testing.assert_allclose(out["disp_conf"], ref[f"{tag}disp_conf"], rtol=0, atol=5e-5)
    if f"{tag}seg" in ref:
        np.testing.assert_array_equal(out["seg"], ref[f"{tag}seg"])
        np.testing.assert_allclose(out["seg_conf"], ref[f"{tag}seg_conf"], rtol=1e-6)   # float32 logits in HBM
        np.testing.assert_allclose(out["dist2edge"], ref[f"{tag}dist2edge"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_depth_preprocessing_matches_reference_goldens(tag):
    g = np.load(GOLD)
    base = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    out = _run_gpu(base, VARIANTS[tag])
    _check(out, {k: g[k] for k in g.files}, tag + "_")
    assert np.isnan(out["depth_after"][out["inval"]]).all() and not np.isnan(out["depth_after"][~out["inval"]]).any()


def test_stereo_confidence_full_size_matches_oracle():
    """480x640, default CLI setting (SSIM confidence on): device warp + SSIM + blend against the oracle."""
    from super_amd import synth
    H, W = 480, 640
    rng = np.random.default_rng(9)
    K = synth.intrinsics()
    inv_K = np.linalg.pinv(K)
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    depth = (0.2 * synth._surface(uu, vv, H, W, 0.3)).astype(np.float32)
    depth[rng.uniform(size=(H, W)) < 0.01] = 0.0
    img = (0.5 + 0.25 * np.sin(uu / 5.0)[None] * np.cos(vv / 7.0)[None] + rng.uniform(-0.2, 0.2, (3, H, W))).astype(np.float32)
    a = 0.004
    T = np.array([[np.cos(a), 0, np.sin(a), -0.0055], [0, 1, 0, 1e-4], [-np.sin(a), 0, np.cos(a), 2e-4], [0, 0, 0, 1]],
                 np.float32)
    base = dict(H=H, W=W, K=K, inv_K=inv_K, depth=depth, color=img, color01=img, stereo_T=T, divterm=1.0 / (2 * 0.6 * 0.6))
    okw = dict(data="superv2", load_depth=True, disable_ssim_conf=False)
    out = _run_gpu(base, okw)
    opt = dpo.default_opt(height=H, width=W, **okw)
    ref = dpo.depth_preprocessing(opt, depth, K, inv_K, img, base["divterm"], stereo_T=T)
    ok = np.isfinite(ref["disp_conf"])
    assert ok.mean() > 0.99
    np.testing.assert_allclose(out["disp_conf"][ok], ref["disp_conf"][ok], rtol=0, atol=1e-4)
    np.testing.assert_array_equal(out["valid"], ref["valid"])
    both = np.isfinite(ref["confs"]) & np.isfinite(out["confs"])
    assert both.mean() > 0.99
    np.testing.assert_allclose(out["confs"][both], ref["confs"][both], rtol=0, atol=3e-5)
    assert float(np.nanstd(ref["disp_conf"])) > 0.01 and float(np.nanmax(np.abs(ref["disp_conf"]))) <= 1.0 + 1e-5


@pytest.mark.parametrize("tag", ["v1seg", "v1n8"])
def test_depth_preprocessing_full_size_matches_oracle(tag):
    """480x640 (the SuPer image size) against the NumPy oracle, then the result is a valid target
    for the LM path: index_map enumerates the valid pixels in row-major order."""
    from super_amd import synth
    H, W = 480, 640
    rng = np.random.default_rng(3)
    K = synth.intrinsics()
    inv_K = np.linalg.pinv(K)
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    depth = (0.2 * synth._surface(uu, vv, H, W, 0.3)).astype(np.float32)
    depth[rng.uniform(size=(H, W)) < 0.01] = 0.0
    color = rng.uniform(0, 255, (3, H, W)).astype(np.float32)
    logits = synth._f32(synth._box_mean(synth._class_logits(uu, vv, H, W, 3, 0.0), 11))
    base = dict(H=H, W=W, K=K, inv_K=inv_K, depth=depth, color=color, divterm=1.0 / (2 * 0.6 * 0.6),
                seg_conf=logits, seg=np.argmax(logits, 0).astype(np.int64))
    out = _run_gpu(base, VARIANTS[tag])
    kw = dict(VARIANTS[tag])
    use_seg = kw.pop("use_seg", False)
    ref = dpo.depth_preprocessing(dpo.default_opt(height=H, width=W, **kw), depth, K, inv_K, color, base["divterm"],
                                  seg=base["seg"] if use_seg else None,
                                  seg_conf=logits.astype(np.float64) if use_seg else None)
    _check(out, {"x_" + k: v for k, v in ref.items()}, "x_")
    T = int(out["valid"].sum())
    assert T == len(out["points"]) and T > 0.5 * H * W
    np.testing.assert_array_equal(out["index_map"].reshape(-1)[out["valid"]], np.arange(T))


def test_valid_mask_file_is_read_like_the_reference(tmp_path):
    """opt.load_valid_mask (superv1): <data_dir>/<valid_mask_dir>/<filename>-left.png, non-zero = valid; the result
    equals passing the same mask as a tensor and the oracle with that mask; a missing file fails loudly."""
    import torch
    from PIL import Image
    from super_amd.data_loader import depth_preprocessing
    g = np.load(GOLD)
    base = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    H, W = int(base["H"]), int(base["W"])
    rng = np.random.default_rng(5)
    mask = rng.uniform(size=(H, W)) > 0.2
    mask[10:20, 15:40] = False
    os.makedirs(tmp_path / "masks")
    Image.fromarray((mask * 255).astype(np.uint8)).save(tmp_path / "masks" / "000001-left.png")

    def run(**extra):
        opt = SimpleNamespace(height=H, width=W, data="superv1", load_valid_mask=True, depth_model="monodepth2",
                              dilate_invalid_kernel=2, normal_model="naive", phase="test", load_depth=True,
                              data_dir=str(tmp_path), valid_mask_dir="masks")
        inputs = {("depth", 0): torch.from_numpy(base["depth"].copy())[None, None].cuda(),
                  ("disp", 0): torch.zeros(1, 1, H, W).cuda(), "inv_K": torch.from_numpy(base["inv_K"])[None],
                  "K": torch.from_numpy(base["K"])[None], ("color", 0): torch.from_numpy(base["color"].copy())[None].cuda(),
                  "divterm": float(base["divterm"]), "filename": ["000001"]}
        inputs.update(extra)
        data, _ = depth_preprocessing(opt, None, inputs)[:2]
        return data, opt

    from_file, opt = run()
    from_tensor, _ = run(valid_mask=torch.from_numpy(mask))
    np.testing.assert_array_equal(from_file.valid.cpu().numpy(), from_tensor.valid.cpu().numpy())
    np.testing.assert_array_equal(from_file.points.cpu().numpy(), from_tensor.points.cpu().numpy())
    ref = dpo.depth_preprocessing(dpo.default_opt(height=H, width=W, data="superv1", load_valid_mask=True,
                                                  dilate_invalid_kernel=2), base["depth"], base["K"], base["inv_K"],
                                  base["color"], float(base["divterm"]), valid_mask=mask)
    np.testing.assert_array_equal(from_file.valid.cpu().numpy(), ref["valid"])
    assert not from_file.valid.view(H, W)[10:20, 15:40].any()
    os.remove(tmp_path / "masks" / "000001-left.png")
    with pytest.raises(FileNotFoundError):
        run()
