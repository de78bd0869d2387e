"""Pin the NumPy restatement of the reference's depth_preprocessing against
golden vectors recorded from the reference itself.  CPU only."""
import os

import numpy as np
import pytest

from oracle import depth_oracle as dpo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dp_60x80.npz")

# same table as tests/golden/make_golden_depth.py VARIANTS
VARIANTS = {
    "v1": dict(data="superv1"),
    "v1dil": dict(data="superv1", dilate_invalid_kernel=2, use_seg=True, del_seg_classes=[1]),
    "v1raft": dict(data="superv1", depth_model="raft_stereo", dilate_invalid_kernel=3),
    "v1n8": dict(data="superv1", normal_model="8neighbors"),
    "v1seg": dict(data="superv1", use_seg=True, del_seg_classes=[2]),
    "v2": dict(data="superv2", load_depth=True),
    "v2range": dict(data="superv2", load_depth=False, depth_width_range=(0.1, 0.8)),
    "v2ssim": dict(data="superv2", load_depth=True, disable_ssim_conf=False),
}


def run_oracle(g, tag):
    kw = dict(VARIANTS[tag])
    use_seg = kw.pop("use_seg", False)
    opt = dpo.default_opt(height=int(g["in_H"]), width=int(g["in_W"]), **kw)
    ssim = "disable_ssim_conf" in kw
    return dpo.depth_preprocessing(opt, g["in_depth"], g["in_K"], g["in_inv_K"], g["in_color01" if ssim else "in_color"],
                                   float(g["in_divterm"]), seg=g["in_seg"] if use_seg else None,
                                   seg_conf=g["in_seg_conf"].astype(np.float64) if use_seg else None,
                                   stereo_T=g["in_stereo_T"] if ssim else None)


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_depth_preprocessing_matches_reference(tag):
    g = np.load(GOLD)
    out = run_oracle(g, tag)
    # discrete outputs: bit-exact
    np.testing.assert_array_equal(out["inval"], g[f"{tag}_inval"])
    np.testing.assert_array_equal(out["valid"], g[f"{tag}_valid"])
    np.testing.assert_array_equal(out["index_map"], g[f"{tag}_index_map"])
    # points: the float32 back-projection is reproduced operation by operation
    np.testing.assert_array_equal(out["points"], g[f"{tag}_points"])
    np.testing.assert_array_equal(out["colors"], g[f"{tag}_colors"])
    np.testing.assert_allclose(out["norms"], g[f"{tag}_norms"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["radii"], g[f"{tag}_radii"], rtol=1e-5)
    np.testing.assert_allclose(out["confs"], g[f"{tag}_confs"], rtol=1e-6)
    if f"{tag}_seg" in g.files:
        np.testing.assert_array_equal(out["seg"], g[f"{tag}_seg"])
        np.testing.assert_allclose(out["seg_conf"], g[f"{tag}_seg_conf"], rtol=1e-12)
        np.testing.assert_allclose(out["dist2edge"], g[f"{tag}_dist2edge"], rtol=0, atol=1e-12)
    if f"{tag}_disp_conf" in g.files:
        # warp (Project3D + grid_sample) and the blend are the reference's own; the SSIM kernel inside is the
        # restatement on both sides (scikit-image absent and unpinned) -- see the oracle's header
        np.testing.assert_allclose(out["disp_conf"], g[f"{tag}_disp_conf"], rtol=0, atol=2e-5)
        assert np.abs(g[f"{tag}_confs"] - g["v2_confs"]).max() > 0.05
    assert 0 < out["valid"].sum() < out["valid"].size
