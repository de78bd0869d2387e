"""HIP surfel fusion vs the reference's goldens, through the C ABI.
Needs an MI355X (-m gpu)."""
import logging
from types import SimpleNamespace

import numpy as np
import pytest

from test_fusion_oracle import GOLD, VARIANTS

pytestmark = pytest.mark.gpu


def _objects(b, okw):
    import torch
    t = lambda a: torch.from_numpy(np.array(a, copy=True)).cuda()
    opt = SimpleNamespace(height=int(b["H"]), width=int(b["W"]), th_dist=0.1, th_cosine_ang=0.4, th_time_steps=30,
                          disable_merging_new_surfels=False, disable_merging_exist_surfels=False,
                          disable_adding_new_surfels=False, disable_removing_unstable_surfels=False,
                          phase="test", method="super", num_neighbors=4)
    for k, v in okw.items():
        setattr(opt, k, v)
    sf = SimpleNamespace(opt=opt, hard_seg=False, evaluate_tracking=False, logger=logging.getLogger("fusion"),
                         points=t(b["sf_points"]), norms=t(b["sf_norms"]), colors=t(b["sf_colors"]), radii=t(b["sf_radii"]),
                         confs=t(b["sf_confs"]), time_stamp=t(b["sf_time_stamp"]), isStable=t(b["sf_isStable"]),
                         knn_indices=t(b["sf_knn_idx"]), knn_w=t(b["sf_knn_w"]),
                         ED_nodes=SimpleNamespace(points=t(b["ed_points"]), radii=t(b["ed_radii"])),
                         projdata=torch.zeros(len(b["sf_points"]), 2).cuda(), time=0)
    inputs = {"K": torch.from_numpy(b["K"])[None], "time": int(b["time"])}
    sfdata = SimpleNamespace(points=t(b["new_points"]), norms=t(b["new_norms"]), colors=t(b["new_colors"]),
                             radii=t(b["new_radii"]), confs=t(b["new_confs"]), valid=t(b["new_valid"]),
                             index_map=t(b["new_index_map"]), time=int(b["time"]))
    return sf, inputs, sfdata


def _check(sf, g, prefix):
    get = lambda k: getattr(sf, k).cpu().numpy()
    assert len(get("points")) == len(g[prefix + "points"])
    np.testing.assert_array_equal(get("isStable"), g[prefix + "isStable"])
    np.testing.assert_array_equal(get("knn_indices"), g[prefix + "knn_indices"])
    np.testing.assert_array_equal(get("time_stamp"), g[prefix + "time_stamp"])
    np.testing.assert_allclose(get("points"), g[prefix + "points"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(get("norms"), g[prefix + "norms"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(get("radii"), g[prefix + "radii"], rtol=1e-12)
    np.testing.assert_allclose(get("confs"), g[prefix + "confs"], rtol=1e-6)
    np.testing.assert_allclose(get("colors"), g[prefix + "colors"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(get("knn_w"), g[prefix + "knn_w"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(get("projdata"), g[prefix + "projdata"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_fusion_matches_reference_goldens(tag):
    from super_amd import fusion
    g = np.load(GOLD)
    b = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    sf, inputs, sfdata = _objects(b, VARIANTS[tag])
    fusion.fuseInputData(sf, inputs, sfdata)
    _check(sf, g, f"{tag}_fuse_")
    fusion.prepareStableIndexNSwapAllModel(sf, inputs, sfdata)
    if f"{tag}_swap_points" in g.files:
        _check(sf, g, f"{tag}_swap_")
    else:
        assert len(sf.points) == int(g[f"{tag}_swap_count"])
    assert bool(sf.isStable.all()) or VARIANTS[tag].get("disable_removing_unstable_surfels", False)


def test_fusion_full_size_matches_oracle():
    """480x640 frame, 120k surfels (with duplicated surfels so that pixels carry several layers),
    2k nodes: fuse + swap on the device == the NumPy oracle; the fused model is a valid LM input."""
    import torch
    from oracle import fusion_oracle as fuo
    from super_amd import fusion, synth
    sc = synth.make_scene(N=100_000, J=2000, H=480, W=640, seed=5, tgt_holes=0.01)
    rng = np.random.default_rng(5)
    P0, N0 = sc.f64("sf_points"), sc.f64("sf_norms")
    dup = rng.choice(sc.N, 20_000, replace=False)
    Pd = P0[dup] + rng.normal(0, 1, (len(dup), 3)) * np.array([4e-4, 4e-4, 2e-3])
    Nd = N0[dup] + rng.normal(0, 0.1, (len(dup), 3))
    Nd /= np.linalg.norm(Nd, axis=1, keepdims=True)
    n = sc.N + len(dup)
    b = dict(H=sc.H, W=sc.W, K=sc.K, sf_points=np.concatenate([P0, Pd]), sf_norms=np.concatenate([N0, Nd]),
             sf_colors=rng.uniform(0, 255, (n, 3)).astype(np.float32), sf_radii=rng.uniform(0.002, 0.004, n),
             sf_confs=rng.uniform(0.2, 3.0, n).astype(np.float32),
             sf_time_stamp=(40.0 - rng.integers(0, 45, n)).astype(np.float32), sf_isStable=rng.uniform(size=n) > 0.05,
             sf_knn_idx=np.concatenate([sc.sf_knn_idx, sc.sf_knn_idx[dup]]),
             sf_knn_w=np.concatenate([sc.f64("sf_knn_w"), sc.f64("sf_knn_w")[dup]]),
             ed_points=sc.f64("ed_points"), ed_radii=sc.f64("ed_radii"), new_points=sc.f64("tgt_points"),
             new_norms=sc.f64("tgt_norms"), new_colors=rng.uniform(0, 255, (sc.T, 3)).astype(np.float32),
             new_radii=rng.uniform(0.002, 0.004, sc.T), new_confs=rng.uniform(0.05, 1.0, sc.T).astype(np.float32),
             new_valid=sc.valid, new_index_map=sc.index_map, time=41)
    okw = dict(th_dist=0.006, th_cosine_ang=0.8)
    sf, inputs, sfdata = _objects(b, okw)
    fusion.fuseInputData(sf, inputs, sfdata)
    m = fuo.Model(b["sf_points"], b["sf_norms"], b["sf_colors"], b["sf_radii"], b["sf_confs"], b["sf_time_stamp"],
                  b["sf_isStable"], b["sf_knn_idx"], b["sf_knn_w"], b["ed_points"], b["ed_radii"])
    new = SimpleNamespace(points=b["new_points"], norms=b["new_norms"], colors=b["new_colors"], radii=b["new_radii"],
                          confs=b["new_confs"], valid=b["new_valid"])
    opt = fuo.default_opt(height=sc.H, width=sc.W, **okw)
    fuo.fuse_input_data(m, opt, b["K"], new, 41)
    ref = {"x_" + k: getattr(m, k) for k in ("points", "norms", "colors", "radii", "confs", "time_stamp", "isStable",
                                              "knn_indices", "knn_w", "projdata")}
    _check(sf, ref, "x_")
    assert len(sf.points) > n and (~sf.isStable[:n].cpu().numpy() & b["sf_isStable"]).sum() > 1000
    fusion.prepareStableIndexNSwapAllModel(sf, inputs, sfdata)
    fuo.swap_stable(m, opt, 41)
    _check(sf, {"y_" + k: getattr(m, k) for k in ("points", "norms", "colors", "radii", "confs", "time_stamp", "isStable",
                                                    "knn_indices", "knn_w", "projdata")}, "y_")
    w = sf.knn_w.cpu().numpy()
    np.testing.assert_allclose(w.sum(1), 1.0, rtol=0, atol=1e-12)
