"""HIP surfel fusion vs the reference's goldens, through the C ABI.
Needs an MI355X (-m gpu)."""
import logging
from types import SimpleNamespace

import numpy as np
import pytest

from test_fusion_oracle import GOLD, GOLD_K6, K6_VARIANTS, SEG, VARIANTS

pytestmark = pytest.mark.gpu


def _objects(b, okw, seg=False):
    import torch
    t = lambda a: torch.from_numpy(np.array(a, copy=True)).cuda()
    opt = SimpleNamespace(height=int(b["H"]), width=int(b["W"]), th_dist=0.1, th_cosine_ang=0.4, th_time_steps=30,
                          disable_merging_new_surfels=False, disable_merging_exist_surfels=False,
                          disable_adding_new_surfels=False, disable_removing_unstable_surfels=False,
                          phase="test", method="super", num_neighbors=int(b.get("num_neighbors", 4)))
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
    if seg:
        sf.hard_seg = bool(okw.get("hard_seg", False))
        sf.seg, sf.seg_conf, sf.dist2edge = t(b["sf_seg"]), t(b["sf_seg_conf"]), t(b["sf_dist2edge"])
        sf.ED_nodes.seg, sf.ED_nodes.seg_conf = t(b["ed_seg"]), t(b["ed_seg_conf"])
        sfdata.seg, sfdata.seg_conf, sfdata.dist2edge = t(b["new_seg"]), t(b["new_seg_conf"]), t(b["new_dist2edge"])
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
    if prefix + "seg" in (g.files if hasattr(g, "files") else g):
        import torch
        assert sf.seg.dtype == torch.long and sf.seg_conf.dtype == torch.float64
        np.testing.assert_array_equal(get("seg"), g[prefix + "seg"])
        np.testing.assert_allclose(get("seg_conf"), g[prefix + "seg_conf"], rtol=0, atol=1e-13)
        np.testing.assert_array_equal(get("dist2edge"), g[prefix + "dist2edge"])


@pytest.mark.parametrize("tag,gold", [(t, GOLD) for t in VARIANTS] + [(t, GOLD_K6) for t in K6_VARIANTS])
def test_fusion_matches_reference_goldens(tag, gold):
    from super_amd import fusion
    g = np.load(gold)
    b = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    sf, inputs, sfdata = _objects(b, VARIANTS[tag], seg=tag in SEG)
    if tag == "track":
        import torch
        sf.track_pts, sf.evaluate_tracking = {}, False
        sf.track_id = torch.from_numpy(b["track_id"].copy()).cuda()
        inputs["filename"] = ["000041"]
    fusion.fuseInputData(sf, inputs, sfdata)
    _check(sf, g, f"{tag}_fuse_")
    if tag == "track":
        np.testing.assert_array_equal(sf.track_id.cpu().numpy(), g["track_fuse_track_id"])
        sf.evaluate_tracking = True
    fusion.prepareStableIndexNSwapAllModel(sf, inputs, sfdata)
    if tag == "track":
        np.testing.assert_array_equal(sf.track_id.cpu().numpy(), g["track_swap_track_id"])
    if f"{tag}_swap_points" in g.files:
        _check(sf, g, f"{tag}_swap_")
    else:
        assert len(sf.points) == int(g[f"{tag}_swap_count"])
    assert bool(sf.isStable.all()) or VARIANTS[tag].get("disable_removing_unstable_surfels", False)


def test_track_point_bookkeeping_on_device_tensors_equals_the_reference_golden():
    """Row f4 with the model on the GPU (what the driver holds): ``init_track_pts`` / ``update_track_pts`` on cuda
    tensors reproduce ``tests/golden/track_48x64.npz`` (recorded from the reference with a non-empty gt)."""
    import os
    import torch
    from super_amd import evaluation as ev
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "track_48x64.npz"))
    keys = ["000010", "000020", "000030"]
    gt = {k: g["gtfile_" + k] for k in keys}
    t = lambda a: torch.from_numpy(np.array(a, copy=True)).cuda()
    sf = SimpleNamespace(points=t(g["in_sf_points"]), isStable=t(g["in_sf_isStable"]), projdata=t(g["projdata"]),
                         track_id=t(g["track_id0"]), track_num=20, gt=gt, gt_strkeys=keys, track_rsts={})
    sfdata = SimpleNamespace(points=t(g["in_new_points"]), index_map=t(g["in_new_index_map"]))
    ev.init_track_pts(sf, sfdata, "000010", th=0.2)
    np.testing.assert_array_equal(sf.track_id.cpu().numpy(), g["init_track_id"])
    np.testing.assert_allclose(sf.track_rsts["000010"].cpu().numpy(), g["init_rsts"], rtol=0, atol=1e-12)
    ev.update_track_pts(sf, sfdata, "000020", th=0.05)
    np.testing.assert_array_equal(sf.track_id.cpu().numpy(), g["upd20_track_id"])
    np.testing.assert_allclose(sf.track_rsts["000020"].cpu().numpy(), g["upd20_rsts"], rtol=0, atol=1e-12)
    sf.projdata = t(g["projdata2"])
    ev.update_track_pts(sf, sfdata, "000010")
    np.testing.assert_allclose(sf.track_rsts["000010"].cpu().numpy(), g["upd10_rsts"], rtol=0, atol=1e-12)


def test_tracked_points_follow_the_surface_through_the_driver():
    """Labelled points (row f4) attached by init_track_pts stay on their surfels through LM, update,
    fusion and swap for several frames: the ids stay assigned and point at live surfels, the recorded
    positions are those surfels' projections, and they drift by no more than a few pixels (the
    synthetic surface deforms in depth; there is no material ground truth to compare with)."""
    import torch
    from super_amd import evaluation as ev, synth
    from driver_harness import FrameLoop as SuPer
    H, W = 96, 128
    K = synth._scaled_intrinsics(H, W)
    inv_K = np.linalg.pinv(K)
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    color = np.random.default_rng(2).uniform(0, 255, (3, H, W)).astype(np.float32)
    opt = SimpleNamespace(height=H, width=W, data="superv1", load_valid_mask=False, depth_model="monodepth2",
                          dilate_invalid_kernel=0, normal_model="naive", phase="test", method="super", load_depth=True,
                          deform_udpate_method="super_edg", mesh_step_size=8, use_derived_gradient=True,
                          sf_point_plane=True, mesh_arap=True, mesh_rot=True, sf_point_plane_weight=1.0,
                          mesh_arap_weight=10.0, mesh_rot_weight=1.0, num_optimize_iterations=10, num_neighbors=4,
                          num_ED_neighbors=4, th_dist=0.02, th_cosine_ang=0.4, th_time_steps=30,
                          disable_merging_new_surfels=False, disable_merging_exist_surfels=False,
                          disable_adding_new_surfels=False, disable_removing_unstable_surfels=False)
    rng = np.random.default_rng(4)
    labels = np.stack([rng.integers(20, W - 20, 20), rng.integers(20, H - 20, 20), np.ones(20)], 1).astype(np.int64)
    gt = {"%06d" % k: torch.from_numpy(labels.copy()) for k in range(5)}      # a camera-fixed pattern: the surface only moves in depth
    model = SuPer(opt)
    for k in range(5):
        depth = synth._surface(uu, vv, H, W, 0.3 + 0.03 * k).astype(np.float32)
        depth[:4] = 0.0
        depth[:, :4] = 0.0
        inputs = {("depth", 0): torch.from_numpy(depth.copy())[None, None], ("disp", 0): torch.zeros(1, 1, H, W),
                  "inv_K": torch.from_numpy(inv_K)[None], "K": torch.from_numpy(K)[None],
                  ("color", 0): torch.from_numpy(color)[None], "divterm": torch.tensor(1.0 / (2 * 0.6 * 0.6)),
                  "filename": ["%06d" % k], "time": k, "ID": torch.tensor([k])}
        if k == 1:       # tracking is switched on once the model exists
            sf = model.sf
            sf.evaluate_tracking, sf.track_pts = True, {}
            sf.gt, sf.gt_strkeys = gt, sorted(gt)
            sf.track_num, sf.track_rsts = 20, {}
            sf.track_id = -torch.ones(20, dtype=torch.long, device="cuda")
            sf.init_track_pts = lambda d, n: ev.init_track_pts(sf, d, n)
            sf.update_track_pts = lambda d, n: ev.update_track_pts(sf, d, n)
        model(SimpleNamespace(), inputs)
    sf = model.sf
    assert int((sf.track_id >= 0).sum()) >= 18            # nearly every label found its surfel and kept it
    live = sf.track_id >= 0
    assert bool(sf.isStable[sf.track_id[live]].all())
    np.testing.assert_array_equal(sf.track_rsts["000004"][live.cpu()][:, :2].cpu().numpy(),
                                  sf.projdata[sf.track_id[live]].cpu().numpy())
    for name in ("000002", "000004"):
        err = ev.evaluate(gt[name].numpy(), sf.track_rsts[name].cpu().numpy())
        assert 0 <= err[live.cpu().numpy()].max() < 4.0, (name, err)


@pytest.mark.parametrize("mode", ["plain", "sem", "hard"])
def test_fusion_full_size_matches_oracle(mode):
    """480x640 frame, 120k surfels (with duplicated surfels so that pixels carry several layers),
    2k nodes: fuse + swap on the device == the NumPy oracle; the fused model is a valid LM input.
    `sem` / `hard`: Semantic-SuPer segmentation fields, Jensen-Shannon weights, class-restricted neighbours."""
    import torch
    from oracle import fusion_oracle as fuo
    from super_amd import fusion, synth
    seg = mode != "plain"
    sc = synth.make_scene(N=100_000, J=2000, H=480, W=640, seed=5, tgt_holes=0.01, semantic=seg, num_classes=3)
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
    kw = {}
    if seg:
        okw.update(method="semantic-super", num_classes=3, hard_seg=mode == "hard")

        def noisy(c):
            c = np.asarray(c, np.float64) + rng.uniform(0, 0.3, np.shape(c))
            return c / c.sum(1, keepdims=True)

        sf_conf = noisy(np.concatenate([sc.sf_seg_conf, sc.sf_seg_conf[dup]]))
        new_conf = noisy(sc.tgt_seg_conf)
        # ED nodes: class distribution of the pixel they project to
        e = np.exp(sc.img_seg_conf.astype(np.float64))
        img = e / e.sum(0, keepdims=True)
        g_ = sc.f64("ed_points")
        u = np.clip(np.rint(g_[:, 0] * sc.K[0, 0] / g_[:, 2] + sc.K[0, 2]).astype(int), 0, sc.W - 1)
        v = np.clip(np.rint(g_[:, 1] * sc.K[1, 1] / g_[:, 2] + sc.K[1, 2]).astype(int), 0, sc.H - 1)
        ed_conf = noisy(img[:, v, u].T)
        b.update(sf_seg=np.argmax(sf_conf, 1), sf_seg_conf=sf_conf, sf_dist2edge=rng.uniform(0, 20, n),
                 new_seg=np.argmax(new_conf, 1), new_seg_conf=new_conf, new_dist2edge=rng.uniform(0, 20, sc.T),
                 ed_seg=np.argmax(ed_conf, 1), ed_seg_conf=ed_conf)
        assert np.bincount(b["ed_seg"], minlength=3).min() >= 4
        kw = dict(seg=b["sf_seg"], seg_conf=b["sf_seg_conf"], dist2edge=b["sf_dist2edge"], ed_seg=b["ed_seg"],
                  ed_seg_conf=b["ed_seg_conf"])
    sf, inputs, sfdata = _objects(b, okw, seg=seg)
    fusion.fuseInputData(sf, inputs, sfdata)
    m = fuo.Model(b["sf_points"], b["sf_norms"], b["sf_colors"], b["sf_radii"], b["sf_confs"], b["sf_time_stamp"],
                  b["sf_isStable"], b["sf_knn_idx"], b["sf_knn_w"], b["ed_points"], b["ed_radii"], **kw)
    new = SimpleNamespace(points=b["new_points"], norms=b["new_norms"], colors=b["new_colors"], radii=b["new_radii"],
                          confs=b["new_confs"], valid=b["new_valid"])
    if seg:
        new.seg, new.seg_conf, new.dist2edge = b["new_seg"], b["new_seg_conf"], b["new_dist2edge"]
    opt = fuo.default_opt(height=sc.H, width=sc.W, **okw)
    fuo.fuse_input_data(m, opt, b["K"], new, 41)
    names = ("points", "norms", "colors", "radii", "confs", "time_stamp", "isStable", "knn_indices", "knn_w",
             "projdata") + (("seg", "seg_conf", "dist2edge") if seg else ())
    ref = {"x_" + k: getattr(m, k) for k in names}
    _check(sf, ref, "x_")
    assert len(sf.points) > n and (~sf.isStable[:n].cpu().numpy() & b["sf_isStable"]).sum() > 1000
    fusion.prepareStableIndexNSwapAllModel(sf, inputs, sfdata)
    fuo.swap_stable(m, opt, 41)
    _check(sf, {"y_" + k: getattr(m, k) for k in names}, "y_")
    w = sf.knn_w.cpu().numpy()
    np.testing.assert_allclose(w.sum(1), 1.0, rtol=0, atol=1e-12)


@pytest.mark.parametrize("gold", [GOLD, GOLD_K6])
@pytest.mark.parametrize("mode", ["plain", "sem", "hard"])
def test_knn_feeder_semantic_branches_match_reference(mode, gold):
    """update_ed / update_sfed_knn (frame 0) with hard_seg class-restricted neighbours and the
    Jensen-Shannon weights of Semantic-SuPer, against the reference's goldens (num_neighbors 4 and 6)."""
    import torch
    from super_amd import nodes
    g = np.load(gold)
    b = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    t = lambda a: torch.from_numpy(np.array(a, copy=True)).cuda()
    opt = SimpleNamespace(method="super" if mode == "plain" else "semantic-super", num_neighbors=int(b.get("num_neighbors", 4)), num_ED_neighbors=4,
                          num_classes=int(b["num_classes"]))
    ed = SimpleNamespace(points=t(b["ed_points"]), radii=t(b["ed_radii"]), seg=t(b["ed_seg"]), seg_conf=t(b["ed_seg_conf"]))
    sf = SimpleNamespace(opt=opt, hard_seg=mode == "hard", ED_nodes=ed, points=t(b["sf_points"]), seg=t(b["sf_seg"]),
                         seg_conf=t(b["sf_seg_conf"]), isStable=torch.ones(len(b["sf_points"]), dtype=torch.bool).cuda())
    nodes.update_ed(sf)
    nodes.update_sfed_knn(sf)
    tol = 1e-6 if mode == "plain" else 1e-13          # the plain path stores float32 weights (LM feeder)
    np.testing.assert_array_equal(ed.knn_indices.cpu().numpy(), g[f"knn_{mode}_ed_idx"])
    np.testing.assert_array_equal(sf.knn_indices.cpu().numpy(), g[f"knn_{mode}_sf_idx"])
    np.testing.assert_allclose(ed.knn_w.cpu().numpy(), g[f"knn_{mode}_ed_w"], rtol=0, atol=1e-6 if mode != "hard" else 1e-13)
    np.testing.assert_allclose(sf.knn_w.cpu().numpy(), g[f"knn_{mode}_sf_w"], rtol=0, atol=tol)
    np.testing.assert_array_equal(sf.isStable.cpu().numpy(), g[f"knn_{mode}_sf_stable"])
    assert sf.knn_indices.dtype == torch.long and sf.knn_w.dtype == torch.float64


def test_class_restricted_knn_with_too_few_nodes_fails_loudly():
    import torch
    from super_amd import nodes
    from super_amd._lib import SuperLMError
    p = torch.rand(50, 3, dtype=torch.float64).cuda()
    n = torch.rand(10, 3, dtype=torch.float64).cuda()
    s1 = torch.zeros(50, dtype=torch.long).cuda()
    s2 = torch.tensor([0, 0, 0, 1, 1, 1, 1, 1, 1, 1]).cuda()
    with pytest.raises(SuperLMError, match="fewer nodes"):
        nodes.find_knn(p, n, num_classes=2, seg1=s1, seg2=s2, k=4)
    d, i = nodes.find_knn(p, n, num_classes=2, seg1=torch.ones(50, dtype=torch.long).cuda(), seg2=s2, k=4)
    assert int(i.min()) >= 3 and bool((d[:, 1:] >= d[:, :-1]).all())


def test_hard_seg_with_too_few_nodes_of_a_class_fails_loudly():
    """The reference asserts len(p2) >= k per class (utils/utils.py:237); the device path reports it."""
    from super_amd import fusion
    from super_amd._lib import SuperLMError
    g = np.load(GOLD)
    b = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    b["ed_seg"] = b["ed_seg"].copy()
    keep = np.nonzero(b["ed_seg"] == 2)[0][:3]
    b["ed_seg"][b["ed_seg"] == 2] = 0
    b["ed_seg"][keep] = 2                      # class 2 keeps 3 nodes only
    sf, inputs, sfdata = _objects(b, VARIANTS["hard"], seg=True)
    with pytest.raises(SuperLMError, match="at least num_neighbors ED nodes"):
        fusion.fuseInputData(sf, inputs, sfdata)


def test_whole_frame_pipeline_depth_lm_update_fusion():
    """SuPer.forward for two tracked frames with every stage on the device (super/super.py:39-77):
    depth_preprocessing -> LM_Solver.LM -> Surfels.update -> fuseInputData ->
    prepareStableIndexNSwapAllModel, chained through the mirrors on one `sf` object.  Each stage is
    checked against its oracle evaluated on the state the stage actually received."""
    import torch
    from oracle import depth_oracle as dpo, fusion_oracle as fuo, lm_oracle as orc
    from super_amd import fusion, nodes, synth
    from super_amd.LM import LM_Solver
    from super_amd.data_loader import depth_preprocessing
    H, W = 60, 80
    K = synth._scaled_intrinsics(H, W)
    inv_K = np.linalg.pinv(K)
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    rng = np.random.default_rng(8)
    color = rng.uniform(0, 255, (3, H, W)).astype(np.float32)
    opt = SimpleNamespace(height=H, width=W, data="superv1", load_valid_mask=False, depth_model="monodepth2",
                          dilate_invalid_kernel=0, normal_model="naive", phase="test", method="super",
                          sf_point_plane=True, mesh_arap=True, mesh_rot=True, sf_point_plane_weight=1.0,
                          mesh_arap_weight=10.0, mesh_rot_weight=1.0, num_optimize_iterations=5,
                          use_derived_gradient=True, num_neighbors=4, num_ED_neighbors=4,
                          th_dist=0.02, th_cosine_ang=0.4, th_time_steps=30, disable_merging_new_surfels=False,
                          disable_merging_exist_surfels=False, disable_adding_new_surfels=False,
                          disable_removing_unstable_surfels=False)

    def frame_inputs(k):
        depth = synth._surface(uu, vv, H, W, 0.3 + 0.04 * k).astype(np.float32)
        depth[:3] = 0.0
        depth[:, :3] = 0.0            # a border without depth
        return depth, {("depth", 0): torch.from_numpy(depth.copy())[None, None].cuda(), ("disp", 0): torch.zeros(1, 1, H, W).cuda(),
                       "inv_K": torch.from_numpy(inv_K)[None], "K": torch.from_numpy(K)[None],
                       ("color", 0): torch.from_numpy(color)[None].cuda(), "divterm": 1.0 / (2 * 0.6 * 0.6),
                       "filename": ["%06d" % k], "time": k, "ID": torch.tensor([k])}

    # ---- frame 0: the model starts as the first frame's points; nodes = every 7th valid pixel on a grid
    _, inputs = frame_inputs(0)
    data, inputs = depth_preprocessing(opt, None, inputs)
    im = data.index_map.cpu().numpy()
    node_rows = im[6:H - 6:7, 6:W - 6:7].reshape(-1)
    node_rows = torch.from_numpy(node_rows[node_rows >= 0]).cuda()
    ed = SimpleNamespace(points=data.points[node_rows].clone(), norms=data.norms[node_rows].clone())
    ed.num, ed.param_num = int(ed.points.shape[0]), 7 * int(ed.points.shape[0])
    d, _ = nodes.find_knn(ed.points, ed.points, k=5)
    ed.radii = d[:, 1:].mean(1)
    n0 = int(data.points.shape[0])
    sf = SimpleNamespace(opt=opt, hard_seg=False, evaluate_tracking=False, ED_nodes=ed, points=data.points.clone(),
                         norms=data.norms.clone(), colors=data.colors.clone(), radii=data.radii.clone(),
                         confs=data.confs.clone(), time_stamp=torch.zeros(n0, device="cuda"),
                         isStable=torch.ones(n0, dtype=torch.bool, device="cuda"), projdata=torch.zeros(n0, 2, device="cuda"),
                         time=0)
    nodes.update_ed(sf)
    nodes.update_sfed_knn(sf)
    lm = LM_Solver(opt)
    oopt = orc.default_opt(num_optimize_iterations=5)
    for k in (1, 2):
        depth, inputs = frame_inputs(k)
        data, inputs = depth_preprocessing(opt, None, inputs)
        ref = dpo.depth_preprocessing(dpo.default_opt(height=H, width=W), depth, K, inv_K, color, inputs["divterm"])
        np.testing.assert_array_equal(data.index_map.cpu().numpy(), ref["index_map"])
        np.testing.assert_array_equal(data.points.cpu().numpy(), ref["points"])
        # LM on the device vs the oracle on the float32-rounded state the HIP path streams
        f32r = lambda t: t.cpu().numpy().astype(np.float32).astype(np.float64)
        fr = orc.Frame(sf_points=f32r(sf.points), sf_knn_idx=sf.knn_indices.cpu().numpy(), sf_knn_w=f32r(sf.knn_w),
                       ed_points=f32r(ed.points), ed_knn_idx=ed.knn_indices.cpu().numpy(), tgt_points=f32r(data.points),
                       tgt_norms=f32r(data.norms), index_map=data.index_map.cpu().numpy(), valid=data.valid.cpu().numpy(),
                       K=K, H=H, W=W)
        beta = lm.LM(sf, inputs, data)
        np.testing.assert_allclose(beta.cpu().numpy(), orc.lm(fr, oopt), rtol=0, atol=1e-4, err_msg=f"frame {k}")
        nodes.update(sf, beta)
        # fusion on the device vs the oracle on the same pre-fusion state
        m = fuo.Model(*(getattr(sf, a).cpu().numpy() for a in ("points", "norms", "colors", "radii", "confs", "time_stamp",
                                                               "isStable", "knn_indices", "knn_w")),
                      ed.points.cpu().numpy(), ed.radii.cpu().numpy())
        new = SimpleNamespace(**{a: getattr(data, a).cpu().numpy() for a in ("points", "norms", "colors", "radii", "confs", "valid")})
        fopt = fuo.default_opt(height=H, width=W, th_dist=opt.th_dist)
        fuo.fuse_input_data(m, fopt, K, new, k)
        fuo.swap_stable(m, fopt, k)
        n_before = int(sf.points.shape[0])
        fusion.fuseInputData(sf, inputs, data)
        fusion.prepareStableIndexNSwapAllModel(sf, inputs, data)
        _check(sf, {"z_" + a: getattr(m, a) for a in ("points", "norms", "colors", "radii", "confs", "time_stamp", "isStable",
                                                        "knn_indices", "knn_w", "projdata")}, "z_")
        assert int(sf.points.shape[0]) >= n_before * 0.9 and bool(sf.isStable.all())
        rec = lm.last_records[0]
        assert rec[-1]["loss"] < rec[0]["loss"]


@pytest.mark.parametrize("derived", [True, False])
def test_super_driver_tracks_a_deforming_surface(derived):
    """tests/driver_harness.FrameLoop (the stage mirrors in the reference driver's call order, super/super.py) over 5 frames of a deforming synthetic surface,
    LM path and first-order (GraphFit / Adam) path: the surfel model follows the surface (its points
    re-project onto the newest depth map to within a fraction of the inter-frame motion), the ED graph
    comes from the grid mesh, the model stays compact."""
    import torch
    from super_amd import synth
    from driver_harness import FrameLoop as SuPer
    H, W = 96, 128
    K = synth._scaled_intrinsics(H, W)
    inv_K = np.linalg.pinv(K)
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    color = np.random.default_rng(2).uniform(0, 255, (3, H, W)).astype(np.float32)
    opt = SimpleNamespace(height=H, width=W, data="superv1", load_valid_mask=False, depth_model="monodepth2",
                          dilate_invalid_kernel=0, normal_model="naive", phase="test", method="super", load_depth=True,
                          deform_udpate_method="super_edg", mesh_step_size=8, use_derived_gradient=derived,
                          sf_point_plane=True, mesh_arap=True, mesh_rot=True, mesh_face=False, sf_point_plane_weight=1.0,
                          mesh_arap_weight=10.0, mesh_rot_weight=1.0, mesh_face_weight=1.0, num_optimize_iterations=10,
                          optimizer="Adam", learning_rate=2e-4, num_neighbors=4, num_ED_neighbors=4, th_dist=0.02,
                          th_cosine_ang=0.4, th_time_steps=30, disable_merging_new_surfels=False,
                          disable_merging_exist_surfels=False, disable_adding_new_surfels=False,
                          disable_removing_unstable_surfels=False)
    model = SuPer(opt)
    models = SimpleNamespace()
    depth_of = lambda k: (synth._surface(uu, vv, H, W, 0.3 + 0.03 * k)).astype(np.float32)
    counts, errs = [], []
    for k in range(5):
        depth = depth_of(k)
        depth[:4] = 0.0
        depth[:, :4] = 0.0
        inputs = {("depth", 0): torch.from_numpy(depth.copy())[None, None], ("disp", 0): torch.zeros(1, 1, H, W),
                  "inv_K": torch.from_numpy(inv_K)[None], "K": torch.from_numpy(K)[None],
                  ("color", 0): torch.from_numpy(color)[None], "divterm": torch.tensor(1.0 / (2 * 0.6 * 0.6)),
                  "filename": ["%06d" % k], "time": k, "ID": torch.tensor([k])}
        deform = model(models, inputs)
        sf = model.sf
        assert torch.isfinite(sf.points).all() and bool(sf.isStable.all())
        counts.append(int(sf.points.shape[0]))
        # model points against the newest depth map at their own pixels
        P = sf.points.cpu().numpy()
        u = np.rint(P[:, 0] * K[0, 0] / P[:, 2] + K[0, 2]).astype(int)
        v = np.rint(P[:, 1] * K[1, 1] / P[:, 2] + K[1, 2]).astype(int)
        ok = (u >= 5) & (u < W - 1) & (v >= 5) & (v < H - 1)
        errs.append(float(np.abs(P[ok, 2] - depth[v[ok], u[ok]]).mean()))
        if k == 0:
            assert deform is None and sf.ED_nodes.num > 50 and sf.ED_nodes.triangles.shape[0] == 3
        else:
            assert deform.shape == ((sf.ED_nodes.num, 7) if derived else (sf.ED_nodes.num + 1, 7))
    motion = float(np.abs(depth_of(4) - depth_of(3))[5:, 5:].mean())      # inter-frame depth change
    print("tracking error per frame", errs, "inter-frame motion", motion)
    assert max(errs[1:]) < (0.7 if derived else 1.2) * motion, (errs, motion)
    assert counts[-1] < 1.3 * counts[0]


@pytest.mark.parametrize("hard", [False, True])
def test_semantic_super_driver_runs_end_to_end(hard):
    """opt.method == "semantic-super" through the driver mirror: depth_preprocessing with segmentation inputs,
    grid-mesh graph with node classes (class-boundary edges dropped under hard_seg), Surfels with
    Jensen-Shannon / class-restricted skinning, GraphFit with the soft segmentation point-plane weight,
    fusion carrying seg / seg_conf / dist2edge.  Properties: every field stays consistent in length, class
    ids follow the fused confidences, hard_seg neighbours share the surfel's class at initialisation, the
    model follows the surface."""
    import torch
    from super_amd import synth
    from driver_harness import FrameLoop as SuPer
    H, W, C = 96, 128, 3
    K = synth._scaled_intrinsics(H, W)
    inv_K = np.linalg.pinv(K)
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    rng = np.random.default_rng(7)
    color = rng.uniform(0, 255, (3, H, W)).astype(np.float32)
    opt = SimpleNamespace(height=H, width=W, data="superv2", load_valid_mask=False, depth_model="monodepth2",
                          dilate_invalid_kernel=0, normal_model="naive", phase="test", method="semantic-super",
                          load_depth=True, num_classes=C, del_seg_classes=[], hard_seg=hard,
                          deform_udpate_method="super_edg", mesh_step_size=8, use_derived_gradient=False,
                          sf_point_plane=False, sf_soft_seg_point_plane=not hard, sf_hard_seg_point_plane=hard,
                          mesh_arap=True, mesh_rot=True, mesh_face=True, sf_point_plane_weight=1.0,
                          mesh_arap_weight=10.0, mesh_rot_weight=1.0, mesh_face_weight=1.0, num_optimize_iterations=10,
                          optimizer="Adam", learning_rate=2e-4, num_neighbors=4, num_ED_neighbors=4, th_dist=0.02,
                          th_cosine_ang=0.4, th_time_steps=30, disable_merging_new_surfels=False,
                          disable_merging_exist_surfels=False, disable_adding_new_surfels=False,
                          disable_removing_unstable_surfels=False)
    model = SuPer(opt)
    models = SimpleNamespace()
    depth_of = lambda k: (synth._surface(uu, vv, H, W, 0.3 + 0.03 * k)).astype(np.float32)
    # three vertical class bands, smooth logits
    logits = np.stack([-((uu - c) / 18.0) ** 2 for c in (20.0, 64.0, 108.0)], 0) * 3.0
    errs = []
    for k in range(4):
        depth = depth_of(k)
        depth[:4] = 0.0
        depth[:, :4] = 0.0
        lg = (logits + rng.normal(0, 0.05, logits.shape)).astype(np.float64)
        inputs = {("depth", 0): torch.from_numpy(depth.copy())[None, None], ("disp", 0): torch.zeros(1, 1, H, W),
                  "inv_K": torch.from_numpy(inv_K)[None], "K": torch.from_numpy(K)[None],
                  ("color", 0): torch.from_numpy(color)[None], "divterm": torch.tensor(1.0 / (2 * 0.6 * 0.6)),
                  ("seg_conf", 0): torch.from_numpy(lg)[None], ("seg", 0): torch.from_numpy(np.argmax(lg, 0))[None, None],
                  "filename": ["%06d" % k], "time": k, "ID": torch.tensor([k])}
        deform = model(models, inputs)
        sf = model.sf
        n = int(sf.points.shape[0])
        assert sf.seg.shape == (n,) and sf.seg_conf.shape == (n, C) and sf.dist2edge.shape == (n,)
        assert sf.knn_w.shape == (n, 4) and torch.isfinite(sf.knn_w).all() and torch.isfinite(sf.points).all()
        np.testing.assert_allclose(sf.seg_conf.sum(1).cpu().numpy(), 1.0, rtol=0, atol=1e-12)
        np.testing.assert_array_equal(sf.seg.cpu().numpy(), sf.seg_conf.argmax(1).cpu().numpy())
        np.testing.assert_allclose(sf.knn_w.sum(1).cpu().numpy(), 1.0, rtol=0, atol=1e-12)
        ed = sf.ED_nodes
        if k == 0:
            assert deform is None and ed.num > 50 and ed.seg.shape == (ed.num,) and len(torch.unique(ed.seg)) == C
            if hard:
                assert bool((ed.seg[ed.edge_index[0]] == ed.seg[ed.edge_index[1]]).all())
                assert bool((ed.seg[sf.knn_indices] == sf.seg[:, None]).all())
                assert bool((ed.seg[ed.knn_indices] == ed.seg[:, None]).all())
        else:
            assert deform.shape == (ed.num + 1, 7) and torch.isfinite(deform).all()
        P = sf.points.cpu().numpy()
        u = np.rint(P[:, 0] * K[0, 0] / P[:, 2] + K[0, 2]).astype(int)
        v = np.rint(P[:, 1] * K[1, 1] / P[:, 2] + K[1, 2]).astype(int)
        ok = (u >= 5) & (u < W - 1) & (v >= 5) & (v < H - 1)
        errs.append(float(np.abs(P[ok, 2] - depth[v[ok], u[ok]]).mean()))
    motion = float(np.abs(depth_of(3) - depth_of(2))[5:, 5:].mean())
    print("semantic driver: tracking error per frame", errs, "inter-frame motion", motion)
    assert max(errs[1:]) < 1.5 * motion, (errs, motion)
