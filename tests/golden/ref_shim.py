"""Import shim for the UNMODIFIED reference hot-path modules (build container only).

The reference (``/root/reference``, read-only, never copied) depends on wheels
that are not installed here and hard-codes ``.cuda()``.  This shim (SURVEY.md
Appendix C) injects stub modules, neutralises the device calls and then imports
``super.LM`` / ``super.loss`` / ``super.utils`` / ``super.nodes`` so that
``make_golden.py`` can run the reference's own arithmetic on CPU and record
golden vectors.  It is never imported on the GPU box (``/root/reference`` does
not exist there) and nothing in the product or the ``-m gpu`` tests uses it.
"""
from __future__ import annotations

import logging
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"


class Data(SimpleNamespace):
    """Stand-in for ``torch_geometric.data.Data`` (attribute bag with ``items()``)."""

    def items(self):
        return self.__dict__.items()

    def keys(self):
        return self.__dict__.keys()

    def __getitem__(self, k):
        return self.__dict__[k]

    def __setitem__(self, k, v):
        self.__dict__[k] = v


def _knn_points(p1, p2, K=1, **_):
    """``pytorch3d.ops.knn_points`` semantics used by the reference
    (``utils/utils.py:217``): squared L2, ascending.  Ties: lowest index."""
    d2 = ((p1[0][:, None, :] - p2[0][None, :, :]) ** 2).sum(-1)
    order = torch.argsort(d2, dim=1, stable=True)[:, :K]
    return torch.gather(d2, 1, order)[None], order[None], None


class _SummaryWriter:
    def __init__(self, *a, **k):
        self.scalars = []

    def add_scalar(self, *a, **k):
        self.scalars.append(a)

    def add_image(self, *a, **k):
        pass

    def add_images(self, *a, **k):
        pass


_installed = False


def install():
    """Idempotently install the stubs and device patches, then return the
    reference modules as a namespace."""
    global _installed
    if not _installed:
        def mod(name, **attrs):
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
            return m

        p3d = mod("pytorch3d")
        p3d.ops = mod("pytorch3d.ops", knn_points=_knn_points, ball_query=None)
        p3d.transforms = mod("pytorch3d.transforms", quaternion_to_matrix=None,
                             matrix_to_quaternion=None)
        tv = mod("torchvision")
        tv.utils = mod("torchvision.utils", make_grid=None)
        tv.transforms = mod("torchvision.transforms")
        tv.models = mod("torchvision.models", ResNet=type("ResNet", (), {}))
        tg = mod("torch_geometric")
        tg.data = mod("torch_geometric.data", Data=Data)
        mod("cv2")
        if "torch.utils.tensorboard" not in sys.modules:
            tb = mod("torch.utils.tensorboard", SummaryWriter=_SummaryWriter)
            torch.utils.tensorboard = tb

        torch.Tensor.cuda = lambda self, *a, **k: self
        _orig_tensor = torch.tensor

        def _tensor(*a, **k):
            k.pop("device", None)
            return _orig_tensor(*a, **k)
        torch.tensor = _tensor
        torch.cuda.empty_cache = lambda: None
        if REFERENCE_ROOT not in sys.path:
            sys.path.insert(0, REFERENCE_ROOT)
        _installed = True

    import super.LM as ref_LM            # noqa: E402
    import super.loss as ref_loss        # noqa: E402
    import super.utils as ref_utils      # noqa: E402
    import super.nodes as ref_nodes      # noqa: E402
    import utils.utils as ref_uutils     # noqa: E402
    import super.deform_mesh as ref_dm    # noqa: E402
    return SimpleNamespace(LM=ref_LM, loss=ref_loss, utils=ref_utils, nodes=ref_nodes,
                           uutils=ref_uutils, deform_mesh=ref_dm)


def _structural_similarity(im1, im2, channel_axis=None, full=False, **kw):
    """scikit-image is not installed here (and not version-pinned by the reference): the restatement of
    its published algorithm in ``oracle/depth_oracle.py::skimage_ssim_full`` stands in, so that the
    reference's own warp (Project3D + grid_sample) and confidence blend run and get recorded.  The SSIM
    values themselves are therefore NOT pinned by the reference (said so in the oracle and DESIGN.md)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import depth_oracle
    assert channel_axis == 0 and full and not kw
    S = depth_oracle.skimage_ssim_full(np.asarray(im1), np.asarray(im2))
    return float(S.mean()), S


def install_data_loader():
    """Additionally import the reference's ``utils.data_loader`` (for ``depth_preprocessing``,
    SURVEY.md 8f row f2): stubs for the image / network modules it imports at module level."""
    ref = install()

    def mod(name, **attrs):
        if name in sys.modules:
            sys.modules[name].__dict__.update(attrs)
            return sys.modules[name]
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    sk = mod("skimage")
    sk.io = mod("skimage.io")
    sk.metrics = mod("skimage.metrics", structural_similarity=_structural_similarity)
    pil = mod("PIL")
    pil.Image = mod("PIL.Image")
    mod("depth.raft_core")
    mod("depth.raft_core.utils")
    mod("depth.raft_core.utils.utils", depth_to_point_cloud=None)
    mod("seg")
    mod("seg.inference", generate_mask=None)
    import utils.data_loader as ref_dl   # noqa: E402
    ref.data_loader = ref_dl
    return ref


def torch_frame(sc, frame_id=1):
    """Build the reference-side ``sf`` / ``inputs`` / ``new_data`` objects (f64/i64
    torch tensors) from a ``super_amd.synth.Scene``."""
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
    ed = Data(points=t(sc.ed_points), norms=t(sc.ed_norms), radii=t(sc.ed_radii),
              knn_indices=t(sc.ed_knn_idx, torch.long), knn_w=t(sc.ed_knn_w),
              num=sc.J, param_num=7 * sc.J)
    log = logging.getLogger("ref_shim")          # defect D1: reference never creates sf.logger
    sf = SimpleNamespace(points=t(sc.sf_points), norms=t(sc.sf_norms),
                         knn_indices=t(sc.sf_knn_idx, torch.long), knn_w=t(sc.sf_knn_w),
                         ED_nodes=ed, logger=log)
    inputs = {("color", 0): torch.zeros(1, 3, sc.H, sc.W), "K": torch.from_numpy(sc.K)[None],
              "ID": torch.tensor([frame_id])}
    new_data = Data(points=t(sc.tgt_points), norms=t(sc.tgt_norms),
                    index_map=t(sc.index_map, torch.long), valid=t(sc.valid, torch.bool))
    return sf, inputs, new_data


def graphfit_frame(sc, stable=None):
    """``src`` / ``trg`` / ``models`` objects for the reference's GraphFit (autograd path)."""
    sf, inputs, new_data = torch_frame(sc)
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
    sf.isStable = torch.ones(sc.N, dtype=torch.bool) if stable is None else t(stable, torch.bool)
    sf.colors = torch.zeros(sc.N, 3)
    sf.time = 0
    sf.summary_writer = _SummaryWriter()
    sf.ED_nodes.triangles = t(sc.ed_triangles, torch.long)
    sf.ED_nodes.triangles_areas = t(sc.ed_triangle_areas)
    new_data.time = 1
    if getattr(sc, "num_classes", 0):      # Semantic-SuPer inputs (data_loader.py:319-331,455-457,494-496)
        sf.seg = t(sc.sf_seg, torch.long)
        sf.seg_conf = t(sc.sf_seg_conf)
        new_data.seg_conf = t(sc.tgt_seg_conf)
        inputs[("seg_conf", 0)] = t(sc.img_seg_conf)[None]
        inputs[("seg", 0)] = t(sc.img_seg, torch.long)[None, None]
    models = SimpleNamespace(renderer=lambda inputs, data, rad=None: torch.zeros(sc.H, sc.W, 3))
    return sf, inputs, new_data, models


def ref_opt(**kw):
    o = SimpleNamespace(sf_point_plane=True, sf_point_plane_weight=1.0, mesh_arap=True,
                        mesh_arap_weight=10.0, mesh_rot=True, mesh_rot_weight=1.0,
                        num_optimize_iterations=10, phase="test", use_derived_gradient=True,
                        num_neighbors=4, num_ED_neighbors=4, method="super", optimizer="SGD",
                        learning_rate=5e-5, mesh_face=False, mesh_face_weight=1.0, sf_corr=False,
                        sf_corr_match_renderimg=False, deform_udpate_method="super_edg",
                        renderer_rad=0.0002, depth_model="monodepth2", save_sample_freq=1000000,
                        sf_hard_seg_point_plane=False, sf_soft_seg_point_plane=False,
                        sf_bn_morph=False, sf_bn_morph_weight=0.1, num_classes=3)
    for k, v in kw.items():
        setattr(o, k, v)
    return o
