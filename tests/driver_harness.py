"""TEST HARNESS (not product code): runs the stage mirrors in the order the reference's per-frame driver calls
them (``SuPer.forward`` -> ``init_surfels`` / ``fusion``, ``super/super.py:23-73``), so that the tests and
``tools/time_driver.py`` can exercise whole sequences through libsuper_lm.so.  A reference user does NOT need
this file: INTEGRATION.md section 2 binds the stage mirrors onto the reference's own ``SuPer`` / ``Surfels``.

    loop = FrameLoop(opt)                       # opt: the reference's option names
    for inputs in frames:                       # inputs[("depth",0)], inputs["K"], inputs["inv_K"], ...
        warp = loop(models, inputs)             # loop.sf = the surfel model, warp = the frame's deform parameters

Stage order per frame: depth_preprocessing -> [frame 0: ED graph, model, KNN feeders, first swap]
                       -> LM or GraphFit -> update -> fuseInputData -> swap.
Depth must be loaded (no depth / segmentation networks here).
"""
from __future__ import annotations

import logging

import torch

from super_amd import fusion, nodes
from super_amd.LM import LM_Solver
from super_amd.data_loader import depth_preprocessing
from super_amd.deform_mesh import GraphFit
from super_amd.graph_encoder import DirectDeformGraph


class SurfelModel:
    """Attribute bag with the field names the stage mirrors read (those of the reference's ``Surfels``,
    ``super/nodes.py:96-149``); the stage mirrors are attached as methods so that ``evaluation`` and the fusion
    mirrors can call each other through the model exactly as they would on the reference object."""
    update_ed = nodes.update_ed
    update_sfed_knn = nodes.update_sfed_knn
    update = nodes.update
    fuseInputData = fusion.fuseInputData
    prepareStableIndexNSwapAllModel = fusion.prepareStableIndexNSwapAllModel


def first_frame_model(opt, models, inputs, data) -> SurfelModel:
    """frame 0: every field of the target frame becomes a model field; stability, time stamps, projections and
    the two skinning tables are initialised (what ``Surfels.__init__`` sets up, ``nodes.py:96-149``)."""
    m = SurfelModel()
    m.opt, m.models, m.evaluate_tracking = opt, models, False
    m.logger = logging.getLogger("driver_harness")
    m.hard_seg = bool(getattr(opt, "hard_seg", False))
    if getattr(opt, "method", "super") == "semantic-super":
        m.power_arg = (0.5, 0.5)
    for name, value in vars(data).items():
        if name != "valid":
            setattr(m, name, value)
    n, dev = int(m.points.shape[0]), m.points.device
    m.sf_num = n
    m.isStable = torch.ones(n, dtype=torch.bool, device=dev)
    if opt.phase == "test":
        m.time_stamp = torch.full((n,), float(m.time), device=dev)
    yx = data.valid.view(opt.height, opt.width).nonzero()
    m.projdata = torch.flip(yx, dims=[-1]).to(torch.float32)       # (x, y) of every valid pixel
    m.update_ed()
    m.update_sfed_knn()
    return m


class FrameLoop:
    def __init__(self, opt):
        if not getattr(opt, "load_depth", True):
            raise NotImplementedError("driver_harness: depth must be loaded (no depth network here)")
        self.opt, self.sf, self.deform_param = opt, None, None
        self.lm = LM_Solver(opt) if opt.use_derived_gradient else None
        self.graph_fit = None if opt.use_derived_gradient else GraphFit(opt)

    def _to_device(self, inputs):
        for key, v in list(inputs.items()):
            if not torch.is_tensor(v) or key == "filename":
                continue
            inputs[key] = v.item() if key == "divterm" else v.cuda()
        return inputs

    def __call__(self, models, inputs):
        opt = self.opt
        target, inputs = depth_preprocessing(opt, models, self._to_device(inputs))
        test = opt.phase == "test"
        if self.sf is None:
            if getattr(opt, "deform_udpate_method", "super_edg") == "super_edg":
                encoder = getattr(models, "mesh_encoder", None) or DirectDeformGraph(opt)
                target.ED_nodes = encoder(inputs, target)
            self.sf = first_frame_model(opt, models, inputs, target)
            if test:
                self.sf.prepareStableIndexNSwapAllModel(inputs, target)
            if self.lm is not None and getattr(opt, "slm_prepare_ahead", True):
                self.lm.prepare_model(self.sf)
            self.deform_param = None
            return None
        if self.lm is not None:
            warp = self.lm.LM(self.sf, inputs, target)
        else:
            warp = self.graph_fit(inputs, self.sf, target, models)
        self.sf.update(warp)
        if test:
            self.sf.fuseInputData(inputs, target)
            self.sf.prepareStableIndexNSwapAllModel(inputs, target)
        # the model is final for the next frame: its model-side prepare starts now, on the library's worker thread and
        # stream, while the caller fetches the next frame (optional -- LM() alone does it all)
        if self.lm is not None and getattr(opt, "slm_prepare_ahead", True):
            self.lm.prepare_model(self.sf)
        self.deform_param = warp
        return warp
