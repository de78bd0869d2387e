"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A float64 PyTorch-CPU restatement of the reference's DEFAULT per-frame optimiser, the
autograd path ``GraphFit`` (``super/deform_mesh.py:198-230,251-379`` with the loss terms of
``super/loss.py:293-401,458-473,502-505``): skin the stable surfels with the local warps,
apply the global row T_g, evaluate face / ARAP / Rot / point-to-plane / flow-correspondence losses, backprop,
scale the global row's gradient by 1/J, step SGD(momentum 0.9) or Adam; plus the
Semantic-SuPer terms of the same function: the hard / soft segmentation weight on the
point-to-plane residuals (``loss.py:346-399``), the optional ``max`` clip (``loss.py:369-370``)
and the semantic-boundary morphing term (``deform_mesh.py:126-194``).  This is the path
BASELINE.json names as the reported CPU baseline (``configs[0]``); ``bench.py`` times it on
the host cores, and it is the parity reference for the hand-derived-gradient HIP version.

Pinned against the reference itself by ``tests/golden/make_golden.py`` (``gf_*`` arrays in
the fixtures): per-term losses and d(loss)/d(deform_verts) at iteration 0, and the final
``deform_verts`` after 10 iterations of SGD and of Adam; ``s60x80_j48_semantic`` pins the
semantic terms the same way and ``s60x80_j48_corr`` (make_golden_corr.py) the flow-correspondence term.  Only ``tests/`` and ``bench.py``'s
``cpu_baseline`` leg import this module.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

F64 = torch.float64


def _t(a, dt=F64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt)


def qrot(q, x):
    """R(q)x for an un-normalised quaternion, broadcasting (super/utils.py:49-54)."""
    w, v = q[..., 0:1], q[..., 1:4]
    c = torch.cross(v.expand_as(x), x, dim=-1)
    return x + 2.0 * w * c + 2.0 * torch.cross(v.expand_as(x), c, dim=-1)


class Problem:
    """Tensors GraphFit reads from ``src`` (Surfels), ``trg`` (Data) and ``inputs``."""

    def __init__(self, sc, stable=None):
        st = np.ones(sc.N, bool) if stable is None else np.asarray(stable, bool)
        self.J = sc.J
        self.p = _t(sc.sf_points)[st]
        self.idx = _t(sc.sf_knn_idx, torch.long)[st]
        self.w = _t(sc.sf_knn_w)[st]
        self.g = _t(sc.ed_points)
        self.e_idx = _t(sc.ed_knn_idx, torch.long)
        self.e_w = _t(sc.ed_knn_w)
        self.tri = _t(sc.ed_triangles, torch.long) if sc.ed_triangles is not None else None
        self.tri_area = _t(sc.ed_triangle_areas) if sc.ed_triangle_areas is not None else None
        self.o = _t(sc.tgt_points)
        self.n = _t(sc.tgt_norms)
        self.index_map = _t(sc.index_map, torch.long)
        self.H, self.W = sc.H, sc.W
        K = sc.K
        self.fx, self.fy, self.cx, self.cy = (float(K[0, 0]), float(K[1, 1]), float(K[0, 2]),
                                               float(K[1, 2]))
        self.C = int(getattr(sc, "num_classes", 0) or 0)
        if self.C:
            self.sf_seg = _t(sc.sf_seg, torch.long)[st]            # src.seg[isStable]
            self.sf_seg_conf = _t(sc.sf_seg_conf)[st]              # src.seg_conf[isStable]
            self.tgt_seg_conf = _t(sc.tgt_seg_conf)                # trg.seg_conf (T,C)
            self.img_seg_conf = _t(sc.img_seg_conf)[None]          # inputs[("seg_conf",0)] (1,C,H,W)
            self.img_seg = _t(sc.img_seg, torch.long)              # inputs[("seg",0)][0,0]
            self.edge_pts = None
        fl = getattr(sc, "flow", None)                             # models.optical_flow(src.rgb, color) (1,2,H,W) f32
        self.flow = None if fl is None else torch.from_numpy(np.ascontiguousarray(fl, dtype=np.float32))


def deform(pb: Problem, dv):
    """``deform_source`` (deform_mesh.py:198-230): local skinning then the global row.
    Returns the deformed node positions (J,3) and surfel points (N,3)."""
    gk = pb.g[pb.idx]                                    # (N,K,3)
    bk = dv[:-1][pb.idx]                                 # (N,K,7)
    t = qrot(bk[..., 0:4], pb.p[:, None, :] - gk) + bk[..., 4:7] + gk
    sf = (pb.w[..., None] * t).sum(1)
    qg, bg = dv[-1:, 0:4], dv[-1:, 4:7]
    verts = qrot(qg, pb.g + dv[:-1, 4:7]) + bg
    sf = qrot(qg, sf) + bg
    return verts, sf


def kld(P, Q, eps=1e-13):
    """``KLD`` (utils/utils.py:244-250)."""
    return (P * (P / (Q + eps) + eps).log()).sum(-1)


def jsd(P, Q, eps=1e-13):
    """``JSD`` (utils/utils.py:252-254)."""
    M = 0.5 * (P + Q)
    return 0.5 * (kld(P, M, eps) + kld(Q, M, eps))


def point_plane(pb: Problem, sf, seg_mode=None, pp_max=None):
    """``DataLoss.autograd_forward`` with ``loss_type='point-plane'`` (loss.py:293-401):
    margin-1 validity on ROUNDED projections, 4 taps all mapped (zero fill), weights
    differentiable through (u,v), sum of (n.(p-o))^2.  ``seg_mode`` 'soft' / 'hard' multiplies
    each squared residual by the detached semantic weight (loss.py:379-399); ``pp_max`` drops
    squared residuals >= max (loss.py:369-370, only without segmentation)."""
    Z = sf[:, 2] + 1e-8
    u_ = sf[:, 0] * pb.fx / Z + pb.cx
    v_ = sf[:, 1] * pb.fy / Z + pb.cy
    ur, vr = torch.round(u_).long(), torch.round(v_).long()
    ok = (vr >= 1) & (vr < pb.H - 2) & (ur >= 1) & (ur < pb.W - 2)
    u, v, P = u_[ok], v_[ok], sf[ok]
    fv, cv, fu, cu = torch.floor(v), torch.ceil(v), torch.floor(u), torch.ceil(u)
    nb = torch.stack([fv, fv, cv, cv], -1)
    mb = torch.stack([fu, cu, fu, cu], -1)
    rows = pb.index_map[nb.long(), mb.long()]            # (M,4)
    tap_ok = (rows >= 0).all(-1)
    feats = [pb.o, pb.n] + ([pb.tgt_seg_conf] if seg_mode else [])
    feat = torch.cat(feats, -1)
    U = torch.zeros(rows.shape + (feat.shape[-1],), dtype=F64)
    U[rows >= 0] = feat[rows[rows >= 0]]
    an = torch.clamp(1 - torch.abs(nb - v[:, None]), min=0)[..., None]
    am = torch.clamp(1 - torch.abs(mb - u[:, None]), min=0)[..., None]
    out = (U * an * am).sum(-2)
    o, n = out[:, 0:3], out[:, 3:6]
    losses = (n[tap_ok] * (P[tap_ok] - o[tap_ok])).sum(-1) ** 2
    if seg_mode:
        with torch.no_grad():
            tconf = out[:, 6:].softmax(1)                   # trg.seg_conf is softmaxed AGAIN (loss.py:357)
            if seg_mode == "soft":
                wgt = torch.exp(-0.1 * jsd(pb.sf_seg_conf[ok], tconf))
            else:
                wgt = (pb.sf_seg[ok] == torch.argmax(tconf, dim=1)).to(F64)
        losses = losses * wgt[tap_ok]
    elif pp_max is not None:
        losses = losses[losses < pp_max]
    return losses.sum(), int(losses.numel())


def corr_term(pb: Problem, sf, loss_type="point-point"):
    """``DataLoss.autograd_forward(..., flow=flow, loss_type=opt.sf_corr_loss_type)`` (loss.py:293-345,401 as
    called at deform_mesh.py:100-109): the UNROUNDED projections are shifted by the optical flow sampled at
    them -- ``F.grid_sample`` on a float32 grid, bilinear, zero padding, align_corners=False, differentiable
    through the grid -- validity is margin 1 on the shifted FLOAT coordinates, then the same 4-tap gather;
    'point-point' sums |p - o|^2, 'point-plane' (n.(p - o))^2."""
    Z = sf[:, 2] + 1e-8
    u_ = sf[:, 0] * pb.fx / Z + pb.cx
    v_ = sf[:, 1] * pb.fy / Z + pb.cy
    grid = torch.stack([u_ * 2 / pb.W - 1, v_ * 2 / pb.H - 1], dim=1).view(1, -1, 1, 2).float()
    loc = F.grid_sample(pb.flow, grid, mode="bilinear", padding_mode="zeros", align_corners=False)[0, :, :, 0]
    u_ = u_ + loc[0]
    v_ = v_ + loc[1]
    ok = (v_ >= 1) & (v_ < pb.H - 2) & (u_ >= 1) & (u_ < pb.W - 2)
    u, v, P = u_[ok], v_[ok], sf[ok]
    fv, cv, fu, cu = torch.floor(v), torch.ceil(v), torch.floor(u), torch.ceil(u)
    nb = torch.stack([fv, fv, cv, cv], -1)
    mb = torch.stack([fu, cu, fu, cu], -1)
    rows = pb.index_map[nb.long(), mb.long()]
    tap_ok = (rows >= 0).all(-1)
    feat = torch.cat([pb.o, pb.n], -1)
    U = torch.zeros(rows.shape + (6,), dtype=F64)
    U[rows >= 0] = feat[rows[rows >= 0]]
    an = torch.clamp(1 - torch.abs(nb - v[:, None]), min=0)[..., None]
    am = torch.clamp(1 - torch.abs(mb - u[:, None]), min=0)[..., None]
    out = (U * an * am).sum(-2)
    o, n = out[:, 0:3], out[:, 3:6]
    d = P[tap_ok] - o[tap_ok]
    if loss_type == "point-point":
        losses = (d ** 2).sum(-1)
    elif loss_type == "point-plane":
        losses = (n[tap_ok] * d).sum(-1) ** 2
    else:
        raise ValueError(loss_type)
    return losses.sum(), int(losses.numel())


def edge_points(img_seg, num_classes, kernel=3, margin=1):
    """Per class, the (x,y) pixels of that class with a pixel of another class in their
    kernel x kernel neighbourhood, image border of ``kernel`` px dropped, row-major order
    (``find_edge_region`` utils/utils.py:276-301 as called at deform_mesh.py:149-165)."""
    seg = np.asarray(img_seg)
    H, W = seg.shape
    r = kernel // 2
    out = []
    for c in range(num_classes):
        m = seg == c
        other = np.pad(~m, r, constant_values=False)      # zero padding: outside is not "other"
        near = np.zeros((H, W), bool)
        for dy in range(kernel):
            for dx in range(kernel):
                near |= other[dy:dy + H, dx:dx + W]
        e = near & m
        e[:kernel] = e[-kernel:] = False
        e[:, :kernel] = e[:, -kernel:] = False
        ey, ex = np.nonzero(e)
        keep = (ex >= margin) & (ex < W - 1 - margin) & (ey >= margin) & (ey < H - 1 - margin)
        out.append(torch.from_numpy(np.stack([ex[keep], ey[keep]], 1).astype(np.float64)).reshape(-1, 2))
    return out


def bn_morph(pb: Problem, sf):
    """Semantic-boundary morphing term (deform_mesh.py:126-194): surfels whose class differs
    from the class of the pixel they project to are pulled towards the 2 nearest boundary pixels
    of their own class; mean over the surfels whose mean squared distance exceeds 15.
    Returns None when no class contributes a list entry, else the (possibly NaN) mean."""
    Z = sf[:, 2] + 1e-8
    x = sf[:, 0] * pb.fx / Z + pb.cx
    y = sf[:, 1] * pb.fy / Z + pb.cy
    grid = torch.stack([x, y], 1)
    sg = torch.stack([x / pb.W * 2 - 1, y / pb.H * 2 - 1], 1)
    with torch.no_grad():
        new_seg = F.grid_sample(pb.img_seg_conf, sg[None, :, None, :].detach(), align_corners=False
                                )[0, :, :, 0].argmax(0)
    val = (new_seg != pb.sf_seg) & (sg[:, 0] > -1) & (sg[:, 0] < 1) & (sg[:, 1] > -1) & (sg[:, 1] < 1)
    if pb.edge_pts is None:
        pb.edge_pts = edge_points(pb.img_seg.numpy(), pb.C)
    parts = []
    for c in range(pb.C):
        cm = (pb.sf_seg == c) & val
        E = pb.edge_pts[c]
        if not bool(cm.any()) or len(E) == 0:
            continue
        g = grid[cm]
        d2 = ((g[:, None, :] - E[None, :, :]) ** 2).sum(-1)
        order = torch.argsort(d2.detach(), dim=1, stable=True)[:, :2]
        kd = torch.sqrt(torch.gather(d2.detach(), 1, order))
        dte = torch.minimum(torch.minimum(g.min(1).values, pb.W - g[:, 0]), pb.H - g[:, 1])
        ok = ~torch.any(kd > dte[:, None], dim=1)
        li = ((E[order][ok] - g[ok][:, None, :]) ** 2).sum(2).mean(1)
        parts.append(li[li > 15])
    if not parts:
        return None
    return torch.cat(parts).mean()


def arap(pb: Problem, dv_local):
    """``ARAPLoss.autograd_forward`` (loss.py:458-473): weighted by the node KNN weights."""
    d = pb.g[:, None, :] - pb.g[pb.e_idx]
    bk = dv_local[pb.e_idx]
    # quirk kept: the reference subtracts the float32-ROUNDED edge vector (loss.py:468)
    r = qrot(bk[..., 0:4], d) + bk[..., 4:7] - d.float().double() - dv_local[:, None, 4:7]
    return (pb.e_w * (r ** 2).sum(-1)).sum()


def rot(dv):
    """``RotLoss.autograd_forward`` over all J+1 rows (loss.py:502-505)."""
    return ((1.0 - (dv[:, 0:4] ** 2).sum(-1)) ** 2).sum()


def face(pb: Problem, verts):
    """Face term (deform_mesh.py:51-60)."""
    c = torch.cross(verts[pb.tri[1]] - verts[pb.tri[0]], verts[pb.tri[2]] - verts[pb.tri[0]], dim=1)
    a = 0.5 * torch.sqrt((c ** 2).sum(1) + 1e-13)
    return ((a - pb.tri_area) ** 2).sum()


def default_opt(**kw):
    o = SimpleNamespace(sf_point_plane=True, sf_point_plane_weight=1.0, mesh_arap=True,
                        mesh_arap_weight=10.0, mesh_rot=True, mesh_rot_weight=1.0, mesh_face=False,
                        mesh_face_weight=1.0, num_optimize_iterations=10, optimizer="SGD",
                        learning_rate=5e-5, sf_soft_seg_point_plane=False, sf_hard_seg_point_plane=False,
                        sf_bn_morph=False, sf_bn_morph_weight=1.0, depth_model="monodepth2", sf_corr=False,
                        sf_corr_weight=0.001, sf_corr_loss_type="point-point")
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def total_loss(pb: Problem, dv, opt):
    """``get_losses`` (deform_mesh.py:25-196) for the geometric terms; returns (loss, dict)."""
    verts, sf = deform(pb, dv)
    terms = {}
    if opt.mesh_face:
        terms["face_losses"] = opt.mesh_face_weight * face(pb, verts)
    if opt.mesh_arap:
        terms["arap_loss"] = opt.mesh_arap_weight * arap(pb, dv[:-1])
    if opt.mesh_rot:
        terms["rot_loss"] = opt.mesh_rot_weight * rot(dv)
    soft = getattr(opt, "sf_soft_seg_point_plane", False)
    hard = getattr(opt, "sf_hard_seg_point_plane", False)
    if opt.sf_point_plane or soft or hard:
        seg_mode = "soft" if soft else ("hard" if hard else None)     # soft wins (loss.py:384)
        pp_max = 2e-5 if (seg_mode is None and getattr(opt, "depth_model", "") == "raft_stereo") else None
        pp, m = point_plane(pb, sf, seg_mode, pp_max)
        terms["point_plane_loss"] = opt.sf_point_plane_weight * pp
        terms["_matched"] = m
    if getattr(opt, "sf_corr", False):
        cl, m = corr_term(pb, sf, getattr(opt, "sf_corr_loss_type", "point-point"))
        terms["corr_loss"] = opt.sf_corr_weight * cl
        terms["_corr_matched"] = m
    if getattr(opt, "sf_bn_morph", False):
        bm = bn_morph(pb, sf)
        if bm is not None:
            terms["sf_bn_morph_loss"] = opt.sf_bn_morph_weight * bm
    loss = sum(v for k, v in terms.items() if not k.startswith("_"))
    return loss, terms


def graphfit(pb: Problem, opt, trace=None):
    """``deform_superedg`` (deform_mesh.py:251-379): identity init of (J+1,7), Niter steps of
    SGD(momentum .9) / Adam at lr = opt.learning_rate with the global row's gradient / J."""
    dv = torch.zeros((pb.J + 1, 7), dtype=F64)
    dv[:, 0] = 1.0
    dv.requires_grad_(True)
    if opt.optimizer == "SGD":
        optim = torch.optim.SGD([dv], lr=opt.learning_rate, momentum=0.9)
    elif opt.optimizer == "Adam":
        optim = torch.optim.Adam([dv], lr=opt.learning_rate)
    else:
        raise ValueError(opt.optimizer)
    for it in range(opt.num_optimize_iterations):
        optim.zero_grad()
        loss, terms = total_loss(pb, dv, opt)
        loss.backward()
        dv.grad[-1] = dv.grad[-1] / pb.J
        if trace is not None:
            trace.append(dict(it=it, loss=float(loss), grad=dv.grad.detach().clone().numpy(),
                              terms={k: (float(v) if torch.is_tensor(v) else v) for k, v in terms.items()}))
        optim.step()
    return dv.detach().numpy()
