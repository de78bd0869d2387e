"""Generate golden vectors by running the REFERENCE itself (build container only).

    python tests/golden/make_golden.py [scene names ...]

imports the unmodified reference hot-path modules from ``/root/reference``
through ``ref_shim`` and records, for small seeded synthetic scenes
(``super_amd.synth``), the inputs at the drop-in boundary and the reference's
outputs: per-term residuals and sparse Jacobians (captured by wrapping
``LossTool.prepare_jtj_jtl``), dense JtJ / jtl from ``prepareCostTerm``, the
match mask (from ``pcd2depth`` + ``bilinear_intrpl_block`` outputs), the
per-iteration ``{loss, u, accepted, beta}`` trace of ``LM_Solver.LM``, its final
beta, ``Surfels.update`` outputs and the KNN feeder outputs.  The ``.npz`` files
are committed; the reference never leaves this box (fixtures are data only).
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

torch.set_num_threads(1)  # bit-deterministic fixtures

SCENES = {
    # name: (scene kwargs, lm opt kwargs, store dense JtJ)
    # target border wider than the source border + 2 % target holes: some surfels
    # project onto invalid pixels / unmapped taps, so the match set is a strict subset
    "s60x80_j48": (dict(N=1500, J=48, H=60, W=80, seed=1, src_border=4, tgt_border=6,
                        tgt_holes=0.02), dict(), True),
    "s120x160_j108": (dict(N=6000, J=108, H=120, W=160, seed=2, src_border=6, tgt_border=8,
                           tgt_holes=0.01), dict(), False),
    "s60x80_j48_dataonly": (dict(N=1500, J=48, H=60, W=80, seed=3, src_border=5, tgt_border=2),
                            dict(mesh_arap=False, mesh_rot=False), False),
    # large motion (dphi) so that some LM steps are rejected
    "s60x80_j48_reject": (dict(N=1500, J=48, H=60, W=80, seed=4, src_border=5, tgt_border=2,
                               dphi=0.9), dict(), False),
    # num_neighbors = 6 (README.md:175 / options.py:49: a documented tunable; loss.py:213-220 and utils.py:30-36 are
    # K-generic): 42-wide Jacobian rows, 21 coupled node pairs per surfel
    "s60x80_j48_k6": (dict(N=1500, J=48, H=60, W=80, seed=6, src_border=5, tgt_border=3, tgt_holes=0.01,
                           n_neighbors=6), dict(num_neighbors=6), True),
    # Semantic-SuPer: 3 wavy class bands, target segmentation moved 6 px (configs[4] terms)
    "s60x80_j48_semantic": (dict(N=1500, J=48, H=60, W=80, seed=5, src_border=5, tgt_border=2,
                                 semantic=True), dict(), False),
}

# GraphFit variants recorded per scene: tag -> opt overrides
GF_VARIANTS = {
    "plain": (("sgd", dict(optimizer="SGD")), ("adam", dict(optimizer="Adam")),
              ("sgdface", dict(optimizer="SGD", mesh_face=True))),
    "semantic": (
        # the configs[4] trio: soft-seg point-plane + face + boundary morphing
        ("soft", dict(optimizer="SGD", sf_point_plane=False, sf_soft_seg_point_plane=True,
                      sf_hard_seg_point_plane=False, mesh_face=True, sf_bn_morph=True)),
        ("softadam", dict(optimizer="Adam", sf_point_plane=False, sf_soft_seg_point_plane=True,
                          sf_hard_seg_point_plane=False, mesh_face=True, sf_bn_morph=True,
                          learning_rate=1e-4)),
        ("hard", dict(optimizer="SGD", sf_point_plane=False, sf_soft_seg_point_plane=False,
                      sf_hard_seg_point_plane=True)),
        ("morph", dict(optimizer="SGD", sf_point_plane=True, sf_bn_morph=True, sf_bn_morph_weight=1e-6)),
        ("clip", dict(optimizer="SGD", depth_model="raft_stereo")),
    ),
}


def _np(t):
    return t.detach().cpu().numpy().copy()


def capture_terms(ref, solver, sf, inputs, new_data, beta):
    """Run prepareCostTerm(grad=True) at ``beta`` recording each term's sparse J and r."""
    rec = []
    orig = ref.loss.LossTool.prepare_jtj_jtl

    def wrapped(Jacobian, loss):
        Jc = Jacobian.coalesce()
        rec.append(dict(idx=_np(Jc.indices()).astype(np.int32), val=_np(Jc.values()).astype(np.float64),
                        shape=tuple(Jc.shape), r=_np(loss).reshape(-1).astype(np.float64)))
        return orig(Jacobian, loss)

    # match-mask capture: outputs of the reference's own projection + bilinear gather
    cap = {}
    orig_p2d = ref.loss.pcd2depth
    orig_bil = ref.loss.LossTool.bilinear_intrpl_block

    def p2d(inp, pcd, **kw):
        out = orig_p2d(inp, pcd, **kw)
        cap["proj"] = tuple(_np(o) for o in out)
        cap["T"] = _np(pcd)
        return out

    def bil(v, u, target_, **kw):
        out = orig_bil(v, u, target_, **kw)
        cap.setdefault("bil", []).append(_np(out[0]))
        return out

    ref.loss.LossTool.prepare_jtj_jtl = staticmethod(wrapped)
    ref.loss.pcd2depth = p2d
    ref.loss.LossTool.bilinear_intrpl_block = staticmethod(bil)
    try:
        for t in solver.losses:
            t.prepare(sf, new_data)
        jtj, jtl = solver.prepareCostTerm(sf, inputs, new_data, beta, grad=True)
    finally:
        ref.loss.LossTool.prepare_jtj_jtl = staticmethod(orig)
        ref.loss.pcd2depth = orig_p2d
        ref.loss.LossTool.bilinear_intrpl_block = staticmethod(orig_bil)
    loss = solver.prepareCostTerm(sf, inputs, new_data, beta, grad=False)

    out = dict(jtj=_np(jtj), jtl=_np(jtl).reshape(-1), loss=float(loss))
    names = [n for n, on in (("data", solver.opt.sf_point_plane), ("arap", solver.opt.mesh_arap),
                             ("rot", solver.opt.mesh_rot)) if on]
    for n, r in zip(names, rec):
        out[f"{n}_Jidx"], out[f"{n}_Jval"], out[f"{n}_r"] = r["idx"], r["val"], r["r"]
        out[f"{n}_Jshape"] = np.array(r["shape"])
    if "proj" in cap:
        v_, u_, coords, proj_valid = cap["proj"]
        valid = _np(new_data.valid)
        vp = valid[np.clip(coords, 0, len(valid) - 1)] & (coords >= 0) & (coords < len(valid))
        o, n = cap["bil"][0], cap["bil"][1]
        ok = ~(np.isnan(o).any(1) | np.isnan(n).any(1))
        cand = np.nonzero(vp)[0]
        # reference: sf_indicies = proj_valid[valid_pair][intrpl_valid] must be all-True (D2)
        assert proj_valid[cand][ok].all(), "scene triggers reference defect D2"
        out["match"] = cand[ok]
        out["T"] = cap["T"]
        out["v_"], out["u_"], out["coords"] = v_, u_, coords
        out["o"], out["n"] = o[ok], n[ok]
    return out


def capture_lm(ref, solver, sf, inputs, new_data):
    calls = []
    orig = solver.prepareCostTerm

    def wrapped(sf_, inputs_, new_data_, beta, grad=False):
        res = orig(sf_, inputs_, new_data_, beta, grad=grad)
        if grad:
            calls.append(("grad", _np(beta)))
        else:
            calls.append(("loss", _np(beta), float(res)))
        return res

    solver.prepareCostTerm = wrapped
    try:
        beta = solver.LM(sf, inputs, new_data)
    finally:
        solver.prepareCostTerm = orig
    losses = [c[2] for c in calls if c[0] == "loss"]
    betas_try = [c[1] for c in calls if c[0] == "loss"]
    betas_in = [c[1] for c in calls if c[0] == "grad"]
    u, v, best = 10.0, 7.5, 1e10
    us, acc = [], []
    for L in losses:
        us.append(u)
        if L < best:
            best = L
            u /= v
            acc.append(True)
        else:
            u *= v
            acc.append(False)
    return dict(lm_beta=_np(beta), lm_loss=np.array(losses), lm_u=np.array(us),
                lm_accepted=np.array(acc), lm_beta_try=np.stack(betas_try),
                lm_beta_in=np.stack(betas_in))


def capture_update(ref, sc, opt, beta):
    sf, _, _ = ref_shim.torch_frame(sc)
    sf.opt = opt
    ref.nodes.Surfels.update(sf, torch.from_numpy(beta))
    return dict(upd_points=_np(sf.points), upd_norms=_np(sf.norms),
                upd_ed_points=_np(sf.ED_nodes.points), upd_ed_norms=_np(sf.ED_nodes.norms))


def capture_knn(ref, sc, opt):
    sf, _, _ = ref_shim.torch_frame(sc)
    sf.opt, sf.hard_seg = opt, False
    sf.isStable = torch.ones(sc.N, dtype=torch.bool)
    ref.nodes.Surfels.update_ed(sf)
    ref.nodes.Surfels.update_sfed_knn(sf)
    return dict(knn_sf_idx=_np(sf.knn_indices), knn_sf_w=_np(sf.knn_w),
                knn_sf_stable=_np(sf.isStable), knn_ed_idx=_np(sf.ED_nodes.knn_indices),
                knn_ed_w=_np(sf.ED_nodes.knn_w))


def capture_graphfit(ref, sc, okw, variants):
    """Reference autograd path (GraphFit): iteration-0 losses and gradient (after the 1/J
    scaling of the global row) and the final deform_verts, per option variant."""
    out = {}
    for tag, extra in variants:
        kw = dict(okw)
        kw.update(extra)
        opt = ref_shim.ref_opt(**kw)
        if getattr(sc, "num_classes", 0):
            opt.num_classes, opt.width, opt.height = sc.num_classes, sc.W, sc.H
        src, inputs, trg, models = ref_shim.graphfit_frame(sc)
        gf = ref.deform_mesh.GraphFit(opt)
        rec = {}
        orig = gf.get_losses

        def wrapped(deform_verts, *a, **k):
            loss, losses = orig(deform_verts, *a, **k)
            if "loss0" not in rec:
                rec["loss0"] = float(loss)
                rec["terms0"] = {n: float(v) for n, v in losses.items()}
                g, = torch.autograd.grad(loss, deform_verts, retain_graph=True)
                g = g.clone()
                g[-1] = g[-1] / src.ED_nodes.num
                rec["grad0"] = _np(g)
            return loss, losses

        gf.get_losses = wrapped
        dv = gf(inputs, src, trg, models)
        out[f"gf_{tag}_final"] = _np(dv)
        if tag == "sgd":   # Surfels.update, autograd variant (global row), on the reference
            sfu, _, _ = ref_shim.torch_frame(sc)
            sfu.opt = ref_shim.ref_opt(use_derived_gradient=False, num_neighbors=okw.get("num_neighbors", 4))
            ref.nodes.Surfels.update(sfu, dv.detach())
            out["gf_upd_points"], out["gf_upd_norms"] = _np(sfu.points), _np(sfu.norms)
            out["gf_upd_ed_points"], out["gf_upd_ed_norms"] = _np(sfu.ED_nodes.points), _np(sfu.ED_nodes.norms)
        out[f"gf_{tag}_loss0"] = rec["loss0"]
        out[f"gf_{tag}_grad0"] = rec["grad0"]
        for n, v in rec["terms0"].items():
            out[f"gf_{tag}_term_{n}"] = v
    return out


def main():
    ref = ref_shim.install()
    for name, (skw, okw, store_jtj) in SCENES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        sc = synth.make_scene(**skw)
        opt = ref_shim.ref_opt(**okw)
        sf, inputs, new_data = ref_shim.torch_frame(sc)
        solver = ref.LM.LM_Solver(opt)
        g = dict(H=sc.H, W=sc.W, K=sc.K)
        for f in ("sf_points", "sf_norms", "sf_knn_idx", "sf_knn_w", "ed_points", "ed_norms",
                  "ed_radii", "ed_knn_idx", "ed_knn_w", "tgt_points", "tgt_norms", "index_map",
                  "valid"):
            g["in_" + f] = getattr(sc, f)
        g["opt_flags"] = np.array([opt.sf_point_plane, opt.mesh_arap, opt.mesh_rot], dtype=bool)
        g["opt_weights"] = np.array([opt.sf_point_plane_weight, opt.mesh_arap_weight,
                                     opt.mesh_rot_weight])

        rng = np.random.default_rng(100 + skw["seed"])
        beta0 = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0]), (sc.J, 1))
        beta1 = beta0 + np.concatenate([rng.normal(0, 0.02, (sc.J, 4)),
                                        rng.normal(0, 0.004, (sc.J, 3))], axis=1)
        for tag, b in (("b0", beta0), ("b1", beta1)):
            t = capture_terms(ref, solver, sf, inputs, new_data, torch.from_numpy(b.copy()))
            # dense JtJ -> its non-zeros (same information, smaller file)
            nz = np.nonzero(t["jtj"])
            t["jtj_nz_idx"], t["jtj_nz_val"] = np.stack(nz).astype(np.int32), t["jtj"][nz]
            t.pop("jtj")
            if not store_jtj:      # big scenes: keep r / jtl / loss / match, drop J and JtJ
                for k in [k for k in t if "_Jidx" in k or "_Jval" in k or k.startswith("jtj_nz")]:
                    t.pop(k)
            for k in ("T", "v_", "u_", "coords"):
                if tag == "b1" or not store_jtj:
                    t.pop(k, None)
            g[f"{tag}_beta"] = b
            for k, v in t.items():
                g[f"{tag}_{k}"] = v
        g.update(capture_lm(ref, solver, sf, inputs, new_data))
        g.update(capture_update(ref, sc, opt, g["lm_beta"]))
        g.update(capture_knn(ref, sc, opt))
        g["in_ed_triangles"], g["in_ed_triangle_areas"] = sc.ed_triangles, sc.ed_triangle_areas
        if name in ("s60x80_j48", "s60x80_j48_reject", "s60x80_j48_k6"):   # (k6, round 6: deform_source / get_losses at num_neighbors = 6)
            g.update(capture_graphfit(ref, sc, okw, GF_VARIANTS["plain"]))
        if sc.num_classes:
            for f in ("img_seg_conf", "img_seg", "tgt_seg_conf", "sf_seg", "sf_seg_conf"):
                g["in_" + f] = getattr(sc, f)
            g["in_num_classes"] = sc.num_classes
            g.update(capture_graphfit(ref, sc, okw, GF_VARIANTS["semantic"]))
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **g)
        print(f"{name}: N={sc.N} J={sc.J} M(b0)={len(g['b0_match'])} "
              f"losses={np.array2string(g['lm_loss'], precision=4)} "
              f"accepted={g['lm_accepted'].astype(int)} -> {os.path.getsize(path)/1024:.0f} KB")


if __name__ == "__main__":
    main()
