"""``slm_prepare_model`` / ``LM_Solver.prepare_model``: the model-side half of a bind, ahead of the frame and asynchronous
(replaces the model-side part of ``loss_term.prepare``, reference ``super/loss.py:212-220,408-426``,
moved to where its inputs become final, ``super/super.py:66-73``).  Through the C ABI, on an MI355X (-m gpu).

* a prepared model + the bind of the target gives the state a plain bind gives (bitwise on the run-to-run reproducible
  data path; against the reference's goldens on the default path);
* a preparation serves ONE bind of the SAME arrays: another model, or the same tensors modified in place, fall back to a
  full bind with the right result;
* an error of the preparation (a surfel KNN index outside [0, J)) is reported by the bind that consumes it; the slot
  stays unbound in between (``slm_run`` refuses it);
* a sequence of frames with the model changing every frame (what the driver does) never mixes up preparations.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import load_golden, ref_opt, torch_frame

pytestmark = pytest.mark.gpu


def _solver(opt, **kw):
    from super_amd.LM import LM_Solver
    o = ref_opt(opt)
    for k, v in kw.items():
        setattr(o, k, v)
    return LM_Solver(o)


@pytest.mark.parametrize("name", ["s60x80_j48", "s120x160_j108", "s60x80_j48_reject"])
def test_prepared_model_then_lm_equals_lm_alone_bitwise(name):
    g, sc, opt = load_golden(name)
    sf, inputs, new_data = torch_frame(sc)
    plain = _solver(opt, slm_data_path=2)
    want = plain.LM(sf, inputs, new_data).cpu().numpy()
    ahead = _solver(opt, slm_data_path=2)
    ahead.prepare_model(sf)
    assert ahead._prepared[0] is not None
    got = ahead.LM(sf, inputs, new_data).cpu().numpy()
    assert ahead._prepared[0] is None                       # consumed
    np.testing.assert_array_equal(got, want)
    assert [r["loss"] for r in ahead.last_records[0]] == [r["loss"] for r in plain.last_records[0]]
    np.testing.assert_allclose(got, g["lm_beta"], rtol=0, atol=1e-7)
    # ... and the next LM() without a preparation binds in full again
    again = ahead.LM(sf, inputs, new_data).cpu().numpy()
    np.testing.assert_array_equal(again, want)


def test_default_data_path_against_the_reference_golden():
    g, sc, opt = load_golden("s120x160_j108")
    sf, inputs, new_data = torch_frame(sc)
    lm = _solver(opt)
    lm.prepare_model(sf)
    beta = lm.LM(sf, inputs, new_data).cpu().numpy()
    np.testing.assert_allclose(beta, g["lm_beta"], rtol=0, atol=1e-7)
    np.testing.assert_allclose([r["loss"] for r in lm.last_records[0]], g["lm_loss"], rtol=1e-6, atol=1e-12)
    assert [r["accepted"] for r in lm.last_records[0]] == [bool(a) for a in g["lm_accepted"]]


def test_a_preparation_for_another_model_is_discarded():
    import torch
    g1, sc1, opt = load_golden("s60x80_j48")
    g2, sc2, _ = load_golden("s120x160_j108")
    f1, f2 = torch_frame(sc1), torch_frame(sc2)
    lm = _solver(opt)
    lm.prepare_model(f2[0])                                 # prepared for frame 2's model ...
    b1 = lm.LM(*f1).cpu().numpy()                           # ... but frame 1 is solved: full bind
    np.testing.assert_allclose(b1, g1["lm_beta"], rtol=0, atol=1e-7)
    # the same tensors modified IN PLACE after the preparation: the stamp (tensor version) no longer matches
    lm.prepare_model(f2[0])
    f2[0].points.add_(0.0)                                  # an in-place op: same values, new version
    b2 = lm.LM(*f2).cpu().numpy()
    np.testing.assert_allclose(b2, g2["lm_beta"], rtol=0, atol=1e-7)
    # C level: the library compares the model pointers itself
    from super_amd import _lib
    from super_amd.LM import BoundFrame, ModelView, _stream_ptr
    h = lm._handle()
    mv = ModelView(f2[0])
    fr = _lib.SlmFrame()
    mv.fill(fr)
    _lib.check(lm.lib.slm_prepare_model(h, 0, C.byref(fr), _stream_ptr(mv.device)), "prepare")
    assert lm.lib.slm_run(h, 1, _stream_ptr(mv.device)) == _lib.SLM_ERR_UNBOUND      # prepared, not bound
    bf = BoundFrame(*f1)                                    # other arrays
    _lib.check(lm.lib.slm_bind_frame(h, 0, C.byref(bf.c), _stream_ptr(bf.device)), "bind")
    _lib.check(lm.lib.slm_run(h, 1, _stream_ptr(bf.device)), "run")
    beta = torch.empty((bf.J, 7), dtype=torch.float64, device=bf.device)
    _lib.check(lm.lib.slm_get_beta(h, 0, beta.data_ptr(), _stream_ptr(bf.device)), "beta")
    np.testing.assert_allclose(beta.cpu().numpy(), g1["lm_beta"], rtol=0, atol=1e-7)


@pytest.mark.parametrize("data_path", [0, 2])
def test_no_copy_model_rewritten_in_place_after_the_preparation(data_path):
    """with inputs that need NO conversion copy (int32 KNN tables, float64 contiguous state) the ModelView
    made by ``LM()`` holds the very pointers ``prepare_model`` passed, so the library's pointer comparison cannot see an
    in-place rewrite -- the mirror must drop the preparation (``slm_discard_prepared``), otherwise the bind consumes a
    plan sorted from the OLD values.  The model is changed for real: surfel rows permuted (points, KNN rows and weights
    together: the same problem in another surfel order, so the stale tuple-sorted streams are wrong row for row)."""
    import torch
    g, sc, opt = load_golden("s120x160_j108")
    sf, inputs, new_data = torch_frame(sc)
    sf.knn_indices = sf.knn_indices.to(torch.int32)          # no-copy inputs
    sf.ED_nodes.knn_indices = sf.ED_nodes.knn_indices.to(torch.int32)
    lm = _solver(opt, slm_data_path=data_path)
    lm.prepare_model(sf)
    mv = lm._prepared[0][1]
    assert mv.sf_points.data_ptr() == sf.points.data_ptr() and mv.sf_knn_idx.data_ptr() == sf.knn_indices.data_ptr()
    torch.cuda.synchronize()                                 # (the worker has read the old values)
    perm = torch.from_numpy(np.random.default_rng(5).permutation(sc.N)).to(sf.points.device)
    sf.points.copy_(sf.points[perm])
    sf.knn_indices.copy_(sf.knn_indices[perm])
    sf.knn_w.copy_(sf.knn_w[perm])
    sf.norms.copy_(sf.norms[perm])
    got = lm.LM(sf, inputs, new_data).cpu().numpy()
    fresh = _solver(opt, slm_data_path=data_path)
    want = fresh.LM(sf, inputs, new_data).cpu().numpy()
    if data_path == 2:
        np.testing.assert_array_equal(got, want)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)
    np.testing.assert_allclose(got, g["lm_beta"], rtol=0, atol=1e-7)   # a permutation of the surfels: the same problem
    # C level: slm_discard_prepared makes the next bind a full one although every pointer is the prepared one
    from super_amd import _lib
    from super_amd.LM import BoundFrame, ModelView, _stream_ptr
    h = lm._handle()
    mv = ModelView(sf)
    fr = _lib.SlmFrame()
    mv.fill(fr)
    _lib.check(lm.lib.slm_prepare_model(h, 0, C.byref(fr), _stream_ptr(mv.device)), "prepare")
    torch.cuda.synchronize()
    inv = torch.argsort(perm)
    for t in (sf.points, sf.knn_indices, sf.knn_w, sf.norms):
        t.copy_(t[inv])                                      # back to the original order, in place
    _lib.check(lm.lib.slm_discard_prepared(h, 0), "discard")
    bf = BoundFrame(sf, inputs, new_data, model=mv)          # the prepared pointers
    _lib.check(lm.lib.slm_bind_frame(h, 0, C.byref(bf.c), _stream_ptr(bf.device)), "bind")
    _lib.check(lm.lib.slm_run(h, 1, _stream_ptr(bf.device)), "run")
    beta = torch.empty((bf.J, 7), dtype=torch.float64, device=bf.device)
    _lib.check(lm.lib.slm_get_beta(h, 0, beta.data_ptr(), _stream_ptr(bf.device)), "beta")
    np.testing.assert_allclose(beta.cpu().numpy(), g["lm_beta"], rtol=0, atol=1e-7)
    assert lm.lib.slm_discard_prepared(h, 5) == _lib.SLM_ERR_INVALID


def test_an_error_of_the_preparation_surfaces_at_the_bind():
    from super_amd import _lib
    g, sc, opt = load_golden("s60x80_j48")
    sf, inputs, new_data = torch_frame(sc)
    sf.knn_indices = sf.knn_indices.clone()
    sf.knn_indices[5, 2] = sc.J + 3                         # an IndexError in the reference
    lm = _solver(opt)
    lm.prepare_model(sf)                                    # returns at once: the worker finds it
    with pytest.raises(_lib.SuperLMError, match="KNN index"):
        lm.LM(sf, inputs, new_data)
    sf2, inputs2, nd2 = torch_frame(sc)                     # the solver is still usable
    np.testing.assert_allclose(lm.LM(sf2, inputs2, nd2).cpu().numpy(), g["lm_beta"], rtol=0, atol=1e-7)


def test_a_sequence_with_the_model_changing_every_frame():
    """Alternating models of different sizes, each prepared ahead right after the previous solve (the driver's order):
    every solve must be the one a fresh solver gives, and the solver must survive its destruction with a preparation
    still queued."""
    names = ["s60x80_j48", "s120x160_j108", "s60x80_j48_reject", "s120x160_j108", "s60x80_j48"]
    data = {n: load_golden(n) for n in set(names)}
    opt = data[names[0]][2]
    lm = _solver(opt)
    frames = [torch_frame(data[n][1]) for n in names]
    lm.prepare_model(frames[0][0])
    for k, n in enumerate(names):
        g, _, o = data[n]
        if (o.sf_point_plane, o.mesh_arap, o.mesh_rot) != (opt.sf_point_plane, opt.mesh_arap, opt.mesh_rot):
            continue
        beta = lm.LM(*frames[k]).cpu().numpy()
        np.testing.assert_allclose(beta, g["lm_beta"], rtol=0, atol=1e-7, err_msg=f"frame {k} ({n})")
        if k + 1 < len(names):
            lm.prepare_model(frames[k + 1][0])
    lm.prepare_model(frames[0][0])
    del lm                                                   # slm_destroy with a queued preparation
