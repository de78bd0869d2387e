"""slm_bind_frames: the frames of a batch bound concurrently (worker threads / streams inside the library) give
the same solver state as slm_bind_frame one by one.  Needs an MI355X (-m gpu)."""
import numpy as np
import pytest

from helpers import GOLDENS, load_golden

pytestmark = pytest.mark.gpu


def _frames(dev):
    from super_amd.engine import DeviceFrame
    out = []
    for name in GOLDENS + GOLDENS[:2] + GOLDENS:         # 10 frames: different sizes / plans, more than 8 workers
        g, sc, _ = load_golden(name)
        out.append((g if name in GOLDENS[:2] else None, DeviceFrame.from_scene(sc, dev)))     # [:2]: default options
    return out


@pytest.mark.parametrize("solver_path", [0, 3])
def test_batch_bind_matches_sequential_binds_and_the_goldens(solver_path):
    import torch
    from super_amd.engine import Engine
    dev = torch.device("cuda", 0)
    fr = _frames(dev)
    B = len(fr)
    kw = dict(max_frames=B, solver_path=solver_path)
    seq = Engine(dev, **kw)
    for i, (_, f) in enumerate(fr):
        seq.bind(i, f)
    seq.run(B)
    bat = Engine(dev, **kw)
    bat.bind_batch([f for _, f in fr])
    bat.run(B)
    torch.cuda.synchronize()
    for i, (g, _) in enumerate(fr):
        a, b = seq.beta(i).cpu().numpy(), bat.beta(i).cpu().numpy()
        np.testing.assert_allclose(b, a, rtol=0, atol=1e-11)
        if g is not None:
            np.testing.assert_allclose(b, g["lm_beta"], rtol=0, atol=1e-6)
        ra, rb = seq.records(i), bat.records(i)
        assert [r["status"] for r in ra] == [r["status"] for r in rb]
        assert [r["accepted"] for r in ra] == [r["accepted"] for r in rb]
    # binding again (steady state: plans cached) and a partial batch at an offset
    bat.bind_batch([f for _, f in fr[:3]], first=4)
    bat.run(B)
    torch.cuda.synchronize()
    for k in range(3):
        np.testing.assert_allclose(bat.beta(4 + k).cpu().numpy(), seq.beta(k).cpu().numpy(), rtol=0, atol=1e-11)
    seq.close()
    bat.close()


def test_batch_bind_reports_the_failing_frame():
    import ctypes as C
    import torch
    from super_amd import _lib
    from super_amd.engine import Engine
    dev = torch.device("cuda", 0)
    fr = _frames(dev)[:3]
    eng = Engine(dev, max_frames=3)
    cs = [f.c_struct() for _, f in fr]
    cs[1].K = 9                                           # num_neighbors must be in 1..8 (round 5: any K there, tests/test_gpu_num_neighbors.py)
    arr = (_lib.SlmFrame * 3)(*cs)
    rc = eng.lib.slm_bind_frames(eng.h, 0, 3, arr, eng.stream)
    assert rc != 0
    assert b"num_neighbors" in eng.lib.slm_last_error()
    eng.close()


def test_running_again_without_binding_keeps_the_first_records():
    """slm_run twice on the same binding continues the loop; the record buffer (num_iterations entries) keeps the
    first run's records and nothing is written past it (this used to fault after a few repetitions)."""
    import torch
    from super_amd.engine import Engine
    dev = torch.device("cuda", 0)
    (g, f), = _frames(dev)[:1]
    eng = Engine(dev, max_frames=1)
    eng.bind(0, f)
    eng.run(1)
    first = eng.records(0)
    beta1 = eng.beta(0).cpu().numpy()
    for _ in range(40):
        eng.run(1)
    torch.cuda.synchronize()
    assert eng.records(0) == first
    np.testing.assert_allclose(beta1, g["lm_beta"], rtol=0, atol=1e-6)
    assert np.isfinite(eng.beta(0).cpu().numpy()).all()
    eng.close()


def test_a_slot_rebound_with_a_much_larger_and_then_smaller_scene_matches_a_fresh_solver():
    """A slot's plan carries the tuple count of its last frame as the bound of the next preparation (no read-back);
    when the scene grows past the bound's head-room the preparation reruns with the exact count, and a shrinking scene
    is covered by the bound (a count above the bound used to run the preparation kernels past their buffers: a memory
    fault).  Either way the state is the one a fresh solver builds (1e-12: the loss sums of a large frame are not
    bit-reproducible from run to run)."""
    import torch
    from super_amd import synth
    from super_amd.engine import DeviceFrame, Engine
    dev = torch.device("cuda", 0)
    small = DeviceFrame.from_scene(synth.make_scene(seed=3, N=1500, J=96, H=120, W=160), dev)
    large = DeviceFrame.from_scene(synth.make_scene(seed=4, N=24000, J=256, H=240, W=320), dev)
    eng = Engine(dev, max_frames=1, num_iterations=4)
    got = []
    for f in (small, large, small, large):
        eng.bind(0, f)
        eng.run(1)
        got.append((eng.beta(0).cpu().numpy().copy(), eng.records(0)))
    eng.close()
    for k, f in enumerate((small, large)):
        ref = Engine(dev, max_frames=1, num_iterations=4)
        ref.bind(0, f)
        ref.run(1)
        want, recs = ref.beta(0).cpu().numpy(), ref.records(0)
        ref.close()
        assert any(r["accepted"] for r in recs)
        for j in (k, k + 2):
            np.testing.assert_allclose(got[j][0], want, rtol=0, atol=1e-12)
            assert [(r["status"], r["accepted"]) for r in got[j][1]] == [(r["status"], r["accepted"]) for r in recs]
            np.testing.assert_allclose([r["loss"] for r in got[j][1]], [r["loss"] for r in recs], rtol=1e-12)


def test_a_surfel_knn_index_out_of_range_is_refused_and_the_slot_stays_usable():
    """The reference indexes the node arrays with sf.knn_indices (super/loss.py:189-197): an index >= J is an
    IndexError there.  Here the preparation reports it from the device (clamped meanwhile: no out-of-bounds access),
    slm_bind_frame fails, the slot counts as unbound, and a good frame binds and solves in it afterwards."""
    import torch
    from super_amd import _lib, synth
    from super_amd.engine import DeviceFrame, Engine
    dev = torch.device("cuda", 0)
    sc = synth.make_scene(seed=5, N=3000, J=64, H=120, W=160)
    good = DeviceFrame.from_scene(sc, dev)
    eng = Engine(dev, max_frames=1, num_iterations=3)
    eng.bind(0, good)
    eng.run(1)
    want = eng.beta(0).cpu().numpy().copy()
    for bad_value in (sc.J, -1, 1 << 20):
        bad = DeviceFrame.from_scene(sc, dev)
        bad.sf_knn_idx[1234, 2] = bad_value
        torch.cuda.synchronize()
        c = bad.c_struct()
        import ctypes as C
        rc = eng.lib.slm_bind_frame(eng.h, 0, C.byref(c), eng.stream)
        assert rc == _lib.SLM_ERR_INVALID and b"sf_knn_idx" in eng.lib.slm_last_error()
        assert eng.lib.slm_run(eng.h, 1, eng.stream) == _lib.SLM_ERR_UNBOUND
        eng.bind(0, good)                                  # hinted or not: the slot works again
        eng.run(1)
        np.testing.assert_allclose(eng.beta(0).cpu().numpy(), want, rtol=0, atol=1e-11)
    eng.close()
