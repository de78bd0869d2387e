"""Pure-fill pivot-column tiles (round 5; ``FrameDev::tile_kind``): a tile of a front's pivot columns that no
assembled block reaches (data term, ARAP, Rot, node diagonals) and that some child maps into is neither zeroed per iteration nor
read by its first toucher -- the pull (``k_fpull``, or the tile's own task of the task graph) starts from zero and stores it.
Differential check through the C ABI on an MI355X: with ``SLM_PURE_FILL=0`` (read per solver at ``slm_create``) every tile is
zeroed and read-modify-written as in rounds 1-4; on the run-to-run reproducible data path the two must agree BITWISE -- same
sums in the same order, a skipped ``0 +`` changes nothing --, in every form of the solver (task graph, per-level launches,
hybrid), over all ten LM iterations, and across a sequence of frames on one solver (the kinds follow the plan's destination
list, which grows by fill-position pairs while the node graph stays).  Reference being replaced: the dense Cholesky of
``super/LM.py:47-49`` -- the oracle comparison of the same runs lives in ``tests/test_gpu_fullsize_parity.py``."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(monkeypatch, pure, **kw):
    import torch
    from super_amd.engine import Engine
    if pure:
        monkeypatch.delenv("SLM_PURE_FILL", raising=False)
    else:
        monkeypatch.setenv("SLM_PURE_FILL", "0")
    e = Engine(torch.device("cuda", 0), data_path=2, **kw)
    monkeypatch.delenv("SLM_PURE_FILL", raising=False)
    return e


@pytest.mark.parametrize("form", [(1, 0), (3, 0), (2, 3), (1, 3)])   # (frames per launch, solver_path): task graph, hybrid, per-level
def test_pure_fill_tiles_change_nothing_bitwise(monkeypatch, form):
    import torch
    from super_amd import synth
    from super_amd.engine import DeviceFrame
    B, sp = form
    dev = torch.device("cuda", 0)
    frames = [DeviceFrame.from_scene(synth.make_scene(seed=s, **synth.WORKLOADS["C1"]), dev) for s in range(B)]
    out = {}
    for pure in (True, False):
        e = _engine(monkeypatch, pure, max_frames=B, solver_path=sp)
        for i, fr in enumerate(frames):
            e.bind(i, fr)
        e.run(B)
        info = e.plan_info(0)
        out[pure] = ([e.beta(i).cpu().numpy().copy() for i in range(B)], [r["loss"] for r in e.records(0)], info)
        assert all(r["status"] == 0 for r in e.records(0))
        e.close()
    on, off = out[True][2], out[False][2]
    assert on["pivot_tiles"] == off["pivot_tiles"] > 0
    assert off["pure_fill_tiles"] == 0
    # (C1: a third of the tiles with the throughput plan, C2: 41 %; fewer with the task-graph solver's larger leaves -- a leaf has no children)
    assert 0.05 * on["pivot_tiles"] < on["pure_fill_tiles"] < 0.7 * on["pivot_tiles"]
    for a, b in zip(out[True][0], out[False][0]):
        np.testing.assert_array_equal(a, b)
    assert out[True][1] == out[False][1]


def test_kinds_follow_the_plan_over_a_sequence_of_frames(monkeypatch):
    """One solver, frames whose coupled-pair lists differ on the same node graph (surfels moved between neighbouring tuples): the
    plan is reused, extended by fill-position pairs or rebuilt -- every frame must equal what the all-zeroed form gives."""
    import torch
    from super_amd import synth
    from super_amd.engine import DeviceFrame
    dev = torch.device("cuda", 0)
    base = dict(N=20_000, J=300, H=240, W=320)
    scenes = [synth.make_scene(seed=3, **base), synth.make_scene(seed=3, **{**base, "N": 14_000}), synth.make_scene(seed=3, **{**base, "N": 26_000}),
              synth.make_scene(seed=3, **base)]
    runs = {}
    for pure in (True, False):
        e = _engine(monkeypatch, pure, max_frames=1)
        betas = []
        for sc in scenes:
            e.bind(0, DeviceFrame.from_scene(sc, dev))
            e.run(1)
            betas.append(e.beta(0).cpu().numpy().copy())
        runs[pure] = betas
        e.close()
    for a, b in zip(runs[True], runs[False]):
        np.testing.assert_array_equal(a, b)
