"""The nested-dissection factorisation as ONE persistent launch over a static task graph (solver_path 2,
csrc/slm_dag.hip) against the reference's goldens, against the per-level launch form (solver_path 3), for
batches of frames with different plans, for reproducibility and for the failure path.  Through the C ABI."""
import numpy as np
import pytest

from helpers import GOLDENS, load_golden, ref_opt, torch_frame
from oracle import lm_oracle as orc

pytestmark = pytest.mark.gpu


def _engine(**kw):
    import torch
    from super_amd.engine import Engine
    return Engine(torch.device("cuda", 0), **kw)


def _dframe(sc, **kw):
    import torch
    from super_amd.engine import DeviceFrame
    return DeviceFrame.from_scene(sc, torch.device("cuda", 0), **kw)


@pytest.mark.parametrize("name", GOLDENS)
def test_lm_on_the_task_graph_solver_matches_reference_goldens(name):
    from super_amd.LM import LM_Solver
    g, sc, opt = load_golden(name)
    o = ref_opt(opt)
    o.slm_solver_path = 2
    lm = LM_Solver(o)
    beta = lm.LM(*torch_frame(sc)).cpu().numpy()
    recs = lm.last_records[0]
    assert all(r["status"] == 0 for r in recs)
    np.testing.assert_allclose([r["loss"] for r in recs], g["lm_loss"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(beta, g["lm_beta"], rtol=0, atol=1e-4)
    assert np.abs(beta - g["lm_beta"]).max() < 1e-7      # observed ~1e-12


def test_task_graph_equals_level_launches_on_a_batch_of_different_plans():
    """three frames of different sizes advance together; per frame the two forms of the numeric phase give
    the same iterations (loss records to 1e-12 relative, beta to 1e-12)"""
    from super_amd import synth
    scenes = [synth.make_scene(N=4000, J=96, H=96, W=128, seed=61, src_border=6, tgt_border=3),
              synth.make_scene(N=2500, J=48, H=60, W=80, seed=62, src_border=5, tgt_border=3),
              synth.make_scene(N=6000, J=140, H=120, W=160, seed=63, src_border=6, tgt_border=4, dphi=0.4)]
    out = {}
    for sp in (3, 2):
        e = _engine(max_frames=3, solver_path=sp)
        for i, sc in enumerate(scenes):
            e.bind(i, _dframe(sc))
        e.run(3)
        out[sp] = [(e.beta(i).cpu().numpy(), e.records(i)) for i in range(3)]
    for i in range(3):
        b0, r0 = out[3][i]
        b2, r2 = out[2][i]
        assert all(r["status"] == 0 for r in r2)
        np.testing.assert_allclose([r["loss"] for r in r2], [r["loss"] for r in r0], rtol=1e-10)
        assert [r["accepted"] for r in r2] == [r["accepted"] for r in r0]
        np.testing.assert_allclose(b2, b0, rtol=0, atol=1e-10)
        ob = orc.lm(orc.Frame.from_scene(scenes[i]), orc.default_opt())
        np.testing.assert_allclose(b2, ob, rtol=0, atol=1e-6)


def test_task_graph_solver_is_bitwise_reproducible():
    """fixed summation order everywhere (extend-adds child 0 before child 1): with the reproducible data path
    two runs give identical bits, whatever order the workgroups ran the tasks in"""
    from super_amd import synth
    sc = synth.make_scene(N=20000, J=400, H=240, W=320, seed=64, src_border=8, tgt_border=4)
    betas = []
    for _ in range(3):
        e = _engine(solver_path=2, data_path=2)
        e.bind(0, _dframe(sc))
        e.run(1)
        assert all(r["status"] == 0 for r in e.records(0))
        betas.append(e.beta(0).cpu().numpy())
    assert (betas[0] == betas[1]).all() and (betas[0] == betas[2]).all()


def test_task_graph_solver_failure_stops_like_the_reference():
    """singular normal matrix (data term only, u0 = 0, nodes without surfels): the factorisation reports the
    non-positive pivot, the loop stops with beta unchanged (super/LM.py:99-103) -- and the launch terminates"""
    from super_amd import synth
    sc = synth.make_scene(N=300, J=96, H=60, W=80, seed=33, src_border=20, tgt_border=3)
    e = _engine(solver_path=2, use_arap=False, use_rot=False, u0=0.0)
    e.bind(0, _dframe(sc))
    e.run(1)
    recs = e.records(0)
    assert recs[0]["status"] == 1 and all(r["status"] == 2 for r in recs[1:])
    np.testing.assert_allclose(e.beta(0).cpu().numpy(), np.tile([1.0, 0, 0, 0, 0, 0, 0], (sc.J, 1)), rtol=0, atol=0)


def test_c2_full_size_task_graph_solver():
    """BASELINE C2 (200 k surfels / 2 000 nodes): 10 iterations on the task graph == per-level launches"""
    from super_amd import synth
    sc = synth.make_scene(seed=0, **synth.WORKLOADS["C2"])
    res = {}
    for sp in (3, 2):
        e = _engine(solver_path=sp)
        e.bind(0, _dframe(sc, state_f64=(sp == 2)))
        e.run(1)
        res[sp] = (e.beta(0).cpu().numpy(), e.records(0))
    assert all(r["status"] == 0 for r in res[2][1])
    np.testing.assert_allclose([r["loss"] for r in res[2][1]], [r["loss"] for r in res[3][1]], rtol=1e-9)
    np.testing.assert_allclose(res[2][0], res[3][0], rtol=0, atol=1e-9)


def test_hybrid_form_on_a_batch_equals_level_launches():
    """batches whose trees have the same depth: per-level launches for the levels with many fronts, the top of the tree
    as tasks of one persistent launch over all frames (solver_path 4, and what solver_path 0 picks for LARGE batches;
    this one -- 4 frames x 400 nodes -- it runs as one task graph: frames x nodes <= SLM_DAG_MAX_NODES) -- same
    iterations as the pure per-level form, different plans per frame"""
    from super_amd import synth
    scenes = [synth.make_scene(N=20000, J=400, H=240, W=320, seed=70 + k, src_border=8, tgt_border=4, dphi=0.15 + 0.05 * k)
              for k in range(4)]
    out = {}
    for sp in (3, 4, 0):
        e = _engine(max_frames=4, solver_path=sp)
        e.bind_batch([_dframe(sc) for sc in scenes])
        e.run(4)
        form = e.lib.slm_debug_last_solver_form(e.h)
        assert form == {3: 0, 4: 2, 0: 1}[sp], (sp, form)
        out[sp] = [(e.beta(i).cpu().numpy(), e.records(i)) for i in range(4)]
    for sp in (4, 0):
        for i in range(4):
            b0, r0 = out[3][i]
            b, r = out[sp][i]
            assert all(x["status"] == 0 for x in r)
            np.testing.assert_allclose([x["loss"] for x in r], [x["loss"] for x in r0], rtol=1e-10)
            assert [x["accepted"] for x in r] == [x["accepted"] for x in r0]
            np.testing.assert_allclose(b, b0, rtol=0, atol=1e-10)
    ob = orc.lm(orc.Frame.from_scene(scenes[1]), orc.default_opt())
    np.testing.assert_allclose(out[0][1][0], ob, rtol=0, atol=1e-6)


def test_hybrid_form_falls_back_when_the_trees_differ_in_depth():
    from super_amd import synth
    scenes = [synth.make_scene(N=4000, J=96, H=96, W=128, seed=61, src_border=6, tgt_border=3),
              synth.make_scene(N=2500, J=48, H=60, W=80, seed=62, src_border=5, tgt_border=3),
              synth.make_scene(N=20000, J=400, H=240, W=320, seed=63, src_border=8, tgt_border=4)]
    e = _engine(max_frames=3, solver_path=4)
    for i, sc in enumerate(scenes):
        e.bind(i, _dframe(sc))
    e.run(3)
    assert e.lib.slm_debug_last_solver_form(e.h) == 0
    for i, sc in enumerate(scenes[:2]):
        np.testing.assert_allclose(e.beta(i).cpu().numpy(), orc.lm(orc.Frame.from_scene(sc), orc.default_opt()), rtol=0, atol=1e-6)


def test_hybrid_form_failure_stops_like_the_reference():
    """a singular frame inside a hybrid batch: its loop stops with beta unchanged, the other frames finish"""
    from super_amd import synth
    good = [synth.make_scene(N=2500, J=96, H=96, W=128, seed=80 + k, src_border=6, tgt_border=3) for k in range(2)]
    bad = synth.make_scene(N=300, J=96, H=96, W=128, seed=33, src_border=30, tgt_border=3)
    e = _engine(max_frames=3, solver_path=4, use_arap=False, use_rot=False, u0=0.0)
    e.bind_batch([_dframe(good[0]), _dframe(bad), _dframe(good[1])])
    e.run(3)
    recs = e.records(1)
    assert recs[0]["status"] == 1 and all(r["status"] == 2 for r in recs[1:])
    np.testing.assert_allclose(e.beta(1).cpu().numpy(), np.tile([1.0, 0, 0, 0, 0, 0, 0], (bad.J, 1)), rtol=0, atol=0)


def test_c2_full_size_hybrid_batch():
    """BASELINE C2 x 3 frames (different plans): the hybrid form (solver_path 4) and what solver_path 0 picks for this
    batch -- one task graph over the three frames (3 x 2 000 nodes <= SLM_DAG_MAX_NODES; two workgroups per CU since
    round 5) -- == per-level launches"""
    from super_amd import synth
    scenes = [synth.make_scene(seed=s, **synth.WORKLOADS["C2"]) for s in range(3)]
    res = {}
    for sp in (3, 4, 0):
        e = _engine(max_frames=3, solver_path=sp)
        e.bind_batch([_dframe(sc) for sc in scenes])
        e.run(3)
        assert e.lib.slm_debug_last_solver_form(e.h) == {3: 0, 4: 2, 0: 1}[sp]
        res[sp] = [(e.beta(i).cpu().numpy(), e.records(i)) for i in range(3)]
        e.close()
    for sp in (4, 0):
        for i in range(3):
            assert all(r["status"] == 0 for r in res[sp][i][1])
            np.testing.assert_allclose([r["loss"] for r in res[sp][i][1]], [r["loss"] for r in res[3][i][1]], rtol=1e-9)
            assert [r["accepted"] for r in res[sp][i][1]] == [r["accepted"] for r in res[3][i][1]]
            np.testing.assert_allclose(res[sp][i][0], res[3][i][0], rtol=0, atol=1e-9)


@pytest.mark.parametrize("top_fronts", ["1", "4", "16"])
def test_hybrid_cut_override_keeps_the_result(top_fronts):
    """SLM_DAG_TOP_FRONTS moves the cut between the per-level launches and the task graph (root only ... four levels):
    same solution (read once per process, hence the subprocess)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        f"sys.path[:0] = [{root!r}, {os.path.join(root, 'python-super_amd')!r}, {os.path.join(root, 'tests')!r}]\n"
        "import torch\n"
        "from oracle import lm_oracle as orc\n"
        "from super_amd import synth\n"
        "from super_amd.engine import DeviceFrame, Engine\n"
        "dev = torch.device('cuda', 0)\n"
        "scs = [synth.make_scene(N=20000, J=400, H=240, W=320, seed=90 + k, src_border=8, tgt_border=4) for k in range(3)]\n"
        "e = Engine(dev, max_frames=3, solver_path=4, num_iterations=4)\n"
        "e.bind_batch([DeviceFrame.from_scene(sc, dev) for sc in scs])\n"
        "e.run(3)\n"
        "form = e.lib.slm_debug_last_solver_form(e.h)\n"
        "errs = [float(np.abs(e.beta(i).cpu().numpy() - orc.lm(orc.Frame.from_scene(scs[i]), orc.default_opt(num_optimize_iterations=4))).max()) for i in range(3)]\n"
        "print('RES', form, max(errs))\n")
    env = dict(os.environ)
    env["SLM_DAG_TOP_FRONTS"] = top_fronts
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    form, err = out.stdout.strip().splitlines()[-1].split()[1:]
    assert int(form) == 2 and float(err) < 1e-6, (top_fronts, form, err)


def test_c4_hybrid_batch_matches_level_launches():
    """500 k surfels / 4 000 nodes (9 levels) x 3 frames: hybrid form == per-level launches, 4 iterations"""
    from super_amd import synth
    scenes = [synth.make_scene(seed=s, **synth.WORKLOADS["C4"]) for s in range(3)]
    frames = [_dframe(sc) for sc in scenes]
    res = {}
    for sp in (3, 0):
        e = _engine(max_frames=3, solver_path=sp, num_iterations=4)
        e.bind_batch(frames)
        e.run(3)
        assert e.lib.slm_debug_last_solver_form(e.h) == (0 if sp == 3 else 2)
        res[sp] = [(e.beta(i).cpu().numpy(), e.records(i)) for i in range(3)]
        e.close()
    for i in range(3):
        assert all(r["status"] == 0 for r in res[0][i][1])
        np.testing.assert_allclose([r["loss"] for r in res[0][i][1]], [r["loss"] for r in res[3][i][1]], rtol=1e-9)
        np.testing.assert_allclose(res[0][i][0], res[3][i][0], rtol=0, atol=1e-9)


def test_an_aborted_task_graph_stops_every_unfinished_slot_of_a_large_batch():
    """after an aborted task-graph launch EVERY slot whose solve did not finish must stop -- also slots
    beyond the first 64 of a batch -- and with a status of its own (a scheduling time-out, not an ill-posed system).
    `slm_debug_dag_abort` leaves the slots as an aborted launch does and runs the launch's own check + the accept step."""
    import torch
    from super_amd import _lib, synth
    from super_amd.engine import DeviceFrame, Engine
    B = 72
    dev = torch.device("cuda", 0)
    sc = synth.make_scene(N=1500, J=48, H=60, W=80, seed=5, src_border=5, tgt_border=3)
    frames = [DeviceFrame.from_scene(sc, dev) for _ in range(B)]
    eng = Engine(dev, max_frames=B, solver_path=2, num_iterations=3)
    eng.bind_batch(frames)
    _lib.check(eng.lib.slm_debug_dag_abort(eng.h, B, eng.stream), "abort")
    eng.run(B)                                           # stopped slots: the remaining iterations are no-ops
    recs = [eng.records(i) for i in range(B)]
    assert [r[0]["status"] for r in recs] == [_lib.SLM_ITER_SOLVER_TIMEOUT] * B
    assert all(x["status"] == _lib.SLM_ITER_NOT_RUN for r in recs for x in r[1:])
    beta = eng.beta(B - 1).cpu().numpy()
    np.testing.assert_array_equal(beta, np.tile([1.0, 0, 0, 0, 0, 0, 0], (sc.J, 1)))      # nothing was accepted
    # the mirror says what happened
    assert _lib.SLM_ITER_SOLVER_TIMEOUT != _lib.SLM_ITER_SOLVER_FAILED
    # a fresh bind of the same batch solves (and a shortened, then restored deadline is accepted by the API)
    _lib.check(eng.lib.slm_debug_dag_timeout(10 ** 9), "timeout")
    _lib.check(eng.lib.slm_debug_dag_timeout(0), "timeout")
    eng.bind_batch(frames)
    eng.run(B)
    assert all(r["status"] == _lib.SLM_ITER_OK for i in range(B) for r in eng.records(i))
    eng.close()


def test_leaf_size_follows_the_solver_form_and_changes_nothing_but_the_plan():
    """Round 5: a solver that runs its launches as ONE task graph (at most two slots, or solver_path 2) dissects down to
    50-node leaves (SLM_ND_LEAF_LATENCY), the per-level / hybrid forms to 18 (SLM_ND_LEAF): fewer, larger fronts -- and the
    same iterations (the elimination order is not part of the result: 1e-9 on beta, accept sequence equal), both against the
    oracle.  The look-ahead entry of the tile factorisation (POTRF(s > 0)) runs in every task-graph solve of this file."""
    from super_amd import synth
    sc = synth.make_scene(N=20000, J=400, H=240, W=320, seed=77, src_border=8, tgt_border=4, dphi=0.2)
    out = {}
    for name, kw in (("graph1", dict(max_frames=1)), ("levels8", dict(max_frames=8, solver_path=3)), ("graph8", dict(max_frames=8, solver_path=2))):
        e = _engine(**kw)
        e.bind(0, _dframe(sc))
        e.run(1)
        out[name] = (e.beta(0).cpu().numpy(), e.records(0), e.plan_info(0))
        e.close()
    assert out["graph1"][2]["fronts"] < out["levels8"][2]["fronts"]          # larger leaves: fewer fronts, fewer levels
    assert out["graph1"][2]["levels"] <= out["levels8"][2]["levels"]
    assert out["graph8"][2]["fronts"] == out["graph1"][2]["fronts"]          # solver_path 2 is a task graph at any slot count
    ob = orc.lm(orc.Frame.from_scene(sc), orc.default_opt())
    for name in out:
        b, r, _ = out[name]
        assert all(x["status"] == 0 for x in r)
        assert [x["accepted"] for x in r] == [x["accepted"] for x in out["levels8"][1]]
        np.testing.assert_allclose(b, out["levels8"][0], rtol=0, atol=1e-9)
        np.testing.assert_allclose(b, ob, rtol=0, atol=1e-6)


def _expect_affine():
    """k_fdag's XCD-affine mode is taken on a device with 8 XCDs (MI355X: 256 CUs in 8 XCDs) unless SLM_DAG_XCD=0."""
    import os
    import torch
    if os.environ.get("SLM_DAG_XCD") == "0":
        return 0
    return 1 if torch.cuda.get_device_properties(0).multi_processor_count == 256 else 0


def test_eight_frames_as_one_task_graph_on_xcd_affine_ticket_streams():
    """Round 5: at a multiple of 8 frames per launch every XCD serves the tasks of 'its' frames from a ticket stream of its
    own (k_fdag mode bit 0): all tasks of a frame on one XCD, operand tiles and gathered update tiles through that XCD's L2.
    8 frames x 400 nodes run as ONE task graph under solver_path 0 (frames x nodes <= SLM_DAG_MAX_NODES); same iterations
    as the per-level launches, different plans per frame."""
    from super_amd import synth
    scenes = [synth.make_scene(N=20000, J=400, H=240, W=320, seed=120 + k, src_border=8, tgt_border=4, dphi=0.1 + 0.03 * k)
              for k in range(8)]
    frames = [_dframe(sc) for sc in scenes]
    out = {}
    for sp in (3, 0, 4):
        e = _engine(max_frames=8, solver_path=sp)
        e.bind_batch(frames)
        e.run(8)
        assert e.lib.slm_debug_last_solver_form(e.h) == {3: 0, 0: 1, 4: 2}[sp]
        # the mode bit must really have been taken (a probe that says "not 8 XCDs" would silently test the
        # non-affine path): 1 on an 8-XCD device, -1 when the solve ran no task graph
        assert e.lib.slm_debug_last_dag_mode(e.h) == {3: -1, 0: _expect_affine(), 4: _expect_affine()}[sp]
        out[sp] = [(e.beta(i).cpu().numpy(), e.records(i)) for i in range(8)]
        e.close()
    for sp in (0, 4):
        for i in range(8):
            b0, r0 = out[3][i]
            b, r = out[sp][i]
            assert all(x["status"] == 0 for x in r)
            np.testing.assert_allclose([x["loss"] for x in r], [x["loss"] for x in r0], rtol=1e-10)
            assert [x["accepted"] for x in r] == [x["accepted"] for x in r0]
            np.testing.assert_allclose(b, b0, rtol=0, atol=1e-10)


@pytest.mark.parametrize("env", [{"SLM_DAG_XCD": "0"}, {"SLM_DAG_WG_PER_CU": "1"}, {"SLM_DAG_DEFER_BOUNDARY": "0"},
                                 {"SLM_DAG_DEFER_BOUNDARY": "9"}, {"SLM_DAG_MAX_NODES": "0"}, {"SLM_FUSE_BEGIN": "0"},
                                 {"SLM_DAG_XCD": "0", "SLM_DAG_WG_PER_CU": "1", "SLM_DAG_DEFER_BOUNDARY": "0", "SLM_DAG_TOP_FRONTS": "2"}])
def test_task_graph_switches_of_round_5_keep_the_result(env):
    """The round-5 switches of the task graph (read once per process, hence the subprocess): no XCD-affine ticket streams,
    one workgroup per CU, the boundary rows of a column listed with / far behind its pivot rows, no frames x nodes rule (the
    hybrid form for a small batch), the round-4 settings together, and round 6's zeroing as a launch of its own again
    (SLM_FUSE_BEGIN=0) -- 8 frames of different plans under solver_path 0 and 4, four iterations, against the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        f"sys.path[:0] = [{root!r}, {os.path.join(root, 'python-super_amd')!r}, {os.path.join(root, 'tests')!r}]\n"
        "import torch\n"
        "from oracle import lm_oracle as orc\n"
        "from super_amd import synth\n"
        "from super_amd.engine import DeviceFrame, Engine\n"
        "dev = torch.device('cuda', 0)\n"
        "scs = [synth.make_scene(N=20000, J=400, H=240, W=320, seed=140 + k, src_border=8, tgt_border=4, dphi=0.1 + 0.02 * k) for k in range(8)]\n"
        "want = [orc.lm(orc.Frame.from_scene(sc), orc.default_opt(num_optimize_iterations=4)) for sc in scs[:2]]\n"
        "res = []\n"
        "for sp in (0, 4):\n"
        "    e = Engine(dev, max_frames=8, solver_path=sp, num_iterations=4)\n"
        "    e.bind_batch([DeviceFrame.from_scene(sc, dev) for sc in scs])\n"
        "    e.run(8)\n"
        "    ok = all(r['status'] == 0 for i in range(8) for r in e.records(i))\n"
        "    err = max(float(np.abs(e.beta(i).cpu().numpy() - want[i]).max()) for i in range(2))\n"
        "    res.append((e.lib.slm_debug_last_solver_form(e.h), ok, err, e.lib.slm_debug_last_dag_mode(e.h)))\n"
        "    e.close()\n"
        "print('RES', res[0][0], res[1][0], int(res[0][1] and res[1][1]), max(res[0][2], res[1][2]), res[0][3], res[1][3])\n")
    full = dict(os.environ)
    full.update(env)
    out = subprocess.run([sys.executable, "-c", code], env=full, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    f0, f4, ok, err, m0, m4 = out.stdout.strip().splitlines()[-1].split()[1:]
    assert int(f0) == (2 if env.get("SLM_DAG_MAX_NODES") == "0" else 1) and int(f4) == 2, (env, f0, f4)
    if env.get("SLM_DAG_XCD") == "0":
        assert int(m0) == 0 and int(m4) == 0, (env, m0, m4)      # the switch really turns the XCD-affine streams off
    else:
        assert int(m0) == _expect_affine() and int(m4) == _expect_affine(), (env, m0, m4)
    assert int(ok) == 1 and float(err) < 1e-6, (env, ok, err)
