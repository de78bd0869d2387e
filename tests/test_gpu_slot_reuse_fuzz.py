"""Slot reuse under changing problem sizes: a seeded random sequence of frames of very different sizes (surfels, ED
nodes, image, node-KNN width, incl. a frame without surfels and one without any valid target pixel) bound into the
slots of ONE solver -- singly and through slm_bind_frames, on every solver form -- gives for every bind the state a
fresh solver builds for that frame.  Grow-only buffers, the cached symbolic plan, the tuple-count hint and the
descriptor mirrors of a slot all outlive a bind; this is the test that they do so harmlessly.  Needs an MI355X (-m gpu)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [dict(N=300, J=12, H=40, W=56, src_border=4, tgt_border=2),             # fewer nodes than one leaf front
          dict(N=1500, J=96, H=120, W=160),
          dict(N=900, J=24, H=40, W=56, src_border=4, tgt_border=2, n_ed_neighbors=6),
          dict(N=24000, J=256, H=240, W=320),
          dict(N=5000, J=108, H=120, W=160, src_border=6, tgt_border=3, n_ed_neighbors=8),
          dict(N=60000, J=700, H=240, W=320),
          dict(N=2000, J=48, H=60, W=80, src_border=5, tgt_border=3)]


def _scenes():
    from super_amd import synth
    scs = [synth.make_scene(seed=50 + i, **kw) for i, kw in enumerate(SHAPES)]
    empty = synth.make_scene(seed=70, **SHAPES[2])
    for name in ("sf_points", "sf_norms", "sf_knn_idx", "sf_knn_w"):
        setattr(empty, name, getattr(empty, name)[:0].copy())
    blind = synth.make_scene(seed=71, **SHAPES[6])
    blind.valid[:] = False
    blind.index_map[:] = -1
    return scs + [empty, blind]


@pytest.mark.parametrize("solver_path,data_path", [(0, 0), (3, 0), (4, 0), (2, 0), (1, 0), (0, 1), (0, 2)])
def test_random_rebinds_match_fresh_solvers(solver_path, data_path):
    import torch
    from super_amd.engine import DeviceFrame, Engine
    dev = torch.device("cuda", 0)
    frames = [DeviceFrame.from_scene(sc, dev) for sc in _scenes()]
    want = []
    for f in frames:
        e = Engine(dev, max_frames=1, num_iterations=4, solver_path=solver_path, data_path=data_path)
        e.bind(0, f)
        e.run(1)
        want.append((e.beta(0).cpu().numpy().copy(), e.records(0)))
        e.close()
    S = 3
    eng = Engine(dev, max_frames=S, num_iterations=4, solver_path=solver_path, data_path=data_path)
    rng = np.random.default_rng(1234 + solver_path + 10 * data_path)
    cur = [0, 3, 7]
    eng.bind_batch([frames[k] for k in cur])

    def check(tag):
        eng.run(S)
        torch.cuda.synchronize()
        for slot, k in enumerate(cur):
            beta, recs = eng.beta(slot).cpu().numpy(), eng.records(slot)
            key = lambda rs: [(r["status"], r["accepted"], r["M_grad"]) for r in rs]
            assert key(recs) == key(want[k][1]), (tag, slot, k)
            np.testing.assert_allclose(beta, want[k][0], rtol=0, atol=1e-9, err_msg=f"{tag}: slot {slot} frame {k}")

    check("first batch")
    for step in range(14):
        # (every slot is bound again in every step: slm_run on a slot that was not would continue its LM loop)
        cur = [int(k) for k in rng.integers(0, len(frames), size=S)]
        if step % 3 == 2:
            eng.bind_batch([frames[k] for k in cur])
        else:
            for slot in rng.permutation(S):
                eng.bind(int(slot), frames[cur[slot]])
        check(f"step {step}")
    eng.close()


@pytest.mark.parametrize("solver_path", [0, 4])
def test_slots_alternate_between_num_neighbors(solver_path):
    """Round 6: the slots of one solver are re-bound with frames of another ``num_neighbors`` from step to step (the frames of
    ONE step share it) -- 4 takes the tuple-sorted assembly, 3 and 6 the K-generic pair path, all on the multifrontal solver:
    the plan cache, the pair-record buffer, the per-surfel pair indices and the descriptor of a slot survive the switch in
    both directions, and every bind gives the state a fresh solver builds for that frame."""
    import torch
    from super_amd import synth
    from super_amd.engine import DeviceFrame, Engine
    dev = torch.device("cuda", 0)
    shapes = [dict(N=1500, J=96, H=120, W=160), dict(N=6000, J=160, H=120, W=160, src_border=6, tgt_border=3),
              dict(N=900, J=24, H=40, W=56, src_border=4, tgt_border=2, n_ed_neighbors=6)]
    scs = {K: [synth.make_scene(seed=400 + 10 * K + i, n_neighbors=K, **kw) for i, kw in enumerate(shapes)] for K in (3, 4, 6)}
    frames = {K: [DeviceFrame.from_scene(sc, dev) for sc in v] for K, v in scs.items()}
    want = {}
    for K, fl in frames.items():
        for i, f in enumerate(fl):
            e = Engine(dev, max_frames=1, num_iterations=4, solver_path=solver_path)
            e.bind(0, f)
            e.run(1)
            want[K, i] = (e.beta(0).cpu().numpy().copy(), e.records(0))
            e.close()
    S = 3
    eng = Engine(dev, max_frames=S, num_iterations=4, solver_path=solver_path)
    rng = np.random.default_rng(77 + solver_path)
    for step, K in enumerate([4, 6, 4, 3, 6, 6, 4, 3, 4]):
        cur = [int(k) for k in rng.integers(0, len(shapes), size=S)]
        if step % 2:
            eng.bind_batch([frames[K][k] for k in cur])
        else:
            for slot in rng.permutation(S):
                eng.bind(int(slot), frames[K][cur[slot]])
        eng.run(S)
        torch.cuda.synchronize()
        for slot, k in enumerate(cur):
            beta, recs = eng.beta(slot).cpu().numpy(), eng.records(slot)
            key = lambda rs: [(r["status"], r["accepted"], r["M_grad"]) for r in rs]
            assert key(recs) == key(want[K, k][1]), (step, K, slot, k)
            np.testing.assert_allclose(beta, want[K, k][0], rtol=0, atol=1e-9, err_msg=f"step {step} K {K} slot {slot} frame {k}")
    eng.close()


def test_rebinding_does_not_leak_device_memory():
    """The buffers of a slot only grow: alternating two frames for a while leaves the free device memory where it
    was once both have been seen."""
    import torch
    from super_amd.engine import DeviceFrame, Engine
    dev = torch.device("cuda", 0)
    scs = _scenes()
    a, b = DeviceFrame.from_scene(scs[3], dev), DeviceFrame.from_scene(scs[1], dev)
    eng = Engine(dev, max_frames=2, num_iterations=2)
    free = []
    for it in range(40):
        eng.bind(0, a if it % 2 else b)
        eng.bind_batch([b, a] if it % 3 else [a, b])
        eng.run(2)
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info(dev)[0])
    eng.close()
    assert free[8] - free[-1] <= 4 << 20, (free[8], free[-1])


@pytest.mark.parametrize("optimizer", ["SGD", "Adam"])
def test_one_graphfit_object_over_frames_of_changing_size_matches_fresh_objects(optimizer):
    """The autograd-path mirror keeps ONE solver handle for the whole sequence (as the reference keeps one GraphFit
    module, super/deform_mesh.py:30-44): frames of very different sizes through the same object give what a fresh
    object gives for each."""
    import torch
    from helpers import torch_frame
    from oracle import graphfit_oracle as gfo
    from super_amd.deform_mesh import GraphFit
    from super_amd import synth

    def opt():
        o = gfo.default_opt(optimizer=optimizer)
        o.deform_udpate_method = "super_edg"
        return o

    shapes = [SHAPES[1], SHAPES[3], SHAPES[0], SHAPES[5], SHAPES[6], SHAPES[3]]
    scs = [synth.make_scene(seed=80 + i, **kw) for i, kw in enumerate(shapes)]
    one = GraphFit(opt())
    for i, sc in enumerate(scs):
        sf, inputs, new_data = torch_frame(sc)
        got = one(inputs, sf, new_data, None).cpu().numpy()
        sf, inputs, new_data = torch_frame(sc)
        want = GraphFit(opt())(inputs, sf, new_data, None).cpu().numpy()
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-10, err_msg=f"frame {i}")
        assert np.abs(want[:, 1:]).max() > 1e-6            # the fit moved something


def test_surfel_sharded_ranks_rebound_with_frames_of_changing_size():
    """The surfel-sharded protocol (super/LM.py:93-133 split over ranks; here two solver contexts on this GPU with the
    exchanges emulated) over a sequence of frames of very different sizes in the same slot: every frame ends at the
    single-GPU solve."""
    import torch
    from helpers import ref_opt, torch_frame
    from oracle import lm_oracle as orc
    from super_amd import _lib, synth
    from super_amd.LM import LM_Solver, _dev_ptr, _stream_ptr
    world = 2
    o = ref_opt(orc.default_opt())
    o.num_optimize_iterations = 4
    ranks = [LM_Solver(o, rank=r, world=world, all_reduce=lambda t: None, broadcast=lambda t: None) for r in range(world)]
    hs = [lm._handle() for lm in ranks]
    lib = ranks[0].lib
    shapes = [SHAPES[1], SHAPES[3], SHAPES[0], SHAPES[5], SHAPES[1]]
    for i, kw in enumerate(shapes):
        sc = synth.make_scene(seed=90 + i, **kw)
        sf, inputs, new_data = torch_frame(sc)
        bfs = [lm._bind(h, 0, sf, inputs, new_data) for lm, h in zip(ranks, hs)]
        dev = bfs[0].device
        st = _stream_ptr(dev)

        def exchange(what, combine):
            bufs = [lm.exchange_buffer(h, 0, what, dev) for lm, h in zip(ranks, hs)]
            for h, b in zip(hs, bufs):
                _lib.check(lib.slm_lm_exchange_get(h, 0, what, _dev_ptr(b), st), "get")
            out = combine(bufs)
            for h in hs:
                _lib.check(lib.slm_lm_exchange_set(h, 0, what, _dev_ptr(out), st), "set")

        for _ in range(int(o.num_optimize_iterations)):
            for h in hs:
                _lib.check(lib.slm_lm_grad_local(h, 1, st), "grad_local")
            exchange(_lib.SLM_X_PAIR_BLOCKS, lambda b: torch.stack(b).sum(0))
            for h in hs:
                _lib.check(lib.slm_lm_solve(h, 1, st), "solve")
            exchange(_lib.SLM_X_DELTA, lambda b: b[0].clone())
            for h in hs:
                _lib.check(lib.slm_lm_loss_local(h, 1, st), "loss_local")
            exchange(_lib.SLM_X_DATA_LOSS, lambda b: torch.stack(b).sum(0))
            for h in hs:
                _lib.check(lib.slm_lm_accept(h, 1, st), "accept")
        betas = []
        for h, bf in zip(hs, bfs):
            beta = torch.empty((bf.J, 7), dtype=torch.float64, device=dev)
            _lib.check(lib.slm_get_beta(h, 0, _dev_ptr(beta), st), "get_beta")
            betas.append(beta.cpu().numpy())
        np.testing.assert_array_equal(betas[0], betas[1])
        single = LM_Solver(o)
        want = single.LM(sf, inputs, new_data).cpu().numpy()
        np.testing.assert_allclose(betas[0], want, rtol=0, atol=1e-9, err_msg=f"frame {i}")
