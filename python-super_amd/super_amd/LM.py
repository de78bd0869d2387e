"""Host-side mirror of the reference's ``LM_Solver`` (``super/LM.py:10-122``) over
libsuper_lm.so.

Same names, argument meaning and error behaviour as the reference class, so that
``SuPer.__init__`` / ``SuPer.fusion`` (``super/super.py:18-19,68``) can construct and
call it unchanged when ``--use_derived_gradient`` is set:

    self.lm = LM_Solver(self.opt)
    deform_param = self.lm.LM(self.sf, inputs, sfdata)        # (J,7) float64

Everything numerical happens in the HIP library; this file only hands the caller's tensors to
the C ABI (device pointers) and converts results back.  The model state (``sf.points``,
``sf.knn_w``, ``ED_nodes.points``) is float64 in the reference and is passed AS IS (the library
reads float64 state, ``slm_frame.state_f64``): nothing is rounded between frames.  float32
state tensors are passed as float32 (the compact layout); indices become int32, the per-frame
target tables float32 (exact: the reference widens float32 back-projections,
``utils/data_loader.py:453-462``).  PyTorch is plumbing (device memory, streams).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import SlmConfig, SlmFrame, SlmIterRecord


def _dev_ptr(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise _lib.SuperLMError("super_amd needs tensors on a HIP device (no CPU fallback)")
    return t.data_ptr()


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _as(t, dtype, device):
    return t.detach().to(device=device, dtype=dtype).contiguous()


def state_dtype(sf, override=None):
    """dtype the library reads the model state in: the dtype of ``sf.points`` (float64 in the
    reference) unless ``override`` ("f32" / "f64", ``opt.slm_state_dtype``) forces one."""
    if override in ("f32", "float32", torch.float32):
        return torch.float32
    if override in ("f64", "float64", torch.float64):
        return torch.float64
    return torch.float64 if sf.points.dtype == torch.float64 else torch.float32


class ModelView:
    """The MODEL side of a frame in ABI layout (what ``slm_prepare_model`` reads): ``sf.points``, ``sf.knn_indices``,
    ``sf.knn_w``, ``ED_nodes.points``, ``ED_nodes.knn_indices``.  Remembers which tensors (and which in-place versions
    of them) it was made from, so that the bind of the next frame can tell whether it is still the same model."""

    FIELDS = ("sf_points", "sf_knn_idx", "sf_knn_w", "ed_points", "ed_knn_idx")

    def __init__(self, sf, device=None, state=None):
        dev = device if device is not None else sf.points.device
        if dev.type != "cuda":
            raise _lib.SuperLMError("super_amd needs tensors on a HIP device (no CPU fallback)")
        ed = sf.ED_nodes
        i32 = torch.int32
        sdt = state_dtype(sf, state)
        self.device = dev
        self.state_dtype = sdt
        self.sources = (sf.points, sf.knn_indices, sf.knn_w, ed.points, ed.knn_indices)
        self.stamp = self._stamp(self.sources)
        self.sf_points = _as(sf.points, sdt, dev)          # no copy when already float64 / contiguous
        self.sf_knn_idx = _as(sf.knn_indices, i32, dev)
        self.sf_knn_w = _as(sf.knn_w, sdt, dev)
        self.ed_points = _as(ed.points, sdt, dev)
        self.ed_knn_idx = _as(ed.knn_indices, i32, dev)
        self.J = int(self.ed_points.shape[0])

    @staticmethod
    def _stamp(tensors):
        return tuple((t.data_ptr(), tuple(t.shape), t.dtype, t._version) for t in tensors)

    def matches(self, sf, device, state):
        ed = sf.ED_nodes
        return (self.device == device and self.state_dtype == state_dtype(sf, state) and
                self.stamp == self._stamp((sf.points, sf.knn_indices, sf.knn_w, ed.points, ed.knn_indices)))

    def fill(self, fr: SlmFrame):
        fr.N, fr.J = int(self.sf_points.shape[0]), self.J
        fr.K, fr.K_ED = int(self.sf_knn_idx.shape[1]), int(self.ed_knn_idx.shape[1])
        for name in self.FIELDS:
            setattr(fr, name, _dev_ptr(getattr(self, name)))
        fr.state_f64 = 1 if self.state_dtype == torch.float64 else 0


class BoundFrame:
    """Device-resident, ABI-layout view of what LM reads from ``sf`` / ``inputs`` /
    ``new_data``.  Holds the tensors alive while the library uses them.
    ``model``: a ``ModelView`` of the same ``sf`` made earlier (``LM_Solver.prepare_model``)."""

    def __init__(self, sf, inputs, new_data, device=None, state=None, model=None):
        dev = device if device is not None else sf.points.device
        if dev.type != "cuda":
            raise _lib.SuperLMError("super_amd needs tensors on a HIP device (no CPU fallback)")
        f32, i32 = torch.float32, torch.int32
        mv = model if model is not None else ModelView(sf, dev, state)
        self.model = mv
        self.device = dev
        self.state_dtype = mv.state_dtype
        for name in ModelView.FIELDS:
            setattr(self, name, getattr(mv, name))
        self.tgt_points = _as(new_data.points, f32, dev)
        self.tgt_norms = _as(new_data.norms, f32, dev)
        self.index_map = _as(new_data.index_map, i32, dev)
        self.tgt_valid = _as(new_data.valid, torch.uint8, dev)
        H, W = inputs[("color", 0)].shape[-2:]
        K = inputs["K"]
        Kh = K[0].detach().to("cpu", torch.float32)        # one tiny D2H read per frame
        self.J = mv.J
        fr = SlmFrame()
        mv.fill(fr)
        fr.T = int(self.tgt_points.shape[0])
        fr.H, fr.W = int(H), int(W)
        fr.fx, fr.fy, fr.cx, fr.cy = float(Kh[0, 0]), float(Kh[1, 1]), float(Kh[0, 2]), float(Kh[1, 2])
        for name in ("tgt_points", "tgt_norms", "index_map", "tgt_valid"):
            setattr(fr, name, _dev_ptr(getattr(self, name)))
        self.c = fr


class LM_Solver():
    """Drop-in for ``super.LM.LM_Solver``; see module docstring."""

    def __init__(self, opt, convs=None, max_frames=1, shard_surfels=False, rank=None, world=None,
                 all_reduce=None, broadcast=None):
        """``LM_Solver(opt, convs)`` as in the reference.  ``shard_surfels=True`` (after
        ``torch.distributed.init_process_group``) splits the surfels of ONE large frame over the GPUs
        of a node: every rank evaluates its share, the block-sparse J^T J / J^T r sums and the loss are
        all-reduced, every rank solves and takes rank 0's step."""
        self.opt = opt
        self.lib = _lib.load()                       # raises if the HIP library is missing
        self.device = torch.device("cuda", torch.cuda.current_device()) \
            if torch.cuda.is_available() else None
        if self.device is None:
            raise _lib.SuperLMError("no HIP device visible: super_amd has no CPU fallback")
        self.phase = opt.phase
        if self.phase == "train":
            self.convs = convs
        self.max_frames = max_frames
        self._solvers = {}                           # (u, v, minimal_loss) -> handle
        self._bound = [None] * max_frames
        self._prepared = [None] * max_frames         # (handle, ModelView) of a pending slm_prepare_model per slot
        self.last_records = None
        self.rank, self.world = 0, 1
        self.sharded = bool(shard_surfels or world is not None)   # (a world of one rank runs the same protocol)
        self._all_reduce, self._broadcast = all_reduce, broadcast
        if self.sharded:
            import torch.distributed as dist
            if world is None:
                world, rank = dist.get_world_size(), dist.get_rank()
            self.rank, self.world = int(rank), int(world)
            if self._all_reduce is None or self._broadcast is None:
                from .dist import default_collectives
                ar, bc = default_collectives()
                self._all_reduce = self._all_reduce or ar                # sum, in place
                self._broadcast = self._broadcast or bc

    # ---- library handle --------------------------------------------------------------
    def _config(self, u, v, minimal_loss):
        o = self.opt
        c = SlmConfig()
        c.num_iterations = int(o.num_optimize_iterations)
        c.phase_test = 1 if o.phase == "test" else 0
        c.use_data, c.use_arap, c.use_rot = int(bool(o.sf_point_plane)), int(bool(o.mesh_arap)), \
            int(bool(o.mesh_rot))
        c.max_frames = self.max_frames
        c.data_path = int(getattr(o, "slm_data_path", 0))
        c.solver_path = int(getattr(o, "slm_solver_path", 0))
        c.w_data = float(getattr(o, "sf_point_plane_weight", 1.0))
        c.w_arap = float(getattr(o, "mesh_arap_weight", 10.0))
        c.w_rot = float(getattr(o, "mesh_rot_weight", 1.0))
        c.u0, c.v, c.minimal_loss0 = float(u), float(v), float(minimal_loss)
        return c

    def _handle(self, u=10, v=7.5, minimal_loss=1e10):
        key = (float(u), float(v), float(minimal_loss))
        h = self._solvers.get(key)
        if h is None:
            cfg = self._config(*key)
            out = C.c_void_p()
            _lib.check(self.lib.slm_create(C.byref(cfg), C.byref(out)), "slm_create")
            h = out
            if self.sharded:
                _lib.check(self.lib.slm_set_shard(h, self.rank, self.world), "slm_set_shard")
            self._solvers[key] = h
        return h

    def __del__(self):
        try:
            for h in self._solvers.values():
                self.lib.slm_destroy(h)
        except Exception:
            pass

    def _frame(self, h, slot, sf, inputs, new_data):
        """The ``BoundFrame`` of a slot (no library call yet); settles what becomes of a model prepared ahead."""
        state = getattr(self.opt, "slm_state_dtype", None)
        dev = sf.points.device
        prepared = self._prepared[slot]
        self._prepared[slot] = None                    # a prepared model serves one bind
        mv = None
        if prepared is not None and prepared[0] is h and prepared[1].matches(sf, dev, state):
            mv = prepared[1]                           # same arrays, untouched since: the library only binds the target side
        elif prepared is not None:
            # The model changed after prepare_model (other tensors, or the same buffers rewritten in place: the version
            # counters moved).  The library compares sizes and pointers only -- with no-copy inputs (int32 tables, state
            # dtype, contiguous) it would take the stale plan for this frame's: drop it explicitly.
            _lib.check(self.lib.slm_discard_prepared(prepared[0], slot), "slm_discard_prepared")
        return BoundFrame(sf, inputs, new_data, state=state, model=mv)

    def _bind(self, h, slot, sf, inputs, new_data):
        bf = self._frame(h, slot, sf, inputs, new_data)
        _lib.check(self.lib.slm_bind_frame(h, slot, C.byref(bf.c), _stream_ptr(bf.device)),
                   "slm_bind_frame")
        self._bound[slot] = bf
        return bf

    def _bind_batch(self, h, frames):
        """The frames of a batch bound CONCURRENTLY (``slm_bind_frames``: one host thread, stream and scratch set per frame
        inside the library, forked from and joined into the caller's stream) -- slots 0 .. len(frames) - 1."""
        bfs = [self._frame(h, i, *fr) for i, fr in enumerate(frames)]
        arr = (SlmFrame * len(bfs))(*[bf.c for bf in bfs])
        _lib.check(self.lib.slm_bind_frames(h, 0, len(bfs), arr, _stream_ptr(bfs[0].device)), "slm_bind_frames")
        for i, bf in enumerate(bfs):
            self._bound[i] = bf
        return bfs

    def prepare_model(self, sf, slot=0, u=10, v=7.5, minimal_loss=1e10):
        """The model-side half of the NEXT frame's ``loss_term.prepare`` (reference ``super/loss.py:212-220,408-426``),
        ahead of time: call it when the current frame is done with the model -- after ``sf.update`` /
        ``fuseInputData`` / ``prepareStableIndexNSwapAllModel`` (``super/super.py:66-73``) -- and the next ``LM()`` on the
        same, untouched ``sf`` only binds its target (``slm_prepare_model``: the sort, the size read-backs and any
        symbolic analysis run on the library's worker thread and stream while the caller fetches the next frame).
        Optional: ``LM()`` alone does everything, as the reference does.  Not for surfel-sharded solvers' first bind
        order.  When the model changes after all -- other tensors or an in-place write, seen in the tensors' version
        counters -- the next ``LM()`` drops the preparation (``slm_discard_prepared``) and prepares in full."""
        h = self._handle(u, v, minimal_loss)
        mv = ModelView(sf, state=getattr(self.opt, "slm_state_dtype", None))
        fr = SlmFrame()
        mv.fill(fr)
        _lib.check(self.lib.slm_prepare_model(h, slot, C.byref(fr), _stream_ptr(mv.device)), "slm_prepare_model")
        self._prepared[slot] = (h, mv)

    # ---- reference surface -------------------------------------------------------------
    @staticmethod
    def Solver(A, b, method="cholesky"):
        """(reference ``super/LM.py:37-51``) solve A x = b for SPD A on the device; raises
        ``RuntimeError`` when the factorisation fails, like ``torch.linalg.cholesky``."""
        if method == "lu":     # deprecated torch.lu path, never selected by any reference caller
            raise NotImplementedError("super_amd.LM_Solver.Solver: method='lu' is not built (Cholesky only)")
        if method != "cholesky":
            raise ValueError(method)
        lib = _lib.load()
        dev = A.device
        A64 = _as(A, torch.float64, dev)
        b64 = _as(b, torch.float64, dev).reshape(-1)
        P = A64.shape[0]
        x = torch.empty(P, dtype=torch.float64, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.slm_solve_dense(P, _dev_ptr(A64), _dev_ptr(b64), _dev_ptr(x),
                                       _dev_ptr(status), _stream_ptr(dev)), "slm_solve_dense")
        if int(status.item()) != 0:
            raise RuntimeError("cholesky: the input is not positive-definite")
        return x.reshape(b.shape).to(b.dtype)

    def prepareCostTerm(self, sf, inputs, new_data, beta, grad=False):
        """(reference ``super/LM.py:54-78``) ``grad=True`` -> (dense JtJ (P,P), jtl (P,1));
        ``grad=False`` -> scalar sum of squared residuals.  For inspection / parity: the LM
        loop itself never materialises the dense matrix."""
        h = self._handle()
        bf = self._bind(h, 0, sf, inputs, new_data)
        st = _stream_ptr(bf.device)
        b64 = _as(beta, torch.float64, bf.device)
        _lib.check(self.lib.slm_set_beta(h, 0, _dev_ptr(b64), st), "slm_set_beta")
        P = 7 * bf.J
        if grad:
            jtj = torch.empty((P, P), dtype=torch.float64, device=bf.device)
            jtl = torch.empty((P, 1), dtype=torch.float64, device=bf.device)
            _lib.check(self.lib.slm_assemble(h, 0, _dev_ptr(jtj), _dev_ptr(jtl), st), "slm_assemble")
            return jtj, jtl
        out = torch.empty(4, dtype=torch.float64, device=bf.device)
        _lib.check(self.lib.slm_loss(h, 0, _dev_ptr(out), st), "slm_loss")
        return out[:3].sum()

    def LM(self, sf, inputs, new_data, u=10, v=7.5, minimal_loss=1e10):
        """(reference ``super/LM.py:81-122``) run ``opt.num_optimize_iterations`` damped
        iterations on the device; returns beta (J,7) float64 on ``sf``'s device."""
        return self.LM_batch([(sf, inputs, new_data)], u=u, v=v, minimal_loss=minimal_loss)[0]

    def LM_batch(self, frames, u=10, v=7.5, minimal_loss=1e10):
        """Many independent frames / hypotheses advanced together in the same launches
        (one slot each).  ``frames`` is a list of ``(sf, inputs, new_data)``."""
        n = len(frames)
        if n < 1 or n > self.max_frames:
            raise ValueError(f"need 1..{self.max_frames} frames, got {n}")
        h = self._handle(u, v, minimal_loss)
        bfs = self._bind_batch(h, frames) if n > 1 else [self._bind(h, 0, *frames[0])]
        dev = bfs[0].device
        st = _stream_ptr(dev)
        if self.sharded:
            self._run_sharded(h, n, dev)
        else:
            _lib.check(self.lib.slm_run(h, n, st), "slm_run")
        betas, self.last_records = [], []
        for i, (bf, fr) in enumerate(zip(bfs, frames)):
            beta = torch.empty((bf.J, 7), dtype=torch.float64, device=dev)
            _lib.check(self.lib.slm_get_beta(h, i, _dev_ptr(beta), st), "slm_get_beta")
            betas.append(beta)
            recs = self.records(h, i, st)
            self.last_records.append(recs)
            self._report(fr[0], fr[1], recs)
        return betas

    # ---- one frame sharded over several GPUs ------------------------------------------------
    def exchange_buffer(self, h, slot, what, dev):
        n = C.c_int64(0)
        _lib.check(self.lib.slm_lm_exchange_size(h, slot, what, C.byref(n)), "slm_lm_exchange_size")
        return torch.empty(n.value, dtype=torch.float64, device=dev)

    def exchange_view(self, h, slot, what, dev):
        """The library's exchange buffer itself as a tensor (aliased, not copied)."""
        from .dist import device_view
        ptr, n = C.c_void_p(), C.c_int64(0)
        _lib.check(self.lib.slm_lm_exchange_ptr(h, slot, what, C.byref(ptr), C.byref(n)), "slm_lm_exchange_ptr")
        return device_view(ptr.value, n.value, dev)

    def _exchange(self, h, n, what, op, bufs):
        # the collective runs in place on the library's buffer (kernels and collective share the current stream)
        for i in range(n):
            op(bufs[i])

    def _run_sharded(self, h, n, dev):
        """The LM loop with the three exchanges per iteration (see include/super_lm.h)."""
        st = _stream_ptr(dev)
        lib = self.lib
        pair = [self.exchange_view(h, i, _lib.SLM_X_PAIR_BLOCKS, dev) for i in range(n)] if self.opt.sf_point_plane else []
        delta = [self.exchange_view(h, i, _lib.SLM_X_DELTA, dev) for i in range(n)]
        loss = [self.exchange_view(h, i, _lib.SLM_X_DATA_LOSS, dev) for i in range(n)] if self.opt.sf_point_plane else []
        for _ in range(int(self.opt.num_optimize_iterations)):
            _lib.check(lib.slm_lm_grad_local(h, n, st), "slm_lm_grad_local")
            if self.opt.sf_point_plane:
                self._exchange(h, n, _lib.SLM_X_PAIR_BLOCKS, self._all_reduce, pair)
            _lib.check(lib.slm_lm_solve(h, n, st), "slm_lm_solve")
            self._exchange(h, n, _lib.SLM_X_DELTA, self._broadcast, delta)
            _lib.check(lib.slm_lm_loss_local(h, n, st), "slm_lm_loss_local")
            if self.opt.sf_point_plane:
                self._exchange(h, n, _lib.SLM_X_DATA_LOSS, self._all_reduce, loss)
            _lib.check(lib.slm_lm_accept(h, n, st), "slm_lm_accept")

    # ---- helpers -----------------------------------------------------------------------
    def records(self, h, slot, stream):
        n = int(self.opt.num_optimize_iterations)
        arr = (SlmIterRecord * max(n, 1))()
        _lib.check(self.lib.slm_get_records(h, slot, arr, n, stream), "slm_get_records")
        return [dict(loss=r.loss, u=r.u, accepted=bool(r.accepted), status=int(r.status),
                     M_grad=int(r.M_grad), M_loss=int(r.M_loss)) for r in arr[:n]]

    def _report(self, sf, inputs, recs):
        for r in recs:
            if r["status"] == _lib.SLM_ITER_SOLVER_FAILED:
                print("\t\tSolver failed: Ill-posed system!")        # super/LM.py:102
            elif r["status"] == _lib.SLM_ITER_SOLVER_TIMEOUT:          # not a property of the matrix: say so
                print("\t\tSolver failed: the task-graph solve timed out (GPU scheduling), beta kept")
        done = [r for r in recs if r["status"] == _lib.SLM_ITER_OK]
        logger = getattr(sf, "logger", None)                         # defect D1: may be absent
        if self.opt.phase == "test" and logger is not None and done:
            fid = inputs["ID"].item() if "ID" in inputs else -1
            logger.info(f"{fid} loss: {done[-1]['loss']}")
