"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A float64 NumPy restatement of the reference's per-frame embedded-deformation
Levenberg-Marquardt step (``--use_derived_gradient``), written from the maths
(SURVEY.md Appendix A), used only as the *checker* for the HIP path:
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
Nothing under ``python-super_amd/`` imports it.

Parity pinning: the reference ships no tests or golden vectors,
so this oracle is pinned against outputs of the reference itself, imported in
the build container by ``tests/golden/make_golden.py`` and committed as
``tests/golden/*.npz`` (``tests/test_oracle_golden.py`` replays them).
Third-party arithmetic outside ``/root/reference``: ``pytorch3d==0.6.2``
``knn_points`` (tie order unpinned by the reference -> lowest index wins here),
``torch.sparse.mm`` / ``torch.linalg.cholesky`` (reduction order unpinned ->
tolerance parity, 1e-4, by design).

Each function cites the reference lines it restates (paths relative to
``/root/reference``).  Conventions: ``beta`` is (J,7) ``[qw,qx,qy,qz,bx,by,bz]``;
the Jacobian row of one data residual has 4x7 entries, columns
``7*node + 0..6``; ``jtl`` is ``-J^T r`` as the reference stores it.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np

# --------------------------------------------------------------------------- a1/a2


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def skew(a):
    """[a]x with skew(a) @ b == a x b.  Reference ``super/utils.py:4-14`` builds the
    stack so that the same identity holds (SURVEY.md a1 probe)."""
    z = np.zeros_like(a[..., 0])
    return np.stack([np.stack([z, -a[..., 2], a[..., 1]], -1),
                     np.stack([a[..., 2], z, -a[..., 0]], -1),
                     np.stack([-a[..., 1], a[..., 0], z], -1)], -2)


def quat_apply(q, x):
    """R(q)x = x + 2 w (v x x) + 2 v x (v x x) for an UN-normalised quaternion
    (reference ``super/utils.py:49-54``; defect D7: never normalised)."""
    w, v = q[..., 0:1], q[..., 1:4]
    c = cross(v, x)
    return x + 2.0 * w * c + 2.0 * cross(v, c)


def quat_apply_jac(q, x):
    """d(R(q)x)/dq, shape (...,3,4), columns (w,x,y,z) (reference ``super/utils.py:59-69``):
    d/dw = 2 (v x x);  d/dv = 2[(v.x) I + v x^T - 2 x v^T - w [x]x]."""
    w, v = q[..., 0:1], q[..., 1:4]
    dw = 2.0 * cross(v, x)
    eye = np.eye(3)
    vx = np.einsum("...i,...j->...ij", v, x)
    dv = 2.0 * ((v * x).sum(-1)[..., None, None] * eye + vx - 2.0 * np.swapaxes(vx, -1, -2)
                - w[..., None] * skew(x))
    return np.concatenate([dw[..., None], dv], axis=-1)


# --------------------------------------------------------------------------- a3


def skin_points(p, g, knn_idx, knn_w, beta, grad=False):
    """ED skinning T(p) = sum_k w_k [R(q_k)(p - g_k) + b_k + g_k] and the
    w-scaled quaternion Jacobian (reference ``super/utils.py:17-38``,
    ``super/loss.py:212-226``).  Returns T(p) (N,3) and Jq (N,K,3,4) or None."""
    gk = g[knn_idx]                                   # (N,K,3)
    d = p[:, None, :] - gk
    bk = beta[knn_idx]                                # (N,K,7)
    t = quat_apply(bk[..., 0:4], d) + bk[..., 4:7] + gk
    T = (knn_w[..., None] * t).sum(axis=1)
    if not grad:
        return T, None
    return T, quat_apply_jac(bk[..., 0:4], d) * knn_w[..., None, None]


# --------------------------------------------------------------------------- a4


def project(P, K, H, W):
    """Pin-hole projection (reference ``utils/utils.py:161-184``): float (v_,u_),
    rounded pixel id ``coords`` and the margin-0 validity test on ROUNDED coords;
    note Z + 1e-8 here but plain Z in dPi (``super/loss.py:161-173``)."""
    fx, fy, cx, cy = (float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]))
    Z = P[:, 2] + 1e-8
    u_ = P[:, 0] * fx / Z + cx
    v_ = P[:, 1] * fy / Z + cy
    with np.errstate(invalid="ignore"):
        u = np.rint(u_)                               # torch.round: half to even
        v = np.rint(v_)
    ok = np.isfinite(u) & np.isfinite(v) & (np.abs(u) < 2**40) & (np.abs(v) < 2**40)
    ui = np.where(ok, u, -1).astype(np.int64)
    vi = np.where(ok, v, -1).astype(np.int64)
    coords = vi * W + ui
    valid = ok & (vi >= 0) & (vi < H - 1) & (ui >= 0) & (ui < W - 1)
    return v_, u_, coords, valid


# --------------------------------------------------------------------------- a5


def bilinear_taps(v_, u_, index_map):
    """Four taps (fl v,fl u),(fl v,ce u),(ce v,fl u),(ce v,ce u) -> row ids in the
    target table, -1 where the tap is outside the image or unmapped
    (reference ``super/loss.py:107-129``)."""
    H, W = index_map.shape
    fv, cv, fu, cu = np.floor(v_), np.ceil(v_), np.floor(u_), np.ceil(u_)
    n = np.stack([fv, fv, cv, cv], axis=1)
    m = np.stack([fu, cu, fu, cu], axis=1)
    ni, mi = n.astype(np.int64), m.astype(np.int64)
    inside = (ni >= 0) & (ni < H) & (mi >= 0) & (mi < W)
    rows = index_map[np.clip(ni, 0, H - 1), np.clip(mi, 0, W - 1)]
    rows = np.where(inside & (rows >= 0), rows, -1)
    return n, m, rows


def bilinear_lookup(v_, u_, table, index_map, grad=False):
    """Bilinear read of ``table`` rows through ``index_map`` with NaN for any
    invalid tap; weights max(1-|n-v|,0) max(1-|m-u|,0); gradient [...,0] = d/du,
    [...,1] = d/dv with sign(m-u), sign(n-v) (>= 0 -> +1)
    (reference ``super/loss.py:106-157``; defect D4 kept: an exactly integer
    coordinate makes floor == ceil and both taps get weight 1)."""
    n, m, rows = bilinear_taps(v_, u_, index_map)
    U = np.full(rows.shape + (table.shape[1],), np.nan)
    ok = rows >= 0
    U[ok] = table[rows[ok]]
    dn = (n - v_[:, None])[..., None]
    dm = (m - u_[:, None])[..., None]
    an = np.maximum(1.0 - np.abs(dn), 0.0)
    am = np.maximum(1.0 - np.abs(dm), 0.0)
    val = (U * an * am).sum(axis=1)
    if not grad:
        return val, None, rows
    sn = np.where(dn >= 0, 1.0, -1.0)
    sm = np.where(dm >= 0, 1.0, -1.0)
    g = np.stack([(U * an * sm).sum(axis=1), (U * am * sn).sum(axis=1)], axis=2)
    return val, g, rows


# --------------------------------------------------------------------------- a6


def dproj(P, fx, fy):
    """d(u,v)/dP (M,2,3), no epsilon on Z (reference ``super/loss.py:161-173``)."""
    Z = P[:, 2]
    out = np.zeros((len(P), 2, 3))
    out[:, 0, 0] = fx / Z
    out[:, 0, 2] = -fx * P[:, 0] / Z**2
    out[:, 1, 1] = fy / Z
    out[:, 1, 2] = -fy * P[:, 1] / Z**2
    return out


# --------------------------------------------------------------------------- frames


@dataclass
class Frame:
    """Everything ``LM_Solver.LM`` reads from ``sf``, ``inputs`` and ``new_data``
    (SURVEY.md §8b), as float64 / int64 arrays."""
    sf_points: np.ndarray
    sf_knn_idx: np.ndarray
    sf_knn_w: np.ndarray
    ed_points: np.ndarray
    ed_knn_idx: np.ndarray
    tgt_points: np.ndarray
    tgt_norms: np.ndarray
    index_map: np.ndarray
    valid: np.ndarray
    K: np.ndarray
    H: int
    W: int

    @staticmethod
    def from_scene(sc) -> "Frame":
        return Frame(sf_points=sc.f64("sf_points"), sf_knn_idx=sc.sf_knn_idx,
                     sf_knn_w=sc.f64("sf_knn_w"), ed_points=sc.f64("ed_points"),
                     ed_knn_idx=sc.ed_knn_idx, tgt_points=sc.f64("tgt_points"),
                     tgt_norms=sc.f64("tgt_norms"), index_map=sc.index_map, valid=sc.valid,
                     K=sc.K, H=sc.H, W=sc.W)

    @property
    def J(self):
        return len(self.ed_points)


# --------------------------------------------------------------------------- a10


def data_term(fr: Frame, beta, lam, grad=False):
    """Point-to-plane ICP residuals r = lam * n.(T(p)-o) on the match set
    S = valid_pair & all-taps-valid & proj_valid, and (grad) the 4x7 Jacobian
    entries per residual (reference ``super/loss.py:222-290``; the
    match-set definition resolves defect D2)."""
    fx, fy = float(fr.K[0, 0]), float(fr.K[1, 1])
    T, Jq = skin_points(fr.sf_points, fr.ed_points, fr.sf_knn_idx, fr.sf_knn_w, beta, grad)
    v_, u_, coords, proj_valid = project(T, fr.K, fr.H, fr.W)
    inrange = (coords >= 0) & (coords < len(fr.valid))
    valid_pair = inrange & fr.valid[np.clip(coords, 0, len(fr.valid) - 1)]
    cand = np.nonzero(valid_pair & proj_valid)[0]
    o, dodc, rows = bilinear_lookup(v_[cand], u_[cand], fr.tgt_points, fr.index_map, grad)
    n, dndc, _ = bilinear_lookup(v_[cand], u_[cand], fr.tgt_norms, fr.index_map, grad)
    ok = ~(np.isnan(o).any(1) | np.isnan(n).any(1))
    sel = cand[ok]
    o, n, rows = o[ok], n[ok], rows[ok]
    Tm = T[sel]
    e = Tm - o
    r = lam * (n * e).sum(1)
    out = SimpleNamespace(match=sel, taps=rows, r=r, T=T, v=v_, u=u_, coords=coords)
    if not grad:
        return out
    Pi = dproj(Tm, fx, fy)                            # (M,2,3)
    A = dodc[ok] @ Pi                                 # do/dT   (M,3,3)
    B = dndc[ok] @ Pi                                 # dn/dT   (M,3,3)
    c = n - np.einsum("mi,mij->mj", n, A) + np.einsum("mi,mij->mj", e, B)   # (M,3)
    w = fr.sf_knn_w[sel]                              # (M,K)
    jq = np.einsum("mi,mkij->mkj", c, Jq[sel])        # (M,K,4): c . (w_k dR(q_k)d_k/dq)
    jb = w[..., None] * c[:, None, :]                 # (M,K,3)
    out.Jrow = lam * np.concatenate([jq, jb], axis=2)  # (M,K,7)
    out.nodes = fr.sf_knn_idx[sel]                    # (M,K)
    return out


# --------------------------------------------------------------------------- a11


def arap_term(fr: Frame, beta, lam, grad=False):
    """ARAP residuals r_{jk} = lam [R(q_k) d + b_k - d - b_j], d = g_j - g_k,
    k in KNN_ED(j), UN-weighted in the LM path (defect D5); row order
    (j*K_ED + slot)*3 + c (reference ``super/loss.py:408-455``)."""
    g, nb = fr.ed_points, fr.ed_knn_idx
    d = g[:, None, :] - g[nb]                         # (J,K,3)
    bk = beta[nb]
    r = quat_apply(bk[..., 0:4], d) + bk[..., 4:7] - d - beta[:, None, 4:7]
    out = SimpleNamespace(r=lam * r.reshape(-1))
    if grad:
        out.Jq = lam * quat_apply_jac(bk[..., 0:4], d)  # (J,K,3,4) -> cols 7k+0..3
        out.lam = lam                                   # +lam at 7k+4+c, -lam at 7j+4+c
    return out


# --------------------------------------------------------------------------- a12


def rot_term(beta, lam, grad=False):
    """Rot residual r_j = lam (1 - |q_j|^2), evaluated in FLOAT32 like the
    reference (``super/loss.py:487-499``)."""
    q = beta[:, 0:4].astype(np.float32)
    lam32 = np.float32(lam)
    r = lam32 * (np.float32(1.0) - (q * q).sum(axis=1, dtype=np.float32))
    out = SimpleNamespace(r=r.astype(np.float64))
    if grad:
        out.Jq = (-lam32 * np.float32(2.0) * q).astype(np.float64)    # (J,4) at 7j+0..3
    return out


# --------------------------------------------------------------------------- a7/a8/a13


def _terms(opt):
    return (bool(opt.sf_point_plane), bool(opt.mesh_arap), bool(opt.mesh_rot))


def total_loss(fr: Frame, beta, opt):
    """sum of squared residuals over the enabled terms, each re-evaluated at
    ``beta`` with a fresh match set (reference ``super/LM.py:70-78``)."""
    use_d, use_a, use_r = _terms(opt)
    s, M = 0.0, 0
    if use_d:
        t = data_term(fr, beta, opt.sf_point_plane_weight)
        s += float((t.r**2).sum())
        M = len(t.r)
    if use_a:
        s += float((arap_term(fr, beta, opt.mesh_arap_weight).r**2).sum())
    if use_r:
        s += float((rot_term(beta, opt.mesh_rot_weight).r**2).sum())
    return s, M


def jacobian_coo(fr: Frame, beta, opt):
    """Per-term sparse Jacobians as (rows, cols, vals, nrows) COO triplets plus the
    residual vectors, in the reference's row/entry order
    (``super/loss.py:178-197,277-288,414-426,447-455,482-499``)."""
    use_d, use_a, use_r = _terms(opt)
    out = {}
    if use_d:
        t = data_term(fr, beta, opt.sf_point_plane_weight, grad=True)
        M = len(t.r)
        rows = np.repeat(np.arange(M), 7 * t.nodes.shape[1])     # (K = num_neighbors columns blocks of 7 per residual)
        cols = (7 * t.nodes[:, :, None] + np.arange(7)[None, None, :]).reshape(-1)
        out["data"] = (rows, cols, t.Jrow.reshape(-1), M, t.r, t)
    if use_a:
        a = arap_term(fr, beta, opt.mesh_arap_weight, grad=True)
        J, Ke = fr.ed_knn_idx.shape
        ridx = np.arange(J * Ke * 3).reshape(J, Ke, 3)
        k = fr.ed_knn_idx
        rows = [np.repeat(ridx[..., None], 4, axis=-1).reshape(-1)]
        cols = [(7 * k[:, :, None, None] + np.arange(4)[None, None, None, :]
                 + np.zeros((1, 1, 3, 1), np.int64)).reshape(-1)]
        vals = [a.Jq.reshape(-1)]
        rows.append(ridx.reshape(-1))
        cols.append((7 * k[:, :, None] + 4 + np.arange(3)[None, None, :]).reshape(-1))
        vals.append(np.full(J * Ke * 3, a.lam))
        rows.append(ridx.reshape(-1))
        cols.append((7 * np.arange(J)[:, None, None] + 4 + np.arange(3)[None, None, :]
                     + np.zeros((1, Ke, 1), np.int64)).reshape(-1))
        vals.append(np.full(J * Ke * 3, -a.lam))
        out["arap"] = (np.concatenate(rows), np.concatenate(cols), np.concatenate(vals),
                       J * Ke * 3, a.r, a)
    if use_r:
        t = rot_term(beta, opt.mesh_rot_weight, grad=True)
        J = len(beta)
        rows = np.repeat(np.arange(J), 4)
        cols = (7 * np.arange(J)[:, None] + np.arange(4)[None, :]).reshape(-1)
        out["rot"] = (rows, cols, t.Jq.reshape(-1), J, t.r, t)
    return out


def normal_equations(fr: Frame, beta, opt, dense=True):
    """JtJ (P,P) and jtl = -Jt r (P,) summed over the enabled terms in the order
    Data, ARAP, Rot (reference ``super/LM.py:54-68``, ``super/loss.py:200-205``)."""
    import scipy.sparse as sp
    P = 7 * fr.J
    JtJ = sp.csr_matrix((P, P))
    jtl = np.zeros(P)
    M = 0
    for name, (rows, cols, vals, nrows, r, _) in jacobian_coo(fr, beta, opt).items():
        Jm = sp.coo_matrix((vals, (rows, cols)), shape=(nrows, P)).tocsr()
        JtJ = JtJ + (Jm.T @ Jm)
        jtl -= Jm.T @ r
        if name == "data":
            M = nrows
    return (JtJ.toarray() if dense else JtJ.tocsr()), jtl, M


# --------------------------------------------------------------------------- a14/a15


def solve_damped(JtJ_dense, jtl, u):
    """(JtJ + u I) delta = jtl by Cholesky; raises ``np.linalg.LinAlgError`` when not
    SPD (reference ``super/LM.py:37-51,97-103``)."""
    from scipy.linalg import cho_factor, cho_solve
    A = JtJ_dense.copy()
    A[np.diag_indices_from(A)] += u
    return cho_solve(cho_factor(A, lower=True, check_finite=False), jtl, check_finite=False)


def default_opt(**kw):
    """Hot-path flags with the reference defaults (``options.py:26-51,213-238``)."""
    o = SimpleNamespace(sf_point_plane=True, sf_point_plane_weight=1.0, mesh_arap=True,
                        mesh_arap_weight=10.0, mesh_rot=True, mesh_rot_weight=1.0,
                        num_optimize_iterations=10, phase="test")
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def solve_damped_sparse(JtJ_csr, jtl, u):
    """The same system through a sparse direct solve (SciPy's SuperLU, symmetric mode, no pivoting off the diagonal) on
    the block-sparse JtJ: what makes full-size traces at 4 k nodes affordable for the checker (the dense factor is
    6 GB / 7 TFLOP at P = 28 000).  Not positive definite -- a non-positive pivot, or a pivot taken off the diagonal --
    raises ``np.linalg.LinAlgError`` like the dense Cholesky of the reference would (``super/LM.py:47-51,99-103``)."""
    import scipy.sparse as sp
    from scipy.sparse.linalg import splu
    n = JtJ_csr.shape[0]
    A = (JtJ_csr + u * sp.identity(n, format="csr")).tocsc()
    try:
        lu = splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    except RuntimeError as e:                                  # exactly singular
        raise np.linalg.LinAlgError(str(e))
    # positive definite <=> elimination on the diagonal (row and column permutations coincide) with positive pivots
    if not np.array_equal(lu.perm_r, lu.perm_c) or not (lu.U.diagonal() > 0.0).all():
        raise np.linalg.LinAlgError("matrix is not positive definite")
    return lu.solve(jtl)


def lm(fr: Frame, opt, u=10.0, v=7.5, minimal_loss=1e10, trace=None, solve="dense"):
    """The damped accept/reject loop (reference ``super/LM.py:81-122``): beta0 =
    identity; per iteration build, damp, solve (failure -> stop), step, re-evaluate
    the loss with a fresh match set, accept (u /= v) or reject (u *= v, roll back).
    Returns beta (J,7).  ``trace`` (a list) receives one dict per iteration.
    ``solve``: "dense" (Cholesky of the dense matrix like the reference) or "sparse" (``solve_damped_sparse``: the same
    system, checked against the dense path on the reference's goldens in tests/test_oracle_golden.py)."""
    beta = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0]), (fr.J, 1))
    best = beta.copy()
    for it in range(opt.num_optimize_iterations):
        JtJ, jtl, M = normal_equations(fr, beta, opt, dense=(solve == "dense"))
        try:
            delta = (solve_damped(JtJ, jtl, u) if solve == "dense" else solve_damped_sparse(JtJ, jtl, u)).reshape(-1, 7)
        except np.linalg.LinAlgError:
            if trace is not None:
                trace.append(dict(it=it, status="solver_failed", u=u))
            break
        beta = beta + delta
        loss, Mn = total_loss(fr, beta, opt)
        u_used = u
        accepted = True
        if opt.phase == "test":
            if loss < minimal_loss:
                minimal_loss = loss
                u /= v
                best = beta.copy()
            else:
                accepted = False
                u *= v
                beta = best.copy()
        if trace is not None:
            trace.append(dict(it=it, loss=loss, u=u_used, accepted=accepted, M_grad=M,
                              M_loss=Mn, beta=beta.copy(), delta=delta.copy()))
    return beta


# --------------------------------------------------------------------------- a16


def _normalize(x, eps=1e-12):
    """torch.nn.functional.normalize: x / max(|x|, eps)."""
    return x / np.maximum(np.linalg.norm(x, axis=-1, keepdims=True), eps)


def apply_update(sf_points, sf_norms, sf_knn_idx, sf_knn_w, ed_points, ed_norms, beta):
    """``Surfels.update`` for the LM path (no global row): skin points, blend and
    normalise normals, translate nodes, rotate node normals
    (reference ``super/nodes.py:193-223``).  Returns the four updated arrays."""
    new_p, _ = skin_points(sf_points, ed_points, sf_knn_idx, sf_knn_w, beta)
    bk = beta[sf_knn_idx]
    rn = quat_apply(bk[..., 0:4], np.broadcast_to(sf_norms[:, None, :], bk[..., 1:4].shape))
    rn = rn + bk[..., 4:7]        # transformQuatT adds b when beta has 7 columns (nodes.py:207-209)
    new_n = _normalize((sf_knn_w[..., None] * rn).sum(axis=1))
    new_g = ed_points + beta[:, 4:7]
    new_gn = _normalize(quat_apply(beta[:, 0:4], ed_norms))
    return new_p, new_n, new_g, new_gn


# --------------------------------------------------------------------------- a17


def knn(points, nodes, k):
    """K nearest nodes per point: squared L2 ascending, ties -> lowest index;
    returns (dist = sqrt(d2), idx) (reference ``utils/utils.py:212-221`` over
    ``pytorch3d.ops.knn_points``, absent here -- see module header)."""
    a = np.asarray(points, np.float64)
    b = np.asarray(nodes, np.float64)
    idx = np.empty((len(a), k), np.int64)
    d2o = np.empty((len(a), k), np.float64)
    for s in range(0, len(a), 16384):
        d2 = ((a[s:s + 16384, None, :] - b[None, :, :]) ** 2).sum(-1)
        order = np.argsort(d2, axis=1, kind="stable")[:, :k]
        idx[s:s + 16384] = order
        d2o[s:s + 16384] = np.take_along_axis(d2, order, axis=1)
    return np.sqrt(d2o), idx


def knn_weights(dist, radii_of_nbrs):
    """softmax(exp(-dist/radius)) (reference ``super/nodes.py:166,191``)."""
    e = np.exp(-dist / radii_of_nbrs)
    s = np.exp(e - e.max(axis=-1, keepdims=True))
    return s / s.sum(axis=-1, keepdims=True)


def surfel_knn(points, ed_points, ed_radii, k, is_stable=None):
    """``Surfels.update_sfed_knn`` (reference ``super/nodes.py:170-191``): indices,
    weights and the stability test any(dist <= radius)."""
    dist, idx = knn(points, ed_points, k)
    rad = ed_radii[idx]
    stable = (dist <= rad).any(axis=1)
    if is_stable is not None:
        stable &= is_stable
    return idx, knn_weights(dist, rad), stable, dist


def node_knn(ed_points, ed_radii, k_ed):
    """``Surfels.update_ed`` (reference ``super/nodes.py:154-168``): K_ED+1 nearest,
    self dropped, weights softmax(exp(-dist/radius_self))."""
    dist, idx = knn(ed_points, ed_points, k_ed + 1)
    dist, idx = dist[:, 1:], idx[:, 1:]
    return idx, knn_weights(dist, ed_radii[:, None]), dist
