"""CPU ORACLE (test infrastructure, NOT product code).

A plain-numpy restatement of PIN-SLAM's per-frame neural-point SDF hot path, used
only as the checker by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  The product path (``pin_slam_amd``) never
imports this package.

Every function cites the reference file:line it restates (paths relative to the
PRBonn/PIN_SLAM tree).  The reference ships no tests or golden vectors (SURVEY.md
section 4), so the oracle is *pinned by executing the reference itself*:
``oracle/make_golden.py`` imports the unmodified reference modules on CPU in the
build container and writes ``tests/golden/*.npz``; ``tests/test_oracle_vs_golden.py``
checks every function below against those fixtures (and, when the reference tree
is present, against live reference calls).

Conventions
* float32 everywhere the reference computes in float32, int64 hashing, and the
  same evaluation order for ``d2`` so neighbour indices are bit-exact.
* Unstable-sort ties in the reference (``torch.sort`` over candidate distances,
  neural_points.py:584) are canonicalised as (d2, candidate order); invalid
  candidates all carry index -1 so their order is immaterial.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
PRIMES = np.array([73856093, 19349669, 83492791], dtype=np.int64)  # neural_points.py:82-84
IDW_EPS = F32(1e-15)  # neural_points.py:665
INVALID_D2 = F32(9e3)  # neural_points.py:583


# --------------------------------------------------------------------------- K1
def search_neighborhood(num_nei_cells: int, search_alpha: float, resolution: float):
    """neural_points.py:910-948 -> (neighbor_dx [Kc,3] int64, max_valid_dist2 float)."""
    n = int(num_nei_cells)
    r = np.arange(-n, n + 1, dtype=np.int64)
    gx, gy, gz = np.meshgrid(r, r, r, indexing="ij")
    dx = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3)
    keep = (dx ** 2).sum(-1) < (n + search_alpha) ** 2
    return np.ascontiguousarray(dx[keep]), 3 * ((n + 1) * resolution) ** 2


def grid_coords(points: np.ndarray, resolution: float) -> np.ndarray:
    """floor(p / res) in float32 true division, then int64 (neural_points.py:963)."""
    return np.floor(points.astype(F32) / F32(resolution)).astype(np.int64)


def hash_slots(cells: np.ndarray, buffer_size: int) -> np.ndarray:
    """fmod(sum(cell*primes), B) with the negative-index wrap of ``table[hash]``
    (neural_points.py:972-978): slot = h < 0 ? h + B : h."""
    h = (cells.astype(np.int64) * PRIMES).sum(-1)
    h = np.fmod(h, np.int64(buffer_size))
    return np.where(h < 0, h + np.int64(buffer_size), h)


def _d2(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    """sum((P - q)**2, -1) in float32, evaluated (dx*dx + dy*dy) + dz*dz like
    torch's 3-element reduction (neural_points.py:992-995)."""
    d = (p.astype(F32) - q.astype(F32)).astype(F32)
    s = d * d
    return ((s[..., 0] + s[..., 1]).astype(F32) + s[..., 2]).astype(F32)


def radius_search(points, table, positions, resolution, neighbor_dx, max_valid_dist2,
                  ts_create=None, travel_dist=None, cur_ts=0, diff_travel_dist_local=None):
    """NeuralPoints.radius_neighborhood_search (neural_points.py:950-1009).

    Returns (dist2 [N,Kc] f32, idx [N,Kc] int64 global indices, -1 = invalid).
    ``travel_dist is not None`` switches the travel-distance window filter on
    (neural_points.py:982-988)."""
    points = np.asarray(points, dtype=F32)
    B = table.shape[0]
    N, Kc = points.shape[0], neighbor_dx.shape[0]
    mv = F32(max_valid_dist2)
    if positions.shape[0] == 0:
        return np.full((N, Kc), mv, F32), np.full((N, Kc), -1, np.int64)
    g = grid_coords(points, resolution)
    cells = g[:, None, :] + neighbor_dx[None, :, :]
    slot = hash_slots(cells, B)
    idx = table[slot].astype(np.int64)
    if travel_dist is not None:
        td = np.asarray(travel_dist, dtype=F32)
        diff = np.abs(td[cur_ts] - td[ts_create[idx]])  # idx == -1 reads the last point, as in torch
        idx = np.where(diff < F32(diff_travel_dist_local), idx, -1)
    d2 = _d2(positions[idx], points[:, None, :])
    d2 = np.where(idx == -1, mv, d2).astype(F32)
    idx = np.where(d2 > mv, -1, idx)
    return d2, idx


# --------------------------------------------------------------------------- K2
def select_knn(d2, idx, k):
    """Top-k of query_feature (neural_points.py:583-589) with the canonical
    (d2, candidate order) tie-break.  ``idx`` is already in the index space that is
    gathered from (local or global); invalid = -1."""
    d2 = np.where(idx == -1, INVALID_D2, d2).astype(F32)
    order = np.argsort(d2, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(d2, order, 1), np.take_along_axis(idx, order, 1)


def quat_rotate(quat, vec):
    """utils/tools.py:428-437 apply_quaternion_rotation: p' = p + w t + (-q_xyz) x t with
    t = 2 (-q_xyz) x p, i.e. rotation by the CONJUGATE quaternion (w,x,y,z): p' = R(q)^T p.
    Returns (p', M) with M = R(q)^T so that p' = M p."""
    q0, q1, q2, q3 = quat[..., 0], quat[..., 1], quat[..., 2], quat[..., 3]
    R = np.stack([
        1 - 2 * (q2 ** 2 + q3 ** 2), 2 * (q1 * q2 - q0 * q3), 2 * (q1 * q3 + q0 * q2),
        2 * (q1 * q2 + q0 * q3), 1 - 2 * (q1 ** 2 + q3 ** 2), 2 * (q2 * q3 - q0 * q1),
        2 * (q1 * q3 - q0 * q2), 2 * (q2 * q3 + q0 * q1), 1 - 2 * (q1 ** 2 + q2 ** 2),
    ], axis=-1).reshape(quat.shape[:-1] + (3, 3)).astype(vec.dtype)
    R = np.swapaxes(R, -1, -2)
    return np.einsum("...ij,...j->...i", R, vec), R


def idw_weights(d2k, valid, nn_count, dtype=F32):
    """neural_points.py:665-683."""
    u = (dtype(1.0) / (d2k.astype(dtype) + dtype(1e-15))).astype(dtype)
    u = np.where(valid, u, dtype(0))
    u = np.where((nn_count == 0)[:, None], dtype(1e-15), u)
    S = u.sum(1, keepdims=True, dtype=dtype)
    w = np.where(valid, u / S, dtype(0)).astype(dtype)
    return u, S, w


def query_feature(points, search, feats, positions, certainties=None, k=6,
                  global2local=None, orientations=None, weighted_first=True,
                  training_mode=False, query_ts=None, ts_update=None):
    """NeuralPoints.query_feature (neural_points.py:530-746) for the geometric feature.

    ``search`` = (d2, idx) from :func:`radius_search` (global indices).  When
    ``global2local`` is given, ``feats/positions/certainties/orientations`` are the
    *local* arrays.  Returns a dict with geo_feat ([N,F+3] or [N,k,F+3]), weight
    [N,k,1], nn_count [N], certainty [N], knn_idx [N,k], knn_d2 [N,k], and -- in
    training mode -- the updated certainties / ts_update copies (the side effects of
    neural_points.py:685-710)."""
    points = np.asarray(points, F32)
    d2, idx = search
    if global2local is not None:
        idx = global2local[idx]  # -1 -> last entry, which is -1 (neural_points.py:573, :505)
    nn_count = (idx >= 0).sum(-1)
    d2k, idxk = select_knn(d2, idx, k)
    valid = idxk >= 0
    F = feats.shape[1]
    gather = np.where(valid, idxk, 0)
    f = np.where(valid[..., None], feats[gather], F32(0)).astype(F32)
    v = (points[:, None, :] - positions[gather]).astype(F32)
    if orientations is not None:  # after_pgo (neural_points.py:645-648)
        v, _ = quat_rotate(orientations[gather].astype(F32), v)
        v = v.astype(F32)
    v = np.where(valid[..., None], v, F32(0))
    fv = np.concatenate([f, v], -1)
    u, S, w = idw_weights(d2k, valid, nn_count)
    out = dict(nn_count=nn_count.astype(np.int64), weight=w[..., None], knn_idx=idxk, knn_d2=d2k)
    if certainties is not None:
        c = np.where(valid, certainties[gather], F32(0))
        out["certainty"] = (c * w).sum(1, dtype=F32)
    if training_mode:
        newc = certainties.copy()
        np.add.at(newc, gather.ravel(), w.ravel())  # invalid -> row 0, weight 0 (neural_points.py:689)
        out["certainties_after"] = newc
        if query_ts is not None and ts_update is not None:
            newts = ts_update.copy()
            ts = np.where(valid, np.asarray(query_ts)[:, None], 0).astype(newts.dtype)
            np.maximum.at(newts, gather.ravel(), ts.ravel())
            out["ts_update_after"] = newts
    out["geo_feat"] = (fv * w[..., None]).sum(1, dtype=F32) if weighted_first else fv
    return out


# --------------------------------------------------------------------------- K3
def unpack_decoder(flat, in_dim, hidden, levels, out_dim=1):
    """Split the flat parameter buffer (layers.0.weight, layers.0.bias, ...,
    lout.weight, lout.bias -- the state_dict order of model/decoder.py:46-53)."""
    flat = np.asarray(flat)
    Ws, bs, o = [], [], 0
    d = in_dim
    for _ in range(levels):
        Ws.append(flat[o:o + hidden * d].reshape(hidden, d)); o += hidden * d
        bs.append(flat[o:o + hidden]); o += hidden
        d = hidden
    Wo = flat[o:o + out_dim * d].reshape(out_dim, d); o += out_dim * d
    bo = flat[o:o + out_dim]; o += out_dim
    assert o == flat.size, (o, flat.size)
    return Ws, bs, Wo, bo


def decoder_param_count(in_dim, hidden, levels, out_dim=1):
    n, d = 0, in_dim
    for _ in range(levels):
        n += hidden * d + hidden
        d = hidden
    return n + out_dim * d + out_dim


def mlp_forward(z, params, keep=False):
    """Decoder.mlp (model/decoder.py:61-80): (Linear+ReLU) x L, Linear out."""
    Ws, bs, Wo, bo = params
    h = z
    acts = [z]
    for W, b in zip(Ws, bs):
        h = np.maximum(h @ W.T + b, 0).astype(z.dtype)
        acts.append(h)
    out = (h @ Wo.T + bo).astype(z.dtype)
    return (out, acts) if keep else out


def mlp_input_jacobian(acts, params):
    """d out[...,0] / d z, back through the ReLU masks (what autograd's get_gradient,
    utils/tools.py:247-260, computes through Decoder.mlp)."""
    Ws, bs, Wo, bo = params
    a = np.broadcast_to(Wo[0], acts[-1].shape).astype(acts[0].dtype)
    for li in range(len(Ws) - 1, -1, -1):
        a = (a * (acts[li + 1] > 0)) @ Ws[li]
    return a


def sdf_and_grad(points, qf, params, sdf_scale, positions, orientations=None, dtype=np.float64):
    """SDF value, analytic d sdf/d q and (not weighted_first) the std over neighbours.

    Restates Tracker.query_source_points (utils/tracker.py:297-335): Decoder.sdf on the
    interpolated feature, autograd gradient through the MLP, the neighbour vectors
    AND the IDW weights.  ``qf`` is the dict returned by
    :func:`query_feature` called with weighted_first=False (per-neighbour vectors);
    the mode evaluated here is chosen by ``weighted_first`` stored in qf['mode'].
    Computed in ``dtype`` (float64 by default: the reference's own float32 autograd
    carries ~5e-5 relative noise, tests compare with 1e-4)."""
    T = dtype
    idxk, d2k = qf["knn_idx"], qf["knn_d2"].astype(T)
    valid = idxk >= 0
    nn = qf["nn_count"]
    fv = qf["fv"].astype(T)  # [N,k,F+3]
    params = tuple([w.astype(T) for w in p] if isinstance(p, list) else p.astype(T) for p in params)
    gather = np.where(valid, idxk, 0)
    diff = (np.asarray(points, T)[:, None, :] - positions[gather].astype(T))  # q - P_t
    u = np.where(valid, 1.0 / (d2k + T(1e-15)), 0.0)
    u = np.where((nn == 0)[:, None], T(1e-15), u)
    S = u.sum(1, keepdims=True)
    w = np.where(valid, u / S, 0.0)
    g_u = np.where(valid[..., None], -2.0 * (u ** 2)[..., None] * diff, 0.0)  # d u_t / d q
    G = g_u.sum(1, keepdims=True)
    dw = np.where(valid[..., None], g_u / S[..., None] - (u / S ** 2)[..., None] * G, 0.0)  # [N,k,3]
    s = T(sdf_scale)
    Fdim = fv.shape[-1] - 3
    if orientations is not None:
        _, R = quat_rotate(orientations[gather].astype(T), diff)
    if qf["mode"] == "weighted_first":
        z = (fv * w[..., None]).sum(1)
        out, acts = mlp_forward(z, params, keep=True)
        a = mlp_input_jacobian(acts, params)  # [N,F+3]
        c = np.einsum("nf,nkf->nk", a, fv)
        if orientations is None:
            direct = a[:, Fdim:] * w.sum(1, keepdims=True)
        else:
            direct = np.einsum("nk,nkij,ni->nj", w, R, a[:, Fdim:])
        grad = s * (direct + np.einsum("nk,nkj->nj", c, dw))
        return (s * out[:, 0]), grad, np.zeros(len(z), T)
    out, acts = mlp_forward(fv, params, keep=True)  # [N,k,1]
    a = mlp_input_jacobian(acts, params)  # [N,k,F+3]
    sk = s * out[..., 0]
    mean = (sk * w).sum(1)
    if orientations is None:
        direct = (w[..., None] * a[..., Fdim:]).sum(1)
    else:
        direct = np.einsum("nk,nkij,nki->nj", w, R, a[..., Fdim:])
    grad = np.einsum("nk,nkj->nj", sk, dw) + s * direct
    std = np.sqrt((w * (sk - mean[:, None]) ** 2).sum(1))
    return mean, grad, std


def query_sdf(points, search, feats, positions, params, sdf_scale, k, weighted_first=True,
              global2local=None, certainties=None, orientations=None, with_grad=True,
              dtype=np.float64):
    """Convenience: radius-search result -> (sdf, grad, std, nn_count, certainty)."""
    qf = query_feature(points, search, feats, positions, certainties, k, global2local,
                       orientations, weighted_first=False)
    qf["fv"] = qf["geo_feat"]
    qf["mode"] = "weighted_first" if weighted_first else "per_neighbor"
    sdf, grad, std = sdf_and_grad(points, qf, params, sdf_scale, positions, orientations, dtype)
    return sdf, grad, std, qf["nn_count"], qf.get("certainty")


# --------------------------------------------------------------------------- colour head (C5)
INTENSITY = np.array([0.299, 0.587, 0.114])  # utils/tools.py:408-410 color_to_intensity


def mlp_input_jacobian_multi(acts, params):
    """d out[..., c] / d z for every output channel c -> [..., C, in]."""
    Ws, bs, Wo, bo = params
    outs = []
    for c in range(Wo.shape[0]):
        a = np.broadcast_to(Wo[c], acts[-1].shape).astype(acts[0].dtype)
        for li in range(len(Ws) - 1, -1, -1):
            a = (a * (acts[li + 1] > 0)) @ Ws[li]
        outs.append(a)
    return np.stack(outs, -2)


def query_color(points, search, cfeats, positions, params, k, weighted_first=True, global2local=None,
                orientations=None, dtype=np.float64):
    """Decoder.regress_color (model/decoder.py:112: sigmoid(mlp)) on the interpolated COLOUR
    features and the per-channel autograd gradients of Tracker.query_source_points
    (utils/tracker.py:342-350).  Returns (color [N,C], color_grad [N,C,3], nn_count)."""
    T = dtype
    qf = query_feature(points, search, cfeats, positions, None, k, global2local, orientations, weighted_first=False)
    fv = qf["geo_feat"].astype(T)
    idxk, d2k = qf["knn_idx"], qf["knn_d2"].astype(T)
    valid = idxk >= 0
    nn = qf["nn_count"]
    params = tuple([w.astype(T) for w in p] if isinstance(p, list) else p.astype(T) for p in params)
    gather = np.where(valid, idxk, 0)
    diff = np.asarray(points, T)[:, None, :] - positions[gather].astype(T)
    u = np.where(valid, 1.0 / (d2k + T(1e-15)), 0.0)
    u = np.where((nn == 0)[:, None], T(1e-15), u)
    S = u.sum(1, keepdims=True)
    w = np.where(valid, u / S, 0.0)
    g_u = np.where(valid[..., None], -2.0 * (u ** 2)[..., None] * diff, 0.0)
    G = g_u.sum(1, keepdims=True)
    dw = np.where(valid[..., None], g_u / S[..., None] - (u / S ** 2)[..., None] * G, 0.0)
    Fdim = fv.shape[-1] - 3
    if orientations is not None:
        _, R = quat_rotate(orientations[gather].astype(T), diff)
    if weighted_first:
        z = (fv * w[..., None]).sum(1)
        out, acts = mlp_forward(z, params, keep=True)
        p = _sigmoid(out)  # [N,C]
        a = mlp_input_jacobian_multi(acts, params) * (p * (1 - p))[..., None]  # [N,C,in]
        c = np.einsum("ncf,nkf->nck", a, fv)
        if orientations is None:
            direct = a[..., Fdim:] * w.sum(1)[:, None, None]
        else:
            direct = np.einsum("nk,nkij,nci->ncj", w, R, a[..., Fdim:])
        return p, direct + np.einsum("nck,nkj->ncj", c, dw), nn
    out, acts = mlp_forward(fv, params, keep=True)  # [N,k,C]
    p = _sigmoid(out)
    a = mlp_input_jacobian_multi(acts, params) * (p * (1 - p))[..., None]  # [N,k,C,in]
    col = (p * w[..., None]).sum(1)
    if orientations is None:
        direct = (w[..., None, None] * a[..., Fdim:]).sum(1)
    else:
        direct = np.einsum("nk,nkij,nkci->ncj", w, R, a[..., Fdim:])
    return col, np.einsum("nkc,nkj->ncj", p, dw) + direct, nn


# --------------------------------------------------------------------------- K5
def expmap(t):
    """utils/tracker.py:784-795."""
    angle = np.linalg.norm(t)
    axis = t / angle
    S = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + S * np.sin(angle) + (S @ S) * (1.0 - np.cos(angle))


def registration_step(points, sdf, grad, std, nn_count, *, valid_nn_k, min_grad_norm=0.5,
                      max_grad_norm=2.0, max_sdf_std=0.25, GM_dist=0.3, GM_grad=0.1,
                      lm_lambda=1e-4, sdf_labels=None, colors=None, color_pred=None, color_grad=None,
                      photo_loss=False, photo_weight=0.01, consist_weight=True, dist_div_grad_norm=False):
    """Tracker.registration_step + implicit_reg (utils/tracker.py:409-524, 615-695),
    geometric term only.  Returns dict(T [4,4] f64, valid_count, residual_cm, N, g)."""
    points = np.asarray(points, np.float64)
    sdf = np.asarray(sdf, np.float64)
    grad = np.asarray(grad, np.float64)
    gn = np.linalg.norm(grad, axis=-1)
    valid = (nn_count >= valid_nn_k) & (gn < max_grad_norm) & (gn > min_grad_norm) & (np.asarray(std) < max_sdf_std)
    n = int(valid.sum())
    if n < 10:  # tracker.py:430-432
        return dict(T=np.eye(4), valid_count=n, residual_cm=0.0, valid=valid)
    p, g, r, gnv = points[valid], grad[valid], sdf[valid], gn[valid]
    if dist_div_grad_norm:  # reg_dist_div_grad_norm, tracker.py:452-456 (the Jacobian keeps the plain gradient)
        r = r / gnv
    if sdf_labels is not None:
        r = r - np.asarray(sdf_labels, np.float64)[valid]
    w = np.ones(n)
    if GM_grad is not None:
        w = w * (GM_grad / (GM_grad + (gnv - 1.0) ** 2)) ** 2
    if GM_dist is not None:
        w = w * (GM_dist / (GM_dist + r ** 2)) ** 2
    photo_res = None
    if colors is not None:  # tracker.py:493-514 (3 channels -> intensity)
        ci = np.asarray(colors, np.float64)[valid] @ INTENSITY
        pi = np.asarray(color_pred, np.float64)[valid] @ INTENSITY
        if (not photo_loss) and consist_weight:
            w = w * np.exp(-np.abs(ci - pi))
    w = w / (2.0 * w.mean())  # tracker.py:524
    J = np.concatenate([np.cross(p, g), g], -1)
    N = J.T @ (w[:, None] * J)
    b = -(J * w[:, None]).T @ r
    if colors is not None and photo_loss:  # implicit_color_reg, tracker.py:699-744
        gi = np.einsum("c,ncj->nj", INTENSITY, np.asarray(color_grad, np.float64)[valid])
        rc = pi - ci
        photo_res = float(np.abs(rc).mean())
        Jc = np.concatenate([np.cross(p, gi), gi], -1)
        N = N + photo_weight * (Jc.T @ (w[:, None] * Jc))
        b = b + photo_weight * (-(Jc * w[:, None]).T @ rc)
    N_raw = N.copy()
    N = N + lm_lambda * np.diag(np.diag(N))
    t = np.linalg.solve(N, b)
    T = np.eye(4)
    T[:3, :3] = expmap(t[:3])
    T[:3, 3] = t[3:]
    return dict(T=T, valid_count=n, residual_cm=float(np.abs(r).mean() * 100.0), N=N_raw,
                g=b, valid=valid, weight=w, photo_residual=photo_res)


def transform_points(points, T):
    """utils/tools.py:534-553 transform_torch: float32 points x float32(T)."""
    T32 = np.asarray(T).astype(F32)
    ph = np.concatenate([np.asarray(points, F32), np.ones((len(points), 1), F32)], 1)
    return (ph @ T32.T)[:, :3].astype(F32)


# --------------------------------------------------------------------------- K6 / K7
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def bce_with_logits(x, y, weight=None):
    """torch.nn.BCEWithLogitsLoss(reduction='mean') as used by utils/loss.py:45-63."""
    l = np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))
    if weight is not None:
        l = l * weight
    return l.mean()


def mlp_backward(acts, params, dout):
    """Backprop d loss/d out [...,1] through Decoder.mlp -> (dz, flat param grad)."""
    Ws, bs, Wo, bo = params
    h = acts[-1].reshape(-1, acts[-1].shape[-1])
    d = dout.reshape(-1, dout.shape[-1])
    gWo, gbo = d.T @ h, d.sum(0)
    dh = d @ Wo
    gW, gb = [None] * len(Ws), [None] * len(Ws)
    for li in range(len(Ws) - 1, -1, -1):
        dh = dh * (acts[li + 1].reshape(-1, acts[li + 1].shape[-1]) > 0)
        x = acts[li].reshape(-1, acts[li].shape[-1])
        gW[li], gb[li] = dh.T @ x, dh.sum(0)
        dh = dh @ Ws[li]
    flat = []
    for a, b in zip(gW, gb):
        flat += [a.ravel(), b.ravel()]
    flat += [gWo.ravel(), gbo.ravel()]
    return dh.reshape(acts[0].shape), np.concatenate(flat)


def mlp_tangent_backward(acts, params, t0, dout):
    """Gradient of sum(dout * D_t0 out) w.r.t. the decoder parameters, where D_t0 out is the directional derivative
    of Decoder.mlp's output along the input direction t0 at the point the forward pass `acts` was taken: what
    autograd's double backward (tools.py:247-260 with create_graph=True, then mapper.py:817) gives for a
    Linear+ReLU stack -- the ReLU patterns are constants almost everywhere, so the derivative network is the
    same weights without biases behind fixed masks."""
    Ws, bs, Wo, bo = params
    masks = [(acts[li + 1].reshape(-1, acts[li + 1].shape[-1]) > 0) for li in range(len(Ws))]
    t = t0.reshape(-1, t0.shape[-1])
    ts = [t]
    for li in range(len(Ws)):
        t = (t @ Ws[li].T) * masks[li]
        ts.append(t)
    d = dout.reshape(-1, dout.shape[-1])
    gWo = d.T @ ts[-1]
    dh = d @ Wo
    flat_w = [None] * len(Ws)
    for li in range(len(Ws) - 1, -1, -1):
        dh = dh * masks[li]
        flat_w[li] = dh.T @ ts[li]
        dh = dh @ Ws[li]
    flat = []
    for li, a in enumerate(flat_w):
        flat += [a.ravel(), np.zeros(Ws[li].shape[0], a.dtype)]
    flat += [gWo.ravel(), np.zeros(Wo.shape[0], gWo.dtype)]
    return np.concatenate(flat)


def eikonal_queries(coord, dec, eps, first=0):
    """mapper.py:682-686 + 986-1008: the 6*n_e central-difference query points in the
    reference's concatenation order (x+, x-, y+, y-, z+, z-), each block [n_e,3].
    ``first`` = phase of a shard of a larger batch (0 for the reference's whole batch)."""
    x = np.asarray(coord, F32)[first::dec]
    e = F32(eps)
    offs = [(e, 0, 0), (-e, 0, 0), (0, e, 0), (0, -e, 0), (0, 0, e), (0, 0, -e)]
    return np.concatenate([(x + np.array(o, F32)).astype(F32) for o in offs], 0)


def train_step(coord, sdf_label, sample_weight, searcher, feats, positions, flat_params,
               dec_shape, sdf_scale, k, *, weighted_first=True, dec=10, eps=0.08, weight_e=0.5,
               loss_weight_on=False, ekional=True, dtype=np.float64, eik_first=0, n_main_global=None,
               n_eik_global=None, analytic=False, orientations=None):
    """One iteration of Mapper.mapping (utils/mapper.py:645-817) on a FIXED batch:
    forward (K1-K3), BCE + Eikonal(numerical gradient) loss, backward.

    ``searcher(points) -> dict`` must return :func:`query_feature` output with
    weighted_first=False vectors (geo_feat [N,k,F+3]) for the given points.
    ``eik_first / n_main_global / n_eik_global`` evaluate one SHARD of a larger batch with the
    losses normalised by the global counts: gradients of the shards then add up to
    the whole-batch gradient.  Returns dict(loss, sdf_loss, eik_loss, feat_grad [M+1,F],
    dec_grad [n_param], sdf_pred); with shard arguments the loss terms are the shard's SUMS
    divided by the global counts.

    ``analytic`` (numerical_grad_on: False, config.py:437-439 -> gradient_decimation 1; config/lidar_slam/
    run_livox.yaml:27): the Eikonal term is taken on the autograd gradient of EVERY sample
    (mapper.py:642-643, 677-678: get_gradient(coord, sdf_pred) with create_graph=True) and the loss is
    differentiated through it.  With c = d loss / d g, the scalar c.g is linear in the per-neighbour
    predictions (through the IDW-weight derivative) and in the derivative network along the direction
    d(decoder input)/dq . c (mlp_tangent_backward); the ReLU patterns contribute nothing."""
    T = dtype
    in_dim, hidden, levels = dec_shape
    params = unpack_decoder(np.asarray(flat_params, T), in_dim, hidden, levels)
    F = feats.shape[1]
    feat_grad = np.zeros((feats.shape[0], F), T)
    s = T(sdf_scale)

    def forward(points):
        qf = searcher(points)
        fv = qf["geo_feat"].astype(T)
        valid = qf["knn_idx"] >= 0
        _, _, w = idw_weights(qf["knn_d2"], valid, qf["nn_count"], dtype=T)
        if weighted_first:
            z = (fv * w[..., None]).sum(1)
        else:
            z = fv
        out, acts = mlp_forward(z, params, keep=True)
        x = out[..., 0]
        pred = s * x if weighted_first else (s * x * w).sum(1)
        return dict(qf=qf, w=w, valid=valid, acts=acts, x=x, pred=pred)

    def backward(fw, dpred):
        """dpred = d loss / d pred [N]."""
        if weighted_first:
            dz, gflat = mlp_backward(fw["acts"], params, (dpred * s)[:, None])
            dfeat = fw["w"][..., None] * dz[:, None, :F]
        else:
            dout = (dpred[:, None] * s * fw["w"])[..., None]
            dz, gflat = mlp_backward(fw["acts"], params, dout)
            dfeat = dz[..., :F]
        dfeat = np.where(fw["valid"][..., None], dfeat, 0.0)
        gather = np.where(fw["valid"], fw["qf"]["knn_idx"], 0)
        np.add.at(feat_grad, gather.reshape(-1), dfeat.reshape(-1, F))
        return gflat

    bs = n_main_global or len(coord)
    fw = forward(coord)
    sigma = s
    xl = fw["pred"] / sigma
    y = _sigmoid(np.asarray(sdf_label, T) / sigma)
    wt = np.abs(np.asarray(sample_weight, T)) if loss_weight_on else None
    sdf_loss = bce_with_logits(xl, y, wt) * len(coord) / bs
    dxl = (_sigmoid(xl) - y) / bs
    if wt is not None:
        dxl = dxl * wt
    dec_grad = backward(fw, dxl / sigma)
    eik_loss = 0.0
    if ekional and weight_e > 0 and analytic:
        qf = fw["qf"]
        valid, w = fw["valid"], fw["w"]
        gather = np.where(valid, qf["knn_idx"], 0)
        diff = np.asarray(coord, T)[:, None, :] - positions[gather].astype(T)
        u = np.where(valid, 1.0 / (qf["knn_d2"].astype(T) + T(1e-15)), 0.0)
        u = np.where((qf["nn_count"] == 0)[:, None], T(1e-15), u)
        S = u.sum(1, keepdims=True)
        g_u = np.where(valid[..., None], -2.0 * (u ** 2)[..., None] * diff, 0.0)
        dw = np.where(valid[..., None], g_u / S[..., None] - (u / S ** 2)[..., None] * g_u.sum(1, keepdims=True), 0.0)
        if orientations is not None:
            _, R = quat_rotate(orientations[gather].astype(T), diff)
        else:
            R = np.broadcast_to(np.eye(3, dtype=T), diff.shape[:2] + (3, 3))
        a = mlp_input_jacobian(fw["acts"], params)  # [N,(k,)F+3]
        fv = qf["geo_feat"].astype(T)
        if weighted_first:
            g = s * (np.einsum("nk,nkij,ni->nj", w, R, a[:, F:]) + np.einsum("nk,nkj->nj", np.einsum("nf,nkf->nk", a, fv), dw))
        else:
            g = np.einsum("nk,nkj->nj", s * fw["x"], dw) + s * np.einsum("nk,nkij,nki->nj", w, R, a[..., F:])
        ne = n_eik_global or len(coord)
        nrm = np.linalg.norm(g, axis=-1)
        eik_loss = ((nrm - 1.0) ** 2).sum() / ne
        with np.errstate(invalid="ignore", divide="ignore"):
            c = np.where(nrm[:, None] > 0, g / nrm[:, None], 0.0) * (2.0 * (nrm - 1.0) * weight_e / ne)[:, None]
        cdw = np.einsum("nj,nkj->nk", c, dw)  # c . d w_t / d q
        chat = np.einsum("nkij,nj->nki", R, c)  # d (neighbour vector) / d q . c
        if weighted_first:
            zdot = np.einsum("nk,nkf->nf", cdw, fv)
            zdot[:, F:] += np.einsum("nk,nki->ni", w, chat)
            dec_grad = dec_grad + mlp_tangent_backward(fw["acts"], params, zdot, np.full((len(coord), 1), s, T))
            dfeat = np.where(valid[..., None], s * cdw[..., None] * a[:, None, :F], 0.0)
            np.add.at(feat_grad, gather.reshape(-1), dfeat.reshape(-1, F))
        else:
            dz, gflat = mlp_backward(fw["acts"], params, (s * cdw)[..., None])
            dfeat = np.where(valid[..., None], dz[..., :F], 0.0)
            np.add.at(feat_grad, gather.reshape(-1), dfeat.reshape(-1, F))
            t0 = np.concatenate([np.zeros(chat.shape[:2] + (F,), T), chat], -1)
            dec_grad = dec_grad + gflat + mlp_tangent_backward(fw["acts"], params, t0, (s * w)[..., None])
    elif ekional and weight_e > 0:
        qe = eikonal_queries(coord, dec, eps, eik_first)
        fe = forward(qe)
        ne_local = len(qe) // 6
        ne = n_eik_global or ne_local
        P = fe["pred"].reshape(6, ne_local)
        g = np.stack([(P[0] - P[1]), (P[2] - P[3]), (P[4] - P[5])], -1) / (2 * T(F32(eps)))
        nrm = np.linalg.norm(g, axis=-1)
        eik_loss = ((nrm - 1.0) ** 2).sum() / ne
        with np.errstate(invalid="ignore", divide="ignore"):
            dg = np.where(nrm[:, None] > 0, g / nrm[:, None], 0.0)
        dg = dg * (2.0 * (nrm - 1.0) * weight_e / ne)[:, None] / (2 * T(F32(eps)))
        dP = np.stack([dg[:, 0], -dg[:, 0], dg[:, 1], -dg[:, 1], dg[:, 2], -dg[:, 2]], 0).reshape(-1)
        dec_grad = dec_grad + backward(fe, dP)
    return dict(loss=sdf_loss + weight_e * eik_loss, sdf_loss=sdf_loss, eik_loss=eik_loss,
                feat_grad=feat_grad, dec_grad=dec_grad, sdf_pred=fw["pred"], fw=fw)


def train_color_step(coord, sdf_label, color_label, sample_weight, searcher, cfeats, flat_params, dec_shape, k, *,
                     weighted_first=True, surface_range=0.25, weight_i=1.0, loss_weight_on=False, dtype=np.float64):
    """Colour term of Mapper.mapping (utils/mapper.py:668-675, 804-812; utils/loss.py:31-42):
    weight_i * mean over surface samples (|sdf_label| < surface_sample_range_m) and channels of
    |sigmoid(mlp_color(z_color)) - colour label|, backward to colour features and colour decoder.
    ``searcher(points)`` returns query_feature output over the COLOUR feature table with
    per-neighbour vectors.  Returns dict(loss, feat_grad, dec_grad, color_pred)."""
    T = dtype
    in_dim, hidden, levels, out_dim = dec_shape
    params = unpack_decoder(np.asarray(flat_params, T), in_dim, hidden, levels, out_dim)
    F = cfeats.shape[1]
    feat_grad = np.zeros((cfeats.shape[0], F), T)
    qf = searcher(coord)
    fv = qf["geo_feat"].astype(T)
    valid = qf["knn_idx"] >= 0
    _, _, w = idw_weights(qf["knn_d2"], valid, qf["nn_count"], dtype=T)
    z = (fv * w[..., None]).sum(1) if weighted_first else fv
    out, acts = mlp_forward(z, params, keep=True)
    p = _sigmoid(out)
    pred = p if weighted_first else (p * w[..., None]).sum(1)
    mask = np.abs(np.asarray(sdf_label, T)) < surface_range
    n_s = int(mask.sum())
    diff = pred - np.asarray(color_label, T)
    wt = np.abs(np.asarray(sample_weight, T))[:, None] if loss_weight_on else 1.0
    loss = weight_i * (wt * np.abs(diff))[mask].mean() if n_s else 0.0
    dpred = np.where(mask[:, None], weight_i * wt * np.sign(diff) / max(n_s * out_dim, 1), 0.0)
    if weighted_first:
        dout = dpred * p * (1 - p)
        dz, gflat = mlp_backward(acts, params, dout)
        dfeat = w[..., None] * dz[:, None, :F]
    else:
        dout = dpred[:, None, :] * w[..., None] * p * (1 - p)
        dz, gflat = mlp_backward(acts, params, dout)
        dfeat = dz[..., :F]
    dfeat = np.where(valid[..., None], dfeat, 0.0)
    np.add.at(feat_grad, np.where(valid, qf["knn_idx"], 0).reshape(-1), dfeat.reshape(-1, F))
    return dict(loss=loss, feat_grad=feat_grad, dec_grad=gflat, color_pred=pred)



# --------------------------------------------------------------------------- semantic head
def log_softmax(x):
    """F.log_softmax(x, dim=-1) (model/decoder.py:100-103)."""
    m = x.max(-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))


def sem_label_prob(z, params):
    """Decoder.sem_label_prob (model/decoder.py:100-103): log_softmax(mlp(features)) over the heads."""
    return log_softmax(mlp_forward(z, params))


def sem_select_mask(sem_label, freespace_label_on=False, decimation=1):
    """The samples the semantic loss runs over (utils/mapper.py:786-799): label_mask = label > 0 (>= 0 with
    freespace_label_on), then every `decimation`-th of the masked samples in order (sem_pred[label_mask][::dec])."""
    lab = np.asarray(sem_label)
    mask = lab >= 0 if freespace_label_on else lab > 0
    idx = np.nonzero(mask)[0][::max(1, int(decimation))]
    sel = np.zeros(lab.shape[0], bool)
    sel[idx] = True
    return sel


def query_sem(points, search, feats, positions, params, k, weighted_first=True, global2local=None, orientations=None,
              dtype=np.float64):
    """Tracker.query_source_points / Mesher.query_points with query_sem (utils/tracker.py:336-341, utils/mesher.py:137-145):
    sem_pred = sem_label_prob(feature) -- per neighbour and summed with the IDW weights when not weighted_first -- and its
    argmax.  Returns (sem_pred [N, S], label [N] int64, nn_count)."""
    T = dtype
    qf = query_feature(points, search, feats, positions, None, k, global2local, orientations, weighted_first=False)
    fv = qf["geo_feat"].astype(T)
    valid = qf["knn_idx"] >= 0
    _, _, w = idw_weights(qf["knn_d2"], valid, qf["nn_count"], dtype=T)
    params = tuple([x.astype(T) for x in p] if isinstance(p, list) else p.astype(T) for p in params)
    if weighted_first:
        pred = sem_label_prob((fv * w[..., None]).sum(1), params)
    else:
        pred = (sem_label_prob(fv, params) * w[..., None]).sum(1)
    return pred, pred.argmax(-1), qf["nn_count"]


def train_sem_step(coord, sem_label, searcher, feats, flat_params, dec_shape, k, *, weighted_first=True, weight_s=1.0,
                   decimation=1, freespace_label_on=False, dtype=np.float64):
    """Semantic term of Mapper.mapping (utils/mapper.py:664-667, 782-800): sem_pred = sem_label_prob(geo_feature)
    (x weight_knn summed over the neighbours when not weighted_first), torch.nn.NLLLoss(mean) over the selected samples
    (sem_select_mask), times weight_s; backward to the GEOMETRY features and the semantic decoder.  ``searcher(points)``
    returns query_feature output over the geometry feature table with per-neighbour vectors.
    Returns dict(loss (weight_s NOT applied, the reference's sem_loss), feat_grad, dec_grad, sem_pred, selected)."""
    T = dtype
    in_dim, hidden, levels, out_dim = dec_shape
    params = unpack_decoder(np.asarray(flat_params, T), in_dim, hidden, levels, out_dim)
    F = feats.shape[1]
    feat_grad = np.zeros((feats.shape[0], F), T)
    qf = searcher(coord)
    fv = qf["geo_feat"].astype(T)
    valid = qf["knn_idx"] >= 0
    _, _, w = idw_weights(qf["knn_d2"], valid, qf["nn_count"], dtype=T)
    z = (fv * w[..., None]).sum(1) if weighted_first else fv
    out, acts = mlp_forward(z, params, keep=True)
    lp = log_softmax(out)                                     # [N, S] or [N, k, S]
    pred = lp if weighted_first else (lp * w[..., None]).sum(1)
    sel = sem_select_mask(sem_label, freespace_label_on, decimation)
    lab = np.asarray(sem_label).astype(np.int64)
    n_s = int(sel.sum())
    rows = np.nonzero(sel)[0]
    loss = float(-pred[rows, lab[rows]].mean()) if n_s else 0.0
    onehot = np.zeros(pred.shape, T)
    onehot[rows, lab[rows]] = 1.0
    coef = np.where(sel, weight_s / max(n_s, 1), 0.0).astype(T)    # d (weight_s * mean NLL) / d (-pred[i, label_i])
    if weighted_first:
        dout = coef[:, None] * (np.exp(lp) - onehot)               # d / d logits: softmax - onehot
        dz, gflat = mlp_backward(acts, params, dout)
        dfeat = w[..., None] * dz[:, None, :F]
    else:
        dout = coef[:, None, None] * w[..., None] * (np.exp(lp) - onehot[:, None, :])
        dz, gflat = mlp_backward(acts, params, dout)
        dfeat = dz[..., :F]
    dfeat = np.where(valid[..., None], dfeat, 0.0)
    np.add.at(feat_grad, np.where(valid, qf["knn_idx"], 0).reshape(-1), dfeat.reshape(-1, F))
    return dict(loss=loss, feat_grad=feat_grad, dec_grad=gflat, sem_pred=pred, selected=sel)


def adam_step(p, g, m, v, step, lr=0.01, b1=0.9, b2=0.99, eps=1e-15):
    """torch.optim.Adam (no amsgrad, no weight decay) as configured by
    utils/tools.py:198-199 (betas (0.9, 0.99), eps = adam_eps)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


# --------------------------------------------------------------------------- K8 / K9
def voxel_down_sample(points, voxel_size):
    """utils/tools.py:583-626 voxel_down_sample_torch: index of the point closest to the
    voxel centre per voxel (distance quantised to 1000 levels, smallest index wins ties);
    result ordered by ascending linearised voxel id like torch.unique."""
    points = np.asarray(points, F32)
    vs = F32(voxel_size)
    offset = np.floor(points.min(0) / vs).astype(np.int64)
    grid = np.floor(points / vs)
    center = ((grid + F32(0.5)) * vs).astype(F32)
    dd = (points - center) ** 2
    dist = ((dd[:, 0] + dd[:, 1]).astype(F32) + dd[:, 2]).astype(F32) ** F32(0.5)
    dist = (dist / dist.max() * F32(999)).astype(np.int64)
    gi = grid.astype(np.int64) - offset
    vsz = gi.max()
    gid = gi[:, 0] + gi[:, 1] * vsz + gi[:, 2] * vsz * vsz
    _, inverse = np.unique(gid, return_inverse=True)
    n = len(points)
    off = 10 ** len(str(n - 1))
    key = np.arange(n, dtype=np.int64) + dist * off
    out = np.full(inverse.max() + 1, np.iinfo(np.int64).max, np.int64)
    np.minimum.at(out, inverse, key)
    return out % off


def map_update(state, points, cur_ts, resolution, travel_dist=None, diff_travel_dist_local=None,
               reboot_ts=0, temporal=True):
    """NeuralPoints.update (neural_points.py:311-416) on a dict state with keys
    table [B] int64, positions [P,3], ts_create [P], ts_update [P].  Appends new points
    (features/certainties are initialised by the caller) and returns the added points.
    Duplicate hash slots inside one call resolve 'last writer wins' in sample order, as the
    CPU reference's sequential index_put_ does (neural_points.py:377)."""
    points = np.asarray(points, F32)
    sel = voxel_down_sample(points, resolution)
    sp = points[sel]
    B = state["table"].shape[0]
    slot = hash_slots(grid_coords(sp, resolution), B)
    hidx = state["table"][slot]
    P = state["positions"].shape[0]
    if P > 0 and cur_ts != reboot_ts:
        d2 = _d2(state["positions"][hidx], sp)
        mask = (hidx == -1) | (d2 > F32(3 * resolution ** 2))
        if temporal:
            td = np.asarray(travel_dist, F32)
            mask |= (td[cur_ts] - td[state["ts_update"][hidx]]) > F32(diff_travel_dist_local)
    else:
        mask = np.ones(len(sp), bool)
    added = sp[mask]
    cur = hidx.copy()
    cur[mask] = np.arange(len(added), dtype=np.int64) + P
    state["table"][slot] = cur
    state["positions"] = np.concatenate([state["positions"], added], 0)
    ts = np.full(len(added), cur_ts, np.int32)
    state["ts_create"] = np.concatenate([state["ts_create"], ts])
    state["ts_update"] = np.concatenate([state["ts_update"], ts])
    return added


def mid_ts(ts_create, ts_update):
    """((point_ts_create + point_ts_update) / 2).int() (neural_points.py:443-447, 803-806, 845-848): true division in
    float32, truncated."""
    return ((np.asarray(ts_create, np.int32) + np.asarray(ts_update, np.int32)).astype(F32) / F32(2)).astype(np.int32)


def local_map_mask(positions, ts_used, sensor_position, radius, travel_dist=None, cur_ts=0,
                   diff_travel_dist_local=None, reboot_ts=None, diff_ts_local=None):
    """NeuralPoints.reset_local_map (neural_points.py:448-507): (local_mask [P] bool,
    global2local [P+1] int64).  travel_dist given: travel-distance window (use_travel_dist, :451-455);
    diff_ts_local given instead: window of frames (:456-458); neither: no time mask.  The radius test runs
    in the dtype of ``sensor_position`` (float32 points - float64 position promotes, :475-479).

    Reference quirk, reproduced on purpose: ``torch.full_like(local_mask, -1).long()``
    (neural_points.py:498) is taken of a *bool* tensor, so the fill value is True -> 1,
    i.e. NON-LOCAL points map to local index 1, not -1; only the padding entry is -1
    (neural_points.py:505)."""
    P = positions.shape[0]
    ts_used = np.asarray(ts_used)
    if travel_dist is not None or diff_ts_local is not None:
        if travel_dist is not None:
            td = np.asarray(travel_dist, F32)
            tm = np.abs(td[cur_ts] - td[ts_used]) < F32(diff_travel_dist_local)
        else:
            tm = np.abs(int(cur_ts) - ts_used.astype(np.int64)) < int(diff_ts_local)
        if reboot_ts is not None:
            tm &= ts_used >= reboot_ts
        if tm.sum() < 100:
            tm = np.ones(P, bool)
    else:
        tm = np.ones(P, bool)
    sp = np.asarray(sensor_position)
    if sp.dtype == np.float64:
        d = (np.asarray(positions, np.float64) - sp) ** 2
        near = ((d[:, 0] + d[:, 1]) + d[:, 2]) < float(radius) ** 2
    else:
        d = (positions - np.asarray(sensor_position, F32)).astype(F32) ** 2
        dist2 = ((d[:, 0] + d[:, 1]).astype(F32) + d[:, 2]).astype(F32)
        near = dist2 < F32(radius ** 2)
    mask = tm & near
    g2l = np.full(P + 1, 1, np.int64)
    g2l[:P][mask] = np.arange(int(mask.sum()))
    g2l[P] = -1
    return mask, g2l


# --------------------------------------------------------------------------- K12-K14: process_frame data path
def sample_rays(points, colors, rnd_surface, rnd_front, rnd_behind, *, surface_range, surface_n, front_n, behind_n,
                free_begin_ratio, free_end_dist, dist_weight_on=True, dist_weight_scale=0.8, max_range=60.0,
                behind_dropoff_on=False):
    """DataSampler.sample (utils/data_sampler.py:18-260) with the three random draws as inputs
    (randn [N*surface_n], rand [N*front_n], rand [N*behind_n], in the reference's order of
    generation).  float32 throughout, operation order of the reference.
    Returns (coord [N*A,3], sdf_label [N*A], color [N*A,C] or None, weight [N*A]) in the
    reference's ray-wise order (sample j of ray i at i*A + j; j = measured, surface.., front.., behind..)."""
    p = np.asarray(points, F32)
    N = p.shape[0]
    A = surface_n + front_n + behind_n + 1
    # torch.linalg.norm on CPU accumulates with fused multiply-adds: fma(z,z, fma(y,y, x*x))
    x, y, z = (p[:, i].astype(np.float64) for i in range(3))
    acc = (x * x).astype(F32).astype(np.float64)
    acc = (y * y + acc).astype(F32).astype(np.float64)
    acc = (z * z + acc).astype(F32)
    dist = np.sqrt(acc).astype(F32)[:, None]
    sr = F32(surface_range)
    half = F32(2.0 * surface_range)  # sigma_ratio * surface_sample_range (python float, then cast)
    disp_s = (np.asarray(rnd_surface, F32).reshape(-1, 1) * sr).astype(F32)
    rep = np.tile(dist, (surface_n, 1))
    ratio_s = (disp_s / rep + F32(1.0)).astype(F32)
    rep = np.tile(dist, (front_n, 1))
    fmax = (F32(1.0) - half / rep).astype(F32)
    fdiff = (fmax - F32(free_begin_ratio)).astype(F32)
    ratio_f = (np.asarray(rnd_front, F32).reshape(-1, 1) * fdiff + F32(free_begin_ratio)).astype(F32)
    disp_f = ((ratio_f - F32(1.0)) * rep).astype(F32)
    rep = np.tile(dist, (behind_n, 1))
    bmax = (F32(free_end_dist) / rep + F32(1.0)).astype(F32)
    bmin = (F32(1.0) + half / rep).astype(F32)
    bdiff = (bmax - bmin).astype(F32)
    ratio_b = (np.asarray(rnd_behind, F32).reshape(-1, 1) * bdiff + bmin).astype(F32)
    disp_b = ((ratio_b - F32(1.0)) * rep).astype(F32)
    disp = np.concatenate([np.zeros_like(dist), disp_s, disp_f, disp_b], 0)
    ratio = np.concatenate([np.ones_like(dist), ratio_s, ratio_f, ratio_b], 0)
    rep_d = np.tile(dist, (A, 1))
    pts = (np.tile(p, (A, 1)) * ratio).astype(F32)
    w = np.ones_like(rep_d)
    ns = N * (surface_n + 1)
    if dist_weight_on:
        w[:ns] = (F32(1 + dist_weight_scale * 0.5) - (rep_d[:ns] / F32(max_range)) * F32(dist_weight_scale)).astype(F32)
    if behind_dropoff_on:
        dmin, dmax = 0.2 * free_end_dist, free_end_dist
        dw = ((F32(dmax) - disp) / F32(dmax - dmin)).astype(F32)
        dw = np.clip(dw, F32(0.0), F32(1.0))
        dw = (dw * F32(0.8) + F32(0.2)).astype(F32)
        w = (w * dw).astype(F32)
    w[ns:] *= F32(-1.0)
    coord = pts.reshape(A, N, 3).transpose(1, 0, 2).reshape(-1, 3)
    label = (-disp[:, 0]).reshape(A, N).T.reshape(-1)
    weight = w[:, 0].reshape(A, N).T.reshape(-1)
    color = None
    if colors is not None:
        c = np.asarray(colors, F32)
        C = c.shape[1]
        call = np.concatenate([np.tile(c, (surface_n + 1, 1)), np.zeros((N * (front_n + behind_n), C), F32)], 0)
        color = call.reshape(A, N, C).transpose(1, 0, 2).reshape(-1, C)
    return np.ascontiguousarray(coord), np.ascontiguousarray(label), color, np.ascontiguousarray(weight)



def sample_sem_labels(labels, surface_n, front_n, behind_n):
    """Semantic labels of DataSampler.sample's output rows (utils/data_sampler.py:59-62, 83-84, 105-106, 184-194 and the final
    point-major reshape :104-107): the measured point and its surface_n close-to-surface samples carry the point's label,
    the free-space samples label 0; row i * A + j belongs to point i."""
    lab = np.asarray(labels, np.int32)
    A = 1 + surface_n + front_n + behind_n
    out = np.zeros((lab.shape[0], A), np.int32)
    out[:, :1 + surface_n] = lab[:, None]
    return out.reshape(-1)

def pool_filter_mask(global_coord, origin, window_radius, pool_capacity=None, discard_index=None):
    """Distance window + random discard of Mapper.process_frame (utils/mapper.py:303-323).  The
    subtraction promotes to float64 (float32 pool - float64 pose column).  `discard_index` are
    the reference's torch.randint draws (indices into the kept list); they are only applied
    when more than pool_capacity samples survive the window."""
    rel = np.asarray(global_coord, F32).astype(np.float64) - np.asarray(origin, np.float64)[None, :]
    d2 = (rel[:, 0] ** 2 + rel[:, 1] ** 2) + rel[:, 2] ** 2
    mask = d2 < float(window_radius) ** 2
    kept = np.nonzero(mask)[0]
    if pool_capacity is not None and len(kept) > pool_capacity:
        assert discard_index is not None and len(discard_index) == len(kept) - pool_capacity
        mask[kept[np.asarray(discard_index, np.int64)]] = False
    return mask


def query_certainty(points, table, positions, certainties, resolution):
    """NeuralPoints.query_certainty (neural_points.py:1011-1033) under the search neighbourhood
    process_frame sets for it (num_nei_cells=1, search_alpha=0: the query's own cell only,
    mapper.py:385-387): max over candidates of the (global) certainty, 0 where invalid."""
    dx, mv = search_neighborhood(1, 0.0, resolution)
    _, idx = radius_search(points, table, positions, resolution, dx, mv)
    cert = np.asarray(certainties, F32)[idx]
    cert = np.where(idx < 0, F32(0.0), cert)
    return cert.max(axis=-1).astype(F32)


def new_sample_index(certainty, sdf_label, new_certainty_thre, surface_range, offset=0):
    """mapper.py:405-416: close-to-surface samples of the current frame in not-yet-certain cells."""
    sel = (np.asarray(certainty, F32) < F32(new_certainty_thre)) & (np.abs(np.asarray(sdf_label, F32)) < F32(surface_range * 3.0))
    return np.nonzero(sel)[0].astype(np.int64) + int(offset)


def adaptive_iter_offset(new_sample_count, cur_sample_count, frame_id, *, adaptive_iters, ratio_less=0.02,
                         ratio_more=0.15, ratio_restart=0.3, freeze_after_frame=40):
    """mapper.py:424-439."""
    if not adaptive_iters:
        return 0
    r = new_sample_count / cur_sample_count
    if r < ratio_less:
        return -5
    if r > ratio_more:
        return 10 if (frame_id > freeze_after_frame and r > ratio_restart) else 5
    return 0


# --------------------------------------------------------------------------- preprocess_frame data path
def _norm3(p):
    """torch.norm / torch.linalg.norm over 3 columns on CPU: fma(z,z, fma(y,y, x*x)), then sqrt."""
    x, y, z = (np.asarray(p[:, i], F32).astype(np.float64) for i in range(3))
    acc = (x * x).astype(F32).astype(np.float64)
    acc = (y * y + acc).astype(F32).astype(np.float64)
    acc = (z * z + acc).astype(F32)
    return np.sqrt(acc).astype(F32)


def crop_frame_mask(points, min_z, max_z, min_range, max_range):
    """crop_frame (dataset/slam_dataset.py:1229-1247): rows kept (order preserved)."""
    p = np.asarray(points, F32)
    dist = _norm3(p)
    return (dist > F32(min_range)) & (dist < F32(max_range)) & (p[:, 2] > F32(min_z)) & (p[:, 2] < F32(max_z))


def intrinsic_correct(points, correct_deg):
    """intrinsic_correct (dataset/slam_dataset.py:1251-1269), float32 like the reference."""
    p = np.array(points, F32, copy=True)
    if correct_deg == 0.0:
        return p
    dist = _norm3(p)
    ang = F32(correct_deg / 180.0 * np.pi)
    v = np.arcsin((p[:, 2] / dist).astype(F32)).astype(F32)
    vc = (v + ang).astype(F32)
    hs = (np.cos(vc).astype(F32) / np.cos(v).astype(F32)).astype(F32)
    p[:, 0] *= hs
    p[:, 1] *= hs
    p[:, 2] = dist * np.sin(vc).astype(F32)
    return p


def rotvec_of(R):
    """log map of a rotation matrix (float64), shortest arc."""
    R = np.asarray(R, np.float64)
    c = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    th = np.arccos(c)
    if th < 1e-12:
        return np.zeros(3)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2.0 * np.sin(th))
    return w * th


def deskewing(points, ts, pose, ts_mid_pose=0.5):
    """deskewing (utils/tools.py:747-779): per-point motion undistortion with the relative pose
    T_last<-cur.  roma.rotmat_slerp(I, R, t) = exp(t * log(R)) (roma's unit-quaternion slerp with
    shortest_arc; t may be negative), evaluated here in float64 and applied in float32."""
    p = np.asarray(points, F32)
    t = np.asarray(ts, F32).reshape(-1)
    t = ((t - t.min()) / (t.max() - t.min())).astype(F32)
    t = (t - F32(ts_mid_pose)).astype(F32)
    T = np.asarray(pose, F32)
    rv = rotvec_of(T[:3, :3].astype(np.float64))
    th = np.linalg.norm(rv)
    ax = rv / th if th > 0 else np.array([1.0, 0.0, 0.0])
    a = t.astype(np.float64) * th
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rm = np.eye(3)[None] + np.sin(a)[:, None, None] * K[None] + (1 - np.cos(a))[:, None, None] * (K @ K)[None]
    out = np.einsum("nij,nj->ni", Rm.astype(F32), p[:, :3]).astype(F32) + (t[:, None] * T[:3, 3][None, :]).astype(F32)
    return out.astype(F32)


# --------------------------------------------------------------------------- Mesher.query_points (utils/mesher.py:40-164)
def mesher_query(points, search, feats, positions, params, sdf_scale, k, weighted_first=True, global2local=None,
                 mask_min_nn_count=4):
    """Forward-only bulk query: (sdf [N] with 0 where no neighbour was found, marching-cubes mask [N])."""
    sdf, _, _, nn, _ = query_sdf(points, search, feats, positions, params, sdf_scale, k, weighted_first=weighted_first,
                                 global2local=global2local, with_grad=False, dtype=np.float32)
    sdf = np.where(nn >= 1, sdf, 0.0).astype(F32)
    return sdf, nn >= mask_min_nn_count


# --------------------------------------------------------------------------- post-loop map maintenance
def transform_batch(points, T):
    """transform_batch_torch (utils/tools.py:556-580): float32 bmm R_i p_i + t_i."""
    p = np.asarray(points, F32)
    T = np.asarray(T).astype(F32)
    return (np.einsum("nij,nj->ni", T[:, :3, :3], p).astype(F32) + T[:, :3, 3]).astype(F32)


def rotmat_to_quat(Rm):
    """utils/tools.py:441-456 (w, x, y, z), float32, no normalisation."""
    Rm = np.asarray(Rm, F32)
    qw = (np.sqrt(F32(1.0) + Rm[:, 0, 0] + Rm[:, 1, 1] + Rm[:, 2, 2]) / F32(2.0)).astype(F32)
    q4 = (F32(4.0) * qw).astype(F32)
    return np.stack([qw, (Rm[:, 2, 1] - Rm[:, 1, 2]) / q4, (Rm[:, 0, 2] - Rm[:, 2, 0]) / q4, (Rm[:, 1, 0] - Rm[:, 0, 1]) / q4], 1).astype(F32)


def quat_multiply(q1, q2):
    """utils/tools.py:499-514."""
    w1, x1, y1, z1 = (np.asarray(q1, F32)[:, i] for i in range(4))
    w2, x2, y2, z2 = (np.asarray(q2, F32)[:, i] for i in range(4))
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], 1).astype(F32)


def adjust_map(positions, orientations, ts_create, pose_diff, ts_update=None):
    """NeuralPoints.adjust_map (neural_points.py:791-817): per-point SE(3) by creation frame (ts_update given:
    config.use_mid_ts, by the mid timestamp)."""
    used = np.asarray(ts_create, np.int64) if ts_update is None else mid_ts(ts_create, ts_update).astype(np.int64)
    pd = np.asarray(pose_diff)
    pos = transform_batch(positions, pd[used])
    dq = rotmat_to_quat(pd[:, :3, :3].astype(F32))
    return pos, quat_multiply(dq[used], orientations)


def voxel_down_sample_min_value(points, voxel_size, value):
    """utils/tools.py:629-668: per voxel the index of the point with the smallest value (1000 levels, lowest index on ties)."""
    points = np.asarray(points, F32)
    vs = F32(voxel_size)
    offset = np.floor(points.min(0) / vs).astype(np.int64)
    gi = np.floor(points / vs).astype(np.int64) - offset
    vsz = gi.max()
    gid = gi[:, 0] + gi[:, 1] * vsz + gi[:, 2] * vsz * vsz
    _, inverse = np.unique(gid, return_inverse=True)
    n = len(points)
    off = 10 ** len(str(n - 1))
    v = np.asarray(value, F32)
    q = (v / v.max() * F32(999)).astype(np.int64)
    key = np.arange(n, dtype=np.int64) + q * off
    out = np.full(inverse.max() + 1, np.iinfo(np.int64).max, np.int64)
    np.minimum.at(out, inverse, key)
    return out % off


def recreate_hash(positions, ts_create, cur_ts, resolution, buffer_size, certainties=None, with_ts=True, ts_update=None):
    """NeuralPoints.recreate_hash(kept_points=True) (neural_points.py:819-870): the rebuilt table (ts_update given:
    config.use_mid_ts)."""
    if with_ts:
        ts = np.asarray(ts_create, np.int64) if ts_update is None else mid_ts(ts_create, ts_update).astype(np.int64)
        value = np.abs(ts - cur_ts).astype(F32)
    else:
        c = np.asarray(certainties, F32)
        value = (c.max() - c).astype(F32)
    sel = voxel_down_sample_min_value(positions, resolution, value)
    table = np.full(buffer_size, -1, np.int64)
    table[hash_slots(grid_coords(np.asarray(positions, F32)[sel], resolution), buffer_size)] = sel
    return table, sel


def merge_map(arrays, cur_ts, resolution, buffer_size, with_ts=False):
    """NeuralPoints.recreate_hash(kept_points=False) (neural_points.py:872-898): the map merged down to one point per
    voxel.  arrays = dict(positions, orientations, ts_create, ts_update, certainties, geo_features [P+1, F]); returns
    (merged dict in the same keys, table over the NEW indices)."""
    _, sel = recreate_hash(arrays["positions"], arrays["ts_create"], cur_ts, resolution, buffer_size,
                           certainties=arrays["certainties"], with_ts=with_ts)
    out = {k: np.asarray(arrays[k])[sel] for k in ("positions", "orientations", "ts_create", "ts_update", "certainties")}
    out["geo_features"] = np.asarray(arrays["geo_features"])[np.concatenate([sel, [-1]])]
    table = np.full(buffer_size, -1, np.int64)
    table[hash_slots(grid_coords(out["positions"], resolution), buffer_size)] = np.arange(len(sel))
    return out, table


def prune_mask(certainties, ts_update, travel_dist, cur_ts, diff_travel_dist_local, thre, global_prune=False):
    """NeuralPoints.prune_map (neural_points.py:748-789): True = pruned."""
    m = np.asarray(certainties, F32) < F32(thre)
    if not global_prune:
        td = np.asarray(travel_dist, F32)
        m &= np.abs(td[cur_ts] - td[np.asarray(ts_update, np.int64)]) > F32(diff_travel_dist_local)
    return m
