"""Batched Opt_with_points / LS_sol (train.py:59-110) in torch, for thousands of ES candidates.

The reference fits every candidate's ETG weights on the host with two <=1000-step gradient
descents (train.py:100-104); at 4096+ candidates per generation that dwarfs the rollout, so
the same iteration is run for all candidates at once (float64, per-candidate stopping rule
identical to LS_sol's `while err > precision and i < 1000`).  Works on any torch device.
"""
import numpy as np
import torch

from .etg import control_times


def ls_sol_batched(A, b, precision=1e-4, alpha=0.05, lamb=1.0, w0=None, max_iter=1000):
    """A [m,n] shared, b [B,m]; returns x [B,n] -- per-row identical to LS_sol(A, b_i, ...)."""
    A = A.to(torch.float64)
    b = b.to(torch.float64)
    Bsz, n = b.shape[0], A.shape[1]
    x = torch.zeros(Bsz, n, dtype=torch.float64, device=b.device) if w0 is None else \
        w0.to(torch.float64).expand(Bsz, n).clone()
    anchor = None if w0 is None else x.clone()
    AtA = A.T @ A
    Atb = b @ A                      # [B,n]
    active = torch.ones(Bsz, dtype=torch.bool, device=b.device)
    for _ in range(max_iter):
        r = x @ A.T - b
        active = active & ((r * r).sum(1) > precision)
        if not bool(active.any()):
            break
        g = x @ AtA - Atb
        if anchor is not None:
            g = g + lamb * (x - anchor)
        x = torch.where(active[:, None], x - alpha * g, x)
    return x


def opt_with_points_batched(ETG, ETG_T, points, b0, w0, precision=1e-4, lamb=0.5, device="cpu"):
    """points [B,6,2] (prior + candidate offsets), b0 [3], w0 [3,20] -> (w [B,3,20], b [B,3]).
    Same as calling Opt_with_points(ETG, ETG_T, points=points[i], b0=b0, w0=w0) for every i."""
    # the 6 x 20 feature matrix depends on the layer and the period alone: built and uploaded once per device
    cache = ETG.__dict__.setdefault("_fit_feats", {})
    # (keyed by everything the matrix depends on: a layer whose dt / sigma / amplitude / phase is changed after a fit gets a new one)
    key = (float(ETG_T), str(device)) + tuple(repr(np.asarray(getattr(ETG, a, None)).tolist()) for a in
                                              ("dt", "T", "H", "sigma_sq", "amp", "phase", "omega", "u"))
    if key not in cache:
        cache[key] = torch.as_tensor(np.array([ETG.update(t) for t in control_times(ETG_T)]), dtype=torch.float64, device=device)
    else:
        ETG.t += ETG.dt * len(control_times(ETG_T))       # (what the update() calls of the uncached path do to the layer's clock)
    feats = cache[key]
    pts = torch.as_tensor(points, dtype=torch.float64, device=device)
    if pts.is_cuda:
        return _fit_hip(pts.contiguous(), feats.contiguous(), b0, w0, precision, lamb)
    b0 = np.asarray(b0, dtype=np.float64)
    b = torch.tensor([b0[0], b0[-1]], dtype=torch.float64, device=device)
    centred = pts - b
    w0t = torch.as_tensor(np.asarray(w0), dtype=torch.float64, device=device)
    x1 = ls_sol_batched(feats, centred[:, :, 0], precision, 0.05, lamb, w0t[0][None])
    x2 = ls_sol_batched(feats, centred[:, :, 1], precision, 0.05, lamb, w0t[-1][None])
    Bsz = pts.shape[0]
    w = torch.zeros(Bsz, 3, feats.shape[1], dtype=torch.float64, device=device)
    w[:, 0], w[:, 2] = x1, x2
    bb = torch.zeros(Bsz, 3, dtype=torch.float64, device=device)
    bb[:, 0], bb[:, 2] = b[0], b[1]
    return w, bb


def _fit_hip(pts, feats, b0, w0, precision, lamb, alpha=0.05, max_iter=1000):
    """The HIP kernel behind the C-ABI (csrc/etg_fit.hip: one lane per candidate-dimension)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    dev = pts.device
    b0 = np.asarray(b0, dtype=np.float64)
    w0x = torch.as_tensor(np.asarray(w0), dtype=torch.float64, device=dev)
    w0t = torch.stack([w0x[0], w0x[-1]]).contiguous()
    nb = pts.shape[0]
    w = torch.empty(nb, 3, feats.shape[1], dtype=torch.float64, device=dev)
    b = torch.empty(nb, 3, dtype=torch.float64, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.etg_fit_etg(C.c_void_p(pts.data_ptr()), nb, C.c_void_p(feats.data_ptr()), C.c_void_p(w0t.data_ptr()),
                               float(b0[0]), float(b0[-1]), float(precision), float(alpha), float(lamb), int(max_iter),
                               C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), stream))
    return w, b
