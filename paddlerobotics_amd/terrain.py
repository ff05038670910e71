"""Terrain tasks as heightfields.

The reference picks its task by name -- `--task_mode stairstair` by default (train.py:462), with the
per-episode geometry drawn from STEP_HEIGHT = arange(0.08, 0.101, 0.002), STEP_WIDTH = arange(0.26,
0.401, 0.02), SLOPE = arange(0.2, 0.401, 0.02) (train.py:48-50) -- but the task geometry itself lives in
rlschool, which is absent.  The definitions below are therefore THIS repo's (parity unpinned): every task
is a height profile along +x (the walking direction), rasterised into the bilinear heightfield the kernels
already support.  Several geometry variants are stacked as "bands" of one heightfield (EtgConfig.hf_bands):
robot e walks on band e % bands, so one batch covers the whole STEP_HEIGHT x STEP_WIDTH x SLOPE range.

    x < 1.0            flat approach (the robot resets at x = 0)
    up section         `n_steps` stairs of (width, height), or a ramp of gradient `slope` reaching the same top
    plateau (1 m)
    down section       stairs or ramp back to z = 0
    flat run-out
"""
import numpy as np

STEP_HEIGHT = np.arange(0.08, 0.101, 0.002)   # train.py:48
SLOPE = np.arange(0.2, 0.401, 0.02)           # train.py:49
STEP_WIDTH = np.arange(0.26, 0.401, 0.02)     # train.py:50

TERRAIN_TASKS = ("stairstair", "stairslope", "slopestair", "slopeslope", "balancebeam", "rough")


def _stairs_up(x, x0, width, height, n):
    """height reached at x on n stairs starting at x0 (riser at the START of each tread)."""
    k = np.floor((x - x0) / width) + 1.0
    return height * np.clip(k, 0.0, float(n))


def profile(task, x, step_height=0.09, step_width=0.3, slope=0.3, n_steps=5, approach=1.0, plateau=1.0):
    """z(x) of one terrain variant; `task` = <up><down> with up/down in {stair, slope}."""
    x = np.asarray(x, dtype=np.float64)
    top = n_steps * step_height
    kinds = {"stairstair": ("stair", "stair"), "stairslope": ("stair", "slope"),
             "slopestair": ("slope", "stair"), "slopeslope": ("slope", "slope")}[task]
    len_up = n_steps * step_width if kinds[0] == "stair" else top / slope
    len_dn = n_steps * step_width if kinds[1] == "stair" else top / slope
    x1 = approach                      # start of the up section
    x2 = x1 + len_up                   # plateau
    x3 = x2 + plateau                  # start of the down section
    if kinds[0] == "stair":
        up = _stairs_up(x, x1, step_width, step_height, n_steps)
    else:
        up = np.clip((x - x1) * slope, 0.0, top)
    if kinds[1] == "stair":
        dn = top - _stairs_up(x, x3, step_width, step_height, n_steps)
    else:
        dn = np.clip(top - (x - x3) * slope, 0.0, top)
    z = np.where(x < x2, up, np.where(x < x3, top, dn))
    return z, x3 + len_dn


def make_task_heightfield(task, variants=16, seed=0, cell=0.02, half_width=1.0, x_min=-1.0, run_out=2.0,
                          n_steps=5, beam_width=0.3, beam_drop=0.5, rough_height=0.05):
    """-> dict(heights [variants*rows, nx] float32, cell, origin (x0, y0), bands, params [variants, 3]).

    params rows are (step_height, step_width, slope) of each band, drawn from the reference's ranges with
    numpy.random.default_rng(seed)."""
    if task not in TERRAIN_TASKS:
        raise ValueError("unknown terrain task %r (have %s)" % (task, ", ".join(TERRAIN_TASKS)))
    rng = np.random.default_rng(seed)
    params = np.stack([rng.choice(STEP_HEIGHT, variants), rng.choice(STEP_WIDTH, variants),
                       rng.choice(SLOPE, variants)], axis=1)
    rows = int(round(2 * half_width / cell)) + 1
    ys = -half_width + cell * np.arange(rows)
    if task in ("balancebeam", "rough"):
        x_max = 8.0
    else:
        x_max = max(profile(task, np.zeros(1), p[0], p[1], p[2], n_steps)[1] for p in params) + run_out
    nx = int(np.ceil((x_max - x_min) / cell)) + 1
    xs = x_min + cell * np.arange(nx)
    H = np.zeros((variants, rows, nx), dtype=np.float32)
    for v in range(variants):
        if task == "balancebeam":
            # a beam of the given width along +x from x = 1: off the beam the ground drops away
            on = (np.abs(ys)[:, None] <= beam_width / 2) | (xs[None, :] < 1.0)
            H[v] = np.where(on, 0.0, -beam_drop)
        elif task == "rough":
            H[v] = rng.uniform(0.0, rough_height, size=(rows, nx))
            H[v][:, xs < 0.5] = 0.0
        else:
            z, _ = profile(task, xs, params[v, 0], params[v, 1], params[v, 2], n_steps)
            H[v] = z[None, :]
    return {"heights": H.reshape(variants * rows, nx), "cell": float(cell), "origin": (float(x_min), float(-half_width)),
            "bands": int(variants), "params": params}
