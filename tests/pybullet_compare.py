#!/usr/bin/env python
"""Gap between a pybullet trajectory dump (tools/pybullet_baseline.py --dump, format in its DUMP_FORMAT) and this repo's
oracle on the same scenario: the measured replacement for "parity unpinned"  on the first box that has
pybullet.  TEST INFRASTRUCTURE (it drives oracle/): lives under tests/.  Needs no pybullet itself.
    python tests/pybullet_compare.py dump.npy [--solver-iters N]
    python tests/pybullet_compare.py --oracle-dump out.npy --steps 400     (the oracle's own trajectory in the dump format)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def oracle_trajectory(steps, solver_iters=None, kp=100.0, friction=None):
    """the same scenario on this repo's oracle -> [steps, 19] in the dump format"""
    from paddlerobotics_amd import a1_model as A
    from paddlerobotics_amd.etg import ETG_layer, Opt_with_points
    from oracle.oracle import OracleSim
    # solver_iters None = the library default = pybullet's own: up to 50 iterations with the 1e-7 residual exit (a count
    # alone switches the exit off)
    sim = OracleSim(A.default_config(1, solver_iters=solver_iters))
    row = A.default_dynamic_row()
    row[21:33] = kp                          # a1.py:75-80 simulation gains (the identified default is 80)
    row[0] = 0.0                             # no control latency in the plain pybullet loop above
    if friction is not None:
        row[1] = friction
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w, b, _ = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
    sim.set_params(dyn=row[None], etg_w=w, etg_b=b)
    sim.reset()
    out = np.zeros((steps, 19))
    for k in range(steps):
        sim.step(np.zeros((1, 12)))
        st = sim.get_state()[0]
        out[k, :7], out[k, 7:] = st[:7], st[13:25]
    return out


def compare(dump_path, solver_iters=None, friction=None):
    """gap between a pybullet dump (--dump on a box that has pybullet) and the oracle on the same scenario: the measured
    replacement for "parity unpinned" (DESIGN.md section 3).  Needs no pybullet."""
    ref = np.load(dump_path)
    mine = oracle_trajectory(ref.shape[0], solver_iters=solver_iters, friction=friction)
    dq = np.abs(ref[:, 7:] - mine[:, 7:]).max(1)
    dp = np.linalg.norm(ref[:, :3] - mine[:, :3], axis=1)
    dot = np.abs((ref[:, 3:7] * mine[:, 3:7]).sum(1)).clip(0, 1)
    dang = 2 * np.arccos(dot)
    marks = [k for k in (1, 10, 50, 100, 200, 400) if k <= ref.shape[0]]
    return {"steps": int(ref.shape[0]), "solver_iters": solver_iters,
            "joint_angle_gap_rad": {str(k): float(dq[:k].max()) for k in marks},
            "base_position_gap_m": {str(k): float(dp[:k].max()) for k in marks},
            "base_orientation_gap_rad": {str(k): float(dang[:k].max()) for k in marks},
            "distance_travelled_m": {"dump": float(ref[-1, 0] - ref[0, 0]), "oracle": float(mine[-1, 0] - mine[0, 0])}}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("dump", nargs="?", default=None)
    ap.add_argument("--oracle-dump", type=str, default=None)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--solver-iters", type=int, default=None, help="exactly this many sweeps per tick instead of pybullet's rule (<= 50, residual 1e-7)")
    a = ap.parse_args()
    if a.oracle_dump:
        np.save(a.oracle_dump, oracle_trajectory(a.steps, a.solver_iters))
        print(json.dumps({"oracle_dump": a.oracle_dump, "steps": a.steps}))
    elif a.dump:
        print(json.dumps(compare(a.dump, a.solver_iters)))
    else:
        ap.error("give a dump file or --oracle-dump")
