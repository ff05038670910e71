"""Why some ticks need many sweeps: the residual of Bullet-order projected Gauss-Seidel, sweep by sweep, for a few stances
(the independent numpy statement of tests/test_oracle_physics.py; residual = max over rows of ((d lambda_r) A_rr)^2, the quantity
pybullet compares with solverResidualThreshold = 1e-7).  CPU only.
usage: python tools/residual_curve.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from paddlerobotics_amd import a1_model as A
from oracle import oracle as O
from tests.test_oracle_physics import _bullet_order_solve


def curve(label, mu, prep=None, body=2, sweeps=60, kick=(0.05, 0.02, 0.0), seed=7):
    cfg = A.default_config(1, settle_ticks=800, solver_iters=50, body_contacts=body)
    sim = O.OracleSim(cfg)
    row = A.default_dynamic_row(); row[1] = mu
    sim.set_params(dyn=row[None]); sim.reset()
    st = sim.get_state()
    rng = np.random.default_rng(seed)
    st[0, 7:10] += np.array(kick)
    st[0, 25:37] += rng.normal(size=12) * 0.3
    if prep:
        prep(st)
    sim.set_state(st)
    s0, lam_prev = sim.get_state()[0].copy(), sim.get_lambda()[0].copy()
    tau = -row[21:33] * (s0[13:25] - A.INIT_MOTOR_ANGLES) - row[33:45] * s0[25:37]
    tr = []
    lam, _, _ = _bullet_order_solve(sim, s0, lam_prev, tau, cfg, mu, sweeps, trace=tr)
    tr = np.array(tr)
    first = int(np.argmax(tr <= 1e-7)) + 2 if (tr <= 1e-7).any() else None      # sweep after which the test passes
    rate = (tr[20] / tr[5]) ** (1 / 15.0) if len(tr) > 20 and tr[5] > 0 and tr[20] > 0 else float("nan")
    on_cone = sum(1 for l in range(4) if lam[3 * l] > 0 and abs(np.hypot(lam[3 * l + 1], lam[3 * l + 2]) - mu * lam[3 * l]) < 1e-9)
    print("%-44s mu %.1f  stops after sweep %-4s residual after 2/4/8/16/32 sweeps: %s  contraction per sweep %.3f  feet on the cone %d/4"
          % (label, mu, first, " ".join("%.1e" % tr[k] for k in (0, 2, 6, 14, 30) if k < len(tr)), rate, on_cone))
    return tr


def kneel(st):
    st[0, 13:19] = np.array([0.05, 1.45, -2.55, -0.05, 1.45, -2.55]); st[0, 2] = 0.118
    st[0, 3:7] = np.array([0.0, np.sin(0.17), 0.0, np.cos(0.17)])
    st[0, 7:10] = np.array([0.4, 0.1, -0.3]); st[0, 10:13] = np.array([0.0, 0.5, 0.8])


if __name__ == "__main__":
    print("stance on four feet, small kick (sticking or sliding feet):")
    for mu in (0.2, 0.5, 1.0, 2.0, 3.2):
        curve("standing, toe spheres only", mu, body=0)
    print("stance on four feet, sideways slide 0.5 m/s:")
    for mu in (0.2, 1.0, 3.2):
        curve("sliding, toe spheres only", mu, body=0, kick=(0.5, 0.2, -0.05))
    print("kneeling: front knee spheres + hind feet (body_friction 0.5):")
    for mu in (0.2, 1.0, 3.2):
        curve("kneeling, default contact set", mu, prep=kneel)
