"""Physics self-consistency of the CPU oracle.  The reference's engine (Bullet) is
absent, so dynamics parity is UNPINNED against it; what can be pinned is
that the oracle's model obeys mechanics and the reference's kinematics:
  - free flight conserves energy / momentum to first order in dt  (M, C consistent)
  - static stance satisfies tau = C - J_c^T f with the contact Jacobian built from the
    reference's analytical_leg_jacobian (a1.py:132-160)
  - sliding decelerates at mu*g
"""
import os

import numpy as np

from paddlerobotics_amd import a1_model as A
from oracle import oracle as O


def quat2mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _free_flight_drift(dt, ticks):
    rng = np.random.default_rng(0)
    sim = O.OracleSim(A.default_config(1, settle_ticks=0, sim_dt=dt))
    row = A.default_dynamic_row()
    row[45:48] = 0
    sim.set_params(dyn=row[None])
    st = np.zeros(37)
    st[2] = 5.0
    qt = rng.normal(size=4)
    st[3:7] = qt / np.linalg.norm(qt)
    st[7:10] = rng.normal(size=3)
    st[10:13] = rng.normal(size=3) * 2
    st[13:25] = A.INIT_MOTOR_ANGLES + rng.normal(size=12) * 0.2
    st[25:37] = rng.normal(size=12) * 3
    sim.set_state(st[None])

    def energy_mom():
        M, _ = sim.dynamics_terms()
        s = sim.get_state()[0]
        R = quat2mat(s[3:7])
        v = np.concatenate([R.T @ s[10:13], R.T @ s[7:10], s[25:37]])
        h = M[:6] @ v
        lin = R @ h[3:6]
        ang = R @ h[:3] + np.cross(s[:3], lin)
        return 0.5 * v @ M @ v, lin, ang
    e0, l0, a0 = energy_mom()
    sim.tick(np.zeros((1, 12)), ticks)
    e1, l1, a1 = energy_mom()
    return abs(e1 - e0) / e0, np.abs(l1 - l0).max(), np.abs(a1 - a0).max()


def test_mass_matrix_spd_and_total_mass():
    sim = O.OracleSim(A.default_config(1, settle_ticks=0))
    M, _ = sim.dynamics_terms()
    assert np.abs(M - M.T).max() < 1e-15
    assert np.linalg.eigvalsh(M).min() > 1e-4
    total = 4.713 * 1.5 + 4 * (0.696 + 1.013 + 0.166 + 0.06)
    assert abs(M[3, 3] - total) < 1e-12 and abs(M[4, 4] - total) < 1e-12


def test_free_flight_conservation_first_order():
    e_a, l_a, a_a = _free_flight_drift(2e-4, 1000)
    e_b, l_b, a_b = _free_flight_drift(5e-5, 4000)
    assert e_a < 3e-4 and l_a < 5e-3 and a_a < 3e-2
    # errors shrink ~4x when dt shrinks 4x (consistent M and C, first-order integrator)
    assert 3.0 < e_a / e_b < 5.0 and 3.0 < l_a / l_b < 5.0 and 3.0 < a_a / a_b < 5.0


def test_static_stance_force_balance_with_reference_jacobian():
    cfg = A.default_config(1, settle_ticks=3000, solver_iters=50)
    sim = O.OracleSim(cfg)
    row = A.default_dynamic_row()
    row[1] = 1.0
    sim.set_params(dyn=row[None])
    sim.reset()
    st = sim.get_state()[0]
    q, qd = st[13:25], st[25:37]
    assert np.abs(qd).max() < 1e-4
    tau = -row[21:33] * (q - A.INIT_MOTOR_ANGLES) - row[33:45] * qd
    M, Cb = sim.dynamics_terms()
    lam = sim.get_lambda()[0] / cfg.sim_dt
    R = quat2mat(st[3:7])
    tot = np.zeros(3)
    for l in range(4):
        f_w = np.array([lam[3 * l + 1], lam[3 * l + 2], lam[3 * l]])
        tot += f_w
        f_b = R.T @ f_w
        ql = q[3 * l:3 * l + 3]
        J = O.leg_jacobian(ql, l)                         # reference foot-centre Jacobian
        # shift to the contact point (sphere bottom): d(off)/dq_j = z_j x off
        off = -A.FOOT_RADIUS * (R.T @ np.array([0, 0, 1.0]))
        ca, sa = np.cos(ql[0]), np.sin(ql[0])
        z1, z2 = np.array([1.0, 0, 0]), np.array([0, ca, sa])
        Jc = J + np.stack([np.cross(z1, off), np.cross(z2, off), np.cross(z2, off)], axis=1)
        assert np.allclose(tau[3 * l:3 * l + 3], Cb[6 + 3 * l:9 + 3 * l] - Jc.T @ f_b, atol=2e-4)
    assert abs(tot[2] - M[3, 3] * 10.0) < 1e-2 and np.abs(tot[:2]).max() < 1e-2


def test_sliding_friction_deceleration():
    cfg = A.default_config(1, settle_ticks=1500, solver_iters=8)
    sim = O.OracleSim(cfg)
    row = A.default_dynamic_row()
    row[1] = 0.2
    sim.set_params(dyn=row[None])
    sim.reset()
    st = sim.get_state()
    st[0, 7], st[0, 8] = 1.0, 0.5
    sim.set_state(st)
    v0 = np.hypot(1.0, 0.5)
    for _ in range(200):
        s = sim.get_state()[0]
        tau = -row[21:33] * (s[13:25] - A.INIT_MOTOR_ANGLES) - row[33:45] * s[25:37]
        sim.tick(tau[None], 1)
    s = sim.get_state()[0]
    decel = (v0 - np.hypot(s[7], s[8])) / (200 * cfg.sim_dt)
    assert abs(decel - 0.2 * 10.0) < 0.15
    assert abs(s[8] / s[7] - 0.5) < 0.08     # friction opposes the slip direction


def test_fp32_oracle_tracks_fp64_short_horizon():
    cfg = A.default_config(2, settle_ticks=100)
    a, b = O.OracleSim(cfg), O.OracleSim(cfg, dtype=np.float32)
    a.reset()
    b.reset()
    rng = np.random.default_rng(1)
    for _ in range(5):
        act = rng.uniform(-0.1, 0.1, size=(2, 12))
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
    assert np.allclose(a.get_state(), b.get_state(), atol=2e-3)
    assert np.allclose(oa, ob, atol=5e-2)


def test_pybullet_dump_format_and_comparison_tool(tmp_path):
    """tools/pybullet_baseline.py + tests/pybullet_compare.py: the trajectory dump format a pybullet box would produce and the gap report against the
    oracle.  No pybullet here: the oracle's own dump (K = 50) stands in for the file, so the gap against K = 50 is zero and
    against K = 2 it is the solver-convergence gap (small, and it grows with the horizon)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import pybullet_baseline
    from tests import pybullet_compare as PB
    assert "[steps, 19]" in pybullet_baseline.DUMP_FORMAT
    traj = PB.oracle_trajectory(30, solver_iters=50)
    assert traj.shape == (30, 19) and np.isfinite(traj).all()
    assert np.abs(np.linalg.norm(traj[:, 3:7], axis=1) - 1).max() < 1e-9
    f = str(tmp_path / "dump.npy")
    np.save(f, traj)
    same = PB.compare(f, solver_iters=50)
    assert same["steps"] == 30 and max(same["joint_angle_gap_rad"].values()) == 0.0
    k2 = PB.compare(f, solver_iters=2)
    assert 0.0 < max(k2["joint_angle_gap_rad"].values()) < 5e-2
    assert pybullet_baseline.run(1) is None or isinstance(pybullet_baseline.run(1), float)      # pybullet absent -> None
