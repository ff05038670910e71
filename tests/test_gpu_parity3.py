"""-m gpu, round 3: the contact solver's stopping rule on the device, and the evidence VERDICT r02 found thin.

  * the residual rule (EtgConfig.solver_residual: pybullet's numSolverIterations = 50 / solverResidualThreshold = 1e-7, the
    library default) through the C-ABI against the fp64 oracle, both lane mappings, flat ground and heightfield; the executed
    sweep count the kernels report (info['solver_sweeps']) against the oracle's per-robot counts;
  * a robot's result does not depend on the robots sharing its wavefront (converged robots are frozen while neighbours sweep on):
    a permuted batch gives bit-identical per-robot results, step kernels and fused rollouts, both mappings;
  * 400-step statistics on the configs[4] heightfield with body contacts and joint limits on, and of the configs[2] closed
    loop, 4096 robots against 512 oracle robots, for the residual rule and for K = 50;
  * the 4-lane mapping at 16384 robots: determinism, batch invariance, a 256-robot oracle sample over 12 steps, a heightfield run;
  * the reference's trained gait (gait_action_list_ETG_exp.npy -> exp_w / exp_b) walks its 600 steps (env_test.py:51-54) at
    ~ vel_d without terminating, both mappings, as it does in the oracle;
  * the fused rollout over a caller-supplied action tape (etg_rollout_actions) against env.step() with the same actions.
"""
import os

import numpy as np
import pytest

from paddlerobotics_amd import a1_model as A

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tests.test_gpu_parity import _need_gpu, _etg_params, _make, _oracle, _lt, _ensemble   # noqa: E402
from tests.parity_util import sens_robots   # noqa: E402
from tests.test_gpu_parity2 import _say, _policy, _population               # noqa: E402

NCPU = os.cpu_count() or 1


def _heightfield():
    """BASELINE configs[4]: 256 x 256 grid, 0.05 m cells, heights U(0, 0.05) m from default_rng(0)"""
    hf = np.random.default_rng(0).uniform(0.0, 0.05, size=(256, 256)).astype(np.float32)
    return dict(heights=hf, cell=0.05, origin=(-6.4, -6.4))


def _rolling_hills():
    """a SMOOTH heightfield (same grid): +-3 cm hills of ~4 m wavelength.  The bilinear surface is C0 everywhere, but here its
    normals jump by ~1e-3 rad at cell edges instead of ~1 rad on the U(0, 0.05) grid, so fp32 and fp64 trajectories stay
    together for many steps and a tight bound means something."""
    x = -6.4 + 0.05 * np.arange(256)
    hf = 0.03 * (1.0 + np.sin(1.5 * x)[None, :] * np.cos(1.3 * x)[:, None])
    return dict(heights=hf.astype(np.float32), cell=0.05, origin=(-6.4, -6.4))


@pytest.mark.parametrize("lanes", [4, 16])
@pytest.mark.parametrize("terrain", ["flat", "heightfield"])
def test_residual_rule_matches_oracle(lanes, terrain):
    """Default solver (up to 50 sweeps, residual 1e-7) against the fp64 oracle over 20 control steps of random residual
    actions.  Bounds ~10x the measured gaps (printed); the executed sweep count of a wave is the count of its slowest robot."""
    _need_gpu()
    n = 64
    hf = _rolling_hills() if terrain == "heightfield" else None
    kw = dict(task="heightfield", heightfield=hf) if hf else {}
    W, B = _etg_params(n, seed=21)
    env = _make(n, lanes_per_robot=lanes, **kw)
    assert (env.cfg.solver_iters, env.cfg.solver_residual) == (50, 1e-7)
    orc = _ensemble(n, terrain=1 if hf else 0, heightfield=hf)
    if hf:
        orc.set_heightfield(hf["heights"])
    env.reset(ETG_w=W, ETG_b=B)
    orc.set_params(etg_w=W, etg_b=B)
    orc.reset()
    rng = np.random.default_rng(2)
    worst_q, worst_p, sq, sp = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    per_wave = 64 // lanes
    waves_compared = 0
    for k in range(20):
        act = rng.uniform(-0.1, 0.1, size=(n, 12))
        _, rg, dg, info = env.step(torch.as_tensor(act, dtype=torch.float32))
        _, ro, do, io = orc.step(act)
        sg, so = env.get_state().cpu().numpy(), orc.get_state()
        worst_q = np.maximum(worst_q, np.abs(sg - so)[:, 13:25].max(1))
        worst_p = np.maximum(worst_p, np.abs(sg - so)[:, :3].max(1))
        sq = np.maximum(sq, orc.spread(slice(13, 25))); sp = np.maximum(sp, orc.spread(slice(0, 3)))
        sw_g = info["solver_sweeps"].cpu().numpy().reshape(-1)
        sw_o = io[:, A.INFO_SWEEPS].reshape(-1, per_wave)
        sw_g = sw_g.reshape(-1, per_wave)
        assert np.all(sw_g == sw_g[:, :1])                          # wave-uniform
        # the wave runs, per tick, the count of its slowest robot: between the largest per-robot step total and the sum
        # (waves whose robots' ensembles are all still one trajectory: a robot past a grip bifurcation counts its own sweeps)
        on = (sq < 1e-5).reshape(-1, per_wave).all(1)
        waves_compared += int(on.sum())
        assert np.all(sw_g[on, 0] >= sw_o[on].max(1) - 2) and np.all(sw_g[on, 0] <= sw_o[on].sum(1) + 2), (k, sw_g[:, 0], sw_o)
        assert np.all(sw_g >= 13) and np.all(sw_g <= 13 * 50)
    assert waves_compared >= 10
    _say("residual rule %s lanes %d: executed sweeps/tick %.2f vs oracle per robot %.2f" % (terrain, lanes, sw_g.mean() / 13, sw_o.mean() / 13))
    sens_robots(worst_q, sq, 1e-3 if hf else 1e-4, "residual rule %s lanes %d joint angles, 20 steps" % (terrain, lanes))   # measured (toe spheres only): 8e-6 / 5e-7 on flat ground
    sens_robots(worst_p, sp, 2e-4 if hf else 2e-5, "residual rule %s lanes %d base position" % (terrain, lanes))
    env.close()


@pytest.mark.parametrize("lanes", [4, 16])
@pytest.mark.parametrize("terrain", ["flat", "heightfield"])
def test_wave_neighbours_do_not_influence_a_robot(lanes, terrain):
    """The same 256 robots in two orders (so every robot shares its wavefront with different neighbours): states, rewards and
    observations are bit-identical per robot, through env.step() and through the fused rollout, under the residual rule."""
    _need_gpu()
    n = 256
    hf = _heightfield() if terrain == "heightfield" else None
    kw = dict(task="heightfield", heightfield=hf) if hf else {}
    W, B = _etg_params(n, seed=23)
    rng = np.random.default_rng(5)
    dyn = np.stack([A.dynamic_dict_to_row(A.param2dynamic_dict(rng.uniform(-0.3, 0.3, 48))) for _ in range(n)])
    perm = rng.permutation(n)
    act = rng.uniform(-0.15, 0.15, size=(6, n, 12)).astype(np.float32)
    res = []
    for order in (np.arange(n), perm):
        env = _make(n, lanes_per_robot=lanes, **kw)
        env.reset(ETG_w=W[order], ETG_b=B[order], dynamic_param=torch.as_tensor(dyn[order], dtype=torch.float32))
        rews = []
        for k in range(6):
            _, r, _, _ = env.step(torch.as_tensor(act[k][order]))
            rews.append(r.cpu().numpy().copy())
        st_step = env.get_state().cpu().numpy().copy()
        ret, ln = env.rollout_openloop(30)
        res.append((order, np.stack(rews, 1), st_step, env.get_state().cpu().numpy().copy(), env.obs.cpu().numpy().copy(),
                    ret.cpu().numpy().copy(), ln.cpu().numpy().copy()))
        env.close()
    (o0, *a), (o1, *b) = res
    inv = np.empty(n, dtype=int)
    inv[o1] = np.arange(n)
    for x, y in zip(a, b):
        assert np.array_equal(x, y[inv])


def _stats_vs_oracle(env, orc, steps, m, label, closed_loop=None):
    """400-step statistics of the GPU batch against the oracle on the first m robots (see test_long_horizon_statistics_match_oracle)"""
    from scipy.stats import ks_2samp
    x0 = env.get_state()[:, 0].cpu().numpy()
    if closed_loop is None:
        ret_g, ln_g = env.rollout_openloop(steps)
    else:
        ret_g, ln_g = env.rollout_policy(closed_loop[0], steps, 0.3)
    ret_g, ln_g = ret_g.cpu().numpy().astype(np.float64), ln_g.cpu().numpy()
    dx_g = env.get_state()[:, 0].cpu().numpy() - x0
    x0o = orc.get_state()[:, 0].copy()
    if closed_loop is None:
        ret_o, ln_o = orc.run_steps(steps)
    else:
        from oracle import oracle as O
        obs = closed_loop[2]
        ret_o, alive, ln_o = np.zeros(m), np.ones(m, bool), np.zeros(m, int)
        for _ in range(steps):
            obs, r, d, _ = orc.step(O.mlp_forward(obs, *closed_loop[1], scale=0.3), want_info=False)
            ret_o += alive * r
            ln_o += alive
            alive &= ~d.astype(bool)
    dx_o = orc.get_state()[:, 0] - x0o
    surv = lambda ln, t: float((ln > t).mean())
    grid = list(range(25, steps, 25))
    gap = max(abs(surv(ln_g[:m], t) - surv(ln_o, t)) for t in grid)
    gap_full = max(abs(surv(ln_g, t) - surv(ln_o, t)) for t in grid)
    agree = float((np.abs(ln_g[:m] - ln_o) <= 1).mean())
    ks_len, ks_ret = ks_2samp(ln_g[:m], ln_o).statistic, ks_2samp(ret_g[:m], ret_o).statistic
    alive = (ln_g[:m] == steps) & (ln_o == steps)
    ks_dx = ks_2samp(dx_g[:m][alive], dx_o[alive]).statistic if alive.sum() > 20 else 0.0
    _say("%s: survivors gpu %.3f (full %.3f) oracle %.3f | survival-curve gap %.4f (full batch %.4f) | same length +-1: %.3f | "
         "KS len %.4f ret %.4f dx %.4f | mean return gpu %.2f oracle %.2f" %
         (label, surv(ln_g[:m], steps - 1), surv(ln_g, steps - 1), surv(ln_o, steps - 1), gap, gap_full, agree, ks_len, ks_ret,
          ks_dx, ret_g[:m].mean(), ret_o.mean()))
    assert np.isfinite(ret_g).all()
    # difference of the mean returns in standard errors of a two-sample comparison (m vs m draws)
    z = abs(ret_g[:m].mean() - ret_o.mean()) / max(np.sqrt((ret_g[:m].var() + ret_o.var()) / m), 1e-9)
    _say("%s: mean-return difference %.2f standard errors (std gpu %.1f oracle %.1f)" % (label, z, ret_g[:m].std(), ret_o.std()))
    return dict(gap=gap, gap_full=gap_full, agree=agree, ks_len=ks_len, ks_ret=ks_ret, ks_dx=ks_dx,
                dret=abs(ret_g[:m].mean() - ret_o.mean()) / max(ret_o.std(), 1e-9), z=z)


@pytest.mark.parametrize("solver,lanes", [("rule", 16), ("k50", 16), ("rule", 4)])
def test_long_horizon_statistics_on_the_heightfield(solver, lanes):
    """configs[4] per GPU: 4096 robots on the random heightfield, 400 control steps, body contacts (deepest of knee / shin /
    trunk corner) and joint-limit stops ON, against 512 fp64 oracle robots.  The terrain is only C0 and its normals jump by
    up to ~1 rad at cell edges, so individual fp32 / fp64 trajectories part within tens of steps (only 20 % of the robots end
    their episode on the same step): what must agree are the DISTRIBUTIONS -- two-sample KS statistics of episode length,
    return and distance below the 0.122 critical value (alpha = 0.001, 512 vs 512), mean returns within 4 standard errors."""
    _need_gpu()
    n, m, steps = 4096, 512, 400
    hf = _heightfield()
    skw = {} if solver == "rule" else dict(solver_iters=50)
    w, b = _population(n)
    env = _make(n, task="heightfield", heightfield=hf, body_contacts=2, joint_limits=True, lanes_per_robot=lanes, **skw)
    assert env.lanes_per_robot == lanes
    env.reset(ETG_w=w, ETG_b=b)
    orc = _oracle(m, terrain=1, heightfield=hf, body_contacts=2, joint_limits=1, **skw)
    orc.threads = NCPU
    orc.set_heightfield(hf["heights"])
    orc.set_params(etg_w=w[:m].double().cpu().numpy(), etg_b=b[:m].double().cpu().numpy())
    orc.reset()
    s = _stats_vs_oracle(env, orc, steps, m, "heightfield + body contacts + joint limits, %s, %d lanes per robot" % (solver, lanes))
    assert s["gap"] < 0.1 and s["gap_full"] < 0.1            # survival curves: same robots / full batch vs the 512-robot sample
    assert s["ks_len"] < 0.1 and s["ks_ret"] < 0.1 and s["ks_dx"] < 0.12       # measured 0.045-0.06 / 0.025-0.03 / 0
    assert s["z"] < 4.0
    if solver == "rule" and lanes == 16:
        # the same-length floor: how often does ANY fp32 evaluation end a robot's episode on the fp64 oracle's step
        # (+-1) on this terrain?  The oracle's own fp32 build sets the yardstick; the GPU must reach it within 3 standard errors
        # of the difference (tools/hf_tracking_probe.py: no hardware approximation separates the two)
        o32 = _oracle(m, dtype=np.float32, terrain=1, heightfield=hf, body_contacts=2, joint_limits=1, **skw)
        o32.threads = NCPU
        o32.set_heightfield(hf["heights"])
        o32.set_params(etg_w=w[:m].double().cpu().numpy(), etg_b=b[:m].double().cpu().numpy())
        o32.reset()
        _, ln32 = o32.run_steps(steps)
        _, ln64 = orc.episode_lengths() if hasattr(orc, "episode_lengths") else (None, None)
        if ln64 is None:                                     # the fp64 lengths: rerun (the first run's were consumed inside the helper)
            orc.reset()
            _, ln64 = orc.run_steps(steps)
        a32 = float((np.abs(ln32 - ln64) <= 1).mean())
        se = np.sqrt((a32 * (1 - a32) + s["agree"] * (1 - s["agree"])) / m)
        _say("heightfield + body contacts: same episode length +-1 as the fp64 oracle: gpu %.3f, fp32 oracle %.3f (standard error of "
             "the difference %.3f)" % (s["agree"], a32, se))
        assert s["agree"] >= a32 - 3.0 * se
    env.close()


@pytest.mark.parametrize("solver", ["rule", "k50"])
def test_long_horizon_statistics_of_the_closed_loop(solver):
    """configs[2]: ETG + residual policy (random init) fused rollout, 4096 robots x 400 steps, against the oracle's closed loop
    (mlp_forward + step) on the first 512 robots."""
    _need_gpu()
    n, m, steps = 4096, 512, 400
    skw = {} if solver == "rule" else dict(solver_iters=50)
    w, b = _population(n)
    pol, ws = _policy()
    env = _make(n, **skw)
    env.reset(ETG_w=w, ETG_b=b)
    orc = _oracle(m, **skw)
    orc.threads = NCPU
    orc.set_params(etg_w=w[:m].double().cpu().numpy(), etg_b=b[:m].double().cpu().numpy())
    obs0 = orc.reset()
    s = _stats_vs_oracle(env, orc, steps, m, "closed loop, %s" % solver, closed_loop=(pol, ws, obs0))
    assert s["gap"] < 0.03 and s["gap_full"] < 0.08
    assert s["agree"] > 0.85      # (toe spheres only: 0.95; a kneeling robot's last step hangs on when a knee sphere grips)
    assert s["ks_len"] < 0.04 and s["ks_ret"] < 0.04 and s["ks_dx"] < 0.1
    assert s["dret"] < 0.08
    env.close()


def test_four_lane_mapping_at_16384_robots():
    """The mapping every batch above 4096 robots gets (lanes_per_robot = 0 -> 4): reruns are bit-identical, a robot does not
    depend on its batch, the first 256 robots track the fp64 oracle over 12 control steps, and a heightfield run stays sane."""
    _need_gpu()
    n, m = 16384, 256
    W, B = _etg_params(m, seed=29)
    Wn, Bn = np.tile(W, (n // m, 1, 1)), np.tile(B, (n // m, 1))
    rng = np.random.default_rng(3)
    act = rng.uniform(-0.1, 0.1, size=(12, m, 12)).astype(np.float32)
    outs = []
    for rep in range(2):
        env = _make(n)
        assert env.lanes_per_robot == 4
        env.reset(ETG_w=Wn, ETG_b=Bn)
        for k in range(12):
            env.step(torch.as_tensor(np.tile(act[k], (n // m, 1))), want_info=False)
        outs.append((env.get_state().cpu().numpy().copy(), env.obs.cpu().numpy().copy(), env.reward.cpu().numpy().copy()))
        if rep == 0:
            ret, ln = env.rollout_openloop(60)
            roll = (ret.cpu().numpy().copy(), ln.cpu().numpy().copy(), env.get_state().cpu().numpy().copy())
        env.close()
    for x, y in zip(*outs):
        assert np.array_equal(x, y)                                             # determinism
    st = outs[0][0]
    assert np.array_equal(st.reshape(n // m, m, -1), np.broadcast_to(st[:m], (n // m, m, st.shape[1])))   # every copy == the sample
    assert np.isfinite(roll[0]).all() and np.isfinite(roll[2]).all()
    small = _make(m, lanes_per_robot=4)
    small.reset(ETG_w=W, ETG_b=B)
    for k in range(12):
        small.step(torch.as_tensor(act[k]), want_info=False)
    assert np.array_equal(small.get_state().cpu().numpy(), st[:m])              # batch invariance: 256 alone == 256 of 16384
    small.close()
    orc = _ensemble(m)
    orc.set_params(etg_w=W, etg_b=B)
    orc.reset()
    sq, sp = np.zeros(m), np.zeros(m)
    for k in range(12):
        orc.step(act[k].astype(np.float64), want_info=False)
        sq = np.maximum(sq, orc.spread(slice(13, 25))); sp = np.maximum(sp, orc.spread(slice(0, 3)))
    so = orc.get_state()
    eq, ep = np.abs(st[:m] - so)[:, 13:25].max(1), np.abs(st[:m] - so)[:, :3].max(1)
    _say("4 lanes, 16384 robots: q err median %.2e max %.2e | pos err max %.2e (256-robot oracle sample, 12 steps)" %
         (np.median(eq), eq.max(), ep.max()))
    sens_robots(eq, sq, 2e-4, "4 lanes, 16384 robots: joint angles, 12 steps (end state; spread = largest along the way)")
    sens_robots(ep, sp, 5e-5, "4 lanes, 16384 robots: base position")
    # heightfield at the same size
    hf = _heightfield()
    envh = _make(n, task="heightfield", heightfield=hf)
    assert envh.lanes_per_robot == 4
    envh.reset(ETG_w=Wn, ETG_b=Bn)
    ret, ln = envh.rollout_openloop(100)
    sth = envh.get_state().cpu().numpy()
    assert np.isfinite(sth).all() and np.isfinite(ret.cpu().numpy()).all()
    assert np.array_equal(sth[:m], sth[m:2 * m])
    ln = ln.cpu().numpy()
    orh = _oracle(64, terrain=1, heightfield=hf)
    orh.threads = NCPU
    orh.set_heightfield(hf["heights"])
    orh.set_params(etg_w=W[:64], etg_b=B[:64])
    orh.reset()
    _, ln_o = orh.run_steps(100)
    _say("4 lanes heightfield: alive after 100 steps gpu %.3f oracle(64) %.3f" % ((ln[:64] == 100).mean(), (ln_o == 100).mean()))
    assert abs((ln[:64] == 100).mean() - (ln_o == 100).mean()) < 0.25      # (64 vs 64 robots on a chaotic terrain: one standard error of the difference is 0.09)
    envh.close()


@pytest.mark.parametrize("lanes", [4, 16])
def test_reference_trained_gait_walks_600_steps(golden, lanes):
    """env_test.py:43-54 with the reference's recorded gait: the ETG fitted to gait_action_list_ETG_exp.npy (exp_w / exp_b of
    tests/golden/etg.npz) walks 600 control steps without terminating, at ~ vel_d = 0.5 m/s (the oracle: 7.7 m)."""
    _need_gpu()
    g = golden("etg")
    from oracle.oracle import OracleSim
    orc = OracleSim(A.default_config(1))
    orc.set_params(etg_w=g["exp_w"], etg_b=g["exp_b"])
    orc.reset()
    xo = orc.get_state()[0, 0]
    _, ln_o = orc.run_steps(600)
    dxo = orc.get_state()[0, 0] - xo
    for skw in ({}, dict(solver_iters=2), dict(solver_iters=50)):
        env = _make(4, lanes_per_robot=lanes, **skw)
        env.reset(ETG_w=g["exp_w"], ETG_b=g["exp_b"])
        x0 = env.get_state()[:, 0].cpu().numpy()
        ret, ln = env.rollout_openloop(600)
        dx = env.get_state()[:, 0].cpu().numpy() - x0
        _say("reference gait, lanes %d %s: length %s, distance %.3f m (oracle %.3f m, %d steps), return %.1f" %
             (lanes, skw or "rule", ln.cpu().numpy().tolist(), dx[0], dxo, ln_o[0], float(ret[0])))
        assert int(ln.min()) == 600 and ln_o[0] == 600
        assert np.all(np.abs(dx - dxo) < 0.15) and np.all(dx > 7.0)
        assert abs(dx[0] / (600 * 0.026) - 0.5) < 0.03                          # ~ vel_d
        env.close()


@pytest.mark.parametrize("lanes", [4, 16])
def test_pyramid_friction_option_matches_oracle(lanes):
    """friction_model = 1 (each tangent direction clamped on its own, the sequential-impulse pyramid) against the oracle."""
    _need_gpu()
    n = 32
    W, B = _etg_params(n, seed=31)
    env = _make(n, lanes_per_robot=lanes, friction_model=1)
    orc = _ensemble(n, friction_model=1)
    env.reset(ETG_w=W, ETG_b=B)
    orc.set_params(etg_w=W, etg_b=B)
    orc.reset()
    rng = np.random.default_rng(8)
    dyn = np.tile(A.default_dynamic_row(), (n, 1))
    sq = np.zeros(n)
    for k in range(15):
        act = rng.uniform(-0.2, 0.2, size=(n, 12))                              # large residuals: feet slide
        env.step(torch.as_tensor(act, dtype=torch.float32), want_info=False)
        orc.step(act, want_info=False)
        sq = np.maximum(sq, orc.spread(slice(13, 25)))
    sg, so = env.get_state().cpu().numpy(), orc.get_state()
    sens_robots(np.abs(sg - so)[:, 13:25].max(1), sq, 2e-4, "pyramid friction lanes %d: joint angles after 15 steps" % lanes)
    # and it is a different model: the disc result differs from it
    disc = _make(n, lanes_per_robot=lanes)
    disc.reset(ETG_w=W, ETG_b=B)
    rng = np.random.default_rng(8)
    for k in range(15):
        disc.step(torch.as_tensor(rng.uniform(-0.2, 0.2, size=(n, 12)), dtype=torch.float32), want_info=False)
    assert np.abs(disc.get_state().cpu().numpy() - sg)[:, 13:25].max() > 1e-4
    env.close(); disc.close()


@pytest.mark.parametrize("lanes", [4, 16])
@pytest.mark.parametrize("variant", ["default", "etg0_filter", "heightfield"])
def test_action_tape_rollout_equals_stepping(lanes, variant, monkeypatch):
    """etg_rollout_actions (one launch per ROLLOUT_CHUNK control steps -- 400 by default, 50 here so that the continuation from
    one launch to the next is part of the test --, commands known in advance) against env.step() with the same
    actions: bookkeeping exact (lengths, done bytes), trajectories to the rounding noise two kernels of the same source
    differ by (FMA contraction; cf. test_fused_rollout_equals_stepping), every recorded column against the stepped info."""
    _need_gpu()
    monkeypatch.setenv("ETG_ROLLOUT_CHUNK", "50")      # (read by etg_create)
    n, T = 64, 60                          # 60 > 50: two launches
    # (toe spheres only: two kernels of one source agree to rounding, and a gripping knee sphere amplifies rounding past any
    # fixed bound within tens of steps -- the default contact set is compared through distributions and against the oracle)
    kw = dict(lanes_per_robot=lanes, body_contacts=0)
    if variant == "etg0_filter":
        kw.update(ETG=0, enable_action_filter=True)
    elif variant == "heightfield":
        kw.update(task="heightfield", heightfield=_rolling_hills())      # (the U(0, 0.05) grid is chaotic within tens of steps)
    W, B = _etg_params(n, seed=41)
    rng = np.random.default_rng(6)
    acts = torch.as_tensor(rng.uniform(-0.15, 0.15, size=(T, n, 12)), dtype=torch.float32, device="cuda:0")
    a, b = _make(n, **kw), _make(n, **kw)
    for e in (a, b):
        e.reset(ETG_w=W, ETG_b=B) if variant != "etg0_filter" else e.reset()
    ret, ln, rec = a.rollout_actions(acts, record=("joint_angle", "obs-IMU", "obs", "reward", "done"))
    q, imu, obs, rew, done = [], [], [], [], []
    for k in range(T):
        o, r, d, info = b.step(acts[k])
        q.append(info["joint_angle"].clone()); imu.append(info["obs-IMU"].clone()); obs.append(o.clone())
        rew.append(r.clone()); done.append(d.clone())
    q, imu, obs, rew, done = (torch.stack(x).cpu().numpy() for x in (q, imu, obs, rew, done))
    ret_b, ln_b = b.episode_stats()
    first = 3                                                   # the first steps: identical arithmetic, rounding not yet amplified
    assert np.abs(rec["joint_angle"].cpu().numpy()[:first] - q[:first]).max() < 1e-5          # measured 2e-6
    same_len = (ln.cpu().numpy() == ln_b.cpu().numpy())
    assert same_len.mean() > 0.9
    run = np.arange(T)[:, None] < np.minimum(ln.cpu().numpy(), ln_b.cpu().numpy())[None, :] - 1     # steps before either episode ended
    tol = 5e-3 if variant == "heightfield" else 5e-4
    eq = np.abs(rec["joint_angle"].cpu().numpy() - q).max(2)
    _say("action tape %s lanes %d: joint gap to stepping, running robots: median %.2e max %.2e" % (variant, lanes, np.median(eq[run]), eq[run].max()))
    assert np.median(eq[run]) < 1e-5 and eq[run].max() < tol
    assert np.abs(rec["obs-IMU"].cpu().numpy() - imu).max(2)[run].max() < 20 * tol
    assert np.abs(rec["obs"].cpu().numpy() - obs)[:first].max() < 2e-3            # normalised rows (x10 / x38); measured 1.5e-4
    # EVERY recorded observation row of a running robot, the last row of each 50-step launch included (row 49: it is written
    # to `obs` by the kernel and copied onto the tape by the host -- ADVICE r3)
    og = np.abs(rec["obs"].cpu().numpy() - obs).max(2)
    assert np.isfinite(rec["obs"].cpu().numpy()).all()
    assert og[run].max() < 400 * tol, (og[run].max(), np.argwhere(og * run > 400 * tol)[:4])
    assert og[49][run[49]].max() < 400 * tol and og[49][run[49]].max() <= 3 * max(og[48][run[48]].max(), og[50][run[50]].max(), 1e-4)
    assert np.array_equal(rec["done"].cpu().numpy()[:first], done[:first])
    assert np.abs(rec["reward"].cpu().numpy()[:first] - rew[:first]).max() < 1e-3
    assert np.abs(ret.cpu().numpy() - ret_b.cpu().numpy())[same_len].max() < 2e-2 * (1 + np.abs(ret_b.cpu().numpy()).max())
    assert np.array_equal(a.obs.cpu().numpy(), rec["obs"].cpu().numpy()[-1])      # the final row is the tape's last
    # one command row for every robot
    a.reset() if variant == "etg0_filter" else a.reset(ETG_w=W, ETG_b=B)
    _, _, rec1 = a.rollout_actions(acts[:5, 0], record=("joint_angle",))
    a.reset() if variant == "etg0_filter" else a.reset(ETG_w=W, ETG_b=B)
    _, _, rec2 = a.rollout_actions(acts[:5, :1].expand(5, n, 12), record=("joint_angle",))
    assert torch.equal(rec1["joint_angle"], rec2["joint_angle"])
    with pytest.raises(ValueError):
        a.rollout_actions(acts[:, :5])
    a.close(); b.close()


def test_dynamics_identification_evaluator_fused_equals_stepping(golden, monkeypatch):
    """make_dynamics_id_evaluator through the action-tape rollout (2 launches per 100-step replay: ROLLOUT_CHUNK 50) gives the fitness of the
    env.step() loop (Dynamic_parallel_model.py:53-77) -- and how long a generation takes either way."""
    _need_gpu()
    import time
    from paddlerobotics_amd import rollout as R
    monkeypatch.setenv("ETG_ROLLOUT_CHUNK", "50")
    n, T = 256, 100
    g = golden("dynid")
    POSE = A.INIT_MOTOR_ANGLES
    rng = np.random.default_rng(14)
    etg = golden("etg")
    rows = {int(r): a for r, a in zip(etg["exp_rows"], etg["exp_act"])}
    exp = np.zeros((T, 12)); last = np.zeros(12)
    for k in range(T):
        last = rows.get(k, last)
        exp[k] = POSE + last
    gait = {"exp": exp, "ori": np.tile(POSE[None], (T, 1))}
    mean_dict = {}
    for key in ("exp", "ori"):
        mean_dict[key + "_motor_mean"], mean_dict[key + "_drpy_mean"] = gait[key], np.zeros((T, 3))
        mean_dict[key + "_motor_std"], mean_dict[key + "_drpy_std"] = np.full((T, 12), 0.05), np.full((T, 3), 0.5)
    cand = torch.as_tensor(rng.uniform(-0.2, 0.2, size=(n, 48)))
    env = _make(n, ETG=0, body_contacts=0)      # (two kernels of one source: see test_action_tape_rollout_equals_stepping)
    out = {}
    for fused in (True, False):
        ev = R.make_dynamics_id_evaluator(env, gait, mean_dict, e_steps=T, fused=fused)
        ev(cand)                                                 # warm (lazy kernel loads)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fit = ev(cand)
        torch.cuda.synchronize()
        out[fused] = (fit.cpu().numpy(), time.perf_counter() - t0, env.episode_stats()[1].cpu().numpy())
    up = (out[True][2] == T) & (out[False][2] == T)              # robots still on their feet at the end of the last replay
    gap = np.abs(out[True][0] - out[False][0])
    _say("dynamics-ID evaluation of %d candidates x 2 gaits x %d steps: fused %.1f ms, env.step loop %.1f ms | fitness gap: "
         "standing robots (%.0f %%) max %.2e, all max %.2e" % (n, T, out[True][1] * 1e3, out[False][1] * 1e3, 100 * up.mean(),
                                                                gap[up].max(), gap.max()))
    # a fallen robot keeps being stepped (as in the reference's loop) and thrashes chaotically: its loss is noise in both paths
    assert up.mean() > 0.5
    assert gap[up].max() < 2e-2 * (1 + np.abs(out[False][0][up]).max())
    assert np.corrcoef(out[True][0][up], out[False][0][up])[0, 1] > 0.999
    env.close()


def _rccl_worker(rank, world, port, n_per_rank, out):
    """one rank of the 2-GPU ES generation: its own GPU, RCCL (backend "nccl"), its slice of the candidates"""
    import torch.distributed as dist
    from paddlerobotics_amd import rollout as R
    from paddlerobotics_amd.es import SimpleGA
    from paddlerobotics_amd.env import make_env
    from paddlerobotics_amd.etg import ETG_layer, Opt_with_points
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    env = make_env("Quadrupedal", num_envs=n_per_rank, device="cuda:%d" % rank)
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w0, b0, prior = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
    ga = SimpleGA(12, sigma_init=0.02, sigma_decay=0.99, sigma_limit=0.005, elite_ratio=0.1, weight_decay=0.005,
                  popsize=world * n_per_rank, param=np.zeros(12), seed=5, device="cuda:%d" % rank)
    ev = R.make_etg_evaluator(env, layer, 0.5, prior, w0, b0, max_step=60)
    fits = [R.es_generation(ga, ev, dist, rank, world).cpu().numpy() for _ in range(2)]
    out[rank] = (np.stack(fits), ga.best_param.cpu().numpy(), dist.get_world_size(), dist.get_backend())
    env.close()
    dist.destroy_process_group()


def test_two_rank_es_generation_over_rccl():
    """BASELINE configs[3] in miniature, the moment two GPUs exist: two ranks, one GPU each, RCCL all_gather of the returns,
    replicated tell -- the gathered fitness vector and the next population equal the single-rank run of the same
    candidates (robots do not depend on their batch).  Skipped on a 1-GPU box (the gloo world-size-2 CPU test covers the
    control flow there: tests/test_es_and_sharding.py)."""
    _need_gpu()
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices (one rank per GPU over RCCL)")
    import socket
    import torch.multiprocessing as mp
    from paddlerobotics_amd import rollout as R
    from paddlerobotics_amd.es import SimpleGA
    from paddlerobotics_amd.etg import ETG_layer, Opt_with_points
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    n_per_rank, world = 64, 2
    out = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(world, port, n_per_rank, out), nprocs=world, join=True)
    env = _make(world * n_per_rank)
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w0, b0, prior = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
    ga = SimpleGA(12, sigma_init=0.02, sigma_decay=0.99, sigma_limit=0.005, elite_ratio=0.1, weight_decay=0.005,
                  popsize=world * n_per_rank, param=np.zeros(12), seed=5, device="cuda:0")
    ev = R.make_etg_evaluator(env, layer, 0.5, prior, w0, b0, max_step=60)
    ref = np.stack([R.es_generation(ga, ev).cpu().numpy() for _ in range(2)])
    for r in range(world):
        fits, best, ws, backend = out[r]
        assert ws == world and backend == "nccl"
        assert np.array_equal(fits, ref)                                  # gather order = candidate order, robots batch-invariant
        assert np.array_equal(best, ga.best_param.cpu().numpy())          # replicated tell
    env.close()


@pytest.mark.parametrize("lanes", [4, 16])
def test_pd_latency_matches_oracle(lanes):
    """EtgConfig.pd_latency (minitaur.py:100,1195-1199; make_env(pd_latency=seconds)): the PD law reads the joint state of
    1.3 ms ago, every sub-step and during the reset settle; cached resets reproduce the simulated one."""
    _need_gpu()
    n = 32
    W, B = _etg_params(n, seed=43)
    env = _make(n, lanes_per_robot=lanes, pd_latency=0.0013)
    orc = _ensemble(n, pd_latency=0.0013)
    env.reset(ETG_w=W, ETG_b=B)
    orc.set_params(etg_w=W, etg_b=B)
    orc.reset()
    s0 = env.get_state().cpu().numpy()
    assert np.abs(s0 - orc.get_state())[:, 13:25].max() < 1e-5
    rng = np.random.default_rng(9)
    acts = rng.uniform(-0.1, 0.1, size=(12, n, 12))
    sq = np.zeros(n)
    for k in range(12):
        env.step(torch.as_tensor(acts[k], dtype=torch.float32), want_info=False)
        orc.step(acts[k], want_info=False)
        sq = np.maximum(sq, orc.spread(slice(13, 25)))
    sg, so = env.get_state().cpu().numpy(), orc.get_state()
    sens_robots(np.abs(sg - so)[:, 13:25].max(1), sq, 1e-4, "pd_latency 1.3 ms lanes %d: joint angles after 12 steps" % lanes)
    env.reset()                                             # from the settle cache
    assert np.array_equal(env.get_state().cpu().numpy(), s0)
    for k in range(12):
        env.step(torch.as_tensor(acts[k], dtype=torch.float32), want_info=False)
    assert np.array_equal(env.get_state().cpu().numpy(), sg)
    env.close()


@pytest.mark.parametrize("variant", ["flat", "heightfield", "student", "bf16"])
def test_four_lane_closed_loop_kernel(variant):
    """etg_rollout_policy on the 4-lanes-per-robot mapping (k_rollout_policy4: 64 robots per workgroup, two stacked policy tiles
    per weight fetch) against predict() + step() on the same mapping, against the 16-lane fused kernel, and against the oracle's
    closed loop (mlp_forward + step, train.py:213-249)."""
    _need_gpu()
    from oracle import oracle as O
    n, m, steps = 128, 64, 12
    kw = {}
    hf = None
    if variant == "heightfield":
        hf = _rolling_hills()
        kw = dict(task="heightfield", heightfield=hf)
    elif variant == "student":
        kw = dict(sensor_mode={"dis": 0})                       # the 46-float observation of BCtrain.py:53-59 (columns 3..48)
    prec = 1 if variant == "bf16" else 0
    obs_dim = 46 if variant == "student" else 49
    pol, ws = _policy(obs_dim)
    W, B = _etg_params(m, seed=47)
    Wn, Bn = np.tile(W, (n // m, 1, 1)), np.tile(B, (n // m, 1))
    fused4, step4, fused16 = _make(n, lanes_per_robot=4, **kw), _make(n, lanes_per_robot=4, **kw), _make(n, lanes_per_robot=16, **kw)
    for e in (fused4, step4, fused16):
        e.reset(ETG_w=Wn, ETG_b=Bn)
    ret4, ln4 = fused4.rollout_policy(pol, steps, 0.3, prec, fused=True)
    ret16, ln16 = fused16.rollout_policy(pol, steps, 0.3, prec, fused=True)
    for _ in range(steps):
        view = step4.obs[:, 3:] if variant == "student" else step4.obs
        step4.step(pol.predict(view.contiguous(), 0.3, prec), want_info=False)
    rets, lns = step4.episode_stats()
    s4, ss, s16 = (e.get_state().cpu().numpy() for e in (fused4, step4, fused16))
    # the oracle ensemble's closed loop (every member on its own observations).  bf16 operands: the members' actions carry a
    # relative noise of 2^-8, the input uncertainty of that arithmetic; otherwise +-1 fp32 ulp
    orc = _ensemble(m, terrain=1 if hf else 0, heightfield=hf, rel_noise=2.0 ** -8 if variant == "bf16" else 0.0)
    if hf:
        orc.set_heightfield(hf["heights"])
    orc.set_params(etg_w=W, etg_b=B)
    orc.reset()
    sq = np.zeros(m)
    for _ in range(steps):
        orc.closed_loop_step(ws, 0.3, col0=3 if variant == "student" else 0)
        sq = np.maximum(sq, orc.spread(slice(13, 25)))
    sqn = np.tile(sq, n // m)
    tol = 5e-3 if variant == "bf16" else 1e-4     # (bf16 operands: an opt-in arithmetic with its own tolerances; SURVEY 8d: 2e-2 on the actor's output)
    # two fp32 evaluations of one trajectory are as far apart as the trajectory is sensitive: the kernel-to-kernel gaps are held
    # to the same per-robot criterion as the gap to the oracle
    sens_robots(np.abs(s4 - ss)[:, 13:25].max(1), sqn, tol, "4-lane closed loop %s: joint gap to predict+step, 12 steps" % variant)
    sens_robots(np.abs(s4 - s16)[:, 13:25].max(1), sqn, tol, "4-lane closed loop %s: joint gap to the 16-lane fused kernel" % variant)
    one = sqn < 1e-5
    assert np.array_equal(ln4.cpu().numpy()[one], lns.cpu().numpy()[one])
    dr = np.abs(ret4.cpu().numpy() - rets.cpu().numpy())[one]
    assert np.all(dr < (0.5 if variant == "bf16" else 5e-3) * (1 + np.abs(rets.cpu().numpy()).max()) * 1e-1 + 5e-3)
    assert np.array_equal(s4[:m], s4[m:])                       # copies of the sample: batch invariance across workgroups
    eq = np.abs(s4[:m] - orc.get_state())[:, 13:25].max(1)
    _say("4-lane closed loop %s vs oracle: q err median %.2e max %.2e" % (variant, np.median(eq), eq.max()))
    sens_robots(eq, sq, 5e-2 if variant == "bf16" else 2e-4, "4-lane closed loop %s vs oracle: joint angles" % variant)
    if variant != "bf16":
        assert np.median(eq) < 1e-5
    with pytest.raises(Exception):
        odd = _make(96, lanes_per_robot=4)                      # not a multiple of 64: the C-ABI refuses, env falls back to stepping
        odd.reset()
        import ctypes as C
        from paddlerobotics_amd import _lib
        _lib.check(odd._lib.etg_rollout_policy(odd._h, pol._h, 2, C.c_float(0.3), 0, 0, C.c_void_p(odd.obs.data_ptr()), None, None, None))
    for e in (fused4, step4, fused16):
        e.close()


def test_long_horizon_statistics_on_the_stairs_task():
    """The reference's default task (`stairstair`, train.py:462; 16 banded stair variants drawn from its STEP_HEIGHT / STEP_WIDTH
    ranges), 4096 robots x 400 control steps under the residual rule against 512 fp64 oracle robots on the same bands."""
    _need_gpu()
    n, m, steps = 4096, 512, 400
    w, b = _population(n)
    env = _make(n, task="stairstair")
    env.reset(ETG_w=w, ETG_b=b)
    hf = env.terrain
    orc = _oracle(m, terrain=1, heightfield=hf)
    orc.threads = NCPU
    orc.set_heightfield(hf["heights"])
    orc.set_params(etg_w=w[:m].double().cpu().numpy(), etg_b=b[:m].double().cpu().numpy())
    orc.reset()
    s = _stats_vs_oracle(env, orc, steps, m, "stairstair, rule")
    # the stair treads are flat, so trajectories stay together: measured gap 0.002, same length +-1 for 99.2 %, KS 0.002 / 0.004 / 0.03
    # (round 5, body spheres colliding: agreement 0.93 -- the last step of a robot kneeling on a stair edge hangs on one grip)
    assert s["gap"] < 0.02 and s["gap_full"] < 0.08 and s["agree"] > 0.88
    assert s["ks_len"] < 0.03 and s["ks_ret"] < 0.03 and s["ks_dx"] < 0.1
    assert s["z"] < 3.0
    env.close()


def test_next_episode_dynamics_are_prepared_ahead():
    """random_param['random_dynamics'] under auto_reset (train.py:253 at scale): the dynamic parameters of every robot's NEXT
    episode are drawn and settled in one launch (etg_prepare_next_dynamics) while the robots run on their current ones; the
    in-kernel restart of etg_step_autoreset installs them.  (1) all robots forced to finish at one step restart on the prepared
    rows: their trajectories equal a twin env reset with exactly these rows (settle simulated from scratch) -- and differ from a
    robot with the old rows; (2) the pending flags: consumed by the restart, refilled by the next refresh, only for the consumed
    robots; (3) both mappings; (4) the stepping cost stays that of the fused auto-reset step."""
    import ctypes as C, time
    from paddlerobotics_amd import _lib as L
    _need_gpu()
    n = 256
    for lanes in (16, 4):
        env = _make(n, auto_reset=True, random_param={"random_dynamics": 1}, random_dynamics_refresh=1000, seed=5, lanes_per_robot=lanes)
        env.reset()
        assert env._nx_on
        pend = torch.empty(n, dtype=torch.uint8, device="cuda:0")
        # robots in the first 64 ticks of their episode still read their pre-reset history from the settle cache that the call
        # replaces: the library leaves them out, so a call right after the reset prepares nothing ...
        assert env._prepare_next_dynamics(None)
        L.check(env._lib.etg_next_dynamics_pending(env._h, C.c_void_p(pend.data_ptr()), env._stream()))
        assert bool((pend == 0).all())
        # ... and their first delayed observations are those of a twin that never called it (same rows, same kernels: bit for bit)
        same = _make(n, auto_reset=True, random_param={"random_dynamics": 1}, random_dynamics_refresh=1000, seed=5, lanes_per_robot=lanes)
        same.reset()
        for _ in range(env._nx_first):            # the env's own first call comes when every robot is old enough
            o1 = env.step(None, want_info=False)[0].clone()
            o2 = same.step(None, want_info=False)[0]
            assert torch.equal(o1, o2)
        same.close()
        rows_next = env._nx_rows.clone()
        L.check(env._lib.etg_next_dynamics_pending(env._h, C.c_void_p(pend.data_ptr()), env._stream()))
        assert bool((pend == 1).all())
        for _ in range(3):
            env.step(None, want_info=False)
        obs, r, d, _ = env.step(None, donef=True)            # every episode ends here: restart on the prepared rows
        assert bool(d.all())
        L.check(env._lib.etg_next_dynamics_pending(env._h, C.c_void_p(pend.data_ptr()), env._stream()))
        assert bool((pend == 0).all())
        twin = _make(n, seed=5, lanes_per_robot=lanes)        # the same rows, installed the slow way: set_params + simulated settle
        obs_t, _ = twin.reset(dynamic_param=rows_next)
        old = _make(n, seed=5, lanes_per_robot=lanes)
        old.reset()                                           # default rows: what the robots would do had nothing been installed
        _lt((obs - obs_t).abs().max().item(), 2e-5, "lanes %d next-episode dynamics: restart observation vs a reset with the same rows" % lanes)
        for _ in range(12):
            env.step(None, want_info=False); twin.step(None, want_info=False); old.step(None, want_info=False)
        se, st, so = env.get_state(), twin.get_state(), old.get_state()
        _lt((se - st)[:, 13:25].abs().max().item(), 2e-4, "lanes %d next-episode dynamics: joints 12 steps after the restart vs the twin" % lanes)
        assert (se - so)[:, 13:25].abs().max().item() > 5e-3, "the installed rows must matter"
        # a second restart before any refresh: the same rows again (exactly the first restart's observation)
        obs2, _, d2, _ = env.step(None, donef=True)
        _lt((obs2 - obs).abs().max().item(), 1e-6, "lanes %d next-episode dynamics: second restart on the same rows" % lanes)
        env.close(); twin.close(); old.close()
        # (2) refresh every 4 steps: only consumed robots get new rows, and they are pending again afterwards
        env = _make(n, auto_reset=True, random_param={"random_dynamics": 1}, random_dynamics_refresh=4, seed=6, lanes_per_robot=lanes)
        env.reset()
        for _ in range(env._nx_first):
            env.step(None, want_info=False)                   # the first call: every robot is old enough, all get rows
        L.check(env._lib.etg_next_dynamics_pending(env._h, C.c_void_p(pend.data_ptr()), env._stream()))
        assert bool((pend == 1).all())
        df = torch.zeros(n, dtype=torch.uint8, device="cuda:0"); df[::3] = 1
        env.step(None, donef=df, want_info=False)             # a third of the robots restart (step 1 of the refresh period)
        L.check(env._lib.etg_next_dynamics_pending(env._h, C.c_void_p(pend.data_ptr()), env._stream()))
        assert torch.equal(pend == 0, df.bool())
        for _ in range(3):
            env.step(None, want_info=False)                   # step 4: a refresh, but the restarted robots are 39 ticks old: left out
        L.check(env._lib.etg_next_dynamics_pending(env._h, C.c_void_p(pend.data_ptr()), env._stream()))
        assert torch.equal(pend == 0, df.bool()) and torch.equal(env._nx_mask.bool(), df.bool())
        for _ in range(4):
            env.step(None, want_info=False)                   # step 8: the next refresh takes them (91 ticks old)
        L.check(env._lib.etg_next_dynamics_pending(env._h, C.c_void_p(pend.data_ptr()), env._stream()))
        assert bool((pend == 1).all())
        assert torch.isfinite(env.get_state()).all()
        env.close()
    # (4) cost: random dynamics + auto reset steps at the fused auto-reset cost (the masked-reset path: ~4 ms per step at 4096)
    N = 4096
    env = _make(N, auto_reset=True, random_param={"random_dynamics": 1}, seed=7)
    env.reset()
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    acts = [(torch.rand(N, 12, device="cuda:0", generator=g) * 2 - 1) * 0.6 for _ in range(8)]
    for k in range(40):
        env.step(acts[k % 8], want_info=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(512):
        env.step(acts[k % 8], want_info=False)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 512 * 1e6
    _say("random dynamics + auto reset, 4096 robots, violent actions: %.1f us per env.step() (two refreshes of the next-episode rows included)" % per)
    assert per < 1000.0 and env._nx_on          # (the masked-reset path measures 3900 us)
    env.close()


def test_auto_reset_keeps_the_start_jitter_on_both_paths():
    """reset(x_noise=1) jitters the start positions; robots restarted by step(auto_reset) start over where their last reset put
    them -- inside the fused launch (checked on the positions) and through the masked-reset path (random dynamics drawn at every
    reset: the settle under new dynamics moves the robot by itself, so the check is that the offsets are not cleared)."""
    _need_gpu()
    n = 64
    env = _make(n, auto_reset=True, seed=3)
    env.reset(x_noise=1)
    x0 = env.get_state()[:, 0].clone()
    assert x0.abs().max().item() > 0.02 and x0.std().item() > 0.01
    for _ in range(3):
        env.step(None, want_info=False)
    env.step(None, donef=True, want_info=False)
    assert (env.get_state()[:, 0] - x0).abs().max().item() < 1e-4
    env.close()
    env = _make(n, auto_reset=True, seed=3, random_param={"random_dynamics": 1}, random_dynamics_refresh=1)
    env.reset(x_noise=1)
    calls = []
    orig = env.set_reset_offsets
    env.set_reset_offsets = lambda xy, ids=None: (calls.append(xy is None), orig(xy, ids))[1]
    env.step(None, donef=True, want_info=False)
    assert calls == [] and torch.isfinite(env.get_state()).all()
    env.reset(env_ids=torch.ones(n, dtype=torch.bool, device="cuda:0"))        # a caller's own masked reset does go back to nominal
    assert calls == [True]
    env.close()


def test_fast_auto_reset_path_comes_back_after_masked_resets():
    """The one-launch restart of etg_step_autoreset (and etg_prepare_next_dynamics) needs every robot to have a cached settle.
    New dynamic parameters for SOME robots invalidate theirs; once masked resets have settled exactly those robots again, the
    library notices (a count of uncached robots reported under the invalidation's sequence number) and returns to the fast path
    -- it used to stay on step + masked reset until the next full reset."""
    import ctypes as C
    from paddlerobotics_amd import _lib as L
    _need_gpu()
    n = 64
    env = _make(n, auto_reset=True, seed=2)
    env.reset()
    rows = torch.as_tensor(np.tile(A.default_dynamic_row(), (n, 1)), dtype=torch.float32, device="cuda:0").contiguous()
    prep = lambda: env._lib.etg_prepare_next_dynamics(env._h, C.c_void_p(rows.data_ptr()), None, env._stream())
    assert prep() == 0                                            # all cached after the full reset
    half = torch.zeros(n, dtype=torch.bool, device="cuda:0"); half[::2] = True
    heavy = rows.clone(); heavy[:, 2] *= 1.2                      # a heavier trunk for half of the robots
    env.set_dynamic_param(heavy, half)
    assert prep() != 0 and b"cached settle" in env._lib.etg_last_error()
    df = half.to(torch.uint8)
    env.step(None, donef=df, want_info=False)                     # the general path: step, then a masked reset that settles them
    torch.cuda.synchronize()
    env.step(None, want_info=False)                               # reads the report: every robot is cached again
    assert prep() == 0
    # a second invalidation that no reset has covered yet keeps the general path
    env.set_dynamic_param(heavy, ~half)
    env.step(None, want_info=False)
    torch.cuda.synchronize()
    assert prep() != 0
    assert torch.isfinite(env.get_state()).all()
    env.close()
