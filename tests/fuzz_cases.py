"""The differential fuzzer's trial generator (test infrastructure): tools/fuzz_parity.py runs batches of trials, the -m gpu
regression tests re-create single ones by (seed, trial) -- tests/test_gpu_regressions.py.

A trial = a random combination of the robot-layer / solver / terrain options, a lane mapping, per-robot ETG parameters, dynamic
rows, strength ratios and pushes.  `setup_trial` draws it, builds the env with make_env() and the oracles FROM THE ENV'S OWN
EtgConfig, installs everything on both sides and resets them; the generator's state after the call is where the trial's action
draws start, so a trial is reproduced bit for bit from (seed, trial, n)."""
import numpy as np
import torch

from paddlerobotics_amd import a1_model as A

NOISE = [0.02, 0.3, 0.0, 0.01, 0.05]


def etg_params(rng, n):
    from paddlerobotics_amd.etg import ETG_layer, Opt_with_points
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w0, b0, prior = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
    W, B = np.zeros((n, 3, 20)), np.zeros((n, 3))
    for i in range(n):
        W[i], B[i], _ = Opt_with_points(layer, ETG_T=0.5, w0=w0, b0=b0, points=prior + 0.02 * rng.normal(size=(6, 2)))
    return W, B


def draw(rng):
    kw = {}
    lanes = int(rng.choice([16, 4]))
    mode = rng.choice(["pose", "pose", "pose", "torque", "hybrid"])
    if mode != "pose":
        kw["motor_control_mode"] = str(mode)
    if rng.random() < 0.25: kw["enable_action_filter"] = True
    if rng.random() < 0.25: kw["enable_action_interpolation"] = True
    if mode == "pose" and rng.random() < 0.25: kw["enable_clip_motor_commands"] = True
    bc = int(rng.choice([0, 0, 0, 1, 2, 3]))
    if bc == 3 and lanes == 16: bc = 2
    kw["body_contacts"] = bc          # (always explicit: the library default is 2)
    if bc in (1, 2) and rng.random() < 0.4: kw["body_friction"] = float(rng.choice([0.0, 0.2, 1.0]))
    if rng.random() < 0.3: kw["joint_limits"] = False
    if rng.random() < 0.3: kw["friction_model"] = 1
    s = rng.random()
    if s < 0.2: kw["solver_iters"] = int(rng.integers(2, 6))
    elif s < 0.35: kw["solver_residual"] = 1e-5
    if rng.random() < 0.3: kw["pd_latency"] = float(rng.choice([0.0005, 0.001, 0.002]))   # (the delayed kd term: marginal from ~2.5 ms -- settles differ by 1e-3 between any two
        # fp32 evaluations -- and a physical blow-up from ~4 ms with the default gains, where both sides go NaN)
    if rng.random() < 0.2: kw.update(warmstart=0.85, warmstart_friction=float(rng.choice([0.0, 0.85])))
    if rng.random() < 0.15: kw["contact_slop"] = 0.0
    if rng.random() < 0.2: kw["foot_restitution"] = float(rng.uniform(0.1, 0.8))
    if rng.random() < 0.25: kw["motor_torque_limits"] = float(rng.uniform(8.0, 30.0))
    if rng.random() < 0.2: kw["ETG"] = 0
    t = rng.random()
    if t < 0.2: kw["task"] = "stairstair"; kw["terrain_seed"] = int(rng.integers(0, 5))
    elif t < 0.4:
        hf = rng.uniform(0.0, 0.04, size=(64, 64)).astype(np.float32)
        kw.update(task="heightfield", heightfield=dict(heights=hf, cell=0.05, origin=(-1.6, -1.6)))
    extras = dict(dyn=rng.random() < 0.5, strength=rng.random() < 0.3, push=rng.random() < 0.3, noise=rng.random() < 0.2, offsets=rng.random() < 0.25)
    if kw.get("task") == "heightfield":
        extras["offsets"] = True       # (identical robots on ONE spot make a terrain trial all-or-nothing: ~20 % of the spots of
                                       # such a field settle > 1e-4 apart between any two fp32 evaluations -- spread them)
    if extras["dyn"] and kw.get("pd_latency", 0.0) > 0.001:
        kw["pd_latency"] = 0.001       # (random kd up to 2.6 on lighter links: the delayed damping term blows up earlier)
    return lanes, kw, extras


def short(kw):
    return {k: (v if k != "heightfield" else "64x64") for k, v in kw.items()}




def trial_rng(seed, trial):
    return np.random.default_rng(1000 * seed + trial)


def setup_trial(seed, trial, N, make_oracles, kw_extra=None):
    """-> dict(rng, lanes, kw, ex, env, orcs, W, B, rows, sr, f, offs, mode, loose).  make_oracles(cfg) -> list of oracle-like objects
    (OracleSim / tests.parity_util.OracleEnsemble) built from a copy of the env's EtgConfig; every installer is called on each."""
    from paddlerobotics_amd.env import make_env
    rng = trial_rng(seed, trial)
    lanes, kw, ex = draw(rng)
    kw.update(kw_extra or {})      # (a regression case pins an option of the trial, e.g. body_blend = 0)
    if ex["noise"]:
        kw["observation_noise_stdev"] = NOISE      # (counter-based draws: the oracle's stream is seeded like the env's)
    offs = rng.uniform(-0.3, 0.3, size=(64, 2)) if ex["offsets"] else None
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0", lanes_per_robot=lanes, seed=trial, **kw)
    cfg = type(env.cfg).from_buffer_copy(env.cfg)
    orcs = make_oracles(cfg)
    if env.terrain is not None:
        for o in orcs: o.set_heightfield(env.terrain["heights"])
    if ex["noise"]:
        for o in orcs: o.set_sensor_noise(NOISE, seed=trial)
    if offs is not None:
        env.set_reset_offsets(torch.as_tensor(offs[:N], dtype=torch.float32))
        for o in orcs: o.set_reset_offsets(offs[:N])
    W = B = None
    if kw.get("ETG", 1):      # (make_env installs the prior gait by default: the oracles get the same per-robot parameters)
        W, B = etg_params(rng, N)
        for o in orcs: o.set_params(etg_w=W, etg_b=B)
    rows = None
    if ex["dyn"]:
        p = torch.as_tensor(rng.uniform(-0.3, 0.3, size=(N, A.DYN_DIM)), dtype=torch.float32)
        rows = A.param2dynamic_rows_torch(p).numpy().astype(np.float64)
        rows[:, 1] = np.maximum(rows[:, 1], 0.05)          # (frictionless feet are legal but chaotic within a step or two)
        env.set_dynamic_param(torch.as_tensor(rows, dtype=torch.float32, device="cuda:0"))
        for o in orcs: o.set_params(dyn=rows)
    sr = None
    if ex["strength"]:
        sr = rng.uniform(0.5, 1.0, size=(N, 12))
        env.set_motor_strength_ratios(torch.as_tensor(sr, dtype=torch.float32))
        for o in orcs: o.set_motor_strength(sr)
    if W is not None:
        env.reset(ETG_w=W, ETG_b=B)
    else:
        env.reset()
    for o in orcs: o.reset()
    f = None
    if ex["push"]:
        f = np.zeros((N, 3)); f[:, :2] = rng.uniform(-15, 15, size=(N, 2))
        env.set_external_force(torch.as_tensor(f, dtype=torch.float32))
        for o in orcs: o.set_external_force(f)
    # (a looser residual threshold stops the sweeps on a last-bit decision worth ~sqrt(threshold) m/s: the floor follows it)
    loose = max(1.0, float(np.sqrt(env.cfg.solver_residual / 1e-7)))
    return dict(rng=rng, lanes=lanes, kw=kw, ex=ex, env=env, orcs=orcs, W=W, B=B, rows=rows, sr=sr, f=f, offs=offs,
                mode=kw.get("motor_control_mode", "pose"), loose=loose)


def trial_action(rng, mode, N, adim):
    """the action rows of one control step of a trial (the order of draws is the trial's)"""
    if mode == "hybrid":
        a = rng.uniform(-1, 1, size=(N, 12, 5))
        a[..., 0] = np.array([0.0, 0.9, -1.8] * 4) + 0.15 * a[..., 0]; a[..., 1] = 80.0; a[..., 2] = 0.0; a[..., 3] = 1.5; a[..., 4] *= 2.0
        return a.reshape(N, 60)
    if mode == "torque":
        return rng.uniform(-4.0, 4.0, size=(N, adim))
    return rng.uniform(-0.2, 0.2, size=(N, adim))
