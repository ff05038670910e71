"""Generate tests/golden/*.npz from the pieces of the reference that can be executed here.

Run ONCE in the build container (needs /root/reference; never on the GPU box):
    python tests/golden/make_golden.py
Only DATA is written (inputs + expected outputs).  Functions are executed from the
reference tree where they lie (AST extraction of pure functions / plain imports);
no reference source text is stored.

Provenance (paths under /root/reference/QuadrupedalRobots/ETGRL):
  kin.npz     a1.py:97-173 foot_position_in_hip_frame_to_joint_angle / foot_position_in_hip_frame /
              analytical_leg_jacobian / foot_positions_in_base_frame on seeded random inputs
  pd.npz      deployment/robots/laikago_motor.py:103-175 convert_to_torque (POSITION + TORQUE modes)
  etg.npz     gait_action_list_ETG_exp.npy and deployment/exp/stairstair/gait_action_list_CPG_*.npy
              (sub-sampled rows) + the (W,b) least-squares fit of the RBF model to them
  opt.npz     train.py:59-126 LS_sol / Opt_with_points / param2dynamic_dict outputs
  mlp.npz     deployment/exp/stairstair/StairStair3_BC1_itr_500383.pt actor weights -> outputs
              (plain torch fp32, model/mujoco_model.py:44-60 semantics)
  ga.npz      alg/es.py SimpleGA seeded ask/tell trace
  filter.npz  deployment/robots/action_filter.py ActionFilterButter (absl stubbed) trace
"""
import ast
import os
import sys
import types

import numpy as np

REF = "/root/reference/QuadrupedalRobots/ETGRL/"
OUT = os.path.dirname(os.path.abspath(__file__))


def extract(path, names, assigns=()):
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"np": np, "math": __import__("math"), "copy": __import__("copy").copy}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") in assigns:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    return ns


def main():
    rng = np.random.default_rng(20260926)

    # ---- kinematics ---------------------------------------------------------------
    a1 = extract(REF + "deployment/robots/a1.py",
                 {"foot_position_in_hip_frame_to_joint_angle", "foot_position_in_hip_frame",
                  "analytical_leg_jacobian", "foot_positions_in_base_frame"},
                 ("COM_OFFSET", "HIP_OFFSETS"))
    n = 64
    ang = np.stack([rng.uniform(-0.6, 0.6, n), rng.uniform(-0.5, 2.0, n), rng.uniform(-2.6, -1.0, n)], 1)
    legs = rng.integers(0, 4, n)
    fk = np.stack([a1["foot_position_in_hip_frame"](ang[i], (-1.0) ** (legs[i] + 1)) for i in range(n)])
    ik = np.stack([a1["foot_position_in_hip_frame_to_joint_angle"](fk[i], (-1.0) ** (legs[i] + 1)) for i in range(n)])
    jac = np.stack([a1["analytical_leg_jacobian"](ang[i], int(legs[i])) for i in range(n)])
    ang12 = np.concatenate([ang[:16], ang[16:32], ang[32:48], ang[48:64]], axis=1)
    fbase = np.stack([a1["foot_positions_in_base_frame"](ang12[i]) for i in range(16)])
    np.savez(os.path.join(OUT, "kin.npz"), ang=ang, legs=legs, fk=fk, ik=ik, jac=jac, ang12=ang12,
             fbase=fbase, hip_offsets=a1["HIP_OFFSETS"])

    # ---- PD motor model -----------------------------------------------------------
    sys.path.insert(0, REF + "deployment")
    from robots import laikago_motor, robot_config  # noqa: E402
    kp = rng.uniform(20, 200, 12)
    kd = rng.uniform(0, 5, 12)
    mm = laikago_motor.LaikagoMotorModel(kp=kp, kd=kd, motor_control_mode=robot_config.MotorControlMode.POSITION)
    q = rng.normal(size=(32, 12))
    qd = rng.normal(size=(32, 12)) * 3
    qdes = rng.normal(size=(32, 12))
    tau = np.stack([mm.convert_to_torque(qdes[i], q[i], qd[i], qd[i], robot_config.MotorControlMode.POSITION)[0]
                    for i in range(32)])
    tq = np.stack([mm.convert_to_torque(qdes[i], q[i], qd[i], qd[i], robot_config.MotorControlMode.TORQUE)[0]
                   for i in range(32)])
    mm1 = laikago_motor.LaikagoMotorModel(kp=100.0, kd=np.array([1, 2, 2] * 4.0 if False else [1., 2., 2.] * 4),
                                          motor_control_mode=robot_config.MotorControlMode.POSITION)
    known = mm1.convert_to_torque(np.array([0, 0.9, -1.8] * 4), np.zeros(12), np.ones(12), np.ones(12),
                                  robot_config.MotorControlMode.POSITION)[0]
    np.savez(os.path.join(OUT, "pd.npz"), kp=kp, kd=kd, q=q, qd=qd, qdes=qdes, tau=tau, tau_torque_mode=tq,
             known=known)

    # ---- ETG fixtures -------------------------------------------------------------
    g1 = np.load(REF + "gait_action_list_ETG_exp.npy")
    g2 = np.load(REF + "deployment/exp/stairstair/gait_action_list_CPG_stairstair7_12_3.npy")
    pose = np.array([0, 0.9, -1.8] * 4)
    base_foot = np.array([0.18, -0.15, -0.23, 0.18, 0.148, -0.23, -0.18, -0.14, -0.23, -0.18, 0.135, -0.23]).reshape(4, 3)
    T, H, amp, sig, dt = 0.5, 20, 0.2, 0.04, 0.026
    phase = np.array([-np.pi / 2, 0])

    def fwd(t):
        return amp * np.sin(phase + t * 2 * np.pi / T)
    u = np.array([fwd(h * T / (H - 0.9)) for h in range(H)])

    def rbf(t):
        return np.exp(-np.sum((fwd(t) - u) ** 2, axis=1) / sig)
    out = {}
    for name, g, off in (("exp", g1, 1), ("stair", g2, 0)):
        fp = np.array([a1["foot_positions_in_base_frame"](pose + r) for r in g])
        d = fp - base_foot
        ts = (np.arange(len(g)) + off) * dt
        Amat = np.array([np.append(rbf(t), 1.0) for t in ts])
        W = np.zeros((3, H))
        b = np.zeros(3)
        for dim in (0, 2):
            sol = np.linalg.lstsq(Amat, d[:, 0, dim], rcond=None)[0]
            W[dim], b[dim] = sol[:H], sol[H]
        rows = np.arange(0, len(g), 7)
        out[name + "_rows"] = rows
        out[name + "_t"] = ts[rows]
        out[name + "_act"] = g[rows]
        out[name + "_w"] = W
        out[name + "_b"] = b
        out[name + "_maxres"] = np.abs(Amat @ np.append(W[0], b[0]) - d[:, 0, 0]).max()
    out["rbf_t"] = np.array([0.0, 0.013, 0.1, 0.26, 0.35, 0.49, 1.3])
    out["rbf"] = np.array([rbf(t) for t in out["rbf_t"]])
    np.savez(os.path.join(OUT, "etg.npz"), **out)

    # ---- Opt_with_points / LS_sol / param2dynamic_dict ------------------------------
    tr = extract(REF + "train.py", {"LS_sol", "Opt_with_points", "param2dynamic_dict"})

    class _ETG:  # minimal stand-in for the absent rlschool ETG_layer: only .update(t) is used (train.py:91)
        def update(self, t):
            return rbf(t)
    w0, b0, pts = tr["Opt_with_points"](ETG=_ETG(), ETG_T=0.5, Footheight=0.1, Steplength=0.05)
    dpts = rng.normal(size=(4, 6, 2)) * 0.02
    w_l, b_l = [], []
    for k in range(4):
        w, b, _ = tr["Opt_with_points"](ETG=_ETG(), ETG_T=0.5, w0=w0, b0=b0, points=pts + dpts[k])
        w_l.append(w)
        b_l.append(b)
    A_ls = rng.normal(size=(6, 20)) * 0.3
    b_ls = rng.normal(size=(6, 1)) * 0.05
    x_ls = tr["LS_sol"](A_ls, b_ls, precision=1e-4, alpha=0.05)
    x_ls_w0 = tr["LS_sol"](A_ls, b_ls, precision=1e-4, alpha=0.05, lamb=0.5, w0=np.full((20, 1), 0.01))
    p48 = rng.uniform(-1.3, 1.3, size=(6, 48))
    p48[0] = 0
    dyn = []
    for r in p48:
        d = tr["param2dynamic_dict"](r)
        dyn.append(np.concatenate([np.atleast_1d(d['control_latency']), np.atleast_1d(d['footfriction']),
                                   np.atleast_1d(d['basemass']), d['baseinertia'], d['legmass'], d['leginertia'],
                                   d['motor_kp'], d['motor_kd'], d['gravity']]))
    np.savez(os.path.join(OUT, "opt.npz"), w0=w0, b0=b0, prior=pts, dpts=dpts, w=np.array(w_l), b=np.array(b_l),
             A_ls=A_ls, b_ls=b_ls, x_ls=x_ls, x_ls_w0=x_ls_w0, p48=p48, dyn=np.array(dyn))

    # ---- policy MLP -----------------------------------------------------------------
    import torch
    sd = torch.load(REF + "deployment/exp/stairstair/StairStair3_BC1_itr_500383.pt", map_location="cpu")
    wts = {k: sd["actor_model." + k].float() for k in
           ("l1.weight", "l1.bias", "l2.weight", "l2.bias", "mean_linear.weight", "mean_linear.bias")}
    torch.manual_seed(0)
    obs = torch.cat([torch.zeros(1, 46), torch.randn(31, 46)], 0)
    with torch.no_grad():
        x = torch.relu(obs @ wts["l1.weight"].T + wts["l1.bias"])
        x = torch.relu(x @ wts["l2.weight"].T + wts["l2.bias"])
        act = torch.tanh(x @ wts["mean_linear.weight"].T + wts["mean_linear.bias"])
    np.savez_compressed(os.path.join(OUT, "mlp.npz"), obs=obs.numpy(), act=act.numpy(),
                        **{k.replace(".", "_"): v.numpy() for k, v in wts.items()})

    # ---- SimpleGA trace ---------------------------------------------------------------
    sys.path.insert(0, REF)
    from alg.es import SimpleGA  # noqa: E402
    np.random.seed(123)
    ga = SimpleGA(12, sigma_init=0.02, sigma_decay=0.99, sigma_limit=0.005, elite_ratio=0.1,
                  weight_decay=0.005, popsize=40, param=np.zeros(12))
    trace = {}
    for it in range(3):
        sol = ga.ask()
        fit = -np.sum((sol - 0.05) ** 2, axis=1) * 100 + np.arange(40) * 1e-3
        ga.tell(fit)
        trace["sol%d" % it] = sol
        trace["fit%d" % it] = fit
        trace["elite%d" % it] = ga.elite_params.copy()
        trace["elite_rewards%d" % it] = ga.elite_rewards.copy()
        trace["best%d" % it] = ga.best_param.copy()
        trace["sigma%d" % it] = np.array(ga.sigma)
    np.savez(os.path.join(OUT, "ga.npz"), **trace)

    # ---- Butterworth action filter ------------------------------------------------------
    absl = types.ModuleType("absl")
    absl.logging = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None)
    sys.modules["absl"] = absl
    sys.modules["absl.logging"] = absl.logging
    from robots import action_filter  # noqa: E402
    f = action_filter.ActionFilterButter(sampling_rate=1 / 0.026, num_joints=12)
    f.init_history(pose)
    xs = pose + rng.normal(size=(20, 12)) * 0.2
    ys = np.stack([f.filter(x) for x in xs])
    np.savez(os.path.join(OUT, "filter.npz"), x=xs, y=ys, init=pose, a=np.asarray(f.a[0]), b=np.asarray(f.b[0]))
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
