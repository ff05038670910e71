"""Robustness sweep: every robot-layer option x both lane mappings x flat / stairs, 150 random-action steps each;
checks that observations / rewards / states stay finite and reports survivors.  GPU only."""
import itertools, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env

N = 512
bad = 0
opts = [dict(), dict(motor_control_mode="torque"), dict(motor_control_mode="hybrid"), dict(enable_action_filter=True),
        dict(enable_action_interpolation=True), dict(enable_clip_motor_commands=True),
        dict(observation_noise_stdev=[0.02, 0.3, 0.0, 0.01, 0.05]), dict(random_param={"random_dynamics": 1, "random_force": 1}),
        dict(body_contacts=True), dict(sensor_mode={"dis": 0, "RNN": {"time_steps": 2, "mode": "stack", "time_interval": 1}}),
        # round 2: joint-limit stops, no trajectory generator, the optional sensors, auto-reset inside the step launch
        dict(joint_limits=True), dict(joint_limits=True, body_contacts=True, motor_control_mode="torque"), dict(ETG=0),
        dict(sensor_mode={"ETG_obs": 1, "footpose": 1, "dynamic_vec": 1, "force_vec": 1, "noise": 1}),
        dict(body_contacts=2), dict(auto_reset=True), dict(auto_reset=True, joint_limits=True, random_param={"random_force": 1}),
        # round 3: all body spheres at once (4-lane kernels), pyramid friction, pd latency, the solver with a fixed count
        dict(body_contacts=3), dict(body_contacts=3, motor_control_mode="torque"), dict(friction_model=1), dict(pd_latency=0.001),
        dict(solver_iters=3), dict(body_contacts=2, friction_model=1, auto_reset=True),
        dict(auto_reset=True, random_param={"random_dynamics": 1}, random_dynamics_refresh=32),
        # round 4: torque limits, restitution, Bullet's own warm start on both row kinds, no slop, strength ratios (set below)
        dict(motor_torque_limits=20.0), dict(foot_restitution=0.5), dict(warmstart=0.85, warmstart_friction=0.85, contact_slop=0.0),
        dict(strength=0.7), dict(strength=0.7, auto_reset=True, random_param={"random_dynamics": 1, "random_force": 1}, random_dynamics_refresh=16),
        dict(motor_control_mode="torque", motor_torque_limits=3.0, strength=0.8)]
for lanes, task, o in itertools.product((16, 4), ("ground", "stairstair"), opts):
    if o.get("body_contacts") == 3 and lanes == 16:
        continue   # three body rows per leg: the 4-lane mapping only
    o = dict(o)
    strength = o.pop("strength", None)
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0", lanes_per_robot=lanes, task=task, seed=3, **o)
    if strength is not None:
        env.set_motor_strength_ratios(strength)
        o["strength"] = strength
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    obs, _ = env.reset(x_noise=1)
    adim = env.action_space.shape[0]
    for k in range(150):
        a = (torch.rand(N, adim, device="cuda:0", generator=g) * 2 - 1) * (0.3 if adim == 12 and o.get("motor_control_mode") != "torque" else 1.0)
        if adim == 60:   # (q_des, kp, qd_des, kd, tau_ff) per motor: PD around the standing pose
            a = a.view(N, 12, 5); a[..., 0] = torch.tensor([0.0, 0.9, -1.8] * 4, device="cuda:0") + 0.1 * a[..., 0]
            a[..., 1] = 80.0; a[..., 2] = 0.0; a[..., 3] = 1.5; a[..., 4] *= 2.0; a = a.reshape(N, 60)
        obs, rew, done, info = env.step(a)
        if done.any() and not o.get("auto_reset"):
            env.reset(env_ids=done)
    ok = bool(torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(env.get_state()).all())
    ret, ln = env.episode_stats()
    print("%-5s lanes %2d %-11s %-70s finite=%s mean episode len %.0f" % ("ok" if ok else "BAD", lanes, task, str(o)[:70], ok, ln.float().mean().item()))
    bad += not ok
    env.close()
print("config matrix:", "all finite" if bad == 0 else "%d configurations produced non-finite values" % bad)
sys.exit(1 if bad else 0)
