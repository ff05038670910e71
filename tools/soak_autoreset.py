"""Soak of the round-2 paths: 4096 robots with auto-reset inside the step kernel, knee contacts, joint-limit stops, sensor
noise, random pushes and the stochastic policy head, on flat ground and on the stairs task; then the same with random
dynamics (general reset path).  Checks finiteness, that episodes keep restarting, and reports throughput.  GPU only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env
from paddlerobotics_amd.policy import MfmaPolicy
from paddlerobotics_amd import a1_model as A

N, STEPS = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for name, kw in (("flat, knees + joint limits + noise + pushes", dict(body_contacts=True, joint_limits=True, sensor_mode={"noise": 1},
                                                                   random_param={"random_force": 1})),
                 ("stairs, knees + joint limits", dict(task="stairstair", body_contacts=True, joint_limits=True)),
                 ("flat, random dynamics (general reset path)", dict(random_param={"random_dynamics": 1, "random_force": 1}))):
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0", auto_reset=True, seed=1, **kw)
    pol = MfmaPolicy(A.OBS_DIM, 12)
    pol.load_state_dict(MfmaPolicy.init_like_reference(A.OBS_DIM, 12, seed=0))
    obs, _ = env.reset()
    resets = torch.zeros((), dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(STEPS):
        act, _ = pol.sample(obs, 0.3)
        obs, rew, done, info = env.step(act, donef=(k % 400 == 399))
        resets += done.sum()
        if k % 100 == 99:
            assert torch.isfinite(obs).all() and torch.isfinite(rew).all(), (name, k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert torch.isfinite(env.get_state()).all()
    _, ln = env.episode_stats()
    assert int(ln.max()) <= 400
    print("soak ok [%s]: %d steps x %d robots, %d restarts, %.1f M env-steps/s incl. the policy sample" % (
        name, STEPS, N, int(resets), N * STEPS / dt / 1e6))
    env.close()
