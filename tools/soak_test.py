"""Soak: 4096 robots, random residual actions through the MFMA policy head, auto-reset of finished robots,
external pushes and randomised dynamics on; checks finiteness and reports throughput.  GPU only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env
from paddlerobotics_amd.policy import MfmaPolicy
from paddlerobotics_amd import a1_model as A

N, STEPS = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 2000
env = make_env("Quadrupedal", num_envs=N, device="cuda:0", task="stairstair",
               random_param={"random_dynamics": 1, "random_force": 1}, seed=1)
pol = MfmaPolicy(A.OBS_DIM, 12)
pol.load_state_dict(MfmaPolicy.init_like_reference(A.OBS_DIM, 12, seed=0))
obs, _ = env.reset()
resets = 0
t_reset = 0.0
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(STEPS):
    act, logp = pol.sample(obs, 0.3)
    obs, rew, done, info = env.step(act, donef=(k % 400 == 399))
    if k % 50 == 49:                       # auto-reset of finished robots (host sync only every 50 steps)
        ids = torch.nonzero(done).flatten()
        if ids.numel():
            torch.cuda.synchronize(); tr = time.perf_counter()
            env.reset(env_ids=ids)
            torch.cuda.synchronize(); t_reset += time.perf_counter() - tr
            resets += int(ids.numel())
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(logp).all(), k
torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = env.get_state()
assert torch.isfinite(st).all()
print("soak ok: %d steps x %d robots, %d resets, %.1f M env-steps/s incl. policy sample, resets, pushes "
      "(%.2f s total, %.2f s of it in %d reset calls with re-randomised dynamics)" %
      (STEPS, N, resets, N * STEPS / dt / 1e6, dt, t_reset, STEPS // 50))
