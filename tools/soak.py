"""Soak run on one GPU: auto-reset envs driven by violent random actions for tens of thousands of control steps, with the options
that switch contacts most (random heightfield, body contacts, random pushes, randomised dynamics).  Everything must stay finite,
finished robots must keep restarting, and the episode-length distribution must not drift between the first and the last quarter
of the run (a leak of state across restarts would show there)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = 4096
rng = np.random.default_rng(0)
hf = dict(heights=rng.uniform(0, 0.05, size=(256, 256)).astype(np.float32), cell=0.05, origin=(-6.4, -6.4))
cases = [("flat, 16 lanes, pushes + random dynamics", dict(lanes_per_robot=16, random_param={"random_dynamics": 1, "random_force": 1})),
         ("heightfield + body_contacts 2, 16 lanes", dict(lanes_per_robot=16, task="heightfield", heightfield=hf, body_contacts=2)),
         ("heightfield + body_contacts 3, 4 lanes", dict(lanes_per_robot=4, task="heightfield", heightfield=hf, body_contacts=3)),
         ("stairs, 4 lanes, action filter + noise", dict(lanes_per_robot=4, task="stairstair", enable_action_filter=True,
                                                         observation_noise_stdev=[0.02, 0.3, 0.0, 0.01, 0.05]))]
bad = 0
for name, kw in cases:
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0", auto_reset=True, seed=1, **kw)
    g = torch.Generator(device="cuda:0"); g.manual_seed(2)
    env.reset(x_noise=1)
    finished = torch.zeros((), device="cuda:0")
    lens = [[], []]
    cur = torch.zeros(N, device="cuda:0")
    ok = torch.ones((), dtype=torch.bool, device="cuda:0")
    t0 = time.perf_counter()
    for k in range(STEPS):
        a = (torch.rand(N, 12, device="cuda:0", generator=g) * 2 - 1) * (0.6 if k % 200 < 150 else 0.05)
        obs, rew, done, info = env.step(a, want_info=False)
        cur += 1
        if k < STEPS // 4 or k >= 3 * STEPS // 4:
            lens[0 if k < STEPS // 4 else 1].append(cur[done.view(-1).bool()].clone())
        cur = torch.where(done.view(-1).bool(), torch.zeros_like(cur), cur)
        finished += done.sum()
        if k % 500 == 0:
            ok &= torch.isfinite(obs).all() & torch.isfinite(rew).all()
    ok &= torch.isfinite(env.get_state()).all()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    l0, l1 = torch.cat(lens[0]).float(), torch.cat(lens[1]).float()
    drift = abs(l0.mean().item() - l1.mean().item()) / max(l0.mean().item(), 1.0)
    good = bool(ok.item()) and finished.item() > N and drift < 0.1
    print("%-4s %-46s %d steps x %d robots in %.1f s (%.0f M env-steps/s through env.step): %d episodes finished, mean length first quarter %.1f "
          "last quarter %.1f (drift %.1f %%)" % ("ok" if good else "BAD", name, STEPS, N, dt, STEPS * N / dt / 1e6, int(finished.item()),
                                                 l0.mean().item(), l1.mean().item(), 100 * drift), flush=True)
    bad += not good
    env.close()
print("soak:", "clean" if bad == 0 else "%d cases failed" % bad)
sys.exit(1 if bad else 0)
