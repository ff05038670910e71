"""per-launch fixed cost of the step kernel: time vs ticks per control step (action_repeat) -> slope and intercept"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env
N = 4096
res = {}
for rep in (1, 2, 4, 13, 26):
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0", action_repeat=rep)
    env.reset()
    for _ in range(20): env.step(None, want_info=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): env.step(None, want_info=False)
    torch.cuda.synchronize(); step_us = (time.perf_counter() - t0) / 300 * 1e6
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.rollout_openloop(300)
    torch.cuda.synchronize(); fused_us = (time.perf_counter() - t0) / 300 * 1e6
    res[rep] = (step_us, fused_us)
    print("action_repeat %2d: env.step %.1f us/step, fused rollout %.1f us/step" % (rep, step_us, fused_us))
    env.close()
(a1, f1), (a13, f13) = res[1], res[13]
print("per tick: %.2f us (stepwise), %.2f us (fused); per-step fixed cost: %.1f us (stepwise), %.1f us (fused)" %
      ((a13 - a1) / 12, (f13 - f1) / 12, a1 - (a13 - a1) / 12, f1 - (f13 - f1) / 12))
