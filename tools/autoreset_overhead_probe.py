"""What step(auto_reset) costs when nobody restarts (the default gait walks on for 400 steps) and when many do: env.step(None)
per control step, 4096 robots.  One MI355X."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env
N = 4096
def run(name, act_amp, **kw):
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0", **kw)
    env.reset()
    g = torch.Generator(device="cuda:0"); g.manual_seed(0)
    acts = [None] if act_amp == 0 else [(torch.rand(N, 12, device="cuda:0", generator=g) * 2 - 1) * act_amp for _ in range(16)]
    for k in range(40): env.step(acts[k % len(acts)], want_info=False)
    env.reset(); torch.cuda.synchronize(); t0 = time.perf_counter(); nd = 0
    for k in range(300):
        _, _, d, _ = env.step(acts[k % len(acts)], want_info=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300 * 1e6
    ret, ln = env.episode_stats()
    print("%-44s %6.1f us per step   (mean episode length now %.0f)" % (name, dt, ln.float().mean().item()), flush=True)
    env.close()
run("zero actions, no auto_reset", 0)
run("zero actions, auto_reset", 0, auto_reset=True)
run("+-0.15 rad actions, no auto_reset", 0.15)
run("+-0.15 rad actions, auto_reset", 0.15, auto_reset=True)
run("+-0.3 rad actions, no auto_reset", 0.3)
run("+-0.3 rad actions, auto_reset", 0.3, auto_reset=True)
