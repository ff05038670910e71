"""Where env.step() spends what the fused rollout does not (4096 robots, one MI355X): kernel durations from HIP events around
single launches (k_step16 alone, back to back), against the wall time per step and the fused per-step time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from paddlerobotics_amd.env import make_env
N = 4096
w, b = bench.etg_population(N, 0, "cuda:0")
env = make_env("Quadrupedal", num_envs=N, device="cuda:0")
env.reset(ETG_w=w, ETG_b=b)
for _ in range(100): env.step(None, want_info=False)
# (1) wall per step, back to back
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(400): env.step(None, want_info=False)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 400 * 1e6
# (2) one event pair around 400 back-to-back steps (device time incl. the gaps between dependent launches)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(400): env.step(None, want_info=False)
e1.record(); torch.cuda.synchronize(); dev = e0.elapsed_time(e1) / 400 * 1e3
# (3) event pairs around SINGLE launches with an idle device in between (kernel alone: no queueing behind a predecessor)
single = []
for _ in range(50):
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); env.step(None, want_info=False); c.record(); torch.cuda.synchronize()
    single.append(a.elapsed_time(c) * 1e3)
single.sort()
env.reset(ETG_w=w, ETG_b=b); env.rollout_openloop(50)
e0.record(); env.rollout_openloop(400); e1.record(); torch.cuda.synchronize(); fused = e0.elapsed_time(e1) / 400 * 1e3
print("env.step(): wall %.2f us per step | device time per step, back to back %.2f us | a single launch on an idle device: median %.2f us (min %.2f)" % (wall, dev, single[25], single[0]))
print("fused rollout: %.2f us per control step" % fused)
