"""Per-phase cycle breakdown of k_step16 on the walking workload of configs[1] (debug build with -DETG_PROFILE_PHASES: gpurun_variants/lib_prof.so).
The kernel reports the counters of robot 0's wave; to sample several waves the population is rotated between runs.
usage: phase_profile2.py [body_contacts] [first_step] [n_samples] [hf]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ETG_LIB"] = os.path.join(ROOT, "gpurun_variants", "lib_prof.so")
import numpy as np, torch
from paddlerobotics_amd.env import make_env
import bench
bc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
first = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nsamp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
N = 4096
hf_kw = {}
if len(sys.argv) > 4 and sys.argv[4] == "hf":      # BASELINE config 5's terrain
    hf_kw = dict(task="heightfield", heightfield=dict(heights=np.random.default_rng(0).uniform(0.0, 0.05, size=(256, 256)).astype(np.float32), cell=0.05, origin=(-6.4, -6.4)))
w, b = bench.etg_population(N, 0, "cuda:0")
names = ["integration + ring + PD + trig + link inertias", "RNEA", "CRBA / H^-1 / P", "Schur + LDL^T + solve", "unconstrained velocity",
         "contact rows + Z (+ body candidates)", "Delassus rows", "row velocities + warm start", "PGS sweeps (feet + body normals)", "apply impulses",
         "body normal columns (Ak)", "body friction phases", "body friction build (build_b)"]
env = make_env("Quadrupedal", num_envs=N, device="cuda:0", lanes_per_robot=16, body_contacts=bc, **hf_kw)
env.reset(ETG_w=w, ETG_b=b)
for _ in range(first): env.step(None)
rows = []
for s in range(nsamp):
    _, _, _, info = env.step(None, want_info=True); torch.cuda.synchronize()
    rows.append(env.info_buf[0::4, :17].cpu().numpy().astype(np.float64))      # one row per wave (its first robot's info row)
R = np.stack(rows)                     # [steps, waves, 17]
tot = R[:, :, 16]
print("body_contacts", bc, "| control steps %d..%d after reset, all %d waves" % (first, first + nsamp - 1, R.shape[1]))
print("cycles per control step and wave: mean %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f  | mean over steps of the slowest wave %.0f (%.1f us at 2.4 GHz)" % (
    tot.mean(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), tot.max(1).mean(), tot.max(1).mean() / 2400))
worst = np.stack([R[s, np.argmax(tot[s])] for s in range(nsamp)]).mean(0)    # the slowest wave of EVERY step (what a Gym step waits for)
mean = R.mean((0, 1))
wsum = tot.sum(0)                                                                # per wave over all sampled steps (what a fused launch waits for)
fused = R[:, np.argmax(wsum)].mean(0)
print("the wave with the largest total over the %d steps: %.0f cycles per step (%.2f x the mean wave); p90 of the waves' totals %.2f x" % (
    nsamp, wsum.max() / nsamp, wsum.max() / wsum.mean(), np.percentile(wsum, 90) / wsum.mean()))
print("  %-52s %12s %14s %16s" % ("cycles per control step", "mean wave", "slowest / step", "slowest in total"))
for k in range(13):
    print("  %-52s %12.0f %14.0f %16.0f" % (names[k], mean[k], worst[k], fused[k]))
print("  %-52s %12.0f %14.0f %16.0f" % ("outside ticks", mean[16] - mean[:13].sum(), worst[16] - worst[:13].sum(), fused[16] - fused[:13].sum()))
print("  %-52s %12.0f %14.0f %16.0f" % ("total", mean[16], worst[16], fused[16]))
