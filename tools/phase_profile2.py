"""Per-phase cycle breakdown of k_step16 on the walking workload of configs[1] (debug build with -DETG_PROFILE_PHASES: gpurun_variants/lib_prof.so).
The kernel reports the counters of robot 0's wave; to sample several waves the population is rotated between runs.
usage: phase_profile2.py [body_contacts] [first_step] [n_samples]"""
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
w, b = bench.etg_population(N, 0, "cuda:0")
names = ["integration + ring + PD + trig + link inertias", "RNEA", "CRBA / H^-1 / P", "Schur + LDL^T + solve", "unconstrained velocity",
         "contact rows + Z (+ body candidates)", "Delassus rows", "row velocities + warm start", "PGS sweeps (feet + body normals)", "apply impulses",
         "body normal columns (Ak)", "body friction phases", "body friction build (build_b)"]
acc = np.zeros(17); sweeps = 0.0
env = make_env("Quadrupedal", num_envs=N, device="cuda:0", lanes_per_robot=16, body_contacts=bc)
for s in range(nsamp):
    sh = 4 * 37 * s
    env.reset(ETG_w=torch.roll(w, sh, 0), ETG_b=torch.roll(b, sh, 0))
    for _ in range(first + 3 * s): env.step(None)
    _, _, _, info = env.step(None, want_info=True); torch.cuda.synchronize()
    acc += env.info_buf[0, :17].cpu().numpy().astype(np.float64)
    sweeps += float(info["solver_sweeps"].float().mean().item()) / 13.0
acc /= nsamp
tot = acc[16]
print("body_contacts", bc, "| control steps %d.. after reset, %d sampled waves | executed sweeps per tick (all waves) %.2f" % (first, nsamp, sweeps / nsamp))
print("total cycles per control step: %.0f  (%.1f us at 2.4 GHz)" % (tot, tot / 2400))
for k in range(13):
    print("  %-52s %9.0f cycles  %5.1f %%  (%.0f per tick)" % (names[k], acc[k], 100 * acc[k] / tot, acc[k] / 13))
print("  outside ticks %9.0f cycles  %5.1f %%" % (tot - acc[:13].sum(), 100 * (tot - acc[:13].sum()) / tot))
