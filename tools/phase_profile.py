"""Per-phase cycle breakdown of one wave of k_step (debug build with -DETG_PROFILE_PHASES)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ETG_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_variants", "lib_prof.so")
import numpy as np, torch
from paddlerobotics_amd.env import make_env
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
# a second argument switches the action filter on: the all-options (non-PLAIN) instantiation with the same physics
kw = dict(enable_action_filter=True) if len(sys.argv) > 2 else {}
env = make_env("Quadrupedal", num_envs=4096, device="cuda:0", lanes_per_robot=lanes, **kw)
env.reset()
for _ in range(20): env.step(None)
env.step(None, want_info=True); torch.cuda.synchronize()
row = env.info_buf[0].cpu().numpy()
# prof[k] = cycles between phase marker k-1 and marker k of physics_tick16 / physics_tick (prof[0]: from the end of the previous tick)
names = ["integration + ring + PD + trig + link inertias", "RNEA", "CRBA / H^-1 / P", "Schur + LDL^T + solve", "unconstrained velocity", "contact rows + Z", "Delassus rows", "row velocities + warm start", "PGS sweeps", "apply impulses (+ joint stops)", "-"]
tot = row[16]
print("lanes_per_robot", lanes)
print("total cycles per control step (wave 0): %.0f  (%.1f us at 2.4 GHz)" % (tot, tot / 2400))
for k in range(10):
    print("  %-48s %9.0f cycles  %5.1f %%  (%.0f per tick)" % (names[k], row[k], 100 * row[k] / tot, row[k] / 13))
print("  outside ticks          %9.0f cycles  %5.1f %%" % (tot - row[:10].sum(), 100 * (tot - row[:10].sum()) / tot))
