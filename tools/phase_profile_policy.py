"""Per-stage cycles of one wave of k_rollout_policy16 (debug build with -DETG_PROFILE_PHASES, gpurun_variants/lib_prof.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ETG_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_variants", "lib_prof.so")
import numpy as np, torch
from paddlerobotics_amd.env import make_env
from paddlerobotics_amd.policy import MfmaPolicy
precision = int(sys.argv[1]) if len(sys.argv) > 1 else 0   # 0 fp32 MFMA, 1 bf16
env = make_env("Quadrupedal", num_envs=4096, device="cuda:0")
env.reset()
pol = MfmaPolicy(49, 12, 256, "cuda:0")
pol.load_state_dict(MfmaPolicy.init_like_reference(49, 12, 256, seed=0))
steps = 50
env.rollout_policy(pol, steps, 0.3, precision)
env.reset()
env.rollout_policy(pol, steps, 0.3, precision)
torch.cuda.synchronize()
row = env.obs[0].cpu().numpy()
names = ["barrier + observation tile -> LDS", "layer 1 (4 k-blocks) + barrier", "layer 2 (16 k-blocks) + barrier", "head + barrier", "bias / tanh / action -> LDS + barrier", "control step (13 ticks)"]
tot = row[:6].sum()
print("precision", precision, " cycles per control step of wave 0: %.0f (%.1f us at 2.4 GHz)" % (tot / steps, tot / steps / 2400))
for k in range(6):
    print("  %-42s %8.0f cycles %5.1f %%" % (names[k], row[k] / steps, 100 * row[k] / tot))
