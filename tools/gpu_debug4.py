import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from paddlerobotics_amd.env import make_env
env = make_env("Quadrupedal", num_envs=64, device="cuda:0", solver_iters=4)
env.reset(); torch.cuda.synchronize()
print("BEFORE_ROLLOUT", flush=True); sys.stderr.write("BEFORE_ROLLOUT\n"); sys.stderr.flush()
ret, ln = env.rollout_openloop(5); torch.cuda.synchronize()
sys.stderr.write("AFTER_ROLLOUT\n"); sys.stderr.flush()
print(ret[:4].cpu().numpy(), ln[:4].cpu().numpy())
env.close()
