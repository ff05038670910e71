import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from paddlerobotics_amd.env import make_env
def sync(tag):
    torch.cuda.synchronize(); print("ok", tag, flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
env = make_env("Quadrupedal", num_envs=n, device="cuda:0", solver_iters=4); sync("create")
env.reset(); sync("reset")
s0 = env.get_state().cpu().numpy(); sync("get_state")
print("reset rows identical:", np.abs(s0 - s0[0]).max())
for k in range(3):
    env.step(None, want_info=False); sync("step %d" % k)
s1 = env.get_state().cpu().numpy()
print("after steps: x moved", np.abs(s1[:,0]-s0[:,0]).max(), "rows identical", np.abs(s1 - s1[0]).max())
ret, ln = env.rollout_openloop(1); sync("rollout 1")
ret, ln = env.rollout_openloop(5); sync("rollout 5")
print(ret[:4].cpu().numpy(), ln[:4].cpu().numpy())
s2 = env.get_state().cpu().numpy()
print("after rollout: x moved", np.abs(s2[:,0]-s1[:,0]).max())
obs, r, d, info = env.step(None, want_info=True); sync("step info")
env.close(); print("closed")
