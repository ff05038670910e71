import os, sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
from paddlerobotics_amd.env import make_env
from paddlerobotics_amd.policy import MfmaPolicy
from paddlerobotics_amd import a1_model as A
N = 4096
def run(label, steps=300, policy=False, want_info=False, **kw):
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0", **kw)
    pol = MfmaPolicy(A.OBS_DIM, 12); pol.load_state_dict(MfmaPolicy.init_like_reference(A.OBS_DIM, 12, seed=0))
    obs, _ = env.reset()
    for _ in range(10): env.step(None, want_info=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        act = pol.sample(obs, 0.3)[0] if policy else None
        obs, rew, done, info = env.step(act, want_info=want_info)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e6
    print("%-52s %7.1f us/step" % (label, dt)); env.close()
run("flat, open loop")
run("flat, open loop, want_info", want_info=True)
run("flat, policy sample", policy=True)
run("stairs (heightfield kernel)", task="stairstair")
run("flat + random_force", random_param={"random_force": 1})
run("stairs + force + policy + info", task="stairstair", random_param={"random_force": 1}, policy=True, want_info=True)
env = make_env("Quadrupedal", num_envs=N, device="cuda:0", task="stairstair", random_param={"random_dynamics": 1})
env.reset(); torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); env.reset(); torch.cuda.synchronize(); print("full reset with random dynamics: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
ids = torch.arange(0, N, 2, device="cuda:0")
t0 = time.perf_counter(); env.reset(env_ids=ids); torch.cuda.synchronize(); print("half reset with random dynamics: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
env2 = make_env("Quadrupedal", num_envs=N, device="cuda:0"); env2.reset(); torch.cuda.synchronize()
t0 = time.perf_counter(); env2.reset(); torch.cuda.synchronize(); print("cached full reset: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); env2.reset(env_ids=ids[:100]); torch.cuda.synchronize(); print("cached partial reset (100 robots): %.3f ms" % ((time.perf_counter() - t0) * 1e3))
