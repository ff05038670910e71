import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from paddlerobotics_amd.env import make_env
N=4096
def run(name, steps=400, **kw):
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0", auto_reset=True, seed=1, **kw)
    g = torch.Generator(device="cuda:0"); g.manual_seed(2)
    env.reset()
    acts = [(torch.rand(N, 12, device="cuda:0", generator=g) * 2 - 1) * 0.6 for _ in range(8)]
    for k in range(50): env.step(acts[k % 8], want_info=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps): env.step(acts[k % 8], want_info=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-40s %.1f us per step" % (name, dt / steps * 1e6), flush=True)
    env.close()
run("auto_reset")
run("auto_reset + random_force", random_param={"random_force": 1})
run("auto_reset + random_dynamics", steps=100, random_param={"random_dynamics": 1})
# pieces of the random-dynamics reset
env = make_env("Quadrupedal", num_envs=N, device="cuda:0", seed=1, random_param={"random_dynamics": 1})
env.reset()
m = torch.zeros(N, dtype=torch.uint8, device="cuda:0"); m[::17] = 1
from paddlerobotics_amd import a1_model as A
def t(f, n=50):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
p = (torch.rand(N, A.DYN_DIM, device="cuda:0") * 2 - 1) * 0.3
print("param2dynamic_rows_torch        %.1f us" % t(lambda: A.param2dynamic_rows_torch(p)))
rows = A.param2dynamic_rows_torch(p)
print("set_dynamic_param (masked)       %.1f us" % t(lambda: env.set_dynamic_param(rows, m)))
print("reset(env_ids=mask) all-in       %.1f us" % t(lambda: env.reset(env_ids=m), 20))
env2 = make_env("Quadrupedal", num_envs=N, device="cuda:0", seed=1); env2.reset()
print("reset(env_ids=mask), fixed dyn   %.1f us" % t(lambda: env2.reset(env_ids=m), 20))
# the prepared-ahead path (etg_prepare_next_dynamics): one refresh, and the step between refreshes
env3 = make_env("Quadrupedal", num_envs=N, device="cuda:0", seed=1, auto_reset=True, random_param={"random_dynamics": 1}, random_dynamics_refresh=10**9)
env3.reset()
mask = torch.ones(N, dtype=torch.uint8, device="cuda:0")
print("_prepare_next_dynamics(all)      %.1f us" % t(lambda: env3._prepare_next_dynamics(mask), 10))
print("_draw_dynamics_rows              %.1f us" % t(lambda: env3._draw_dynamics_rows(), 20))
g = torch.Generator(device="cuda:0"); g.manual_seed(2)
acts = [(torch.rand(N, 12, device="cuda:0", generator=g) * 2 - 1) * 0.6 for _ in range(8)]
k = [0]
def st():
    env3.step(acts[k[0] % 8], want_info=False); k[0] += 1
print("env.step between refreshes       %.1f us" % t(st, 300))
