import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from paddlerobotics_amd.env import make_env
N=4096
g = torch.Generator(device="cuda:0"); g.manual_seed(2)
acts = [(torch.rand(N, 12, device="cuda:0", generator=g) * 2 - 1) * 0.6 for _ in range(8)]
def probe(name, scale, amp=1.0, **kw):
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0", seed=1, auto_reset=True, random_dynamics_scale=scale, random_dynamics_refresh=10**9, **kw)
    env.reset()
    for k in range(100): env.step(acts[k % 8] * amp, want_info=False)
    sw = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(200):
        o, r, d, info = env.step(acts[k % 8] * amp, want_info=(k % 20 == 0))
        if k % 20 == 0: sw.append(info["solver_sweeps"].float())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200 * 1e6
    s = torch.stack(sw)
    print("%-48s %.1f us per step; sweeps per tick: mean %.2f, max over robots %.1f" % (name, dt, s.mean().item() / 13, s.max().item() / 13), flush=True)
    env.close()
probe("fixed dynamics, violent actions", 0.3)
probe("random dynamics 0.3, violent actions", 0.3, random_param={"random_dynamics": 1})
probe("random dynamics 0.3, gentle actions", 0.3, amp=0.1, random_param={"random_dynamics": 1})
probe("random dynamics 0.1, violent actions", 0.1, random_param={"random_dynamics": 1})
probe("random dynamics 0.3, violent, K = 4 sweeps", 0.3, random_param={"random_dynamics": 1}, solver_iters=4)
probe("random dynamics 0.3, violent, rule capped at 8", 0.3, random_param={"random_dynamics": 1}, solver_iters=8, solver_residual=1e-7)
probe("random dynamics 0.3, violent, rule capped at 16", 0.3, random_param={"random_dynamics": 1}, solver_iters=16, solver_residual=1e-7)
