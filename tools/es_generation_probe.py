"""Where one ES generation of 4096 candidates x 401 control steps spends its time beyond the rollout (bench leg
`es_generation`): each phase timed with a device synchronize on both sides, medians of 7.  One MI355X."""
import os, sys, time, statistics as S
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env
from paddlerobotics_amd.etg import ETG_layer, Opt_with_points
from paddlerobotics_amd.etg_fit import opt_with_points_batched
from paddlerobotics_amd.es import SimpleGA
from paddlerobotics_amd import rollout as R

N = 4096
env = make_env("Quadrupedal", num_envs=N, device="cuda:0")
layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
w0, b0, prior = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
solver = SimpleGA(12, sigma_init=0.02, sigma_decay=0.99, sigma_limit=0.005, elite_ratio=0.1, weight_decay=0.005, popsize=N, param=np.zeros(12), device="cuda:0")
evaluate = R.make_etg_evaluator(env, layer, 0.5, prior, w0, b0, max_step=400)
for _ in range(3):
    R.es_generation(solver, evaluate)
pr = torch.as_tensor(np.asarray(prior), dtype=torch.float64, device="cuda:0")
T = {k: [] for k in ("ask", "fit", "reset", "rollout 401", "tell", "whole generation")}
sync = lambda: torch.cuda.synchronize()
for _ in range(7):
    sync(); t = time.perf_counter(); sol = solver.ask(); sync(); T["ask"].append(time.perf_counter() - t)
    t = time.perf_counter(); w, b = opt_with_points_batched(layer, 0.5, pr[None] + sol.reshape(-1, 6, 2), b0, w0, device="cuda:0"); w = w.float(); b = b.float(); sync(); T["fit"].append(time.perf_counter() - t)
    t = time.perf_counter(); env.reset(ETG_w=w, ETG_b=b); sync(); T["reset"].append(time.perf_counter() - t)
    t = time.perf_counter(); ret, _ = env.rollout_openloop(401); sync(); T["rollout 401"].append(time.perf_counter() - t)
    t = time.perf_counter(); solver.tell(ret); sync(); T["tell"].append(time.perf_counter() - t)
    sync(); t = time.perf_counter(); R.es_generation(solver, evaluate); sync(); T["whole generation"].append(time.perf_counter() - t)
for k, v in T.items():
    print("%-18s %8.3f ms" % (k, S.median(v) * 1e3))
