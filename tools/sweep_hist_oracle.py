"""Sweep-count histograms of the contact solver's stopping rule (EtgConfig.solver_residual, the library default: at most 50
sweeps, squared row residual <= 1e-7) on the BASELINE workloads, from the ORACLE -- the rule's definition, per robot and tick
(the kernels execute, per wave, the count of the slowest robot sharing the wave: bench.py reports that next to it).
256 robots x 400 control steps each; CPU only.  usage: python tools/sweep_hist_oracle.py > profiles/r03_sweep_histograms.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paddlerobotics_amd import a1_model as A  # noqa: E402
from paddlerobotics_amd.policy import MfmaPolicy  # noqa: E402
from paddlerobotics_amd.terrain import make_task_heightfield  # noqa: E402
from oracle import oracle as O  # noqa: E402
import bench  # noqa: E402

N, STEPS, THREADS = 256, 400, min(32, os.cpu_count() or 1)


def report(name, h, alive):
    k = np.arange(64)
    nz = np.nonzero(h)[0]
    print("%-58s mean %.2f  max %2d  alive at the end %.2f | " % (name, (h * k).sum() / h.sum(), nz.max(), alive) +
          " ".join("%d:%.3f" % (i, h[i] / h.sum()) for i in nz))


def run(name, closed_loop=False, warmstart=0.85, **kw):
    hf = kw.get("heightfield")
    cfg = A.default_config(N, warmstart=warmstart, terrain=1 if hf else 0, **kw)
    sim = O.OracleSim(cfg, threads=THREADS)
    if hf:
        sim.set_heightfield(hf["heights"])
    w, b = bench.etg_population(N, 0, "cpu")
    sim.set_params(etg_w=w.double().numpy(), etg_b=b.double().numpy())
    obs = sim.reset()
    sim.sweep_hist()
    if not closed_loop:
        _, ln = sim.run_steps(STEPS)
        alive = float((ln == STEPS).mean())
    else:
        sd = MfmaPolicy.init_like_reference(A.OBS_DIM, 12, seed=0)
        ws = [sd["actor_model." + k].numpy() for k in ("l1.weight", "l1.bias", "l2.weight", "l2.bias", "mean_linear.weight", "mean_linear.bias")]
        up = np.ones(N, bool)
        for _ in range(STEPS):
            obs, _, d, _ = sim.step(O.mlp_forward(obs, *ws, scale=0.3), want_info=False)
            up &= ~d.astype(bool)
        alive = float(up.mean())
    report(name, sim.sweep_hist(), alive)


if __name__ == "__main__":
    print("# ticks by the number of sweeps the residual rule ran (fraction of all ticks), oracle fp64, %d robots x %d control steps x 13 ticks" % (N, STEPS))
    print("# robots keep being stepped after their episode ended, as in the benchmark's timed region")
    run("configs[1]: flat, ETG open loop (the headline workload)")
    run("configs[2]: flat, ETG + residual MLP policy (random init)", closed_loop=True)
    hf = np.random.default_rng(0).uniform(0.0, 0.05, size=(256, 256)).astype(np.float32)
    run("configs[4] per GPU: random heightfield U(0, 0.05)", heightfield=dict(heights=hf, cell=0.05, origin=(-6.4, -6.4)))
    run("configs[4] + body contacts (2) + joint limits", heightfield=dict(heights=hf, cell=0.05, origin=(-6.4, -6.4)), body_contacts=2, joint_limits=1)
    run("stairstair (the reference's default task, train.py:462)", heightfield=make_task_heightfield("stairstair", variants=16, seed=0))
    run("configs[1] with pybullet's warm-start factor 0.1", warmstart=0.1)
    run("configs[1] with friction_model = 1 (per-direction clamp)", friction_model=1)
