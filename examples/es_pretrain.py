#!/usr/bin/env python
"""ES pre-training of the ETG gait on the GPU simulator -- the batched counterpart of
QuadrupedalRobots/ETGRL/pretrain.py:220-243 (40 candidates x 401 serial steps per generation there;
here every candidate is one robot of the batch).  Usage: python examples/es_pretrain.py [--popsize 4096]
Several GPUs: python -m torch.distributed.run --nproc-per-node G --master-addr 127.0.0.1 examples/es_pretrain.py
(--popsize is then the GLOBAL population, sharded G ways; one all_gather of the returns per generation, tell()
replicated on every rank -- the xparl scatter/gather of model/Dynamic_parallel_model.py:157-171)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env  # noqa: E402
from paddlerobotics_amd.es import SimpleGA  # noqa: E402
from paddlerobotics_amd.etg import ETG_layer, Opt_with_points  # noqa: E402
from paddlerobotics_amd import rollout as R  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--popsize", type=int, default=4096)
    ap.add_argument("--generations", type=int, default=10)
    ap.add_argument("--max-step", type=int, default=400)
    ap.add_argument("--sigma", type=float, default=0.02)
    args = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    dist = None
    dev = "cuda:%d" % local
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    env = make_env("Quadrupedal", num_envs=args.popsize // world, device=dev)
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w0, b0, prior = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)     # train.py:298-299
    solver = SimpleGA(12, sigma_init=args.sigma, sigma_decay=0.99, sigma_limit=0.005, elite_ratio=0.1,
                      weight_decay=0.005, popsize=args.popsize, param=np.zeros(12), device=dev)  # train.py:288-295
    evaluate = R.make_etg_evaluator(env, layer, 0.5, prior, w0, b0, max_step=args.max_step)
    for g in range(args.generations):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fit = R.es_generation(solver, evaluate, dist, rank, world)
        _, length = env.episode_stats()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        x = R.gather_returns(env.get_state()[:, 0].contiguous(), dist)
        if rank == 0:
            print("gen %2d  fitness max %8.1f mean %8.1f | episode length mean %5.1f | best x %.2f m | %.2f s "
                  "(%.1f M env-steps/s incl. fit+reset)" % (g, fit.max().item(), fit.mean().item(),
                                                          length.float().mean().item(), x[fit.argmax()].item(), dt,
                                                          args.popsize * (args.max_step + 1) / dt / 1e6))
    if rank == 0:
        np.savez("es_pretrain_result.npz", param=solver.get_best_param().cpu().numpy())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
