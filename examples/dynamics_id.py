#!/usr/bin/env python
"""Dynamics identification on the GPU simulator -- the batched counterpart of QuadrupedalRobots/ETGRL/model/
Dynamic_parallel_model.py (ES over the 48 dynamic parameters: every candidate replays recorded gaits on a robot with
`param2dynamic_dict(candidate)` and is scored by loss_func against recorded joint angles / rpy rates; there 10 xparl
workers x K candidates, here every candidate is one robot of the batch).

There are no robot recordings in the reference tree (`mean_dict` comes from the SharePoint data folder), so this example
manufactures them: a robot with hidden "true" parameters replays two gaits, and the ES is asked to recover a parameter
vector that reproduces those recordings.  Usage: python examples/dynamics_id.py [--popsize 1024] [--generations 20]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd import a1_model as A  # noqa: E402
from paddlerobotics_amd.env import make_env  # noqa: E402
from paddlerobotics_amd.es import SimpleGA  # noqa: E402
from paddlerobotics_amd.etg import ETG_layer, Opt_with_points, etg_joint_action  # noqa: E402
from paddlerobotics_amd import rollout as R  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--popsize", type=int, default=1024)
    ap.add_argument("--generations", type=int, default=20)
    ap.add_argument("--steps", type=int, default=100)          # e_step of sample_episode, Dynamic_parallel_model.py:53
    ap.add_argument("--sigma", type=float, default=0.1)
    ap.add_argument("--device", type=str, default="cuda:0")
    ap.add_argument("--step-loop", action="store_true",
                    help="replay through env.step() per control step (the reference's loop shape) instead of the fused action-tape rollout")
    args = ap.parse_args()
    T, N = args.steps, args.popsize
    # two recorded joint-target sequences: the ETG prior gait ("exp") and standing still ("ori")
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w0, b0, _ = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
    pose = A.INIT_MOTOR_ANGLES
    gait = {"exp": np.stack([pose + etg_joint_action(layer, w0, b0, (k + 1) * 0.026) for k in range(T)]),
            "ori": np.tile(pose[None], (T, 1))}
    env = make_env("Quadrupedal", num_envs=N, device=args.device, task="ground", ETG=0)      # Dynamic_parallel_model.py:49
    # the "real robot": robot 0 with hidden parameters
    rng = np.random.default_rng(0)
    truth = rng.uniform(-0.4, 0.4, size=48)
    mean_dict = {}
    rows = A.param2dynamic_rows(np.tile(truth[None], (N, 1)))
    for key in gait:
        env.reset(dynamic_param=rows)
        mot, dr = [], []
        for i in range(T):
            act = torch.as_tensor(gait[key][i] - pose, dtype=torch.float32, device=env.device).expand(N, 12)
            _, _, _, info = env.step(act, donef=False)
            mot.append(info["joint_angle"][0].cpu().numpy()); dr.append(info["obs-IMU"][0, 3:].cpu().numpy())
        mean_dict[key + "_motor_mean"], mean_dict[key + "_drpy_mean"] = np.array(mot), np.array(dr)
        mean_dict[key + "_motor_std"], mean_dict[key + "_drpy_std"] = np.full((T, 12), 0.05), np.full((T, 3), 0.5)
    evaluate = R.make_dynamics_id_evaluator(env, gait, mean_dict, e_steps=T, fused=not args.step_loop)
    solver = SimpleGA(48, sigma_init=args.sigma, sigma_decay=0.995, sigma_limit=0.02, elite_ratio=0.1, weight_decay=0.005,
                      popsize=N, param=np.zeros(48), device=args.device)                     # ES_ParallelModel defaults
    for g in range(args.generations):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fit = R.es_generation(solver, evaluate)
        torch.cuda.synchronize()
        best = solver.get_best_param().cpu().numpy()
        print("gen %2d  reward max %7.3f mean %7.3f | |best - truth| on kp/kd/mass %.3f | %.2f s (%.1f M env-steps/s)" % (
            g, fit.max().item(), fit.mean().item(), np.abs(best - truth)[[2, 6, 7, 8] + list(range(21, 45))].mean(),
            time.perf_counter() - t0, N * 2 * T / (time.perf_counter() - t0) / 1e6))
    np.save("dynamic_param_identified.npy", solver.get_best_param().cpu().numpy())            # ES_ParallelModel.save


if __name__ == "__main__":
    main()
