#!/usr/bin/env python
"""Roll the ETG open-loop for 600 steps and dump info["ETG_act"] -- the batched counterpart of
QuadrupedalRobots/ETGRL/env_test.py:47-58, which produces the gait_action_list_*.npy files the
real-robot deployment consumes (deployment/test.py:86-96).  Usage:
    python examples/export_gait.py --load es_pretrain_result.npz --out gait_action_list_ETG_gpu.npy"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env  # noqa: E402
from paddlerobotics_amd.etg import ETG_layer, Opt_with_points  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--load", type=str, default="", help=".npz with w,b (train.py:301) or with param (12 ETG offsets)")
    ap.add_argument("--out", type=str, default="gait_action_list_ETG_gpu.npy")
    ap.add_argument("--steps", type=int, default=600)
    args = ap.parse_args()
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w, b, prior = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
    if args.load:
        data = np.load(args.load)
        if "w" in data:
            w, b = data["w"], data["b"]
        else:
            w, b, _ = Opt_with_points(layer, ETG_T=0.5, w0=w, b0=b, points=prior + data["param"].reshape(-1, 2))
    env = make_env("Quadrupedal", num_envs=1, device="cuda:0")
    env.reset(ETG_w=w, ETG_b=b)
    rows = []
    for _ in range(args.steps):
        _, _, _, info = env.step(None, donef=False)
        rows.append(info["ETG_act"][0].cpu().numpy().astype(np.float64))
    np.save(args.out, np.asarray(rows))
    print("saved", args.out, np.asarray(rows).shape)


if __name__ == "__main__":
    main()
