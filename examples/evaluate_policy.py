#!/usr/bin/env python
"""Closed-loop evaluation of a trained actor on the GPU simulator -- the batched counterpart of
run_evaluate_episodes (QuadrupedalRobots/ETGRL/train.py:182-211): N robots, one episode each, the reference's
checkpoint format (`torch.save(state_dict)` with actor_model.* keys, mujoco_agent.py:61-65) and ETG file (.npz with
w, b; train.py:386-390).

    python examples/evaluate_policy.py --actor model.pt --etg ETG_models/Slope_ETG.npz --task stairstair
Without --actor a random-initialised actor of the reference architecture is used (BASELINE config 3)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env  # noqa: E402
from paddlerobotics_amd.policy import MfmaPolicy  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--actor", type=str, default="")
    ap.add_argument("--etg", type=str, default="")
    ap.add_argument("--task", type=str, default="ground")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--max-step", type=int, default=600)          # train.py:373
    ap.add_argument("--act-bound", type=float, default=0.3)       # train.py:488
    ap.add_argument("--student", action="store_true", help="46-float observation (no BaseDisplacement), BCtrain.py:53-59")
    args = ap.parse_args()
    env = make_env("Quadrupedal", num_envs=args.num_envs, device="cuda:0", task=args.task, ETG_path=args.etg,
                   sensor_mode={"dis": 0} if args.student else None)
    obs_dim = env.observation_space.shape[0]
    pol = MfmaPolicy(obs_dim, 12)
    if args.actor:
        pol.restore(args.actor)
    else:
        pol.load_state_dict(MfmaPolicy.init_like_reference(obs_dim, 12, seed=0))
    obs, _ = env.reset()
    success = torch.zeros(args.num_envs, device="cuda:0")
    for steps in range(1, args.max_step + 2):
        obs, rew, done, info = env.step(pol.predict(obs, args.act_bound), donef=(steps > args.max_step))
        success += (info["velx"] >= 0.3).float() * (1 - env.done.float())        # train.py:156
    ret, length = env.episode_stats()
    print("episodes %d | return mean %.1f max %.1f | length mean %.1f | survivors %.1f %% | mean x %.2f m" %
          (args.num_envs, ret.mean().item(), ret.max().item(), length.float().mean().item(),
           100.0 * (length > args.max_step).float().mean().item(), env.get_state()[:, 0].mean().item()))


if __name__ == "__main__":
    main()
