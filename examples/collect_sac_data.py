#!/usr/bin/env python
"""Filling the replay memory of the reference's SAC loop from the GPU simulator -- the data half of
QuadrupedalRobots/ETGRL/train.py (run_train_episode :129-179 and run_EStrain_episode :213-249 with --es_rpm): warm-up
episodes with uniform actions, then episodes with the stochastic actor; every transition of a robot whose episode is
still running goes to a DeviceReplayMemory in HBM, and `sample_batch` hands the learner device tensors.

The learner itself (alg/sac.py) is out of scope here; the loop below shows where `agent.learn(*batch)` plugs in.
Usage: python examples/collect_sac_data.py [--num-envs 1024] [--episodes 3] [--max-step 100]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env  # noqa: E402
from paddlerobotics_amd.policy import MfmaPolicy  # noqa: E402
from paddlerobotics_amd.replay import DeviceReplayMemory, collect_recorded, collect_transitions  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=1024)
    ap.add_argument("--episodes", type=int, default=3)
    ap.add_argument("--max-step", type=int, default=100)
    ap.add_argument("--batch-size", type=int, default=256)       # BATCH_SIZE, train.py:37
    ap.add_argument("--device", type=str, default="cuda:0")
    args = ap.parse_args()
    env = make_env("Quadrupedal", num_envs=args.num_envs, device=args.device, task="stairstair")   # the reference's default task
    obs_dim, act_dim = env.observation_space.shape[0], env.action_space.shape[0]
    actor = MfmaPolicy(obs_dim, act_dim, device=args.device)
    actor.load_state_dict(MfmaPolicy.init_like_reference(obs_dim, act_dim, seed=0))                # or actor.restore("model.pt")
    rpm = DeviceReplayMemory(max_size=int(1e6), obs_dim=obs_dim, act_dim=act_dim, device=args.device)   # MEMORY_SIZE, train.py:35
    for ep in range(args.episodes):
        mode = "uniform" if ep == 0 else "sample"                # rpm.size() < WARMUP_STEPS -> uniform actions, train.py:141-142
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ret, length, infos = collect_transitions(env, rpm, args.max_step, policy=actor, action_bound=0.3, mode=mode, x_noise=1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        stored = int(length.sum().item())
        print("episode %d (%s): %d transitions in %.1f ms (%.1f M/s), memory holds %d, mean return %.2f, success rate %.2f"
              % (ep, mode, stored, dt * 1e3, stored / dt / 1e6, rpm.size(), ret.mean().item(), infos["success_rate"].mean().item()))
        batch_obs, batch_action, batch_reward, batch_next_obs, batch_terminal = rpm.sample_batch(args.batch_size)
        # critic_loss, actor_loss = agent.learn(batch_obs, batch_action, batch_reward, batch_next_obs, batch_terminal)   # alg/sac.py
        assert batch_obs.shape == (args.batch_size, obs_dim) and batch_terminal.min().item() >= 0.0
    # the ES phase (run_EStrain_episode with --es_rpm, train.py:213-249: deterministic actor, every candidate's episode kept for
    # the learner) goes through the fused closed-loop kernel, which records the steps itself
    flat = make_env("Quadrupedal", num_envs=args.num_envs, device=args.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ret, length = collect_recorded(flat, rpm, args.max_step, actor, action_bound=0.3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("recorded ES episode: %d transitions in %.1f ms (%.1f M/s), memory holds %d" % (int(length.sum().item()), dt * 1e3,
                                                                                       int(length.sum().item()) / dt / 1e6, rpm.size()))
    flat.close()
    env.close()


if __name__ == "__main__":
    main()
