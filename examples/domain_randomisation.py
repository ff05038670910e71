#!/usr/bin/env python
"""Domain-randomised stepping at scale -- the environment side of QuadrupedalRobots/ETGRL/train.py with
`--random_dynamic 1 --random_force 1` (random_param, train.py:253): every robot draws new dynamic parameters
(param2dynamic_dict of U(-1, 1) * scale, train.py:112-126) for every episode and is pushed at random.

With thousands of robots some episode ends on nearly every control step.  `auto_reset=True` restarts finished robots inside
the step launch; the parameters of every robot's NEXT episode are drawn, derived and settled ahead of time in one launch every
`random_dynamics_refresh` control steps (etg_prepare_next_dynamics), so no step waits for a settle.  The contact solver's sweep
cap is lowered from pybullet's 50: with the reference's friction range (mu up to 3.2) a few robots per batch would otherwise
hold their waves at the cap.

Usage: python examples/domain_randomisation.py [--num-envs 4096] [--steps 2000]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env  # noqa: E402
from paddlerobotics_amd.policy import MfmaPolicy  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--scale", type=float, default=0.3, help="random_dynamics_scale: half-width of the draw in [-1, 1] units")
    ap.add_argument("--refresh", type=int, default=256, help="control steps between preparations of the next episodes' dynamics")
    ap.add_argument("--sweep-cap", type=int, default=8)
    ap.add_argument("--device", type=str, default="cuda:0")
    args = ap.parse_args()
    env = make_env("Quadrupedal", num_envs=args.num_envs, device=args.device, auto_reset=True, seed=0,
                   random_param={"random_dynamics": 1, "random_force": 1}, random_dynamics_scale=args.scale,
                   random_dynamics_refresh=args.refresh, solver_iters=args.sweep_cap, solver_residual=1e-7)
    obs_dim, act_dim = env.observation_space.shape[0], env.action_space.shape[0]
    actor = MfmaPolicy(obs_dim, act_dim, device=args.device)
    actor.load_state_dict(MfmaPolicy.init_like_reference(obs_dim, act_dim, seed=0))                # or actor.restore("model.pt")
    obs, _ = env.reset(x_noise=1)
    episodes = torch.zeros((), device=args.device)
    length_sum = torch.zeros((), device=args.device)
    run = torch.zeros(args.num_envs, device=args.device)
    act = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        act = actor.predict(obs, 0.3, out=act)                   # agent.predict(obs) * action_bound (train.py:225-228)
        obs, reward, done, _ = env.step(act, want_info=False)     # finished robots already hold their reset observation
        run += 1
        d = done.view(-1).bool()
        episodes += d.sum()
        length_sum += run[d].sum()
        run = torch.where(d, torch.zeros_like(run), run)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_ep = max(int(episodes.item()), 1)
    print("%d robots x %d control steps in %.2f s: %.1f M env-steps/s through predict + step, %d episodes (mean length %.1f steps), "
          "prepared-ahead dynamics %s" % (args.num_envs, args.steps, dt, args.num_envs * args.steps / dt / 1e6, n_ep,
                                          length_sum.item() / n_ep, "on" if env._nx_on else "off (masked resets)"))
    env.close()


if __name__ == "__main__":
    main()
