#!/usr/bin/env python
"""Golden vectors for the stochastic policy head (run in the build container only; /root/reference is read).

Input : deployment/exp/stairstair/StairStair3_BC1_itr_500383.pt (the reference's trained student actor,
        keys actor_model.{l1,l2,mean_linear,std_linear}.{weight,bias}).
Output: tests/golden/mlp_sample.npz -- obs, noise, and (action, log_prob) of SAC.sample as the reference
        defines it (alg/sac.py:65-76 on top of Actor.forward model/mujoco_model.py:53-60: log_std clamped to
        [-20, 2], reparameterised Normal, tanh squashing, log-prob correction with 1e-6), evaluated with plain
        torch fp32 (alg/sac.py itself needs `parl`, which is absent, so its few lines are restated here with
        torch.distributions.Normal exactly as it calls it).
"""
import os

import numpy as np
import torch
from torch.distributions import Normal

REF = "/root/reference/QuadrupedalRobots/ETGRL/"
OUT = os.path.dirname(os.path.abspath(__file__))

sd = torch.load(REF + "deployment/exp/stairstair/StairStair3_BC1_itr_500383.pt", map_location="cpu")
w = {k: sd["actor_model." + k].float() for k in ("l1.weight", "l1.bias", "l2.weight", "l2.bias", "mean_linear.weight",
                                                  "mean_linear.bias", "std_linear.weight", "std_linear.bias")}
torch.manual_seed(1)
obs = torch.randn(40, 46)
noise = torch.randn(40, 12)
with torch.no_grad():
    x = torch.relu(obs @ w["l1.weight"].T + w["l1.bias"])
    x = torch.relu(x @ w["l2.weight"].T + w["l2.bias"])
    mean = x @ w["mean_linear.weight"].T + w["mean_linear.bias"]
    log_std = torch.clamp(x @ w["std_linear.weight"].T + w["std_linear.bias"], min=-20.0, max=2.0)
    normal = Normal(mean, log_std.exp())
    x_t = mean + log_std.exp() * noise                       # rsample() with the noise made explicit
    action = torch.tanh(x_t)
    log_prob = normal.log_prob(x_t) - torch.log((1 - action.pow(2)) + 1e-6)
    log_prob = log_prob.sum(1, keepdim=True)
np.savez_compressed(os.path.join(OUT, "mlp_sample.npz"), obs=obs.numpy(), noise=noise.numpy(), action=action.numpy(),
                    log_prob=log_prob.numpy(), log_std=log_std.numpy(),
                    std_linear_weight=w["std_linear.weight"].numpy(), std_linear_bias=w["std_linear.bias"].numpy())
print("wrote mlp_sample.npz; log_std range", float(log_std.min()), float(log_std.max()))
