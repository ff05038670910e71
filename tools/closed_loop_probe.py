"""Fused closed loop (etg_rollout_policy) against policy.predict() + env.step() per control step, per kernel instantiation:
which of the two env.rollout_policy should take.  One MI355X."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env, _ptr
from paddlerobotics_amd.policy import MfmaPolicy
from paddlerobotics_amd import a1_model as A, _lib
import ctypes as C
import numpy as np

hf = np.random.default_rng(0).uniform(0.0, 0.05, size=(256, 256)).astype(np.float32)
HF = dict(task="heightfield", heightfield=dict(heights=hf, cell=0.05, origin=(-6.4, -6.4)))
cases = [(4096, 16, {}), (4096, 16, dict(enable_action_filter=True)), (4096, 16, dict(body_contacts=2)), (4096, 16, HF),
         (4096, 16, dict(body_contacts=2, **HF)), (16384, 4, {}), (16384, 4, dict(enable_action_filter=True)), (16384, 4, dict(body_contacts=2)), (16384, 4, HF)]
for N, lanes, kw in cases:
    pol = MfmaPolicy(A.OBS_DIM, 12, device="cuda:0"); pol.load_state_dict(MfmaPolicy.init_like_reference(A.OBS_DIM, 12, seed=0))
    a = make_env("Quadrupedal", num_envs=N, device="cuda:0", lanes_per_robot=lanes, **kw)
    ret = torch.empty(N, device="cuda:0"); ln = torch.empty(N, dtype=torch.int32, device="cuda:0")
    def fused(n):
        _lib.check(a._lib.etg_rollout_policy(a._h, pol._h, n, C.c_float(0.3), 0, 0, _ptr(a.obs), _ptr(ret), _ptr(ln), a._stream()))
    def stepped(n):
        for k in range(n): a.step(pol.predict(a.obs, 0.3), want_info=False)
    out = []
    for f in (fused, stepped):
        a.reset(); f(20); a.reset(); torch.cuda.synchronize(); t0 = time.perf_counter(); f(200); torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 200 * 1e6)
    print("%5d robots, %2d lanes, %-46s fused %6.1f us per step | predict + step %6.1f us per step | fused / stepped %.2f"
          % (N, lanes, {k: (v if k != "heightfield" else "256x256") for k, v in kw.items()}, out[0], out[1], out[0] / out[1]), flush=True)
    a.close()
