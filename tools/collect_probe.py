"""Throughput of the SAC data path: replay.collect_transitions (policy.predict / sample + env.step + replay writes per
control step) at 4096 robots, next to the bare stepping loop.  GPU only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env
from paddlerobotics_amd.policy import MfmaPolicy
from paddlerobotics_amd.replay import DeviceReplayMemory, collect_transitions, collect_recorded

N, T = 4096, 400
env = make_env("Quadrupedal", num_envs=N, device="cuda:0")
pol = MfmaPolicy(49, 12); pol.load_state_dict(MfmaPolicy.init_like_reference(49, 12, seed=0))
rpm = DeviceReplayMemory(N * (T + 1), 49, 12)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def bare():
    env.reset()
    act = None
    for s in range(1, T + 2):
        act = pol.predict(env.obs, 0.3, out=act)
        env.step(act, donef=(s > T), want_info=False)

for mode in ("predict", "sample"):
    dt = timed(lambda: collect_transitions(env, rpm, T, policy=pol, mode=mode))
    print("collect_transitions(%s): %.1f us per control step, %.1f M env-steps/s" % (mode, dt / (T + 1) * 1e6, N * (T + 1) / dt / 1e6))
dt = timed(lambda: collect_recorded(env, rpm, T, pol))
print("collect_recorded (fused kernel): %.1f us per control step, %.1f M env-steps/s" % (dt / (T + 1) * 1e6, N * (T + 1) / dt / 1e6))
dt = timed(lambda: collect_recorded(env, rpm, T, pol, mode="sample"))
print("collect_recorded (sample):       %.1f us per control step, %.1f M env-steps/s" % (dt / (T + 1) * 1e6, N * (T + 1) / dt / 1e6))
dt = timed(lambda: (env.reset(), env.rollout_policy(pol, T + 1, 0.3)))
print("fused rollout, nothing stored:   %.1f us per control step, %.1f M env-steps/s" % (dt / (T + 1) * 1e6, N * (T + 1) / dt / 1e6))
dt = timed(bare)
print("predict + step only:          %.1f us per control step, %.1f M env-steps/s" % (dt / (T + 1) * 1e6, N * (T + 1) / dt / 1e6))
