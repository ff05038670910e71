"""etg_rollout_actions (fused tape, recording joint_angle + obs-IMU like the dynamics-identification evaluator) against the same
tape through env.step() with the info reads, per mapping.  One MI355X."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env
for N, lanes, kw in ((4096, 16, {}), (4096, 16, dict(body_contacts=2)), (16384, 4, {}), (16384, 4, dict(body_contacts=2)), (16384, 4, dict(body_contacts=3))):
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0", lanes_per_robot=lanes, **kw)
    g = torch.Generator(device="cuda:0"); g.manual_seed(0)
    T = 100
    tape = (torch.rand(T, N, 12, device="cuda:0", generator=g) * 2 - 1) * 0.2
    def fused():
        env.reset(); env.rollout_actions(tape)
    def stepped():
        env.reset()
        qa = torch.empty(T, N, 12, device="cuda:0"); im = torch.empty(T, N, 6, device="cuda:0")
        for k in range(T):
            _, _, _, info = env.step(tape[k])
            qa[k] = info["joint_angle"]; im[k] = info["obs-IMU"]
    out = []
    for f in (fused, stepped):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter(); f(); f(); torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / (2 * T) * 1e6)
    print("%5d robots, %2d lanes, %-22s fused tape %6.1f us per step | env.step() + info reads %6.1f us per step | ratio %.2f" % (N, lanes, kw, out[0], out[1], out[0] / out[1]), flush=True)
    env.close()
