"""Why env.step() with auto_reset costs 272 us under +-0.6 rad actions (profiles/r04_random_dynamics.txt) when the oracle's
sweep counts do not rise: time per step, sweeps executed per tick (info column 63), restarts per step, for a few variants."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env
N = 4096


def run(name, amp, cycle, steps=300, **kw):
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0", auto_reset=True, seed=1, **kw)
    g = torch.Generator(device="cuda:0"); g.manual_seed(2)
    env.reset()
    acts = [(torch.rand(N, 12, device="cuda:0", generator=g) * 2 - 1) * amp for _ in range(cycle)]
    for k in range(50): env.step(acts[k % cycle], want_info=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps): env.step(acts[k % cycle], want_info=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sw, mx, dn, big = 0.0, 0.0, 0.0, 0.0
    for k in range(40):
        _, _, d, info = env.step(acts[k % cycle])
        s = env.info_buf.view(N, -1)[:, 63]
        sw += s.mean().item() / 13; mx = max(mx, s.max().item() / 13); dn += d.float().mean().item(); big += (s > 13 * 10).float().mean().item()
    print("%-58s %6.1f us per step | sweeps per tick (executed, per robot's wave) mean %.2f max %.1f | robots in a wave above 10 sweeps per tick %.3f | restarts per step %.4f"
          % (name, dt / steps * 1e6, sw / 40, mx, big / 40, dn / 40), flush=True)
    env.close()


run("amp 0.3, 8 actions cycled", 0.3, 8)
run("amp 0.3, 8 cycled, toe spheres only (rounds 1-4)", 0.3, 8, body_contacts=0)
run("amp 0.3, 8 cycled, solver_preset locomotion_gym", 0.3, 8, solver_preset="locomotion_gym")
run("amp 0.3, 64 cycled, random_dynamics", 0.3, 64, random_param={"random_dynamics": 1})
run("amp 0.3, 64 cycled, random_dynamics, toe spheres only", 0.3, 64, random_param={"random_dynamics": 1}, body_contacts=0)
run("amp 0.3, 64 cycled, random_dynamics, locomotion_gym", 0.3, 64, random_param={"random_dynamics": 1}, solver_preset="locomotion_gym")
run("amp 0.3, 64 cycled, random_dynamics + random_force", 0.3, 64, random_param={"random_dynamics": 1, "random_force": 1})
run("amp 0.3, 64 cycled, random_dynamics, joint_limits off", 0.3, 64, random_param={"random_dynamics": 1}, joint_limits=False)
run("amp 0.6, 64 cycled, random_dynamics", 0.6, 64, random_param={"random_dynamics": 1})
run("amp 0.6, 8 actions cycled", 0.6, 8)
run("amp 0.6, 64 actions cycled", 0.6, 64)
run("amp 0.6, 8 cycled, joint_limits off", 0.6, 8, joint_limits=False)
run("amp 0.6, 8 cycled, solver_iters 8", 0.6, 8, solver_iters=8)
run("amp 0.6, 8 cycled, 4 lanes per robot", 0.6, 8, lanes_per_robot=4)


def run_fused(name, amp, steps=200, **kw):
    """the same workload through etg_rollout_actions (no per-step barrier between the waves; robots that fall stay down)"""
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0", seed=1, **kw)
    g = torch.Generator(device="cuda:0"); g.manual_seed(2)
    tape = (torch.rand(steps, N, 12, device="cuda:0", generator=g) * 2 - 1) * amp
    env.reset(); env.rollout_actions(tape, record=())
    env.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    ret, ln, _ = env.rollout_actions(tape, record=())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    env.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps): env.step(tape[k], want_info=False)
    torch.cuda.synchronize(); ds = time.perf_counter() - t0
    print("%-58s fused %6.1f us per step | the same tape through env.step() %6.1f us per step | mean episode length %.0f of %d"
          % (name, dt / steps * 1e6, ds / steps * 1e6, ln.float().mean().item(), steps), flush=True)
    env.close()


run_fused("fused tape, amp 0.3", 0.3)
run_fused("fused tape, amp 0.3, random_dynamics", 0.3, random_param={"random_dynamics": 1})
run_fused("fused tape, amp 0.6", 0.6)
run_fused("fused tape, amp 0.6, random_dynamics", 0.6, random_param={"random_dynamics": 1})
