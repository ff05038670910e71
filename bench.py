#!/usr/bin/env python
"""bench.py -- env-steps/sec of the batched A1 simulator (BASELINE.json metric).

A "step" is one env.step() of the whole batch: one 0.026 s control step = 13 physics ticks
for every robot on the rank's GPU.  Workload at N=1 GPU: BASELINE.json configs[1]
("4096 parallel A1, flat terrain, ETG open-loop (no policy net), 1 MI355X"); --config 3 adds
the residual MLP policy (configs[2]).  With --gpus N the driver launches one rank per GPU;
robots shard embarrassingly (4096 per rank, weak scaling) and the only collective is one
all_gather of the episode returns after the timed rollout (configs[3]).

Rank 0 prints ONE JSON line (see the contract in the task description); `roofline` is the
dynamics kernel against the HBM roofline, `cpu_baseline` the oracle timed on host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from paddlerobotics_amd import a1_model as A  # noqa: E402
from paddlerobotics_amd.etg import ETG_layer, Opt_with_points  # noqa: E402
from paddlerobotics_amd.etg_fit import opt_with_points_batched  # noqa: E402

SOLVER_ITERS = 2           # library default (K=2 vs K=50 differ by <0.4 mm after 5 m)
HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)
# PMC figures per CONTROL STEP at N = 4096 (profiles/r01_pmc_16lane_kernels.txt, r01_pmc_4lane_kernels.txt, r01_pmc_cfg3_kernels.txt; separate
# passes, FETCH_SIZE / WRITE_SIZE in KB).  Calibrated for this code's access width -- one dword per lane, coalesced
# SoA -- with tools/ubench/pmc_calib.hip (a 512 MiB copy): FETCH_SIZE reports exactly 1/2 of the bytes read (as the
# micro-arch guide found for wide reads), WRITE_SIZE is exact; hence the factor 2 on the fetch term.  k_rollout16
# runs 50 control steps per launch, so its launch totals are divided by 50.  Scaled linearly with N.
PMC_TRAFFIC_BYTES_AT_4096 = {"k_rollout16": (2 * 17514.5 + 30193.4) * 1024.0 / 50.0, "k_step16": (2 * 3838.5 + 3076.0) * 1024.0,
                             "k_step": (2 * 3879.5 + 3076.0) * 1024.0, "k_rollout": (2 * 17543.8 + 30192.0) * 1024.0 / 50.0,
                             "k_rollout_policy16": (2 * 20892.0 + 33264.0) * 1024.0 / 50.0}
# VALU instructions one wave issues per control step (SQ_INSTS_VALU / SQ_WAVES, same files) and the VALU issue
# capacity of a SIMD measured with tools/ubench/occupancy_rate.hip (8 resident waves of v_fma_f32: 0.384
# wave-instructions per SIMD-cycle at the nominal 2.4 GHz; a lone wave issues one VALU instruction per 4.9-5.4
# cycles, i.e. about 0.2 -- issue_rate2.hip).
PMC_VALU_PER_WAVE = {"k_rollout16": 974745600.0 / 1024.0 / 50.0, "k_step16": 19988480.0 / 1024.0, "k_step": 7204352.0 / 256.0,
                     "k_rollout": 356351744.0 / 256.0 / 50.0, "k_rollout_policy16": 1013550265.0 / 1024.0 / 50.0}
VALU_PEAK_PER_SIMD_CYCLE = 0.384
NOMINAL_HZ = 2.4e9
BYTES_PER_STEP_CFG2 = 816  # 564 B + 252 B per-env ETG w,b
BYTES_PER_STEP_CFG3 = 808 + 252


def etg_population(n, seed, device):
    """BASELINE config 2: per-env ETG control points = prior + N(0, 0.02^2) (SimpleGA first ask)."""
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w0, b0, prior = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)
    rng = np.random.default_rng(seed)
    pts = prior[None] + 0.02 * rng.normal(size=(n, 6, 2))
    w, b = opt_with_points_batched(layer, 0.5, pts, b0, w0, device=device)
    return w.float(), b.float()


def device_copy_bandwidth(dev, nbytes=1 << 30, reps=5):
    """attainable HBM bandwidth on this box: device-to-device copy, read + write bytes per second"""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3)


def cpu_baseline(n_envs, steps, threads):
    """The CPU oracle (port of the path, fp64 like stock pybullet) on a bounded sample."""
    from oracle.oracle import OracleSim
    cfg = A.default_config(n_envs, solver_iters=SOLVER_ITERS)
    sim = OracleSim(cfg, threads=threads)
    w, b = etg_population(n_envs, 0, "cpu")
    sim.set_params(etg_w=w.double().numpy(), etg_b=b.double().numpy())
    sim.reset()
    act = np.zeros((n_envs, 12))
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step(act, want_info=False)
    dt = time.perf_counter() - t0
    return n_envs * steps / dt


def max_over_ranks(seconds, dist, dev):
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def es_generation_leg(env, world, rank, dist, barrier, max_step=400):
    """two warm-up and three timed ES generations (mean) over the population of world x N candidates (candidate i = robot i)."""
    from paddlerobotics_amd import rollout as R
    from paddlerobotics_amd.es import SimpleGA
    from paddlerobotics_amd.etg import ETG_layer, Opt_with_points
    N = env.num_envs
    layer = ETG_layer(0.5, 0.026, 20, 0.04, np.array([-np.pi / 2, 0]), 0.2, 0.5)
    w0, b0, prior = Opt_with_points(layer, ETG_T=0.5, Footheight=0.1, Steplength=0.05)        # train.py:298-299
    solver = SimpleGA(12, sigma_init=0.02, sigma_decay=0.99, sigma_limit=0.005, elite_ratio=0.1, weight_decay=0.005,
                      popsize=world * N, param=np.zeros(12), device=str(env.device))           # train.py:288-295
    evaluate = R.make_etg_evaluator(env, layer, 0.5, prior, w0, b0, max_step=max_step)
    for _ in range(2):      # the first generation skips the elite merge: warm both code paths (lazy kernel loads)
        R.es_generation(solver, evaluate, dist, rank, world)
    GENS = 3
    barrier()
    t0 = time.perf_counter()
    for _ in range(GENS):
        fit = R.es_generation(solver, evaluate, dist, rank, world)
    barrier()
    dt = (time.perf_counter() - t0) / GENS
    dt = max_over_ranks(dt, dist, env.device)
    return {"value": world * N * (max_step + 1) / dt, "unit": "env-steps/s", "ms": dt * 1e3, "population": world * N,
            "control_steps": max_step + 1, "fitness_mean": float(fit.mean().item()),
            "includes": "SimpleGA.ask, etg_fit_etg (batched Opt_with_points), reset, fused open-loop rollout, "
                        "all_gather of the returns, SimpleGA.tell"}


def main():
    global SOLVER_ITERS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--num-envs", type=int, default=4096, help="robots per GPU")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 5),
                    help="BASELINE.json configs (1-based): 2 open loop, 3 + MLP policy, 5 open loop on the random heightfield")
    ap.add_argument("--precision", type=int, default=0, help="policy MFMA: 0 fp32, 1 bf16")
    ap.add_argument("--solver-iters", type=int, default=SOLVER_ITERS, help="PGS sweeps per tick")
    ap.add_argument("--lanes", type=int, default=0, choices=(0, 4, 16),
                    help="kernel mapping, lanes per robot (0 = library default: 16 up to 4096 robots, else 4)")
    ap.add_argument("--body-contacts", action="store_true", help="knee spheres collide too (16-lane heightfield kernels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-es-generation", action="store_true",
                    help="skip the extra ES-generation leg (profiling runs: keeps the per-kernel averages clean)")
    ap.add_argument("--stepwise", action="store_true",
                    help="open-loop configs: time env.step() per control step instead of the fused open-loop rollout")
    args = ap.parse_args()
    SOLVER_ITERS = args.solver_iters

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the simulator has no CPU path")
    # ETG_BENCH_BACKEND=gloo is a dry-run aid (control flow of the N > 1 path on a box with fewer GPUs than ranks: ranks
    # share devices, collectives are staged through the host); the measured configuration is always RCCL, one GPU per rank
    backend = os.environ.get("ETG_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from paddlerobotics_amd.env import make_env
    from paddlerobotics_amd.policy import MfmaPolicy
    from paddlerobotics_amd import rollout as R
    N = args.num_envs
    terrain_kw = {}
    if args.config == 5:   # BASELINE config 5: 256x256 grid, 0.05 m cells, heights U(0, 0.05) m from default_rng(0)
        hf = np.random.default_rng(0).uniform(0.0, 0.05, size=(256, 256)).astype(np.float32)
        terrain_kw = dict(task="heightfield", heightfield=dict(heights=hf, cell=0.05, origin=(-6.4, -6.4)))
    env = make_env("Quadrupedal", num_envs=N, device=str(dev), solver_iters=args.solver_iters,
                   lanes_per_robot=args.lanes, body_contacts=args.body_contacts, **terrain_kw)
    lanes = env.lanes_per_robot
    w, b = etg_population(N, seed=rank, device=dev)
    env.reset(ETG_w=w, ETG_b=b)
    policy = None
    act = None
    if args.config == 3:
        policy = MfmaPolicy(A.OBS_DIM, 12, device=str(dev))
        policy.load_state_dict(MfmaPolicy.init_like_reference(A.OBS_DIM, 12, seed=0))
        act = torch.zeros(N, 12, device=dev)

    def one_step():
        if policy is not None:
            policy.predict(env.obs, 0.3, args.precision, out=act)   # act_bound 0.3, train.py:488
            env.step(act, want_info=False)
        else:
            env.step(None, want_info=False)

    for _ in range(args.warmup):
        one_step()
    if dist is not None:   # warm the one collective of the path too (first-use setup of RCCL's all_gather is not the metric)
        R.gather_returns(env.episode_stats()[0], dist)
    # per-launch duration of the dynamics kernel: HIP event pairs on the launch stream inside the timed region,
    # one pair per EVENT_EVERY launches, spanning EVENT_SPAN back-to-back launches of the step kernel (1 when the
    # policy kernel runs in between).  A pair around EVERY launch costs ~7 us of stream time per step -- 13 % of
    # a 50 us step (tools/launch_gap_probe.py) -- while un-instrumented launches run gap-free, so per-launch pairs
    # would mostly measure the events.
    EVENT_EVERY = 16
    EVENT_SPAN = min(1 if args.config == 3 else 8, max(args.steps, 1))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(max(1, args.steps // EVENT_EVERY))]

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Open-loop configs (2, 5): the hot path is the episode loop with zero residual action (run_episode,
    # pretrain.py:129-154), whose batched counterpart is etg_rollout_openloop -- up to 50 control steps per launch with
    # state, control variables and tick constants in registers.  Config 3 needs the policy between steps, so it
    # (and --stepwise) goes through env.step() once per control step.
    fused = not args.stepwise                                 # config 3: etg_rollout_policy, the policy tile inside the kernel
    barrier()
    t0 = time.perf_counter()
    if fused:
        ev = ev[:1]
        EVENT_SPAN = args.steps                               # one event pair around the K fused steps
        ev[0][0].record()
        if policy is None:
            env.rollout_openloop(args.steps)
        else:
            env.rollout_policy(policy, args.steps, 0.3, args.precision)
        ev[0][1].record()
    for k in range(0 if fused else args.steps):
        slot, phase = divmod(k, EVENT_EVERY)
        if policy is not None:
            policy.predict(env.obs, 0.3, args.precision, out=act)
        if phase == 0 and slot < len(ev):
            ev[slot][0].record()
        env.step(act if policy is not None else None, want_info=False)
        if phase == EVENT_SPAN - 1 and slot < len(ev):
            ev[slot][1].record()
    # episode returns / lengths were accumulated inside the step kernel (alive-masked); for N > 1 the
    # one exchange of the path -- all_gather of the returns (configs[3]; cf. the xparl scatter/gather
    # of model/Dynamic_parallel_model.py:157-160) -- is part of the timed region
    ret, length = env.episode_stats()
    if dist is not None:
        allret = R.gather_returns(ret, dist)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, dev)
    kern_ms = float(np.mean([a.elapsed_time(b_) for a, b_ in ev])) / EVENT_SPAN if ev else float("nan")
    survivors = float((length == args.steps + args.warmup).float().mean().item())
    stepwise = None
    if fused:   # for reference, not part of `value`: the same K steps through env.step(), one launch per control step
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        barrier()
        dt1 = time.perf_counter() - t1
        stepwise = {"value": world * N * args.steps / dt1, "ms_per_step": dt1 / args.steps * 1e3,
                    "note": "env.step() per control step (%s)" % ("k_step16" if lanes == 16 else "k_step") + (", policy.predict() before each" if policy is not None else "")}

    # BASELINE configs[3] in miniature, reported next to `value` (never part of it): ONE full ES generation on the
    # same robots -- SimpleGA.ask, batched Opt_with_points on the device, reset, 401 open-loop control steps, the
    # all_gather of the returns (RCCL for N > 1), SimpleGA.tell replicated on every rank (train.py:398-418).
    es_gen = None
    if args.config != 3 and not args.stepwise and not args.no_es_generation:
        try:
            es_gen = es_generation_leg(env, world, rank, dist, barrier)
        except Exception as e:                                   # noqa: BLE001 - an optional leg must not lose the line
            es_gen = {"error": repr(e)[:200]}

    if rank == 0:
        total_steps = world * N * args.steps
        value = total_steps / elapsed
        bytes_per = BYTES_PER_STEP_CFG3 if args.config == 3 else BYTES_PER_STEP_CFG2
        kname = ("k_rollout16" if fused else "k_step16") if lanes == 16 else ("k_rollout" if fused else "k_step")
        if fused and policy is not None:
            kname = "k_rollout_policy16"
        achieved = bytes_per * N / (kern_ms * 1e-3)
        out = {
            "metric": "env-steps/sec, 4096 A1 quadrupeds; 1/2/4/8-GPU scaling",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("configs[1]: %d parallel A1 per GPU, flat terrain, ETG open-loop, per-env ETG "
                                    "params = prior + N(0,0.02^2)" % N) if args.config == 2 else
                       ("configs[4] (per GPU): %d parallel A1 per GPU, random heightfield 256x256 x 0.05 m, ETG open-loop, "
                        "per-env ETG params" % N) if args.config == 5 else
                       ("configs[2]: %d parallel A1 per GPU, flat, ETG + residual MLP policy (random init, "
                        "precision %d)" % (N, args.precision)),
                       "robots_per_gpu": N, "action_repeat": 13, "sim_dt": 0.002, "solver_iters": args.solver_iters, "lanes_per_robot": lanes,
                       "body_contacts": bool(args.body_contacts), "parallelism": "env-shard x%d" % world},
            "path": ("env.step per control step" if not fused else
                     "etg_rollout_openloop: fused kernel, up to 50 control steps per launch" if policy is None else
                     "etg_rollout_policy: policy MFMA tile + control step fused, up to 50 control steps per launch"),
            "roofline": {"bound": "hbm", "kernel": "etg::" + kname, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK,
                         "traffic": (PMC_TRAFFIC_BYTES_AT_4096[kname] * N / 4096.0) if PMC_TRAFFIC_BYTES_AT_4096[kname] else None,
                         "kernel_ms": kern_ms, "kernel_ms_is": "per control step" if fused else "per launch",
                         "algorithmic_bytes_per_env_step": bytes_per,
                         # the ceiling that actually binds: VALU issue of one wave per SIMD
                         "valu_issue": {"achieved": PMC_VALU_PER_WAVE[kname] / (kern_ms * 1e-3 * NOMINAL_HZ),
                                        "peak": VALU_PEAK_PER_SIMD_CYCLE, "unit": "wave-instr/SIMD-cycle @2.4GHz",
                                        "frac": PMC_VALU_PER_WAVE[kname] / (kern_ms * 1e-3 * NOMINAL_HZ) / VALU_PEAK_PER_SIMD_CYCLE,
                                        "single_wave_limit": 0.2} if N * lanes <= 1024 * 64 else None,
                         "note": "VALU-issue-bound by construction (~1e3 FLOP/B, SURVEY 8d): one wave per SIMD, "
                                 "1 VALU issue / 4 cycles; see DESIGN.md section 7"},
            "survivors": survivors,
        }
        if stepwise is not None:
            out["stepwise"] = stepwise
        if es_gen is not None:
            out["es_generation"] = es_gen
        out["roofline"]["hbm_copy_measured_GBps"] = device_copy_bandwidth(dev) / 1e9
        if policy is not None:
            # the stand-alone policy kernel (what env.step-wise callers launch), 50 back-to-back launches in one event pair
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                policy.predict(env.obs, 0.3, args.precision, out=act)
            e1.record()
            torch.cuda.synchronize(dev)
            pol_ms = e0.elapsed_time(e1) / 50.0
            flops = 2.0 * N * (A.OBS_DIM * 256 + 256 * 256 + 256 * 12)          # 162 304 FLOP per env-step
            peak = 157.3 if args.precision == 0 else 2500.0                     # dense fp32 / bf16 MFMA peaks, TFLOP/s
            out["policy_roofline"] = {"bound": "mfma", "kernel": "k_policy", "achieved": flops / (pol_ms * 1e-3) / 1e12,
                                      "peak": peak, "unit": "TFLOP/s", "frac": flops / (pol_ms * 1e-3) / 1e12 / peak,
                                      "kernel_ms": pol_ms, "dtype": "f32" if args.precision == 0 else "bf16"}
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is an N = 1 exercise
            cores = os.cpu_count() or 1
            one = cpu_baseline(64, 60, 1)
            allc = cpu_baseline(max(64, 16 * cores), 40, cores)
            out["cpu_baseline"] = {"value": allc, "unit": "env-steps/s", "cores": cores, "kind": "port",
                                   "sample": "oracle/etgsim_oracle.cpp fp64, %d envs x 40 steps on %d threads; "
                                             "single-thread: %.0f env-steps/s (64 envs x 60 steps)" %
                                             (max(64, 16 * cores), cores, one),
                                   "single_thread": one}
            # the reference's real engine, if this box happens to have it; never expected here
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import pybullet_baseline
                pb = pybullet_baseline.run(100)
            except Exception:                                   # noqa: BLE001 - a broken pybullet must not kill the bench
                pb = None
            out["cpu_baseline"]["pybullet_env_steps_per_s"] = pb   # None: pybullet unavailable, baseline is the port
        print(json.dumps(out))
    env.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
