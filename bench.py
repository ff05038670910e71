#!/usr/bin/env python
"""bench.py -- env-steps/sec of the batched A1 simulator (BASELINE.json metric).

A "step" is one env.step() of the whole batch: one 0.026 s control step = 13 physics ticks for every robot on the
rank's GPU.  Workload at N = 1 GPU: BASELINE.json configs[1] ("4096 parallel A1, flat terrain, ETG open-loop (no
policy net), 1 MI355X"); --config 3 adds the residual MLP policy (configs[2]), --config 5 is the random heightfield.

`python bench.py --gpus N` with N > 1 and no launcher environment starts the N ranks itself (re-executes under
torch.distributed.run, one rank per GPU, RCCL); under the driver's launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.
Robots shard embarrassingly (4096 per rank, weak scaling); the only collective is one all_gather of the episode
returns after every timed rollout (configs[3]).

Timing: the clocks are warmed with >= 200 ms of real stepping, then REPEATS (>= 5) repeats are
timed, each = reset (untimed), W warm-up steps (untimed), EXACTLY K timed steps bracketed by barrier + synchronize on
both sides with the max over ranks taken; `value` is the MEDIAN repeat, min / max are reported next to it.  Rank 0
prints ONE JSON line; `roofline` is the dynamics kernel against the HBM roofline (HIP events on the launch stream,
inside the timed region), `cpu_baseline` the oracle on the host cores.
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from paddlerobotics_amd import a1_model as A  # noqa: E402
from paddlerobotics_amd.etg import ETG_layer, Opt_with_points  # noqa: E402
from paddlerobotics_amd.etg_fit import opt_with_points_batched  # noqa: E402

# contact solver of the headline = the library default = pybullet's documented rule: up to 50 sweeps
# per tick, stop when the squared velocity-level row residual of a sweep is <= 1e-7.  --solver-iters K alone = exactly K sweeps.
SOLVER = A.solver_rule()   # (sweep cap, residual threshold)
HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)
BYTES_PER_STEP_CFG2 = 816  # 564 B + 252 B per-env ETG w,b
BYTES_PER_STEP_CFG3 = 808 + 252
REPEATS = 5
CLOCK_WARM_SECONDS = 0.25
CLOCK_WARM_STEPS = 600     # untimed scratch-env steps (~57 ms) right before every timed repeat's barrier, in launches of K steps
# PMC figures (HBM traffic, VALU instructions per wave) are NOT measured by this process: they come from separate
# `rocprofv3 --pmc` passes of this same command (tools/pmc_gpu.sh), summarised per kernel and control step in this file.
# The file is stamped with the hash of the kernel sources it was collected on (tools/make_pmc_json.py); when the sources have
# changed since, the figures are reported as STALE (roofline.traffic = null) instead of being passed off as this build's.
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc.json")
VALU_PEAK_GUIDE = 0.5              # MI355X_MICROARCH.md: a SIMD issues one wave64 VALU instruction every 2 cycles
VALU_PEAK_MEASURED = 0.384         # tools/ubench/occupancy_rate.hip: 8 resident waves of v_fma_f32 per SIMD
NOMINAL_HZ = 2.4e9


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


def cpu_baseline(n_envs, steps, threads, solver=None, hist=None, heightfield=None, body_contacts=2):
    """The CPU oracle (port of the path, fp64 like stock pybullet) on a bounded sample, with the SAME contact-solver rule as
    the GPU leg it stands next to.  Persistent workers: every thread runs its slice of the robots through ALL the steps
    (etgo_run_steps), so no thread is spawned or joined per step.  hist (a list): receives the sample's sweep histogram."""
    from oracle.oracle import OracleSim
    it, res = solver if solver is not None else SOLVER
    cfg = A.default_config(n_envs, solver_iters=it, solver_residual=res, terrain=1 if heightfield else 0, heightfield=heightfield,
                           body_contacts=body_contacts)
    sim = OracleSim(cfg, threads=threads)
    if heightfield:
        sim.set_heightfield(heightfield["heights"])
    w, b = etg_population(n_envs, 0, "cpu")
    sim.set_params(etg_w=w.double().numpy(), etg_b=b.double().numpy())
    sim.reset()
    sim.sweep_hist()
    t0 = time.perf_counter()
    sim.run_steps(steps, threads=threads)
    dt = time.perf_counter() - t0
    if hist is not None:
        hist.append(sim.sweep_hist())
    return n_envs * steps / dt


def sweep_summary(h):
    """ticks by sweep count -> {mean, max, fraction of ticks per count}"""
    h = np.asarray(h, dtype=np.float64)
    k = np.arange(len(h))
    nz = np.nonzero(h)[0]
    return {"mean_sweeps_per_tick": float((h * k).sum() / max(h.sum(), 1.0)), "max_sweeps": int(nz.max()) if len(nz) else 0,
            "fraction_of_ticks_by_sweeps": {str(int(i)): round(float(h[i] / h.sum()), 4) for i in nz}}


def usable_cpus():
    """host threads this process may actually run on: the affinity mask, capped by the cgroup CPU quota if there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:                                           # noqa: BLE001
            continue
    return n, quota


def max_over_ranks(seconds, dist, dev):
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def multi_gpu_report(local_seconds, K, N, dist, dev, ret_probe, barrier):
    """`multi_gpu` of the bench line (N > 1 ranks): every rank's own median time per control step and env-steps/s, the one
    exchange of the path -- the all_gather of the N x world fp32 episode returns -- timed on its own, and who reports the world
    size (the collective library, not the flags).  Works on any backend (the CPU suite runs it under gloo, world size 2)."""
    from paddlerobotics_amd import rollout as R
    world = dist.get_world_size()
    on_dev = dist.get_backend() == "nccl"
    mine = torch.tensor([local_seconds / K * 1e3], dtype=torch.float64, device=dev if on_dev else "cpu")
    allr = torch.empty(world, dtype=torch.float64, device=mine.device)
    dist.all_gather_into_tensor(allr, mine)
    for _ in range(3):
        R.gather_returns(ret_probe, dist)
    barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        g = R.gather_returns(ret_probe, dist)
    if ret_probe.is_cuda:
        torch.cuda.synchronize(ret_probe.device)
    gather_ms = (time.perf_counter() - t0) / 20 * 1e3
    return {"ms_per_step_by_rank": [float(x) for x in allr.cpu().tolist()],
            "env_steps_per_s_by_rank": [N / (float(x) * 1e-3) for x in allr.cpu().tolist()],
            "return_gather": {"ms": max_over_ranks(gather_ms, dist, dev), "elements": int(g.numel()), "bytes": int(g.numel()) * 4,
                              "collective": "all_gather_into_tensor", "backend": dist.get_backend(),
                              "note": "20 back-to-back gathers of the N x world fp32 returns, host clock, max over ranks; one such "
                                      "gather is inside every timed repeat"},
            "world_size_reported_by": "rccl (torch.distributed backend nccl)" if on_dev else "torch.distributed/" + dist.get_backend(),
            "world_size": world}


def es_generation_leg(env, world, rank, dist, barrier, max_step=400):
    """two warm-up and three timed ES generations (mean) over the population of world x N candidates (candidate i = robot i)."""
    from paddlerobotics_amd import rollout as R
    from paddlerobotics_amd.es import SimpleGA
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run --nproc-per-node N bench.py ...`."""
    backend = os.environ.get("ETG_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if backend == "nccl" and have < n:
        raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible -- one rank per GPU is the measured configuration "
                         "(ETG_BENCH_BACKEND=gloo runs a control-flow dry run with ranks sharing devices)" % (n, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    global SOLVER
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=REPEATS, help="timed repeats of --steps (median reported; >= 5 by default)")
    ap.add_argument("--num-envs", type=int, default=4096, help="robots per GPU")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 5),
                    help="BASELINE.json configs (1-based): 2 open loop, 3 + MLP policy, 5 open loop on the random heightfield")
    ap.add_argument("--precision", type=int, default=0, help="policy MFMA: 0 fp32, 1 bf16")
    ap.add_argument("--solver-iters", type=int, default=None,
                    help="PGS sweeps per tick: alone = exactly that many (residual test off); default: up to 50 with the residual exit")
    ap.add_argument("--solver-residual", type=float, default=None,
                    help="squared velocity-level row residual at which a tick stops sweeping (default 1e-7; 0 = fixed count)")
    ap.add_argument("--lanes", type=int, default=0, choices=(0, 4, 16),
                    help="kernel mapping, lanes per robot (0 = library default: 16 up to 4096 robots -- 8192 with body rows -- else 4)")
    ap.add_argument("--body-contacts", type=int, default=2, choices=(0, 1, 2),
                    help="link shapes that collide besides the toe spheres (EtgConfig.body_contacts; default 2 = the library default: one "
                         "contact per leg, with friction, on the deepest of knee / shin midpoint / trunk corner; 0 = toe spheres only)")
    ap.add_argument("--foot-friction", type=float, default=None,
                    help="foot friction coefficient of every robot (default: param2dynamic_dict(zeros) = 0.2, on which the open-loop gait skates)")
    ap.add_argument("--no-joint-limits", dest="joint_limits", action="store_false",
                    help="switch the joint-limit stops (a1.py:186-195; on by default, as Bullet enforces the URDF's) off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the legs reported next to `value` (stepwise, auto-reset, K = 50, config 3, ES generation): "
                         "profiling runs, keeps the per-kernel averages clean")
    ap.add_argument("--no-es-generation", action="store_true", help="skip only the ES-generation leg")
    ap.add_argument("--stepwise", action="store_true",
                    help="time env.step() per control step instead of the fused rollout")
    args = ap.parse_args()
    SOLVER = A.solver_rule(args.solver_iters, args.solver_residual)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)                                    # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the simulator has no CPU path")
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d" % (args.gpus, world, world), file=sys.stderr)
    # ETG_BENCH_BACKEND=gloo is a dry-run aid (control flow of the N > 1 path on a box with fewer GPUs than ranks: ranks
    # share devices, collectives are staged through the host); the measured configuration is always RCCL, one GPU per rank
    backend = os.environ.get("ETG_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible)" % (local_rank, torch.cuda.device_count()))
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
        world = dist.get_world_size()                             # what the collective library reports

    from paddlerobotics_amd.env import make_env
    from paddlerobotics_amd.policy import MfmaPolicy
    from paddlerobotics_amd import rollout as R
    N = args.num_envs
    K = args.steps
    terrain_kw = {}
    if args.config == 5:   # BASELINE config 5: 256x256 grid, 0.05 m cells, heights U(0, 0.05) m from default_rng(0)
        hf = np.random.default_rng(0).uniform(0.0, 0.05, size=(256, 256)).astype(np.float32)
        terrain_kw = dict(task="heightfield", heightfield=dict(heights=hf, cell=0.05, origin=(-6.4, -6.4)))
    env_kw = dict(num_envs=N, device=str(dev), lanes_per_robot=args.lanes, body_contacts=args.body_contacts,
                  joint_limits=args.joint_limits, **terrain_kw)
    solver_kw = dict(solver_iters=SOLVER[0], solver_residual=SOLVER[1])
    env = make_env("Quadrupedal", **solver_kw, **env_kw)
    lanes = env.lanes_per_robot
    w, b = etg_population(N, seed=rank, device=dev)
    dyn_row = None
    if args.foot_friction is not None:
        row = A.default_dynamic_row().copy()
        row[1] = args.foot_friction
        dyn_row = torch.as_tensor(np.tile(row, (N, 1)), dtype=torch.float32, device=dev)

    def make_policy():
        p = MfmaPolicy(A.OBS_DIM, 12, device=str(dev))
        p.load_state_dict(MfmaPolicy.init_like_reference(A.OBS_DIM, 12, seed=0))
        return p
    policy = make_policy() if args.config == 3 else None
    act = torch.zeros(N, 12, device=dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    stat_buf = (torch.empty(N, device=dev), torch.empty(N, dtype=torch.int32, device=dev))   # episode returns / lengths of a rollout
    wall_local = []      # this rank's own wall seconds of the last timed_repeats() call (before the max over ranks)
    live_steps = []      # this rank's env-steps of still-running episodes inside the timed region, per repeat of the last call

    step_groups = [1]     # sub-batches of the stepping legs (env.run_groups): 1 = one launch per control step

    def run_steps(e, pol, n, fused):
        """n control steps of the hot path: the fused rollouts (etg_rollout_openloop / etg_rollout_policy, <= 400 control steps
        per launch) or env.step() per control step (policy.predict() before each in closed loop)"""
        if n <= 0:
            return e.episode_stats()
        if fused:   # the fused entry points return the episode statistics themselves (etg_episode_stats inside the C call)
            return e.rollout_openloop(n, out=stat_buf) if pol is None else e.rollout_policy(pol, n, 0.3, args.precision)
        if step_groups[0] > 1:   # the Gym loop per sub-batch on its own stream: one barrier per group instead of one per batch
            return e.run_groups(n, step_groups[0]) if pol is None else e.rollout_policy(pol, n, 0.3, args.precision, fused=False, groups=step_groups[0])
        for _ in range(n):
            if pol is not None:
                pol.predict(e.obs, 0.3, args.precision, out=act)
                e.step(act, want_info=False)
            else:
                e.step(None, want_info=False)
        return e.episode_stats()

    def timed_repeats(e, pol, fused, repeats, events=False):
        """`repeats` x (reset, W untimed warm-up steps, EXACTLY K timed steps + the return gather).  Returns the per-repeat
        wall seconds (max over ranks), the per-repeat kernel milliseconds per control step (HIP event pair on the launch
        stream around the K steps) and the fraction of robots still alive after the last repeat."""
        wall, kern = [], []
        wall_local.clear()
        live_steps.clear()
        surv = float("nan")
        for _ in range(repeats):
            e.reset(ETG_w=w, ETG_b=b, **({} if dyn_row is None else {'dynamic_param': dyn_row}))
            run_steps(e, pol, args.warmup, fused)
            # the reset and a short warm-up leave the GPU mostly idle for a millisecond or more and its clocks drop: a timed
            # region of K = 20 steps (0.8 ms) then runs 14 % slower than the same steps inside a long run.  Keep the chip
            # loaded right up to the barrier with untimed stepping of a scratch env (warm clocks).
            # ... in launches of the timed region's own length: what precedes the region sets the clock it runs at (all 1024 waves
            # busy without a break -> ~2.25 GHz; launches of <= 50 steps, whose tails leave SIMDs idle -> ~2.34 GHz: 4 % on a
            # 20-step region, profiles/r06_ab_experiments.txt section 14), so the chip enters the region in the state a steady run
            # of such launches leaves it in
            for _ in range((CLOCK_WARM_STEPS + K - 1) // K):
                run_steps(warm_env, None, K, True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            if events:
                e0.record()        # (an empty stream: the event pair brackets the same device work as the host clock)
            t0 = time.perf_counter()
            # episode returns / lengths are accumulated inside the kernels (alive-masked) and copied out by the same call; for
            # N > 1 the one exchange of the path -- all_gather of the returns (configs[3]; cf. the xparl scatter/gather of
            # model/Dynamic_parallel_model.py:157-160) -- is part of the timed region
            ret, length = run_steps(e, pol, K, fused)
            if events:
                e1.record()
            if dist is not None:
                R.gather_returns(ret, dist)
            barrier()
            local = time.perf_counter() - t0
            wall.append(max_over_ranks(local, dist, dev))
            wall_local.append(local)
            if events:
                kern.append(e0.elapsed_time(e1) / K)
            if not getattr(e, "auto_reset", False):
                surv = float((length == K + args.warmup).float().mean().item())
                # env-steps of robots whose episode was still running (the fused rollouts do not simulate a finished robot: `value`
                # counts N x K as BASELINE's metric does, live_env_steps_per_s counts these)
                live_steps.append(float((length - args.warmup).clamp(min=0).sum().item()))
        return wall, kern, surv

    # ---- warm everything: lazy kernel loads, the collective, and the clocks (>= 200 ms of real stepping)
    warm_env = make_env("Quadrupedal", **solver_kw, **env_kw)     # scratch robots for the per-repeat clock warm
    warm_env.reset(ETG_w=w, ETG_b=b)
    env.reset(ETG_w=w, ETG_b=b)
    fused = not args.stepwise
    run_steps(env, policy, max(args.warmup, 4), fused)
    run_steps(env, policy, 2, False)
    if dist is not None:
        R.gather_returns(env.episode_stats()[0], dist)
    torch.cuda.synchronize(dev)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < CLOCK_WARM_SECONDS:
        run_steps(env, policy, 100, fused)
        torch.cuda.synchronize(dev)

    repeats = max(1, args.repeats)
    wall, kern, survivors = timed_repeats(env, policy, fused, repeats, events=True)
    elapsed = float(np.median(wall))
    kern_ms = float(np.median(kern))
    live_value = (world * float(np.median(live_steps)) / elapsed) if live_steps else None
    # how unevenly the sweeps load the wavefronts: per-wave shader-clock cycles of the LAST timed repeat's launches (every launch
    # ends with a chip-wide barrier: it lasts as long as its slowest wave)
    imbalance = None
    if fused and policy is None:
        try:
            cyc = env.rollout_wave_cycles().double()
            if cyc.numel() and cyc.shape[0] > 0:
                per_launch_max = cyc.max(dim=1).values
                tot = cyc.sum(dim=0)
                imbalance = {"value": float(per_launch_max.sum().item() / tot.mean().item()), "launches": int(cyc.shape[0]), "waves": int(cyc.shape[1]),
                             "kernel_cycles_per_step": float(per_launch_max.sum().item()) / K, "mean_wave_cycles_per_step": float(tot.mean().item()) / K,
                             "slowest_wave_total_over_mean": float(tot.max().item() / tot.mean().item()),
                             "is": "sum over the launches of the slowest wave's cycles / mean over the waves of their total cycles, last timed "
                                   "repeat (etg_rollout_wave_cycles: clock64 at kernel entry and exit of every wavefront)"}
                # the shader clock the timed launches effectively ran at: boxes of this pool differ by ~5 % in it, and a 20-step
                # launch runs ~5 % below a 400-step one on the same box (profiles/r06_ab_experiments.txt sections 0 and 11) -- the
                # cycle count is the kernel's, the clock is the box's
                if kern_ms > 0:
                    imbalance["effective_clock_ghz"] = imbalance["kernel_cycles_per_step"] / (kern_ms * 1e-3) * 1e-9
        except Exception as e:                                       # noqa: BLE001 - diagnostics must not lose the line
            imbalance = {"error": repr(e)[:200]}
    # N > 1: every rank's own median step time, and the one exchange of the path (the all_gather of the returns) on its own
    multi = None
    if dist is not None:
        multi = multi_gpu_report(float(np.median(wall_local)), K, N, dist, dev, env.episode_stats()[0], barrier)

    # ---- legs reported NEXT to `value`, never part of it
    extra = {}
    if not args.no_extra_legs and not args.stepwise:
        def leg(e, pol, fz, note, reps=3):
            wl, _, sv = timed_repeats(e, pol, fz, reps)
            m = float(np.median(wl))
            out_ = {"value": world * N * K / m, "ms_per_step": m / K * 1e3, "survivors": sv, "note": note}
            if live_steps:
                out_["live_env_steps_per_s"] = world * float(np.median(live_steps)) / m
            return out_
        extra["stepwise"] = leg(env, policy, False, "env.step() per control step (%s)%s" % (
            "k_step16" if lanes == 16 else "k_step", ", policy.predict() before each" if policy is not None else ""))
        # the same loop with the batch as G sub-batches on G streams (etg_step_range): per-robot results identical
        # (tests/test_gpu_groups.py), group g's step k + 1 starts when ITS slowest wavefront has finished
        # (their streams have to sit on different hardware queues of the 4 the runtime uses, or the groups serialise at ~2x:
        # env.tune_groups measures the candidates, retrying a colliding one on fresh streams)
        g_best, g_table = env.tune_groups((2, 4))
        by_g = {}
        for G in sorted(g for g in g_table if g > 1):
            step_groups[0] = G
            try:
                by_g[G] = leg(env, policy, False, "env.run_groups(K, %d): %d sub-batches of %d robots, each stepping on its own stream" % (G, G, -(-N // G)))
            finally:
                step_groups[0] = 1
        best = min(by_g, key=lambda g: by_g[g]["ms_per_step"])
        extra["stepwise_groups"] = {"best_groups": best, "ms_per_step_by_groups": {"1": extra["stepwise"]["ms_per_step"], **{str(g): by_g[g]["ms_per_step"] for g in by_g}},
                                    "tune_groups_us_per_step": {str(g): round(t, 1) for g, t in g_table.items()}, **by_g[best]}
        # rounds 1-5 for continuity: finished robots simulated on inside the fused rollout (accumulators masked)
        env.set_rollout_mode(simulate_finished=True)
        try:
            extra["simulate_finished"] = leg(env, policy, True, "the same fused rollout with etg_set_rollout_mode(h, 1): robots whose episode "
                                             "has ended are simulated on (the behaviour of rounds 1-5; the reference's loops leave at done)")
        finally:
            env.set_rollout_mode(simulate_finished=False)
        if policy is None:
            # what the timed env-step contains: with auto-reset no terminated robot is stepped on
            envr = make_env("Quadrupedal", auto_reset=True, **solver_kw, **env_kw)
            extra["stepwise_auto_reset"] = leg(envr, None, False, "env.step(auto_reset=True): finished robots restart from the "
                                               "settle cache inside the timed region (etg_reset with the done mask after every step)")
            envr.close()
            # fixed sweep counts next to the residual rule: K = 50 (the cap, never stopping early) and K = 2 (the default of
            # rounds 1-2); each with the CPU port timed at the same setting (rank 0, N = 1 only)
            for kfix, reps in ((50, 2), (2, 3)):
                if SOLVER == (kfix, 0.0):
                    continue
                envk = make_env("Quadrupedal", solver_iters=kfix, solver_residual=0.0, **env_kw)
                extra["solver_iters_%d" % kfix] = leg(envk, None, True, "the same fused rollout with exactly %d PGS sweeps per tick "
                                                      "(no residual test)" % kfix, reps=reps)
                envk.close()
            if SOLVER == A.solver_rule():
                # the OTHER candidate for the absent env layer's engine settings (a1_model.solver_preset): the
                # locomotion_gym_env lineage's int(300 / action_repeat) = 23 iterations with the friction pyramid
                ps = A.solver_preset("locomotion_gym", 13)
                envp = make_env("Quadrupedal", solver_preset="locomotion_gym", **env_kw)
                extra["locomotion_gym_preset"] = leg(envp, None, True, "the same fused rollout with solver_preset = 'locomotion_gym': at most %d "
                                                     "sweeps per tick (residual exit 1e-7 still on), friction pyramid" % ps["solver_iters"])
                envp.close()
            if args.body_contacts:
                # rounds 1-4's model for continuity: only the toe spheres collide (shins and trunk pass through the floor)
                kwf = dict(env_kw, body_contacts=0)
                envf = make_env("Quadrupedal", **solver_kw, **kwf)
                extra["toe_spheres_only"] = leg(envf, None, True, "the same fused rollout with body_contacts = 0: the headline model of rounds "
                                                "1-4 (no body rows; the robots sink instead of kneeling)", reps=5)
                envf.close()
        if args.config == 2:
            pol3 = make_policy()
            wl, kn, sv = timed_repeats(env, pol3, True, 3, events=True)
            m = float(np.median(wl))
            extra["config3"] = {"value": world * N * K / m, "ms_per_step": m / K * 1e3, "kernel_ms_per_step": float(np.median(kn)),
                                "survivors": sv, "note": "configs[2]: ETG + residual MLP policy (random init, fp32 MFMA), "
                                "etg_rollout_policy: policy tile + control step fused, <= 400 control steps per launch"}
            if args.precision == 0:     # the opt-in bf16 tile next to it (reduced precision: a side note, never the config-3 number)
                args.precision = 1
                try:
                    wl, kn, sv = timed_repeats(env, pol3, True, 3, events=True)
                finally:
                    args.precision = 0
                m = float(np.median(wl))
                extra["config3_bf16_policy"] = {"value": world * N * K / m, "ms_per_step": m / K * 1e3, "kernel_ms_per_step": float(np.median(kn)),
                                                "survivors": sv, "dtype": "bf16 policy tile, f32 physics",
                                                "note": "the same fused closed loop with precision = 1: bf16 MFMA on fragments packed at load "
                                                        "time; an opt-in arithmetic with its own tolerances, NOT the parity path"}
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                pol3.predict(env.obs, 0.3, args.precision, out=act)
            e1.record()
            torch.cuda.synchronize(dev)
            pol_ms = e0.elapsed_time(e1) / 50.0
            flops = 2.0 * N * (A.OBS_DIM * 256 + 256 * 256 + 256 * 12)          # 162 304 FLOP per env-step
            peak = 157.3 if args.precision == 0 else 2500.0                     # dense fp32 / bf16 MFMA peaks, TFLOP/s
            extra["config3"]["policy_roofline"] = {
                "bound": "mfma", "kernel": "k_policy (stand-alone policy forward, 50 back-to-back launches in one event pair)",
                "achieved": flops / (pol_ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": flops / (pol_ms * 1e-3) / 1e12 / peak, "kernel_ms": pol_ms, "dtype": "f32" if args.precision == 0 else "bf16"}
        if policy is None and not args.no_es_generation:
            # BASELINE configs[3] in miniature: full ES generations on the same robots -- SimpleGA.ask, batched
            # Opt_with_points on the device, reset, 401 open-loop control steps, the all_gather of the returns (RCCL
            # for N > 1), SimpleGA.tell replicated on every rank (train.py:398-418)
            try:
                extra["es_generation"] = es_generation_leg(env, world, rank, dist, barrier)
            except Exception as e:                                   # noqa: BLE001 - an optional leg must not lose the line
                extra["es_generation"] = {"error": repr(e)[:200]}

    if rank == 0:
        value = world * N * K / elapsed
        bytes_per = BYTES_PER_STEP_CFG3 if args.config == 3 else BYTES_PER_STEP_CFG2
        kname = ("k_rollout16" if fused else "k_step16") if lanes == 16 else ("k_rollout" if fused else "k_step")
        if fused and policy is not None:   # fp32 operands: one wave = 4 robots + their policy tile (round 6); bf16: the workgroup tile
            kname = ("k_rollout_policy16w" if args.precision == 0 else "k_rollout_policy16") if lanes == 16 else "k_rollout_policy"
        achieved = bytes_per * N / (kern_ms * 1e-3)
        pmc, pmc_state = {}, "absent"
        try:
            from paddlerobotics_amd.build import kernel_source_hash
            pj = json.load(open(PMC_FILE))
            if pj.get("kernel_source_hash") == kernel_source_hash():
                pmc, pmc_state = pj.get("config%d" % args.config, {}).get(kname, {}), "current"
            else:
                pmc_state = "stale: %s was collected on kernel sources %s, this run's are %s" % (
                    os.path.relpath(PMC_FILE, ROOT), pj.get("kernel_source_hash"), kernel_source_hash())
        except Exception:                                           # noqa: BLE001 - the counters file is optional
            pmc = {}
        traffic = pmc.get("traffic_bytes_per_control_step_at_4096")
        valu = pmc.get("valu_insts_per_wave_per_control_step")
        out = {
            "metric": "env-steps/sec, 4096 A1 quadrupeds; 1/2/4/8-GPU scaling",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "live_env_steps_per_s": live_value, "imbalance": imbalance, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("configs[1]: %d parallel A1 per GPU, flat terrain, ETG open-loop, per-env ETG "
                                    "params = prior + N(0,0.02^2)" % N) if args.config == 2 else
                       ("configs[4] (per GPU): %d parallel A1 per GPU, random heightfield 256x256 x 0.05 m, ETG open-loop, "
                        "per-env ETG params" % N) if args.config == 5 else
                       ("configs[2]: %d parallel A1 per GPU, flat, ETG + residual MLP policy (random init, "
                        "precision %d)" % (N, args.precision)),
                       "robots_per_gpu": N, "action_repeat": 13, "sim_dt": 0.002,
                       "solver": {"rule": ("projected Gauss-Seidel in Bullet's order (joint-limit rows, all normal rows, then the friction "
                                           "pairs); per tick sweep until max_rows ((d lambda_r) A_rr)^2 <= residual_threshold, at most "
                                           "max_sweeps (pybullet: numSolverIterations 50, solverResidualThreshold 1e-7)") if SOLVER[1] > 0
                                  else "projected Gauss-Seidel in Bullet's order, fixed sweep count",
                                  "max_sweeps": SOLVER[0], "residual_threshold": SOLVER[1],
                                  "settings": {"friction": "implicit cone: the friction pair projected on the disc mu * lambda_n (pybullet enableConeFriction = 1)",
                                               "warmstart_normal": env.cfg.warmstart, "warmstart_friction": env.cfg.warmstart_friction,
                                               "contact_slop": env.cfg.contact_slop, "erp": env.cfg.erp, "contact_margin": env.cfg.contact_margin,
                                               "joint_limits": "unilateral rows inside the sweeps", "source": "DESIGN.md section 2 (pybullet's server settings)"}},
                       "solver_iters": SOLVER[0], "lanes_per_robot": lanes,
                       "body_contacts": int(args.body_contacts), "body_friction": env.cfg.body_friction, "foot_friction": args.foot_friction if args.foot_friction is not None else 0.2, "joint_limits": bool(args.joint_limits),   # (the stops are on by default)
                       "auto_reset": False, "parallelism": "env-shard x%d" % world,
                       "world_size_reported_by": ("torch.distributed/" + dist.get_backend()) if dist is not None else "single process"},
            "timing": {"repeats": repeats, "value_is": "median repeat", "ms_per_step_min": min(wall) / K * 1e3,
                       "ms_per_step_max": max(wall) / K * 1e3, "value_min": world * N * K / max(wall), "value_max": world * N * K / min(wall),
                       "clock_warm_s": CLOCK_WARM_SECONDS, "each_repeat": "reset (untimed), %d warm-up steps (untimed), >= %d untimed steps of a "
                       "scratch env in launches of %d steps (the timed region's own length) to keep the clocks where a steady run of "
                       "such launches has them, barrier + synchronize, %d timed steps + return gather, barrier + synchronize, "
                       "max over ranks" % (args.warmup, CLOCK_WARM_STEPS, K, K)},
            "path": ("env.step per control step" if not fused else
                     "etg_rollout_openloop: fused kernel, up to 400 control steps per launch" if policy is None else
                     "etg_rollout_policy: policy MFMA tile + control step fused, up to 400 control steps per launch"),
            "roofline": {"bound": "hbm", "kernel": "etg::" + kname, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK,
                         "traffic": (traffic * N / 4096.0) if traffic else None,
                         "traffic_source": ("%s: 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes of this command "
                                            "(tools/pmc_gpu.sh) on the same kernel sources (hash checked), not counters of this run"
                                            % os.path.relpath(PMC_FILE, ROOT)) if traffic else None,
                         "pmc_file": pmc_state,
                         "kernel_ms": kern_ms, "kernel_ms_is": "per control step (median over the repeats; HIP event pair around "
                                                               "the K timed steps on the launch stream)",
                         "algorithmic_bytes_per_env_step": bytes_per,
                         "note": "VALU-issue-bound by construction (~1e3 FLOP/B, SURVEY 8d): one wave per SIMD; DESIGN.md section 7"},
            "survivors": survivors,
            "finished_episodes": "not simulated: a robot's fused rollout ends with its episode, as the reference's loops do (pretrain.py:137-153, "
                                 "train.py:226-247); `value` = N x K / t (every robot counted for every step, BASELINE's metric), "
                                 "live_env_steps_per_s = steps of still-running episodes / t; leg `simulate_finished` = rounds 1-5",
        }
        if valu and N * lanes <= 1024 * 64:
            # the ceiling that actually binds: VALU issue of one wave per SIMD
            v = valu / (kern_ms * 1e-3 * NOMINAL_HZ)
            out["roofline"]["valu_issue"] = {"achieved": v, "peak": VALU_PEAK_GUIDE, "unit": "wave-instr/SIMD-cycle @2.4GHz",
                                             "frac": v / VALU_PEAK_GUIDE, "peak_is": "MI355X_MICROARCH.md: one wave64 VALU instruction per 2 SIMD cycles",
                                             "peak_measured": VALU_PEAK_MEASURED, "frac_of_measured": v / VALU_PEAK_MEASURED,
                                             "peak_measured_is": "tools/ubench/occupancy_rate.hip: 8 resident waves of v_fma_f32 per SIMD",
                                             "single_wave_limit": 0.2,
                                             "valu_insts_per_wave_step_source": "%s (SQ_INSTS_VALU / SQ_WAVES)" % os.path.relpath(PMC_FILE, ROOT)}
        out.update(extra)
        if "stepwise" in extra:
            # the Gym surface itself, next to the fused headline, as plain top-level scalars (north_star names env.step())
            front = {"stepwise_env_step_value": extra["stepwise"]["value"], "stepwise_env_step_ms_per_step": extra["stepwise"]["ms_per_step"]}
            keys = list(out)
            i = keys.index("ms_per_step") + 1
            out = {**{k: out[k] for k in keys[:i]}, **front, "survivors": out["survivors"], **{k: out[k] for k in keys[i:] if k != "survivors"}}
        if multi is not None:
            out["multi_gpu"] = multi
        out["roofline"]["hbm_copy_measured_GBps"] = device_copy_bandwidth(dev) / 1e9
        if policy is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                policy.predict(env.obs, 0.3, args.precision, out=act)
            e1.record()
            torch.cuda.synchronize(dev)
            pol_ms = e0.elapsed_time(e1) / 50.0
            flops = 2.0 * N * (A.OBS_DIM * 256 + 256 * 256 + 256 * 12)
            peak = 157.3 if args.precision == 0 else 2500.0
            out["policy_roofline"] = {"bound": "mfma", "kernel": "k_policy", "achieved": flops / (pol_ms * 1e-3) / 1e12,
                                      "peak": peak, "unit": "TFLOP/s", "frac": flops / (pol_ms * 1e-3) / 1e12 / peak,
                                      "kernel_ms": pol_ms, "dtype": "f32" if args.precision == 0 else "bf16"}
        # sweeps the kernels EXECUTED (per wave: the slowest robot of the 4 / 16 sharing it sets the count), from the info column
        # of plain env.step() calls over the same K steps on the headline robots -- untimed
        try:
            env.reset(ETG_w=w, ETG_b=b)
            acc = torch.zeros((), device=dev)
            first = None
            for k in range(K):
                _, _, _, inf = env.step(None)
                acc += inf["solver_sweeps"].float().mean()
                if k == 25:
                    first = float(acc.item()) / 26.0 / 13.0
            out["config"]["solver"]["executed_sweeps_per_tick_per_wave"] = {
                "mean_over_the_%d_steps" % K: float(acc.item()) / K / 13.0, "first_26_steps_mean": first,
                "source": "info['solver_sweeps'] (ETG_INFO_SWEEPS) of env.step()"}
        except Exception as e:                                       # noqa: BLE001 - diagnostics must not lose the line
            out["config"]["solver"]["executed_sweeps_per_tick_per_wave"] = {"error": repr(e)[:200]}
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is an N = 1 exercise
            logical = os.cpu_count() or 1
            affinity, quota = usable_cpus()
            hf_cpu = terrain_kw.get("heightfield")
            one = cpu_baseline(64, 60, 1, heightfield=hf_cpu, body_contacts=args.body_contacts)
            # the box reports `logical` CPUs, but a container may be allowed fewer (affinity mask / cgroup quota): probe the
            # thread count instead of trusting cpu_count(), and report the best figure with the threads that produced it
            n_all, s_all = max(64, 16 * logical), 50
            cand = sorted({t for t in (logical, affinity, int(quota) if quota else 0, 128, 64, 32, 16) if 1 <= t <= logical}, reverse=True)
            hist = []
            probe = {t: cpu_baseline(n_all, s_all, t, hist=hist, heightfield=hf_cpu, body_contacts=args.body_contacts) for t in cand}
            cores = max(probe, key=probe.get)
            allc = probe[cores]
            out["cpu_baseline"] = {"value": allc, "unit": "env-steps/s", "cores": cores, "kind": "port",
                                   "solver": {"max_sweeps": SOLVER[0], "residual_threshold": SOLVER[1]},
                                   "sample": "oracle/etgsim_oracle.cpp fp64, same contact-solver rule as `value`, %d envs x %d steps on %d "
                                             "persistent threads (each runs its robots through all the steps; best of the thread counts "
                                             "probed); single-thread: %.0f env-steps/s (64 envs x 60 steps)" % (n_all, s_all, cores, one),
                                   "single_thread": one,
                                   "host": {"logical_cpus": logical, "affinity_cpus": affinity, "cgroup_cpu_quota": quota,
                                            "threads_probed": {str(t): v for t, v in probe.items()}},
                                   "gpu_over_cpu": value / allc,
                                   "gpu_over_cpu_denominator": "port (this repo's fp64 oracle on all host threads); pybullet itself is "
                                                               "not available on the box, so the >=100x-over-pybullet clause is unmeasured"}
            # per-robot sweep counts of the rule on this workload (the oracle is the rule's definition): first 50 control steps
            out["config"]["solver"]["oracle_sweeps_per_tick_per_robot"] = sweep_summary(hist[0])
            # like-for-like CPU figures for the fixed-count legs
            for kfix in (50, 2):
                key = "solver_iters_%d" % kfix
                if key in out:
                    v = cpu_baseline(n_all if kfix == 2 else max(64, n_all // 4), s_all, cores, solver=(kfix, 0.0), heightfield=hf_cpu,
                                     body_contacts=args.body_contacts)
                    out[key]["cpu_baseline"] = {"value": v, "unit": "env-steps/s", "cores": cores, "kind": "port"}
                    out[key]["gpu_over_cpu"] = out[key]["value"] / v
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
    warm_env.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
