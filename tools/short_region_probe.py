"""Where the K = 20 timed region of bench.py spends its time beyond the kernel (driver flags: --steps 20 --warmup 5).
One MI355X.  For each host wait mode (HIP's default, hipDeviceScheduleSpin, hipDeviceScheduleBlockingSync) prints, as medians
of 15 repeats: the wall time of  barrier | rollout_openloop(20) | barrier , the HIP event pair around the same device work, the
host's enqueue time, and the same with the raw C call without the statistics copy-out (etg_rollout_actions, ret = len = NULL,
zero tape) to price the k_episode_stats launch."""
import ctypes as C, os, statistics as S, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlerobotics_amd.env import make_env, _ptr
from paddlerobotics_amd import _lib

dev = torch.device("cuda:0")
N, K = 4096, 20
hip = C.CDLL("libamdhip64.so")                      # the runtime torch has already loaded (same SONAME)
hip.hipSetDeviceFlags.argtypes = [C.c_uint]
env = make_env("Quadrupedal", num_envs=N, device="cuda:0", seed=0)
warm = make_env("Quadrupedal", num_envs=N, device="cuda:0", seed=1)
env.reset(); warm.reset()
buf = (torch.empty(N, device=dev), torch.empty(N, dtype=torch.int32, device=dev))
tape = torch.zeros(K, N, 12, device=dev)
lib = env._lib


def with_stats():
    env.rollout_openloop(K, out=buf)


def without_stats():
    _lib.check(lib.etg_rollout_actions(env._h, _ptr(tape), K, _ptr(env.obs), None, None, None, None, None, None, None, env._stream()))


def region(fn, reps=15):
    wall, evt, enq = [], [], []
    for _ in range(reps):
        env.reset()
        env.rollout_openloop(5, out=buf)
        warm.rollout_openloop(600, out=buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        e1.record()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        wall.append((t2 - t0) * 1e6); enq.append((t1 - t0) * 1e6); evt.append(e0.elapsed_time(e1) * 1e3)
    return S.median(wall), S.median(evt), S.median(enq), min(wall)


for _ in range(3):
    warm.rollout_openloop(400, out=buf)
print("K = %d control steps, %d robots; microseconds, medians of 15 (min wall in brackets)" % (K, N))
for name, flag in (("default (auto)", 0), ("hipDeviceScheduleSpin", 1), ("hipDeviceScheduleYield", 2), ("hipDeviceScheduleBlockingSync", 4), ("default (auto) again", 0)):
    rc = hip.hipSetDeviceFlags(flag)
    for what, fn in (("rollout_openloop + stats", with_stats), ("rollout kernel only", without_stats)):
        w, e, q, mn = region(fn)
        print("%-32s %-26s wall %7.1f [%7.1f] = %.2f per step | events %7.1f = %.2f per step | host enqueue %5.1f | wall - events %5.1f   (rc %d)"
              % (name, what, w, mn, w / K, e, e / K, q, w - e, rc))
