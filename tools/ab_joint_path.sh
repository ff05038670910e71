#!/bin/bash
# A/B the default build against gpurun_variants/*.so where the joint-limit rows are busy: env.step(auto_reset) under +-0.3 / +-0.6
# rad actions (tools/violent_probe.py's first lines), then the headline (tools/ab_bench3.sh's first line), interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/jp.py <<'P'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from paddlerobotics_amd.env import make_env
N = 4096
for amp in (0.3, 0.6):
    env = make_env("Quadrupedal", num_envs=N, device="cuda:0", auto_reset=True, seed=1, lanes_per_robot=int(os.environ.get("AB_LANES", "0")))
    g = torch.Generator(device="cuda:0"); g.manual_seed(2)
    env.reset()
    acts = [(torch.rand(N, 12, device="cuda:0", generator=g) * 2 - 1) * amp for _ in range(8)]
    for k in range(50): env.step(acts[k % 8], want_info=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(300): env.step(acts[k % 8], want_info=False)
    torch.cuda.synchronize()
    print("%-10s lanes %s env.step(auto_reset), +-%.1f rad: %.1f us per step" % (os.path.basename(os.environ.get("ETG_LIB", "default")), os.environ.get("AB_LANES", "auto"), amp, (time.perf_counter() - t0) / 300 * 1e6), flush=True)
    env.close()
P
for round in 1 2; do
  for lib in default $(ls $R/gpurun_variants/*.so 2>/dev/null); do
    if [ "$lib" = default ]; then unset ETG_LIB; else export ETG_LIB=$lib; fi
    python /tmp/jp.py 2>/dev/null
    AB_LANES=4 python /tmp/jp.py 2>/dev/null
    python $R/bench.py --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s headline %.2f M env-steps/s kernel %.2f us' % ('$(basename $lib)', d['value']/1e6, d['roofline']['kernel_ms']*1e3))"
  done
done
