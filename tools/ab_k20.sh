#!/bin/bash
# A/B the default build against gpurun_variants/*.so at the driver's flags (--steps 20 --warmup 5), interleaved rounds
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2 3; do
  for lib in default $(ls $R/gpurun_variants/*.so 2>/dev/null); do
    if [ "$lib" = default ]; then unset ETG_LIB; else export ETG_LIB=$lib; fi
    python $R/bench.py --steps 20 --warmup 5 --repeats 9 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s K=20: %.2f M env-steps/s, %.2f us per step wall, kernel %.2f us' % ('$(basename $lib)', d['value']/1e6, d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
  done
done
