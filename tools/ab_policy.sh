#!/bin/bash
# A/B the closed-loop (config 3) rollout of the default build against gpurun_variants/*.so, interleaved rounds
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python $R/bench.py --config 3 --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %-24s' % ('$LIBNAME', '$*'), '%.2f M env-steps/s' % (d['value']/1e6), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3), 'k_policy alone %.2f us (%.2f of fp32 MFMA peak)' % (d['policy_roofline']['kernel_ms']*1e3, d['policy_roofline']['frac']))"; }
for round in 1 2; do
  for lib in default $(ls $R/gpurun_variants/*.so 2>/dev/null); do
    if [ "$lib" = default ]; then unset ETG_LIB; else export ETG_LIB=$lib; fi
    LIBNAME=$(basename $lib)
    one
  done
done
