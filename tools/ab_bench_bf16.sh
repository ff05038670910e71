#!/bin/bash
# A/B the default build against gpurun_variants/*.so on the closed loop with the bf16 policy tile (precision = 1) and on k_policy
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for lib in default $(ls $R/gpurun_variants/*.so 2>/dev/null); do
    if [ "$lib" = default ]; then unset ETG_LIB; else export ETG_LIB=$lib; fi
    for prec in 0 1; do
      python $R/bench.py --config 3 --precision $prec --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-12s precision %d' % ('$(basename $lib)', $prec), '%.2f M env-steps/s' % (d['value']/1e6), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3), 'k_policy %.2f us' % (d['policy_roofline']['kernel_ms']*1e3))"
    done
  done
done
