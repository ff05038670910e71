#!/bin/bash
# A/B of build variants on the non-PLAIN kernels: knee rows + joint limits on (flat ground) and the stairs task
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for lib in default $(ls $R/gpurun_variants/*.so 2>/dev/null); do
    if [ "$lib" = default ]; then unset ETG_LIB; else export ETG_LIB=$lib; fi
    python $R/bench.py --steps 200 --warmup 20 --body-contacts --joint-limits --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib)', 'knees+limits %.2f M env-steps/s' % (d['value']/1e6), 'kernel %.1f us' % (d['roofline']['kernel_ms']*1e3))"
  done
done
