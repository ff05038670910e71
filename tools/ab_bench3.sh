#!/bin/bash
# A/B the default build against gpurun_variants/*.so on three workloads (interleaved rounds): headline, 4-lane big batch, heightfield
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python $R/bench.py --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %-32s' % ('$LIBNAME', '$*'), '%.2f M env-steps/s' % (d['value']/1e6), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3))"; }
for round in 1 2; do
  for lib in default $(ls $R/gpurun_variants/*.so 2>/dev/null); do
    if [ "$lib" = default ]; then unset ETG_LIB; else export ETG_LIB=$lib; fi
    LIBNAME=$(basename $lib)
    one
    if [ $round = 1 ]; then one --lanes 4 --num-envs 16384; one --config 5; one --solver-iters 2; fi
  done
done
