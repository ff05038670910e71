#!/bin/bash
# A/B the default build against gpurun_variants/*.so on env.step() per control step (bench.py --stepwise), interleaved rounds
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python $R/bench.py --stepwise --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %-28s' % ('$LIBNAME', '$*'), '%.2f M env-steps/s' % (d['value']/1e6), '%.2f us per step' % (d['ms_per_step']*1e3))"; }
for round in 1 2; do
  for lib in default $(ls $R/gpurun_variants/*.so 2>/dev/null); do
    if [ "$lib" = default ]; then unset ETG_LIB; else export ETG_LIB=$lib; fi
    LIBNAME=$(basename $lib)
    one
    if [ $round = 1 ]; then one --config 5; one --config 3; fi
  done
done
