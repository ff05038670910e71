#!/bin/bash
# A/B on env.step() per control step, 4-lane mapping at 16384 robots
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for lib in default $(ls $R/gpurun_variants/*.so 2>/dev/null); do
    if [ "$lib" = default ]; then unset ETG_LIB; else export ETG_LIB=$lib; fi
    python $R/bench.py --stepwise --lanes 4 --num-envs 16384 --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-12s' % '$(basename $lib)', '%.2f M env-steps/s' % (d['value']/1e6), '%.2f us per step' % (d['ms_per_step']*1e3))"
  done
done
