#!/bin/bash
# A/B the default build against alternative builds in gpurun_variants/ on the heightfield workload (bench.py --config 5),
# both lane mappings (interleaved rounds)
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for lanes in 16 4; do
    for lib in default $(ls $R/gpurun_variants/*.so 2>/dev/null); do
      if [ "$lib" = default ]; then unset ETG_LIB; else export ETG_LIB=$lib; fi
      python $R/bench.py --config 5 --lanes $lanes --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib) lanes $lanes', '%.2f M env-steps/s' % (d['value']/1e6), 'kernel %.1f us' % (d['roofline']['kernel_ms']*1e3))"
    done
  done
done
