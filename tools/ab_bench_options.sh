#!/bin/bash
# A/B of the all-options (non-PLAIN) kernels: default build vs gpurun_variants/*.so on option configurations
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in default $(ls $R/gpurun_variants/*.so 2>/dev/null); do
  if [ "$lib" = default ]; then unset ETG_LIB; else export ETG_LIB=$lib; fi
  for a in "" "--body-contacts" "--config 5 --body-contacts" "--lanes 4 --num-envs 16384"; do
    python $R/bench.py --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-12s %-30s' % ('$(basename $lib)', '$a'), '%.2f M env-steps/s' % (d['value']/1e6), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3))"
  done
  python $R/tools/phase_profile.py 16 filter 2>/dev/null | grep -E "total|PGS|Schur"
done
