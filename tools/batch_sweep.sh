#!/bin/bash
# robots per GPU x kernel mapping: env-steps/s of the fused open-loop rollout (bench.py, 200 steps, 3 repeats)
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "1024 16" "2048 16" "4096 16" "8192 16" "1024 4" "4096 4" "8192 4" "16384 4" "32768 4" "65536 4"; do
  set -- $cfg
  python $R/bench.py --num-envs $1 --lanes $2 --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('robots %6d  lanes/robot %2d  %7.1f M env-steps/s  %6.1f us per control step' % ($1, $2, d['value']/1e6, d['ms_per_step']*1e3))"
done
