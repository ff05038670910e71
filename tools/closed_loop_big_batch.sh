#!/bin/bash
# closed loop (configs[2]) at 4096 and 16384 robots per GPU on both mappings, fp32 and bf16 policy tiles
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python $R/bench.py --config 3 --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-52s' % '--config 3 $*', '%7.2f M env-steps/s   kernel %6.2f us per control step' % (d['value']/1e6, d['roofline']['kernel_ms']*1e3))"; }
one
one --precision 1
one --lanes 16 --num-envs 16384
one --lanes 4 --num-envs 16384
one --lanes 4 --num-envs 16384 --precision 1
