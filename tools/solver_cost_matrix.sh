#!/bin/bash
# what a sweep and its residual test cost: the fused headline rollout under fixed sweep counts with and without the test
# (a vanishing threshold never stops early, so "--solver-iters K --solver-residual 1e-30" = K sweeps + K tests)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python $R/bench.py --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-50s' % '$*', '%.2f M env-steps/s' % (d['value']/1e6), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3), d['config']['solver'].get('executed_sweeps_per_tick_per_wave'))"; }
run
run --solver-iters 1
run --solver-iters 2
run --solver-iters 3
run --solver-iters 4
run --solver-iters 6
run --solver-iters 2 --solver-residual 1e-30
run --solver-iters 4 --solver-residual 1e-30
run --solver-iters 6 --solver-residual 1e-30
run --lanes 4 --num-envs 16384
run --lanes 4 --num-envs 16384 --solver-iters 2
run --config 5
run --config 5 --solver-iters 2
