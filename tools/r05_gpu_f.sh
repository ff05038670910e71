#!/bin/bash
# A/B of the default library against a variant (OLD=gpurun_variants/<name>.so, default lib_u2old), interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r05f; mkdir -p $O
one() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-extra-legs "$@" 2>$O/$tag.err | tail -1 > $O/$tag.json; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); print("%-28s %7.2f M  %.2f us/step kernel %.2f us  survivors %.3f sweeps %s" % ("$tag", d["value"]/1e6, d["ms_per_step"]*1e3, d["roofline"]["kernel_ms"]*1e3, d["survivors"], list(d["config"]["solver"]["executed_sweeps_per_tick_per_wave"].values())[0]))
except Exception as e: print("$tag failed", e)
PY
}
python tools/invariance_probe.py 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_solver_rule.py -m gpu -x -q 2>&1 | tail -4
for round in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export ETG_LIB=$R/gpurun_variants/${OLD:-lib_u2old}.so; else unset ETG_LIB; fi
    one ${v}_k20_$round --steps 20 --warmup 5
    one ${v}_k400_$round --steps 400 --warmup 20 --repeats 3
    one ${v}_step_$round --steps 20 --warmup 5 --stepwise
    one ${v}_cfg3_$round --steps 20 --warmup 5 --config 3
    one ${v}_cfg5_$round --steps 100 --warmup 10 --repeats 3 --config 5
    one ${v}_toe_$round --steps 20 --warmup 5 --body-contacts 0
  done
done
unset ETG_LIB
