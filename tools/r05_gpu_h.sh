#!/bin/bash
# the heightfield workload at K = 100 / 200, default library and the pre-diet variant, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r05h; mkdir -p $O
one() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-extra-legs "$@" 2>$O/$tag.err | tail -1 > $O/$tag.json; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); print("%-28s %7.2f M  %.2f us/step kernel %.2f us  survivors %.3f sweeps %s" % ("$tag", d["value"]/1e6, d["ms_per_step"]*1e3, d["roofline"]["kernel_ms"]*1e3, d["survivors"], list(d["config"]["solver"]["executed_sweeps_per_tick_per_wave"].values())[0]))
except Exception as e: print("$tag failed", e)
PY
}
for round in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export ETG_LIB=$R/gpurun_variants/lib_u2old.so; else unset ETG_LIB; fi
    one ${v}_cfg5_k100_$round --steps 100 --warmup 10 --repeats 3 --config 5
    one ${v}_cfg5_k200_$round --steps 200 --warmup 50 --repeats 3 --config 5
  done
done
