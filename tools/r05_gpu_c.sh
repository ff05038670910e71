#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r05c; mkdir -p $O
one() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-extra-legs "$@" 2>$O/$tag.err | tail -1 > $O/$tag.json; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); print("%-28s %7.2f M  %.2f us/step kernel %.2f us  survivors %.3f sweeps %s" % ("$tag", d["value"]/1e6, d["ms_per_step"]*1e3, d["roofline"]["kernel_ms"]*1e3, d["survivors"], list(d["config"]["solver"]["executed_sweeps_per_tick_per_wave"].values())[0]))
except Exception as e: print("$tag failed", e)
PY
}
one l4_16384_bc2 --lanes 4 --num-envs 16384 --steps 100 --warmup 10 --repeats 3
one l4_16384_bc0 --lanes 4 --num-envs 16384 --steps 100 --warmup 10 --repeats 3 --body-contacts 0
one l4_4096_bc2 --lanes 4 --num-envs 4096 --steps 100 --warmup 10 --repeats 3
one cfg5_bc0 --config 5 --steps 100 --warmup 10 --repeats 3 --body-contacts 0
one cfg5_bc2 --config 5 --steps 100 --warmup 10 --repeats 3
one bc2_8192 --num-envs 8192 --lanes 16 --steps 100 --warmup 10 --repeats 3
