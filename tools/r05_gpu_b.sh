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
python tools/invariance_probe.py 2>&1 | tail -2
one bc2_k20 --steps 20 --warmup 5 --body-contacts 2
one bc2_k400 --steps 400 --warmup 20 --repeats 3 --body-contacts 2
one bc0_k20 --steps 20 --warmup 5 --body-contacts 0
one bc2_k20_step --steps 20 --warmup 5 --body-contacts 2 --stepwise
one bc2_k20_i23 --steps 20 --warmup 5 --body-contacts 2 --solver-iters 23 --solver-residual 1e-7
one bc2_k20_i8 --steps 20 --warmup 5 --body-contacts 2 --solver-iters 8 --solver-residual 1e-7
one cfg3_k20 --steps 20 --warmup 5 --config 3
one bc2_k100 --steps 100 --warmup 5 --repeats 3
