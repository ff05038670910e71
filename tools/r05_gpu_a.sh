#!/bin/bash
# round 5, first GPU pass: parity of the new body rows + cost of the new default next to the toe-spheres model
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_parity3.py -q -m gpu -k "body or knee or belly or wave_neighbours or long_horizon" -x > $O/pytest_body.log 2>&1
tail -5 $O/pytest_body.log
one() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-extra-legs "$@" 2>$O/$tag.err | tail -1 > $O/$tag.json; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); print("%-28s %7.2f M  %.2f us/step kernel %.2f us  survivors %.3f" % ("$tag", d["value"]/1e6, d["ms_per_step"]*1e3, d["roofline"]["kernel_ms"]*1e3, d["survivors"]))
except Exception as e: print("$tag failed", e)
PY
}
one bc2_k20 --steps 20 --warmup 5 --body-contacts 2
one bc0_k20 --steps 20 --warmup 5 --body-contacts 0
one bc2_k400 --steps 400 --warmup 20 --repeats 3 --body-contacts 2
one bc0_k400 --steps 400 --warmup 20 --repeats 3 --body-contacts 0
one bc2_mu1_k200 --steps 200 --warmup 20 --repeats 3 --body-contacts 2 --foot-friction 1.0
one bc0_mu1_k200 --steps 200 --warmup 20 --repeats 3 --body-contacts 0 --foot-friction 1.0
one bc1_k400 --steps 400 --warmup 20 --repeats 3 --body-contacts 1
one bc2_k20_step --steps 20 --warmup 5 --body-contacts 2 --stepwise
