#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r05c; mkdir -p $O
one() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-extra-legs "$@" 2>$O/$tag.err | tail -1 > $O/$tag.json; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); print("%-34s %7.2f M  %.2f us/step kernel %.2f us  survivors %.3f" % ("$tag", d["value"]/1e6, d["ms_per_step"]*1e3, d["roofline"]["kernel_ms"]*1e3, d["survivors"]))
except Exception as e: print("$tag failed", e)
PY
}
for round in 1 2; do
for v in default $VARIANTS; do
  if [ $v = default ]; then unset ETG_LIB; else export ETG_LIB=$R/gpurun_variants/$v.so; fi
  one ${v}_cfg5 --steps 100 --warmup 10 --repeats 3 --config 5
  one ${v}_cfg5_k20 --steps 20 --warmup 5 --config 5
done; done
export ETG_LIB=$R/gpurun_variants/$VARIANTS.so
timeout 900 python -m pytest tests -q -m gpu -n 4 -k "heightfield or stairs or terrain or wave_neighbours" 2>&1 | tail -3
