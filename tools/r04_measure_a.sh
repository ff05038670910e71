#!/bin/bash
# round 4, evidence pass A (one MI355X): the bench line (default flags and the driver's), rocprofv3 kernel-trace summary and
# the PMC passes of configs[1] and configs[2]; the stamped profiles/r04_pmc.json is assembled from them (tools/make_pmc_json.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final; mkdir -p $O
cd $R
for cfg in 2 3; do
  bash tools/profile_gpu.sh r04_cfg$cfg --config $cfg > $O/profile_cfg$cfg.log 2>&1
  if [ $cfg = 2 ]; then P="fetch write valu lds"; else P="fetch write valu mfma"; fi
  PMC_PASSES="$P" PMC_BENCH_ARGS="--config $cfg" bash tools/pmc_gpu.sh r04_cfg$cfg > $O/pmc_cfg$cfg.log 2>&1
done
python tools/make_pmc_json.py r04 > $O/make_pmc.log 2>&1; cp profiles/r04_pmc.json gpurun_out/r04_pmc.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 700 $O/bench_default.json; echo
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; head -c 400 $O/bench_steps20.json; echo
ls $O gpurun_out | head -40
