#!/bin/bash
# round 5 artefacts (one MI355X): bench lines, rocprofv3 kernel stats, PMC passes, phase profile, tail probes.  usage: r05_measure.sh a|b
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r05_final; mkdir -p $O
if [ "$1" = a ]; then
  python bench.py > $O/r05_bench_default.json 2> $O/bench_default.err; tail -c 600 $O/r05_bench_default.json | head -c 300; echo
  python bench.py --steps 20 --warmup 5 > $O/r05_bench_steps20.json 2> $O/bench_steps20.err
  for c in 2 3 5; do
    bash tools/profile_gpu.sh r05_cfg$c --config $c > $O/profile_cfg$c.log 2>&1
    f=$(find gpurun_out/prof_r05_cfg$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r05_cfg${c}_kernel_stats.csv
    cp gpurun_out/prof_r05_cfg$c/bench.json $O/r05_cfg${c}_bench.json
    head -4 $O/r05_cfg${c}_kernel_stats.csv
  done
  python tools/phase_profile2.py 2 5 20 2>&1 | grep -v amdgpu > $O/phase_bc2.txt; python tools/phase_profile2.py 0 5 20 2>&1 | grep -v amdgpu > $O/phase_bc0.txt
  head -3 $O/phase_bc2.txt
else
  bash tools/pmc_gpu.sh r05_cfg2 > $O/pmc_cfg2.log 2>&1; tail -12 $O/pmc_cfg2.log
  PMC_PASSES="fetch write valu mfma" PMC_BENCH_ARGS="--config 3" bash tools/pmc_gpu.sh r05_cfg3 > $O/pmc_cfg3.log 2>&1; tail -6 $O/pmc_cfg3.log
  cp gpurun_out/pmc_r05_cfg2.txt gpurun_out/pmc_r05_cfg3.txt gpurun_out/pmc_r05_cfg2.json gpurun_out/pmc_r05_cfg3.json $O/ 2>/dev/null
  python tools/violent_probe.py 2>&1 | grep -v amdgpu > $O/r05_tail_latency.txt; cat $O/r05_tail_latency.txt
fi
