#!/bin/bash
# The measurement package of a round on one MI355X (run through gpurun; every output under gpurun_out/<tag>_measure/, the files that are
# judged are copied to profiles/ by hand afterwards).  usage: tools/measure_round.sh <tag> a|b|c
#   a: the bench lines (default K = 400, the driver's --steps 20 --warmup 5, --config 3, --config 5 at K = 200) and the
#      rocprofv3 --kernel-trace --stats summaries of the same command per config (tools/profile_gpu.sh)
#   b: the PMC passes (tools/pmc_gpu.sh: one counter group per pass, no trace domains mixed in) for configs 2 and 3
#   c: ROLLOUT_CHUNK sweep (control steps per fused launch; ETG_ROLLOUT_CHUNK) at K = 400, interleaved so that every value sees the same box state
TAG=$1; PART=$2
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${TAG}_measure; mkdir -p $O
if [ "$PART" = a ]; then
  python bench.py > $O/${TAG}_bench_default.json 2> $O/bench_default.err
  python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_steps20.json 2> $O/bench_steps20.err
  python bench.py --config 3 --no-cpu-baseline > $O/${TAG}_bench_config3.json 2> $O/bench_config3.err
  python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_config3_steps20.json 2>> $O/bench_config3.err
  python bench.py --config 5 --steps 200 --no-cpu-baseline > $O/${TAG}_bench_config5.json 2> $O/bench_config5.err
  for c in 2 3 5; do
    bash tools/profile_gpu.sh ${TAG}_cfg$c --config $c > $O/profile_cfg$c.log 2>&1
    f=$(find gpurun_out/prof_${TAG}_cfg$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_cfg${c}_kernel_stats.csv
    cp gpurun_out/prof_${TAG}_cfg$c/bench.json $O/${TAG}_cfg${c}_bench_under_rocprof.json
    head -4 $O/${TAG}_cfg${c}_kernel_stats.csv
  done
elif [ "$PART" = b ]; then
  bash tools/pmc_gpu.sh ${TAG}_cfg2 > $O/pmc_cfg2.log 2>&1; tail -12 $O/pmc_cfg2.log
  PMC_PASSES="fetch write valu mfma" PMC_BENCH_ARGS="--config 3" bash tools/pmc_gpu.sh ${TAG}_cfg3 > $O/pmc_cfg3.log 2>&1; tail -8 $O/pmc_cfg3.log
  cp gpurun_out/pmc_${TAG}_cfg2.txt gpurun_out/pmc_${TAG}_cfg3.txt gpurun_out/pmc_${TAG}_cfg2.json gpurun_out/pmc_${TAG}_cfg3.json $O/ 2>/dev/null
else
  : > $O/${TAG}_chunk_sweep.txt
  for round in 1 2 3; do
    for chunk in 50 100 200 400 25; do
      ETG_ROLLOUT_CHUNK=$chunk python bench.py --steps 400 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs > $O/chunk.json 2>> $O/chunk.err
      python - $chunk $round >> $O/${TAG}_chunk_sweep.txt <<PY
import json, sys
d = json.loads(open("$O/chunk.json").read().strip().splitlines()[-1])
im = d.get("imbalance") or {}
print("round %s ROLLOUT_CHUNK %4s: value %.3f M env-steps/s  %.2f us per control step (min %.2f max %.2f)  launches %s  imbalance %.4f  slowest-wave-total/mean %.4f"
      % (sys.argv[2], sys.argv[1], d["value"] / 1e6, d["ms_per_step"] * 1e3, d["timing"]["ms_per_step_min"] * 1e3, d["timing"]["ms_per_step_max"] * 1e3,
         im.get("launches"), im.get("value", float("nan")), im.get("slowest_wave_total_over_mean", float("nan"))))
PY
    done
  done
  cat $O/${TAG}_chunk_sweep.txt
fi
