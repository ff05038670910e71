#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (run on the GPU box via gpurun)
set -e
cd /tmp && export TMPDIR=/tmp
# launches of 50 control steps (the library's default is 400): every k_rollout* launch of the run then has the same length and the
# summary's AverageNs / 50 is the per-control-step kernel time that bench.py's own HIP events report (roofline.kernel_ms)
export ETG_ROLLOUT_CHUNK=50
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $R/bench.py --steps 200 --warmup 50 --repeats 3 --no-cpu-baseline --no-extra-legs ${@:2} > $OUT/bench.json 2> $OUT/stderr.log || true
find $OUT -name "*kernel_stats*" | head -3
f=$(find $OUT -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
tail -1 $OUT/bench.json
