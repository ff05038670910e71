#!/bin/bash
# round 4, first GPU pass: the -m gpu suite, the heightfield bit-tracking probe with the IEEE build variants, the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04b; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout 500 -s > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
grep -E "passed|failed|FAILED|rc " $O/gputest.log | tail -15
timeout 400 python tools/hf_tracking_probe.py --n 256 --libs $R/gpurun_variants/ieee_hf.so $R/gpurun_variants/nocontract.so $R/gpurun_variants/ieee_all.so > $O/hf_tracking.txt 2>&1
cat $O/hf_tracking.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench_steps20.err; head -c 400 $O/bench_steps20.json
