#!/bin/bash
# round 3, GPU call: A/B of the build variants, then the GPU tests named on the command line
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$1
bash $R/tools/ab_bench.sh > $R/gpurun_out/$1/ab.txt 2>&1
cat $R/gpurun_out/$1/ab.txt
shift
timeout 900 python -m pytest "$@" -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "\[parity\]|passed|failed|FAILED|Error|error|assert" | cut -c1-400
