#!/bin/bash
# round 4, evidence pass B (one MI355X): the -m gpu suite with its [parity] lines, the heightfield kernel summary (configs[4] per
# GPU), the soak run and the cost of env.step() under the reference's randomisation ranges
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/finalb; mkdir -p $O
cd $R
timeout 700 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/pytest_rc.txt
grep -o "\[parity\].*" $O/pytest_gpu.log > $O/parity_report.txt; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
bash tools/profile_gpu.sh r04_cfg5 --config 5 > $O/profile_cfg5.log 2>&1
timeout 300 python tools/random_dynamics_probe.py > $O/random_dynamics.txt 2>&1; head -4 $O/random_dynamics.txt
timeout 500 python tools/soak.py 20000 > $O/soak.txt 2>&1; cat $O/soak.txt
timeout 200 python tools/step_cost_probe.py > $O/step_cost.txt 2>&1; head -20 $O/step_cost.txt
