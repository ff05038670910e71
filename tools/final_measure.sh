#!/bin/bash
# the round's measurement pass on one MI355X (run through gpurun): GPU tests with the [parity] lines, the bench line with
# the default flags and with the driver's, rocprofv3 kernel-trace summaries and PMC passes for configs 2 / 3 / 5.
# Outputs under gpurun_out/final/ (+ gpurun_out/pmc_<tag>.*, gpurun_out/prof_<tag>/); copy what is judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/pytest_rc.txt
grep -o "\[parity\].*" $O/pytest_gpu.log > $O/parity_report.txt; tail -3 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json | head -c 300; echo
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
for cfg in 2 3 5; do
  bash tools/profile_gpu.sh ${TAG}_cfg$cfg --config $cfg > $O/profile_cfg$cfg.log 2>&1
  PMC_BENCH_ARGS="--config $cfg" bash tools/pmc_gpu.sh ${TAG}_cfg$cfg > $O/pmc_cfg$cfg.log 2>&1
done
python tools/config_matrix_check.py > $O/config_matrix.log 2>&1; tail -1 $O/config_matrix.log
bash tools/solver_cost_matrix.sh > $O/solver_cost_matrix.txt 2>&1
bash tools/batch_sweep.sh > $O/batch_sweep.txt 2>&1
ls $O
