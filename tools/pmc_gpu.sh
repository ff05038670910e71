#!/bin/bash
# PMC counters of the bench command, one counter group per pass (no trace domains mixed in)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/pmc_$1
mkdir -p $OUT $R/gpurun_out
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o pmc -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-es-generation $PMC_BENCH_ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
run lds SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
python - > $R/gpurun_out/pmc_$1.txt <<PY
import csv, glob, collections, re
for name in ("fetch","write","valu","lds"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True)
    if not fs: print(name, "no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"etg::(k_step16|k_rollout16|k_rollout_policy16|k_step|k_rollout)<", r["Kernel_Name"])   # 16- / 4-lanes-per-robot step kernels
        if m:
            acc[(m.group(1), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kern, k), v in sorted(acc.items()):
        print("%-12s %-22s mean per launch %14.1f  (n=%d)" % (kern, k, sum(v)/len(v), len(v)))
PY
mkdir -p $R/gpurun_out
cat $R/gpurun_out/pmc_$1.txt
