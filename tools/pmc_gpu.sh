#!/bin/bash
# PMC counters of the bench command, one counter group per pass (no trace domains mixed in; gpurun refuses --pmc with
# sys/hip/hsa traces).  usage: tools/pmc_gpu.sh <tag> ; extra bench flags through PMC_BENCH_ARGS (e.g. "--config 3").
# Writes gpurun_out/pmc_<tag>.txt (raw means per launch) and gpurun_out/pmc_<tag>.json (per kernel and control step,
# the format bench.py reads from profiles/r04_pmc.json).
cd /tmp && export TMPDIR=/tmp
# every fused launch of this run covers <= STEPS control steps (the library's default is 400 per launch: the clock-warming rollouts
# of bench.py would otherwise be longer launches than the timed ones, and the per-launch counters below are divided by STEPS)
export ETG_ROLLOUT_CHUNK=50
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/pmc_$1
STEPS=50
mkdir -p $OUT $R/gpurun_out
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o pmc -- python $R/bench.py --steps $STEPS --warmup 5 --repeats 2 --no-cpu-baseline --no-extra-legs $PMC_BENCH_ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
# PMC_PASSES selects the counter groups (default: all five)
PASSES=${PMC_PASSES:-"fetch write valu lds mfma"}
for p in $PASSES; do
  case $p in
    fetch) run fetch FETCH_SIZE ;;
    write) run write WRITE_SIZE ;;
    valu) run valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES ;;
    lds) run lds SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY ;;
    mfma) run mfma SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES ;;
  esac
done
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_]*MFMA[A-Z_0-9]*" | sort -u > $R/gpurun_out/pmc_$1_mfma_counter_names.txt
python - $1 $STEPS > $R/gpurun_out/pmc_$1.txt <<PY
import csv, glob, collections, re, json, sys
tag, steps = sys.argv[1], int(sys.argv[2])
means = collections.defaultdict(dict)
for name in ("fetch", "write", "valu", "lds", "mfma"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True)
    if not fs: print(name, "no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"::(k_step16|k_rollout16|k_rollout_policy16w|k_rollout_policy16|k_step|k_rollout|k_policy)<", r["Kernel_Name"])   # etg:: or (anonymous namespace)::
        if m:
            acc[(m.group(1), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kern, k), v in sorted(acc.items()):
        # the timed launches of the fused kernels cover STEPS control steps each; the largest value group = those launches
        big = ([x for x in v if x >= 0.5 * max(v)] if kern.startswith("k_rollout") else v) or v
        means[kern][k] = sum(big) / len(big)
        print("%-18s %-26s mean per launch %14.1f  (n=%d of %d)" % (kern, k, means[kern][k], len(big), len(v)))
js = {}
for kern, c in means.items():
    per = float(steps) if kern.startswith("k_rollout") else 1.0
    e = {}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:   # KB; FETCH_SIZE counts half of dword-per-lane coalesced reads (tools/ubench/pmc_calib.hip)
        e["traffic_bytes_per_control_step_at_4096"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / per
        e["fetch_kb_per_launch"], e["write_kb_per_launch"] = c["FETCH_SIZE"], c["WRITE_SIZE"]
    if "SQ_INSTS_VALU" in c and c.get("SQ_WAVES"):
        e["valu_insts_per_wave_per_control_step"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"] / per
        e["wave_cycles_per_wave_per_control_step"] = c.get("SQ_WAVE_CYCLES", 0.0) / c["SQ_WAVES"] / per
    for k in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in c: e[k.lower() + "_per_launch"] = c[k]
    e["control_steps_per_launch"] = per
    js[kern] = e
json.dump(js, open("$R/gpurun_out/pmc_%s.json" % tag, "w"), indent=1)
PY
cat $R/gpurun_out/pmc_$1.txt
