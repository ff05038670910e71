"""profiles/<tag>_pmc.json (what bench.py reads for `roofline.traffic` / `valu_issue`) from the per-config outputs of
tools/pmc_gpu.sh: gpurun_out/pmc_<tag>_cfg<N>.json -> {"config<N>": {kernel: {...}}}.  usage: make_pmc_json.py [tag=r03]"""
import json
import os
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paddlerobotics_amd.build import kernel_source_hash  # noqa: E402
out = {"_source": "tools/pmc_gpu.sh: separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU ... | SQ_INSTS_LDS ... | "
                  "SQ_INSTS_MFMA ...) of `python bench.py --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-extra-legs [--config N]` "
                  "on one MI355X; FETCH_SIZE counts half the bytes of dword-per-lane coalesced reads (tools/ubench/pmc_calib.hip), hence "
                  "traffic = 2 x FETCH_SIZE + WRITE_SIZE"}
# the kernels these counters describe: bench.py compares the stamp with the sources it runs and marks the figures stale otherwise
out["kernel_source_hash"] = kernel_source_hash()
for cfg in (2, 3, 5):
    p = os.path.join(ROOT, "gpurun_out", "pmc_%s_cfg%d.json" % (TAG, cfg))
    if os.path.exists(p):
        out["config%d" % cfg] = json.load(open(p))
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_pmc.json" % TAG), "w"), indent=1)
print(sorted(out))
