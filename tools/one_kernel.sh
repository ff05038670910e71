#!/bin/bash
# Compile ONE instantiation of a tick kernel (15-30 s instead of the whole library) and print its registers, scratch and static
# instruction mix: tools/one_kernel.sh 'k_rollout16<true, true, true>' 'KCfg, DevState, int, float*, StatOut' [-DFLAG ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); V=$R/gpurun_variants/one; mkdir -p $V
KERN=$1; ARGS=$2; shift; shift
TAG=$(echo "$KERN $*" | tr -c 'A-Za-z0-9' '_')
cat > $V/$TAG.hip <<EOT
#define ETG_TU_PARTS 99
#define ETG_TU_PART 98
#include "$R/paddlerobotics_amd/csrc/etg_kernels.hip"
namespace etg { template __global__ void $KERN($ARGS); }
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only -Wno-unused-value "$@" -c -o $V/$TAG.o $V/$TAG.hip
python - $V/$TAG.o <<'EOP'
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "..", "..", "tools"))
import kernel_isa_stats as K
for sym, g in K.stats(sys.argv[1], ["k_"]).items():
    print(sym[:60], {k: g[k] for k in ("vgpr", "agpr", "scratch", "valu", "dpp", "acc_moves", "s_nop", "mfma", "all") if k in g})
EOP
