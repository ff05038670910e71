#!/bin/bash
# A/B build variants of the kernel library: tools/build_variant.sh NAME [-DFLAG ...] -> gpurun_variants/NAME.so
# (the product's own parallel build -- paddlerobotics_amd/build.py -- with the extra flags, objects under gpurun_variants/NAME.so.obj/)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); V=$R/gpurun_variants; mkdir -p $V
NAME=$1; shift
cd $R && python -m paddlerobotics_amd.build --lib $V/$NAME.so "$@" > $V/$NAME.log 2>&1 || { tail -20 $V/$NAME.log; exit 1; }
echo built $V/$NAME.so
