#!/bin/bash
# A/B build variants of the kernel library: tools/build_variant.sh NAME [-DFLAG ...] -> gpurun_variants/NAME.so
# Only etg_kernels.hip is recompiled (with the extra flags); the other three sources are compiled once into gpurun_variants/obj/.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/paddlerobotics_amd/csrc; V=$R/gpurun_variants; mkdir -p $V/obj
NAME=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only -Wno-unused-value"
for s in policy_mlp etg_fit etg_replay; do
  if [ ! -f $V/obj/$s.o ] || [ $C/$s.hip -nt $V/obj/$s.o ] || [ $C/policy_core.h -nt $V/obj/$s.o ]; then /opt/rocm/bin/hipcc $F -c -o $V/obj/$s.o $C/$s.hip; fi
done
/opt/rocm/bin/hipcc $F "$@" -c -o $V/obj/k_$NAME.o $C/etg_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$NAME.so $V/obj/k_$NAME.o $V/obj/policy_mlp.o $V/obj/etg_fit.o $V/obj/etg_replay.o
echo built $V/$NAME.so
