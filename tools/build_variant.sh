#!/bin/bash
# build an alternative libetgsim into gpurun_variants/<name>.so with extra compiler flags (for tools/ab_bench.sh)
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/gpurun_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only \
  -Wno-unused-value "$@" -o $R/gpurun_variants/$name.so $R/paddlerobotics_amd/csrc/etg_kernels.hip $R/paddlerobotics_amd/csrc/policy_mlp.hip $R/paddlerobotics_amd/csrc/etg_fit.hip $R/paddlerobotics_amd/csrc/etg_replay.hip
