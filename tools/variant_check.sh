#!/bin/bash
# correctness of the build variants in gpurun_variants/ (ETG_LIB=<variant>): the main oracle-parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in $(ls $R/gpurun_variants/*.so 2>/dev/null); do
  echo "== $(basename $lib)"
  ETG_LIB=$lib timeout 600 python -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_parity2.py -m gpu -q -s -p no:cacheprovider -k "reset_and_step_match or closed_loop or both_kernel_mappings or etg_act_matches or long_horizon or heightfield_terrain or knee" 2>&1 | grep -E "parity\]|passed|failed|Error" | cut -c1-160
done
