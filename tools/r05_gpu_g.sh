#!/bin/bash
# bisect of two round-4 option tests over the library variants
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in ${VARIANTS:-default lib_u2new lib_u2old}; do
  if [ $v = default ]; then unset ETG_LIB; else export ETG_LIB=$R/gpurun_variants/$v.so; fi
  echo "== $v"
  timeout 600 python -m pytest tests/test_gpu_parity4.py -m gpu -q -s -k "restitution or warmstart_085" 2>&1 | grep "parity\] round-4 option .* lanes\|passed\|failed\|^E  "
done
