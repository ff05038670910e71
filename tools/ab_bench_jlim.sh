R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for lib in default $(ls $R/gpurun_variants/*.so); do
  if [ "$lib" = default ]; then unset ETG_LIB; else export ETG_LIB=$lib; fi
  for a in "" "--joint-limits"; do
  python $R/bench.py --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra-legs $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-20s %-16s' % ('$(basename $lib)', '$a'), '%.2f M env-steps/s' % (d['value']/1e6), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3))"
  done
done
done
