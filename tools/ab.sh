#!/bin/bash
# A/B timing of two builds of the C ABI on one box: the shipped csrc/libbxh.so against
# csrc/libbxh_ab.so (another build of the same sources, e.g. with a -D switch), alternating,
# bench.py without its CPU legs.  Usage: tools/ab.sh [rounds] [extra bench.py flags]
R=${1:-2}; shift
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  for v in main ab; do
    if [ $v = ab ]; then export BXH_LIB=$PWD/baryonyx_amd/csrc/libbxh_ab.so; else unset BXH_LIB; fi
    python bench.py --steps 40 --warmup 8 --cpu-seconds 0 --e2e-rounds 0 "$@" 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
  done
done
