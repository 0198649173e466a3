#!/bin/bash
# A/B timing of two builds of the C ABI on one box, alternating, bench.py without its CPU legs:
# csrc/libbxh_ab.so against the control -- csrc/libbxh_ab0.so when it exists (e.g. a development
# build of the unchanged sources, -DBXH_DEV_ONLY_NP=5: 25 s instead of 4 min), else the shipped
# csrc/libbxh.so.  Usage: tools/ab.sh [rounds] [extra bench.py flags]
R=${1:-2}; shift
mkdir -p gpurun_out
CTL=$PWD/baryonyx_amd/csrc/libbxh_ab0.so
for i in $(seq 1 $R); do
  for v in main ab; do
    if [ $v = ab ]; then export BXH_LIB=$PWD/baryonyx_amd/csrc/libbxh_ab.so;
    elif [ -f $CTL ]; then export BXH_LIB=$CTL; else unset BXH_LIB; fi
    python bench.py --steps 40 --warmup 8 --cpu-seconds 0 --e2e-rounds 0 "$@" 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
  done
done
