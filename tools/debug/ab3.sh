#!/bin/bash
for i in 1 2; do
for v in main ab o2; do
  if [ $v = ab ]; then export BXH_LIB=$PWD/baryonyx_amd/csrc/libbxh_ab.so; elif [ $v = o2 ]; then export BXH_LIB=$PWD/baryonyx_amd/csrc/libbxh_tl2.so; else unset BXH_LIB; fi
  python bench.py --steps 40 --warmup 8 --cpu-seconds 0 --e2e-rounds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
done; done
