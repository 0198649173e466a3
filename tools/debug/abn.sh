#!/bin/bash
# A/B/C...: bench.py (C2, no CPU legs) with each of the given libraries in baryonyx_amd/csrc, alternating, R rounds
R=${1:-2}; shift
for i in $(seq 1 $R); do
  for lib in "$@"; do
    BXH_LIB=$PWD/baryonyx_amd/csrc/$lib python bench.py --steps 40 --warmup 8 --cpu-seconds 0 --e2e-rounds 0 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
  done
done
