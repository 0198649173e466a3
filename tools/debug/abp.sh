#!/bin/bash
# the ab library with and without column prefixes, and the shipped one
run() { python bench.py --steps 40 --warmup 8 --cpu-seconds 0 --e2e-rounds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"; }
for i in 1 2; do
unset BXH_LIB BXH_PREFIX; run main
export BXH_LIB=$PWD/baryonyx_amd/csrc/libbxh_ab.so
BXH_PREFIX=0 run ab-noprefix
BXH_PREFIX=1 run ab-prefix
done
