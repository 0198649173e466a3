#!/bin/bash
# A/B/C.. timing of several builds of the C ABI on one box: tools/debug/abn.sh "<lib suffixes>" "<config:batch ...>" [rounds]
# e.g. tools/debug/abn.sh "main ab1 ab2" "c2:64 c5:256" 2
LIBS=${1:-"main ab1"}; CFGS=${2:-"c2:64"}; R=${3:-2}
for i in $(seq 1 $R); do
  for cb in $CFGS; do
    c=${cb%%:*}; b=${cb##*:}
    for v in $LIBS; do
      if [ $v = main ]; then unset BXH_LIB; else export BXH_LIB=$PWD/baryonyx_amd/csrc/libbxh_$v.so; fi
      python bench.py --config $c --batch $b --steps 40 --warmup 8 --cpu-seconds 0 --e2e-rounds 0 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $b $v', round(d['value']/1e6,3), round(d['roofline']['avg_launch_ms'],4))"
    done
  done
done
