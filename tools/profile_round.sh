#!/bin/bash
# Collect the rocprofv3 summaries a round's numbers are quoted from (run on the
# GPU box through gpurun; results land in gpurun_out/<round>/ and the ones to be
# judged are copied into profiles/<round>/ afterwards).
#   tools/profile_round.sh r02 [c2|c4|short|all]
set -o pipefail
R=${1:-r02}; WHAT=${2:-all}
export TMPDIR=/tmp
O=gpurun_out/$R; mkdir -p $O
run() { echo "== $*" >&2; "$@"; }
if [ "$WHAT" = c2 ] || [ "$WHAT" = all ]; then
  run rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_stats -- python3 bench.py --steps 30 --warmup 3 --cpu-seconds 0 --e2e-rounds 0 > $O/c2_stats.log 2>&1 || exit 1
  run rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c2_fetch -- python3 bench.py --steps 10 --warmup 2 --cpu-seconds 0 --e2e-rounds 0 > $O/c2_fetch.log 2>&1 || exit 1
  run rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c2_write -- python3 bench.py --steps 10 --warmup 2 --cpu-seconds 0 --e2e-rounds 0 > $O/c2_write.log 2>&1 || exit 1
  run python3 bench.py --steps 60 --warmup 5 > $O/bench_line_c2.json 2> $O/bench_line_c2.err || exit 1
fi
if [ "$WHAT" = c4 ] || [ "$WHAT" = all ]; then
  run rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -- python3 bench.py --config c4 --cost-kind real --steps 4 --warmup 1 --cpu-seconds 0 > $O/c4_stats.log 2>&1 || exit 1
  run rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c4_fetch -- python3 bench.py --config c4 --cost-kind real --steps 2 --warmup 1 --cpu-seconds 0 > $O/c4_fetch.log 2>&1 || exit 1
  run rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c4_write -- python3 bench.py --config c4 --cost-kind real --steps 2 --warmup 1 --cpu-seconds 0 > $O/c4_write.log 2>&1 || exit 1
  run python3 bench.py --config c4 --cost-kind real --steps 4 --warmup 1 > $O/bench_line_c4_real.json 2> $O/bench_line_c4_real.err || exit 1
  run python3 bench.py --config c4 --cost-kind int --steps 4 --warmup 1 --cpu-seconds 0 > $O/bench_line_c4_int.json 2> $O/bench_line_c4_int.err || exit 1
fi
if [ "$WHAT" = c2b256 ] || [ "$WHAT" = all ]; then
  # secondary C2 line: 256 restarts, one workgroup per instance (staged kernel)
  run python3 bench.py --batch 256 --steps 20 --warmup 2 --cpu-seconds 0 --e2e-rounds 0 > $O/bench_line_c2_b256.json 2> $O/bench_line_c2_b256.err || exit 1
fi
if [ "$WHAT" = short ] || [ "$WHAT" = all ]; then
  for cfg in c1 c3 c5; do
    for b in 64 256 1024; do
      run python3 bench.py --config $cfg --batch $b --steps 30 --warmup 3 --cpu-seconds 6 --e2e-rounds 0 > $O/bench_line_${cfg}_b$b.json 2> $O/bench_line_${cfg}_b$b.err || exit 1
    done
  done
  for cfg in c1 c3; do   # the staged kernel on the same shapes, for comparison
    for b in 64 256 1024; do
      BXH_NARROW=0 python3 bench.py --config $cfg --batch $b --steps 30 --warmup 3 --cpu-seconds 0 --e2e-rounds 0 > $O/bench_line_${cfg}_b${b}_staged.json 2> $O/bench_line_${cfg}_b${b}_staged.err || exit 1
    done
  done
  run rocprofv3 --kernel-trace --stats --output-format csv -d $O/c1_stats -- python3 bench.py --config c1 --batch 256 --steps 30 --warmup 3 --cpu-seconds 0 --e2e-rounds 0 > $O/c1_stats.log 2>&1 || exit 1
  run rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- python3 bench.py --config c5 --batch 256 --steps 30 --warmup 3 --cpu-seconds 0 --e2e-rounds 0 > $O/c5_stats.log 2>&1 || exit 1
  run rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -- python3 bench.py --config c3 --batch 256 --steps 30 --warmup 3 --cpu-seconds 0 --e2e-rounds 0 > $O/c3_stats.log 2>&1 || exit 1
fi
echo done
