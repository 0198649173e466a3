#!/bin/bash
# Round-5 evidence, collected on the GPU box through gpurun (results land in gpurun_out/r05/;
# the summaries to be judged are copied into profiles/r05/ afterwards):
#   tools/profile_r05.sh c2      kernel stats, FETCH_SIZE / WRITE_SIZE passes and the bench line of C2
#   tools/profile_r05.sh short   c1 / c3 / c5: bench lines at 64 / 256 / 1024 restarts, kernel stats and --
#                                new in round 5 -- the PMC passes (FETCH_SIZE, WRITE_SIZE, SQ cycles) at
#                                256 restarts, so that "issue-bound, not memory-bound" has counters behind it
#   tools/profile_r05.sh multi   two ranks on the one GPU (gloo): the exchange inside the timed region
#   tools/profile_r05.sh host    tools/e2e_workers.py: host microseconds per round of optimize() by workers
#   tools/profile_r05.sh c4      configs[3] (100 000 x 1 000 000) with integer and with real costs: kernel stats,
#                                FETCH_SIZE / WRITE_SIZE passes, bench lines (and the integer line with every tied
#                                row sorted, BXH_WIDE_TIEDFAST=0)
# Each rocprofv3 command profiles `python3 bench.py` directly (no wrapper between `--` and the program).
set -o pipefail
WHAT=${1:-all}
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
run() { echo "== $*" >&2; "$@"; }
if [ "$WHAT" = c2 ] || [ "$WHAT" = all ]; then
  run rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_stats -- python3 bench.py --steps 30 --warmup 3 --cpu-seconds 0 --e2e-rounds 0 > $O/c2_stats.log 2>&1 || exit 1
  run rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c2_fetch -- python3 bench.py --steps 10 --warmup 2 --cpu-seconds 0 --e2e-rounds 0 > $O/c2_fetch.log 2>&1 || exit 1
  run rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c2_write -- python3 bench.py --steps 10 --warmup 2 --cpu-seconds 0 --e2e-rounds 0 > $O/c2_write.log 2>&1 || exit 1
  run rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/c2_sq -- python3 bench.py --steps 6 --warmup 2 --cpu-seconds 0 --e2e-rounds 0 > $O/c2_sq.log 2>&1 || exit 1
  run python3 bench.py --steps 60 --warmup 5 > $O/bench_line_c2.json 2> $O/bench_line_c2.err || exit 1
  run python3 bench.py --batch 256 --steps 20 --warmup 2 --cpu-seconds 0 --e2e-rounds 0 > $O/bench_line_c2_b256.json 2> $O/bench_line_c2_b256.err || exit 1
fi
if [ "$WHAT" = short ] || [ "$WHAT" = all ]; then
  for cfg in c1 c3 c5; do
    for b in 64 256 1024; do
      run python3 bench.py --config $cfg --batch $b --steps 30 --warmup 3 --cpu-seconds 6 --e2e-rounds 0 > $O/bench_line_${cfg}_b$b.json 2> $O/bench_line_${cfg}_b$b.err || exit 1
    done
    run rocprofv3 --kernel-trace --stats --output-format csv -d $O/${cfg}_stats -- python3 bench.py --config $cfg --batch 256 --steps 30 --warmup 3 --cpu-seconds 0 --e2e-rounds 0 > $O/${cfg}_stats.log 2>&1 || exit 1
    run rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${cfg}_fetch -- python3 bench.py --config $cfg --batch 256 --steps 12 --warmup 2 --cpu-seconds 0 --e2e-rounds 0 > $O/${cfg}_fetch.log 2>&1 || exit 1
    run rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${cfg}_write -- python3 bench.py --config $cfg --batch 256 --steps 12 --warmup 2 --cpu-seconds 0 --e2e-rounds 0 > $O/${cfg}_write.log 2>&1 || exit 1
    run rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/${cfg}_sq -- python3 bench.py --config $cfg --batch 256 --steps 12 --warmup 2 --cpu-seconds 0 --e2e-rounds 0 > $O/${cfg}_sq.log 2>&1 || exit 1
  done
fi
if [ "$WHAT" = multi ] || [ "$WHAT" = all ]; then
  BENCH_SHARE_GPU=1 run python3 bench.py --gpus 2 --steps 20 --warmup 3 --cpu-seconds 0 --e2e-rounds 0 --exchange-every 5 > $O/bench_line_c2_two_ranks_one_gpu.json 2> $O/bench_line_c2_two_ranks_one_gpu.err || exit 1
fi
if [ "$WHAT" = host ] || [ "$WHAT" = all ]; then
  run python3 tools/e2e_workers.py c1 200 > $O/e2e_workers_c1.jsonl 2>&1 || exit 1
  run python3 tools/e2e_workers.py c5 100 64 256 1024 > $O/e2e_workers_c5.jsonl 2>&1 || exit 1
  run python3 tools/e2e_workers.py c2 50 64 256 1024 > $O/e2e_workers_c2.jsonl 2>&1 || exit 1
fi
if [ "$WHAT" = c4 ]; then
  for kind in int real; do
    C4="python3 bench.py --config c4 --cost-kind $kind --warmup 1 --cpu-seconds 0"
    run rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_${kind}_stats -- $C4 --steps 4 > $O/c4_${kind}_stats.log 2>&1 || exit 1
    run rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c4_${kind}_fetch -- $C4 --steps 3 > $O/c4_${kind}_fetch.log 2>&1 || exit 1
    run rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c4_${kind}_write -- $C4 --steps 3 > $O/c4_${kind}_write.log 2>&1 || exit 1
  done
  run python3 bench.py --config c4 --cost-kind int --steps 4 --warmup 1 > $O/bench_line_c4_int.json 2> $O/bench_line_c4_int.err || exit 1
  run python3 bench.py --config c4 --cost-kind real --steps 4 --warmup 1 --cpu-seconds 0 > $O/bench_line_c4_real.json 2> $O/bench_line_c4_real.err || exit 1
  BXH_WIDE_TIEDFAST=0 run python3 bench.py --config c4 --cost-kind int --steps 4 --warmup 1 --cpu-seconds 0 > $O/bench_line_c4_int_sorted.json 2> $O/bench_line_c4_int_sorted.err || exit 1
fi
echo done
