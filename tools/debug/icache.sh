#!/bin/bash
# instruction-cache and issue counters of the bench kernel (diagnostic; own PMC pass)
export TMPDIR=/tmp
mkdir -p gpurun_out/icache
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --kernel-trace --output-format csv -d gpurun_out/icache/run -- python3 bench.py --steps 6 --warmup 2 --cpu-seconds 0 --e2e-rounds 0 > gpurun_out/icache/log.txt 2>&1
python3 - <<'P'
import csv,glob,collections
fs=glob.glob("gpurun_out/icache/run/*/*_counter_collection.csv")
acc=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "sweep_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, len(v), sum(v[1:])/max(1,len(v)-1))
P
