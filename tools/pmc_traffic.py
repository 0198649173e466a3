#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of
`bench.py`) into the per-launch HBM traffic figure bench.py reports as
`roofline.traffic`.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/w -- python3 bench.py ...
    python3 tools/pmc_traffic.py out/f/*/*_counter_collection.csv out/w/*/*_counter_collection.csv \
        "<config.workload of the bench line>" [kernel form, bxh_kernel_form()] > profiles/rNN/pmc_traffic[_cfg].json

Units and corrections follow MI355X_MICROARCH.md (HBM): both counters are in
KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B per lane)
loads at 64 bytes, so it is doubled; WRITE_SIZE is taken as is.
"""
import csv
import json
import statistics
import sys


def per_launch(path, counter, kernel="sweep_kernel"):
    vals = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    return vals[1:] if len(vals) > 2 else vals  # first launch runs on cold caches


def main():
    fetch = per_launch(sys.argv[1], "FETCH_SIZE")
    write = per_launch(sys.argv[2], "WRITE_SIZE")
    f_kib, w_kib = statistics.median(fetch), statistics.median(write)
    out = {
        "workload": sys.argv[3] if len(sys.argv) > 3 else None,
        "kernel": "bxh::sweep_kernel*",
        "kernel_form": int(sys.argv[4]) if len(sys.argv) > 4 else 44,
        "launches_sampled": [len(fetch), len(write)],
        "FETCH_SIZE_KiB_per_launch": f_kib,
        "WRITE_SIZE_KiB_per_launch": w_kib,
        "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B "
                      "requests at 64 B)",
        "traffic_bytes_per_launch": (2.0 * f_kib + w_kib) * 1024.0,
        "uncorrected_bytes_per_launch": (f_kib + w_kib) * 1024.0,
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
