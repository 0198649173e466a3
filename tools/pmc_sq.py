#!/usr/bin/env python3
"""Summarise an SQ-counter pass of rocprofv3 (tools/profile_r05.sh: --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS) for the sweep kernel: medians per launch
and the shares of the wave cycles (MI355X_MICROARCH.md, rocprofv3 PMC slots: SQ_WAIT_ANY = wave
parked at s_waitcnt / a barrier, SQ_ACTIVE_INST_ANY = issuing; the counters count quad-cycles).

    python3 tools/pmc_sq.py out/sq/*/*_counter_collection.csv [kernel substring] > profiles/rNN/<cfg>_pmc_sq.json
"""
import csv
import json
import statistics
import sys


def main():
    path = sys.argv[1]
    kernel = sys.argv[2] if len(sys.argv) > 2 else "sweep_kernel"
    vals = {}
    names = set()
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"]:
                names.add(row["Kernel_Name"].split("(")[0])
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    med = {k: statistics.median(v[1:] if len(v) > 2 else v) for k, v in vals.items()}
    wc = med.get("SQ_WAVE_CYCLES", 0.0)
    out = {"kernel": sorted(names), "launches_sampled": {k: len(v) for k, v in vals.items()},
           "median_per_launch": med,
           "share_of_wave_cycles": {k: (v / wc if wc else None) for k, v in med.items() if k != "SQ_WAVE_CYCLES"}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
