#!/usr/bin/env python3
"""The reference CLI's benchmark mode (`baryonyx --bench name file.csv`,
app/src/benchmark.cpp:613-774) over the MI355X host library: `file.csv` has a
header `file,<solver>,...` and one row per instance with the value every solver
found (`inf`: none); the instances are `<csv stem>/<model>.lp` (or `.scp`, the
OR-Library form).  Every instance is run through baryonyx::optimize, the values
go into a new column `name`, and `<csv stem>-stats.csv` ranks the solvers per
instance.

  python tools/bx_bench.py benchmarks/scp.csv --name bx-mi355x -p time_limit:60 -p thread:64
"""
from __future__ import annotations

import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fmt10g(v: float) -> str:
    return "inf" if math.isinf(v) and v > 0 else ("-inf" if math.isinf(v) else f"{v:.10g}")


def load(path):
    """-> (solver names, [(model, [values])])  (bench::load, :332-470)"""
    with open(path) as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    head = [t.strip() for t in lines[0].split(",")]
    if not head or head[0] != "file":
        raise ValueError(f"{path}: the header must start with `file`")
    solvers = head[1:]
    rows = []
    for no, ln in enumerate(lines[1:], 2):
        tok = [t.strip() for t in ln.split(",")]
        if len(tok) != len(head) or not tok[0]:
            raise ValueError(f"{path}:{no}: {len(head)} columns expected")
        rows.append((tok[0], [float(t) if t else math.inf for t in tok[1:]]))
    return solvers, rows


def save(path, solvers, rows, name, current):
    with open(path, "w") as f:
        f.write(",".join(["file", *solvers, name]) + "\n")
        for (model, vals), cur in zip(rows, current):
            f.write(",".join([model, *[fmt10g(v) for v in vals], fmt10g(cur)]) + "\n")


def save_stats(path, solvers, rows, name, current, maximize):
    """rank of every solver per instance (1 = best; ties share a rank)"""
    with open(path, "w") as f:
        f.write(",".join(["file", *solvers, name]) + "\n")
        for (model, vals), cur, mx in zip(rows, current, maximize):
            allv = vals + [cur]
            key = [(-v if mx else v) if math.isfinite(v) else math.inf for v in allv]
            order = sorted(set(key))
            f.write(",".join([model, *[str(order.index(k) + 1) for k in key]]) + "\n")


def instance_path(csv_path, model):
    d = csv_path[: csv_path.rfind(".csv")] + os.sep
    for cand in (model, model + ".lp", model + ".scp"):
        if os.path.isfile(d + cand):
            return d + cand
    return d + (model if model.endswith(".lp") else model + ".lp")


def run(csv_path, name, params, runner=None):
    from baryonyx_amd import host_ffi as H

    solvers, rows = load(csv_path)
    current, maximize = [], []
    for model, _ in rows:
        path = instance_path(csv_path, model)
        print(f"- optimizing {model} ({path})", flush=True)
        try:
            if runner is not None:
                value, mx = runner(path)
            else:
                info = H.parse(path)
                if info.status != 0:
                    raise RuntimeError(f"parse error {info.status}")
                res, _ = H.run(path, 1, H.default_params(**params))
                value, mx = (res.value if res.status == 0 else math.inf), bool(info.maximize)
        except Exception as e:  # as the reference: report, keep going
            print(f"  failure: {e}", file=sys.stderr)
            value, mx = math.inf, False
        current.append(value)
        maximize.append(mx)
    found = sum(math.isfinite(v) for v in current)
    print(f"{name} found {found} feasible solution(s) / {len(current)}")
    for j, s in enumerate(solvers):
        d = [((vals[j] - c) / vals[j]) * 100.0 for (_, vals), c in zip(rows, current)
             if math.isfinite(vals[j]) and math.isfinite(c) and vals[j] != 0]
        if d:
            print(f"Average distance from the solution of {s}: {sum(d) / len(rows):.10g}%")
    save(csv_path, solvers, rows, name, current)
    save_stats(csv_path[: csv_path.rfind(".")] + "-stats.csv", solvers, rows, name, current, maximize)
    return current


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--name", default="bx-mi355x")
    ap.add_argument("-p", action="append", default=[], metavar="name:value",
                    help="solver parameter (as the CLI's -p; e.g. time_limit:60 thread:64 seed:1)")
    a = ap.parse_args()
    params = {}
    for kv in a.p:
        k, v = kv.split(":", 1)
        k = k.replace("-", "_")
        params[k] = float(v) if any(c in v for c in ".e") else int(v)
    params.setdefault("time_limit", 10.0)
    run(a.csv, a.name, params)


if __name__ == "__main__":
    main()
