"""Trace of the oracle (CPU restatement; test infrastructure) on small4.lp with the
parameters of lib/test/optimize.cpp:55-71 as context_set_solver_parameters leaves
them (kappa 0 / 1e-3 / 0.6, w = 20, alpha = 0, theta 0.5, delta 0.2): the number
of violated rows per sweep from the Bastert point, and how often a `sum y - x = 0`
row re-opens a `sum x = 1` row.  Used for the note in tests/test_reference_held.py.

    python tools/trace_small4.py [sweeps]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baryonyx_amd import host_ffi as H  # noqa: E402
from oracle import oracle_ffi as O  # noqa: E402


def main():
    sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 700
    inst = H.merged(os.path.join(ROOT, "tests", "golden", "lp", "small4.lp"))
    s = O.Solver(inst, float_type=1, sort_mode=0, cost_norm=4, seed=1)
    s.set_x(np.zeros(s.nwords, np.uint64))
    s.order_init(0)
    kappa, best = 0.0, inst.m
    xrows_open = 0
    for it in range(sweeps):
        r = s.sweep(kappa, 0.2, 0.5)
        v = s.violated()
        best = min(best, r)
        xrows_open += int((v < 8).sum())
        if it < 10 or it % 50 == 0:
            print(f"sweep {it:4d} kappa {kappa:.3f} violated {r:3d} (x rows: {int((v < 8).sum())}) best {best}")
        if r == 0:
            print("feasible at sweep", it)
            return
        if it > 20:
            kappa += 1e-3
        if kappa > 0.6:
            print(f"kappa_max reached at sweep {it}: {r} rows violated, best {best}; "
                  f"x rows open on average {xrows_open / (it + 1):.1f} of 8")
            return


if __name__ == "__main__":
    main()
