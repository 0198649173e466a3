#!/usr/bin/env python3
"""Writes tests/golden/vectors.json: sweep-by-sweep digests of the CPU oracle on
small seeded instances (run from the repo root: `python tests/golden/make_vectors.py`).

These are regression vectors of the ORACLE (they were not produced by the
reference, which cannot be built here): `tests/test_oracle_vectors.py` checks
that the oracle still reproduces them, `tests/test_gpu_parity.py` that the HIP
path does.  Instances come from baryonyx_amd.synth generators (seeded), so only
the generator arguments are stored, not the matrices."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baryonyx_amd import synth  # noqa: E402
from oracle import oracle_ffi as O  # noqa: E402

CASES = [
    # name, generator, args, float_type, maximize, order, push sweeps
    ("scp-int", "make_scp", [60, 400, 0.06, "int", 11], 1, False, 0, [3]),
    ("scp-real-f32-rand", "make_scp", [50, 300, 0.08, "real", 15], 0, False, 2, [2, 5]),
    ("spp-int-max", "make_spp", [40, 300, 0.06, "int", 17], 1, True, 1, []),
    ("pm1-ineq-real", "make_pm1", [50, 250, 0.06, "real", 20], 1, False, 5, [4]),
    ("pm1-eq-int", "make_pm1", [40, 250, 0.06, "int", 21, True], 1, False, 3, [3]),
    ("zmix-int-max", "make_zmix", [24, 60, "int", 6, 13, True], 1, True, 0, [4]),
    ("z-knapsack-real", "make_z", [20, 60, 0.2, "real", 12], 1, True, 6, []),
]
SWEEPS = 6


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]


def run(case):
    name, gen, args, ft, mx, order, pushes = case
    inst = getattr(synth, gen)(*args)
    s = O.Solver(inst, float_type=ft, sort_mode=1, cost_norm=4, seed=4242, maximize=mx)
    s.set_x(np.zeros(s.nwords, np.uint64))
    s.order_init(order)
    delta = s.compute_delta(0.5)
    out = []
    kappa = 0.0
    for it in range(SWEEPS):
        push = it in pushes
        rem = s.sweep(kappa, delta, 0.5, is_push=push, obj_amp=5.0 if push else 0.0)
        out.append(dict(remaining=int(rem), rng=int(s.rng), x=digest(s.get_x()),
                        pi=digest(s.get_pi()), P=digest(s.get_P())))
        kappa += 0.03
    return dict(name=name, generator=gen, args=args, float_type=ft, maximize=mx, order=order,
                pushes=pushes, seed=4242, theta=0.5, kappa_step=0.03, obj_amp=5.0, sweeps=out)


if __name__ == "__main__":
    vec = [run(c) for c in CASES]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors.json")
    with open(path, "w") as f:
        json.dump(vec, f, indent=1)
    print("wrote", path, len(vec), "cases")
