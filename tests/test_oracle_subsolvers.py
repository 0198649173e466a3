"""CPU: the oracle's branch-and-bound and exhaustive row sub-solvers against the
reference's own known answers (lib/test/lib.cpp:376-563, restated)."""
import ctypes as C

import numpy as np
import pytest


def _sum(values, factors, r_size, result):
    """`sum` helper of lib/test/lib.cpp:38-51."""
    e = min(r_size - 1, result)
    return float(sum(values[: e + 1])), int(sum(factors[: e + 1]))


def _arrays(R):
    v = np.array([r[0] for r in R], np.float64)
    i = np.array([r[1] for r in R], np.int32)
    f = np.array([r[2] for r in R], np.int32)
    return v, i, f


BB_CASES = [
    ([(7.0, 0, 13), (4.0, 1, 12), (3.0, 2, 8), (3.0, 3, 10)], 0, 30, 25, 11.0),
    ([(16.0, 0, 2), (19.0, 1, 3), (23.0, 2, 4), (26.0, 3, 5)], 0, 7, 7, 42.0),
    ([(1.0, 0, -1), (1.0, 1, -1), (1.0, 2, -1), (1.0, 3, 1)], -2, -2, -2, 2.0),
]


@pytest.mark.parametrize("R,bmin,bmax,f_exp,r_exp", BB_CASES)
def test_branch_and_bound_known_answers(oracle, R, bmin, bmax, f_exp, r_exp):
    v, i, f = _arrays(R)
    L = oracle.lib()
    res = L.orc_bb_solve(1, len(R), v.ctypes.data_as(oracle.f64p), i.ctypes.data_as(oracle.i32p),
                         f.ctypes.data_as(oracle.i32p), bmin, bmax)
    assert -1 <= res <= len(R)
    r, ff = _sum(v, f, len(R), res)
    assert ff == f_exp and r == r_exp


EX_CASES = [
    ([(7.0, 0, 13), (4.0, 1, 12), (3.0, 2, 8), (3.0, 3, 10)], 0, 30, 25, 11.0),
    ([(16.0, 0, 2), (19.0, 1, 3), (23.0, 2, 4), (26.0, 3, 5)], 0, 7, 7, 42.0),
    ([(1.0, 0, -1), (1.0, 1, -1), (1.0, 2, -1), (1.0, 3, 1)], -2, -2, -2, 4.0),
    ([(1.0, 0, -1), (-0.5, 1, -1)], -2, -1, -1, 1.0),
]


@pytest.mark.parametrize("R,bmin,bmax,f_exp,r_exp", EX_CASES)
def test_exhaustive_known_answers(oracle, R, bmin, bmax, f_exp, r_exp):
    v, i, f = _arrays(R)
    row = f.copy()
    L = oracle.lib()
    res = L.orc_exhaustive_solve(1, len(R), row.ctypes.data_as(oracle.i32p), bmin, bmax,
                                 v.ctypes.data_as(oracle.f64p), i.ctypes.data_as(oracle.i32p),
                                 f.ctypes.data_as(oracle.i32p))
    assert -1 <= res <= len(R)
    r, ff = _sum(v, f, len(R), res)
    assert ff == f_exp and r == r_exp
