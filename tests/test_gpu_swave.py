"""-m gpu: the 01 / 101 solvers' short-row kernel in its wave-per-row form
(baryonyx_amd/csrc/bxh_swave.hpp, `bxh_kernel_form` 7: every row of 1 .. 63
elements; a wave works a row, up to four rows of the list that share no column
side by side; tie shuffles, coin flips and R's leftovers in row order) against
the oracle, sweep by sweep, bit-exact on x / pi / P / violated rows / engine:

* every small case and every row order of test_gpu_parity.py whose rows fit,
* instances built to form groups of 2 - 4 rows: few elements per row over many
  columns, integer costs (tied rows next to tie-free ones), all four kinds,
* zero reduced costs (the scan of the inequality solvers flips coins) and rows
  whose scan runs on to R[r] (covering rows with nothing that stops it),
* long lists (several tables of 64 rows), columns of more than 16 entries,
* the scp41 shape of BASELINE.json and SAT-like rows of three elements.
"""
import numpy as np
import pytest

from baryonyx_amd import synth
from test_gpu_parity import CASES, ORDER_CASES, _custom, run_case

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def swave_env(monkeypatch):
    monkeypatch.delenv("BXH_GROUPS", raising=False)
    monkeypatch.delenv("BXH_FORCE_V1", raising=False)
    monkeypatch.delenv("BXH_NARROW", raising=False)
    monkeypatch.setenv("BXH_SWAVE", "1")


def _fits(inst):
    r = np.diff(inst.row_ptr)
    return r.min() >= 1 and r.max() <= 63 and len(inst.q_a) == 0


SMALL = [c for c in CASES if _fits(c[1]())]
ORDERS = [c for c in ORDER_CASES if _fits(c[1]())]


def test_the_lists_are_not_empty():
    assert len(SMALL) >= 4 and len(ORDERS) >= 4


@pytest.mark.parametrize("case", SMALL, ids=[c[0] for c in SMALL])
def test_swave_small_cases(oracle, bxh, case):
    _, maker, float_type, maximize, order, x0p = case
    run_case(oracle, bxh, maker(), float_type, maximize, order, x0p, expect_form=7)


@pytest.mark.parametrize("case", ORDERS, ids=[c[0] for c in ORDERS])
def test_swave_row_orders(oracle, bxh, case):
    _, maker, float_type, maximize, order, x0p, push_at = case
    run_case(oracle, bxh, maker(), float_type, maximize, order, x0p, sweeps=20, push_at=push_at, expect_form=7)


GROUPS = [
    # (name, maker, float_type, maximize, order, x0 density)
    ("scp-int", lambda: synth.make_scp(220, 2500, 0.008, "int", 501), 1, False, 0, 0.0),
    ("scp-real-rev", lambda: synth.make_scp(180, 1800, 0.01, "real", 502), 1, False, 1, 0.02),
    ("spp-int-f32", lambda: synth.make_spp(150, 2600, 0.01, "int", 503), 0, False, 0, 0.0),
    ("pm1-ineq-max-random", lambda: synth.make_pm1(200, 2400, 0.01, "int", 504), 1, True, 2, 0.2),
    ("pm1-eq-real", lambda: synth.make_pm1(160, 2000, 0.012, "real", 505, equalities=True), 1, False, 0, 0.1),
    ("pm1-ineq-lagrangian", lambda: synth.make_pm1(140, 1500, 0.015, "int", 506), 1, False, 5, 0.3),
    ("scp-infeasibility-order", lambda: synth.make_scp(130, 1600, 0.012, "int", 507), 1, False, 3, 0.0),
    ("long-columns", lambda: synth.make_scp(700, 500, 0.05, "real", 508), 1, False, 0, 0.05),
    ("long-columns-pm1-f32", lambda: synth.make_pm1(600, 400, 0.06, "int", 509), 0, False, 1, 0.1),
    ("scp41-shape", lambda: synth.config_c1("int"), 1, False, 0, 0.0),
]


@pytest.mark.parametrize("case", GROUPS, ids=[c[0] for c in GROUPS])
def test_swave_groups(oracle, bxh, case):
    _, maker, float_type, maximize, order, x0p = case
    inst = maker()
    assert _fits(inst), np.diff(inst.row_ptr).max()
    run_case(oracle, bxh, inst, float_type, maximize, order, x0p, batch=3, sweeps=10, push_at=(3, 7),
             expect_form=7)


def zero_cost_cover(seed, kind):
    """Covering / packing rows over variables most of which cost NOTHING: reduced
    costs of exactly zero (stop_iterating flips a coin, itm-common.hpp:150-163) and
    scans that run on to R[r] (nothing stops them)."""
    rs = np.random.RandomState(seed)
    n, m = 400, 160
    rows = [np.sort(rs.choice(n, size=rs.randint(2, 12), replace=False)) for _ in range(m)]
    cost = np.where(rs.random_sample(n) < 0.6, 0.0, rs.randint(1, 5, n)).astype(float)
    nnz = sum(len(r) for r in rows)
    if kind == 3:
        fac = np.where(rs.random_sample(nnz) < 0.3, -1, 1)
        cmin, cmax, at = [], [], 0
        for r in rows:
            negs = int((fac[at:at + len(r)] < 0).sum())
            at += len(r)
            cmin.append(1 - negs)
            cmax.append(len(r))
        return _custom(rows, n, cmin, cmax, cost, factor=fac, kind=3)
    return _custom(rows, n, [1] * m, [2 ** 30] * m, cost, kind=1)


@pytest.mark.parametrize("maximize", [False, True], ids=["min", "max"])
@pytest.mark.parametrize("kind", [1, 3], ids=["ineq01", "ineq101"])
def test_swave_coins_and_leftovers(oracle, bxh, kind, maximize):
    inst = zero_cost_cover(600 + kind, kind)
    run_case(oracle, bxh, inst, 1, maximize, 0, 0.1, batch=3, sweeps=10, push_at=(4,), expect_form=7)


def sat_like(seed):
    """Clauses of three literals (a row `sum +-x >= 1 - negatives`): what the
    reference's SAT fixtures look like (flat30-7, uf50-0448)."""
    rs = np.random.RandomState(seed)
    n, m = 150, 500
    rows = [np.sort(rs.choice(n, size=3, replace=False)) for _ in range(m)]
    fac = np.where(rs.random_sample(3 * m) < 0.5, -1, 1)
    cmin = [1 - int((fac[3 * k:3 * k + 3] < 0).sum()) for k in range(m)]
    # (a few variables cost something: an all-zero objective has nothing to normalise by)
    cost = np.where(rs.random_sample(n) < 0.1, 1.0, 0.0)
    return _custom(rows, n, cmin, [3] * m, cost, factor=fac, kind=3)


@pytest.mark.parametrize("order", [0, 2])
def test_swave_sat_like_rows(oracle, bxh, order):
    run_case(oracle, bxh, sat_like(700 + order), 1, False, order, 0.5, batch=2, sweeps=8, push_at=(),
             expect_form=7)


def test_swave_run_of_many_sweeps_equals_single_sweeps(oracle, bxh):
    """bxh_run on the wave-per-row form: launches of up to seven sweeps of solver_functor's
    loop (kappa moved on the device, the row list of a sweep read by the next) against the same
    loop driven sweep by sweep -- sweeps, stop, kappa, x, pi, P, engine and the x of the best sweep."""
    from test_gpu_narrow import RUN_CASES, run_equals_single_sweeps

    for case in RUN_CASES:
        assert _fits(case[1]())
        run_equals_single_sweeps(oracle, bxh, case, 7)
    # a list of several tables (more than 64 rows violated at the start), random row order
    run_equals_single_sweeps(oracle, bxh, ("scp-200", lambda: synth.make_scp(200, 1000, 0.02, "int", 41), 1, 1, 2), 7)
