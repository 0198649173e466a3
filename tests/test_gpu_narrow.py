"""-m gpu: the kernel for short rows (baryonyx_amd/csrc/bxh_narrow.hpp: one or
four waves per instance, rows of at most 256 elements, the next row's column
lines read ahead and staged in LDS, the previous row's P store recomputed in the
staged copy)
against the oracle, sweep by sweep, bit-exact on x / pi / P / violated rows /
engine state:

* every small case of test_gpu_parity.py (01 and +-1 solvers, min / max, push
  sweeps, integer costs with their tied rows, every row order of compute_order),
* 1, 2 and 4 row elements per lane (rows of up to 64 / 128 / 256 elements),
  columns of one and of two prefetched 16-entry pieces, and columns longer than
  what is prefetched (the rest is fetched on demand),
* consecutive rows that share most of their columns (every column of a row then
  needs the recomputed slot),
* the aa04 and scp41 shapes of BASELINE.json at 70 instances (more than one
  wave per CU).
"""
import numpy as np
import pytest

from baryonyx_amd import synth
from test_gpu_parity import CASES, ORDER_CASES, _custom, run_case

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["1", "4"], ids=["1wave", "4waves"])
def narrow_env(request, monkeypatch):
    """One wave per instance, and four (one per SIMD of the CU, two barriers per row)."""
    monkeypatch.delenv("BXH_GROUPS", raising=False)
    monkeypatch.delenv("BXH_FORCE_V1", raising=False)
    monkeypatch.setenv("BXH_NARROW", "1")
    monkeypatch.setenv("BXH_NARROW_W", request.param)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_narrow_small_cases(oracle, bxh, case):
    _, maker, float_type, maximize, order, x0p = case
    run_case(oracle, bxh, maker(), float_type, maximize, order, x0p, expect_form=5)


@pytest.mark.parametrize("case", ORDER_CASES, ids=[c[0] for c in ORDER_CASES])
def test_narrow_row_orders(oracle, bxh, case):
    _, maker, float_type, maximize, order, x0p, push_at = case
    run_case(oracle, bxh, maker(), float_type, maximize, order, x0p, sweeps=20, push_at=push_at,
             expect_form=5)


SHAPES = [
    # (name, maker, float_type, maximize, order, x0 density): NE / NPC of the kernel in the name
    ("ne1-npc1-scp", lambda: synth.make_scp(150, 900, 0.04, "real", 201), 1, False, 0, 0.0),
    ("ne2-npc1-spp", lambda: synth.make_spp(120, 2400, 0.04, "int", 202), 1, False, 0, 0.0),
    ("ne2-npc2-scp", lambda: synth.make_scp(400, 2000, 0.05, "real", 203), 1, False, 1, 0.01),
    ("ne4-npc1-scp-f32", lambda: synth.make_scp(100, 4000, 0.045, "int", 204), 0, False, 0, 0.0),
    ("ne4-pm1-max", lambda: synth.make_pm1(180, 3600, 0.05, "real", 205), 1, True, 2, 0.2),
    ("ne2-pm1-eq", lambda: synth.make_pm1(300, 1500, 0.06, "int", 206, equalities=True), 1, False, 0, 0.1),
    ("long-columns-on-demand", lambda: synth.make_scp(900, 400, 0.1, "real", 207), 1, False, 0, 0.05),
    ("long-columns-pm1-f32", lambda: synth.make_pm1(700, 300, 0.12, "int", 208), 0, False, 1, 0.1),
]


@pytest.mark.parametrize("case", SHAPES, ids=[c[0] for c in SHAPES])
def test_narrow_lane_and_piece_counts(oracle, bxh, case):
    _, maker, float_type, maximize, order, x0p = case
    inst = maker()
    run_case(oracle, bxh, inst, float_type, maximize, order, x0p, batch=3, sweeps=8, push_at=(3, 6),
             expect_form=5)


def overlapping_rows(kind, seed):
    """Consecutive rows share most of their columns (a sliding window over the
    variables): nearly every column of a row holds the slot the previous row has
    just written."""
    rs = np.random.RandomState(seed)
    n, m, width = 300, 120, 90
    rows = []
    for k in range(m):
        start = (k * 2) % (n - width)
        cols = np.arange(start, start + width)
        rows.append(np.sort(rs.choice(cols, size=rs.randint(40, width), replace=False)))
    nnz = sum(len(r) for r in rows)
    if kind == 3:
        fac = np.where(rs.random_sample(nnz) < 0.3, -1, 1)
        cmin, cmax, at = [], [], 0
        for r in rows:
            negs = int((fac[at:at + len(r)] < 0).sum())
            at += len(r)
            cmin.append(1 - negs)
            cmax.append(3 - negs)
        return _custom(rows, n, cmin, cmax, rs.randint(1, 40, n).astype(float), factor=fac, kind=3)
    return _custom(rows, n, [1] * m, [2 ** 30] * m, rs.random_sample(n) * 10 + 1, kind=1)


@pytest.mark.parametrize("kind", [1, 3], ids=["ineq01", "ineq101"])
@pytest.mark.parametrize("order", [0, 1])
def test_narrow_rows_sharing_columns(oracle, bxh, kind, order):
    inst = overlapping_rows(kind, 300 + kind)
    run_case(oracle, bxh, inst, 1, False, order, 0.1, batch=3, sweeps=10, push_at=(4, 7), expect_form=5)


@pytest.mark.parametrize("name", ["c1", "c3"])
def test_narrow_baseline_shapes_many_instances(oracle, bxh, name, monkeypatch):
    """scp41 / aa04 shapes with 70 instances (more waves than one per CU on a
    quarter of the chip), and the kernel chosen by bxh_create itself."""
    monkeypatch.delenv("BXH_NARROW", raising=False)
    inst = synth.config_c1("int") if name == "c1" else synth.config_c3("int")
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=70, sweeps=4, push_at=(2,), expect_form=5)
