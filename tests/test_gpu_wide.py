"""-m gpu: the wide kernel (baryonyx_amd/csrc/bxh_wide.hpp: columns of any
length, pi beyond LDS, G workgroups sharing every row -- the shape of
BASELINE.json configs[3], synthetic SCP 100k x 1M) against the oracle, sweep by
sweep, bit-exact on x / pi / P / violated rows / engine state:

* the small cases of test_gpu_parity.py forced onto it with 1, 2 and 3
  workgroups per instance (01 and +-1 solvers, min / max, push sweeps, ties),
* ragged rows, rows longer than the LDS sort path (the global-memory sort),
  columns of hundreds of entries,
* an XL-shaped reduced instance (m >= 40 000 rows so that pi cannot live in
  LDS, columns of ~2 000 entries), which bxh_create sends to the wide kernel by
  itself.
"""
import numpy as np
import pytest

from baryonyx_amd import synth
from test_gpu_parity import CASES, EDGE_CASES, _custom, run_case

pytestmark = pytest.mark.gpu

WIDE_CASES = [c for c in CASES if c[4] in (0, 1, 2)]


@pytest.mark.parametrize("groups", ["w1", "w2", "w3"])
@pytest.mark.parametrize("case", WIDE_CASES, ids=[c[0] for c in WIDE_CASES])
def test_wide_kernel_small_cases(oracle, bxh, case, groups, monkeypatch):
    monkeypatch.delenv("BXH_FORCE_V1", raising=False)
    monkeypatch.setenv("BXH_GROUPS", groups)
    _, maker, float_type, maximize, order, x0p = case
    run_case(oracle, bxh, maker(), float_type, maximize, order, x0p, expect_form=100 + int(groups[1:]))


@pytest.mark.parametrize("groups", ["w1", "w4"])
@pytest.mark.parametrize("case", EDGE_CASES, ids=[c[0] for c in EDGE_CASES])
def test_wide_kernel_edge_shapes(oracle, bxh, case, groups, monkeypatch):
    """rows of one element next to a row over every variable, interior ranks,
    a row of 1300 elements (beyond the LDS sort: sorted in global memory),
    columns of 300 entries."""
    monkeypatch.setenv("BXH_GROUPS", groups)
    _, maker, float_type, maximize, order, x0p = case
    run_case(oracle, bxh, maker(), float_type, maximize, order, x0p, sweeps=8, push_at=(3, 6))


def test_wide_kernel_long_tied_rows(oracle, bxh, monkeypatch):
    """Rows of ~3000 elements with integer costs: every row of the first sweep
    holds runs of equal reduced costs (the long-row sort and its shuffles)."""
    monkeypatch.setenv("BXH_GROUPS", "w4")
    inst = synth.make_scp(40, 30000, 0.1, "int", 95)
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=2, sweeps=3, push_at=(2,), expect_form=104)


def test_wide_kernel_is_chosen_for_long_columns(oracle, bxh, monkeypatch):
    """Columns of ~300 entries: bxh_create picks the wide kernel by itself."""
    monkeypatch.delenv("BXH_GROUPS", raising=False)
    monkeypatch.delenv("BXH_FORCE_V1", raising=False)
    inst = synth.make_scp(3000, 600, 0.1, "real", 96)
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=2, sweeps=3, push_at=(2,))
    inst = synth.make_pm1(2500, 500, 0.08, "int", 97)
    run_case(oracle, bxh, inst, 0, True, 1, 0.1, batch=3, sweeps=3, push_at=(1,))


@pytest.mark.parametrize("groups", [None, "w4"], ids=["auto", "w4"])
@pytest.mark.parametrize("float_type", [1, 0], ids=["f64", "f32"])
def test_xl_shaped_reduced_instance(oracle, bxh, float_type, groups, monkeypatch):
    """m = 40 000 rows (pi: 320 KB, beyond LDS), n = 5 000, 5 % density:
    columns of ~2 000 entries, rows of ~250 -- configs[3]'s column shape at a
    size the oracle replays in seconds."""
    monkeypatch.delenv("BXH_GROUPS", raising=False)
    if groups:
        monkeypatch.setenv("BXH_GROUPS", groups)
    monkeypatch.delenv("BXH_FORCE_V1", raising=False)
    inst = synth.make_scp(40000, 5000, 0.05, "real" if float_type else "int", 98)
    theta = 0.5
    batch = 2
    orcs = [oracle.Solver(inst, float_type=float_type, sort_mode=1, cost_norm=4, seed=500 + b)
            for b in range(batch)]
    dev = bxh.DeviceBatch(inst, orcs[0].cost(), batch, float_type=float_type)
    assert dev.kernel_form > 100, dev.kernel_form
    delta = orcs[0].compute_delta(theta)
    # x0: everything set except the variables of 60 rows -> those rows (and
    # hardly any other) are violated: a sweep of ~60 row updates per instance
    from helpers import pack_bits
    rs = np.random.RandomState(7)
    for b, s in enumerate(orcs):
        rows = rs.choice(inst.m, size=60, replace=False)
        x = np.ones(inst.n, np.uint8)
        for k in rows:
            x[inst.col[inst.row_ptr[k]:inst.row_ptr[k + 1]]] = 0
        x0 = pack_bits(x)
        s.set_x(x0)
        dev.set_x(b, x0)
        dev.set_rng(b, s.rng)
        s.order_init(0)
    rem = dev.order_init(0)
    for b, s in enumerate(orcs):
        assert rem[b] == len(s.violated()) and rem[b] >= 60
        assert np.array_equal(dev.violated(b), s.violated())
    from helpers import bits_equal, first_diff
    kappa = np.array([0.05, 0.1])
    for it in range(3):
        st = dev.sweep(kappa, delta, theta)
        for b, s in enumerate(orcs):
            r = s.sweep(float(kappa[b]), delta, theta)
            tag = f"sweep {it} inst {b}"
            assert st["remaining"][b] == r, tag
            assert np.array_equal(dev.get_x(b), s.get_x()), tag + " x"
            pd, po = dev.get_pi(b), s.get_pi()
            assert bits_equal(pd, po), f"{tag} pi first diff at {first_diff(pd, po)}"
            Pd, Po = dev.get_P(b), s.get_P()
            assert bits_equal(Pd, Po), f"{tag} P first diff at {first_diff(Pd, Po)}"
            assert st["rng"][b] == s.rng, tag + " rng"
            assert st["row_updates"][b] == s.row_updates, tag
        kappa = kappa + 0.05
    dev.close()
