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
    monkeypatch.setenv("BXH_SWAVE", "0")  # (the wave-per-row form would take the rows of up to 63 elements)
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
    monkeypatch.delenv("BXH_SWAVE", raising=False)
    inst = synth.config_c1("int") if name == "c1" else synth.config_c3("int")
    # (scp41's rows have at most 35 elements: the wave-per-row form, bxh_swave.hpp; aa04's up to 150)
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=70, sweeps=4, push_at=(2,),
             expect_form=7 if name == "c1" else 5)
    monkeypatch.setenv("BXH_SWAVE", "0")
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=70, sweeps=4, push_at=(2,), expect_form=5)


@pytest.mark.parametrize("name", ["aa04", "ne4-pm1-max", "ne2-npc2-scp"])
def test_narrow_two_workgroups_per_cu_build(oracle, bxh, name, monkeypatch):
    """The 256-register build of the four-wave kernel (what bxh_create launches when the batch
    exceeds the CUs and two LDS plans fit one: two workgroups per CU), forced onto small batches,
    and once by itself with more instances than CUs."""
    monkeypatch.setenv("BXH_NARROW_W", "4")
    monkeypatch.setenv("BXH_NARROW_OCC", "2")
    if name == "aa04":
        run_case(oracle, bxh, synth.config_c3("int"), 1, False, 0, 0.0, batch=5, sweeps=5, push_at=(2,),
                 expect_form=5)
        monkeypatch.delenv("BXH_NARROW_OCC")
        monkeypatch.delenv("BXH_NARROW_W")
        monkeypatch.delenv("BXH_NARROW", raising=False)
        run_case(oracle, bxh, synth.config_c3("real"), 1, False, 0, 0.0, batch=300, sweeps=2, expect_form=5)
    else:
        _, maker, float_type, maximize, order, x0p = [c for c in SHAPES if c[0] == name][0]
        run_case(oracle, bxh, maker(), float_type, maximize, order, x0p, batch=3, sweeps=8, push_at=(3, 6),
                 expect_form=5)


def test_narrow_lists_shorter_than_the_row_ring(oracle, bxh):
    """The constants of a row wait in a ring of eight, fetched four rows ahead: lists of 1 .. 9
    rows (the violated rows of later sweeps) end inside the ring's first round."""
    for m in (1, 2, 3, 4, 5, 8, 9):
        inst = synth.make_scp(m, 120, 0.3, "real", 400 + m)
        run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=2, sweeps=4, push_at=(2,), expect_form=5)


RUN_CASES = [
    ("scp-real", lambda: synth.make_scp(60, 400, 0.06, "real", 311), 1, 1, 0),
    ("spp-int-f32-reversing", lambda: synth.make_spp(40, 300, 0.06, "int", 312), 0, 0, 1),
    ("pm1-random-order", lambda: synth.make_pm1(50, 250, 0.06, "real", 313), 1, 3, 2),
]


@pytest.mark.parametrize("case", RUN_CASES, ids=[c[0] for c in RUN_CASES])
def test_run_of_many_sweeps_equals_single_sweeps(oracle, bxh, case):
    run_equals_single_sweeps(oracle, bxh, case, 5)


def run_equals_single_sweeps(oracle, bxh, case, form):
    """bxh_run -- solver_functor's loop (itm-solver-common.hpp:135-213) with up to
    `max_sweeps` sweeps per launch, kappa moved on the device by the host's own
    increments -- against the same loop driven sweep by sweep (one launch each,
    the path every other test pins on the oracle): same number of sweeps, same
    stop, same kappa, x, pi, P, engine; and the x kept for the sweep with the
    fewest violated rows."""
    _, maker, float_type, kind_unused, order = case
    inst = maker()
    F = np.float64 if float_type else np.float32
    s0 = oracle.Solver(inst, float_type=float_type, sort_mode=1, cost_norm=4, seed=5)
    cost, delta, theta = s0.cost(), s0.compute_delta(0.5), 0.5
    kappa_step, kappa_max, alpha, w_limit = F(2e-3), F(0.6), F(1.0), 3
    inc = np.array([float(F(kappa_step * F(np.power(F(r) / F(inst.m), alpha)))) for r in range(inst.m + 1)])
    batch = 3

    def fresh():
        d = bxh.DeviceBatch(inst, cost, batch, float_type=float_type)
        assert d.kernel_form == form, d.kernel_form
        for b in range(batch):
            d.set_x(b, random_x(inst.n, 0.02 * b, 90 + b))
            d.set_rng(b, 1000 + b)
        d.order_init(order)
        return d

    from helpers import random_x, bits_equal  # noqa: F401  (used by fresh)
    one, many = fresh(), fresh()
    # sweep by sweep, per instance state on the host
    kap = [F(0.01 * (b + 1)) for b in range(batch)]
    done = [None] * batch
    best = [(2 ** 31 - 1, None, None)] * batch
    sweeps = [0] * batch
    limit = 40
    for i in range(limit):
        act = np.array([done[b] is None for b in range(batch)], np.uint8)
        if not act.any():
            break
        st = one.sweep(np.array([float(k) for k in kap]), delta, theta, order=order, active=act)
        for b in range(batch):
            if done[b] is not None:
                continue
            sweeps[b] += 1
            rem = int(st["remaining"][b])
            if rem == 0:
                done[b] = 1
                continue
            if rem < best[b][0]:
                best[b] = (rem, i, one.get_x(b))
            if i > w_limit:
                kap[b] = F(kap[b] + F(inc[rem]))
            if kap[b] > kappa_max:
                done[b] = 2
    # the same in launches of 7 sweeps (each instance stops on its own; the
    # others of the batch go on: a launch leaves a stopped instance alone next time)
    kap2 = np.array([0.01 * (b + 1) for b in range(batch)])
    kap2 = np.array([float(F(k)) for k in kap2])
    done2, best2, sweeps2 = [None] * batch, [(2 ** 31 - 1, None, None)] * batch, [0] * batch
    i0 = [0] * batch
    while any(d is None and i0[b] < limit for b, d in enumerate(done2)):
        for b in range(batch):  # (one instance per launch: their loop indices differ once one has stopped)
            if done2[b] is not None or i0[b] >= limit:
                continue
            act = np.zeros(batch, np.uint8)
            act[b] = 1
            # run() takes no mask in this wrapper: use a one-instance view through kappa only
            stt, rs = many.run_masked(kap2, delta, theta, order, min(7, limit - i0[b]), i0[b], w_limit,
                                      float(kappa_max), inc, act)
            r = rs[b]
            sweeps2[b] += int(r["sweeps"])
            if r["best_remaining"] < best2[b][0]:
                best2[b] = (int(r["best_remaining"]), int(r["best_i"]), many.get_best_x(b))
            kap2[b] = r["kappa"]
            i0[b] += int(r["sweeps"])
            if r["stop"] in (1, 2):
                done2[b] = int(r["stop"])
    for b in range(batch):
        assert sweeps2[b] == sweeps[b], b
        assert done2[b] == done[b], b
        assert float(F(kap[b])) == kap2[b], b
        assert np.array_equal(one.get_x(b), many.get_x(b)), b
        assert bits_equal(one.get_pi(b), many.get_pi(b)), b
        assert bits_equal(one.get_P(b), many.get_P(b)), b
        assert one.get_rng(b) == many.get_rng(b), b
        assert best2[b][0] == best[b][0] and best2[b][1] == best[b][1], b
        if best[b][2] is not None:
            assert np.array_equal(best[b][2], best2[b][2]), b
    one.close()
    many.close()

