"""-m gpu: the HIP row update (through the C ABI) against the CPU oracle, sweep
by sweep, on the same seeded inputs.  Bit-exact on x / violated rows /
remaining / RNG state, and -- because the kernel keeps the reference's
summation order -- bit-exact on pi and P too (tolerance 0; north_star allows
1e-6 relative)."""
import numpy as np
import pytest

from baryonyx_amd import synth
from helpers import bits_equal, first_diff, random_x

pytestmark = pytest.mark.gpu

CASES = [
    # (name, maker, float_type, maximize, order, x0_density)
    ("scp-real-f64", lambda: synth.make_scp(60, 400, 0.06, "real", 11), 1, False, 0, 0.0),
    ("scp-int-f64", lambda: synth.make_scp(60, 400, 0.06, "int", 12), 1, False, 0, 0.0),
    ("scp-int-f32", lambda: synth.make_scp(60, 400, 0.06, "int", 13), 0, False, 0, 0.02),
    ("scp-real-rev", lambda: synth.make_scp(50, 300, 0.08, "real", 14), 1, False, 1, 0.01),
    ("scp-int-rand", lambda: synth.make_scp(50, 300, 0.08, "int", 15), 1, False, 2, 0.01),
    ("scp-int-max", lambda: synth.make_scp(50, 300, 0.08, "int", 16), 1, True, 0, 0.5),
    ("spp-int-f64", lambda: synth.make_spp(40, 300, 0.06, "int", 17), 1, False, 0, 0.0),
    ("spp-real-f32", lambda: synth.make_spp(40, 300, 0.06, "real", 18), 0, False, 0, 0.05),
    ("pm1-ineq-int", lambda: synth.make_pm1(50, 250, 0.06, "int", 19), 1, False, 0, 0.1),
    ("pm1-ineq-real-max", lambda: synth.make_pm1(50, 250, 0.06, "real", 20), 1, True, 0, 0.3),
    ("pm1-eq-int", lambda: synth.make_pm1(40, 250, 0.06, "int", 21, equalities=True), 1, False, 0, 0.1),
    ("pm1-eq-real-f32", lambda: synth.make_pm1(40, 250, 0.06, "real", 22, equalities=True), 0, False, 2, 0.1),
    ("c1-scp41-shaped", lambda: synth.config_c1("int"), 1, False, 0, 0.0),
]


def run_case(oracle, bxh, inst, float_type, maximize, order, x0p, batch=3, sweeps=12, push_at=(5, 9),
             expect_form=None):
    theta = 0.5
    orcs = []
    for b in range(batch):
        s = oracle.Solver(inst, float_type=float_type, sort_mode=1, cost_norm=4,
                          seed=1000 + b, maximize=maximize)
        orcs.append(s)
    cost = orcs[0].cost()
    delta = orcs[0].compute_delta(theta)
    dev = bxh.DeviceBatch(inst, cost, batch, float_type=float_type, maximize=maximize,
                          qcost=orcs[0].qcost())
    assert dev.rmax == orcs[0].rmax
    if expect_form is not None:
        assert dev.kernel_form == expect_form, dev.kernel_form

    for b, s in enumerate(orcs):
        x0 = random_x(inst.n, x0p, 77 + b) if x0p > 0 else np.zeros(s.nwords, np.uint64)
        s.set_x(x0)
        dev.set_x(b, x0)
        dev.set_rng(b, s.rng)
        s.order_init(order)
    rem = dev.order_init(order)
    for b, s in enumerate(orcs):
        assert np.array_equal(dev.violated(b), s.violated()), f"initial violated list, inst {b}"
        if order != 7:  # pi_sign_change walks iota(n); remaining counts the violated rows
            assert rem[b] == len(s.violated())

    kappa = np.array([0.0 + 0.01 * (b % 64) for b in range(batch)])  # (kappa < 1)
    for it in range(sweeps):
        push = it in push_at
        amp = 5.0 if push else 0.0
        st = dev.sweep(kappa, delta, theta, is_push=push, obj_amp=amp, order=order)
        for b, s in enumerate(orcs):
            r = s.sweep(float(kappa[b]), delta, theta, is_push=push, obj_amp=amp)
            tag = f"sweep {it} inst {b}"
            assert st["remaining"][b] == r, tag
            assert np.array_equal(dev.get_x(b), s.get_x()), tag + " x"
            assert np.array_equal(dev.violated(b), s.violated()), tag + " violated"
            pd, po = dev.get_pi(b), s.get_pi()
            assert bits_equal(pd, po), f"{tag} pi first diff at {first_diff(pd, po)}"
            Pd, Po = dev.get_P(b), s.get_P()
            assert bits_equal(Pd, Po), f"{tag} P first diff at {first_diff(Pd, Po)}"
            assert st["rng"][b] == s.rng, tag + " rng"
            assert st["row_updates"][b] == s.row_updates, tag + " row_updates"
        kappa = kappa + 0.02
    form = dev.kernel_form
    dev.close()
    return form


@pytest.mark.parametrize("groups", ["1", "2", "4", "p2", "p4"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_sweeps_match_oracle(oracle, bxh, case, groups, monkeypatch):
    """Staged kernel, with 1, 2 and 4 cooperating workgroups (column owners)
    per instance: the result must not depend on how the columns are split."""
    monkeypatch.delenv("BXH_FORCE_V1", raising=False)
    monkeypatch.setenv("BXH_GROUPS", groups)
    _, maker, float_type, maximize, order, x0p = case
    run_case(oracle, bxh, maker(), float_type, maximize, order, x0p)


@pytest.mark.parametrize("case", CASES[::3], ids=[c[0] for c in CASES[::3]])
def test_sweeps_match_oracle_lane_per_column_kernel(oracle, bxh, case, monkeypatch):
    """The lane-per-column kernel stays as the path for instances whose
    longest column exceeds 128 elements: keep it pinned too."""
    monkeypatch.setenv("BXH_FORCE_V1", "1")
    _, maker, float_type, maximize, order, x0p = case
    run_case(oracle, bxh, maker(), float_type, maximize, order, x0p)


def test_scpnrh_shaped_rows_with_column_marks(oracle, bxh, monkeypatch):
    """The pipelined sweep classifies the rows of a consecutive list from static gaps; any other
    list, and every list with BXH_STATIC_LATE=0, goes by the column marks.  Full sweeps of the
    bench shape's rows through the marks (what rounds 1 - 2 ran everywhere)."""
    monkeypatch.setenv("BXH_GROUPS", "p4")
    monkeypatch.setenv("BXH_STATIC_LATE", "0")
    inst = synth.make_scp(120, 10000, 0.05, "int", 31)
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=2, sweeps=4, push_at=(2,), expect_form=44)
    inst = synth.make_scp(100, 6000, 0.08, "real", 33)
    run_case(oracle, bxh, inst, 1, False, 1, 0.0, batch=2, sweeps=3, push_at=(), expect_form=44)


@pytest.mark.parametrize("groups", ["1", "4", "p4"])
def test_scpnrh_shaped_rows(oracle, bxh, groups, monkeypatch):
    """Rows of ~500 elements and columns of ~50 (the bench shape, fewer rows):
    several passes per row, more than one piece per column."""
    monkeypatch.setenv("BXH_GROUPS", groups)
    inst = synth.make_scp(120, 10000, 0.05, "int", 31)
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=2, sweeps=4, push_at=(2,))
    inst = synth.make_scp(100, 6000, 0.08, "real", 32)
    run_case(oracle, bxh, inst, 0, False, 0, 0.0, batch=2, sweeps=4, push_at=(3,))


@pytest.mark.parametrize("groups", ["4", "p4"])
@pytest.mark.parametrize("float_type", [1, 0])
def test_long_pm1_rows_with_push_sweeps(oracle, bxh, groups, float_type, monkeypatch):
    """Rows of ~600 elements with -1 coefficients (inequalities-101), push sweeps among
    them: in the pipelined form that is the third pass of the chain's row classification
    (entries past 512), the second prefetched value of a thread, and -- push sweeps of the
    101 solvers amplify by the COLUMN's cost -- the cost of every entry, not only of the late
    ones, moved into LDS by LDS-DMA.  Rows 64 apart meet the same mark code again."""
    monkeypatch.setenv("BXH_GROUPS", groups)
    inst = synth.make_pm1(140, 2000, 0.3, "int", 61)
    assert int(np.diff(inst.row_ptr).max()) > 560
    run_case(oracle, bxh, inst, float_type, False, 0, 0.1, batch=2, sweeps=5, push_at=(1, 3),
             expect_form=44 if groups == "p4" else None)


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("groups", ["4", "p4"])
def test_consecutive_rows_starting_inside_the_matrix(oracle, bxh, order, groups, monkeypatch):
    """The pipelined sweep takes the late columns of a row from static gaps when the row list is
    a run of consecutive rows.  Here the run starts at row 12: the rows 0 .. 11
    all hold one extra column that is set in x0, so they start satisfied and rows 12 .. m-1 are
    the list -- the first rows of the walk have column neighbours in rows that are NOT walked
    (late by the static rule, with no previous row to patch from), in both directions."""
    monkeypatch.setenv("BXH_GROUPS", groups)
    base = synth.make_scp(90, 700, 0.12, "real", 71)
    rows = [list(base.col[base.row_ptr[k]:base.row_ptr[k + 1]]) for k in range(base.m)]
    extra = base.n  # a new column, first in the rows 0 .. 11
    for k in range(12):
        rows[k] = [extra] + rows[k]
    row_ptr = np.zeros(base.m + 1, np.int32)
    row_ptr[1:] = np.cumsum([len(r) for r in rows])
    col = np.concatenate([np.asarray(r, np.int32) for r in rows])
    inst = synth.Instance(m=base.m, n=base.n + 1, row_ptr=row_ptr, col=col,
                          factor=np.ones(col.shape[0], np.int32), cmin=base.cmin, cmax=base.cmax,
                          obj_var=np.arange(base.n + 1, dtype=np.int32),
                          obj_factor=np.concatenate([base.obj_factor, [3.25]]), maximize=False, kind=1,
                          name="scp-run-from-12")

    theta = 0.5
    orc = oracle.Solver(inst, float_type=1, sort_mode=1, cost_norm=4, seed=5)
    dev = bxh.DeviceBatch(inst, orc.cost(), 2, float_type=1)
    if groups == "p4":
        assert dev.kernel_form == 44
    delta = orc.compute_delta(theta)
    x0 = np.zeros(orc.nwords, np.uint64)
    x0[extra >> 6] |= np.uint64(1) << np.uint64(extra & 63)
    orcs = [orc, oracle.Solver(inst, float_type=1, sort_mode=1, cost_norm=4, seed=6)]
    for b, s in enumerate(orcs):
        s.set_x(x0)
        dev.set_x(b, x0)
        dev.set_rng(b, s.rng)
        s.order_init(order)
    dev.order_init(order)
    assert list(orcs[0].violated()) == list(range(12, inst.m))
    for it in range(4):
        st = dev.sweep(0.03 * it, delta, theta, order=order)
        for b, s in enumerate(orcs):
            r = s.sweep(0.03 * it, delta, theta)
            assert st["remaining"][b] == r
            assert np.array_equal(dev.get_x(b), s.get_x())
            assert bits_equal(dev.get_pi(b), s.get_pi()) and bits_equal(dev.get_P(b), s.get_P())
            assert st["rng"][b] == s.rng
    dev.close()


ORDER_CASES = [
    # (name, maker, float_type, maximize, order, x0_density, push_at)
    ("infeas-decr-pm1", lambda: synth.make_pm1(50, 250, 0.06, "int", 41), 1, False, 3, 0.3, (5, 9)),
    ("infeas-incr-pm1-eq", lambda: synth.make_pm1(40, 250, 0.06, "real", 42, equalities=True), 1, False, 4, 0.3, (5, 9)),
    ("infeas-decr-scp-f32", lambda: synth.make_scp(60, 400, 0.06, "int", 43), 0, False, 3, 0.0, (4,)),
    ("lagrangian-decr-scp", lambda: synth.make_scp(60, 400, 0.06, "real", 44), 1, False, 5, 0.0, (5, 9)),
    ("lagrangian-incr-spp-max", lambda: synth.make_spp(40, 300, 0.06, "int", 45), 1, True, 6, 0.2, (5, 9)),
    ("pi-sign-change", lambda: synth.make_scp(80, 60, 0.15, "real", 46), 1, False, 7, 0.1, (5, 9)),
    ("pi-sign-change-pm1", lambda: synth.make_pm1(90, 70, 0.12, "int", 47), 1, False, 7, 0.2, (3, 4)),
    ("cycle", lambda: synth.make_scp(80, 60, 0.15, "int", 48), 1, False, 8, 0.1, tuple(range(1, 20))),
    ("cycle-pm1-f32", lambda: synth.make_pm1(90, 70, 0.12, "real", 49), 0, False, 8, 0.2,
     tuple(range(0, 20, 2))),
]


@pytest.mark.parametrize("groups", ["1", "4", "p4"])
@pytest.mark.parametrize("case", ORDER_CASES, ids=[c[0] for c in ORDER_CASES])
def test_row_orders_match_oracle(oracle, bxh, case, groups, monkeypatch):
    """The sorting orders of compute_order (infeasibility, lagrangian),
    pi_sign_change and cycle (itm-common.hpp:682-880), including the
    value-initialised m_order a cycle run meets first."""
    monkeypatch.delenv("BXH_FORCE_V1", raising=False)
    monkeypatch.setenv("BXH_GROUPS", groups)
    _, maker, float_type, maximize, order, x0p, push_at = case
    run_case(oracle, bxh, maker(), float_type, maximize, order, x0p, sweeps=20, push_at=push_at)


def test_row_orders_lane_per_column_kernel(oracle, bxh, monkeypatch):
    monkeypatch.setenv("BXH_FORCE_V1", "1")
    for case in (ORDER_CASES[0], ORDER_CASES[3], ORDER_CASES[7]):
        _, maker, float_type, maximize, order, x0p, push_at = case
        run_case(oracle, bxh, maker(), float_type, maximize, order, x0p, sweeps=20, push_at=push_at)


def test_pi_sign_change_needs_n_le_m(bxh):
    inst = synth.make_scp(20, 50, 0.2, "int", 50)
    dev = bxh.DeviceBatch(inst, inst.dense_costs(), 1)
    with pytest.raises(RuntimeError):
        dev.order_create(7)
    dev.close()


Z_CASES = [
    # (name, maker, float_type, maximize, order, x0_density)
    ("zmix-int-max", lambda: synth.make_zmix(24, 60, "int", 6, maxlen=13, maximize=True), 1, True, 0, 0.0),
    ("zmix-real-min", lambda: synth.make_zmix(24, 60, "real", 7, maxlen=13, maximize=False), 1, False, 0, 0.2),
    ("zmix-int-f32", lambda: synth.make_zmix(30, 80, "int", 8, maxlen=12, maximize=True), 0, True, 2, 0.1),
    ("zmix-real-rev", lambda: synth.make_zmix(30, 50, "real", 9, maxlen=12, maximize=False), 1, False, 1, 0.5),
    ("zmix-infeas", lambda: synth.make_zmix(24, 60, "int", 10, maxlen=12, maximize=True), 1, True, 3, 0.3),
    ("z-knapsack", lambda: synth.make_z(20, 60, 0.2, "int", 11), 1, True, 0, 0.0),
    ("z-knapsack-real", lambda: synth.make_z(20, 60, 0.2, "real", 12), 1, True, 5, 0.3),
]


Z_WAVE_CASES = [
    # rows short enough for the wave-per-row form, few enough columns for neighbours to clash
    # and many enough for groups of two to four rows; integer costs: tied rows draw from the
    # engine in row order; the rows of kinds 0 / 1 and the long ones go the general way in between
    ("z-wave-int", lambda: synth.make_z(120, 900, 0.03, "int", 71), 1, False, 0, 0.0),
    ("z-wave-real-max", lambda: synth.make_z(100, 600, 0.04, "real", 72), 1, True, 1, 0.2),
    ("z-wave-f32-random-order", lambda: synth.make_z(150, 1200, 0.025, "int", 73), 0, False, 2, 0.1),
    ("z-wave-mixed-kinds", lambda: synth.make_zmix(80, 400, "int", 74, maxlen=14, maximize=True), 1, True, 0, 0.1),
    ("z-wave-mixed-kinds-min", lambda: synth.make_zmix(90, 500, "real", 75, maxlen=16, maximize=False), 1, False, 0, 0.3),
    ("z-wave-long-list", lambda: synth.make_z(400, 2500, 0.012, "int", 76), 1, False, 0, 0.0),
    # rows of 64 .. 127 elements (round 5: two elements per lane of the wave that takes the row) next
    # to short ones and to rows of 128 and more (the one-row path); integer costs tie, real ones do not
    ("z-wave-two-per-lane", lambda: synth.make_z(90, 3000, 0.03, "int", 81), 1, False, 0, 0.0),
    ("z-wave-two-per-lane-real", lambda: synth.make_z(80, 2400, 0.035, "real", 82), 1, True, 1, 0.2),
    ("z-wave-two-per-lane-f32", lambda: synth.make_z(70, 2600, 0.04, "int", 83), 0, False, 2, 0.1),
    ("z-wave-around-64", lambda: synth.make_z(120, 3200, 0.02, "int", 84), 1, True, 0, 0.3),
    ("z-wave-around-128", lambda: synth.make_z(60, 3000, 0.041, "real", 85), 1, False, 0, 0.1),
    # ... and with columns enough for such rows to share none: groups of long (and short) rows
    ("z-wave-two-per-lane-groups", lambda: synth.make_z(100, 30000, 0.003, "int", 86), 1, False, 0, 0.0),
    ("z-wave-two-per-lane-groups-real", lambda: synth.make_z(100, 24000, 0.0035, "real", 87), 1, True, 2, 0.2),
]


@pytest.mark.parametrize("kmax", [65535, 70000], ids=["a-16-bits", "a-beyond-16-bits"])
def test_z_large_coefficients(oracle, bxh, kmax, monkeypatch):
    """The wave form reads a slot's row and |coefficient| as one 32-bit word: coefficients of up
    to 65535 take it, larger ones the plain form (bxh_create decides) -- both equal to the oracle."""
    monkeypatch.delenv("BXH_ZWAVE", raising=False)
    monkeypatch.delenv("BXH_GROUPS", raising=False)
    inst = synth.make_z(100, 700, 0.035, "int", 77, kmax=kmax)
    assert int(inst.factor.max()) > 60000
    run_case(oracle, bxh, inst, 1, True, 0, 0.3, batch=2, sweeps=6, push_at=(3,),
             expect_form=6 if kmax <= 65535 else 3)


@pytest.mark.parametrize("zwave", ["wave-per-row", "plain"])
@pytest.mark.parametrize("case", Z_CASES + Z_WAVE_CASES, ids=[c[0] for c in Z_CASES + Z_WAVE_CASES])
def test_z_sweeps_match_oracle(oracle, bxh, case, zwave, monkeypatch):
    """inequalities-Z (general integer coefficients): linear scan, branch and
    bound, exhaustive patterns and their `unconstrained` variants, followed by
    local_affect's repair pass (itm-solver-inequalities-Z.cpp:295-564) -- in the
    kernel's two forms: a wave per row with independent neighbours of the list side
    by side (what a linear objective gets up to two workgroups per CU), and the
    plain one (BXH_ZWAVE=0: larger batches, quadratic objectives)."""
    monkeypatch.setenv("BXH_ZWAVE", "1" if zwave == "wave-per-row" else "0")
    _, maker, float_type, maximize, order, x0p = case
    form = run_case(oracle, bxh, maker(), float_type, maximize, order, x0p)
    assert form == (6 if zwave == "wave-per-row" else 3), form


def with_quadratic(inst, seed, terms):
    """Add `terms` random quadratic objective terms f * x_a * x_b (some
    squares) to an instance."""
    rs = np.random.RandomState(seed)
    a = rs.randint(0, inst.n, size=terms).astype(np.int32)
    b = rs.randint(0, inst.n, size=terms).astype(np.int32)
    b[::7] = a[::7]
    inst.q_a, inst.q_b = a, b
    inst.q_factor = np.round(rs.uniform(-20, 20, size=terms), 1)
    return inst


Q_CASES = [
    # (name, maker, float_type, maximize, order, x0_density)
    ("q-scp-int", lambda: with_quadratic(synth.make_scp(60, 400, 0.06, "int", 61), 1, 300), 1, False, 0, 0.1),
    ("q-scp-real-f32-max", lambda: with_quadratic(synth.make_scp(50, 300, 0.08, "real", 62), 2, 200), 0, True, 2, 0.3),
    ("q-spp", lambda: with_quadratic(synth.make_spp(40, 300, 0.06, "int", 63), 3, 250), 1, False, 0, 0.05),
    ("q-pm1-ineq", lambda: with_quadratic(synth.make_pm1(50, 250, 0.06, "real", 64), 4, 200), 1, False, 1, 0.2),
    ("q-pm1-eq", lambda: with_quadratic(synth.make_pm1(40, 250, 0.06, "int", 65, equalities=True), 5, 200), 1, True, 0, 0.2),
    ("q-zmix", lambda: with_quadratic(synth.make_zmix(24, 60, "int", 66, maxlen=12, maximize=True), 6, 80), 1, True, 0, 0.2),
    ("q-zmix-min", lambda: with_quadratic(synth.make_zmix(24, 60, "real", 67, maxlen=12, maximize=False), 7, 80), 1, False, 0, 0.4),
]


@pytest.mark.parametrize("case", Q_CASES, ids=[c[0] for c in Q_CASES])
def test_quadratic_objective_matches_oracle(oracle, bxh, case, monkeypatch):
    """quadratic_cost_type (itm-common.hpp:1150-1440): c(j, x) depends on the
    current x in the reduced costs, in the push amplification and, for the Z
    solver, in the repair pass (sequential there)."""
    monkeypatch.delenv("BXH_FORCE_V1", raising=False)
    _, maker, float_type, maximize, order, x0p = case
    run_case(oracle, bxh, maker(), float_type, maximize, order, x0p)


def test_quadratic_objective_lane_per_column_kernel(oracle, bxh, monkeypatch):
    monkeypatch.setenv("BXH_FORCE_V1", "1")
    for case in (Q_CASES[0], Q_CASES[3]):
        _, maker, float_type, maximize, order, x0p = case
        run_case(oracle, bxh, maker(), float_type, maximize, order, x0p)


def _custom(rows, n, cmin, cmax, costs, factor=None, kind=1, maximize=False, name="edge"):
    row_ptr = np.zeros(len(rows) + 1, np.int32)
    for k, r in enumerate(rows):
        row_ptr[k + 1] = row_ptr[k] + len(r)
    col = np.concatenate([np.asarray(r, np.int32) for r in rows])
    fac = np.ones(col.shape[0], np.int32) if factor is None else np.asarray(factor, np.int32)
    return synth.Instance(m=len(rows), n=n, row_ptr=row_ptr, col=col, factor=fac,
                          cmin=np.asarray(cmin, np.int32), cmax=np.asarray(cmax, np.int32),
                          obj_var=np.arange(n, dtype=np.int32), obj_factor=np.asarray(costs, np.float64),
                          maximize=maximize, kind=kind, name=name)


def edge_ragged():
    """Rows of 1, 2 and 3 elements next to a row over every variable; a
    variable no row mentions; a variable in every row."""
    rs = np.random.RandomState(70)
    n = 90
    rows = [[0], [0, 1], [0, 2, 3], list(range(0, 89)), [0, 5], [0] + list(range(10, 60)), [0, 7, 8, 9]]
    for _ in range(25):
        rows.append([0] + sorted(rs.choice(np.arange(1, 89), size=int(rs.randint(1, 12)), replace=False)))
    m = len(rows)
    return _custom(rows, n, [1] * m, [len(r) for r in rows], rs.randint(1, 30, size=n))


def edge_long_rows():
    """A row longer than twice the workgroup (the bitonic-sort path) and
    equality rows with an interior right-hand side (interior ranks)."""
    rs = np.random.RandomState(71)
    n = 1300
    rows = [list(range(n)), list(range(0, n, 2))]
    cmin, cmax = [3, 5], [3, 5]
    for _ in range(30):
        size = int(rs.randint(20, 200))
        rows.append(sorted(rs.choice(n, size=size, replace=False)))
        b = int(rs.randint(1, 6))
        cmin.append(b)
        cmax.append(b + int(rs.randint(0, 3)))
    return _custom(rows, n, cmin, cmax, 1 + 99 * rs.random_sample(n))


def edge_long_columns():
    """Columns with more than 128 entries (the staged kernel cannot take
    them: lane-per-column kernel) -- 300 rows that all share 5 variables."""
    rs = np.random.RandomState(72)
    n = 400
    rows = []
    for _ in range(300):
        rest = sorted(rs.choice(np.arange(5, n), size=int(rs.randint(3, 15)), replace=False))
        rows.append([0, 1, 2, 3, 4] + rest)
    m = len(rows)
    return _custom(rows, n, [2] * m, [6] * m, rs.randint(1, 50, size=n))


EDGE_CASES = [
    ("ragged", edge_ragged, 1, False, 0, 0.0),
    ("ragged-f32-rand", edge_ragged, 0, False, 2, 0.3),
    ("long-rows", edge_long_rows, 1, False, 0, 0.0),
    ("long-rows-max", edge_long_rows, 1, True, 1, 0.5),
    ("long-columns", edge_long_columns, 1, False, 0, 0.1),
]


@pytest.mark.parametrize("groups", ["1", "4", "p4"])
@pytest.mark.parametrize("case", EDGE_CASES, ids=[c[0] for c in EDGE_CASES])
def test_edge_shapes_match_oracle(oracle, bxh, case, groups, monkeypatch):
    """Ragged rows (down to one element), a variable in no row and one in every
    row, rows longer than the workgroup, interior ranks, columns too long for
    the staged kernel."""
    monkeypatch.delenv("BXH_FORCE_V1", raising=False)
    monkeypatch.setenv("BXH_GROUPS", groups)
    _, maker, float_type, maximize, order, x0p = case
    run_case(oracle, bxh, maker(), float_type, maximize, order, x0p, sweeps=8, push_at=(3, 6))


@pytest.mark.parametrize("lookahead", ["2", "3"])
def test_pipelined_sweep_mark_wrap_and_lookahead(oracle, bxh, lookahead, monkeypatch):
    """Pipelined sweep (row chain + producers): more than 255 rows per sweep, so
    that the one-byte column marks come round (a column last seen 255 rows
    earlier is then summed by the chain although no recent row holds it:
    both sides must take the same decision), with both look-ahead depths; a
    wide instance where most columns are early and a narrow one where nearly
    every column is late."""
    monkeypatch.delenv("BXH_FORCE_V1", raising=False)
    monkeypatch.setenv("BXH_GROUPS", "p4")
    monkeypatch.setenv("BXH_LOOKAHEAD", lookahead)
    inst = synth.make_scp(600, 3000, 0.01, "int", 91)
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=2, sweeps=5, push_at=(3,), expect_form=44)
    inst = synth.make_scp(520, 60, 0.3, "real", 92)
    run_case(oracle, bxh, inst, 1, False, 1, 0.2, batch=2, sweeps=5, push_at=(2,))
    inst = synth.make_pm1(300, 900, 0.02, "int", 93)
    run_case(oracle, bxh, inst, 0, True, 2, 0.1, batch=2, sweeps=5, push_at=(4,))


def test_invalid_inputs_are_refused(bxh):
    """bxh_create validates what the Solver constructors assume."""
    inst = synth.make_scp(10, 30, 0.2, "int", 73)
    bad = synth.make_scp(10, 30, 0.2, "int", 73)
    bad.col = bad.col.copy()
    bad.col[0] = 99  # column out of range
    with pytest.raises(RuntimeError):
        bxh.DeviceBatch(bad, bad.dense_costs(), 1)
    bad = synth.make_scp(10, 30, 0.2, "int", 73)
    bad.factor = bad.factor.copy()
    bad.factor[0] = 3  # not a 0/1 coefficient for the 01 solver
    with pytest.raises(RuntimeError):
        bxh.DeviceBatch(bad, bad.dense_costs(), 1)
    bad = synth.make_scp(10, 30, 0.2, "int", 73)
    bad.cmin = bad.cmin.copy()
    bad.cmin[0] = 1000  # min > max after clamping
    bad.cmax = bad.cmax.copy()
    bad.cmax[0] = 2
    with pytest.raises(RuntimeError):
        bxh.DeviceBatch(bad, bad.dense_costs(), 1)
    dev = bxh.DeviceBatch(inst, inst.dense_costs(), 2)
    with pytest.raises(RuntimeError):
        dev.sweep(-0.5, 0.01, 0.5)  # negative kappa
    with pytest.raises(RuntimeError):
        dev.order_create(42)
    dev.close()


def test_rejected_draw_in_a_tie_shuffle(oracle, bxh):
    """uniform_int_distribution rejects a raw draw now and then; every later
    draw of the row then shifts.  Engine state 898 meets a rejection while
    shuffling a run of 300 equal reduced costs (the all-lanes shuffle gives up
    and thread 0 redoes it), 899 does not."""
    rs = np.random.RandomState(80)
    n = 300
    rows = [list(range(n))] + [sorted(rs.choice(n, size=40, replace=False)) for _ in range(20)]
    m = len(rows)
    inst = _custom(rows, n, [1] * m, [len(r) for r in rows], np.full(n, 7.0))
    for seed in (898, 899):
        s = oracle.Solver(inst, float_type=1, sort_mode=1, cost_norm=4, seed=1)
        dev = bxh.DeviceBatch(inst, s.cost(), 1)
        x0 = np.zeros(s.nwords, np.uint64)
        s.set_x(x0)
        dev.set_x(0, x0)
        s.rng = seed
        dev.set_rng(0, seed)
        s.order_init(0)
        dev.order_init(0)
        d = s.compute_delta(0.5)
        for it in range(3):
            st = dev.sweep(0.1, d, 0.5)
            rem = s.sweep(0.1, d, 0.5)
            assert st["remaining"][0] == rem
            assert st["rng"][0] == s.rng
            assert np.array_equal(dev.get_x(0), s.get_x())
            assert bits_equal(dev.get_P(0), s.get_P())
            assert bits_equal(dev.get_pi(0), s.get_pi())
        dev.close()


def _load_vectors():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("vec", _load_vectors(), ids=lambda v: v["name"])
def test_device_reproduces_committed_vectors(oracle, bxh, vec):
    """The HIP path against the committed digests (tests/golden/vectors.json):
    x, pi, P, engine state and remaining after every sweep."""
    import hashlib

    def digest(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]

    inst = getattr(synth, vec["generator"])(*vec["args"])
    s = oracle.Solver(inst, float_type=vec["float_type"], sort_mode=1, cost_norm=4,
                      seed=vec["seed"], maximize=vec["maximize"])
    dev = bxh.DeviceBatch(inst, s.cost(), 1, float_type=vec["float_type"], maximize=vec["maximize"])
    dev.set_x(0, np.zeros(s.nwords, np.uint64))
    dev.set_rng(0, s.rng)
    dev.order_init(vec["order"])
    delta = s.compute_delta(vec["theta"])
    kappa = 0.0
    for it, exp in enumerate(vec["sweeps"]):
        push = it in vec["pushes"]
        st = dev.sweep(kappa, delta, vec["theta"], is_push=push,
                       obj_amp=vec["obj_amp"] if push else 0.0, order=vec["order"])
        got = dict(remaining=int(st["remaining"][0]), rng=int(st["rng"][0]), x=digest(dev.get_x(0)),
                   pi=digest(dev.get_pi(0)), P=digest(dev.get_P(0)))
        assert got == exp, f"sweep {it}"
        kappa += vec["kappa_step"]
    dev.close()
