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


@pytest.mark.parametrize("groups", ["w1", "w2", "w3", "w2-sorted"])
@pytest.mark.parametrize("case", WIDE_CASES, ids=[c[0] for c in WIDE_CASES])
def test_wide_kernel_small_cases(oracle, bxh, case, groups, monkeypatch):
    monkeypatch.delenv("BXH_FORCE_V1", raising=False)
    _tied(monkeypatch, "sorted" if groups.endswith("-sorted") else "fast")
    groups = groups.split("-")[0]
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


TIED = pytest.mark.parametrize("tied", ["fast", "sorted"])


def _tied(monkeypatch, tied):
    """"sorted": every row with equal reduced costs goes through the sort path (BXH_WIDE_TIEDFAST=0);
    "fast": the default -- the tied keys and their counts out of the tie table, no sorted order, where
    the selection allows it."""
    if tied == "sorted":
        monkeypatch.setenv("BXH_WIDE_TIEDFAST", "0")
    else:
        monkeypatch.delenv("BXH_WIDE_TIEDFAST", raising=False)


@TIED
@pytest.mark.parametrize("groups,cols,own", [("w4", 30000, False), ("w3", 30000, False), ("w2", 70000, False),
                                             ("w4", 30000, True)],
                         ids=["w4-chunk-each", "w3-one-idle", "w2-two-chunks-each", "w4-own-sort"])
def test_wide_kernel_long_tied_rows(oracle, bxh, groups, cols, own, tied, monkeypatch):
    """Rows of ~3000 (7000) elements with integer costs: every row of the first
    sweep holds runs of equal reduced costs (the long-row sort and its shuffles).
    The workgroups of an instance sort the row together -- a chunk of 1024 pairs
    each (4 workgroups), 2 chunks of 2048 for 3 workgroups (one has none), 4 chunks
    of 2048 for 2 workgroups -- or (BXH_WIDE_OWN_SORT) every one the whole row."""
    monkeypatch.setenv("BXH_GROUPS", groups)
    _tied(monkeypatch, tied)
    if own:
        monkeypatch.setenv("BXH_WIDE_OWN_SORT", "1")
    inst = synth.make_scp(40, cols, 0.1, "int", 95)
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=2, sweeps=3, push_at=(2,),
             expect_form=100 + int(groups[1:]))


@TIED
@pytest.mark.parametrize("ncost", [2, 6, 40], ids=["runs-of-1500", "runs-of-500", "runs-of-75"])
def test_wide_kernel_long_rows_tied_runs_side_by_side(oracle, bxh, ncost, tied, monkeypatch):
    """40 rows of 3000 elements whose integer costs take `ncost` values, among
    4200 rows (so that the pi window -- the LDS scratch of the long-row path -- is
    64 KB as at the XL shape): the tied runs of a row are shuffled side by side
    from jumps of the engine, a wave per run (<= 512), all threads (<= 2048) or
    one thread (longer), and with runs this long some draw is rejected in almost
    every row (the scheme restarts behind that run)."""
    monkeypatch.setenv("BXH_GROUPS", "w4")
    _tied(monkeypatch, tied)
    rs = np.random.RandomState(400 + ncost)
    n, nlong, m = 12000, 40, 4200
    rows = [np.sort(rs.choice(n, size=3000, replace=False)) for _ in range(nlong)]
    rows += [np.array([rs.randint(n)]) for _ in range(m - nlong)]
    costs = rs.randint(1, ncost + 1, n).astype(float)
    inst = _custom(rows, n, [1] * m, [2 ** 30] * m, costs, kind=1)
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=2, sweeps=3, push_at=(2,), expect_form=104)


@pytest.mark.parametrize("ncost", [2500, 14000], ids=["1700-tied-keys", "around-1000-tied-keys"])
def test_wide_kernel_many_tied_keys(oracle, bxh, ncost, monkeypatch):
    """Rows of 6000 elements over `ncost` integer cost values: ~1700 tied keys a row (more than the 1024 the
    path without a sort takes: the row is sorted) and ~980 (rows on either side of that limit)."""
    monkeypatch.setenv("BXH_GROUPS", "w4")
    rs = np.random.RandomState(77 + ncost)
    n, nlong, m = 16000, 30, 4200
    rows = [np.sort(rs.choice(n, size=6000, replace=False)) for _ in range(nlong)]
    rows += [np.array([rs.randint(n)]) for _ in range(m - nlong)]
    costs = rs.randint(1, ncost + 1, n).astype(float)
    inst = _custom(rows, n, [1] * m, [2 ** 30] * m, costs, kind=1)
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=2, sweeps=3, push_at=(2,), expect_form=104)


@TIED
def test_wide_kernel_rejected_draw_among_runs_of_a_lane_each(oracle, bxh, tied, monkeypatch):
    """60 rows of 20 000 elements (columns of their own) over 48 cost values: ~48 runs of ~420 equal
    reduced costs a row, none beyond 512 -- every run is shuffled by a lane of its own, and whether a draw
    of the row is rejected (which shifts the engine for every run behind it) is decided from one comparison
    per draw against the widest range, the exact test only for the draws that pass it.  A run of 420
    elements sees a rejected draw with probability ~0.3 %, a row with ~13 %: over 60 rows and three sweeps
    one is certain for every practical purpose (and checked once by breaking the exact test on purpose:
    the case fails then)."""
    monkeypatch.setenv("BXH_GROUPS", "w16")
    _tied(monkeypatch, tied)
    rs = np.random.RandomState(4242)
    nlong, rlen, m = 60, 20000, 4200
    n = nlong * rlen
    rows = [np.arange(i * rlen, (i + 1) * rlen) for i in range(nlong)]
    rows += [np.array([rs.randint(n)]) for _ in range(m - nlong)]
    costs = rs.randint(1, 49, n).astype(float)
    inst = _custom(rows, n, [1] * m, [2 ** 30] * m, costs, kind=1)
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=2, sweeps=3, push_at=(2,), expect_form=116)


@pytest.mark.parametrize("scan", ["band", "gather"])
def test_wide_violated_scan_over_several_bands(oracle, bxh, scan, monkeypatch):
    """compute_violated_constraints of the wide layout for all instances at once: the band scan (x of a
    band of 61 440 columns in LDS, a row's stretch inside the band found by two searches; 01 matrices with
    ascending columns, up to 16 instances) over 150 000 columns = three bands, a row in one, two or all of
    them, against the scan that gathers x out of L2 (BXH_WIDE_BAND_SCAN=0) -- both against the oracle's
    violated lists after every sweep."""
    monkeypatch.setenv("BXH_GROUPS", "w2")
    if scan == "gather":
        monkeypatch.setenv("BXH_WIDE_BAND_SCAN", "0")
    else:
        monkeypatch.delenv("BXH_WIDE_BAND_SCAN", raising=False)
    rs = np.random.RandomState(5150)
    n, m = 150000, 700
    rows = []
    for k in range(m):
        kind = k % 4
        if kind == 0:    # anywhere
            cols = rs.choice(n, size=rs.randint(1, 60), replace=False)
        elif kind == 1:  # inside one band
            b0 = rs.randint(3) * 61440
            cols = b0 + rs.choice(min(61440, n - b0), size=rs.randint(1, 40), replace=False)
        elif kind == 2:  # around a band border
            cols = 61440 * (1 + rs.randint(2)) + rs.choice(np.arange(-30, 30), size=rs.randint(2, 30), replace=False)
        else:            # the ends of the column range
            cols = np.concatenate([rs.choice(50, size=3, replace=False), n - 1 - rs.choice(50, size=3, replace=False)])
        rows.append(np.sort(cols))
    costs = (1 + 99 * rs.rand(n)).astype(float)
    inst = _custom(rows, n, [1] * m, [2 ** 30] * m, costs, kind=1)
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=5, sweeps=3, push_at=(2,), expect_form=102)


def test_wide_kernel_bench_geometry(oracle, bxh, monkeypatch):
    """16 instances x 16 workgroups = one workgroup per CU, the launch geometry of
    `bench.py --config c4` (and the 16-instance form of the shared violated scan),
    on an instance the oracle replays in seconds: every instance of the batch is
    compared after every sweep."""
    monkeypatch.setenv("BXH_GROUPS", "w16")
    inst = synth.make_scp(3000, 2500, 0.08, "real", 131)
    run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=16, sweeps=2, push_at=(1,), expect_form=116)


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


DEVICE_BUILD_CASES = [
    ("scp-int", lambda: synth.make_scp(300, 900, 0.05, "int", 101), 1, False, 0),
    ("scp-real-f32-rev", lambda: synth.make_scp(200, 700, 0.08, "real", 102), 0, False, 1),
    ("spp", lambda: synth.make_spp(120, 600, 0.05, "int", 103), 1, False, 0),
    ("pm1-ineq-max", lambda: synth.make_pm1(150, 500, 0.06, "real", 104), 1, True, 2),
    ("pm1-eq", lambda: synth.make_pm1(100, 400, 0.06, "int", 105, equalities=True), 1, False, 0),
    ("long-columns", lambda: synth.make_scp(5000, 300, 0.1, "real", 106), 1, False, 0),
]


@pytest.mark.parametrize("groups", ["w1", "w3"])
@pytest.mark.parametrize("case", DEVICE_BUILD_CASES, ids=[c[0] for c in DEVICE_BUILD_CASES])
def test_device_built_matrix_matches_oracle(oracle, bxh, case, groups, monkeypatch):
    """bxh_create_device: the CSR is handed over in device memory and the
    column view / bounds are built by kernels (the only way in for the 2e9
    elements of the XL shape) -- same results as the oracle, whose constructor
    builds them on the host."""
    import torch

    from helpers import bits_equal, first_diff, random_x

    monkeypatch.setenv("BXH_GROUPS", groups)
    _, maker, float_type, maximize, order = case
    inst = maker()
    batch = 2
    orcs = [oracle.Solver(inst, float_type=float_type, sort_mode=1, cost_norm=4, seed=700 + b,
                          maximize=maximize) for b in range(batch)]
    dev_t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in dict(
        row_ptr=inst.row_ptr.astype(np.int32), col=inst.col.astype(np.int32),
        factor=inst.factor.astype(np.int32), cmin=inst.cmin.astype(np.int32),
        cmax=inst.cmax.astype(np.int32), cost=orcs[0].cost()).items()}
    dev = bxh.DeviceBatch.on_device(
        inst.m, inst.n, inst.nnz, dev_t["row_ptr"].data_ptr(), dev_t["col"].data_ptr(),
        dev_t["factor"].data_ptr(), dev_t["cmin"].data_ptr(), dev_t["cmax"].data_ptr(),
        dev_t["cost"].data_ptr(), inst.kind, maximize, batch, float_type=float_type, keep=dev_t)
    del dev_t  # the handle owns copies
    torch.cuda.empty_cache()
    assert dev.kernel_form == 100 + int(groups[1:]) and dev.rmax == orcs[0].rmax
    delta = orcs[0].compute_delta(0.5)
    # bxh_row_bytes of a device-built handle == the host formula
    F = 8 if float_type else 4
    clen = np.bincount(inst.col, minlength=inst.n)
    r = np.diff(inst.row_ptr)
    want = (np.add.reduceat(clen[inst.col], inst.row_ptr[:-1]) * (4 + 2 * F) + r * (4 + 5 * F)
            + 2 * F + ((r + 7) // 8) * 2)
    assert np.array_equal(dev.row_bytes(), want.astype(np.float64))
    for b, s in enumerate(orcs):
        x0 = random_x(inst.n, 0.05, 31 + b)
        s.set_x(x0)
        dev.set_x(b, x0)
        dev.set_rng(b, s.rng)
        s.order_init(order)
    rem = dev.order_init(order)
    for b, s in enumerate(orcs):
        assert rem[b] == len(s.violated())
        assert np.array_equal(dev.violated(b), s.violated())
    # P travels both ways through the device-side permutation
    Pset = np.random.RandomState(5).standard_normal(inst.nnz)
    if not float_type:
        Pset = Pset.astype(np.float32).astype(np.float64)
    for b, s in enumerate(orcs):
        s.set_P(Pset * (b + 1))
        dev.set_P(b, Pset * (b + 1))
        assert bits_equal(dev.get_P(b), s.get_P())
    kappa = np.array([0.02, 0.05])
    for it in range(6):
        push = it == 3
        st = dev.sweep(kappa, delta, 0.5, is_push=push, obj_amp=5.0 if push else 0.0, order=order)
        for b, s in enumerate(orcs):
            rr = s.sweep(float(kappa[b]), delta, 0.5, is_push=push, obj_amp=5.0 if push else 0.0)
            tag = f"sweep {it} inst {b}"
            same_x = np.array_equal(dev.get_x(b), s.get_x())
            assert st["remaining"][b] == rr, (tag, "x equal" if same_x else "x differs",
                                             list(dev.violated(b)), list(s.violated()))
            assert same_x, tag + " x"
            assert np.array_equal(dev.violated(b), s.violated()), tag + " violated"
            pd, po = dev.get_pi(b), s.get_pi()
            assert bits_equal(pd, po), f"{tag} pi first diff at {first_diff(pd, po)}"
            Pd, Po = dev.get_P(b), s.get_P()
            assert bits_equal(Pd, Po), f"{tag} P first diff at {first_diff(Pd, Po)}"
            assert st["rng"][b] == s.rng, tag + " rng"
        kappa = kappa + 0.03
    dev.close()
