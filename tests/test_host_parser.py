"""CPU: the C++ host library's LP reader against what the reference's own io
tests expect of the fixture files (lib/test/io.cpp:661-726) and of a few
inline problems of lib/test/solve.cpp."""
import os

import pytest

from baryonyx_amd import host_ffi as H

LP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lp")
MIN, MAX = 0, 1


def lp(name):
    return os.path.join(LP, name)


def test_geom_30a():
    i = H.parse(lp("geom-30a-3-ext_1000_support.lp"))
    assert i.status == 0 and i.maximize == MIN
    assert i.variables == 819 and i.n_binary == 90


def test_general():
    i = H.parse(lp("general.lp"))
    assert i.status == 0 and i.maximize == MIN and i.variables == 3 and i.n_binary == 0


def test_sudoku_and_queens():
    assert H.parse(lp("sudoku.lp")).variables == 81
    q = H.parse(lp("8_queens_puzzle.lp"))
    assert q.variables == 64 and q.n_binary == 64


@pytest.mark.parametrize("name,ptype", [
    ("assignment_problem_1.lp", 0), ("assignment_problem_2.lp", 0), ("assignment_problem_3.lp", 0),
    ("8_queens_puzzle.lp", 3), ("prepro.lp", 4),
    ("negative-coeff.lp", 1), ("negative-coeff2.lp", 1), ("negative-coeff3.lp", 1),
    ("small4.lp", 1), ("geom-30a-3-ext_1000_support.lp", 1),
    ("flat30-7.lp", 4), ("uf50-0448.lp", 4), ("aim-50-1_6-yes1-2.lp", 4), ("negative-coeff4.lp", 4),
])
def test_solver_family_per_fixture(name, ptype):
    """which itm-solver-* each fixture is routed to
    (problem_solver_type: 0 eq01, 1 eq101, 3 ineq01, 4 ineq101)."""
    assert H.parse(lp(name)).problem_type == ptype


def test_inline_problems(tmp_path):
    p = tmp_path / "real_cost.lp"  # lib/test/solve.cpp:258-271
    p.write_text("minimize\n- 0.1 a - 0.5 b - 0.9 c - 1e-7 d\nSubject to:\n-a -b -c <= -1\n"
                 "-a -b -c >= -3\n-a -c >= -2\n-a -c <= -1\na + c >= 1\n+ b + c +d >= 2\n"
                 "Binaries\na b c d\nEnd\n")
    i = H.parse(str(p))
    assert i.status == 0 and i.variables == 4 and i.less == 2 and i.greater == 4 and i.n_binary == 4
    q = tmp_path / "quad.lp"  # lib/test/solve.cpp:236-246
    q.write_text("minimize\n-5a + [ 2a * b + 3c * d ] /2 + 10b\nSubject to:\na + b >= 0\n"
                 "b + c + d >= 0\nBinaries\na b c d\nEnd\n")
    i = H.parse(str(q))
    assert i.status == 0 and i.n_quadratic == 2 and i.n_objective == 2
    bad = tmp_path / "bad.lp"
    bad.write_text("minimise\n x\nst\n x >= 1\nend\n")
    assert H.parse(str(bad)).status == 5  # bad_objective_function_type
    assert H.parse(str(tmp_path / "missing.lp")).status == 1  # file_not_found


def test_merged_constraints_merge_equal_rows(tmp_path):
    """make_merged_constraints: <= and >= on the same left-hand side become one
    row (lib/src/merged-constraint.cpp:63-127)."""
    p = tmp_path / "m.lp"
    p.write_text("minimize\n a + b + c\nst\n a + b + c >= 1\n a + b + c <= 2\n a + b = 1\n"
                 "binary\n a b c\nend\n")
    inst = H.merged(str(p))
    assert inst.m == 2 and inst.n == 3
    rows = {(int(inst.cmin[k]), int(inst.cmax[k])) for k in range(inst.m)}
    assert rows == {(1, 2), (1, 1)}


def test_scp_file_is_the_same_problem_as_its_lp_file(tmp_path):
    """OR-Library's set-covering format read directly: the
    merged constraints and the objective equal those of the same instance
    written as a CPLEX-LP file."""
    import numpy as np
    from baryonyx_amd import synth
    inst = synth.make_scp(60, 400, 0.06, "int", 12)
    lp, scp = str(tmp_path / "a.lp"), str(tmp_path / "a.scp")
    synth.write_lp(inst, lp)
    synth.write_scp(inst, scp)
    a, b = H.merged(lp, preprocessor=0), H.merged(scp, preprocessor=0)
    assert (a.m, a.n, a.kind, a.maximize) == (b.m, b.n, b.kind, b.maximize) == (60, 400, 1, False)
    for f in ("row_ptr", "col", "factor", "cmin", "cmax"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert np.array_equal(a.dense_costs(), b.dense_costs()) and np.array_equal(a.dense_costs(), inst.dense_costs())
    info = H.parse(scp)
    assert (info.status, info.variables, info.greater, info.n_binary) == (0, 400, 60, 400)


def test_long_rows_parse_in_linear_time(tmp_path):
    """The reference merges a variable named twice in a function by scanning the
    elements read so far (lib/src/parser.cpp:1133-1141, 1159-1170): quadratic in
    the row length.  Rows of 20 000 entries and an objective of 100 000 terms --
    the XL set-covering shape at a tenth of its width -- must parse in seconds."""
    import time
    import numpy as np
    from baryonyx_amd import synth
    inst = synth.make_scp(40, 100000, 0.2, "int", 3)
    path = str(tmp_path / "wide.lp")
    synth.write_lp(inst, path)
    t0 = time.perf_counter()
    info = H.parse(path)
    dt = time.perf_counter() - t0
    assert info.status == 0 and info.variables == 100000 and info.nnz == inst.nnz
    assert dt < 20.0, dt  # (the quadratic scan: 40 rows x 2e4^2 / 2 + 1e5^2 / 2 compares ~ 1.3e10)


def test_scp_row_listing_a_column_twice_keeps_one_element(tmp_path):
    """A row of an OR-Library file is the set of the columns that cover it: an index
    listed twice stays one element with factor 1 (summed like an LP row it would be a
    factor 2 and move the instance to the inequalities-Z solver)."""
    import numpy as np
    p = tmp_path / "dup.scp"
    p.write_text("2 4\n 3 1 4 1\n 3 1 2 1\n 4 4 3 4 2\n")
    inst = H.merged(str(p), preprocessor=0)
    assert (inst.m, inst.n, inst.kind) == (2, 4, 1)
    assert list(inst.row_ptr) == [0, 2, 5]
    assert list(inst.col) == [0, 1, 3, 2, 1] or sorted(inst.col[:2]) == [0, 1]
    assert np.all(inst.factor == 1)


def test_cost_ratio_row_orders_p1_p2(tmp_path):
    """pre_constraint_order p1 .. p4 (merged-constraint.cpp:211-284): variable cost / degree,
    a row ranks by its cheapest variable (p1: dearest first, p2 - p4: cheapest first), and the
    elements of every row go by falling cost / degree.  Worked out by hand: degrees a 2, b 2,
    c 1, d 1 -> cost / degree a 1, b 3, c 8, d 2; row minima r0 {a, b} 1, r1 {b, c} 3,
    r2 {a, d} 1.  p1 ranks [r1, r0, r2], p2 ranks [r0, r2, r1] (three items: libstdc++ sorts
    them by insertion, ties keep their order).  The reference then writes the rank list into
    the rows' ids IN FILE ORDER (`csts[i].id = ratio[i].first`) and sorts the rows by id:
    p1 -> ids (1, 0, 2) -> rows r1, r0, r2; p2 -> ids (0, 2, 1) -> rows r0, r2, r1."""
    p = tmp_path / "o.lp"
    p.write_text("minimize\n 2 a + 6 b + 8 c + 2 d\nst\n r0: a + b >= 1\n r1: b + c >= 1\n r2: a + d >= 1\n"
                 "binary\n a b c d\nend\n")
    a, b, c, d = 0, 1, 2, 3
    r0, r1, r2 = (b, a), (c, b), (d, a)  # elements by falling cost / degree
    for order, expect in ((8, [r1, r0, r2]), (9, [r0, r2, r1]), (10, [r0, r2, r1]), (11, [r0, r2, r1])):
        inst = H.merged(str(p), pre_order=order, preprocessor=0)
        assert inst.m == 3 and inst.n == 4
        rows = [tuple(int(j) for j in inst.col[inst.row_ptr[k]:inst.row_ptr[k + 1]]) for k in range(3)]
        assert rows == expect, (order, rows)


def test_parameter_validation_matches_the_restatement():
    """context_set_solver_parameters of the host library (lib/src/private.cpp:50-172
    restated in host/api.cpp) against the tests' own restatement (helpers.validated_params,
    what the oracle-side reference tests run with): same fields for a spread of raw values,
    incl. the reference tests' out-of-range kappa triples, `limit = -1` and fractional `w`."""
    import ctypes as C
    import itertools
    import math

    from helpers import validated_params
    L = H.lib()
    L.bxc_validated_params.argtypes = [C.POINTER(H.Params), C.POINTER(H.Params)]
    grid = dict(
        limit=[-1, 0, 50, 1000],
        w=[-1.0, 0.05, 0.5, 20.0, 5000.0],
        kappa=[(0.0, 1e-3, 0.6), (0.3, 1e-2, 100.0), (0.1, 1e-17, 1.0), (0.5, 0.1, 0.4), (-0.1, 0.1, 0.5),
               (0.0, 1.0, 0.6)],
        theta=[0.5, 1.5], delta=[-1.0, 0.2, -0.5], alpha=[1.0, -2.0],
    )
    fields = ("theta delta kappa_min kappa_step kappa_max alpha w limit pushes_limit "
              "pushing_iteration_limit init_population_size").split()
    count = 0
    for limit, w, (kmin, kstep, kmax), theta, delta, alpha in itertools.product(*grid.values()):
        raw = dict(limit=limit, w=w, kappa_min=kmin, kappa_step=kstep, kappa_max=kmax, theta=theta,
                   delta=delta, alpha=alpha, pushes_limit=-3 if count % 2 else 7,
                   pushing_iteration_limit=0 if count % 3 else 11, init_population_size=3 if count % 5 else 40)
        want = validated_params(**raw)
        got = H.Params()
        L.bxc_validated_params(C.byref(H.default_params(**raw)), C.byref(got))
        for f in fields:
            a, b = getattr(got, f), want[f]
            if f == "limit":
                b = min(b, 2 ** 63 - 1)
            assert a == b or (isinstance(b, float) and math.isinf(b) and math.isinf(a)), (raw, f, a, b)
        count += 1
    assert count == 4 * 5 * 6 * 2 * 3 * 2


INT_MIN, INT_MAX = -2 ** 31, 2 ** 31 - 1
BAD_BOUND = 8  # file_format_error_tag::bad_bound


def _bounds_file(tmp_path, lines, tail="end\n"):
    p = tmp_path / "b.lp"
    p.write_text("minimize\nx + y + z + e1\nst\nx + y + z + e1 >= 1\nbounds\n" + lines + "\n" + tail)
    return str(p)


@pytest.mark.parametrize("lines,expected", [
    # name op v -> [0, v]: the operator is read and ignored (parser.cpp:944-956)
    ("x <= 40", {"x": (0, 40)}),
    ("x >= 3", {"x": (0, 3)}),
    ("x = 1", {"x": (0, 1)}),                      # capmo1_direct.lp:13726 `t = 1`
    ("x = 0", {"x": (0, 0)}),                      # prepro.lp:22 `t = 0`
    ("x == 7\ny =< 5\nz => 6", {"x": (0, 7), "y": (0, 5), "z": (0, 6)}),
    ("x < 4\ny > 4", {"x": (0, 4), "y": (0, 4)}),
    ("x <= -2", {"x": (0, -2)}),                   # no order check on this form
    ("x <= + 8\ny <= - 8", {"x": (0, 8), "y": (0, -8)}),
    ("x <= 2.7\ny <= -2.7\nz <= 1e3", {"x": (0, 2), "y": (0, -2), "z": (0, 1000)}),  # static_cast<int>
    ("x <= inf\ny <= infinity\nz <= -inf", {"x": (0, INT_MAX), "y": (0, INT_MAX), "z": (0, INT_MIN)}),
    # l op name op r -> [l, r] (:910-942)
    ("2 <= x <= 9", {"x": (2, 9)}),
    ("2 >= x >= 9", {"x": (2, 9)}),
    ("2 = x = 9", {"x": (2, 9)}),
    ("- 3 <= x <= - 1\n+ 1 <= y <= + 2", {"x": (-3, -1), "y": (1, 2)}),
    ("-inf <= x <= 5\n-5 <= y <= inf\n- infinity <= z <= infinity",
     {"x": (INT_MIN, 5), "y": (-5, INT_MAX), "z": (INT_MIN, INT_MAX)}),
    ("1.9 <= x <= 9.9", {"x": (1, 9)}),
    # a bare name -> free (:958-961)
    ("x\ny <= 3", {"x": (INT_MIN, INT_MAX), "y": (0, 3)}),
    # l op name -> [l, +inf] and the NEXT token goes with it (:924-928 returns read + 1)
    ("3 <= x\ny\nz <= 2", {"x": (3, INT_MAX), "y": (0, INT_MAX), "z": (0, 2)}),
    # the last line for a variable wins; an empty section changes nothing
    ("x <= 4\nx <= 6", {"x": (0, 6)}),
    ("", {}),
])
def test_bound_forms_follow_the_reference(tmp_path, lines, expected):
    """lib/src/parser.cpp:907-965 (read_bound) with :540-561 (set_bound_variable)."""
    path = _bounds_file(tmp_path, lines)
    assert H.parse(path).status == 0
    got = dict(zip(["x", "y", "z", "e1"], H.variables(path)))
    for name in ("x", "y", "z", "e1"):
        lo, hi = expected.get(name, (0, INT_MAX))
        assert got[name] == (lo, hi, 0), (name, got[name])


@pytest.mark.parametrize("lines,tail", [
    ("9 <= x <= 2", "end\n"),          # l > r (:936-937)
    ("3 <= x", "end\n"),               # the left-only form takes `end` with it; the file's end is no keyword
    ("x free", "end\n"),               # `free` is not a keyword of the reference: an unknown variable
    ("e1 <= 4", "end\n"),              # first character e: taken for a number (:118-129), sscanf fails
    ("E1 <= 4", "end\n"),
    ("w <= 4", "end\n"),               # unknown variable (:542-544)
    ("x <= y", "end\n"),               # no number behind the operator
    ("x <=", "end\n"),
    ("<= 4", "end\n"),
    ("2 x <= 4", "end\n"),             # a number must be followed by an operator
    ("x|y <= 4", "end\n"),             # `|` is no name character (:68-96)
    ("x <= 4", ""),                    # no keyword closes the section (:1216)
])
def test_bad_bounds(tmp_path, lines, tail):
    assert H.parse(_bounds_file(tmp_path, lines, tail)).status == BAD_BOUND


def test_binary_section_overrides_bounds(tmp_path):
    """set_boolean_variable (parser.cpp:507-516) runs after the bounds."""
    path = _bounds_file(tmp_path, "x <= 40\n2 <= y <= 9", "binary\nx\ngeneral\ny\nend\n")
    v = H.variables(path)
    assert v[0] == (0, 1, 1) and v[1] == (2, 9, 2) and v[2] == (0, INT_MAX, 0)


def test_fixture_bound_sections():
    """What the reference's io tests hold (lib/test/io.cpp:143-149, :699-704) and
    the two bound lines its solver tests depend on."""
    for lo, hi, ty in H.variables(lp("sudoku.lp")):
        assert (lo, hi, ty) == (1, 9, 2)
    assert all(v == (0, 1, 2) for v in H.variables(lp("general.lp")))
    # prepro.lp: `t = 0` -> [0, 0], then `binaries ... t` -> [0, 1]
    assert all(v == (0, 1, 1) for v in H.variables(lp("prepro.lp")))
    # capmo1_direct.lp: `t = 1` -> [0, 1], t in no constraint and not binary
    caps = H.variables(lp("capmo1_direct.lp"))
    assert sum(1 for v in caps if v[2] != 1) == 1
    assert [v for v in caps if v[2] != 1] == [(0, 1, 0)]
