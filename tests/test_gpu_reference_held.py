"""-m gpu: the reference's own solver tests (lib/test/solve.cpp,
lib/test/optimize.cpp) against the HIP path, through a COMPILED C++ client of
<baryonyx/core> (tests/cpp/client_reference.cpp: make_context ->
context_set_solver_parameters -> make_problem -> solve / optimize ->
is_valid_solution / compute_solution, as lib/test/solve.cpp:292-303 does), and
every case against the CPU oracle run with the same validated parameters and
seed.  Expectations and the list of what is not reproduced:
tests/test_reference_held.py."""
import os
import subprocess

import numpy as np
import pytest

from baryonyx_amd import host_ffi as H
from helpers import oracle_params, pack_bits
import test_reference_held as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LP = T.LP


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    """Build the client against include/ and libbaryonyx_amd.so, run it once."""
    out = tmp_path_factory.mktemp("client")
    exe = str(out / "client_reference")
    host = os.path.join(ROOT, "baryonyx_amd", "host")
    H.lib()  # fails loudly when the host library is missing
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "client_reference.cpp"), "-L" + host,
                           "-lbaryonyx_amd", "-Wl,-rpath," + host,
                           "-Wl,-rpath," + os.path.join(ROOT, "baryonyx_amd", "csrc"), "-o", exe])
    proc = subprocess.run([exe, LP, str(out)], capture_output=True, text=True, timeout=1500)
    assert proc.returncode == 0, proc.stderr[-2000:]
    cases, affected = {}, {}
    for line in proc.stdout.splitlines():
        tok = line.split()
        if tok and tok[0] == "case":
            cases[tok[1]] = dict(t.split("=", 1) for t in tok[2:])
        elif tok and tok[0] == "affected":
            affected[tok[1]] = {t.split("=")[0]: int(t.split("=")[1]) for t in tok[2:]}
    assert proc.stdout.rstrip().endswith("done")
    return cases, affected


def _ok(c, value=True):
    assert "exception" not in c, c
    assert c["parsed"] == "1" and c["status"] == "0" and c["ok"] == "1" and c["valid"] == "1", c
    if value:
        assert float(c["value"]) == float(c["recomputed"])


def _same_as_oracle(oracle, c, inst, seed, raw, cap=T.CAP):
    """status / value / x of the client's solve == the oracle's solve()."""
    ores, oxw = oracle.solve(inst, 1, oracle_params(oracle, cap=cap, sort_mode=1, seed=seed, **raw))
    assert int(c["status"]) == ores.status
    if ores.status == 0:
        assert float(c["value"]) == ores.value
    return ores, oxw


SEEDS = T.SEEDS


@pytest.mark.parametrize("seed", SEEDS)
def test_preprocessor(client, oracle, seed):
    """solve.cpp:77-193."""
    cases, affected = client
    c = cases[f"prepro-s{seed}"]
    _ok(c)
    assert c["affected"] == "21" and c["names"] == "2"
    a = affected[f"prepro-s{seed}"]
    expect = dict(w=0, a=0, t=0, ZZ=1, c1=0, c2=0, c3=0, c4=0, c5=0, c6=0, d1=1, d2=1, d3=1, d4=0, d5=0,
                  d6=0, b=1)  # :96-166
    for name, value in expect.items():
        assert a[name] == value, name
    assert int(c["nsol"]) >= 1 and float(c["value"]) > 6.0
    _same_as_oracle(oracle, c, H.merged(os.path.join(LP, "prepro.lp")), seed, dict(cost_norm=4))


@pytest.mark.parametrize("seed", SEEDS)
def test_preprocessor_2(client, oracle, seed):
    """solve.cpp:197-240 (capmo1_direct.lp, default parameters): success, valid,
    recomputed value == value, **value < 1 156 908** (:217); and the oracle's
    point."""
    c = client[0][f"capmo1_direct-s{seed}"]
    _ok(c)
    assert float(c["value"]) < 1156908
    assert c["affected"] == "1"  # `t`: [0, 1] by parser.cpp:944-956, in no constraint, cost > 0 -> 0
    _same_as_oracle(oracle, c, H.merged(os.path.join(LP, "capmo1_direct.lp")), seed, {})


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("name,text,raw", [
    ("real_cost", T.REAL_COST, {}), ("negative-coeff5", T.NEGATIVE_COEFF5, {}),
    ("z_coefficient_a", T.Z_COEFFICIENT_A, {}), ("z_coefficient_b", T.Z_COEFFICIENT_B, {}),
])
def test_inline_problems(client, oracle, tmp_path, name, text, raw, seed):
    """solve.cpp:261-290, :404-429, :596-639."""
    c = client[0][f"{name}-s{seed}"]
    _ok(c)
    if name == "real_cost":
        assert float(c["value"]) < 0.0
    _, inst = T._merged_text(tmp_path, text)
    ores, oxw = _same_as_oracle(oracle, c, inst, seed, raw)
    assert np.array_equal(pack_bits(np.array([int(b) for b in c["x"]][: inst.n], np.uint8)), oxw)


@pytest.mark.parametrize("seed", SEEDS)
def test_quadratic_preprocessor(client, seed):
    """solve.cpp:242-259: parses, solve() returns a solution."""
    c = client[0][f"quadratic-s{seed}"]
    assert c["parsed"] == "1" and c["ok"] == "1"


FILE_CASES = [
    ("assignment", "assignment_problem_1.lp", {}),
    ("negative-coeff", "negative-coeff.lp", dict(limit=50)),
    ("negative-coeff2", "negative-coeff2.lp", {}),
    ("negative-coeff3", "negative-coeff3.lp", dict(limit=10000)),
    ("negative-coeff4", "negative-coeff4.lp", dict(limit=50)),
    ("flat30-7", "flat30-7.lp", dict(limit=200000, delta=0.001, kappa_min=0.3, kappa_step=1e-10, kappa_max=1.0,
                                     order=1)),
    ("uf50-0448", "uf50-0448.lp", dict(limit=200000, theta=0.5, delta=1.0, kappa_min=0.1, kappa_step=1e-17,
                                       kappa_max=1.0, alpha=2.0, w=60, order=2)),
    ("aim-50", "aim-50-1_6-yes1-2.lp", dict(limit=200000, theta=0.6, delta=0.01, kappa_step=2 * 10e-4,
                                            kappa_max=100.0, alpha=1.0, w=20)),
]


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("name,file,raw", FILE_CASES, ids=[f[0] for f in FILE_CASES])
def test_fixture_files(client, oracle, name, file, raw, seed):
    """solve.cpp:292-303, :337-402, :531-593."""
    c = client[0][f"{name}-s{seed}"]
    _ok(c)
    if name == "negative-coeff2":
        assert int(c["affected"]) + int(c["names"]) == 4  # :367-368
    _same_as_oracle(oracle, c, H.merged(os.path.join(LP, file)), seed, raw)


@pytest.mark.parametrize("seed", SEEDS)
def test_8_queens_fixed_cost(client, oracle, seed):
    """solve.cpp:431-474."""
    c = client[0][f"queens_fixed-s{seed}"]
    _ok(c)
    assert c["x"].count("1") == 8
    inst = H.merged(os.path.join(LP, "8_queens_puzzle.lp"))
    inst.obj_factor = np.array(T.QUEENS_FIXED_COST, np.float64)
    _same_as_oracle(oracle, c, inst, seed, T.QUEENS_PARAMS)


@pytest.mark.parametrize("draw", range(10))
def test_random_costs(client, oracle, draw):
    """solve.cpp:305-335 and :476-508."""
    for name, file, raw in (("assignment_random", "assignment_problem_1.lp", T.RANDOM_COST_PARAMS),
                            ("queens_random", "8_queens_puzzle.lp", dict(order=3, **T.QUEENS_PARAMS))):
        c = client[0][f"{name}-d{draw}"]
        _ok(c)
        inst = H.merged(os.path.join(LP, file))
        inst.obj_factor = T.drawn_costs(draw, len(inst.obj_factor))
        _same_as_oracle(oracle, c, inst, 1 + draw, raw)


@pytest.mark.parametrize("seed", SEEDS)
def test_qap_solve(client, oracle, seed):
    """solve.cpp:510-529: valid whenever solve() succeeds; same outcome as the oracle."""
    c = client[0][f"qap_solve-s{seed}"]
    assert "exception" not in c
    if c["ok"] == "1":
        assert c["valid"] == "1"
    _same_as_oracle(oracle, c, H.merged(os.path.join(LP, "small4.lp")), seed,
                    dict(limit=20000, theta=0.5, delta=0.01, kappa_step=10e-4, kappa_max=10.0, alpha=0.0, w=20))


def test_n_queens(client, oracle):
    """optimize.cpp:89-165: all 30 found; each equal to the oracle's optimize()
    after the same number of rounds."""
    cplex = T._cplex_values()
    for i in range(30):
        c = client[0][f"n-queens-{i}"]
        assert "exception" not in c, c
        assert c["remaining"] == "0" and c["status"] == "0" and c["valid"] == "1", (i, c)
        assert float(c["value"]) == float(c["recomputed"]) >= cplex[i]
        inst = H.merged(os.path.join(LP, "n-queens", f"n-queens-problem-{i}.lp"))
        res, _ = oracle.optimize(inst, 1, oracle_params(oracle, seed=100 + i, **T.NQUEENS_PARAMS), thread=2,
                                 rounds=1500)
        assert (res.status, res.remaining_constraints, res.value) == (0, 0, float(c["value"])), i


def test_qap_optimize(client, oracle):
    """optimize.cpp:55-87 on the HIP path: the same outcome as the oracle (eight
    rows stay violated; tests/test_reference_held.py explains)."""
    c = client[0]["qap_optimize"]
    assert "exception" not in c, c
    assert c["status"] == "3" and c["remaining"] == "8"
