"""-m gpu: baryonyx::solve / baryonyx::optimize of the C++ host library (every
sweep a HIP launch through the C ABI) on the reference's fixture files, with
the reference tests' own expectations (lib/test/solve.cpp: status == success,
is_valid_solution, compute_solution == reported value), and against the CPU
oracle run with the same parameters and seed."""
import os

import numpy as np
import pytest

from baryonyx_amd import host_ffi as H
from baryonyx_amd import synth
from helpers import is_feasible, objective, pack_bits

pytestmark = pytest.mark.gpu
LP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lp")
CAP = 200000

CASES = [
    ("assignment_problem_1.lp", dict()),
    ("assignment_problem_2.lp", dict()),
    ("negative-coeff.lp", dict(limit=50, w=0.05)),
    ("negative-coeff3.lp", dict(limit=10000, w=0.05)),
    ("negative-coeff4.lp", dict(limit=50, w=0.05)),
    ("8_queens_puzzle.lp", dict(limit=CAP, theta=0.5, delta=0.02, alpha=1.0, w=40.0)),
    ("flat30-7.lp", dict(limit=CAP, delta=0.001, kappa_min=0.3, kappa_step=1e-10, kappa_max=1.0,
                         order=1, w=0.0)),
    ("aim-50-1_6-yes1-2.lp", dict(limit=CAP, theta=0.6, delta=0.01, alpha=1.0, w=20.0)),
    # the other constraint orders through the whole solve() loop
    ("assignment_problem_1.lp", dict(order=3)),
    ("assignment_problem_2.lp", dict(order=4)),
    ("negative-coeff4.lp", dict(limit=50, w=0.05, order=5)),
    ("negative-coeff.lp", dict(limit=50, w=0.05, order=6)),
    ("aim-50-1_6-yes1-2.lp", dict(limit=CAP, theta=0.6, delta=0.01, alpha=1.0, w=20.0, order=5)),
]


@pytest.mark.parametrize("name,params", CASES,
                         ids=[c[0] + ("-order%d" % c[1]["order"] if "order" in c[1] else "") for c in CASES])
def test_solve_fixture_matches_oracle(oracle, name, params):
    path = os.path.join(LP, name)
    kw = dict(seed=123456789)
    kw.update(params)
    res, x = H.run(path, 0, H.default_params(**kw))
    assert res.status == 0 and res.valid == 1
    assert res.recomputed_value == res.value

    # same run on the CPU oracle (canonical tie rule == the device's):
    # identical status / objective / x
    inst = H.merged(path)
    okw = dict(seed=123456789, sort_mode=1, limit=1000, w=50.0)
    okw.update(params)
    if "w" in params and params.get("limit", 1000) != CAP and params["w"] < 1:
        okw["w"] = params["limit"] * params["w"]  # context_set_solver_parameters
    ores, oxw = oracle.solve(inst, 1, oracle.default_params(**okw))
    assert ores.status == res.status
    assert ores.value == res.value
    assert np.array_equal(pack_bits(x[: inst.n].astype(np.uint8)), oxw)


def test_optimize_bounded_budget(tmp_path):
    """optimize() with a sweep budget on a small covering instance: the batch
    of restarts returns a feasible point no worse than the first sweep's."""
    inst = synth.make_scp(60, 400, 0.06, "int", 12)
    path = str(tmp_path / "scp.lp")
    synth.write_lp(inst, path)
    solve_res, _ = H.run(path, 0, H.default_params(seed=5, init_policy_random=0.0))
    assert solve_res.status == 0 and solve_res.valid == 1
    res, x = H.run(path, 1, H.default_params(seed=5, thread=16, sweep_budget=400, time_limit=60.0))
    assert res.status == 0 and res.valid == 1 and res.remaining_constraints == 0
    assert res.value == res.recomputed_value
    assert res.value <= solve_res.value
