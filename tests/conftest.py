import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line(
        "markers",
        "outside_reference_domain: the oracle is EXPECTED to leave the reference's working domain in "
        "this test -- a row update ends with its row violated, where the reference's default build "
        "stops at bx_expects(is_valid_constraint) (itm-common.hpp:464, "
        "itm-solver-inequalities-Z.cpp:436).  Parity there is HIP == oracle == the reference's "
        "BARYONYX_FULL_OPTIMIZATION build only; the count is reported, not asserted.")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import oracle_ffi

    oracle_ffi.lib()
    return oracle_ffi


@pytest.fixture(scope="session")
def bxh():
    """The HIP device layer through its C ABI; never falls back to anything."""
    from baryonyx_amd import _ffi

    _ffi.lib()
    return _ffi


_DOMAIN_REPORT = []


@pytest.fixture(autouse=True)
def reference_domain(request):
    """Every test that drives the oracle claims reference behaviour: the oracle
    must then never pass a point where the reference's default build fails its
    contract check `bx_expects(is_valid_constraint(slv, k, x))` behind a row
    update (oracle/itm_oracle.hpp: Solver::contract_violations).  Tests marked
    `outside_reference_domain` report their count instead."""
    from oracle import oracle_ffi

    before = oracle_ffi.contract_violations_total()
    yield
    fired = oracle_ffi.contract_violations_total() - before
    if request.node.get_closest_marker("outside_reference_domain") is not None:
        _DOMAIN_REPORT.append((request.node.nodeid, fired))
        return
    if fired and os.environ.get("BX_DOMAIN_REPORT_ONLY") == "1":  # survey run: list, do not fail
        _DOMAIN_REPORT.append((request.node.nodeid + "  [UNMARKED]", fired))
        return
    assert fired == 0, (
        f"{fired} row update(s) of the oracle ended with their row violated: the reference's default "
        "build stops there (itm-common.hpp:464 / itm-solver-inequalities-Z.cpp:436); mark the test "
        "`outside_reference_domain` if that is what it is about")


def pytest_terminal_summary(terminalreporter):
    if _DOMAIN_REPORT:
        terminalreporter.write_sep("-", "tests outside the reference's working domain (contract check would fire)")
        for nodeid, fired in _DOMAIN_REPORT:
            terminalreporter.write_line(f"{fired:8d}  {nodeid}")


# ---------------------------------------------------------------------------
# The parity cases that are OUTSIDE the reference's working domain, in one place.
# Instances with mixed-sign general coefficients (synth.make_zmix, the random
# instances of test_gpu_zfuzz) make a Z row update end with its own row violated:
# select_variables' running sum / the branch-and-bound treat the negative
# coefficients through their absolute values, and the reference's default build
# stops there at `bx_expects(is_valid_constraint(*this, k, x))`
# (itm-solver-inequalities-Z.cpp:436; itm-common.hpp:464 for the 01 / 101 solvers).
# The reference's own tests hold no such instance; it does run them when built with
# BARYONYX_FULL_OPTIMIZATION (debug.hpp:103-111), and that is what the oracle and
# the HIP path are compared on here.  Every OTHER test asserts a count of zero.
# (function name, substring of the parametrised id or "" for every case)
OUTSIDE_REFERENCE_DOMAIN = [
    ("test_z_sweeps_match_oracle", "zmix-"),
    ("test_z_sweeps_match_oracle", "z-wave-mixed-kinds"),
    ("test_quadratic_objective_matches_oracle", "q-zmix"),
    ("test_solve_general_coefficients_matches_oracle", "zmix"),
    ("test_z_forms_on_random_instances", ""),
]


def pytest_collection_modifyitems(config, items):
    for item in items:
        fn = getattr(item, "originalname", None) or item.name.split("[")[0]
        for name, part in OUTSIDE_REFERENCE_DOMAIN:
            if fn == name and part in item.name:
                item.add_marker(pytest.mark.outside_reference_domain)
