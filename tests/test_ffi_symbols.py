"""CPU: the C-ABI library loads and exports every symbol include/bxh.h declares;
the oracle and the host library load too.  No compute call is made."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "bxh.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bxh_[a-z_A-Z0-9]+)\s*\(", text)))


def test_header_and_binding_agree(bxh):
    assert header_functions() == sorted(bxh.SYMBOLS)


def test_library_exports_every_declared_symbol(bxh):
    lib = ctypes.CDLL(bxh.LIB_PATH)
    for name in header_functions():
        assert hasattr(lib, name), name


def test_no_cpu_fallback_symbols(bxh):
    """The device library must not carry a host implementation of the sweep."""
    out = os.popen(f"nm -D --defined-only {bxh.LIB_PATH}").read()
    assert "orc_" not in out and "oracle" not in out.lower()


def test_oracle_and_host_libraries_load(oracle):
    from baryonyx_amd import host_ffi

    L = host_ffi.lib()
    for name in ("bxc_default_params", "bxc_parse", "bxc_merged", "bxc_run"):
        assert hasattr(L, name)
    out = os.popen(f"nm -DC --defined-only {host_ffi.LIB_PATH}").read()
    for sym in ("baryonyx::solve(", "baryonyx::optimize(", "baryonyx::make_problem(",
                "baryonyx::make_context(", "baryonyx::context_set_solver_parameters(",
                "baryonyx::is_valid_solution(", "baryonyx::compute_solution("):
        assert sym in out, sym
