"""The rbaryonyx surface (include/rbaryonyx.h): the two functions the R package exports
(rbaryonyx/src/rbaryonyx.cpp:368-392, :519-551), their argument lists and the nine-field result
(:280-288)."""
import ctypes as C
import os
import re

import pytest

from baryonyx_amd import host_ffi as H, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ["solution_found", "error_found", "value", "duration", "variables", "constraints",
          "remaining_constraints", "minimize", "solutions"]


def test_exports_and_header_agree():
    L = H.lib()
    hdr = open(os.path.join(ROOT, "include", "rbaryonyx.h")).read()
    for name in ("bxr_solve_01lp_problem", "bxr_optimize_01lp_problem", "bxr_free_result"):
        assert name in hdr
        getattr(L, name)
    # argument names, in order, of both declarations
    def names(fn):
        body = hdr[hdr.index(fn + "("):]
        body = body[:body.index(");")]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return [a.split()[-1].lstrip("*") for a in body[body.index("(") + 1:].split(",")]
    assert names("bxr_solve_01lp_problem") == ["file_path"] + [n for n, _, _ in H.R_SOLVE_ARGS] + ["out"]
    assert names("bxr_optimize_01lp_problem") == ["file_path"] + [n for n, _, _ in H.R_OPTIMIZE_ARGS] + ["out"]
    assert len(H.R_SOLVE_ARGS) == 23 and len(H.R_OPTIMIZE_ARGS) == 31  # rbaryonyx.cpp:369-392, 520-551


def test_missing_file_is_the_empty_problem(tmp_path):
    """make_problem on a file that is not there returns an empty raw_problem (lpcore.cpp:70-83);
    the R glue does not look at its status and solves it, and solve_problem answers an empty
    problem with `success`, one solution of value objective.value = 0 (itm-solver-common.hpp:
    300-304).  No device is needed for that answer."""
    r = H.solve_01lp_problem(str(tmp_path / "nothing.lp"), verbose=0)
    assert list(r) == FIELDS
    assert r["solution_found"] is True and r["error_found"] is False
    assert r["solutions"] == [0.0] and r["value"] == 0.0
    assert (r["variables"], r["constraints"]) == (0, 0)
    o = H.optimize_01lp_problem(str(tmp_path / "nothing.lp"), verbose=0)
    assert o["error_found"] is False and o["variables"] == 0


@pytest.mark.gpu
def test_solve_and_optimize_through_the_r_surface(tmp_path):
    inst = synth.make_scp(40, 200, 0.08, "int", 5)
    lp = str(tmp_path / "a.lp")
    synth.write_lp(inst, lp)
    r = H.solve_01lp_problem(lp, seed=7, limit=500, verbose=0)
    assert list(r) == FIELDS
    assert r["solution_found"] and not r["error_found"] and r["minimize"]
    # (solve_problem leaves result.remaining_constraints at its default, INT_MAX, core:732-736 /
    # itm-solver-common.hpp:216-224 -- only optimize_problem fills it -- and the glue passes it on)
    assert r["variables"] > 0 and r["constraints"] > 0 and r["remaining_constraints"] == 2**31 - 1
    assert r["solutions"] and r["value"] == r["solutions"][-1] > 0
    # the same call through baryonyx::solve with the parameters the glue sets (pre_order = memory)
    p = H.default_params(seed=7, limit=500, pre_order=1, time_limit=10.0, thread=1)
    ref, _ = H.run(lp, 0, p)
    assert ref.status == 0 and ref.value == r["value"]
    o = H.optimize_01lp_problem(lp, seed=7, thread=8, time_limit=2.0, verbose=0)
    assert o["solution_found"] and o["value"] <= r["value"] and o["remaining_constraints"] == 0
