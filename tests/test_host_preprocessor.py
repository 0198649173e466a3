"""The unit-propagation preprocessor of the host library
(baryonyx_amd/host/preprocessor.cpp) against the expectations of the
reference's own test file lib/test/preprocess.cpp (:47-327)."""
import os

import pytest

from baryonyx_amd import host_ffi as H

LP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lp")


def _write(tmp_path, text):
    p = tmp_path / "pb.lp"
    p.write_text(text)
    return str(p)


BQP = ("minimize\n"
       "-5a + [ 2a * b + 6c * d ] /2 + 10b\n"
       "Subject to:\n"
       "- a + b >= 0\n"
       "b + c - d >= 0\n"
       "Binaries\n"
       "a b c d\n"
       "End\n")
BQP2 = ("minimize\n"
        "-5a + 10a + [ 2a * b + 6c * d ] /2 + 10b - b -b "
        "-b + [2 a * b + 6 c * d] / 2\n"
        "Subject to:\n"
        "- a + b >= 0\n"
        "b + c - d >= 0\n"
        "Binaries\n"
        "a b c d\n"
        "End\n")


@pytest.mark.parametrize("text,lin,quad", [
    (BQP, [(0, -5.0), (1, 10.0)], [(0, 1, 1.0), (2, 3, 3.0)]),   # preprocess.cpp:50-147
    (BQP2, [(0, 5.0), (1, 7.0)], [(0, 1, 2.0), (2, 3, 6.0)]),    # :149-197
])
def test_bqp_clean(tmp_path, text, lin, quad):
    r = H.preprocess(_write(tmp_path, text))
    i = r["info"]
    assert (i.variables, i.affected, i.equal, i.less, i.greater) == (4, 0, 0, 0, 2)
    assert r["linear"] == lin      # variables a, b, c, d are 0..3 in file order
    assert r["quadratic"] == quad


def test_bound_affectation():  # preprocess.cpp:200-234
    path = os.path.join(LP, "bound.lp")
    i = H.preprocess(path)["info"]
    assert (i.variables, i.affected, i.equal, i.less, i.greater) == (6, 0, 0, 2, 0)
    i = H.preprocess(path, fix_zero="f")["info"]
    assert (i.variables, i.affected, i.equal, i.less, i.greater) == (5, 1, 0, 2, 0)


def test_cleaning_affected_variables():  # preprocess.cpp:236-246
    assert H.parse(os.path.join(LP, "prepro.lp")).variables == 23
    i = H.preprocess(os.path.join(LP, "prepro.lp"))["info"]
    assert (i.variables, i.affected) == (2, 21)


@pytest.mark.parametrize("var,value", [(0, True), (0, False), (1, True), (1, False)])
def test_affect_variable(var, value):  # preprocess.cpp:248-299
    r = H.preprocess(os.path.join(LP, "prepro.lp"), affect=(var, value))
    assert (r["info"].variables, r["info"].affected) == (2, 21)
    assert (r["affect"].variables, r["affect"].affected) == (0, 23)


def test_split():  # preprocess.cpp:301-327
    r = H.preprocess(os.path.join(LP, "prepro.lp"), split=0)
    assert (r["split1"].variables, r["split1"].affected) == (0, 23)
    assert (r["split0"].variables, r["split0"].affected) == (0, 23)


def test_capmo1_direct_unused_variable_goes_to_zero():
    """lib/test/solve.cpp:197-240 depends on it: capmo1_direct.lp's bound line `t = 1`
    is [0, 1] for the reference's reader (parser.cpp:944-956), t (cost +14 156 111) is
    in no constraint, and the preprocessor leaves such a variable of a minimisation
    at its lower bound -- nothing is added to the objective's constant.  (Rounds 1 - 4
    read the line as t fixed at 1: constant 14 156 111, every value too large by it.)"""
    r = H.preprocess(os.path.join(LP, "capmo1_direct.lp"))
    assert r["info"].affected == 1
    assert r["info"].objective_constant == 0.0
    assert r["info"].variables == 10100
