"""CPU: the product's `.lp` reader (baryonyx_amd/host/lp_parser.cpp) against an independent
restatement of the reference's reader (tests/lp_grammar_restatement.py, from
lib/src/parser.cpp:268-1257) -- on every fixture file the reference's tests use, on texts
written for the reference's less obvious readings, and on generated token soups.  Both must
give the same status and the same problem: variables in order of first appearance with
their bounds and types, objective, quadratic terms, the three constraint lists.  (The
loader is the one piece of the path that the oracle cannot check: the oracle is fed by
it -- which is how the bound-section bug of rounds 1 - 4 survived.)"""
import glob
import os
import random

import pytest

from baryonyx_amd import host_ffi as H
import lp_grammar_restatement as R

LP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lp")
FIXTURES = sorted(glob.glob(os.path.join(LP, "*.lp")) + glob.glob(os.path.join(LP, "n-queens", "*.lp")))


def _same(path, text):
    status, dump = H.dump(path)
    try:
        rstatus, prob = R.parse(text)
    except R.ReferenceAborts:
        assert status != 0, "the reference aborts on this text; the product must at least refuse it"
        return "aborts"
    assert status == rstatus, (status, rstatus)
    if status == 0:
        want = R.canonical(prob)
        if dump != want:
            a, b = dump.split("\n"), want.split("\n")
            for i, (x, y) in enumerate(zip(a, b)):
                assert x == y, f"line {i}: product {x!r} != reference {y!r}"
            assert len(a) == len(b)
    return status


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.relpath(f, LP) for f in FIXTURES])
def test_fixture_files_read_as_the_reference_reads_them(path):
    with open(path, encoding="latin-1", newline="") as f:
        text = f.read()
    assert _same(path, text) == 0


QUIRKS = {
    # a number token runs over e / E: `3ex` is 3 times x, `2e1y` is 20 y (:385-447)
    "number-eats-e": "minimize 3ex + 2e1y + 4E\nst\nc1: 3ex + 2Ey >= 1\nbinary x y\nend\n",
    # `1e+5` is `1e`, `+`, `5`: 1, then the constant 5
    "exponent-sign-splits": "minimize x + 1e+5 y\nst\nx + y >= 1\nend\n",
    # a second number in a row is a NAME (digits are name characters, :68-96)
    "number-then-number": "minimize 3 5 + x\nst\n2 7 + x >= 1\nend\n",
    # a constant in a constraint names the EMPTY variable (:1143)
    "constant-in-constraint": "minimize x\nst\nx + 3 >= 1\nend\n",
    # a sign in front of a sign: the first one is -1 times nothing
    "double-sign": "minimize - - x + y\nst\nx - + y >= 1\nend\n",
    # labels, all operator spellings, signed right-hand sides, truncation of reals
    "operators": "maximize obj: 2.7 a - 0.5 b\nsubject to:\nc1: a + b =< 3\nc2: a - b => -2\n"
                 "c3: 2.9 a == 1\nc4: a < + 4\nc5: b > - 1\nc6: a = 2\nc7: - 2.5 a - 1.5 b <= 7.9\nend\n",
    # a variable named twice adds up, also to zero (the element stays)
    "duplicates": "minimize x + x - 2 x + y\nst\nx + y - x >= 1\n2 y + 3 y <= 9\nend\n",
    # quadratic block: products, squares (`^2` and `^ 2`), `/2` and `/ 2`, the sign in front
    "quadratic": "minimize x - [ 2 x * y + 3 y ^2 - z ^ 2 + 4 y * x ] / 2 + [ x * z ]/2 + 3\nst\nx + y + z >= 1\nend\n",
    # section spellings with and without the colon
    "sections": "MINIMUM x + y\nS.T.:\nx + y >= 1\nBOUNDS:\nx <= 4\nBINARIES:\ny\nGENERALS:\nx\nEND:\n",
    "sush-that": "min x\nsush that\nx >= 1\nend\n",
    # `such that` is NOT a spelling of the reference (:152): the objective runs on
    "such-that": "min x\nsuch that\nx >= 1\nend\n",
    # comments: only a word that STARTS with a backslash
    "comments": "minimize x \\ this is dropped\n + y\\z\nst\nx + y\\z >= 1 \\* so is this\nend\n",
    # objective label, keyword as first token behind the sense
    "objective-label": "minimize cost: x + y\nst\nx >= 1\nend\n",
    "no-objective": "minimize\nst\nx + y >= 1\nend\n",
    # names with every punctuation the reference accepts; `inf` is a number
    "names": "minimize a!b + c#d + e$f + g(1) + h,i + j.k + l;m + n?o + p@q + r_s + t{u} + v~w + \"z\"\n"
             "st\na!b + g(1) >= 1\nend\n",
    "inf-is-a-number": "minimize inf + x\nst\nx >= 1\nend\n",
    # a name with a character outside the set is a bad objective / constraint
    "bad-name-objective": "minimize x|y\nst\nx >= 1\nend\n",
    "bad-name-constraint": "minimize x\nst\nx + y|z >= 1\nend\n",
    # trailing text behind end, a missing end
    "trailing": "minimize x\nst\nx >= 1\nend\nmore\n",
    "missing-end": "minimize x\nst\nx >= 1\nbinary x\n",
    # binaries / generals of unknown variables
    "unknown-binary": "minimize x\nst\nx >= 1\nbinary y\nend\n",
    "unknown-general": "minimize x\nst\nx >= 1\ngeneral y\nend\n",
    # no white space at all between tokens
    "dense": "minimize 2x+3y-z\nst\nc:2x+3y-z>=1\nd:x-y<=2\nbound\n0<=x<=5\nend\n",
    # where the reference trips its own contract check (pop_front(0)): an error here
    "operator-first": "minimize x\nst\n>= 1\nend\n",
    "missing-rhs": "minimize x\nst\nx >=\nend\n",
    "stray-in-objective": "minimize x = 3\nst\nx >= 1\nend\n",
    "empty": "",
    "only-sense": "minimize",
}


@pytest.mark.parametrize("name", sorted(QUIRKS))
def test_the_reference_s_less_obvious_readings(tmp_path, name):
    path = str(tmp_path / "q.lp")
    with open(path, "w", encoding="latin-1", newline="") as f:
        f.write(QUIRKS[name])
    _same(path, QUIRKS[name])


def test_some_readings_spelled_out(tmp_path):
    """What a few of the texts above come to, so that the restatement is not the only witness."""
    def dump(text):
        p = str(tmp_path / "s.lp")
        with open(p, "w", newline="") as f:
            f.write(text)
        return H.dump(p)
    st, d = dump(QUIRKS["number-eats-e"])
    assert st == 0 and "var x 0 1 1" in d and "var y 0 1 1" in d and "obj 0 3\n" in d + "\n"
    assert "obj 1 20" in d and "constant 4" in d  # 3 x + 20 y + 4
    st, d = dump(QUIRKS["number-then-number"])
    assert st == 0 and d.split("\n")[2] == "var 5 0 2147483647 0"  # the variable named `5`
    st, d = dump(QUIRKS["constant-in-constraint"])
    assert st == 0 and "var  0 2147483647 0" in d  # the empty name
    st, d = dump(QUIRKS["such-that"])
    assert st != 0
    st, d = dump(QUIRKS["dense"])
    assert st == 0 and "var x 0 5 0" in d and "ge 0 1 [c] 2*0 3*1 -1*2" in d and "le 1 2 [d] 1*0 -1*1" in d


ALPHABET = (["minimize", "st", "bounds", "binary", "general", "end", "subject", "to", "sush", "that", "bin", "gen"]
            + ["x", "y", "z1", "e2", "E3", "w_1", "a.b", "q(1)", "inf", "free"]
            + ["0", "1", "2", "3.5", "10", "1e3", "2e", ".5", "7.", "1e+2", "3ex", "2E1y"]
            + ["+", "-", "+", "-", "<", ">", "=", "<=", ">=", "=<", "=>", "==", ":", "[", "]", "*", "^", "^2", "/", "/2"]
            + ["\n", "\n", "\\c", "|"])


def _soup(rng):
    """A text that mostly follows the grammar, with tokens swapped for random ones here and there."""
    def term():
        t = ""
        if rng.random() < 0.6:
            t += rng.choice(["+", "-", "+ ", "- "])
        if rng.random() < 0.7:
            t += rng.choice(["1", "2", "3.5", "10", "1e3", "2e", ".5", "7."]) + rng.choice(["", " "])
        return t + rng.choice(["x", "y", "z1", "e2", "E3", "w_1", "a.b", "q(1)"])
    parts = [rng.choice(["minimize", "maximize", "min", "MAX", "minimum"]), "\n"]
    if rng.random() < 0.3:
        parts += ["obj:", " "]
    for _ in range(rng.randint(0, 5)):
        parts += [term(), " "]
    if rng.random() < 0.25:
        parts += ["+ [ 2 x * y + y ^2 ] / 2 "]
    parts += ["\n", rng.choice(["st", "subject to", "s.t.", "st:", "sush that"]), "\n"]
    for c in range(rng.randint(0, 4)):
        if rng.random() < 0.5:
            parts += ["c%d:" % c, rng.choice(["", " "])]
        for _ in range(rng.randint(1, 4)):
            parts += [term(), " "]
        parts += [rng.choice(["<=", ">=", "=", "<", ">", "=<", "=>", "=="]), " ",
                  rng.choice(["", "- ", "+ ", "-"]), rng.choice(["0", "1", "2", "3.5", "10"]), "\n"]
    if rng.random() < 0.6:
        parts += [rng.choice(["bounds", "bound", "Bounds:"]), "\n"]
        for _ in range(rng.randint(0, 3)):
            v = rng.choice(["x", "y", "z1", "e2", "w_1"])
            parts += [rng.choice([f"{v} <= 4", f"{v} >= 1", f"{v} = 1", f"0 <= {v} <= 9", f"-1 <= {v} <= 3",
                                  f"2 <= {v}", v, f"- inf <= {v} <= 5", f"{v} <= 2.5"]), "\n"]
    if rng.random() < 0.6:
        parts += [rng.choice(["binary", "binaries", "bin"]), "\n", " ".join(rng.sample(["x", "y", "z1", "w_1"], 2)), "\n"]
    if rng.random() < 0.3:
        parts += ["general", "\n", rng.choice(["x", "e2", "a.b"]), "\n"]
    parts += ["end", "\n"]
    # noise: replace / insert / drop a few tokens
    for _ in range(rng.choice([0, 0, 0, 1, 1, 2, 3])):
        i = rng.randrange(len(parts))
        what = rng.random()
        if what < 0.4:
            parts[i] = rng.choice(ALPHABET)
        elif what < 0.7:
            parts.insert(i, rng.choice(ALPHABET) + rng.choice(["", " "]))
        else:
            del parts[i]
    return "".join(parts)


def test_generated_texts(tmp_path):
    rng = random.Random(20260105)
    path = str(tmp_path / "g.lp")
    seen = {}
    for i in range(3000):
        text = _soup(rng)
        with open(path, "w", encoding="latin-1", newline="") as f:
            f.write(text)
        try:
            out = _same(path, text)
        except AssertionError as e:
            raise AssertionError(f"text {i}:\n{text}\n{e}")
        seen[out] = seen.get(out, 0) + 1
    # the generator reaches the good path, several kinds of refusal and the abort cases
    assert seen.get(0, 0) > 300 and len(seen) >= 5, seen
