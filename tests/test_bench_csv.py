"""tools/bx_bench.py: the CSV side of the reference CLI's --bench mode
(app/src/benchmark.cpp:332-774): a column is appended, the other solvers'
values survive, a missing instance becomes `inf`."""
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bx_bench  # noqa: E402


def write_csv(tmp_path):
    p = tmp_path / "scp.csv"
    p.write_text("file,status,cplex,lsp\nscp41,429,429,429\nscp42,512,512,inf\nmissing,1,1,1\n")
    os.makedirs(tmp_path / "scp")
    return str(p)


def test_csv_round_trip_with_a_stub_runner(tmp_path):
    csv = write_csv(tmp_path)

    def runner(path):
        if "missing" in path:
            raise FileNotFoundError(path)
        return (430.0 if "scp41" in path else 512.0), False

    cur = bx_bench.run(csv, "bx-mi355x", {}, runner=runner)
    assert cur[:2] == [430.0, 512.0] and math.isinf(cur[2])
    solvers, rows = bx_bench.load(csv)
    assert solvers == ["status", "cplex", "lsp", "bx-mi355x"]
    assert rows[0] == ("scp41", [429.0, 429.0, 429.0, 430.0])
    assert rows[1][1][2] == math.inf and rows[2][1][3] == math.inf
    stats = open(str(tmp_path / "scp-stats.csv")).read().splitlines()
    assert stats[0] == "file,status,cplex,lsp,bx-mi355x"
    assert stats[1] == "scp41,1,1,1,2" and stats[2] == "scp42,1,1,2,1"


@pytest.mark.gpu
def test_csv_bench_on_the_gpu(tmp_path):
    """two small covering instances, one as .lp, one in OR-Library's .scp form"""
    from baryonyx_amd import synth
    csv = tmp_path / "mini.csv"
    csv.write_text("file,best\na,1\nb,1\n")
    os.makedirs(tmp_path / "mini")
    synth.write_lp(synth.make_scp(40, 200, 0.08, "int", 5), str(tmp_path / "mini" / "a.lp"))
    synth.write_scp(synth.make_scp(50, 300, 0.06, "int", 6), str(tmp_path / "mini" / "b.scp"))
    cur = bx_bench.run(str(csv), "bx-mi355x", dict(seed=3, thread=8, sweep_budget=200, time_limit=60.0))
    assert all(math.isfinite(v) and v > 0 for v in cur)
    solvers, rows = bx_bench.load(str(csv))
    assert solvers == ["best", "bx-mi355x"] and [r[1][1] for r in rows] == cur
