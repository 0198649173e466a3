"""The device's restatement of libstdc++'s distributions over std::minstd_rand0
(baryonyx_amd/csrc/bxh_rng.hpp: generate_canonical / bernoulli_distribution,
uniform_int_distribution<size_t> with its up-scaling branch -- what crossover and
mutation of optimize() draw, lib/src/itm-optimizer-common.hpp:359-371,494-526)
against libstdc++ itself, draw for draw, on the host (tests/cpp/rng_check.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rng_restatement_equals_libstdcxx(tmp_path):
    exe = str(tmp_path / "rng_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "baryonyx_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "cpp", "rng_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "bad 0" in out.stdout, out.stdout[-500:]
