"""The scripts under tools/ are run by hand on the GPU box (profiling, A/B timing, diagnostics): here they
are only checked to parse -- a typo in one of them costs a GPU call to find."""
import os
import py_compile
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _files(ext):
    out = []
    for base, _, names in os.walk(os.path.join(ROOT, "tools")):
        out += [os.path.join(base, n) for n in sorted(names) if n.endswith(ext)]
    return out


@pytest.mark.parametrize("path", _files(".py"), ids=lambda p: os.path.relpath(p, ROOT))
def test_python_tools_compile(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)


@pytest.mark.parametrize("path", _files(".sh"), ids=lambda p: os.path.relpath(p, ROOT))
def test_shell_tools_parse(path):
    subprocess.run(["bash", "-n", path], check=True)
