"""-m gpu: randomised differential test of the inequalities-Z kernel's two forms
against the oracle on instances built to hit the rare paths of the wave-per-row
form side by side: one-sided rows whose scan finds nothing (the `unconstrained`
kinds then call branch-and-bound / the exhaustive patterns from inside a group),
tied rows (integer costs: shuffles in row order from the one engine) next to
tie-free ones, rows of exactly 63 / 64 elements (the wave path's limit), rows
that clash with a neighbour (group boundaries) and long lists (several tables)."""
import numpy as np
import pytest

from baryonyx_amd import synth
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
INT_MIN = -(2 ** 31)


def make_zfuzz(seed, m, n, maximize):
    rs = np.random.RandomState(seed)
    rows, facs, cmin, cmax = [], [], [], []
    for k in range(m):
        t = rs.randint(0, 8)
        if t == 0:      # a long row: the general path between two groups
            size = int(rs.choice([63, 64, 70]))
        elif t in (1, 2):  # short one-sided rows with a negative coefficient: the scan may find nothing
            size = int(rs.randint(3, 9))
        elif t == 3:    # one-sided, ten or more elements: unconstrained branch and bound
            size = int(rs.randint(10, 16))
        elif t == 4:    # both sides bounded: the constrained kinds (their branch and bound enumerates: short)
            size = int(rs.randint(4, 13))
        else:
            size = int(rs.randint(4, 40))
        size = min(size, n)
        cols = np.sort(rs.choice(n, size=size, replace=False))
        f = rs.randint(1, 6, size=size).astype(np.int32)
        if t in (1, 2, 3):
            neg = rs.random_sample(size) < 0.35
            neg[0] = True
            f[neg] *= -1
            f[np.abs(f) < 2] *= 2
        lower, upper = int(f[f < 0].sum()), int(f[f > 0].sum())
        pick = rs.random_sample(size) < 0.5
        ssum = int(f[pick].sum())
        if t in (1, 2, 3):
            lo, hi = INT_MIN, min(ssum, -1) if lower < 0 else ssum  # a window the running sum can jump over
            hi = max(hi, lower)
        elif t == 4:    # both sides: the constrained kinds (general path)
            lo, hi = max(ssum - 1, lower), min(ssum + 1, upper)
        else:
            lo, hi = INT_MIN, max(ssum, (upper + 1) // 2)
        rows.append(cols)
        facs.append(f)
        cmin.append(lo)
        cmax.append(hi)
    row_ptr, col = synth._rows_to_csr(rows, n)
    cost = rs.randint(1, 8, size=n).astype(np.float64) if seed % 2 else 1.0 + 99.0 * rs.random_sample(n)
    return synth.Instance(m=m, n=n, row_ptr=row_ptr, col=col, factor=np.concatenate(facs).astype(np.int32),
                          cmin=np.asarray(cmin, np.int64).clip(INT_MIN, None).astype(np.int32),
                          cmax=np.asarray(cmax, np.int32), obj_var=np.arange(n, dtype=np.int32), obj_factor=cost,
                          maximize=maximize, kind=4, name=f"zfuzz-{seed}")


@pytest.mark.parametrize("zwave", ["1", "0"], ids=["wave-per-row", "plain"])
@pytest.mark.parametrize("seed", range(12))
def test_z_forms_on_random_instances(oracle, bxh, seed, zwave, monkeypatch):
    monkeypatch.setenv("BXH_ZWAVE", zwave)
    m = 90 + 37 * (seed % 4)
    n = [260, 700, 1500][seed % 3]
    inst = make_zfuzz(900 + seed, m, n, maximize=bool(seed & 1))
    order = [0, 3, 1, 2, 4, 5, 0, 6, 0, 2, 1, 0][seed]  # (pi_sign_change / cycle need n <= m)
    form = run_case(oracle, bxh, inst, float_type=1 if seed % 5 else 0, maximize=bool(seed & 1), order=order,
                    x0p=[0.0, 0.2, 0.5][seed % 3], batch=2, sweeps=8, push_at=(4,))
    assert form == (6 if zwave == "1" else 3)
