"""Synthetic 0-1 LP instances shaped like the BASELINE.json configs.

OR-Library / telebus files are not vendored by the reference
(`benchmarks/download.sh:36-40`) and there is no network, so the bench and the
parity tests run on generated instances `G(kind, m, n, density, cost_kind,
seed)` as SURVEY.md section 8(d) defines them.  Output is the merged-constraint
CSR form a baryonyx Solver constructor receives
(`lib/src/itm-common.hpp:55-71`, `lib/src/merged-constraint.cpp:297-343`).

numpy's legacy `RandomState` (MT19937) is used because its stream is frozen
across numpy versions: the same seed gives the same instance on every box.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

INT_MIN = -(2**31)
INT_MAX = 2**31 - 1


@dataclass
class Instance:
    """Merged constraints + objective of one 0-1 LP (all numpy, C order)."""

    m: int
    n: int
    row_ptr: np.ndarray  # int32 [m+1]
    col: np.ndarray  # int32 [nnz]
    factor: np.ndarray  # int32 [nnz]
    cmin: np.ndarray  # int32 [m]
    cmax: np.ndarray  # int32 [m]
    obj_var: np.ndarray  # int32 [n_obj]
    obj_factor: np.ndarray  # float64 [n_obj]
    obj_constant: float = 0.0
    q_a: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    q_b: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    q_factor: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    maximize: bool = False
    kind: int = 1  # 0 eq01, 1 ineq01, 2 eq101, 3 ineq101, 4 ineqZ
    name: str = ""

    @property
    def nnz(self) -> int:
        return int(self.col.shape[0])

    def dense_costs(self) -> np.ndarray:
        c = np.zeros(self.n, np.float64)
        np.add.at(c, self.obj_var, self.obj_factor)
        return c


def _rows_to_csr(rows, n):
    row_ptr = np.zeros(len(rows) + 1, np.int32)
    for k, r in enumerate(rows):
        row_ptr[k + 1] = row_ptr[k] + len(r)
    col = (
        np.concatenate(rows).astype(np.int32)
        if rows
        else np.zeros(0, np.int32)
    )
    return row_ptr, col


def _random_support(rs, m, n, density):
    """Row supports: each column w.p. `density`, rows non-empty, then every
    empty column is added to one random row."""
    rows = []
    used = np.zeros(n, bool)
    for _ in range(m):
        mask = rs.random_sample(n) < density
        if not mask.any():
            mask[rs.randint(0, n)] = True
        used |= mask
        rows.append(np.flatnonzero(mask))
    for j in np.flatnonzero(~used):
        k = rs.randint(0, m)
        rows[k] = np.union1d(rows[k], [j])
    return rows


def _costs(rs, n, cost_kind):
    if cost_kind == "int":
        return rs.randint(1, 101, size=n).astype(np.float64)
    if cost_kind == "real":
        return 1.0 + 99.0 * rs.random_sample(n)
    raise ValueError(cost_kind)


def make_scp(m, n, density, cost_kind="int", seed=1) -> Instance:
    """Set covering: min c.x, every row sum(x) >= 1 (inequalities-01 path)."""
    rs = np.random.RandomState(seed)
    rows = _random_support(rs, m, n, density)
    row_ptr, col = _rows_to_csr(rows, n)
    c = _costs(rs, n, cost_kind)
    return Instance(
        m=m,
        n=n,
        row_ptr=row_ptr,
        col=col,
        factor=np.ones(col.shape[0], np.int32),
        cmin=np.ones(m, np.int32),
        cmax=np.full(m, INT_MAX, np.int32),
        obj_var=np.arange(n, dtype=np.int32),
        obj_factor=c,
        maximize=False,
        kind=1,
        name=f"scp-{m}x{n}-d{density}-{cost_kind}-s{seed}",
    )


def make_spp(m, n, density, cost_kind="int", seed=4) -> Instance:
    """Set partitioning: min c.x, every row sum(x) == 1 (equalities-01 path).
    A feasible partition is planted: the first columns are random disjoint
    blocks covering all rows; the others are random."""
    rs = np.random.RandomState(seed)
    perm = rs.permutation(m)
    col_rows = []
    pos = 0
    while pos < m:
        size = int(rs.randint(1, max(2, int(2 * density * m) + 1)))
        col_rows.append(np.sort(perm[pos : pos + size]))
        pos += size
    if len(col_rows) > n:
        raise ValueError("n too small for the planted partition")
    while len(col_rows) < n:
        mask = rs.random_sample(m) < density
        if not mask.any():
            mask[rs.randint(0, m)] = True
        col_rows.append(np.flatnonzero(mask))
    order = rs.permutation(n)  # hide the planted columns
    rows = [[] for _ in range(m)]
    for newj, oldj in enumerate(order):
        for k in col_rows[oldj]:
            rows[k].append(newj)
    rows = [np.array(sorted(r), dtype=np.int64) for r in rows]
    row_ptr, col = _rows_to_csr(rows, n)
    c = _costs(rs, n, cost_kind)
    return Instance(
        m=m,
        n=n,
        row_ptr=row_ptr,
        col=col,
        factor=np.ones(col.shape[0], np.int32),
        cmin=np.ones(m, np.int32),
        cmax=np.ones(m, np.int32),
        obj_var=np.arange(n, dtype=np.int32),
        obj_factor=c,
        maximize=False,
        kind=0,
        name=f"spp-{m}x{n}-d{density}-{cost_kind}-s{seed}",
    )


def make_pm1(m, n, density, cost_kind="int", seed=3, equalities=False, neg=0.3) -> Instance:
    """Covering rows with each coefficient negated w.p. `neg` and the
    right-hand side shifted accordingly (the +/-1 solvers)."""
    rs = np.random.RandomState(seed)
    rows = _random_support(rs, m, n, density)
    row_ptr, col = _rows_to_csr(rows, n)
    factor = np.where(rs.random_sample(col.shape[0]) < neg, -1, 1).astype(np.int32)
    cmin = np.zeros(m, np.int32)
    cmax = np.zeros(m, np.int32)
    planted = rs.random_sample(n) < 0.1  # equalities: rhs of a planted point
    for k in range(m):
        f = factor[row_ptr[k] : row_ptr[k + 1]]
        nneg = int((f < 0).sum())
        if equalities:
            cols = col[row_ptr[k] : row_ptr[k + 1]]
            cmin[k] = cmax[k] = int((f * planted[cols]).sum())
        else:
            cmin[k] = 1 - nneg  # clause: sum_pos x + sum_neg (1 - x) >= 1
            cmax[k] = INT_MAX
    c = _costs(rs, n, cost_kind)
    return Instance(
        m=m,
        n=n,
        row_ptr=row_ptr,
        col=col,
        factor=factor,
        cmin=cmin,
        cmax=cmax,
        obj_var=np.arange(n, dtype=np.int32),
        obj_factor=c,
        maximize=False,
        kind=2 if equalities else 3,
        name=f"pm1-{m}x{n}-d{density}-{cost_kind}-s{seed}",
    )


def make_z(m, n, density, cost_kind="int", seed=5, kmax=5, short_rows=0.2) -> Instance:
    """Knapsack-like rows sum(a.x) <= b, a uniform in 1..kmax, b = half the
    row weight; `short_rows` of the rows have fewer than 10 elements
    (exhaustive sub-solver), the rest use branch-and-bound / linear.  The
    objective is maximised (profits), telebus-like general coefficients."""
    rs = np.random.RandomState(seed)
    rows = []
    for k in range(m):
        if rs.random_sample() < short_rows:
            size = int(rs.randint(2, 10))
            rows.append(np.sort(rs.choice(n, size=size, replace=False)))
        else:
            mask = rs.random_sample(n) < density
            if mask.sum() < 4:
                mask[rs.choice(n, size=4, replace=False)] = True
            rows.append(np.flatnonzero(mask))
    row_ptr, col = _rows_to_csr(rows, n)
    factor = rs.randint(1, kmax + 1, size=col.shape[0]).astype(np.int32)
    cmin = np.full(m, INT_MIN, np.int32)
    cmax = np.zeros(m, np.int32)
    for k in range(m):
        cmax[k] = max(1, int(factor[row_ptr[k] : row_ptr[k + 1]].sum()) // 2)
    c = _costs(rs, n, cost_kind)
    return Instance(
        m=m,
        n=n,
        row_ptr=row_ptr,
        col=col,
        factor=factor,
        cmin=cmin,
        cmax=cmax,
        obj_var=np.arange(n, dtype=np.int32),
        obj_factor=c,
        maximize=True,
        kind=4,
        name=f"z-{m}x{n}-d{density}-{cost_kind}-s{seed}",
    )


def make_zmix(m, n, cost_kind="int", seed=6, maxlen=14, maximize=True, kmax=5) -> Instance:
    """Small rows covering every sub-solver choice of the Z solver
    (itm-solver-inequalities-Z.cpp:125-187): -1/+1 rows (linear scan), rows
    with general coefficients bounded on both sides (branch-and-bound from 10
    elements, exhaustive below), equalities, and one-sided rows (the
    `unconstrained` variants).  Every row has a feasible 0/1 assignment."""
    rs = np.random.RandomState(seed)
    rows, facs, cmin, cmax = [], [], [], []
    for k in range(m):
        t = k % 6
        if t in (2, 5):
            size = int(rs.randint(3, 10))
        elif t == 0:
            size = int(rs.randint(6, 21))
        else:
            size = int(rs.randint(10, maxlen + 1))
        cols = np.sort(rs.choice(n, size=size, replace=False))
        if t == 0:
            f = rs.choice([-1, 1], size=size, p=[0.3, 0.7]).astype(np.int32)
        else:
            f = rs.randint(1, kmax + 1, size=size).astype(np.int32)
            f[rs.random_sample(size) < 0.2] *= -1
            if np.abs(f).max() < 2:
                f[0] = 2 if f[0] > 0 else -2
        lower = int(f[f < 0].sum())
        upper = int(f[f > 0].sum())
        pick = rs.random_sample(size) < 0.5
        ssum = int(f[pick].sum())  # reachable activity
        if t == 0:
            lo, hi = INT_MIN, max(ssum, lower + 1)
        elif t in (1, 2):
            lo, hi = ssum - 1, ssum + 2
            lo = max(lo, lower + 1)
            hi = min(hi, upper - 1)
            if lo > hi or not (lo <= ssum <= hi):
                lo = hi = ssum
        elif t == 3:
            lo = hi = ssum
        else:
            lo, hi = INT_MIN, max(ssum, lower)
        rows.append(cols)
        facs.append(f)
        cmin.append(lo)
        cmax.append(hi)
    row_ptr, col = _rows_to_csr(rows, n)
    c = _costs(rs, n, cost_kind)
    return Instance(
        m=m,
        n=n,
        row_ptr=row_ptr,
        col=col,
        factor=np.concatenate(facs).astype(np.int32),
        cmin=np.asarray(cmin, np.int64).clip(INT_MIN, None).astype(np.int32),
        cmax=np.asarray(cmax, np.int32),
        obj_var=np.arange(n, dtype=np.int32),
        obj_factor=c,
        maximize=maximize,
        kind=4,
        name=f"zmix-{m}x{n}-{cost_kind}-s{seed}",
    )


# BASELINE.json configs: C1 scp41-shaped, C2 scpnrh1-shaped,
# C3 aa04-shaped.
def config_c1(cost_kind="int"):
    return make_scp(200, 1000, 0.02, cost_kind, seed=41)


def config_c5(cost_kind="int"):
    """telebus-shaped stand-in (general integer coefficients, inequalities-Z):
    1000 knapsack rows over 5000 variables, ~50 elements per long row."""
    return make_z(1000, 5000, 0.01, cost_kind, seed=5)


def config_c2(cost_kind="int"):
    return make_scp(1000, 10000, 0.05, cost_kind, seed=1)


def config_c3(cost_kind="int"):
    return make_spp(426, 7195, 0.017, cost_kind, seed=4)


def write_lp(inst: Instance, path: str) -> None:
    """Write an Instance as a CPLEX-LP file (one row per merged constraint:
    `=` when min == max, `>=` / `<=` for the finite side(s))."""
    with open(path, "w") as f:
        f.write("maximize\n" if inst.maximize else "minimize\n")
        terms = [f"{'+' if c >= 0 else '-'} {abs(c):.17g} x{j}" for j, c in zip(inst.obj_var, inst.obj_factor)]
        for i in range(0, len(terms), 8):
            f.write(" " + " ".join(terms[i : i + 8]) + "\n")
        if len(inst.q_a):  # CPLEX-LP quadratic section: the bracket holds twice the factors
            q = []
            for a, b, c in zip(inst.q_a, inst.q_b, inst.q_factor):
                c2 = 2.0 * float(c)
                sign = "+" if c2 >= 0 else "-"
                q.append(f"{sign} {abs(c2):.17g} x{a} ^ 2" if a == b else f"{sign} {abs(c2):.17g} x{a} * x{b}")
            f.write(" + [ " + " ".join(q) + " ] / 2\n")
        f.write("subject to\n")
        for k in range(inst.m):
            s = slice(inst.row_ptr[k], inst.row_ptr[k + 1])
            lhs = " ".join(f"{'+' if a >= 0 else '-'} {abs(int(a))} x{j}" if abs(a) != 1
                           else f"{'+' if a > 0 else '-'} x{j}"
                           for a, j in zip(inst.factor[s], inst.col[s]))
            lo, hi = int(inst.cmin[k]), int(inst.cmax[k])
            if lo == hi:
                f.write(f"c{k}: {lhs} = {lo}\n")
            else:
                if lo > INT_MIN:
                    f.write(f"c{k}g: {lhs} >= {lo}\n")
                if hi < INT_MAX:
                    f.write(f"c{k}l: {lhs} <= {hi}\n")
        f.write("binary\n")
        names = [f"x{j}" for j in range(inst.n)]
        for i in range(0, inst.n, 16):
            f.write(" " + " ".join(names[i : i + 16]) + "\n")
        f.write("end\n")


def write_scp(inst: Instance, path: str) -> None:
    """Write a set-covering Instance in OR-Library's scp format (`m n`, the
    costs, then per row its length and 1-based column indices)."""
    assert inst.kind == 1 and not inst.maximize and np.all(inst.factor == 1)
    with open(path, "w") as f:
        f.write(f" {inst.m} {inst.n}\n")
        c = inst.dense_costs()
        for i in range(0, inst.n, 12):
            f.write(" " + " ".join(f"{v:.17g}" for v in c[i:i + 12]) + "\n")
        for k in range(inst.m):
            cols = inst.col[inst.row_ptr[k]:inst.row_ptr[k + 1]] + 1
            f.write(f" {len(cols)}\n")
            for i in range(0, len(cols), 12):
                f.write(" " + " ".join(str(int(v)) for v in cols[i:i + 12]) + "\n")
