"""CPU: the oracle's equalities-101 row update checked by hand, row by row, on
small4.lp -- the instance on which the restatement does not reach what
lib/test/optimize.cpp:72-79 expects (tests/test_reference_held.py).

Every row update of the first sweeps is recomputed here in plain Python floats
(IEEE doubles, no contraction) straight from the reference's text:

  compute_reduced_costs   itm-solver-equalities-101.cpp:161-195
  select_variables        itm-solver-equalities-101.cpp:197-206
  affect                  itm-common.hpp:382-467
  decrease_preference     itm-solver-equalities-101.cpp:150-156

and compared with the oracle's pi, P (bit for bit) and x after the same
update.  The one thing not recomputed is the order calculator_sort leaves
inside a run of EQUAL reduced costs (itm-common.hpp:139-147: unstable sort +
shuffle): pi_k and d depend on the sorted VALUES only, so the check takes the
set of selected variables from the oracle's x and verifies that it is a legal
outcome (selected + 1 of them, none larger than an unselected one)."""
import os

import numpy as np

from baryonyx_amd import host_ffi as H
from helpers import bits_equal, x_bits

LP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lp")


class HandSolver:
    """P in CSR-element order, pi per row, x as 0/1 array."""

    def __init__(self, inst, cost):
        self.inst, self.c = inst, cost
        self.P = np.zeros(inst.nnz)
        self.pi = np.zeros(inst.m)
        self.x = np.zeros(inst.n, np.int64)
        # CSC in the reference's order: a stable sort of the CSR elements by column
        rows = np.repeat(np.arange(inst.m), np.diff(inst.row_ptr))
        order = np.argsort(inst.col, kind="stable")
        self.col_elems = [[] for _ in range(inst.n)]
        for e in order:
            self.col_elems[inst.col[e]].append((int(e), int(rows[e])))

    def reduced_costs(self, k):
        inst = self.inst
        out = []
        for e in range(inst.row_ptr[k], inst.row_ptr[k + 1]):
            sum_a_pi = 0.0
            sum_a_p = 0.0
            for (he, hrow) in self.col_elems[inst.col[e]]:
                a = float(inst.factor[he])
                sum_a_pi += a * self.pi[hrow]
                sum_a_p += a * self.P[he]
            v = self.c[inst.col[e]] - sum_a_pi - sum_a_p
            f = int(inst.factor[e])
            if f < 0:
                v = -v
            out.append((v, e, f))
        return out

    def update_row(self, k, kappa, delta, theta, x_after):
        """Returns (selected, d).  x_after: the oracle's x after this update."""
        inst = self.inst
        for e in range(inst.row_ptr[k], inst.row_ptr[k + 1]):
            self.P[e] *= theta
        R = self.reduced_costs(k)
        r_size = len(R)
        c_size = sum(1 for (_, _, f) in R if f < 0)
        assert inst.cmin[k] == inst.cmax[k]
        selected = min(int(inst.cmin[k]) + c_size, r_size) - 1
        values = sorted(v for (v, _, _) in R)
        # membership from the oracle's x: "up" = 1 for a > 0, 0 for a < 0
        up = [(x_after[inst.col[e]] == 1) == (f > 0) for (_, e, f) in R]
        assert sum(up) == selected + 1
        if 0 < sum(up) < r_size:
            assert max(v for (v, _, _), u in zip(R, up) if u) <= min(v for (v, _, _), u in zip(R, up) if not u)
        d = delta
        if selected < 0:
            d += (kappa / (1.0 - kappa)) * (values[0] / 2.0)
        elif selected + 1 >= r_size:
            d += (kappa / (1.0 - kappa)) * (values[selected] * 1.5)
        else:
            self.pi[k] += (values[selected] + values[selected + 1]) / 2.0
            d += (kappa / (1.0 - kappa)) * (values[selected + 1] - values[selected])
        for (_, e, f), u in zip(R, up):
            j = inst.col[e]
            if u:  # itm-common.hpp:445-455: selected entries
                if f < 0:
                    self.x[j] = 0
                    self.P[e] -= d
                else:
                    self.x[j] = 1
                    self.P[e] += d
            else:  # :457-467
                if f < 0:
                    self.x[j] = 1
                    self.P[e] += d
                else:
                    self.x[j] = 0
                    self.P[e] -= d
        return selected, d


def test_small4_first_sweeps_by_hand(oracle):
    inst = H.merged(os.path.join(LP, "small4.lp"))
    assert inst.kind == 2  # equalities-101
    orc = oracle.Solver(inst, float_type=1, sort_mode=0, cost_norm=4, seed=1)  # loo norm, the default
    hand = HandSolver(inst, orc.cost())
    assert hand.c.max() == 1.0 and np.all(hand.c[256:] == 0.0)  # the 16 x variables cost nothing
    orc.set_x(np.zeros(orc.nwords, np.uint64))  # the Bastert point: every cost is >= 0
    orc.order_init(0)
    theta, delta = 0.5, 0.2
    kappa = 0.0
    checked = 0
    negated_selected = 0
    for it in range(30):
        rows = orc.violated().copy()
        if it == 0:
            assert list(rows) == list(range(8))  # the eight `sum x = 1` rows
        for k in rows:
            orc.update_rows([int(k)], kappa, delta, theta)
            x_after = x_bits(orc.get_x(), inst.n).astype(np.int64)
            sel, d = hand.update_row(int(k), kappa, delta, theta, x_after)
            assert np.array_equal(hand.x, x_after)
            assert bits_equal(hand.pi, orc.get_pi()), (it, k)
            assert bits_equal(hand.P, orc.get_P()), (it, k)
            # the updated row is satisfied (affect's bx_expects, itm-common.hpp:464)
            e0, e1 = inst.row_ptr[k], inst.row_ptr[k + 1]
            assert inst.cmin[k] == int((inst.factor[e0:e1] * x_after[inst.col[e0:e1]]).sum())
            checked += 1
            if k >= 8 and x_after[inst.col[e1 - 1]] == 0:
                negated_selected += 1  # the row chose x_ij = 0 (the negated variable "up")
        orc.order_init(0)  # compute_violated_constraints, as compute_order::run ends (itm-common.hpp:871-877)
        if it > 20:
            kappa += 1e-3
    assert checked > 500 and negated_selected > 0
    # after 30 sweeps the rows still do not close: the y rows undo the x rows
    assert len(orc.violated()) > 0


def test_preference_of_a_negated_entry_works_against_the_rows_choice(oracle):
    """The reference's text, evaluated on one row `y1 + y2 - x = 0` (b = 0,
    c_size = 1): when the row selects the negated variable (x = 0), affect does
    `P -= d` on its entry (itm-common.hpp:449-451); compute_reduced_costs forms
    r_x = c - a*pi - a*P with a = -1 and negates it, so the next reduced cost of
    the negated variable is HIGHER by d -- it is less likely to be selected again.
    For a +1 entry the same step lowers it by d (reinforces).  This is the
    mechanism behind small4.lp's cycling; it is the reference's behaviour and the
    oracle (and the HIP path) keep it."""
    from baryonyx_amd.synth import Instance
    inst = Instance(m=1, n=3, row_ptr=np.array([0, 3], np.int32), col=np.array([0, 1, 2], np.int32),
                    factor=np.array([1, 1, -1], np.int32), cmin=np.array([0], np.int32),
                    cmax=np.array([0], np.int32), obj_var=np.array([0, 1, 2], np.int32),
                    obj_factor=np.array([3.0, 5.0, 1.0]), kind=2)
    orc = oracle.Solver(inst, float_type=1, sort_mode=1, cost_norm=0, seed=1)
    orc.set_x(np.array([0b001], np.uint64))  # y1 = 1: the row is violated
    delta = 0.25
    # R = {3, 5, -1 (negated x)}: selected = min(0 + 1, 3) - 1 = 0 -> the negated variable
    orc.update_rows([0], 0.0, delta, 1.0)
    assert list(x_bits(orc.get_x(), 3)) == [0, 0, 0]
    P = orc.get_P()
    assert list(P) == [-delta, -delta, -delta]  # selected negated entry: P -= d as well
    pi = orc.get_pi()[0]
    assert pi == (-1.0 + 3.0) / 2.0
    r_x_next = -(1.0 - (-1.0) * pi - (-1.0) * P[2])  # what the next visit computes for x
    r_x_without_P = -(1.0 + pi)
    assert r_x_next == r_x_without_P + delta  # moved AWAY from being selected
    r_y1_next = 3.0 - pi - P[0]
    assert r_y1_next == (3.0 - pi) + delta  # unselected +1 entry: pushed away too, as intended
