"""Lock-step restart driver over one DeviceBatch (Python plumbing for bench.py
and the tests; the C++ host library mirrors the same control flow).

Every instance follows the reference's per-worker loop
(`optimize_functor::operator()`, lib/src/itm-optimizer-common.hpp:620-752):

    reinit -> up to `limit` sweeps of compute_order::run with the kappa
    schedule (kappa += kappa_step * (remaining/m)^alpha once i > w) ->
    on success `pushes_limit` x (push_and_run + up to
    `pushing_iteration_limit` sweeps) -> next restart,

with P and pi carried over between restarts (`reset()` is never called,
:649-663).  One `step()` advances every instance by exactly one sweep; an
instance whose restart is over is re-initialised before its next sweep, so the
batch stays full.  The evolutionary re-initialisation (`best_solution_recorder`
crossover / mutation, :359-554) is not on this path yet: restarts use the
kappa-improve branch of `reinit` (:538-543) and, once that is exhausted,
restart from the incumbent with `init_kappa_improve_start`.
"""
from __future__ import annotations

import numpy as np

NORM_NONE, NORM_RANDOM, NORM_L1, NORM_L2, NORM_LOO = range(5)


def normalize_costs(c, norm=NORM_LOO, float_type=1):
    """normalize_costs (lib/src/itm-common.hpp:967-998, :1095-1127) for a linear
    objective, in the device float kind; returns float64 holding those values."""
    ft = np.float32 if float_type == 0 else np.float64
    c = np.asarray(c, np.float64).astype(ft)
    if norm == NORM_NONE:
        return c.astype(np.float64)
    if norm == NORM_L1:
        div = ft(0)
        for v in np.abs(c):  # sequential sum, as the reference
            div = ft(div + v)
    elif norm == NORM_L2:
        div = ft(0)
        for v in c:
            div = ft(div + ft(v * v))
    elif norm == NORM_LOO:
        div = c.max()
    else:
        raise NotImplementedError("random cost norm needs the worker's std engine")
    if np.isfinite(div) and div != 0 and abs(div) >= np.finfo(ft).tiny:
        c = (c / div).astype(ft)
    return c.astype(np.float64)


def compute_delta(norm_cost, theta, float_type=1):
    """compute_delta, lib/src/itm-common.hpp:917-933."""
    ft = np.float32 if float_type == 0 else np.float64
    c = np.asarray(norm_cost).astype(ft)
    nz = np.abs(c[c != 0])
    mini = nz.min() if nz.size else np.finfo(ft).max
    theta = ft(theta)
    return float(ft(mini - ft(theta * mini)))


class Params:
    """The solver_parameters fields this path reads (core:410-461 defaults)."""

    def __init__(self, **kw):
        self.theta = 0.5
        self.delta = -1.0
        self.kappa_min = 0.0
        self.kappa_step = 1e-3
        self.kappa_max = 0.6
        self.alpha = 1.0
        self.pushing_k_factor = 0.9
        self.pushing_objective_amplifier = 5.0
        self.init_kappa_improve_start = 0.0
        self.init_kappa_improve_increase = 0.02
        self.init_kappa_improve_stop = 0.2
        self.w = 50.0  # context_set_solver_parameters: w < 1 -> w * limit
        self.limit = 1000
        self.pushes_limit = 100
        self.pushing_iteration_limit = 50
        self.order = 0
        self.cost_norm = NORM_LOO
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


MAIN, PUSH, PUSH_ITER = 0, 1, 2


class RestartPool:
    def __init__(self, dev, inst, params: Params, float_type=1, seed=1):
        self.dev = dev
        self.inst = inst
        self.p = params
        self.ft = np.float32 if float_type == 0 else np.float64
        self.B = dev.batch
        B = self.B
        self.m = inst.m
        self.kappa = np.zeros(B, np.float64)
        self.kappa_start = np.zeros(B, np.float64)
        self.kappa_append = np.zeros(B, np.float64)
        self.phase = np.full(B, MAIN, np.int32)
        self.i = np.zeros(B, np.int64)  # sweep index inside the current loop
        self.push = np.zeros(B, np.int32)
        self.need_reinit = np.ones(B, bool)
        self.best_value = np.full(B, np.inf if not inst.maximize else -np.inf)
        self.best_remaining = np.full(B, np.iinfo(np.int32).max, np.int64)
        self.restarts = np.zeros(B, np.int64)
        self.sweeps = 0
        self.costs = inst.dense_costs()
        self.delta = None
        for b in range(B):  # one std engine per worker, distinct seeds
            dev.set_rng(b, 1 + (seed * 7919 + b * 104729) % 2147483645)

    def set_delta(self, delta):
        self.delta = float(delta)

    def _reinit(self, b):
        p = self.p
        # best_solution_recorder::reinit, itm-optimizer-common.hpp:528-554
        if self.kappa_append[b] < p.init_kappa_improve_stop:
            self.kappa_append[b] += p.init_kappa_improve_increase
        else:
            self.kappa_append[b] = p.init_kappa_improve_start
        k = p.kappa_min + (p.kappa_max - p.kappa_min) * self.kappa_append[b]
        self.kappa_start[b] = float(self.ft(k))
        self.kappa[b] = self.kappa_start[b]
        self.phase[b] = MAIN
        self.i[b] = 0
        self.push[b] = 0
        self.need_reinit[b] = False
        self.restarts[b] += 1

    def _kappa_step(self, b, remaining):
        p, ft = self.p, self.ft
        frac = ft(ft(remaining) / ft(self.m))
        inc = ft(ft(p.kappa_step) * ft(np.power(frac, ft(p.alpha))))
        self.kappa[b] = float(ft(ft(self.kappa[b]) + inc))

    def objective(self, b):
        from .util import x_bits

        x = x_bits(self.dev.get_x(b), self.inst.n).astype(np.float64)
        return float((self.costs * x).sum() + self.inst.obj_constant)

    def step(self, record_objective=True):
        """One sweep of every instance (push sweeps for those in a push phase
        are a second launch with is_push=1)."""
        p = self.p
        B = self.B
        for b in range(B):
            if self.need_reinit[b]:
                self._reinit(b)

        is_push = self.phase == PUSH
        total_rows = 0
        statuses = np.zeros(B, dtype=self.dev._status.dtype)
        for push_flag in (False, True):
            act = (is_push == push_flag).astype(np.uint8)
            if not act.any():
                continue
            kap = np.where(is_push, np.float64(self.ft(p.pushing_k_factor)), self.kappa)
            st = self.dev.sweep(
                kap, self.delta, p.theta, is_push=push_flag,
                obj_amp=p.pushing_objective_amplifier if push_flag else 0.0,
                order=p.order, active=act,
            )
            sel = act.astype(bool)
            statuses[sel] = st[sel]
            total_rows += int(st["rows_in_sweep"][sel].sum())
        self.sweeps += 1

        rem = statuses["remaining"]
        for b in range(B):
            r = int(rem[b])
            ph = self.phase[b]
            if r == 0 and record_objective:
                v = self.objective(b)
                better = v > self.best_value[b] if self.inst.maximize else v < self.best_value[b]
                if better:
                    self.best_value[b] = v
            if ph == MAIN:
                if r == 0:
                    self.best_remaining[b] = 0
                    if p.pushes_limit > 0:
                        self.phase[b] = PUSH
                        self.push[b] = 0
                    else:
                        self.need_reinit[b] = True
                    continue
                self.best_remaining[b] = min(self.best_remaining[b], r)
                if self.i[b] > int(p.w):
                    self._kappa_step(b, r)
                self.i[b] += 1
                if self.kappa[b] > p.kappa_max or self.i[b] >= p.limit:
                    self.need_reinit[b] = True  # try_advance + continue
            elif ph == PUSH:
                # after push_and_run: kappa = kappa_start, then the iteration loop
                self.kappa[b] = self.kappa_start[b]
                self.i[b] = 0
                if p.pushing_iteration_limit > 0:
                    self.phase[b] = PUSH_ITER
                else:
                    self._next_push(b)
            else:  # PUSH_ITER
                if r == 0:
                    self._next_push(b)
                    continue
                if self.i[b] > p.w:
                    self._kappa_step(b, r)
                self.i[b] += 1
                if self.kappa[b] > p.kappa_max or self.i[b] >= p.pushing_iteration_limit:
                    self._next_push(b)
        return total_rows, statuses

    def _next_push(self, b):
        self.push[b] += 1
        if self.push[b] >= self.p.pushes_limit:
            self.need_reinit[b] = True
        else:
            self.phase[b] = PUSH
