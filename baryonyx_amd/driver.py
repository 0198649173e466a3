"""Host-side helpers of bench.py and the tests (Python plumbing): the
solver_parameters fields the hot path reads, normalize_costs and compute_delta
for a linear objective (lib/src/itm-common.hpp:917-998, :1095-1127).  The
drivers themselves -- solver_functor, optimize_functor, the recorder -- are C++
(baryonyx_amd/host/itm_gpu.cpp)."""
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
