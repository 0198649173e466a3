"""Shared helpers of the parity tests (oracle side = test infrastructure)."""
import numpy as np


def x_bits(words, n):
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:n]


def pack_bits(bits):
    n = len(bits)
    nwords = n // 64 + 1
    padded = np.zeros(nwords * 64, np.uint8)
    padded[:n] = bits
    return np.packbits(padded, bitorder="little").view(np.uint64).copy()


def random_x(n, p, seed):
    rs = np.random.RandomState(seed)
    return pack_bits((rs.random_sample(n) < p).astype(np.uint8))


def is_feasible(inst, words):
    x = x_bits(words, inst.n).astype(np.int64)
    v = np.add.reduceat(inst.factor.astype(np.int64) * x[inst.col], inst.row_ptr[:-1])
    return bool(np.all((inst.cmin <= v) & (v <= inst.cmax)))


def objective(inst, words):
    x = x_bits(words, inst.n).astype(np.float64)
    return float((inst.dense_costs() * x).sum() + inst.obj_constant)


def bits_equal(a, b):
    """Bitwise equality of two float64 arrays (NaN-safe, -0.0 != +0.0)."""
    a = np.ascontiguousarray(a, np.float64).view(np.uint64)
    b = np.ascontiguousarray(b, np.float64).view(np.uint64)
    return bool(np.array_equal(a, b))


def first_diff(a, b):
    a = np.ascontiguousarray(a, np.float64).view(np.uint64)
    b = np.ascontiguousarray(b, np.float64).view(np.uint64)
    d = np.flatnonzero(a != b)
    return None if d.size == 0 else int(d[0])


# ---------------------------------------------------------------------------
# context_set_solver_parameters, lib/src/private.cpp:50-172, restated for the
# tests that run the ORACLE with "the solver_parameters a reference test sets":
# the oracle takes parameters as the solver sees them (after the range checks).
REFERENCE_DEFAULTS = dict(  # lib/include/baryonyx/core:410-461
    time_limit=-1.0, theta=0.5, delta=-1.0, kappa_min=0.0, kappa_step=1.0e-3, kappa_max=0.6, alpha=1.0,
    pushing_k_factor=0.9, pushing_objective_amplifier=5.0, init_policy_random=0.5, w=0.05, seed=0,
    thread=0, limit=1000, pushes_limit=100, pushing_iteration_limit=50, init_population_size=100,
    order=0, init_policy=0, cost_norm=4,
)
LONG_MAX = 2 ** 63 - 1


def validated_params(**raw):
    """What ctx->parameters holds after context_set_solver_parameters(ctx, params)
    with `params` = the defaults overwritten by `raw`."""
    import math
    p = dict(REFERENCE_DEFAULTS)
    p.update(raw)
    q = dict(REFERENCE_DEFAULTS)
    q["time_limit"] = math.inf if p["time_limit"] <= 0 else p["time_limit"]
    if 0 <= p["theta"] <= 1:
        q["theta"] = p["theta"]
    if p["delta"] == -1 or p["delta"] >= 0:
        q["delta"] = p["delta"]
    # :63-69 -- the whole kappa triple is kept or dropped together
    if (p["kappa_min"] < p["kappa_max"] and 0 <= p["kappa_min"] < 1 and p["kappa_max"] <= 1
            and 0 <= p["kappa_step"] < 1):
        q["kappa_min"], q["kappa_step"], q["kappa_max"] = p["kappa_min"], p["kappa_step"], p["kappa_max"]
    if p["alpha"] >= 0:
        q["alpha"] = p["alpha"]
    for k in ("pushing_k_factor", "pushing_objective_amplifier"):
        if math.isfinite(p[k]):
            q[k] = p[k]
    if 0 <= p["init_policy_random"] <= 1:
        q["init_policy_random"] = p["init_policy_random"]
    q["seed"], q["thread"] = p["seed"], p["thread"]
    q["limit"] = LONG_MAX if p["limit"] <= 0 else p["limit"]
    if p["w"] < 0:
        q["w"] = 0.0
    elif p["limit"] > 0:
        q["w"] = p["limit"] * p["w"] if p["w"] < 1 else min(float(p["limit"]), p["w"])
    else:
        q["w"] = float(math.floor(p["w"]))
    q["pushes_limit"] = max(0, p["pushes_limit"])
    q["pushing_iteration_limit"] = max(0, p["pushing_iteration_limit"])
    if p["init_population_size"] >= 5:
        q["init_population_size"] = p["init_population_size"]
    for k in ("order", "init_policy", "cost_norm"):
        q[k] = p[k]
    return q


def oracle_params(oracle, cap=None, sort_mode=1, **raw):
    """OrcParams for the reference test parameters `raw` (validated as above);
    `cap` bounds an unlimited sweep count (`limit = -1`)."""
    q = validated_params(**raw)
    q.pop("time_limit"), q.pop("thread")
    if cap is not None:
        q["limit"] = min(q["limit"], cap)
    q["limit"] = min(q["limit"], 2 ** 62)
    return oracle.default_params(sort_mode=sort_mode, **q)
