"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; baryonyx_amd/ (the product) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)
u64p = C.POINTER(C.c_uint64)


class OrcProblem(C.Structure):
    _fields_ = [
        ("m", C.c_int32),
        ("n", C.c_int32),
        ("nnz", C.c_int64),
        ("row_ptr", i32p),
        ("col", i32p),
        ("factor", i32p),
        ("cmin", i32p),
        ("cmax", i32p),
        ("n_obj", C.c_int32),
        ("obj_var", i32p),
        ("obj_factor", f64p),
        ("n_qobj", C.c_int32),
        ("q_a", i32p),
        ("q_b", i32p),
        ("q_factor", f64p),
        ("obj_constant", C.c_double),
    ]


class OrcParams(C.Structure):
    _fields_ = [
        (name, C.c_double)
        for name in (
            "theta delta kappa_min kappa_step kappa_max alpha pushing_k_factor "
            "pushing_objective_amplifier init_policy_random "
            "init_crossover_bastert_insertion "
            "init_crossover_solution_selection_mean "
            "init_crossover_solution_selection_stddev "
            "init_mutation_variable_mean init_mutation_variable_stddev "
            "init_mutation_value_mean init_mutation_value_stddev "
            "init_kappa_improve_start init_kappa_improve_increase "
            "init_kappa_improve_stop w"
        ).split()
    ] + [
        ("limit", C.c_int64),
        ("seed", C.c_uint32),
        ("pushes_limit", C.c_int32),
        ("pushing_iteration_limit", C.c_int32),
        ("init_population_size", C.c_int32),
        ("order", C.c_int32),
        ("init_policy", C.c_int32),
        ("cost_norm", C.c_int32),
        ("sort_mode", C.c_int32),
    ]


class OrcResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("remaining_constraints", C.c_int32),
        ("has_solution_vector", C.c_int32),
        ("contract_violations", C.c_int32),
        ("value", C.c_double),
        ("loop", C.c_int64),
        ("row_updates", C.c_uint64),
        ("sweeps", C.c_uint64),
    ]


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so with the committed Makefile."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = ["oracle_capi.cpp", "oracle_capi.h", "itm_oracle.hpp", "itm_oracle_z.hpp", "itm_oracle_opt.hpp"]
    stale = force or not os.path.exists(so)
    if not stale:
        t = os.path.getmtime(so)
        stale = any(os.path.getmtime(os.path.join(_HERE, s)) > t for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_solver_create.restype = C.c_void_p
        L.orc_solver_create.argtypes = [
            C.POINTER(OrcProblem), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
        ]
        L.orc_solver_destroy.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        for name in ("orc_solver_nwords", "orc_solver_rmax"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int
        L.orc_solver_compute_delta.argtypes = [C.c_void_p, C.c_double]
        L.orc_solver_compute_delta.restype = C.c_double
        L.orc_solver_get_cost.argtypes = [C.c_void_p, f64p]
        L.orc_solver_reset.argtypes = [C.c_void_p]
        L.orc_solver_reset.restype = None
        L.orc_solver_get_qcost.argtypes = [C.c_void_p, i32p, i32p, f64p]
        L.orc_solver_get_qcost.restype = C.c_int
        L.orc_solver_get_x.argtypes = [C.c_void_p, u64p]
        L.orc_solver_set_x.argtypes = [C.c_void_p, u64p]
        L.orc_solver_init_x.argtypes = [C.c_void_p, C.c_int, C.c_double]
        for name in ("get_pi", "set_pi", "get_P", "set_P"):
            getattr(L, "orc_solver_" + name).argtypes = [C.c_void_p, f64p]
        L.orc_solver_get_rng.argtypes = [C.c_void_p]
        L.orc_solver_get_rng.restype = C.c_uint32
        L.orc_solver_set_rng.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_solver_row_updates.argtypes = [C.c_void_p]
        L.orc_solver_row_updates.restype = C.c_uint64
        L.orc_solver_tie_rows.argtypes = [C.c_void_p]
        L.orc_solver_tie_rows.restype = C.c_uint64
        L.orc_solver_contract_violations.argtypes = [C.c_void_p]
        L.orc_solver_contract_violations.restype = C.c_uint64
        L.orc_contract_violations_total.argtypes = []
        L.orc_contract_violations_total.restype = C.c_uint64
        L.orc_solver_set_contract_checks.argtypes = [C.c_void_p, C.c_int]
        L.orc_solver_set_contract_checks.restype = None
        L.orc_solver_objective.argtypes = [C.c_void_p]
        L.orc_solver_objective.restype = C.c_double
        L.orc_solver_order_init.argtypes = [C.c_void_p, C.c_int]
        L.orc_solver_violated.argtypes = [C.c_void_p, i32p]
        L.orc_solver_violated.restype = C.c_int
        L.orc_solver_sweep.argtypes = [
            C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double,
        ]
        L.orc_solver_sweep.restype = C.c_int
        L.orc_solver_update_rows.argtypes = [
            C.c_void_p, i32p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double,
        ]
        L.orc_solver_update_rows.restype = C.c_int
        L.orc_solve.argtypes = [
            C.POINTER(OrcProblem), C.c_int, C.c_int, C.c_int,
            C.POINTER(OrcParams), C.POINTER(OrcResult), u64p,
        ]
        L.orc_solve.restype = C.c_int
        L.orc_optimize.argtypes = [
            C.POINTER(OrcProblem), C.c_int, C.c_int, C.c_int, C.POINTER(OrcParams), C.c_int,
            C.c_int64, C.POINTER(OrcResult), i32p, f64p, u64p, u64p,
            C.POINTER(C.c_uint32), C.POINTER(C.c_int64), f64p,
        ]
        L.orc_optimize.restype = C.c_int
        L.orc_default_params.argtypes = [C.POINTER(OrcParams)]
        L.orc_bb_solve.argtypes = [C.c_int, C.c_int, f64p, i32p, i32p, C.c_int, C.c_int]
        L.orc_bb_solve.restype = C.c_int
        L.orc_exhaustive_solve.argtypes = [
            C.c_int, C.c_int, i32p, C.c_int, C.c_int, f64p, i32p, i32p,
        ]
        L.orc_exhaustive_solve.restype = C.c_int
        L.orc_bitarray_set_value.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int]
        L.orc_bitarray_set_value.restype = None
        _LIB = L
    return _LIB


def _p(a, typ):
    return a.ctypes.data_as(typ)


class Problem:
    """Keeps the numpy arrays alive next to the C struct that points at them."""

    def __init__(self, inst):
        self.inst = inst
        self.arrays = dict(
            row_ptr=np.ascontiguousarray(inst.row_ptr, np.int32),
            col=np.ascontiguousarray(inst.col, np.int32),
            factor=np.ascontiguousarray(inst.factor, np.int32),
            cmin=np.ascontiguousarray(inst.cmin, np.int32),
            cmax=np.ascontiguousarray(inst.cmax, np.int32),
            obj_var=np.ascontiguousarray(inst.obj_var, np.int32),
            obj_factor=np.ascontiguousarray(inst.obj_factor, np.float64),
            q_a=np.ascontiguousarray(inst.q_a, np.int32),
            q_b=np.ascontiguousarray(inst.q_b, np.int32),
            q_factor=np.ascontiguousarray(inst.q_factor, np.float64),
        )
        a = self.arrays
        self.c = OrcProblem(
            inst.m, inst.n, inst.nnz,
            _p(a["row_ptr"], i32p), _p(a["col"], i32p), _p(a["factor"], i32p),
            _p(a["cmin"], i32p), _p(a["cmax"], i32p),
            len(a["obj_var"]), _p(a["obj_var"], i32p), _p(a["obj_factor"], f64p),
            len(a["q_a"]), _p(a["q_a"], i32p), _p(a["q_b"], i32p), _p(a["q_factor"], f64p),
            float(inst.obj_constant),
        )


def default_params(**kw) -> OrcParams:
    p = OrcParams()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class Solver:
    """Step-wise oracle Solver + compute_order for one instance."""

    def __init__(self, inst, float_type=1, sort_mode=1, cost_norm=4, seed=1, kind=None, maximize=None):
        self.L = lib()
        self.pb = Problem(inst)
        self.inst = inst
        self.kind = inst.kind if kind is None else kind
        self.maximize = inst.maximize if maximize is None else maximize
        self.h = self.L.orc_solver_create(
            C.byref(self.pb.c), self.kind, float_type, int(self.maximize), sort_mode, cost_norm, seed
        )
        if not self.h:
            raise RuntimeError(self.L.orc_last_error().decode())
        self.nwords = self.L.orc_solver_nwords(self.h)
        self.rmax = self.L.orc_solver_rmax(self.h)

    def close(self):
        if self.h:
            self.L.orc_solver_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compute_delta(self, theta):
        return self.L.orc_solver_compute_delta(self.h, theta)

    def cost(self):
        out = np.zeros(self.inst.n, np.float64)
        self.L.orc_solver_get_cost(self.h, _p(out, f64p))
        return out

    def reset(self):
        """Solver::reset(): P = pi = 0."""
        self.L.orc_solver_reset(self.h)

    def qcost(self):
        """(q_ptr, q_id, q_factor) of the normalised quadratic costs, or None."""
        cnt = self.L.orc_solver_get_qcost(self.h, None, None, None)
        if cnt <= 0:
            return None
        qptr = np.zeros(self.inst.n + 1, np.int32)
        qid = np.zeros(cnt, np.int32)
        qf = np.zeros(cnt, np.float64)
        self.L.orc_solver_get_qcost(self.h, _p(qptr, i32p), _p(qid, i32p), _p(qf, f64p))
        return qptr, qid, qf

    def get_x(self):
        out = np.zeros(self.nwords, np.uint64)
        self.L.orc_solver_get_x(self.h, _p(out, u64p))
        return out

    def set_x(self, words):
        w = np.ascontiguousarray(words, np.uint64)
        assert w.shape[0] == self.nwords
        self.L.orc_solver_set_x(self.h, _p(w, u64p))

    def init_x(self, policy=0, random=0.5):
        self.L.orc_solver_init_x(self.h, policy, random)

    def get_pi(self):
        out = np.zeros(self.inst.m, np.float64)
        self.L.orc_solver_get_pi(self.h, _p(out, f64p))
        return out

    def set_pi(self, v):
        v = np.ascontiguousarray(v, np.float64)
        self.L.orc_solver_set_pi(self.h, _p(v, f64p))

    def get_P(self):
        out = np.zeros(self.inst.nnz, np.float64)
        self.L.orc_solver_get_P(self.h, _p(out, f64p))
        return out

    def set_P(self, v):
        v = np.ascontiguousarray(v, np.float64)
        self.L.orc_solver_set_P(self.h, _p(v, f64p))

    @property
    def rng(self):
        return self.L.orc_solver_get_rng(self.h)

    @rng.setter
    def rng(self, s):
        self.L.orc_solver_set_rng(self.h, s)

    @property
    def row_updates(self):
        return self.L.orc_solver_row_updates(self.h)

    @property
    def tie_rows(self):
        return self.L.orc_solver_tie_rows(self.h)

    @property
    def contract_violations(self):
        """Row updates behind which the reference's default build would have
        stopped (bx_expects(is_valid_constraint), itm-common.hpp:464,
        itm-solver-inequalities-Z.cpp:436): above zero, the run is outside the
        reference's working domain."""
        return self.L.orc_solver_contract_violations(self.h)

    def set_contract_checks(self, on):
        """False: no check and no count (the reference's BARYONYX_FULL_OPTIMIZATION
        build, debug.hpp:103-111) -- for timing."""
        self.L.orc_solver_set_contract_checks(self.h, int(bool(on)))

    def objective(self):
        return self.L.orc_solver_objective(self.h)

    def order_init(self, order=0):
        self.L.orc_solver_order_init(self.h, order)

    def violated(self):
        n = self.L.orc_solver_violated(self.h, None)
        out = np.zeros(max(n, 1), np.int32)
        self.L.orc_solver_violated(self.h, _p(out, i32p))
        return out[:n]

    def sweep(self, kappa, delta, theta, is_push=False, obj_amp=0.0):
        r = self.L.orc_solver_sweep(self.h, kappa, delta, theta, int(is_push), obj_amp)
        if r < 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        return r

    def update_rows(self, rows, kappa, delta, theta, is_push=False, obj_amp=0.0):
        rows = np.ascontiguousarray(rows, np.int32)
        r = self.L.orc_solver_update_rows(
            self.h, _p(rows, i32p), len(rows), kappa, delta, theta, int(is_push), obj_amp
        )
        if r < 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        return bool(r)


def contract_violations_total():
    """The process-wide count of Solver.contract_violations (every Solver, solve
    and optimize so far)."""
    return lib().orc_contract_violations_total()


def solve(inst, float_type=1, params=None, kind=None, maximize=None):
    """Whole solver_functor run; returns (OrcResult, x words)."""
    L = lib()
    pb = Problem(inst)
    p = params or default_params()
    res = OrcResult()
    xw = np.zeros(inst.n // 64 + 1, np.uint64)
    rc = L.orc_solve(
        C.byref(pb.c),
        inst.kind if kind is None else kind,
        float_type,
        int(inst.maximize if maximize is None else maximize),
        C.byref(p),
        C.byref(res),
        _p(xw, u64p),
    )
    if rc != 0:
        raise RuntimeError(L.orc_last_error().decode())
    return res, xw


def optimize(inst, float_type=1, params=None, thread=4, rounds=100, kind=None, maximize=None):
    """optimize_problem for `rounds` rounds of the deterministic schedule (see
    itm_oracle_opt.hpp).  Returns (OrcResult, dict(pop_remaining, pop_value,
    pop_hash, pop_x, worker_rng, worker_calls))."""
    L = lib()
    pb = Problem(inst)
    p = params or default_params()
    res = OrcResult()
    pop = p.init_population_size
    nwords = inst.n // 64 + 1
    out = dict(
        pop_remaining=np.zeros(pop, np.int32), pop_value=np.zeros(pop, np.float64),
        pop_hash=np.zeros(pop, np.uint64), pop_x=np.zeros((pop, nwords), np.uint64),
        worker_rng=np.zeros(thread, np.uint32), worker_calls=np.zeros(thread, np.int64),
    )
    secs = np.zeros(1, np.float64)
    rc = L.orc_optimize(
        C.byref(pb.c), inst.kind if kind is None else kind, float_type,
        int(inst.maximize if maximize is None else maximize), C.byref(p), thread, rounds,
        C.byref(res), _p(out["pop_remaining"], i32p), _p(out["pop_value"], f64p),
        _p(out["pop_hash"], u64p), _p(out["pop_x"], u64p),
        out["worker_rng"].ctypes.data_as(C.POINTER(C.c_uint32)),
        out["worker_calls"].ctypes.data_as(C.POINTER(C.c_int64)), _p(secs, f64p),
    )
    if rc != 0:
        raise RuntimeError(L.orc_last_error().decode())
    out["sweep_seconds"] = float(secs[0])
    return res, out
