"""ctypes binding of the C++ host library (baryonyx_amd/host/libbaryonyx_amd.so),
which exposes baryonyx::make_problem / solve / optimize through a small C
surface (host/c_api.cpp).  Plumbing for tests and bench.py."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST = os.path.join(_HERE, "host")
LIB_PATH = os.path.join(HOST, "libbaryonyx_amd.so")
SOURCES = ["lp_parser.cpp", "problem.cpp", "preprocessor.cpp", "itm_gpu.cpp", "api.cpp", "c_api.cpp",
           "rsurface.cpp"]


class Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "time_limit theta delta kappa_min kappa_step kappa_max alpha pushing_k_factor "
        "pushing_objective_amplifier init_policy_random w").split()] + [
        ("limit", C.c_longlong), ("seed", C.c_uint), ("thread", C.c_int),
        ("pushes_limit", C.c_int), ("pushing_iteration_limit", C.c_int),
        ("init_population_size", C.c_int), ("order", C.c_int), ("float_type", C.c_int),
        ("init_policy", C.c_int), ("cost_norm", C.c_int), ("pre_order", C.c_int),
        ("device", C.c_int), ("sweep_budget", C.c_longlong), ("preprocessor", C.c_int),
        ("pad_", C.c_int)] + [(n, C.c_double) for n in (
            "init_crossover_bastert_insertion init_crossover_solution_selection_mean "
            "init_crossover_solution_selection_stddev init_mutation_variable_mean "
            "init_mutation_variable_stddev init_mutation_value_mean init_mutation_value_stddev "
            "init_kappa_improve_start init_kappa_improve_increase init_kappa_improve_stop").split()]


class ProblemInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "status variables equal greater less nnz maximize problem_type n_objective "
        "n_quadratic n_binary").split()] + [("objective_constant", C.c_double)]


class PreprocessInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "variables affected equal greater less n_objective n_quadratic problem_type").split()] + [
        ("objective_constant", C.c_double)]


class Result(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "status remaining_constraints variables constraints n_solutions valid").split()] + [
        ("loop", C.c_longlong), ("value", C.c_double), ("recomputed_value", C.c_double),
        ("duration", C.c_double)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(HOST, s) for s in SOURCES] + [os.path.join(HOST, "private.hpp")]
    bxh = os.path.join(_HERE, "csrc", "libbxh.so")
    stale = force or not os.path.exists(LIB_PATH)
    if not stale:
        t = os.path.getmtime(LIB_PATH)
        stale = any(os.path.getmtime(s) > t for s in srcs) or os.path.getmtime(bxh) > t
    if stale:
        subprocess.check_call(
            ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-pthread",
             "-I" + os.path.join(_HERE, "..", "include"), "-o", LIB_PATH,
             *[os.path.join(HOST, s) for s in SOURCES],
             "-L" + os.path.join(_HERE, "csrc"), "-lbxh", "-Wl,-rpath,$ORIGIN/../csrc"])
    return LIB_PATH


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        L.bxc_default_params.argtypes = [C.POINTER(Params)]
        L.bxc_parse.argtypes = [C.c_char_p, C.POINTER(ProblemInfo)]
        ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
        L.bxc_variables.argtypes = [C.c_char_p, ip, ip, ip, C.c_int]
        L.bxc_dump_problem.argtypes = [C.c_char_p, C.c_char_p, C.c_int, ip]
        L.bxc_merged.argtypes = [C.c_char_p, C.c_int, C.c_int, ip, ip, ip, ip, ip, ip, ip, ip, ip, ip, dp, dp, ip, ip,
                                 ip, ip, ip, dp]
        L.bxc_preprocess.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(PreprocessInfo), C.c_int, C.c_int,
                                     C.POINTER(PreprocessInfo), C.c_int, C.POINTER(PreprocessInfo),
                                     C.POINTER(PreprocessInfo), dp, ip, dp, ip, ip]
        L.bxc_run.argtypes = [C.c_char_p, C.c_int, C.POINTER(Params), C.POINTER(Result),
                              C.POINTER(C.c_byte), C.c_char_p, C.c_int]
        ullp = C.POINTER(C.c_ulonglong)
        L.bxc_optimize_trace.argtypes = [C.c_char_p, C.POINTER(Params), C.POINTER(Result), C.c_int, C.c_int,
                                         C.c_int, ip, dp, ullp, ullp, C.POINTER(C.c_uint), C.POINTER(C.c_longlong),
                                         C.POINTER(C.c_longlong), C.c_char_p, C.c_int]
        _LIB = L
    return _LIB


class RResult(C.Structure):
    """bxr_result, include/rbaryonyx.h (the named list of rbaryonyx.cpp:280-288)"""
    _fields_ = [("solution_found", C.c_int), ("error_found", C.c_int), ("value", C.c_double),
                ("duration", C.c_double), ("variables", C.c_int), ("constraints", C.c_int),
                ("remaining_constraints", C.c_int), ("minimize", C.c_int), ("n_solutions", C.c_int),
                ("solutions", C.POINTER(C.c_double))]


# the R exports' arguments with their defaults, in order (rbaryonyx.cpp:368-392, :519-551)
R_SOLVE_ARGS = [("limit", int, 1000), ("theta", float, 0.5), ("delta", float, -1.0), ("pre_constraint_order", int, 1),
                ("constraint_order", int, 0), ("kappa_min", float, 0.0), ("kappa_step", float, 1.0e-3),
                ("kappa_max", float, 0.6), ("alpha", float, 1.0), ("w", float, 0.05), ("time_limit", float, 10.0),
                ("seed", int, -1), ("thread", int, 1), ("norm", int, 4), ("pushing_k_factor", float, 0.9),
                ("pushing_objective_amplifier", float, 5.0), ("pushes_limit", int, 100),
                ("pushing_iteration_limit", int, 50), ("init_policy", int, 0), ("init_policy_random", float, 0.5),
                ("float_type", int, 1), ("storage_type", int, 1), ("verbose", int, 1)]
R_OPTIMIZE_ARGS = R_SOLVE_ARGS[:18] + [
    ("init_crossover_bastert_insertion", float, 0.01), ("init_crossover_solution_selection_mean", float, 0.0),
    ("init_crossover_solution_selection_stddev", float, 0.3), ("init_mutation_variable_mean", float, 0.0001),
    ("init_mutation_variable_stddev", float, 0.001), ("init_mutation_value_mean", float, 0.5),
    ("init_mutation_value_stddev", float, 0.2), ("init_kappa_improve_start", float, 0.0),
    ("init_kappa_improve_increase", float, 0.02), ("init_kappa_improve_stop", float, 0.2),
    ("float_type", int, 1), ("storage_type", int, 1), ("verbose", int, 1)]


def _r_call(name, spec, file_path, kw):
    L = lib()
    fn = getattr(L, name)
    fn.argtypes = [C.c_char_p] + [C.c_int if t is int else C.c_double for _, t, _ in spec] + [C.POINTER(RResult)]
    fn.restype = C.c_int
    unknown = set(kw) - {n for n, _, _ in spec}
    if unknown:
        raise TypeError(f"{name}: unknown argument(s) {sorted(unknown)}")
    args = [t(kw.get(n, d)) for n, t, d in spec]
    out = RResult()
    rc = fn(os.fsencode(file_path), *args, C.byref(out))
    if rc != 0:
        raise RuntimeError(f"{name} -> {rc}")
    ret = {f: getattr(out, f) for f, _ in RResult._fields_ if f not in ("solutions", "n_solutions")}
    for f in ("solution_found", "error_found", "minimize"):
        ret[f] = bool(ret[f])
    ret["solutions"] = [out.solutions[i] for i in range(out.n_solutions)]
    L.bxr_free_result.argtypes = [C.POINTER(RResult)]
    L.bxr_free_result(C.byref(out))
    return ret


def solve_01lp_problem(file_path, **kw):
    """rbaryonyx's solve_01lp_problem (rbaryonyx.cpp:368-447) -> its nine-field list as a dict."""
    return _r_call("bxr_solve_01lp_problem", R_SOLVE_ARGS, file_path, kw)


def optimize_01lp_problem(file_path, **kw):
    """rbaryonyx's optimize_01lp_problem (rbaryonyx.cpp:519-618)."""
    return _r_call("bxr_optimize_01lp_problem", R_OPTIMIZE_ARGS, file_path, kw)


def default_params(**kw) -> Params:
    p = Params()
    lib().bxc_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def parse(path) -> ProblemInfo:
    info = ProblemInfo()
    lib().bxc_parse(os.fsencode(path), C.byref(info))
    return info


def dump(path):
    """(status, canonical text) of make_problem on a file: type, constant, variables with
    bounds and types, objective, quadratic terms, the three constraint lists."""
    L = lib()
    need = C.c_int()
    st = L.bxc_dump_problem(os.fsencode(path), None, 0, C.byref(need))
    buf = C.create_string_buffer(need.value + 1)
    L.bxc_dump_problem(os.fsencode(path), buf, need.value + 1, C.byref(need))
    return st, buf.value.decode("latin-1")


def variables(path):
    """[(min, max, variable_type)] of an LP file's variables in order of first
    appearance, as make_problem leaves them; raises with the reader's status."""
    info = parse(path)
    if info.status != 0:
        raise ValueError(info.status)
    n = max(info.variables, 1)
    lo, hi, ty = (np.zeros(n, np.int32) for _ in range(3))
    ip = C.POINTER(C.c_int)
    rc = lib().bxc_variables(os.fsencode(path), lo.ctypes.data_as(ip), hi.ctypes.data_as(ip),
                             ty.ctypes.data_as(ip), n)
    assert rc == 0
    return [(int(lo[i]), int(hi[i]), int(ty[i])) for i in range(info.variables)]


def merged(path, pre_order=0, preprocessor=1):
    """The merged constraints + linear objective of an LP file as a
    baryonyx_amd.synth.Instance (what a Solver constructor receives);
    `preprocessor` as solver_parameters::preprocessor_options (1 = all, the
    default of solve / optimize)."""
    from .synth import Instance

    L = lib()
    m, n, nnz, nobj, mx, pt = (C.c_int() for _ in range(6))
    const = C.c_double()
    nq = C.c_int()
    rc = L.bxc_merged(os.fsencode(path), pre_order, preprocessor, m, n, nnz, nobj, None, None, None, None,
                      None, None, None, C.byref(const), mx, pt, nq, None, None, None)
    if rc != 0:
        raise RuntimeError(f"bxc_merged({path}) -> {rc}")
    row_ptr = np.zeros(m.value + 1, np.int32)
    col = np.zeros(nnz.value, np.int32)
    fac = np.zeros(nnz.value, np.int32)
    cmin = np.zeros(m.value, np.int32)
    cmax = np.zeros(m.value, np.int32)
    ov = np.zeros(nobj.value, np.int32)
    of = np.zeros(nobj.value, np.float64)
    qa, qb = np.zeros(nq.value, np.int32), np.zeros(nq.value, np.int32)
    qf = np.zeros(nq.value, np.float64)
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    rc = L.bxc_merged(os.fsencode(path), pre_order, preprocessor, m, n, nnz, nobj,
                      row_ptr.ctypes.data_as(ip), col.ctypes.data_as(ip), fac.ctypes.data_as(ip),
                      cmin.ctypes.data_as(ip), cmax.ctypes.data_as(ip), ov.ctypes.data_as(ip),
                      of.ctypes.data_as(dp), C.byref(const), mx, pt, nq, qa.ctypes.data_as(ip),
                      qb.ctypes.data_as(ip), qf.ctypes.data_as(dp))
    assert rc == 0
    kind = {0: 0, 1: 2, 2: 4, 3: 1, 4: 3, 5: 4}[pt.value]  # problem_solver_type -> bxh_kind
    return Instance(m=m.value, n=n.value, row_ptr=row_ptr, col=col, factor=fac, cmin=cmin,
                    cmax=cmax, obj_var=ov, obj_factor=of, obj_constant=const.value,
                    maximize=bool(mx.value), kind=kind, name=os.path.basename(path),
                    q_a=qa, q_b=qb, q_factor=qf)


def run(path, mode=0, params=None):
    """baryonyx::solve (mode 0) / optimize (mode 1) on an LP file through the
    C++ host library; returns (Result, x as int8 array)."""
    L = lib()
    p = params or default_params()
    info = parse(path)
    x = np.zeros(max(info.variables, 1), np.int8)
    res = Result()
    err = C.create_string_buffer(256)
    rc = L.bxc_run(os.fsencode(path), mode, C.byref(p), C.byref(res),
                   x.ctypes.data_as(C.POINTER(C.c_byte)), err, 256)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return res, x


def preprocess(path, fix_zero=None, affect=None, split=None):
    """baryonyx::preprocess on an LP file; returns a dict with the reduced
    problem's sizes and objective, plus `affect` (var, value) / `split` (var)
    results when asked (lib/src/preprocessor.cpp:866-913)."""
    L = lib()
    info, aff, s1, s0 = (PreprocessInfo() for _ in range(4))
    raw = parse(path)
    nlin = max(raw.n_objective + raw.n_quadratic, 1)
    lf, lv = np.zeros(nlin, np.float64), np.zeros(nlin, np.int32)
    nq = max(raw.n_quadratic, 1)
    qf, qa, qb = np.zeros(nq, np.float64), np.zeros(nq, np.int32), np.zeros(nq, np.int32)
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    rc = L.bxc_preprocess(os.fsencode(path), None if fix_zero is None else fix_zero.encode(),
                          C.byref(info), -1 if affect is None else affect[0],
                          0 if affect is None else int(affect[1]), C.byref(aff),
                          -1 if split is None else split, C.byref(s1), C.byref(s0),
                          lf.ctypes.data_as(dp), lv.ctypes.data_as(ip), qf.ctypes.data_as(dp),
                          qa.ctypes.data_as(ip), qb.ctypes.data_as(ip))
    if rc != 0:
        raise RuntimeError(f"bxc_preprocess({path}) -> {rc}")
    return dict(info=info, affect=aff, split1=s1, split0=s0,
                linear=[(int(lv[i]), float(lf[i])) for i in range(info.n_objective)],
                quadratic=[(int(qa[i]), int(qb[i]), float(qf[i])) for i in range(info.n_quadratic)])


def last_optimize_timing():
    """(host set-up, device set-up, rounds) of this thread's last optimize(), seconds."""
    out = (C.c_double * 3)()
    lib().bxc_last_optimize_timing(out)
    return float(out[0]), float(out[1]), float(out[2])


def last_optimize_host_rounds():
    """(host seconds inside the rounds of this thread's last optimize() that were not
    spent waiting for the device, rounds)."""
    sec, n = C.c_double(), C.c_longlong()
    lib().bxc_last_optimize_host_rounds(C.byref(sec), C.byref(n))
    return sec.value, n.value


def optimize_trace(path, params, nvars):
    """baryonyx::optimize with the recorder's final state: (Result, dict) with
    the population best first (x over the `nvars` preprocessed variables), every
    worker's engine state and restart count, and the rounds run."""
    L = lib()
    pop, thread = params.init_population_size, params.thread
    if thread <= 0:  # optimize_problem's default pool (itm_gpu.cpp: get_thread_number)
        thread = 64
    nwords = nvars // 64 + 1
    out = dict(
        pop_remaining=np.zeros(pop, np.int32), pop_value=np.zeros(pop, np.float64),
        pop_hash=np.zeros(pop, np.uint64), pop_x=np.zeros((pop, nwords), np.uint64),
        worker_rng=np.zeros(thread, np.uint32), worker_calls=np.zeros(thread, np.int64),
    )
    res = Result()
    rounds = C.c_longlong()
    err = C.create_string_buffer(256)
    ip, dp, ullp = C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_ulonglong)
    rc = L.bxc_optimize_trace(
        os.fsencode(path), C.byref(params), C.byref(res), pop, nwords, thread,
        out["pop_remaining"].ctypes.data_as(ip), out["pop_value"].ctypes.data_as(dp),
        out["pop_hash"].ctypes.data_as(ullp), out["pop_x"].ctypes.data_as(ullp),
        out["worker_rng"].ctypes.data_as(C.POINTER(C.c_uint)),
        out["worker_calls"].ctypes.data_as(C.POINTER(C.c_longlong)), C.byref(rounds), err, 256)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    out["rounds"] = rounds.value
    return res, out
