"""ctypes binding of the device layer's C ABI (include/bxh.h).

Python is plumbing here (tests, bench.py, torch.distributed launch): the
product is `baryonyx_amd/csrc/libbxh.so`.  There is no fallback: if the shared
library is missing or HIP cannot run, every call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libbxh.so")

i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)

BXH_OK = 0
BXH_ERR_INVALID = -1
BXH_ERR_OOM = -2
BXH_ERR_UNSUPPORTED = -3
BXH_ERR_DEVICE = -4

# every symbol include/bxh.h declares (tests check the library exports them)
SYMBOLS = [
    "bxh_last_error", "bxh_create", "bxh_create_device", "bxh_destroy", "bxh_set_stream", "bxh_batch",
    "bxh_nwords", "bxh_rmax", "bxh_kernel_form", "bxh_sweep_includes_scan", "bxh_run_supported", "bxh_run", "bxh_get_best_x", "bxh_set_x", "bxh_get_x", "bxh_set_x_all", "bxh_get_x_all", "bxh_set_rng_all", "bxh_pop_create", "bxh_pop_set", "bxh_pop_get", "bxh_pop_store", "bxh_reinit", "bxh_set_pi", "bxh_get_pi",
    "bxh_set_P", "bxh_get_P", "bxh_set_rng", "bxh_get_rng", "bxh_reset", "bxh_reset_all",
    "bxh_order_create", "bxh_order_init", "bxh_order_init_masked", "bxh_get_violated", "bxh_sweep", "bxh_sweep_async",
    "bxh_status_sync", "bxh_row_bytes", "bxh_kernel_time", "bxh_set_objective", "bxh_incumbent",
]


class BxhProblem(C.Structure):
    _fields_ = [
        ("m", C.c_int32),
        ("n", C.c_int32),
        ("nnz", C.c_int64),
        ("row_ptr", i32p),
        ("col", i32p),
        ("factor", i32p),
        ("cmin", i32p),
        ("cmax", i32p),
        ("cost", f64p),
        ("n_quad", C.c_int32),
        ("q_ptr", i32p),
        ("q_id", i32p),
        ("q_factor", f64p),
    ]


class BxhSweepArgs(C.Structure):
    _fields_ = [
        ("kappa", f64p),
        ("delta", C.c_double),
        ("theta", C.c_double),
        ("obj_amp", C.c_double),
        ("is_push", C.c_int32),
        ("order", C.c_int32),
        ("active", u8p),
    ]


class BxhStatus(C.Structure):
    _fields_ = [
        ("remaining", C.c_int32),
        ("pi_changed", C.c_int32),
        ("rng", C.c_uint32),
        ("rows_in_sweep", C.c_uint32),
        ("row_updates", C.c_uint64),
        ("alg_bytes", C.c_double),
    ]


STATUS_DTYPE = np.dtype(
    [
        ("remaining", np.int32),
        ("pi_changed", np.int32),
        ("rng", np.uint32),
        ("rows_in_sweep", np.uint32),
        ("row_updates", np.uint64),
        ("alg_bytes", np.float64),
    ]
)

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
]


def build(force: bool = False) -> str:
    """Compile csrc/bxh_device.hip for gfx950 into csrc/libbxh.so (in-tree, so
    it travels with the repository snapshot)."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "bxh.h"))
    stale = force or not os.path.exists(LIB_PATH)
    if not stale:
        t = os.path.getmtime(LIB_PATH)
        stale = any(os.path.getmtime(s) > t for s in srcs)
    if stale:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call(
            [hipcc, *HIPCC_FLAGS, "-o", LIB_PATH, os.path.join(CSRC, "bxh_device.hip")]
        )
    return LIB_PATH


_LIB = None


def lib():
    """Load libbxh.so; raises if it has not been built (no silent fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
            )
        # A process that also uses torch (device tensors handed to
        # bxh_create_device, torch.distributed) must end up with ONE HIP runtime:
        # torch's wheel bundles its own libamdhip64, and a library loaded before
        # it binds the system one -- whose calls then fail with "no ROCm-capable
        # device" once torch has initialised.  Loading torch first makes libbxh
        # resolve to the runtime torch brought.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        # BXH_LIB: another build of the same C ABI (A/B timing of kernel variants)
        L = C.CDLL(os.environ.get("BXH_LIB") or LIB_PATH)
        L.bxh_last_error.restype = C.c_char_p
        L.bxh_create.argtypes = [
            C.POINTER(BxhProblem), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            C.POINTER(C.c_void_p),
        ]
        L.bxh_create_device.argtypes = L.bxh_create.argtypes
        L.bxh_destroy.argtypes = [C.c_void_p]
        L.bxh_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        for nm in ("bxh_batch", "bxh_nwords", "bxh_rmax", "bxh_kernel_form"):
            getattr(L, nm).argtypes = [C.c_void_p]
        L.bxh_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bxh_get_best_x.argtypes = [C.c_void_p, C.c_int, u64p]
        L.bxh_run_supported.argtypes = [C.c_void_p]
        L.bxh_set_x.argtypes = [C.c_void_p, C.c_int, u64p]
        L.bxh_get_x.argtypes = [C.c_void_p, C.c_int, u64p]
        L.bxh_set_x_all.argtypes = [C.c_void_p, u64p]
        L.bxh_get_x_all.argtypes = [C.c_void_p, u64p]
        L.bxh_set_rng_all.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        for nm in ("bxh_set_pi", "bxh_get_pi", "bxh_set_P", "bxh_get_P"):
            getattr(L, nm).argtypes = [C.c_void_p, C.c_int, f64p]
        L.bxh_set_rng.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        L.bxh_get_rng.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
        L.bxh_reset.argtypes = [C.c_void_p, C.c_int]
        L.bxh_order_init.argtypes = [C.c_void_p, C.c_int, i32p]
        L.bxh_order_create.argtypes = [C.c_void_p, C.c_int]
        L.bxh_reset_all.argtypes = [C.c_void_p]
        L.bxh_order_init_masked.argtypes = [C.c_void_p, C.c_int, u8p, i32p]
        L.bxh_get_violated.argtypes = [C.c_void_p, C.c_int, i32p, i32p]
        L.bxh_sweep.argtypes = [C.c_void_p, C.POINTER(BxhSweepArgs), C.c_void_p]
        L.bxh_sweep_async.argtypes = [C.c_void_p, C.POINTER(BxhSweepArgs)]
        L.bxh_status_sync.argtypes = [C.c_void_p, C.c_void_p]
        L.bxh_row_bytes.argtypes = [C.c_void_p, f64p]
        L.bxh_kernel_time.argtypes = [C.c_void_p, f64p, u64p, C.c_int]
        L.bxh_set_objective.argtypes = [C.c_void_p, f64p, C.c_double, C.c_int]
        L.bxh_incumbent.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


class BxhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"bxh error {code}: {msg}")
        self.code = code


def _check(rc):
    if rc != 0:
        raise BxhError(rc, lib().bxh_last_error().decode())


def _p(a, typ):
    return a.ctypes.data_as(typ)


class RunArgs(C.Structure):
    _fields_ = [("kappa", C.POINTER(C.c_double)), ("delta", C.c_double), ("theta", C.c_double),
                ("order", C.c_int), ("max_sweeps", C.c_int), ("first_i", C.c_int), ("w_limit", C.c_int),
                ("kappa_max", C.c_double), ("kappa_inc", C.POINTER(C.c_double)),
                ("active", C.POINTER(C.c_ubyte))]


INCUMBENT_DTYPE = np.dtype([("key_hi", np.int64), ("key_lo", np.int64), ("value", np.float64),
                            ("remaining", np.int32), ("instance", np.int32)])  # bxh_incumbent_record


RUN_STATUS_DTYPE = np.dtype([("sweeps", np.int32), ("best_remaining", np.int32), ("best_i", np.int32),
                             ("stop", np.int32), ("kappa", np.float64)])


class DeviceBatch:
    """`batch` Solver instances of one constraint matrix on one MI355X.

    Mirrors the reference's Solver concept: `x`, `pi`, `P`,
    `rng` per instance, `order_init` == compute_order::init, `sweep` ==
    compute_order::run / push_and_run for every instance at once.
    """

    def __init__(self, inst, cost, batch, kind=None, float_type=1, maximize=None, device=0, qcost=None):
        self.L = lib()
        self.inst = inst
        self.batch = int(batch)
        self.kind = inst.kind if kind is None else kind
        self.maximize = inst.maximize if maximize is None else maximize
        self._keep = dict(
            row_ptr=np.ascontiguousarray(inst.row_ptr, np.int32),
            col=np.ascontiguousarray(inst.col, np.int32),
            factor=np.ascontiguousarray(inst.factor, np.int32),
            cmin=np.ascontiguousarray(inst.cmin, np.int32),
            cmax=np.ascontiguousarray(inst.cmax, np.int32),
            cost=np.ascontiguousarray(cost, np.float64),
        )
        k = self._keep
        assert k["cost"].shape[0] == inst.n
        nq, qp, qi, qf = 0, None, None, None
        if qcost is not None:  # (q_ptr, q_id, q_factor): quadratic_cost_type's adjacency
            k["q_ptr"] = np.ascontiguousarray(qcost[0], np.int32)
            k["q_id"] = np.ascontiguousarray(qcost[1], np.int32)
            k["q_factor"] = np.ascontiguousarray(qcost[2], np.float64)
            nq = int(k["q_id"].shape[0])
            qp, qi, qf = _p(k["q_ptr"], i32p), _p(k["q_id"], i32p), _p(k["q_factor"], f64p)
        pb = BxhProblem(
            inst.m, inst.n, inst.nnz,
            _p(k["row_ptr"], i32p), _p(k["col"], i32p), _p(k["factor"], i32p),
            _p(k["cmin"], i32p), _p(k["cmax"], i32p), _p(k["cost"], f64p),
            nq, qp, qi, qf,
        )
        h = C.c_void_p()
        _check(self.L.bxh_create(C.byref(pb), self.kind, float_type, int(self.maximize),
                                 self.batch, device, C.byref(h)))
        self._finish(h, inst.m, inst.n, inst.nnz)

    @classmethod
    def on_device(cls, m, n, nnz, row_ptr, col, factor, cmin, cmax, cost, kind, maximize, batch,
                  float_type=1, device=0, keep=None):
        """bxh_create_device: the arguments row_ptr .. cost are DEVICE addresses
        (ints, e.g. torch tensors' data_ptr(); factor may be 0 = all +1); `keep`
        holds whatever owns that memory until the handle has copied it."""
        self = cls.__new__(cls)
        self.L = lib()
        self.inst = None
        self.batch = int(batch)
        self.kind = int(kind)
        self.maximize = bool(maximize)
        cast = lambda a, t: C.cast(C.c_void_p(int(a)), t) if a else None
        pb = BxhProblem(
            int(m), int(n), int(nnz),
            cast(row_ptr, i32p), cast(col, i32p), cast(factor, i32p),
            cast(cmin, i32p), cast(cmax, i32p), cast(cost, f64p),
            0, None, None, None,
        )
        h = C.c_void_p()
        _check(self.L.bxh_create_device(C.byref(pb), self.kind, float_type, int(self.maximize),
                                        self.batch, device, C.byref(h)))
        del keep
        self._finish(h, m, n, nnz)
        return self

    def _finish(self, h, m, n, nnz):
        self.h = h
        self.m, self.n, self.nnz = int(m), int(n), int(nnz)
        self.nwords = self.L.bxh_nwords(self.h)
        self.rmax = self.L.bxh_rmax(self.h)
        self.kernel_form = self.L.bxh_kernel_form(self.h)
        self.sweep_includes_scan = self.L.bxh_sweep_includes_scan(self.h) != 0
        self._status = np.zeros(self.batch, STATUS_DTYPE)

    def close(self):
        if getattr(self, "h", None):
            self.L.bxh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _check(self.L.bxh_set_stream(self.h, C.c_void_p(stream_ptr)))

    def set_x(self, i, words):
        w = np.ascontiguousarray(words, np.uint64)
        assert w.shape[0] == self.nwords
        _check(self.L.bxh_set_x(self.h, i, _p(w, u64p)))

    def set_x_all(self, words):
        w = np.ascontiguousarray(words, np.uint64)
        assert w.shape == (self.batch, self.nwords)
        self._xall = w  # keep alive until the stream has consumed it
        _check(self.L.bxh_set_x_all(self.h, _p(w, u64p)))

    def set_x_all_device(self, device_ptr, keep=None):
        """x of every instance from DEVICE memory ([batch][nwords] u64), stream-ordered."""
        self._xall = keep
        _check(self.L.bxh_set_x_all(self.h, C.cast(C.c_void_p(int(device_ptr)), u64p)))

    def get_x(self, i):
        w = np.zeros(self.nwords, np.uint64)
        _check(self.L.bxh_get_x(self.h, i, _p(w, u64p)))
        return w

    def set_pi(self, i, v):
        v = np.ascontiguousarray(v, np.float64)
        assert v.shape[0] == self.m
        _check(self.L.bxh_set_pi(self.h, i, _p(v, f64p)))

    def get_pi(self, i):
        v = np.zeros(self.m, np.float64)
        _check(self.L.bxh_get_pi(self.h, i, _p(v, f64p)))
        return v

    def set_P(self, i, v):
        v = np.ascontiguousarray(v, np.float64)
        assert v.shape[0] == self.nnz
        _check(self.L.bxh_set_P(self.h, i, _p(v, f64p)))

    def get_P(self, i):
        v = np.zeros(self.nnz, np.float64)
        _check(self.L.bxh_get_P(self.h, i, _p(v, f64p)))
        return v

    def get_x_all(self):
        w = np.zeros((self.batch, self.nwords), np.uint64)
        _check(self.L.bxh_get_x_all(self.h, _p(w, u64p)))
        return w

    def set_objective(self, cost, constant=0.0, maximize=False):
        """The ORIGINAL linear costs (Cost::results sums those), once, for incumbent()."""
        c = np.ascontiguousarray(cost, np.float64)
        assert c.shape == (self.n,)
        _check(self.L.bxh_set_objective(self.h, _p(c, f64p), float(constant), int(bool(maximize))))

    def incumbent(self, rank=0, record_device=None, x_device=None, read_back=True):
        """bxh_incumbent: the batch's best instance (fewest violated rows, then the
        objective), found on the device.  record_device / x_device: device pointers
        (ints) that receive the 32-byte record and the winner's x in stream order;
        read_back: also return the record (one 32-byte transfer), else None."""
        rec = np.zeros(1, INCUMBENT_DTYPE) if read_back else None
        _check(self.L.bxh_incumbent(
            self.h, int(rank), C.c_void_p(int(record_device) if record_device else None),
            C.c_void_p(int(x_device) if x_device else None),
            rec.ctypes.data_as(C.c_void_p) if read_back else None))
        return rec[0] if read_back else None

    def set_rng_all(self, states):
        s = np.ascontiguousarray(states, np.uint32)
        assert s.shape == (self.batch,)
        _check(self.L.bxh_set_rng_all(self.h, s.ctypes.data_as(C.POINTER(C.c_uint32))))

    def set_rng(self, i, s):
        _check(self.L.bxh_set_rng(self.h, i, int(s)))

    def get_rng(self, i):
        s = C.c_uint32()
        _check(self.L.bxh_get_rng(self.h, i, C.byref(s)))
        return s.value

    def reset(self, i):
        _check(self.L.bxh_reset(self.h, i))

    def reset_all(self):
        """Solver::reset() of every instance (P = pi = 0), stream-ordered."""
        _check(self.L.bxh_reset_all(self.h))

    def order_create(self, order):
        """compute_order's constructor for every instance."""
        _check(self.L.bxh_order_create(self.h, int(order)))

    def order_init(self, order=0, sync=True, active=None):
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        actp = _p(act, u8p) if act is not None else None
        if not sync:
            _check(self.L.bxh_order_init_masked(self.h, order, actp, None))
            return None
        rem = np.zeros(self.batch, np.int32)
        _check(self.L.bxh_order_init_masked(self.h, order, actp, _p(rem, i32p)))
        return rem

    def violated(self, i):
        rows = np.zeros(max(self.m, self.n), np.int32)
        cnt = C.c_int32()
        _check(self.L.bxh_get_violated(self.h, i, _p(rows, i32p), C.byref(cnt)))
        return rows[: cnt.value].copy()

    def _args(self, kappa, delta, theta, is_push, obj_amp, order, active):
        kap = np.ascontiguousarray(np.broadcast_to(np.asarray(kappa, np.float64), (self.batch,)))
        act = None
        if active is not None:
            act = np.ascontiguousarray(active, np.uint8)
        a = BxhSweepArgs(_p(kap, f64p), delta, theta, obj_amp, int(is_push), order,
                         _p(act, u8p) if act is not None else None)
        return a, (kap, act)

    def sweep(self, kappa, delta, theta, is_push=False, obj_amp=0.0, order=0, active=None):
        a, keep = self._args(kappa, delta, theta, is_push, obj_amp, order, active)
        _check(self.L.bxh_sweep(self.h, C.byref(a), self._status.ctypes.data_as(C.c_void_p)))
        return self._status.copy()

    def run(self, kappa, delta, theta, order, max_sweeps, first_i, w_limit, kappa_max, kappa_inc):
        """bxh_run: up to max_sweeps sweeps of solver_functor's loop in one launch;
        returns (status array, run status array)."""
        k = np.ascontiguousarray(np.broadcast_to(np.asarray(kappa, np.float64), (self.batch,)).copy())
        inc = np.ascontiguousarray(kappa_inc, np.float64)
        a = RunArgs(k.ctypes.data_as(C.POINTER(C.c_double)), float(delta), float(theta), int(order),
                    int(max_sweeps), int(first_i), int(w_limit), float(kappa_max),
                    inc.ctypes.data_as(C.POINTER(C.c_double)), None)
        rs = np.zeros(self.batch, RUN_STATUS_DTYPE)
        _check(self.L.bxh_run(self.h, C.byref(a), self._status.ctypes.data_as(C.c_void_p),
                              rs.ctypes.data_as(C.c_void_p)))
        return self._status.copy(), rs

    def run_masked(self, kappa, delta, theta, order, max_sweeps, first_i, w_limit, kappa_max, kappa_inc,
                   active):
        """bxh_run for the instances `active` selects (uint8 per instance)."""
        k = np.ascontiguousarray(np.broadcast_to(np.asarray(kappa, np.float64), (self.batch,)).copy())
        inc = np.ascontiguousarray(kappa_inc, np.float64)
        act = np.ascontiguousarray(active, np.uint8)
        a = RunArgs(k.ctypes.data_as(C.POINTER(C.c_double)), float(delta), float(theta), int(order),
                    int(max_sweeps), int(first_i), int(w_limit), float(kappa_max),
                    inc.ctypes.data_as(C.POINTER(C.c_double)), act.ctypes.data_as(C.POINTER(C.c_ubyte)))
        rs = np.zeros(self.batch, RUN_STATUS_DTYPE)
        _check(self.L.bxh_run(self.h, C.byref(a), self._status.ctypes.data_as(C.c_void_p),
                              rs.ctypes.data_as(C.c_void_p)))
        return self._status.copy(), rs

    def get_best_x(self, i):
        w = np.zeros(self.nwords, np.uint64)
        _check(self.L.bxh_get_best_x(self.h, int(i), w.ctypes.data_as(C.POINTER(C.c_uint64))))
        return w

    def sweep_async(self, kappa, delta, theta, is_push=False, obj_amp=0.0, order=0, active=None):
        a, keep = self._args(kappa, delta, theta, is_push, obj_amp, order, active)
        _check(self.L.bxh_sweep_async(self.h, C.byref(a)))

    def status(self):
        _check(self.L.bxh_status_sync(self.h, self._status.ctypes.data_as(C.c_void_p)))
        return self._status.copy()

    def kernel_time(self, reset=False):
        """(total ms, launches) of the sweep kernels, from HIP events."""
        ms = C.c_double()
        n = C.c_uint64()
        _check(self.L.bxh_kernel_time(self.h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def row_bytes(self):
        b = np.zeros(self.m, np.float64)
        _check(self.L.bxh_row_bytes(self.h, _p(b, f64p)))
        return b
