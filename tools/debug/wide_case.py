"""Diagnostic: one parity case of tests/test_gpu_parity.py on the wide kernel; where the first sweep differs.
    python tools/debug/wide_case.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("BXH_GROUPS", "w1")
import numpy as np
from baryonyx_amd import _ffi as bxh, synth
from oracle import oracle_ffi as oracle
from test_gpu_parity import random_x

def go(tag, float_type, order, x0p=0.1, seed=22, eq=True, batch=3, sweeps=1):
    inst = synth.make_pm1(40, 250, 0.06, "real", seed, equalities=eq)
    theta = 0.5
    orcs = [oracle.Solver(inst, float_type=float_type, sort_mode=1, cost_norm=4, seed=1000 + b, maximize=False)
            for b in range(batch)]
    cost = orcs[0].cost()
    delta = orcs[0].compute_delta(theta)
    dev = bxh.DeviceBatch(inst, cost, batch, float_type=float_type, maximize=False, qcost=orcs[0].qcost())
    for b, s in enumerate(orcs):
        x0 = random_x(inst.n, x0p, 77 + b) if x0p > 0 else np.zeros(s.nwords, np.uint64)
        s.set_x(x0); dev.set_x(b, x0); dev.set_rng(b, s.rng); s.order_init(order)
    dev.order_init(order)
    kappa = np.array([0.0 + 0.01 * (b % 64) for b in range(batch)])
    rp = inst.row_ptr
    for it in range(sweeps):
        rows_before = [s.violated().copy() for s in orcs]
        st = dev.sweep(kappa, delta, theta, is_push=False, obj_amp=0.0, order=order)
        for b, s in enumerate(orcs):
            r = s.sweep(float(kappa[b]), delta, theta, is_push=False, obj_amp=0.0)
            pd, po = dev.get_pi(b), s.get_pi()
            Pd, Po = dev.get_P(b), s.get_P()
            bad_pi = np.nonzero(pd.view(np.uint32 if float_type == 0 else np.uint64) != po.view(np.uint32 if float_type == 0 else np.uint64))[0]
            bad_P = np.nonzero(Pd != Po)[0]
            rows_P = sorted(set(np.searchsorted(rp, bad_P, side="right") - 1))
            xd, xo = dev.get_x(b), s.get_x()
            print(tag, "sweep", it, "inst", b, "remaining", st["remaining"][b], r, "rng", st["rng"][b] == s.rng,
                  "pi rows differ", bad_pi[:8], "P rows differ", rows_P[:8], "x words differ", int((xd != xo).sum()))
            if len(rows_P):
                k = rows_P[0]
                print("   first differing row", k, "r =", rp[k + 1] - rp[k], "position in the sweep's list:",
                      list(rows_before[b]).index(k) if k in rows_before[b] else None, "of", len(rows_before[b]),
                      "bmin/bmax", inst.cmin[k], inst.cmax[k])
                e0, e1 = rp[k], rp[k + 1]
                print("   P dev", np.array2string(Pd[e0:e1], precision=4, max_line_width=250))
                print("   P orc", np.array2string(Po[e0:e1], precision=4, max_line_width=250))
                print("   factor", inst.factor[e0:e1])
        kappa = kappa + 0.02
    import ctypes as C
    L = bxh.lib()
    L.bxh_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    t = np.zeros((batch, 32), np.uint64)
    L.bxh_debug_profile(dev.h, t.ctypes.data_as(C.POINTER(C.c_uint64)), 0)
    for b in range(batch):
        if t[b, 15] == 0:
            continue
        k = int(t[b, 15]) - 1
        r = int(t[b, 14] >> np.uint64(32)); sel = int(np.int32(t[b, 14] & np.uint64(0xffffffff)))
        vals = t[b, :11].view(np.uint32)[:r].view(np.float32)
        ids = t[b, 11:14].view(np.uint8)[:r].astype(int)
        e0 = rp[k]
        fac = inst.factor[e0:e0 + r]
        cols = inst.col[e0:e0 + r]
        want = np.where(fac < 0, -1.0, 1.0) * np.asarray(cost, np.float64)[cols]
        order = np.lexsort((np.arange(r), vals))
        okv = np.array_equal(vals.astype(np.float64), want.astype(np.float32).astype(np.float64))
        print(f"   dump inst {b}: row {k} r {r} sel {sel} values as expected: {okv}; order as sorted: {np.array_equal(order, ids)}")
        if not okv:
            print("      vals", vals); print("      want", want.astype(np.float32))
        if not np.array_equal(order, ids):
            print("      ids ", ids); print("      sort", order); print("      vals", vals)
    dev.close()

B = int(os.environ.get("BX_BATCH", "3"))
go("f32 eq order2", 0, 2, batch=B)
go("f32 eq order2 again", 0, 2, batch=B)
