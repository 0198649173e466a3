"""Debug aid: the `cycle` order case of tests/test_gpu_parity.py on the narrow
kernel, printing where P departs from the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("BXH_NARROW", "1")
import numpy as np
from baryonyx_amd import synth, _ffi as bxh
from oracle import oracle_ffi as oracle
from helpers import random_x, bits_equal

inst = synth.make_scp(80, 60, 0.15, "int", 48)
float_type, order, x0p, batch = 1, 8, 0.1, 3
theta = 0.5
orcs = [oracle.Solver(inst, float_type=float_type, sort_mode=1, cost_norm=4, seed=1000 + b) for b in range(batch)]
cost = orcs[0].cost(); delta = orcs[0].compute_delta(theta)
dev = bxh.DeviceBatch(inst, cost, batch, float_type=float_type, qcost=orcs[0].qcost())
print("form", dev.kernel_form, "rmax", dev.rmax)
for b, s in enumerate(orcs):
    x0 = random_x(inst.n, x0p, 77 + b)
    s.set_x(x0); dev.set_x(b, x0); dev.set_rng(b, s.rng); s.order_init(order)
dev.order_init(order)
kappa = np.array([0.01 * b for b in range(batch)])
rp = inst.row_ptr
for it in range(8):
    push = it >= 1
    amp = 5.0 if push else 0.0
    Pbefore = [s.get_P().copy() for s in orcs]
    st = dev.sweep(kappa, delta, theta, is_push=push, obj_amp=amp, order=order)
    for b, s in enumerate(orcs):
        r = s.sweep(float(kappa[b]), delta, theta, is_push=push, obj_amp=amp)
        Pd, Po = dev.get_P(b), s.get_P()
        bad = np.nonzero(Pd.view(np.uint64) != Po.view(np.uint64))[0]
        changed = np.nonzero(Pbefore[b].view(np.uint64) != Po.view(np.uint64))[0]
        rows_changed = sorted(set(np.searchsorted(rp, changed, side="right") - 1))
        print(f"sweep {it} inst {b}: rows_in_sweep {st['rows_in_sweep'][b]} rem {st['remaining'][b]}/{r} x_ok {np.array_equal(dev.get_x(b), s.get_x())} "
              f"pi_ok {bits_equal(dev.get_pi(b), s.get_pi())} bad P {len(bad)} rows touched by oracle {rows_changed[:12]}")
        if len(bad):
            rows_bad = sorted(set(np.searchsorted(rp, bad, side="right") - 1))
            print("   bad rows", rows_bad, "first", bad[:6], Pd[bad[:6]], Po[bad[:6]], "before", Pbefore[b][bad[:6]])
    kappa = kappa + 0.02
