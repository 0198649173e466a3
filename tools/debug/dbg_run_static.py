import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
os.environ["BXH_GROUPS"] = sys.argv[1]
order = int(sys.argv[2])
from baryonyx_amd import synth, _ffi as bxh
from oracle import oracle_ffi as oracle
from helpers import bits_equal, first_diff
base = synth.make_scp(90, 700, 0.12, "real", 71)
rows = [list(base.col[base.row_ptr[k]:base.row_ptr[k + 1]]) for k in range(base.m)]
extra = base.n
for k in range(12):
    rows[k] = [extra] + rows[k]
row_ptr = np.zeros(base.m + 1, np.int32); row_ptr[1:] = np.cumsum([len(r) for r in rows])
col = np.concatenate([np.asarray(r, np.int32) for r in rows])
inst = synth.Instance(m=base.m, n=base.n + 1, row_ptr=row_ptr, col=col, factor=np.ones(col.shape[0], np.int32), cmin=base.cmin, cmax=base.cmax, obj_var=np.arange(base.n + 1, dtype=np.int32), obj_factor=np.concatenate([base.obj_factor, [3.25]]), maximize=False, kind=1, name="x")
orc = oracle.Solver(inst, float_type=1, sort_mode=1, cost_norm=4, seed=5)
dev = bxh.DeviceBatch(inst, orc.cost(), 1, float_type=1)
print("form", dev.kernel_form)
delta = orc.compute_delta(0.5)
x0 = np.zeros(orc.nwords, np.uint64); x0[extra >> 6] |= np.uint64(1) << np.uint64(extra & 63)
orc.set_x(x0); dev.set_x(0, x0); dev.set_rng(0, orc.rng); orc.order_init(order); dev.order_init(order)
for it in range(4):
    vl = list(orc.violated())
    st = dev.sweep(0.03 * it, delta, 0.5, order=order)
    r = orc.sweep(0.03 * it, delta, 0.5)
    ok = np.array_equal(dev.get_x(0), orc.get_x())
    pd, po = dev.get_pi(0), orc.get_pi()
    print("sweep", it, "list", vl[:5], "..", len(vl), "consecutive", all(b - a == 1 for a, b in zip(vl, vl[1:])), "rem", st["remaining"][0], r, "x", ok, "pi", bits_equal(pd, po), first_diff(pd, po) if not bits_equal(pd, po) else None, "P", bits_equal(dev.get_P(0), orc.get_P()))
