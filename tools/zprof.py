import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baryonyx_amd import _ffi, synth
from baryonyx_amd.driver import compute_delta, normalize_costs
_ffi.LIB_PATH = os.path.join(_ffi.CSRC, "libbxh_prof.so")
os.environ.pop("BXH_LIB", None)
L = _ffi.lib()
L.bxh_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
inst = synth.config_c5("int"); batch = int(os.environ.get("ZPROF_BATCH", "64"))
cost = normalize_costs(inst.dense_costs(), 4, 1); delta = compute_delta(cost, 0.5, 1)
dev = _ffi.DeviceBatch(inst, cost, batch, float_type=1)
from baryonyx_amd.util import pack_bits
bast = (inst.dense_costs() > 0) if inst.maximize else (inst.dense_costs() < 0)
x0 = np.tile(pack_bits(bast.astype(np.uint8)), (batch, 1))
names = ["rc", "rc-barrier", "sort+ties", "fz", "select+sub", "pi,d", "affect+repair", "stale+violated",
         "w:stale,table,group", "w:descr,loads issued", "w:rc", "w:rank", "w:B1 wait", "w:ties,select", "w:pi,affect",
         "w:B2 wait"]
names[1] = "w:pi (+general rc-barrier)"; names[2] = "w:affect,repair (+general sort)"; names[7] = "w:table refill (+general stale)"
for step in range(3):
    dev.set_x_all(x0); dev.order_init(0, sync=False); st = dev.sweep(0.02 * (step + 1), delta, 0.5)
    t = np.zeros((batch, 32), np.uint64); L.bxh_debug_profile(dev.h, t.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
    ms, n = dev.kernel_time(reset=True); rows = st["rows_in_sweep"].mean()
    print(f"step {step}: kernel {ms/n:.2f} ms rows {rows:.0f}")
    for i, nm in enumerate(names):
        print(f"   {nm:16s} {t[:, i].mean() / 100.0 / rows:8.3f} us/row")
