"""Phase profile of the wave-per-row short-row kernel (bxh_swave.hpp; diagnostic -DBXH_PROFILE
build, libbxh_prof.so): us per row of every phase as wave 0 sees it, config c1."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baryonyx_amd import _ffi, synth
from baryonyx_amd.driver import compute_delta, normalize_costs
_ffi.LIB_PATH = os.path.join(_ffi.CSRC, "libbxh_prof.so")
os.environ.pop("BXH_LIB", None)
L = _ffi.lib()
L.bxh_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
kind = sys.argv[2] if len(sys.argv) > 2 else "int"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
inst = getattr(synth, "config_" + cfg)(kind)
cost = normalize_costs(inst.dense_costs(), 4, 1); delta = compute_delta(cost, 0.5, 1)
dev = _ffi.DeviceBatch(inst, cost, batch, float_type=1)
print("form", dev.kernel_form, "rmax", dev.rmax)
from baryonyx_amd.util import pack_bits
bast = (inst.dense_costs() > 0) if inst.maximize else (inst.dense_costs() < 0)
x0 = np.tile(pack_bits(bast.astype(np.uint8)), (batch, 1))
names = ["setup", "table, group start", "descr, loads issued", "reduced costs (load wait)", "rank, select",
         "B1 wait", "ties / coins", "pi, d, affect", "B2 wait", "stale (in 1), tail"]
for step in range(4):
    if step < 2:
        dev.set_x_all(x0); dev.order_init(0, sync=False)
    st = dev.sweep(0.02 * (step + 1), delta, 0.5)
    t = np.zeros((batch, 32), np.uint64); L.bxh_debug_profile(dev.h, t.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
    ms, n = dev.kernel_time(reset=True); rows = max(st["rows_in_sweep"].mean(), 1)
    print(f"step {step}: kernel {ms/max(n,1):.3f} ms rows {rows:.0f}")
    for i, nm in enumerate(names):
        per = "us/row" if 1 <= i <= 8 else "us/sweep"
        v = t[:, i].mean() / 100.0 / (rows if 1 <= i <= 8 else 1)
        print(f"   {nm:28s} {v:9.3f} {per}")
