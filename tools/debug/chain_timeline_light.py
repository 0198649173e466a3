import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from baryonyx_amd import _ffi, synth
from baryonyx_amd.driver import compute_delta, normalize_costs
from baryonyx_amd.util import pack_bits
_ffi.LIB_PATH = os.path.join(_ffi.CSRC, os.environ.get("BXH_TL_LIB", "libbxh_tl.so"))
os.environ.pop("BXH_LIB", None)
L = _ffi.lib()
L.bxh_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
inst = synth.config_c2("int"); batch = 64
cost = normalize_costs(inst.dense_costs(), 4, 1); delta = compute_delta(cost, 0.5, 1)
dev = _ffi.DeviceBatch(inst, cost, batch, float_type=1)
bast = (inst.dense_costs() > 0) if inst.maximize else (inst.dense_costs() < 0)
x0 = np.tile(pack_bits(bast.astype(np.uint8)), (batch, 1))
pts = [int(a) for a in sys.argv[1].split(",")]
for step in range(4):
    dev.set_x_all(x0); dev.order_init(0, sync=False)
    t = np.zeros((batch, 32), np.uint64); L.bxh_debug_profile(dev.h, t.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
    st = dev.sweep(0.02 * (step + 1), delta, 0.5)
    t = np.zeros((batch, 32), np.uint64); L.bxh_debug_profile(dev.h, t.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
    ms, n = dev.kernel_time(reset=True)
    tl = t.reshape(-1)[: 8 * 8 * 16].reshape(8, 8, 16).astype(np.int64)
    if step < 2: continue
    print(f"step {step}: kernel {ms / n:.3f} ms")
    for w in range(8):
        seq = []
        for r in range(8):
            for p in pts: seq.append((tl[r, w, p], r, p))
        d = {}
        for i in range(1, len(seq)):
            key = (seq[i-1][2], seq[i][2]); d.setdefault(key, []).append(seq[i][0] - seq[i-1][0])
        print(" wave", w, {f"{a}->{b}": int(np.median(v)) for (a, b), v in d.items()})
