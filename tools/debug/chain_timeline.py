"""Timeline of the row chain (pipelined sweep, C2 bench shape): a -DBXH_TIMELINE build of the
library (csrc/libbxh_tl.so) writes the shader clock of every wave at 13 points of rows 200..207
of instance 0; this prints, per wave, the average time between consecutive points and the
arrival of every wave at the three barriers relative to the first.  Diagnostic only.
  hipcc ... -DBXH_TIMELINE -o baryonyx_amd/csrc/libbxh_tl.so baryonyx_amd/csrc/bxh_device.hip
  python tools/debug/chain_timeline.py [real|int]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baryonyx_amd import _ffi, synth
from baryonyx_amd.driver import compute_delta, normalize_costs
from baryonyx_amd.util import pack_bits
_ffi.LIB_PATH = os.path.join(_ffi.CSRC, "libbxh_tl.so")
os.environ.pop("BXH_LIB", None)
L = _ffi.lib()
L.bxh_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
kind = sys.argv[1] if len(sys.argv) > 1 else "real"
inst = synth.config_c2(kind); batch = 64
cost = normalize_costs(inst.dense_costs(), 4, 1); delta = compute_delta(cost, 0.5, 1)
dev = _ffi.DeviceBatch(inst, cost, batch, float_type=1)
bast = (inst.dense_costs() > 0) if inst.maximize else (inst.dense_costs() < 0)
x0 = np.tile(pack_bits(bast.astype(np.uint8)), (batch, 1))
names = ["top", "fetch issued", "classify (others)", "late sums (staging)", "values in LDS", "before B1",
         "after B1", "next loads, message, loc", "keys/filter/minima", "after B2", "uniform selection",
         "affect", "after B3"]
NP = len(names)
for step in range(4):
    dev.set_x_all(x0); dev.order_init(0, sync=False)
    t = np.zeros((batch, 32), np.uint64); L.bxh_debug_profile(dev.h, t.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
    st = dev.sweep(0.02 * (step + 1), delta, 0.5)
    t = np.zeros((batch, 32), np.uint64); L.bxh_debug_profile(dev.h, t.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
    ms, n = dev.kernel_time(reset=True)
    tl = t.reshape(-1)[: 8 * 8 * 16].reshape(8, 8, 16).astype(np.int64)  # [row][wave][point]
    if step < 2:
        continue
    print(f"step {step}: kernel {ms / n:.2f} ms, form {dev.kernel_form}")
    cyc = tl[:, :, 1:NP] - tl[:, :, 0:NP - 1]          # per row, wave, phase
    row = tl[1:, 0, 0] - tl[:-1, 0, 0]
    print("  cycles per row (wave 0, top to top):", row.tolist())
    print("  phase                         " + " ".join(f"w{w:<5d}" for w in range(8)))
    for p in range(1, NP):
        print(f"  {names[p]:28s}  " + " ".join(f"{cyc[:, w, p - 1].mean():6.0f}" for w in range(8)))
    # (a row with more than 64 late columns gives its first staging wave a second chunk: the
    # points 13 / 15 then belong to that second pass -- in the C2 instance 1.5 % of the rows,
    # three of the eight sampled here)
    for nm, a, b in (("  late: staged", 2, 13), ("  late: summed (+ values asked)", 13, 15), ("  late: written", 15, 3)):
        dd = tl[:, :4, b] - tl[:, :4, a]
        print(f"  {nm:28s}  " + " ".join(f"{np.median(dd[:, w]):6.0f}" for w in range(4)) + "   (medians)")
    d = tl[:, :4, 13] - tl[:, :4, 2]
    print("  late chunk staged (cycles from the top of the late block), rows x staging waves:")
    for rr in range(8):
        print("     " + " ".join(f"{d[rr, w]:6d} (pieces {tl[rr, w, 14] // 1000}, late {tl[rr, w, 14] % 1000})" for w in range(4)))
    for b, p in (("B1", 5), ("B2", 8), ("B3", 11)):
        arr = tl[:, :, p] - tl[:, :, p].min(axis=1, keepdims=True)
        print(f"  arrival at {b} (cycles after the first wave): " + " ".join(f"{arr[:, w].mean():6.0f}" for w in range(8)))

    # ---- the first producer of the same instance (points 0..14, two chunks at most) ----
    pt = t.reshape(-1)[1024:2048].reshape(8, 8, 16).astype(np.int64)
    pn = ["top", "barrier A", "descriptor + classify", "message (thread 0 polls)", "barrier B",
          "chunk 1 staged", "chunk 1 next loads issued", "chunk 1 summed", "chunk 1 stored",
          "chunk 2 staged", "chunk 2 next loads issued", "chunk 2 summed", "chunk 2 stored",
          "chunk loop left", "descriptor of row + 2 asked"]
    prow = pt[1:, 0, 0] - pt[:-1, 0, 0]
    print("  producer 1, cycles per row (wave 0, top to top):", prow.tolist())
    print("  producer phase                " + " ".join(f"w{w:<5d}" for w in range(8)))
    for p in range(1, 15):
        cells = []
        for w in range(8):
            a, b = pt[:, w, p], pt[:, w, p - 1]
            if p in (9, 13):  # behind an optional second chunk: from whatever came last
                b = np.where(pt[:, w, 12] > pt[:, w, 8], pt[:, w, 12], pt[:, w, 8]) if p == 13 else pt[:, w, 8]
            ok = (a > 0) & (b > 0) & (a >= b)
            cells.append(f"{(a - b)[ok].mean():6.0f}" if ok.any() else "     -")
        print(f"  {pn[p]:28s}  " + " ".join(cells))
