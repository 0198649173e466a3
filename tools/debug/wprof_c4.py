"""Diagnostic: where a long tied row of configs[3] (integer costs) spends its time in sweep_kernel_wide.
Needs csrc/libbxh_wprof.so = a -DBXH_WPROF build (e.g. with -DBXH_DEV_ONLY_NP=5 -DBXH_DEV_KEEP_WIDE).
    python tools/debug/wprof_c4.py [int|real] [steps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BXH_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                                     "baryonyx_amd", "csrc", "libbxh_wprof.so")
import numpy as np
import torch
import bench
kind = sys.argv[1] if len(sys.argv) > 1 else "int"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sys.argv = ["bench.py", "--config", "c4", "--cost-kind", kind]
args = bench.parse()
torch.cuda.set_device(0)
wl = bench.XLWorkload(args, torch, 0, 0, 1)
from baryonyx_amd import _ffi
L = _ffi.lib()
L.bxh_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
wl.step(0)
t = np.zeros((wl.batch, 32), np.uint64)
L.bxh_debug_profile(wl.dev.h, t.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
wl.dev.kernel_time(reset=True)
for k in range(1, 1 + steps):
    wl.step(k)
ms, n = wl.dev.kernel_time()
L.bxh_debug_profile(wl.dev.h, t.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
names = ["sort", "tie detection", "runs listed", "rejection pre-pass", "short runs (lane each)",
         "long runs / behind a rejection", "thread 0: rest, selection, pi, d"]
rows = steps * args.xl_sweep_rows
print(f"kernel {ms / n:.2f} ms per launch, {rows} row updates per instance in {steps} launches")
names += [None, "  sort: own chunk sorted, published", "  sort: first meeting", "  sort: final positions",
          "  sort: second meeting", "  sort: the copy of the sorted row",
          "tied, no sort: keys ordered, engine"]
tot = t[:, :len(names)].astype(np.float64).mean(axis=0)
for nm, v in zip(names, tot):
    if nm:
        print(f"  {nm:36s} {v / rows / 2400.0:8.2f} us per row update (2.4 GHz ticks)")

st = t[:, 16:23].astype(np.float64)
rows_t = st[:, 3].mean()
if rows_t > 0:
    print(f"tied long rows per instance: {rows_t:.0f} of {rows}; per tied row: {st[:, 0].mean() / rows_t:.1f} runs, "
          f"{st[:, 1].mean() / rows_t:.0f} of {st[:, 4].mean() / rows_t:.0f} elements in runs, longest run {st[:, 2].max():.0f}, "
          f"distinct keys {st[:, 6].mean() / rows_t:.0f} on average, {st[:, 5].max():.0f} at most")
tf = t[:, 24:27].astype(np.float64).mean(axis=0)
print(f"tied rows per instance: {tf[0]:.0f} without a sort ({tf[1]:.0f} of them with a split run), {tf[2]:.0f} sorted")
rp = t[:, [27, 28, 29, 30, 31, 23, 7]].astype(np.float64).mean(axis=0)
print("every row, by phase (us per row update at 2.4 GHz ticks; the tied path's stamps above lie inside 'selection and affect'):")
for nm, v in zip(["decrease_preference", "the column sums", "keys into the tie table, the record", "first meeting",
                  "selection and affect", "the keys leave the table", "second meeting"], rp):
    print(f"  {nm:36s} {v / rows / 2400.0:8.2f}")
print(f"  {'sum':36s} {rp.sum() / rows / 2400.0:8.2f}   (kernel: {ms / n * 1000.0 / args.xl_sweep_rows:.1f} us per row update)")
