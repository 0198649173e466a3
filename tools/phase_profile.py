#!/usr/bin/env python3
"""Diagnostic: per-phase wall-clock shares of the sweep kernel (needs
baryonyx_amd/csrc/libbxh_prof.so built with -DBXH_PROFILE). Not a benchmark."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baryonyx_amd import _ffi, synth  # noqa: E402
from baryonyx_amd.driver import compute_delta, normalize_costs  # noqa: E402

_ffi.LIB_PATH = os.path.join(_ffi.CSRC, "libbxh_prof.so")
L = _ffi.lib()
L.bxh_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]

float_type = 0 if "--float" in sys.argv else 1
cost_kind = "real" if "--real" in sys.argv else "int"
batch = 64
inst = synth.config_c2(cost_kind)
cost = normalize_costs(inst.dense_costs(), 4, float_type)
delta = compute_delta(cost, 0.5, float_type)
dev = _ffi.DeviceBatch(inst, cost, batch, float_type=float_type)
x0 = np.zeros((batch, dev.nwords), np.uint64)
names = ["rc:first-issue", "consts", "rc:final-barrier", "combine+exchange", "sort:other|fast:merge-partials",
         "thread0|fast:scan+keys", "affect", "violated", "sort:wave-bitonic|fast:keys+hash+reduce",
         "sort:barrier1|fast:barrier", "sort:rank-search|fast:d", "desc:next-row", "rc:wait+stage", "rc:issue-next",
         "rc:sum", "rc:tail"]
for step in range(4):
    dev.set_x_all(x0)
    dev.order_init(0, sync=False)
    dev.sweep(0.02 * (step + 1), delta, 0.5)
    t = np.zeros((batch, 32), np.uint64)
    L.bxh_debug_profile(dev.h, t.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
    tot = t[:, :16].sum(axis=1).mean()
    ms, n = dev.kernel_time(reset=True)
    print(f"step {step}: kernel {ms/n:.2f} ms; ticks/instance {tot:.0f} (= {tot/100e3:.2f} ms @100MHz)")
    for i, nm in enumerate(names):
        print(f"   {nm:12s} {t[:, i].mean() / tot * 100:6.2f} %   {t[:, i].mean() / 100.0 / 1000:8.3f} us/row")
    if t[:, 24:32].sum() > 0:  # pipelined sweep: the first producer of every instance
        for i, nm in enumerate(["wait: waves of the previous row", "descriptor + classify", "message poll",
                                "barrier", "sums + stores (wave 0)", "descriptor prefetch", "wait for loads + staging",
                                "issue of the next loads"]):
            print(f"   producer {nm:34s} {t[:, 24 + i].mean() / 100.0 / 1000:8.3f} us/row")
    for i, nm in enumerate(["barrier rc-final", "barrier exchange", "barrier minima", "barrier affect"]):
        print(f"   skew {nm:18s} last-first {t[:, 16 + i].mean() / 100.0 / 1000:6.3f} us/row   "
              f"wave 0 waits {t[:, 20 + i].mean() / 100.0 / 1000:6.3f} us/row")
