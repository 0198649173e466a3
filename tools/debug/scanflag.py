import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BXH_GROUPS"] = "w2"
import numpy as np
from baryonyx_amd import _ffi, synth
from baryonyx_amd.driver import normalize_costs
inst = synth.make_scp(60, 400, 0.06, "real", 11)
dev = _ffi.DeviceBatch(inst, normalize_costs(inst.dense_costs(), 4, 1), 3, float_type=1)
print("form", dev.kernel_form, "includes_scan", dev.sweep_includes_scan, dev.L.bxh_sweep_includes_scan(dev.h))
