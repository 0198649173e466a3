"""Diagnostic: where a bounded baryonyx::optimize spends its wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from baryonyx_amd import host_ffi as H, synth
inst = synth.config_c2("int")
path = "/tmp/c2_probe.scp"
synth.write_scp(inst, path)
minst = H.merged(path)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
H.optimize_trace(path, H.default_params(seed=1, thread=64, sweep_budget=2, time_limit=3600.0), minst.n)
t0 = time.perf_counter()
res, tr = H.optimize_trace(path, H.default_params(seed=20260101, thread=64, sweep_budget=rounds, time_limit=3600.0), minst.n)
print("wall %.3f s, value %s, calls %s" % (time.perf_counter() - t0, res.value, tr["worker_calls"].sum()))
