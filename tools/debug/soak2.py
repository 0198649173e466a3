import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib
from baryonyx_amd import synth, _ffi as bxh
from oracle import oracle_ffi as oracle
tp = importlib.import_module("test_gpu_parity")
inst = synth.config_c1("int")
for batch, order in ((300, 0), (120, 2), (300, 2), (64, 2)):
    try:
        tp.run_case(oracle, bxh, inst, 1, False, order, 0.0, batch=batch, sweeps=3, push_at=(1,), expect_form=5)
        print("batch", batch, "order", order, "ok")
    except AssertionError as e:
        print("batch", batch, "order", order, "FAIL", str(e)[:80])
