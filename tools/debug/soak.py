"""One-off soak runs (not part of the suite): more sweeps of the multi-workgroup /
multi-wave kernels against the oracle than the tests do."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib
from baryonyx_amd import synth, _ffi as bxh
from oracle import oracle_ffi as oracle
tp = importlib.import_module("test_gpu_parity")
which = sys.argv[1]
if which == "wide":
    os.environ["BXH_GROUPS"] = "w16"
    inst = synth.make_scp(3000, 2500, 0.08, "int", 231)
    tp.run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=16, sweeps=30, push_at=(7, 19), expect_form=116)
elif which == "narrow4":
    os.environ["BXH_NARROW_W"] = "4"
    inst = synth.config_c3("int")
    tp.run_case(oracle, bxh, inst, 1, False, 0, 0.0, batch=70, sweeps=15, push_at=(5, 11), expect_form=5)  # (run_case: kappa = 0.01 b + 0.02 sweep < 1)
elif which == "narrow1":
    inst = synth.config_c1("int")
    tp.run_case(oracle, bxh, inst, 1, False, 2, 0.0, batch=90, sweeps=5, push_at=(2,), expect_form=5)
print("soak", which, "ok")
