"""baryonyx::solve wall time on small instances: one launch per sweep
(BXC_SINGLE_SWEEPS=1) against up to 256 sweeps per launch (bxh_run, the default
where the short-row kernel runs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baryonyx_amd import host_ffi as H, synth
import tempfile
LP = os.path.join(ROOT, "tests", "golden", "lp")
cases = [("8_queens_puzzle.lp", dict(limit=200000, theta=0.5, delta=0.02, alpha=1.0, w=40.0)),
         ("scp41-shaped", dict(limit=5000, w=20.0, kappa_step=1e-4)),
         ("aa04-shaped", dict(limit=2000, w=20.0, kappa_step=1e-4, kappa_max=0.9))]
with tempfile.TemporaryDirectory() as d:
    for name, kw in cases:
        if name.endswith(".lp"):
            path = os.path.join(LP, name)
        else:
            path = os.path.join(d, name + ".lp")
            synth.write_lp(synth.config_c1("int") if name.startswith("scp41") else synth.config_c3("int"), path)
        H.run(path, 0, H.default_params(seed=3, limit=10))  # warm-up
        t0 = time.perf_counter()
        res, _ = H.run(path, 0, H.default_params(seed=123456789, **kw))
        dt = time.perf_counter() - t0
        print(f"{name:22s} status {res.status} loop {res.loop:7d} remaining {res.remaining_constraints if res.status else 0:5d} "
              f"value {res.value:12.3f}  {dt*1e3:9.1f} ms  ({os.environ.get('BXC_SINGLE_SWEEPS') and 'one launch per sweep' or 'bxh_run'})")
