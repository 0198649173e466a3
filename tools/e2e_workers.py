"""How the host side of baryonyx::optimize scales with the number of workers on
ONE GPU: for 64 / 256 / 1024 / 2048 workers, a fixed number of
rounds of the product's optimize() on a BASELINE shape, reporting per round the wall
time and the host's share outside its wait for the device (restart bookkeeping,
launches, recorder updates: one host thread per process).  If that share grows
linearly with the workers and stops being hidden behind the device, 8 x 256 workers
per process need one bookkeeping thread per device.

    python tools/e2e_workers.py [c1|c2|c3|c5] [rounds] [workers ...]
"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from baryonyx_amd import host_ffi as H  # noqa: E402
from baryonyx_amd import synth  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    counts = [int(a) for a in sys.argv[3:]] or [64, 256, 1024, 2048]
    inst = {"c1": synth.config_c1, "c2": synth.config_c2, "c3": synth.config_c3, "c5": synth.config_c5}[cfg]("int")
    out = []
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "instance.lp")
        synth.write_lp(inst, path)
        n = H.merged(path).n
        for workers in counts:
            kw = dict(thread=workers, time_limit=3600.0)
            H.optimize_trace(path, H.default_params(seed=1, sweep_budget=2, **kw), n)  # warm-up
            t0 = time.perf_counter()
            res, _ = H.optimize_trace(path, H.default_params(seed=20260101, sweep_budget=rounds, **kw), n)
            wall = time.perf_counter() - t0
            host_setup, device_setup, rounds_s = H.last_optimize_timing()
            host_s, nr = H.last_optimize_host_rounds()
            out.append({
                "config": cfg, "workers": workers, "rounds": nr, "wall_s": wall,
                "host_setup_s": host_setup, "device_setup_s": device_setup,
                "us_per_round": rounds_s / max(nr, 1) * 1e6,
                "host_us_per_round": host_s / max(nr, 1) * 1e6,
                "host_us_per_round_per_worker": host_s / max(nr, 1) / workers * 1e6,
                "status": int(res.status), "value": float(res.value),
            })
            print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
