#!/usr/bin/env python3
"""bench.py -- row-updates/s of the in-the-middle inner loop on scpnrh-shaped
instances (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
         --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A *step* is one restart-sweep of every solver instance of the rank's batch:
x <- the Bastert start (all rows violated), compute_order::init, one full
compute_order::run sweep (lib/src/itm-common.hpp:660-680,800-880) with the
kappa of that restart (best_solution_recorder::reinit's kappa-improve cycle),
P and pi carried over from the previous restart exactly as optimize_functor
does (lib/src/itm-optimizer-common.hpp:649-663) -- i.e. m row updates per
instance and step, the workload BASELINE.md section 2 timed on the CPU.  Every
8th restart starts a new solve (Solver::reset(): P = pi = 0): restarting from
the same infeasible point otherwise multiplies pi by ~100 per restart and
overflows (fp32 after 23 restarts, fp64 after ~150).  Workload at every N (weak
scaling: fixed per-GPU work): BASELINE.json configs[1], an scpnrh1-shaped
synthetic set-covering instance (1000 x 10000, 5 % density, inequalities-01
path), 64 restarts per GPU; restarts are sharded over ranks with no data-path
collective, the incumbent objective is min-reduced over RCCL at the end.

Rank 0 prints ONE JSON line (see the task contract) with the extra objects
  roofline     -- algorithmic bytes (SURVEY.md 8(d) formula) / sweep-kernel
                  time from HIP events on the launch stream, vs 8 TB/s HBM
  cpu_baseline -- the CPU oracle (a port of the reference path) timed on this
                  host on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RESET_EVERY = 8  # restarts per solve in the synthetic workload (see step())


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="restarts per GPU")
    ap.add_argument("--float", dest="float_kind", choices=["double", "float"], default="double")
    ap.add_argument("--cost-kind", choices=["int", "real"], default="int")
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c5"], default="c2")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="wall budget of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores")
    return ap.parse_args()


def make_instance(args):
    from baryonyx_amd import synth

    return {"c1": synth.config_c1, "c2": synth.config_c2, "c3": synth.config_c3,
            "c5": synth.config_c5}[args.config](
        args.cost_kind
    )


def kappa_cycle(params, k):
    """kappa of the k-th restart: best_solution_recorder::reinit's
    kappa-improve branch (itm-optimizer-common.hpp:538-543), wrapping to
    init_kappa_improve_start when init_kappa_improve_stop is reached."""
    steps = int(round((params.init_kappa_improve_stop - params.init_kappa_improve_start)
                      / params.init_kappa_improve_increase))
    app = params.init_kappa_improve_start + params.init_kappa_improve_increase * (1 + k % max(steps, 1))
    return params.kappa_min + (params.kappa_max - params.kappa_min) * app


def cpu_run(inst, params, float_type, seconds, nthreads):
    """The same restart-sweep workload on host threads: every thread owns one
    oracle Solver (as every reference worker owns one Solver,
    itm-optimizer-common.hpp:649-653) and repeats {x = Bastert point,
    compute_order::init, one sweep} for ~`seconds`."""
    from oracle import oracle_ffi as O

    totals = [0] * nthreads
    stop_at = [0.0]

    def work(t):
        from baryonyx_amd.util import pack_bits
        s = O.Solver(inst, float_type=float_type, sort_mode=1, cost_norm=4, seed=1000 + t)
        delta = s.compute_delta(params.theta)
        bast = (inst.dense_costs() > 0) if inst.maximize else (inst.dense_costs() < 0)
        x0 = pack_bits(bast.astype(np.uint8))
        k = 0
        while time.perf_counter() < stop_at[0]:
            if k % RESET_EVERY == 0:
                s.reset()
            s.set_x(x0)
            s.order_init(params.order)
            s.sweep(kappa_cycle(params, k), delta, params.theta)
            k += 1
        totals[t] = s.row_updates

    ths = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    t0 = time.perf_counter()
    stop_at[0] = t0 + seconds
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t0
    return sum(totals) / dt, sum(totals), dt


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by
    the cgroup CPU quota (the GPU boxes expose 256 logical CPUs but run the job
    under a 16-CPU quota; oversubscribing it only measures the throttle)."""
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            ncpu = min(ncpu, max(1, int(float(quota) / float(period) + 0.5)))
    except OSError:
        pass
    return ncpu


def cpu_baseline(inst, params, float_type, seconds, threads):
    nthreads = threads if threads > 0 else usable_cores()
    one, n1, t1 = cpu_run(inst, params, float_type, seconds * 0.3, 1)
    allc, na, ta = cpu_run(inst, params, float_type, seconds * 0.7, nthreads)
    return {
        "value": allc,
        "unit": "row-updates/s",
        "cores": nthreads,
        "kind": "port",
        "one_thread_value": one,
        "sample": f"oracle Solver per thread, restart-sweeps of the same instance: "
                  f"{nthreads} threads x {ta:.1f}s = {na} row updates; 1 thread x {t1:.1f}s = {n1}",
    }


def kernel_label(form):
    """bxh_kernel_form -> the kernel the roofline line is about."""
    if form == 1:
        return "bxh::sweep_kernel (lane per column, 1 workgroup / instance)"
    if form == 2:
        return "bxh::sweep_kernel_v2 (staged, 1 workgroup / instance)"
    if form == 3:
        return "bxh::sweep_kernel_z"
    if 20 < form < 40:
        return f"bxh::sweep_kernel_v2 (staged, {form - 20} lock-step column owners / instance)"
    if form > 40:
        return (f"bxh::sweep_kernel_v2 (pipelined sweep: 1 row chain + {form - 41} producer "
                "workgroups / instance)")
    return "bxh::sweep_kernel"


def pmc_traffic(workload, float_type):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this very
    workload (profiles/r*/pmc_traffic.json, made by tools/pmc_traffic.py); the
    counters cannot be read from inside the process, so any other workload or
    float type reports null."""
    best = None
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for rnd in sorted(os.listdir(root)) if os.path.isdir(root) else []:
        path = os.path.join(root, rnd, "pmc_traffic.json")
        if os.path.isfile(path):
            with open(path) as f:
                rec = json.load(f)
            if rec.get("workload") == workload and float_type == 1:
                best = rec["traffic_bytes_per_launch"]
    return best


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU; BENCH_SHARE_GPU=1 (test rigs with one GPU) lets the ranks
    # share device 0 and talk over gloo instead of RCCL
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if share else "nccl", rank=rank, world_size=world)

    from baryonyx_amd import _ffi
    from baryonyx_amd.driver import Params, compute_delta, normalize_costs

    float_type = 1 if args.float_kind == "double" else 0
    F = 8 if float_type else 4
    inst = make_instance(args)
    params = Params()
    cost = normalize_costs(inst.dense_costs(), params.cost_norm, float_type)
    delta = compute_delta(cost, params.theta, float_type)

    dev = _ffi.DeviceBatch(inst, cost, args.batch, float_type=float_type, device=local_rank)
    dev.set_stream(torch.cuda.current_stream().cuda_stream)
    for b in range(args.batch):  # one std engine per worker, distinct seeds
        dev.set_rng(b, 1 + ((1 + rank) * 7919 + b * 104729) % 2147483645)
    scan_bytes = inst.nnz * 4 + inst.nnz / 8  # violated-row scan, per instance and sweep
    # the Bastert point (init_with_bastert, itm-common.hpp:255-283): x_j = 1 where
    # the cost pulls that way -- all zeros for a min-cost covering instance
    from baryonyx_amd.util import pack_bits
    bast = (inst.dense_costs() > 0) if inst.maximize else (inst.dense_costs() < 0)
    x0 = np.tile(pack_bits(bast.astype(np.uint8)), (args.batch, 1))
    assert x0.shape[1] == dev.nwords

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(k):
        # one restart of every instance: x <- Bastert point, compute_order::init,
        # one full sweep.  P and pi carry over from restart to restart as in
        # optimize_functor (itm-optimizer-common.hpp:658-663 never resets the
        # Solver), but restarting from the same infeasible point each time makes
        # pi grow a hundredfold per restart: every RESET_EVERY-th restart begins
        # a new solve (Solver::reset(), P = pi = 0) so that any --steps stays
        # finite in fp64 and fp32.  The first sweep after a reset is the
        # tie-heaviest one (reduced costs == costs).
        if k % RESET_EVERY == 0:
            dev.reset_all()
        dev.set_x_all(x0)
        dev.order_init(params.order, sync=False)
        kap = np.full(args.batch, kappa_cycle(params, k))
        dev.sweep_async(kap, delta, params.theta, order=params.order)

    for k in range(args.warmup):
        step(k)
    barrier()
    dev.kernel_time(reset=True)
    st0 = dev.status()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches = dev.kernel_time()
    st1 = dev.status()
    rows_total = int(st1["row_updates"].sum() - st0["row_updates"].sum())
    bytes0, bytes1 = float(st0["alg_bytes"].sum()), float(st1["alg_bytes"].sum())

    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    rows_t = torch.tensor([float(rows_total)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rows_t, op=dist.ReduceOp.SUM)
    dt_max = float(tmax.item())
    rows_all = float(rows_t.item())

    # incumbent objective after the timed region: min over all restarts, then
    # allreduce(min) over ranks (RCCL over xGMI)
    best = np.inf
    from baryonyx_amd.util import x_bits
    dense = inst.dense_costs()
    for b in range(args.batch):
        if st1["remaining"][b] == 0:
            xb = x_bits(dev.get_x(b), inst.n).astype(np.float64)
            best = min(best, float((dense * xb).sum() + inst.obj_constant))
    best_t = torch.tensor([best], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(best_t, op=dist.ReduceOp.MIN)

    if rank == 0:
        # algorithmic bytes: B(k) summed on the device over every
        # row updated, plus the violated-row scan of every instance and sweep
        mean_row_bytes = (bytes1 - bytes0) / max(rows_total, 1)
        alg = (bytes1 - bytes0) + args.steps * args.batch * scan_bytes
        per_launch_bytes = alg / max(launches, 1)
        per_launch_s = kernel_ms / 1e3 / max(launches, 1)
        achieved = per_launch_bytes / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
        shape = {"c1": "scp41-shaped SCP", "c2": "scpnrh1-shaped SCP", "c3": "aa04-shaped SPP",
                 "c5": "telebus-shaped knapsack rows"}[args.config]
        solver = ["equalities-01", "inequalities-01", "equalities-101", "inequalities-101",
                  "inequalities-Z"][inst.kind]
        workload = (f"{inst.name}: {shape} {inst.m}x{inst.n} nnz={inst.nnz}, "
                    f"{solver}, {args.batch} restarts/GPU, {args.cost_kind} costs")
        out = {
            "metric": "row-updates/sec (itm inner loop)",
            "value": rows_all / dt_max,
            "unit": "row-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if float_type else "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "restarts_per_gpu": args.batch,
                "row_updates": rows_all,
                "best_objective": None if not np.isfinite(best_t.item()) else float(best_t.item()),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(workload, float_type),
                "kernel": kernel_label(dev.kernel_form),
                "launches": int(launches),
                "avg_launch_ms": kernel_ms / max(launches, 1),
                "alg_bytes_per_row_update": mean_row_bytes,
            },
        }
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(inst, params, float_type, args.cpu_seconds,
                                               args.cpu_threads)
        elif args.cpu_seconds > 0:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    dev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
