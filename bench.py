#!/usr/bin/env python3
"""bench.py -- row-updates/s of the in-the-middle inner loop on scpnrh-shaped
instances (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
         --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --gpus N ...      (no WORLD_SIZE in the environment: starts its
                                     N ranks itself, as child processes)

A *step* is one restart-sweep of every solver instance of the rank's batch:
x <- the restart point, compute_order::init, one full compute_order::run sweep
(lib/src/itm-common.hpp:660-680,800-880) with the kappa of that restart
(best_solution_recorder::reinit's kappa-improve cycle), P and pi carried over
from the previous restart exactly as optimize_functor does
(lib/src/itm-optimizer-common.hpp:649-663).

Default workload at every N (weak scaling: fixed per-GPU work): BASELINE.json
configs[1], an scpnrh1-shaped synthetic set-covering instance (1000 x 10000,
5 % density, inequalities-01 path), 64 restarts per GPU; the restart point is
the Bastert point (all rows violated: m row updates per instance and step) and
every 8th restart starts a new solve (Solver::reset(): P = pi = 0) -- restarting
from the same infeasible point otherwise multiplies pi by ~100 per restart and
overflows (fp32 after 23 restarts, fp64 after ~150).

--config c4 is BASELINE.json configs[3], the synthetic SCP 100k x 1M at 2 %
(2e9 elements; generated on the device, handed over with bxh_create_device).
One full sweep of one instance would be 1e5 row updates of 0.8 GB each (80 TB),
so the restart point of a step is a point with `--xl-sweep-rows` violated rows
(everything set except the variables of those rows; the rows rotate through the
matrix from step to step): compute_order::init finds them, the sweep updates
them.  The number of instances is what fits in HBM beside the matrix (printed
in config.workload; --batch overrides).

Restarts are sharded over ranks with no data-path collective; the incumbent
objective is min-reduced over RCCL at the end and the winner's x broadcast
(baryonyx_amd/dist.py).

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
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RESET_EVERY = 8  # restarts per solve in the synthetic workload (see step())
INT_MAX = 2**31 - 1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0,
                    help="restarts per GPU (0: 64; 256 for c5; what fits in HBM for c4)")
    ap.add_argument("--float", dest="float_kind", choices=["double", "float"], default="double")
    ap.add_argument("--cost-kind", choices=["int", "real"], default="int")
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4", "c5"], default="c2")
    ap.add_argument("--xl-rows", type=int, default=100000, help="c4: rows (100k = BASELINE configs[3])")
    ap.add_argument("--xl-cols", type=int, default=1000000, help="c4: columns")
    ap.add_argument("--xl-density", type=float, default=0.02)
    ap.add_argument("--xl-sweep-rows", type=int, default=128,
                    help="c4: violated rows per restart point (row updates per instance and step)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="wall budget of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores")
    ap.add_argument("--exchange-every", type=int, default=8,
                    help="ranks > 1: allreduce(min) of the incumbent + broadcast of its x every T steps "
                         "(inside the timed region) and once at the end; 0 = at the end only")
    ap.add_argument("--e2e-rounds", type=int, default=100,
                    help="rounds of the end-to-end baryonyx::optimize run next to the oracle's (0 = skip)")
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


CPU_SORT_MODE = 0  # the oracle's SORT_STD (the reference's own std::sort); 1 = the canonical stable rule


def cpu_run(inst, params, float_type, seconds, nthreads, rows=None):
    """The same restart-sweep workload on host threads: every thread owns one
    oracle Solver (as every reference worker owns one Solver,
    itm-optimizer-common.hpp:649-653) and repeats {x = restart point,
    compute_order::init, one sweep} for ~`seconds`.  rows: update these rows
    only (the c4 sample) instead of restarting from the Bastert point."""
    from oracle import oracle_ffi as O

    totals = [0] * nthreads
    stop_at = [0.0]

    def work(t):
        from baryonyx_amd.util import pack_bits
        # sort_mode 0: calculator_sort with std::sort, as the reference runs it
        # (itm-common.hpp:134-148); contract checks off: the reference's
        # BARYONYX_FULL_OPTIMIZATION build (debug.hpp:103-111), its faster one
        s = O.Solver(inst, float_type=float_type, sort_mode=CPU_SORT_MODE, cost_norm=4, seed=1000 + t)
        s.set_contract_checks(False)
        delta = s.compute_delta(params.theta)
        bast = (inst.dense_costs() > 0) if inst.maximize else (inst.dense_costs() < 0)
        x0 = pack_bits(bast.astype(np.uint8))
        k = 0
        while time.perf_counter() < stop_at[0]:
            if rows is not None:
                s.update_rows(np.roll(rows, -k)[: max(1, len(rows) // 8)], kappa_cycle(params, k),
                              delta, params.theta)
            else:
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
        "sample": f"oracle Solver per thread (sort_mode={CPU_SORT_MODE}: "
                  f"{'std::sort as the reference' if CPU_SORT_MODE == 0 else 'canonical stable sort'}; "
                  f"contract checks off = the reference's full-optimisation build), restart-sweeps of the "
                  f"same instance: {nthreads} threads x {ta:.1f}s = {na} row updates; "
                  f"1 thread x {t1:.1f}s = {n1}",
    }


def kernel_label(form):
    """bxh_kernel_form -> the kernel the roofline line is about."""
    if form == 1:
        return "bxh::sweep_kernel (lane per column, 1 workgroup / instance)"
    if form == 2:
        return "bxh::sweep_kernel_v2 (staged, 1 workgroup / instance)"
    if form == 3:
        return "bxh::sweep_kernel_z"
    if form == 6:
        return "bxh::sweep_kernel_z (a wave per row, up to 4 independent rows of an instance side by side)"
    if form == 5:
        return "bxh::sweep_kernel_narrow (short rows: one or four waves / instance)"
    if form == 7:
        return "bxh::sweep_kernel_sw (short rows: a wave per row, up to 4 independent rows of an instance side by side)"
    if form > 100:
        return f"bxh::sweep_kernel_wide ({form - 100} workgroups / instance share every row)"
    if 20 < form < 40:
        return f"bxh::sweep_kernel_v2 (staged, {form - 20} lock-step column owners / instance)"
    if form > 40:
        return (f"bxh::sweep_kernel_v2 (pipelined sweep: 1 row chain + {form - 41} producer "
                "workgroups / instance)")
    return "bxh::sweep_kernel"


def pmc_traffic(workload, float_type, form):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this very
    workload AND kernel form (profiles/r*/pmc_traffic.json, made by
    tools/pmc_traffic.py); the counters cannot be read from inside the process,
    so any other workload, float type or kernel form reports null."""
    best = None
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for rnd in sorted(os.listdir(root)) if os.path.isdir(root) else []:
        for name in sorted(os.listdir(os.path.join(root, rnd))):
            if not (name.startswith("pmc_traffic") and name.endswith(".json")):
                continue
            with open(os.path.join(root, rnd, name)) as f:
                rec = json.load(f)
            if ((rec.get("workload") or "").split(":")[0] == workload.split(":")[0] and float_type == 1
                    and rec.get("kernel_form", 44) == form):
                best = rec["traffic_bytes_per_launch"]
    return best


# ---------------------------------------------------------------- c4 (XL) ----
def xl_generate(torch, m, n, density, cost_kind, seed, device):
    """G(scp, m, n, density, cost_kind, seed) of SURVEY.md 8(d) built on the
    device: per row Binomial(n, density) uniform column draws, sorted, duplicates
    dropped (rows stay non-empty).  Returns row_ptr (int32 [m+1]), col (int32
    [nnz]), costs (float64 [n]) as device tensors."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    counts = torch.binomial(torch.full((m,), float(n), device=device),
                            torch.full((m,), density, device=device), generator=g).to(torch.int64)
    counts.clamp_(min=1)
    cap = int(counts.sum().item())
    col = torch.empty(cap, dtype=torch.int32, device=device)
    lens = torch.empty(m, dtype=torch.int64, device=device)
    chunk = max(1, min(m, (1 << 25) // max(1, int(counts.max().item()))))
    at = 0
    for r0 in range(0, m, chunk):
        c = counts[r0:r0 + chunk]
        K = int(c.max().item())
        u = torch.rand((c.shape[0], K), device=device, generator=g)
        cols = (u * n).to(torch.int32).clamp_(max=n - 1)
        cols[torch.arange(K, device=device)[None, :] >= c[:, None]] = n  # beyond the row's draws
        cols, _ = torch.sort(cols, dim=1)
        valid = cols < n
        valid[:, 1:] &= cols[:, 1:] != cols[:, :-1]
        ln = valid.sum(1)
        lens[r0:r0 + chunk] = ln
        flat = cols[valid]
        col[at:at + flat.shape[0]] = flat
        at += int(flat.shape[0])
        del u, cols, valid, flat
    col = col[:at]
    row_ptr = torch.zeros(m + 1, dtype=torch.int64, device=device)
    row_ptr[1:] = torch.cumsum(lens, 0)
    assert int(row_ptr[-1].item()) == at < INT_MAX
    if cost_kind == "int":
        cost = torch.randint(1, 101, (n,), device=device, generator=g).to(torch.float64)
    else:
        cost = 1.0 + 99.0 * torch.rand(n, device=device, generator=g, dtype=torch.float64)
    return row_ptr.to(torch.int32), col, cost


class XLWorkload:
    """BASELINE configs[3] on one GPU (see the module docstring)."""

    def __init__(self, args, torch, rank, local_rank, float_type):
        from baryonyx_amd import _ffi
        from baryonyx_amd.driver import Params, compute_delta, normalize_costs

        self.torch = torch
        self.args = args
        device = torch.device("cuda", local_rank)
        m, n, d = args.xl_rows, args.xl_cols, args.xl_density
        F = 8 if float_type else 4
        t0 = time.perf_counter()
        row_ptr, col, cost = xl_generate(torch, m, n, d, args.cost_kind, 7, device)
        nnz = int(col.shape[0])
        self.m, self.n, self.nnz = m, n, nnz
        self.params = Params()
        self.dense = cost.cpu().numpy()
        cost_h = normalize_costs(self.dense, self.params.cost_norm, float_type)
        self.delta = compute_delta(cost_h, self.params.theta, float_type)
        cost_n = torch.from_numpy(np.ascontiguousarray(cost_h, np.float64)).to(device)
        cmin = torch.ones(m, dtype=torch.int32, device=device)
        cmax = torch.full((m,), INT_MAX, dtype=torch.int32, device=device)
        torch.cuda.synchronize()
        gen_s = time.perf_counter() - t0
        # instances: what fits beside the matrix (CSR columns + column view +
        # element -> slot map = 12 B per element, the build's counters 1 KB per
        # column for a moment) with 3 % of HBM left alone
        torch.cuda.empty_cache()
        free, total = torch.cuda.mem_get_info(device)
        nnzp = nnz + 16 * n
        shared = 8 * nnz + 4 * nnzp + 1100 * n
        per_inst = nnzp * F + 64 * m * F + (1 << 26)
        fit = int((free - shared - 0.03 * total) // per_inst)
        batch = args.batch if args.batch > 0 else max(1, min(fit, 42))
        self.batch = batch
        t0 = time.perf_counter()
        self.dev = _ffi.DeviceBatch.on_device(
            m, n, nnz, row_ptr.data_ptr(), col.data_ptr(), 0, cmin.data_ptr(), cmax.data_ptr(),
            cost_n.data_ptr(), 1, False, batch, float_type=float_type, device=local_rank)
        torch.cuda.synchronize()
        self.build_s = time.perf_counter() - t0
        self.row_ptr, self.col = row_ptr, col  # restart points are made from the CSR
        self.row_ptr_h = row_ptr.cpu().numpy().astype(np.int64)
        del cmin, cmax, cost_n
        torch.cuda.empty_cache()
        self.dev.set_stream(torch.cuda.current_stream().cuda_stream)
        for b in range(batch):
            self.dev.set_rng(b, 1 + ((1 + rank) * 7919 + b * 104729) % 2147483645)
        self.S = max(1, min(args.xl_sweep_rows, m))
        self.scan_bytes = nnz * 4 + nnz / 8
        self.name = (f"xl-scp-{m}x{n}-d{d}-{args.cost_kind}-s7: synthetic SCP {m}x{n} nnz={nnz}, "
                     f"inequalities-01, {batch} restarts/GPU (what fits in "
                     f"{total / 1e9:.0f} GB of HBM beside the matrix: {per_inst / 1e9:.1f} GB each), "
                     f"{self.S} row updates per instance and step, {args.cost_kind} costs; "
                     f"generated on the device in {gen_s:.1f}s, matrix built in {self.build_s:.1f}s")
        self.weights = (1 << torch.arange(8, device=device, dtype=torch.int32)).to(torch.uint8)
        self.nwords = self.dev.nwords
        self._x = None

    def restart_points(self, k):
        """x of every instance: everything set except the variables of S
        consecutive rows (rows (k * batch + b) * S ..., wrapping)."""
        torch = self.torch
        m, n, B, S = self.m, self.n, self.batch, self.S
        bits = torch.ones((B, self.nwords * 64), dtype=torch.uint8, device=self.col.device)
        bits[:, n:] = 0
        for b in range(B):
            r0 = ((k * B + b) * S) % m
            r1 = min(m, r0 + S)
            e0, e1 = int(self.row_ptr_h[r0]), int(self.row_ptr_h[r1])
            bits[b, self.col[e0:e1].long()] = 0
        words = (bits.view(B, self.nwords * 8, 8) * self.weights).sum(2, dtype=torch.int32).to(torch.uint8)
        return words.contiguous()

    def step(self, k):
        x = self.restart_points(k)
        # (torch's default stream is the legacy null stream, which bxh_set_stream
        # takes as "use the handle's own": x must be complete before the copy)
        self.torch.cuda.current_stream().synchronize()
        self.dev.set_x_all_device(x.data_ptr(), keep=(x, self._x))
        self._x = x
        self.dev.order_init(self.params.order, sync=False)
        kap = np.full(self.batch, kappa_cycle(self.params, k))
        self.dev.sweep_async(kap, self.delta, self.params.theta, order=self.params.order)

    def objective(self, b):
        from baryonyx_amd.util import x_bits
        xb = x_bits(self.dev.get_x(b), self.n).astype(np.float64)
        return float((self.dense * xb).sum())

    def cpu_baseline(self, float_type, seconds, threads, mean_row_bytes):
        """The oracle cannot hold the 2e9-element matrix in a bench's time
        budget; it is timed on a slice with the same columns (same m, same
        density: ~2000 entries per column) and fewer of them, and its algorithmic
        byte rate is converted to row updates of the full instance."""
        torch = self.torch
        from baryonyx_amd import synth
        nthreads = threads if threads > 0 else usable_cores()
        n2 = max(1000, self.n // 40)
        rp, cl, cs = xl_generate(torch, self.m, n2, self.args.xl_density, self.args.cost_kind, 8,
                                 self.col.device)
        nz = int(cl.shape[0])
        inst = synth.Instance(
            m=self.m, n=n2, row_ptr=rp.cpu().numpy().astype(np.int32), col=cl.cpu().numpy(),
            factor=np.ones(nz, np.int32), cmin=np.ones(self.m, np.int32),
            cmax=np.full(self.m, INT_MAX, np.int32), obj_var=np.arange(n2, dtype=np.int32),
            obj_factor=cs.cpu().numpy(), maximize=False, kind=1, name="xl-slice")
        del rp, cl, cs
        F = 8 if float_type else 4
        clen = np.bincount(inst.col, minlength=n2).astype(np.float64)
        rows = np.arange(0, self.m, max(1, self.m // 256), dtype=np.int32)[:256]
        r = np.diff(inst.row_ptr)[rows].astype(np.float64)
        rb = np.array([clen[inst.col[inst.row_ptr[k]:inst.row_ptr[k + 1]]].sum() for k in rows])
        slice_row_bytes = float((rb * (4 + 2 * F) + r * (4 + 5 * F) + 2 * F).mean())
        one, n1, t1 = cpu_run(inst, self.params, float_type, seconds * 0.3, 1, rows=rows)
        allc, na, ta = cpu_run(inst, self.params, float_type, seconds * 0.7, nthreads, rows=rows)
        scale = slice_row_bytes / max(mean_row_bytes, 1.0)
        return {
            "value": allc * scale,
            "unit": "row-updates/s",
            "cores": nthreads,
            "kind": "port",
            "one_thread_value": one * scale,
            "sample": f"oracle Solver per thread on a {self.m}x{n2} slice at the same density "
                      f"(columns of ~{clen.mean():.0f} entries, rows of ~{r.mean():.0f}: "
                      f"{slice_row_bytes / 1e6:.1f} MB algorithmic per row update): {nthreads} threads x "
                      f"{ta:.1f}s = {na} slice row updates ({allc * slice_row_bytes / 1e9:.1f} GB/s "
                      f"algorithmic), converted at {mean_row_bytes / 1e6:.0f} MB per row update of the "
                      f"full instance",
        }


# ------------------------------------------------------------ c1 .. c5 -------
class StdWorkload:
    def __init__(self, args, torch, rank, local_rank, float_type):
        from baryonyx_amd import _ffi
        from baryonyx_amd.driver import Params, compute_delta, normalize_costs
        from baryonyx_amd.util import pack_bits

        self.args = args
        inst = make_instance(args)
        self.inst = inst
        self.m, self.n, self.nnz = inst.m, inst.n, inst.nnz
        self.params = Params()
        cost = normalize_costs(inst.dense_costs(), self.params.cost_norm, float_type)
        self.delta = compute_delta(cost, self.params.theta, float_type)
        # (BASELINE.json: configs[1] names 64 restarts per GPU, configs[4] -- c5 -- 256)
        self.batch = args.batch if args.batch > 0 else (256 if args.config == "c5" else 64)
        self.dev = _ffi.DeviceBatch(inst, cost, self.batch, float_type=float_type, device=local_rank)
        self.dev.set_stream(torch.cuda.current_stream().cuda_stream)
        # one std engine per worker, distinct seeds (one transfer)
        self.dev.set_rng_all(np.array([1 + ((1 + rank) * 7919 + b * 104729) % 2147483645
                                       for b in range(self.batch)], np.uint32))
        self.scan_bytes = inst.nnz * 4 + inst.nnz / 8  # violated-row scan, per instance and sweep
        # the Bastert point (init_with_bastert, itm-common.hpp:255-283): x_j = 1 where
        # the cost pulls that way -- all zeros for a min-cost covering instance
        bast = (inst.dense_costs() > 0) if inst.maximize else (inst.dense_costs() < 0)
        self.x0 = np.tile(pack_bits(bast.astype(np.uint8)), (self.batch, 1))
        assert self.x0.shape[1] == self.dev.nwords
        # (resident in HBM like every other input of the timed region: a restart copies it
        # device to device, in stream order)
        self.x0_dev = torch.from_numpy(np.ascontiguousarray(self.x0).view(np.int64)).to(torch.device("cuda", local_rank))
        shape = {"c1": "scp41-shaped SCP", "c2": "scpnrh1-shaped SCP", "c3": "aa04-shaped SPP",
                 "c5": "telebus-shaped knapsack rows"}[args.config]
        solver = ["equalities-01", "inequalities-01", "equalities-101", "inequalities-101",
                  "inequalities-Z"][inst.kind]
        self.name = (f"{inst.name}: {shape} {inst.m}x{inst.n} nnz={inst.nnz}, "
                     f"{solver}, {self.batch} restarts/GPU, {args.cost_kind} costs")
        self.dense = inst.dense_costs()

    def step(self, k):
        # one restart of every instance: x <- Bastert point, compute_order::init,
        # one full sweep.  P and pi carry over from restart to restart as in
        # optimize_functor (itm-optimizer-common.hpp:658-663 never resets the
        # Solver), but restarting from the same infeasible point each time makes
        # pi grow a hundredfold per restart: every RESET_EVERY-th restart begins
        # a new solve (Solver::reset(), P = pi = 0) so that any --steps stays
        # finite in fp64 and fp32.  The first sweep after a reset is the
        # tie-heaviest one (reduced costs == costs).
        p = self.params
        if k % RESET_EVERY == 0:
            self.dev.reset_all()
        self.dev.set_x_all_device(self.x0_dev.data_ptr(), keep=self.x0_dev)
        self.dev.order_init(p.order, sync=False)
        kap = np.full(self.batch, kappa_cycle(p, k))
        self.dev.sweep_async(kap, self.delta, p.theta, order=p.order)

    def objective(self, b):
        from baryonyx_amd.util import x_bits
        xb = x_bits(self.dev.get_x(b), self.n).astype(np.float64)
        return float((self.dense * xb).sum() + self.inst.obj_constant)

    def cpu_baseline(self, float_type, seconds, threads, mean_row_bytes):
        return cpu_baseline(self.inst, self.params, float_type, seconds, threads)


def end_to_end(inst, float_type, rounds, threads):
    """baryonyx::optimize (C++ host library, every sweep a launch + a status
    read-back) for a fixed number of rounds on this instance, next to the
    oracle's optimize() under the same deterministic schedule: objective at an
    equal sweep budget (identical by construction: tests/test_gpu_optimize.py)
    and the time both take.  The CPU figure is the CPU time the oracle's workers
    spend inside their sweeps divided by the cores they could run on (the
    reference's workers run freely; the oracle's lock-step conductor is test
    scaffolding and is not charged)."""
    import tempfile

    from baryonyx_amd import host_ffi as H
    from baryonyx_amd import synth
    from oracle import oracle_ffi as O

    workers = 64
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "instance.scp")
        synth.write_scp(inst, path)
        minst = H.merged(path)
        hp = H.default_params(seed=20260101, thread=workers, float_type=float_type, sweep_budget=rounds,
                              time_limit=3600.0)
        H.optimize_trace(path, H.default_params(seed=1, thread=workers, float_type=float_type,
                                                sweep_budget=2, time_limit=3600.0), minst.n)  # warm-up
        t0 = time.perf_counter()
        res, tr = H.optimize_trace(path, hp, minst.n)
        gpu_s = time.perf_counter() - t0
        host_setup_s, device_setup_s, rounds_s = H.last_optimize_timing()
    okw = {k: getattr(hp, k) for k in (
        "theta delta kappa_min kappa_step kappa_max alpha pushing_k_factor "
        "pushing_objective_amplifier init_policy_random w limit seed pushes_limit "
        "pushing_iteration_limit init_population_size order init_policy cost_norm").split()}
    t0 = time.perf_counter()
    ores, otr = O.optimize(minst, float_type, O.default_params(sort_mode=1, **okw), thread=workers,
                           rounds=rounds)
    cpu_wall = time.perf_counter() - t0
    cores = threads if threads > 0 else usable_cores()
    return {
        "what": f"baryonyx::optimize, {workers} workers, {rounds} rounds of one sweep each "
                f"(restarts, crossover / mutation, push phases as the reference schedules them)",
        "rounds": rounds,
        "gpu_seconds": gpu_s,
        # of which: the rounds themselves (every sweep of every worker, restarts, the
        # recorder's bookkeeping), the device set-up (handles, population upload), and
        # the reference's own host set-up (storage's constructor: 100 start points with
        # their pre-solves and evaluations -- not charged to the CPU figure either)
        "gpu_seconds_rounds": rounds_s,
        "gpu_seconds_device_setup": device_setup_s,
        "host_seconds_population_init": host_setup_s,
        "cpu_seconds": otr["sweep_seconds"] / min(cores, workers),
        "cpu_cores": min(cores, workers),
        "cpu_sweep_cpu_seconds": otr["sweep_seconds"],
        "cpu_wall_seconds_lockstep": cpu_wall,
        "objective_at_budget": {"gpu": res.value if res.status == 0 else None,
                                "cpu": ores.value if ores.status == 0 else None},
        "identical_population": bool(np.array_equal(tr["pop_hash"], otr["pop_hash"])
                                     and np.array_equal(tr["worker_rng"], otr["worker_rng"])),
        "row_updates": int(ores.row_updates),
    }


def spawn_ranks(n):
    """`bench.py --gpus N` outside a launcher: start the N ranks as children
    (before this process touches the GPU) and pass rank 0's line through."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    raise SystemExit(rc)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
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
        world = dist.get_world_size()

    float_type = 1 if args.float_kind == "double" else 0
    wl = (XLWorkload if args.config == "c4" else StdWorkload)(args, torch, rank, local_rank, float_type)
    dev = wl.dev

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        wl.step(k)
    if world > 1:
        # the collectives of the exchange step once, untimed: the first all-reduce /
        # broadcast of a process group sets its communicators up (seconds with RCCL)
        from baryonyx_amd.dist import reduce_incumbent
        reduce_incumbent(np.inf, np.zeros(dev.nwords, np.uint64), device="cpu" if share else "cuda")
        dist.all_reduce(torch.zeros(1, dtype=torch.float64, device="cpu" if share else "cuda"),
                        op=dist.ReduceOp.MAX)
    barrier()
    dev.kernel_time(reset=True)
    st0 = dev.status()
    # the exchange step of the restart pool: every T steps the ranks
    # agree on the incumbent, inside the timed region.  bxh_incumbent finds the
    # rank's best instance ON THE DEVICE (fewest violated rows, then the objective)
    # and leaves its key, objective and x in a tensor; the collectives of
    # baryonyx_amd/dist.py -- all-reduce(MIN) of the key's two words, then the
    # winner's record in one masked all-reduce -- work on that tensor, and the
    # carried incumbent stays on the device too: no read-back, no host loop.
    from baryonyx_amd.dist import read_record, record_words, reduce_incumbent_device
    maximize = bool(getattr(getattr(wl, "inst", None), "maximize", False))
    dev.set_objective(wl.dense, float(getattr(getattr(wl, "inst", None), "obj_constant", 0.0)), maximize)
    cdev = torch.device("cuda", local_rank)
    loc = torch.zeros(record_words(dev.nwords), dtype=torch.int64, device=cdev)
    incumbent = {"rec": None, "exchanges": 0}

    def exchange():
        dev.incumbent(rank, record_device=loc.data_ptr() + dev.nwords * 8, x_device=loc.data_ptr(),
                      read_back=False)
        # (the gloo test rig -- ranks sharing one GPU -- reduces on the host)
        incumbent["rec"] = reduce_incumbent_device(loc.cpu() if share else loc, carried=incumbent["rec"])
        incumbent["exchanges"] += 1

    if world > 1:
        exchange()  # untimed once: the kernels' first launch, the tensors' first use
        incumbent = {"rec": None, "exchanges": 0}
        barrier()

    t0 = time.perf_counter()
    for k in range(args.steps):
        wl.step(args.warmup + k)
        if world > 1 and args.exchange_every > 0 and (k + 1) % args.exchange_every == 0:
            exchange()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches = dev.kernel_time()
    st1 = dev.status()
    rows_total = int(st1["row_updates"].sum() - st0["row_updates"].sum())
    bytes0, bytes1 = float(st0["alg_bytes"].sum()), float(st1["alg_bytes"].sum())

    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else "cuda")
    rows_t = torch.tensor([float(rows_total)], dtype=torch.float64, device="cpu" if share else "cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rows_t, op=dist.ReduceOp.SUM)
    dt_max = float(tmax.item())
    rows_all = float(rows_t.item())

    # incumbent after the timed region: one more exchange (at world size 1 the same
    # device search without collectives), ONE read-back of the winner's record, and its
    # exact objective recomputed from x (the device sums in a tree: bxh_incumbent)
    exchange()
    _, xw_best, _, rem_best = read_record(incumbent["rec"])
    best_all = np.inf
    if rem_best == 0:
        from baryonyx_amd.util import x_bits
        best_all = float((wl.dense * x_bits(xw_best, dev.n).astype(np.float64)).sum()
                         + float(getattr(getattr(wl, "inst", None), "obj_constant", 0.0)))

    if rank == 0:
        # algorithmic bytes: B(k) summed on the device over every
        # row updated, plus the violated-row scan of every instance and sweep
        mean_row_bytes = (bytes1 - bytes0) / max(rows_total, 1)
        # (the violated-row scan belongs to the sweep kernel's bytes only where it
        # runs inside that kernel: the wide kernel's is a launch of its own)
        scan = wl.scan_bytes if dev.sweep_includes_scan else 0.0
        alg = (bytes1 - bytes0) + args.steps * wl.batch * scan
        per_launch_bytes = alg / max(launches, 1)
        per_launch_s = kernel_ms / 1e3 / max(launches, 1)
        achieved = per_launch_bytes / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
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
                "workload": wl.name,
                "restarts_per_gpu": wl.batch,
                "row_updates": rows_all,
                "rows_in_last_sweep": [int(v) for v in st1["rows_in_sweep"][:4]],
                "best_objective": None if not np.isfinite(best_all) else float(best_all),
                "incumbent_exchanges": incumbent["exchanges"],
                "world_size": world,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(wl.name, float_type, dev.kernel_form),
                "kernel": kernel_label(dev.kernel_form),
                "launches": int(launches),
                "avg_launch_ms": kernel_ms / max(launches, 1),
                "alg_bytes_per_row_update": mean_row_bytes,
            },
        }
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = wl.cpu_baseline(float_type, args.cpu_seconds, args.cpu_threads,
                                                  mean_row_bytes)
        elif args.cpu_seconds > 0:
            out["cpu_baseline"] = None
    dev.close()
    if rank == 0:
        if args.e2e_rounds > 0 and world == 1 and args.config != "c4":
            out["end_to_end"] = end_to_end(wl.inst, float_type, args.e2e_rounds, args.cpu_threads)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
