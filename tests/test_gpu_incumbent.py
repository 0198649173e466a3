"""-m gpu: bxh_incumbent -- the batch's best instance found on the device (fewest
violated rows, then the objective: storage's order,
lib/src/itm-optimizer-common.hpp:424-442; raw_result_compare, lib/src/result.hpp:96-107), its 32-byte record and the
winner's x left in device memory for the exchange step (baryonyx_amd/dist.py)."""
import numpy as np
import pytest

from baryonyx_amd import synth
from baryonyx_amd.dist import pack_key, read_record, record_words
from baryonyx_amd.util import pack_bits, x_bits
from helpers import is_feasible

pytestmark = pytest.mark.gpu


def _violated(inst, words):
    x = x_bits(words, inst.n).astype(np.int64)
    v = np.add.reduceat(inst.factor.astype(np.int64) * x[inst.col], inst.row_ptr[:-1])
    return int(np.sum((v < inst.cmin) | (v > inst.cmax)))


def _batch(bxh, oracle, inst, xs, float_type=1):
    orc = oracle.Solver(inst, float_type=float_type, sort_mode=1, cost_norm=4, seed=3)
    dev = bxh.DeviceBatch(inst, orc.cost(), len(xs), float_type=float_type, device=0)
    for b, x in enumerate(xs):
        dev.set_x(b, x)
    rem = dev.order_init(0)  # compute_order::init: the violated rows of every instance
    return dev, rem


@pytest.mark.parametrize("cost_kind,maximize", [("int", False), ("real", False), ("int", True)])
def test_incumbent_is_the_recorders_choice(bxh, oracle, cost_kind, maximize):
    inst = synth.make_scp(60, 700, 0.05, cost_kind, 11)
    rng = np.random.default_rng(5)
    dense = inst.dense_costs()
    # instances 0..3 infeasible with various counts, 4..8 feasible with various objectives
    xs = [pack_bits((rng.random(inst.n) < p).astype(np.uint8)) for p in (0.0, 0.004, 0.008, 0.012)]
    xs += [pack_bits((rng.random(inst.n) < p).astype(np.uint8)) for p in (0.5, 0.35, 0.9, 0.35, 0.6)]
    xs[7] = xs[5].copy()  # an exact tie of two feasible instances: the lower instance wins
    dev, rem = _batch(bxh, oracle, inst, xs)
    for b, x in enumerate(xs):
        assert rem[b] == _violated(inst, x)
    assert any(r == 0 for r in rem) and any(r > 0 for r in rem)
    dev.set_objective(dense, 2.5, maximize)
    rec = dev.incumbent(rank=5)
    obj = [float((dense * x_bits(x, inst.n)).sum() + 2.5) for x in xs]
    feas = [b for b in range(len(xs)) if rem[b] == 0]
    want = (max if maximize else min)(feas, key=lambda b: (obj[b], -b if maximize else b))
    assert rec["instance"] == want and rec["remaining"] == 0
    if cost_kind == "int":
        assert rec["value"] == obj[want]  # integer costs: the tree sum is exact
    else:
        assert rec["value"] == pytest.approx(obj[want], rel=1e-13)
    assert (int(rec["key_hi"]), int(rec["key_lo"])) == pack_key(0, float(rec["value"]), 5, maximize)
    dev.close()


def test_nobody_feasible_and_device_side_record(bxh, oracle):
    import torch

    inst = synth.make_scp(50, 400, 0.05, "int", 12)
    rng = np.random.default_rng(6)
    xs = [pack_bits((rng.random(inst.n) < p).astype(np.uint8)) for p in (0.0, 0.01, 0.02, 0.015, 0.02)]
    dev, rem = _batch(bxh, oracle, inst, xs)
    assert all(r > 0 for r in rem)
    dev.set_objective(inst.dense_costs(), 0.0, False)
    want = int(np.argmin(rem))  # the fewest violated rows; ties to the lower instance
    rec = dev.incumbent(rank=0)
    assert rec["instance"] == want and rec["remaining"] == rem[want] and np.isinf(rec["value"])
    # the same search with the record and x left in a tensor (what bench.py all-reduces)
    dev.set_stream(torch.cuda.current_stream().cuda_stream)
    loc = torch.zeros(record_words(dev.nwords), dtype=torch.int64, device="cuda")
    assert dev.incumbent(3, record_device=loc.data_ptr() + dev.nwords * 8, x_device=loc.data_ptr(),
                         read_back=False) is None
    value, xw, owner, remaining = read_record(loc)
    assert np.isinf(value) and owner == 3 and remaining == rem[want]
    assert np.array_equal(xw, xs[want]) and not is_feasible(inst, xw)
    dev.close()


def test_incumbent_needs_the_objective(bxh, oracle):
    inst = synth.make_scp(20, 100, 0.1, "int", 13)
    dev, _ = _batch(bxh, oracle, inst, [pack_bits(np.zeros(inst.n, np.uint8))])
    with pytest.raises(bxh.BxhError):
        dev.incumbent()
    dev.close()


def test_rccl_takes_the_collectives_of_the_exchange():
    """The exchange step's three collectives (baryonyx_amd/dist.py) on RCCL itself: a process
    group of ONE rank on the GPU -- every all-reduce is the identity, but the backend has to
    accept the tensors as they are: int64 on the device, MIN and SUM.  (More ranks need more
    GPUs than a test box has; the multi-rank logic runs on gloo, tests/test_dist_gloo.py.)"""
    import os
    import socket

    import torch
    import torch.distributed as dist

    from baryonyx_amd.dist import make_record, read_record

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        rec = make_record(193.0, np.array([0xFFFFFFFFFFFFFFFF, 5], np.uint64), 0, 0, device="cuda")
        n = rec.shape[0] - 4
        hi = rec[n:n + 1].clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MIN)
        lo = rec[n + 1:n + 2].clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        out = rec * ((rec[n:n + 1] == hi) & (rec[n + 1:n + 2] == lo)).to(torch.int64)
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        value, xw, owner, remaining = read_record(out)
        assert (value, xw.tolist(), owner, remaining) == (193.0, [0xFFFFFFFFFFFFFFFF, 5], 0, 0)
    finally:
        dist.destroy_process_group()
