"""CPU, gloo, world sizes 2 and 4: the exchange step of the multi-GPU restart
pool -- one all-reduce(MIN) of the packed 63-bit key (violated rows, objective,
rank), then the winner's x and exact objective in one masked all-reduce
(baryonyx_amd/dist.py)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from baryonyx_amd.dist import REMAINING_MAX, ordered_bits, pack_key


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_key_orders_like_the_recorder():
    """storage::sort (itm-optimizer-common.hpp:424-442): fewer violated rows first,
    then the better objective; the rank breaks what is left."""
    vals = [-1e30, -7.5, -1.0, -0.0, 0.0, 1e-300, 1.0, 193.0, 194.0, 7.3e6, 1e30, float("inf")]
    imgs = [ordered_bits(v) for v in vals]
    assert imgs == sorted(imgs) and imgs[3] < imgs[4]  # -0.0 sorts in front of +0.0: harmless
    assert [ordered_bits(v, True) for v in vals] == sorted((ordered_bits(v, True) for v in vals), reverse=True)
    # remaining dominates the objective, the objective dominates the rank
    assert pack_key(0, 1e30, 63) < pack_key(1, -1e30, 0)
    assert pack_key(3, 193.0, 5) < pack_key(3, 194.0, 0)
    assert pack_key(3, 193.0, 1) < pack_key(3, 193.0, 2)
    assert pack_key(10 ** 9, 0.0, 0) == pack_key(REMAINING_MAX, 0.0, 0)  # clipped
    assert 0 <= pack_key(REMAINING_MAX, float("inf"), 63) < 2 ** 63     # a non-negative int64
    # integer objectives below 2**27 keep every bit
    assert pack_key(0, 134217727.0, 0) < pack_key(0, 134217728.0, 0)
    assert pack_key(0, 194.0, 0, maximize=True) < pack_key(0, 193.0, 0, maximize=True)


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from baryonyx_amd.dist import reduce_incumbent

    out = []
    # the highest rank holds the better (smaller) objective
    v, x, o = reduce_incumbent(100.0 - 10 * rank, np.full(5, rank + 1, np.uint64))
    out.append((v, x.tolist(), o))
    # maximise: rank 0 wins
    v, x, o = reduce_incumbent(100.0 - 10 * rank, np.full(5, rank + 1, np.uint64), maximize=True)
    out.append((v, x.tolist(), o))
    # nobody feasible
    v, x, o = reduce_incumbent(float("inf"), np.zeros(5, np.uint64))
    out.append((v, x.tolist(), o))
    # a tie goes to the lowest rank
    v, x, o = reduce_incumbent(7.0, np.full(5, 10 + rank, np.uint64))
    out.append((v, x.tolist(), o))
    # nobody feasible, but rank 1 has the fewest violated rows: it wins whatever its value
    v, x, o = reduce_incumbent(50.0 + rank, np.full(3, 20 + rank, np.uint64), remaining=9 if rank != 1 else 4)
    out.append((v, x.tolist(), o))
    # a feasible rank beats every infeasible one, and x keeps all 64 bits of a word
    v, x, o = reduce_incumbent(1e6 if rank == world - 1 else -5.0, np.full(2, 0xFFFFFFFFFFFFFFFF - rank, np.uint64),
                               remaining=0 if rank == world - 1 else 2)
    out.append((v, x.tolist(), o))
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_incumbent_reduction(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(1, world):
        assert got[r] == got[0]
    res = got[0]
    last = world - 1
    assert res[0] == (100.0 - 10 * last, [last + 1] * 5, last)
    assert res[1] == (100.0, [1] * 5, 0)
    assert res[2][0] == float("inf") and res[2][2] == 0
    assert res[3] == (7.0, [10] * 5, 0)
    assert res[4] == (51.0, [21] * 3, 1)
    assert res[5] == (1e6, [0xFFFFFFFFFFFFFFFF - last] * 2, last)
