"""CPU, world_size 2, gloo: the exchange step of the multi-GPU restart pool
(min-reduce of the incumbent + broadcast of the winner's x)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from baryonyx_amd.dist import reduce_incumbent

    out = []
    # rank 1 holds the better (smaller) objective
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
    q.put((rank, out))
    dist.destroy_process_group()


def test_incumbent_reduction_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1]
    res = got[0]
    assert res[0] == (90.0, [2] * 5, 1)
    assert res[1] == (100.0, [1] * 5, 0)
    assert res[2][0] == float("inf")
    assert res[3] == (7.0, [10] * 5, 0)
