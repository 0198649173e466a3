"""CPU, gloo, world sizes 2, 4 and 8: the exchange step of the multi-GPU restart
pool -- all-reduce(MIN) of the key's first word (violated rows, 44 bits of the
objective), a second one of its second word among the ranks that tie (the other
20 bits, then the rank: the comparison is exact), then the winner's x and exact
objective in one masked all-reduce (baryonyx_amd/dist.py)."""
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
    then the better objective -- exactly, in every bit --; the rank breaks what is
    left.  A key is the pair (hi, lo), compared lexicographically."""
    vals = [-1e30, -7.5, -1.0, -0.0, 0.0, 1e-300, 1.0, 193.0, 194.0, 7.3e6, 1e30, float("inf")]
    imgs = [ordered_bits(v) for v in vals]
    assert imgs == sorted(imgs) and imgs[3] < imgs[4]  # -0.0 sorts in front of +0.0: harmless
    assert [ordered_bits(v, True) for v in vals] == sorted((ordered_bits(v, True) for v in vals), reverse=True)
    # remaining dominates the objective, the objective dominates the rank
    assert pack_key(0, 1e30, 63) < pack_key(1, -1e30, 0)
    assert pack_key(3, 193.0, 5) < pack_key(3, 194.0, 0)
    assert pack_key(3, 193.0, 1) < pack_key(3, 193.0, 2)
    assert pack_key(10 ** 9, 0.0, 0) == pack_key(REMAINING_MAX, 0.0, 0)  # clipped
    hi, lo = pack_key(REMAINING_MAX, float("inf"), 63)
    assert 0 <= hi < 2 ** 63 and 0 <= lo < 2 ** 63                      # non-negative int64 words
    hi, lo = pack_key(REMAINING_MAX, -0.0, 63, maximize=True)
    assert 0 <= hi < 2 ** 63 and 0 <= lo < 2 ** 63
    assert pack_key(0, 194.0, 0, maximize=True) < pack_key(0, 193.0, 0, maximize=True)
    # EXACT: neighbouring doubles differ, whatever their size (the one-word key of round 4
    # kept 26 mantissa bits: integer objectives from 2**27 on, or close real costs, tied)
    for v in (134217727.0, 2.0 ** 40, 7.3e6 + 0.125, 1e300, -1e-300, -5404140.0):
        up = float(np.nextafter(np.float64(v), np.inf))
        assert pack_key(0, v, 63) < pack_key(0, up, 0)
        assert pack_key(0, up, 63, maximize=True) < pack_key(0, v, 0, maximize=True)
        assert pack_key(0, v, 0)[0] <= pack_key(0, up, 0)[0]  # the first word never orders them wrongly


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
    # objectives one ulp apart at 1e12 (the first key word ties): the better VALUE wins, here
    # the highest rank's, not the lowest rank
    base = np.float64(1e12)
    mine = float(base) if rank == world - 1 else float(np.nextafter(base, np.inf))
    v, x, o = reduce_incumbent(mine, np.full(2, 30 + rank, np.uint64))
    out.append((v, x.tolist(), o))
    # the carried incumbent (device form): exchange 1 -- nobody feasible, rank 0 has the fewest
    # violated rows; exchange 2 -- still nobody feasible, every rank's new point is worse: the
    # carried point must survive with ITS count (round 4 re-offered it as "0 violated rows");
    # exchange 3 -- the last rank turns feasible and wins
    from baryonyx_amd.dist import make_record, read_record, reduce_incumbent_device
    inf = float("inf")
    inc = reduce_incumbent_device(make_record(inf, np.full(2, 40 + rank, np.uint64), 3 + rank, rank))
    out.append(read_record(inc)[0:1] + (read_record(inc)[1].tolist(),) + read_record(inc)[2:])
    inc = reduce_incumbent_device(make_record(inf, np.full(2, 50 + rank, np.uint64), 7 + rank, rank), carried=inc)
    out.append(read_record(inc)[0:1] + (read_record(inc)[1].tolist(),) + read_record(inc)[2:])
    inc = reduce_incumbent_device(make_record(12.5 if rank == world - 1 else inf, np.full(2, 60 + rank, np.uint64),
                                              0 if rank == world - 1 else 5, rank), carried=inc)
    out.append(read_record(inc)[0:1] + (read_record(inc)[1].tolist(),) + read_record(inc)[2:])
    # ... and a later, worse feasible point everywhere does not displace it
    inc = reduce_incumbent_device(make_record(13.0, np.full(2, 70 + rank, np.uint64), 0, rank), carried=inc)
    out.append(read_record(inc)[0:1] + (read_record(inc)[1].tolist(),) + (read_record(inc)[3],))
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
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
    assert res[6] == (1e12, [30 + last] * 2, last)
    inf = float("inf")
    assert res[7] == (inf, [40] * 2, 0, 3)
    assert res[8][0] == inf and res[8][1] == [40] * 2 and res[8][3] == 3  # the carried point, its own count
    assert res[9][:2] == (12.5, [60 + last] * 2) and res[9][2] == last and res[9][3] == 0
    assert res[10] == (12.5, [60 + last] * 2, 0)
