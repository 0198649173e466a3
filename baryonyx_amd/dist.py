"""Multi-GPU exchange step of the restart pool: every rank packs its incumbent
into ONE 63-bit key -- (violated rows, order-preserving image of the objective,
rank) -- and a single RCCL all-reduce(MIN) over xGMI names the winner on every
rank at once; the winner's x (and its exact objective) then travel in one more
all-reduce in which only the winner contributes.  No host round trip between
the two collectives (the message is 8 B, latency-bound).
Restarts are sharded over ranks with no data-path exchange; the reference
shares its population through a mutex (itm-optimizer-common.hpp:97-99,
804-857) -- this is the one-process-per-GPU counterpart, used by bench.py.
(The product's baryonyx::optimize drives several devices from ONE process and
needs no collective: DESIGN.md section 6.)"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

REMAINING_BITS, OBJECTIVE_BITS, RANK_BITS = 19, 38, 6  # 63 bits: the key stays a non-negative int64
REMAINING_MAX = (1 << REMAINING_BITS) - 1


def ordered_bits(value: float, maximize: bool = False) -> int:
    """Order-preserving 64-bit image of a double (smaller value <-> smaller
    unsigned image; complemented when maximising so that MIN picks the best)."""
    b = int(np.float64(value).view(np.uint64))
    b = (~b & 0xFFFFFFFFFFFFFFFF) if b >> 63 else (b | (1 << 63))
    return (~b & 0xFFFFFFFFFFFFFFFF) if maximize else b


def pack_key(remaining: int, value: float, rank: int, maximize: bool = False) -> int:
    """(remaining << 44) | (top 38 bits of the objective's image << 6) | rank.
    38 bits keep the sign, the exponent and 26 mantissa bits: integer objectives
    below 2**27 compare exactly; closer values fall to the lower rank."""
    r = min(max(int(remaining), 0), REMAINING_MAX)
    o = ordered_bits(value, maximize) >> (64 - OBJECTIVE_BITS)
    return (r << (OBJECTIVE_BITS + RANK_BITS)) | (o << RANK_BITS) | (rank & ((1 << RANK_BITS) - 1))


def reduce_incumbent(value: float, x_words: np.ndarray, maximize: bool = False, device=None,
                     remaining: int | None = None):
    """All ranks pass their best (value, x[, violated rows]); all ranks get the
    global best.  `value` is +inf / -inf when the rank has no feasible point
    (`remaining` then defaults to "as bad as it gets").  Returns (best_value,
    best_x_words, owner_rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value), np.array(x_words, copy=True), 0
    world, rank = dist.get_world_size(), dist.get_rank()
    if world > (1 << RANK_BITS):
        raise ValueError("reduce_incumbent: more ranks than the key has room for")
    dev = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")
    if remaining is None:
        remaining = 0 if np.isfinite(value) else REMAINING_MAX
    mine = pack_key(remaining, value, rank, maximize)
    key = torch.tensor([mine], dtype=torch.int64, device=dev)
    dist.all_reduce(key, op=dist.ReduceOp.MIN)  # collective 1: who holds the incumbent
    # collective 2: the winner's x and exact objective; everybody else adds zeros.  The
    # mask is formed on the device from the reduced key: no host sync in between.
    payload = np.concatenate([np.ascontiguousarray(x_words, dtype=np.uint64),
                              np.array([np.float64(value)]).view(np.uint64)]).view(np.int64)
    t = torch.from_numpy(payload.copy()).to(dev)
    t = t * (key == mine).to(torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    out = t.cpu().numpy().view(np.uint64)
    owner = int(key.item()) & ((1 << RANK_BITS) - 1)
    return float(out[-1:].view(np.float64)[0]), out[:-1].copy(), owner
