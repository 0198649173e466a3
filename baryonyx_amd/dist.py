"""Multi-GPU exchange step of the restart pool.  Restarts are sharded over the
ranks with no data-path exchange; every T sweeps the ranks agree on the
incumbent, as the reference's workers do through the mutex of
best_solution_recorder (lib/src/itm-optimizer-common.hpp:97-99, :804-857):

* every rank's best point is described by a KEY of two non-negative int64 words
  compared lexicographically --
      hi = violated rows (19 bits) << 44 | top 44 bits of the objective's ordered image
      lo = low 20 bits of that image   <<  6 | rank (6 bits)
  (the layout of baryonyx_amd/csrc/bxh_incumbent.hpp, which builds the key on the
  device; `pack_key` below is the same rule on the host);
* ONE RCCL all-reduce(MIN) of `hi` over xGMI settles every comparison that differs
  in the count of violated rows or in the first 44 bits of the objective; a second
  all-reduce(MIN) of `lo`, in which only the ranks that tie in `hi` take part,
  makes the comparison EXACT in all 64 bits of the objective (storage::sort
  compares exact values, itm-optimizer-common.hpp:424-442) and picks the lowest
  rank among equals;
* the winner's x and its exact objective then travel in one more all-reduce in
  which only the winner contributes.

Everything between the collectives is a device-side operation on 8-byte tensors:
no host round trip, nothing is read back until the caller asks (`read_record`).
(The product's baryonyx::optimize drives several devices from ONE process and
needs no collective: DESIGN.md section 6.)"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

REMAINING_BITS, RANK_BITS = 19, 6
HI_OBJECTIVE_BITS, LO_OBJECTIVE_BITS = 44, 20
REMAINING_MAX = (1 << REMAINING_BITS) - 1
RANK_MASK = (1 << RANK_BITS) - 1
I64_MAX = (1 << 63) - 1


def ordered_bits(value: float, maximize: bool = False) -> int:
    """Order-preserving 64-bit image of a double (smaller value <-> smaller
    unsigned image; complemented when maximising so that MIN picks the best)."""
    b = int(np.float64(value).view(np.uint64))
    b = (~b & 0xFFFFFFFFFFFFFFFF) if b >> 63 else (b | (1 << 63))
    return (~b & 0xFFFFFFFFFFFFFFFF) if maximize else b


def pack_key(remaining: int, value: float, rank: int, maximize: bool = False) -> tuple[int, int]:
    """(hi, lo) of a point with `remaining` violated rows and objective `value`
    held by `rank`: bxh::incumbent_pack on the host.  All 64 bits of the
    objective's image are in the pair: no two different objectives share a key."""
    r = min(max(int(remaining), 0), REMAINING_MAX)
    o = ordered_bits(value, maximize)
    hi = (r << HI_OBJECTIVE_BITS) | (o >> LO_OBJECTIVE_BITS)
    lo = ((o & ((1 << LO_OBJECTIVE_BITS) - 1)) << RANK_BITS) | (rank & RANK_MASK)
    return hi, lo


def key_remaining(hi: int) -> int:
    return int(hi) >> HI_OBJECTIVE_BITS


def _world():
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(), dist.get_rank()


def record_words(nwords: int) -> int:
    """int64 words of an incumbent record: x blocks, key hi, key lo, the bits of the
    objective, (violated rows | instance << 32) -- bxh_incumbent writes the x part
    and the 32-byte bxh_incumbent_record behind it."""
    return nwords + 4


def read_record(rec: torch.Tensor):
    """-> (value, x words, owner rank, violated rows): the one read-back."""
    p = rec.cpu().numpy()
    n = p.shape[0] - 4
    return (float(p[n + 2:n + 3].view(np.float64)[0]), p[:n].view(np.uint64).copy(), int(p[n + 1]) & RANK_MASK,
            int(p[n + 3]) & 0xFFFFFFFF)


def make_record(value: float, x_words: np.ndarray, remaining: int, rank: int, maximize: bool = False,
                device="cpu") -> torch.Tensor:
    """A record built on the host (tests, callers without bxh_incumbent)."""
    hi, lo = pack_key(remaining, value, rank, maximize)
    tail = np.array([hi, lo, int(np.float64(value).view(np.int64)), int(remaining) & 0xFFFFFFFF], np.int64)
    body = np.ascontiguousarray(x_words, dtype=np.uint64).view(np.int64)
    return torch.from_numpy(np.concatenate([body, tail])).to(device)


def reduce_incumbent_device(rec: torch.Tensor, carried: torch.Tensor | None = None) -> torch.Tensor:
    """rec: this rank's best point as an int64[record_words(nwords)] tensor on the
    collective's device (bxh_incumbent fills it in place).  `carried`: the result
    of the previous exchange; this rank offers it again when it is at least as
    good as its new point -- compared on the device, keys without their rank
    field -- so the incumbent never gets worse.  Returns the winner's record,
    identical on every rank, still on the device: nothing is read back here."""
    world, rank = _world()
    if world > (1 << RANK_BITS):
        raise ValueError("reduce_incumbent: more ranks than the key has room for")
    n = rec.shape[0] - 4
    if carried is not None:
        c_hi, c_lo = carried[n:n + 1], carried[n + 1:n + 2] >> RANK_BITS
        n_hi, n_lo = rec[n:n + 1], rec[n + 1:n + 2] >> RANK_BITS
        keep = (c_hi < n_hi) | ((c_hi == n_hi) & (c_lo <= n_lo))
        rec = torch.where(keep, carried, rec)
        rec[n + 1] = ((rec[n + 1] >> RANK_BITS) << RANK_BITS) | rank  # whoever offers it, holds it
    if world == 1:
        return rec.clone()
    hi = rec[n:n + 1].clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MIN)  # collective 1: violated rows, 44 bits of the objective
    tie = rec[n:n + 1] == hi
    lo = torch.where(tie, rec[n + 1:n + 2], torch.full_like(hi, I64_MAX))
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)  # collective 2: the other 20 bits, then the rank
    mine = tie & (rec[n + 1:n + 2] == lo)
    out = rec * mine.to(torch.int64)
    dist.all_reduce(out, op=dist.ReduceOp.SUM)  # collective 3: the winner's x, key and exact objective
    return out


def reduce_incumbent(value: float, x_words: np.ndarray, maximize: bool = False, device=None,
                     remaining: int | None = None):
    """Host-side convenience over reduce_incumbent_device: all ranks pass their
    best (value, x[, violated rows]); all ranks get the global best.  `value` is
    +inf / -inf when the rank has no feasible point (`remaining` then defaults to
    "as bad as it gets").  Returns (best_value, best_x_words, owner_rank)."""
    world, rank = _world()
    if world == 1:
        return float(value), np.array(x_words, copy=True), 0
    dev = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")
    if remaining is None:
        remaining = 0 if np.isfinite(value) else REMAINING_MAX
    rec = make_record(value, x_words, remaining, rank, maximize, dev)
    best, xw, owner, _ = read_record(reduce_incumbent_device(rec))
    return best, xw, owner
