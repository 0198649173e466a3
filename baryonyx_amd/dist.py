"""Multi-GPU exchange step of the restart pool: the incumbent objective is
min-reduced (max-reduced when maximising) over the ranks with RCCL, then the
owner broadcasts its x -- the only collective on the path (restarts are
sharded over ranks with no data-path exchange).  The reference
shares its population through a mutex (itm-optimizer-common.hpp:97-99); this
is the one-process-per-GPU counterpart."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def reduce_incumbent(value: float, x_words: np.ndarray, maximize: bool = False, device=None):
    """All ranks pass their best (value, x); all ranks get the global best.
    `value` is +inf / -inf when the rank has no feasible point.  Returns
    (best_value, best_x_words, owner_rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value), np.array(x_words, copy=True), 0
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")
    v = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(v, op=dist.ReduceOp.MAX if maximize else dist.ReduceOp.MIN)
    best = float(v.item())
    # lowest rank holding the best value owns the broadcast
    mine = rank if float(value) == best else world
    o = torch.tensor([mine], dtype=torch.int64, device=dev)
    dist.all_reduce(o, op=dist.ReduceOp.MIN)
    owner = int(o.item()) % world
    xw = torch.from_numpy(np.ascontiguousarray(x_words).view(np.int64).copy()).to(dev)
    dist.broadcast(xw, src=owner)
    return best, xw.cpu().numpy().view(np.uint64), owner
