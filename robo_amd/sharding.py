"""Multi-GPU sharding of the two independent axes of the hot path.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo"
in the CPU tests).  No data-path collective: every rank holds a full replica of the fitted
GP (the fit is deterministic, so running it on every rank is cheaper than broadcasting L)
and evaluates its own contiguous shard.  The only exchanges are

* candidate shard:  all-gather of one (max, global index) pair per rank, then the same
  np.argmax tie-break on every rank (lowest global index, NaN maximal);
* sample shard:     all-gather of the per-rank partial sums of M acquisition values and a
  RANK-ORDERED local sum: deterministic and identical on every rank (an all-reduce SUM
  guarantees neither).  It equals the single-GPU sample-order accumulation of
  MarginalizationGPMCMC.compute up to fp64 re-association (partial sums are formed per shard).
"""
import numpy as np


def dist_info():
    """(torch.distributed module or None, rank, world_size) of the initialised default process group"""
    # torch is plumbing for multi-GPU runs only: a process group can only have been initialised by code that already
    # imported torch.distributed, so a single-process run never pays the ~1 s import (it used to, on the first
    # maximize() of every BO run)
    import sys
    dist = sys.modules.get("torch.distributed")
    if dist is None:
        return None, 0, 1
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def shard_range(n_items, rank, world):
    """Contiguous [begin, end) of rank's shard; the first n_items % world ranks get one more
    (config 3 of BASELINE.json: 50 samples over 4 GPUs -> 13/13/12/12)."""
    base, rem = divmod(int(n_items), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def better(v_a, i_a, v_b, i_b):
    """np.argmax order: NaN beats everything, then larger value, then lower index."""
    a_nan, b_nan = np.isnan(v_a), np.isnan(v_b)
    if a_nan != b_nan:
        return bool(a_nan)
    if not a_nan and v_a != v_b:
        return bool(v_a > v_b)
    return i_a < i_b


def reduce_argmax(pairs):
    """pairs: iterable of (value, global_index) -> the winning (value, index)."""
    best = None
    for v, i in pairs:
        if i < 0:
            continue
        if best is None or better(v, i, best[0], best[1]):
            best = (float(v), int(i))
    return best


def allgather_argmax(local_max, local_global_index, device=None):
    """Exchange the per-shard incumbents (16 B per rank) -> global (max, argmax)."""
    if dist_info()[2] == 1:
        return float(local_max), int(local_global_index)
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    # the index travels as an exact float64 (< 2^53) next to the value: one 16-byte message
    mine = torch.tensor([float(local_max), float(local_global_index)], dtype=torch.float64, device=dev)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    pairs = [(float(t[0].item()), int(t[1].item())) for t in out]
    return reduce_argmax(pairs)


def allgather_ordered_sum(partial_sum, device=None):
    """Sum per-rank partial acquisition sums in rank order (deterministic on every rank)."""
    if dist_info()[2] == 1:
        return np.asarray(partial_sum, dtype=np.float64)
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    mine = torch.as_tensor(np.ascontiguousarray(partial_sum, dtype=np.float64)).to(dev)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    total = out[0].clone()
    for t in out[1:]:
        total += t
    return total.cpu().numpy()


def allgather_rows(row, device=None):
    """every rank contributes one fp64 row (D,) -> (world, D), identical on every rank"""
    if dist_info()[2] == 1:
        return np.asarray(row, dtype=np.float64)[None, :]
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    mine = torch.as_tensor(np.ascontiguousarray(row, dtype=np.float64)).to(dev)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return np.stack([t.cpu().numpy() for t in out])


def assert_replicated(what, values):
    """Every rank passes the same small vector of numbers, or ALL ranks raise: the check in front of a sharded
    maximisation (ranks that drifted apart -- other seeds, another number of maximize() calls -- would otherwise
    dead-lock in the exchange or score a point against another rank's model)."""
    _, rank, world = dist_info()
    if world == 1:
        return
    rows = allgather_rows(np.asarray(values, dtype=np.float64))
    if not all(np.array_equal(rows[0], r, equal_nan=True) for r in rows[1:]):
        raise RuntimeError("%s differ across ranks (rank %d of %d holds %r): sharded maximisation needs identical "
                           "seeds and call sequences on every rank" % (what, rank, world, list(values)))


def sharded_argmax(acq, X):
    """Candidate shard of one acquisition maximisation (SURVEY.md 8e axis 1): every rank evaluates its contiguous
    slice of the SAME candidate matrix X against its own replica of the model and the per-shard incumbents are
    exchanged (16 B per rank).  Returns the global np.argmax index, identical on every rank and identical to the
    single-process result (values are computed per candidate, so sharding does not change them)."""
    _, rank, world = dist_info()
    if world == 1:
        return int(acq.argmax(X)) if hasattr(acq, "argmax") else int(np.argmax(acq(X)))
    b, e = shard_range(X.shape[0], rank, world)
    if e > b:
        vals = np.asarray(acq(X[b:e]), dtype=np.float64).reshape(-1)
        if vals.shape[0] != e - b:          # EI's whole-batch collapse to [[0]] (ei.py:72-74): every value is 0
            vals = np.zeros(e - b)
        j = int(np.argmax(vals))
        local = (float(vals[j]), b + j)
    else:
        local = (-np.inf, -1)
    return allgather_argmax(local[0], local[1])[1]
