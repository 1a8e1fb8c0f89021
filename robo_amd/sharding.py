"""Multi-GPU sharding of the two independent axes of the hot path.

One process per GPU.  No data-path collective: every rank holds a full replica of the fitted GP (the fit is
deterministic, so running it on every rank is cheaper than broadcasting L) and evaluates its own contiguous shard.
The exchanges run INSIDE librobo_hip.so (robo_amd/csrc/comm.hip: RCCL all-gathers on the library's stream, device
pointers on both sides) through a :class:`robo_amd._lib.Comm`:

* candidate shard:  all-gather of one (max, global index, flags) triple per rank, then the np.argmax tie-break
  (lowest global index, NaN maximal) -- ``robo_acq_eval_cand_sharded``;
* sample shard:     all-gather of the per-rank partial sums of M acquisition values and a RANK-ORDERED sum:
  deterministic and identical on every rank (an all-reduce SUM guarantees neither) --
  ``robo_acq_eval_marginal_cand_sharded``.  It equals the single-GPU sample-order accumulation of
  MarginalizationGPMCMC.compute up to fp64 re-association (partial sums are formed per shard).

The communicator needs one 128-byte id on every rank.  ``init_comm`` takes it from the caller (any launcher);
without that call, a process whose ``torch.distributed`` default group is initialised gets its communicator on first
use, the id travelling through ``broadcast_object_list`` -- torch is then rendezvous plumbing only, nothing of the
data path touches it.
"""
import numpy as np

_comm = None            # this process's library communicator (robo_amd._lib.Comm)


def init_comm(rank, world, comm_id, ctx=None):
    """Join the job: rank in [0, world), comm_id = the bytes rank 0 got from ``robo_amd._lib.Comm.create_id()``."""
    global _comm
    from robo_amd import _lib
    if _comm is not None:
        _comm.close()
    _comm = _lib.Comm(ctx if ctx is not None else _lib.default_context(), rank, world, comm_id)
    return _comm


def close_comm():
    global _comm
    if _comm is not None:
        _comm.close()
        _comm = None


def _torch_group():
    """(rank, world) of an initialised torch.distributed default group, else None -- without importing torch: a
    group can only have been initialised by code that already imported torch.distributed"""
    import sys
    dist = sys.modules.get("torch.distributed")
    if dist is None or not (dist.is_available() and dist.is_initialized()):
        return None
    return dist.get_rank(), dist.get_world_size()


def comm():
    """the library communicator of this process, or None in a single-process run"""
    global _comm
    if _comm is not None:
        return _comm if _comm.world > 1 else None
    tg = _torch_group()
    if tg is None or tg[1] == 1:
        return None
    import torch.distributed as dist
    from robo_amd import _lib
    box = [_lib.Comm.create_id() if tg[0] == 0 else None]
    dist.broadcast_object_list(box, src=0)              # rendezvous only: 128 bytes, once per process
    return init_comm(tg[0], tg[1], box[0])


def dist_info():
    """(communicator or None, rank, world_size)"""
    c = comm()
    if c is None:
        return None, 0, 1
    return c, c.rank, c.world


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


def allgather_argmax(local_max, local_global_index):
    """Exchange the per-shard incumbents of values computed OUTSIDE the fused entry point (any acquisition function,
    e.g. the information gain) -> global (max, argmax).  The index travels as an exact float64 (< 2^53)."""
    c = comm()
    if c is None:
        return float(local_max), int(local_global_index)
    rows = c.allgather([float(local_max), float(local_global_index)])
    return reduce_argmax((float(r[0]), int(r[1])) for r in rows)


BEST_MSG_LEN = 4        # (max, global index or -1, flag word, status of the sender's local half): comm.hip BEST_MSG


def exchange_best(comm, compute_local):
    """The host-side form of the candidate shard's exchange, for the cases the fused library call does not cover (an
    empty shard, out-of-box candidates, a non-device model): ``compute_local()`` -> (value, global index) or None for an
    empty shard.  ONE place builds and parses the four-double message the library's own exchange carries (comm.hip
    comm_pack_best_kernel / comm_best_kernel); a rank whose compute_local() raises still joins the collective with its
    status set -- the other ranks get an error instead of a hang -- and then re-raises.
    -> (global max, global argmax, flags OR-ed)"""
    from robo_amd import _lib
    msg, err = [0.0, -1.0, 0.0, 0.0], None
    try:
        local = compute_local()
        if local is not None:
            msg[0], msg[1] = float(local[0]), float(local[1])
            if len(local) > 2:
                msg[2] = float(local[2])
    except Exception as e:      # noqa: BLE001 -- the collective below must be issued on every rank
        msg, err = [0.0, -1.0, 0.0, float(_lib.RUNTIME_ERROR)], e
    rows = comm.allgather(msg)
    if err is not None:
        raise err
    flags = 0
    for r in rows:
        flags |= int(r[2])
        if int(r[3]) != _lib.OK:
            _lib.check(int(r[3]), "the local half of another rank's shard failed (status %d)" % int(r[3]))
    best = reduce_argmax((float(r[0]), int(r[1])) for r in rows)
    mx, am = best if best is not None else (0.0, -1)
    return mx, am, flags


def allgather_ordered_sum(partial_sum):
    """Sum per-rank partial vectors in rank order (deterministic on every rank) -- the generic form for quantities
    that are not acquisition sums (the mixture posterior of GaussianProcessMCMC.predict)."""
    c = comm()
    part = np.ascontiguousarray(partial_sum, dtype=np.float64)
    if c is None:
        return part
    rows = c.allgather(part.reshape(-1))
    total = rows[0].copy()
    for r in rows[1:]:
        total += r
    return total.reshape(part.shape)


def allgather_rows(row):
    """every rank contributes one fp64 row (D,) -> (world, D), identical on every rank"""
    c = comm()
    if c is None:
        return np.asarray(row, dtype=np.float64)[None, :]
    return c.allgather(row)


def assert_replicated(what, values):
    """Every rank passes the same small vector of numbers, or ALL ranks raise: the check in front of a sharded
    maximisation (ranks that drifted apart -- other seeds, another number of maximize() calls -- would otherwise
    dead-lock in the exchange or score a point against another rank's model)."""
    c = comm()
    if c is None:
        return
    rows = allgather_rows(np.asarray(values, dtype=np.float64))
    if not all(np.array_equal(rows[0], r, equal_nan=True) for r in rows[1:]):
        raise RuntimeError("%s differ across ranks (rank %d of %d holds %r): sharded maximisation needs identical "
                           "seeds and call sequences on every rank" % (what, c.rank, c.world, list(values)))


def sharded_argmax(acq, X):
    """Candidate shard of one acquisition maximisation (SURVEY.md 8e axis 1): every rank evaluates its contiguous
    slice of the SAME candidate matrix X against its own replica of the model and the per-shard incumbents are
    exchanged.  Returns the global np.argmax index, identical on every rank and identical to the single-process
    result (values are computed per candidate, so sharding does not change them).  Closed-form acquisitions on a
    device GP go through the fused entry point (posterior, acquisition, local argmax, RCCL all-gather and the
    cross-rank tie-break in one library call)."""
    c, rank, world = dist_info()
    if world == 1:
        return int(acq.argmax(X)) if hasattr(acq, "argmax") else int(np.argmax(acq(X)))
    b, e = shard_range(X.shape[0], rank, world)
    # which exchange runs is decided by the acquisition's CLASS, never by this rank's shard: a rank whose slice is empty
    # (fewer candidates than ranks) must issue the same collective as the others
    if hasattr(acq, "argmax_sharded"):
        return acq.argmax_sharded(c, X[b:e], b)
    if e > b:
        vals = np.asarray(acq(X[b:e]), dtype=np.float64).reshape(-1)
        if vals.shape[0] != e - b:          # EI's whole-batch collapse to [[0]] (ei.py:72-74): every value is 0
            vals = np.zeros(e - b)
        j = int(np.argmax(vals))
        local = (float(vals[j]), b + j)
    else:
        local = (-np.inf, -1)
    return allgather_argmax(local[0], local[1])[1]
