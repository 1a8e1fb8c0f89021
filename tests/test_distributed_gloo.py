"""The N > 1 exchange steps of the sharded hot path on CPU: world_size 2, gloo backend."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from robo_amd import sharding
    # candidate shard: global acquisition vector with a tie across the shard boundary and a NaN case
    M = 1000
    y = np.random.RandomState(5).randint(0, 50, size=M).astype(float)
    y[[123, 777]] = 99.0                        # tie: the lower global index must win
    b, e = sharding.shard_range(M, rank, world)
    j = int(np.argmax(y[b:e]))
    v, i = sharding.allgather_argmax(y[b + j], b + j)
    assert (v, i) == (99.0, 123), (v, i)
    y[900] = np.nan
    j = int(np.argmax(y[b:e]))
    v, i = sharding.allgather_argmax(y[b + j], b + j)
    assert i == 900 and np.isnan(v)
    # sample shard: rank-ordered sum of partial acquisition sums, identical on every rank
    S = 7
    acq = np.random.RandomState(6).rand(S, 64)
    sb, se = sharding.shard_range(S, rank, world)
    total = sharding.allgather_ordered_sum(acq[sb:se].sum(axis=0))
    np.testing.assert_allclose(total, acq.sum(axis=0), rtol=1e-14)
    np.save(os.path.join(out_dir, "total_%d.npy" % rank), total)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_exchange_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "total_0.npy")
    b = np.load(tmp_path / "total_1.npy")
    np.testing.assert_array_equal(a, b)        # bit-identical on both ranks
