"""CPU tier: the N>1 path (frame sharding + table all-gather) with gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from picasso_amd import dist as pdist
from picasso_amd.backend import LOC_COLUMNS


def test_shard_frames_partition():
    for F in (0, 1, 7, 100, 10000, 200001):
        for world in (1, 2, 3, 8):
            ranges = [pdist.shard_frames(F, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == F
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1


def _fake_table(rank, n, cap):
    rng = np.random.default_rng(100 + rank)
    cols = {}
    for name, dt in LOC_COLUMNS:
        if name == "frame":
            cols[name] = np.sort(rng.integers(rank * 1000, rank * 1000 + 1000, n)).astype(dt)
        elif name == "iterations":
            cols[name] = rng.integers(1, 100, n).astype(dt)
        else:
            cols[name] = rng.normal(size=n).astype(dt)
    t = torch.zeros((len(LOC_COLUMNS), cap), dtype=torch.int32)
    t[:, :n] = pdist.columns_to_table(cols)
    return cols, t


def _worker(rank, world, port, counts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cols, table = _fake_table(rank, counts[rank], cap=counts[rank] + 17)
        out = pdist.table_to_columns(pdist.allgather_table(table, counts[rank]))
        q.put((rank, {k: v.tolist() for k, v in out.items()}))
    except Exception as exc:          # report instead of leaving the parent waiting
        q.put((rank, repr(exc)))
        raise
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("counts", [(5, 9), (0, 4), (3, 0)])
def test_allgather_table_world2(counts):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, counts, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=60) for _ in range(2))
    assert all(isinstance(v, dict) for v in results.values()), results
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = {name: np.concatenate([_fake_table(r, counts[r], counts[r] + 1)[0][name] for r in range(2)])
            for name, _ in LOC_COLUMNS}
    for r in range(2):                         # every rank holds the whole table, in rank (= frame) order
        for name, dt in LOC_COLUMNS:
            got = np.array(results[r][name], dtype=dt)
            assert np.array_equal(got, want[name]), (r, name)
        assert np.all(np.diff(np.array(results[r]["frame"], dtype=np.int64)) >= 0)
