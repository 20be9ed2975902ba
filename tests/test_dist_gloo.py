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


# ---------------------------------------------------------------------------
# RCC undrift over two frame shards == single-process undrift (same pair function)
# ---------------------------------------------------------------------------
def _cpu_pair_shifts(segments, box, roi, pairs=None):
    """The reference's get_image_shift per pair with numpy's FFT and the bounded Gaussian fit of the oracle
    (its restatement of scipy's TRF, pinned against scipy in test_oracle_golden.py) -- stands in for the GPU
    correlation and fit in the gloo tests."""
    from oracle import oracle as orc
    n, Y, X = segments.shape
    if pairs is None:
        pairs = [(i, j) for i in range(n - 1) for j in range(i + 1, n)]
    out = []
    for i, j in pairs:
        w = orc.peak_window(segments[i], segments[j], box, roi)
        if w is None or w[4] is None:
            out.append((0, 0))
        else:
            ym, xm, Y_, X_, win = w
            out.append(orc.image_shift_from_window(win, box, ym, xm, Y_, X_, Y, X))
    return out


def _cpu_render(locs, info, blur_method=None, min_blur_width=0.0):
    from oracle import oracle as orc
    return orc.render(locs["x"].to_numpy(), locs["y"].to_numpy(), 1.0, [(0, 0), (info[0]["Height"], info[0]["Width"])],
                      locs["lpx"].to_numpy(), locs["lpy"].to_numpy(), blur_method, min_blur_width)


def _undrift_inputs():
    import pandas as pd
    from conftest import golden
    g = golden("undrift_rcc")
    locs = pd.DataFrame({"frame": g["frame"], "x": g["x"], "y": g["y"], "lpx": g["lpx"], "lpy": g["lpy"]})
    info = [{"Frames": int(g["frames"]), "Height": int(g["size"]), "Width": int(g["size"]), "Pixelsize": 130}]
    return g, locs, info


def _undrift_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, locs, info = _undrift_inputs()
        lo, hi = pdist.shard_frames(info[0]["Frames"] + 333, world, rank)       # shard edges that cut through segments
        hi = info[0]["Frames"] if rank == world - 1 else hi
        shard = locs[(locs["frame"] >= lo) & (locs["frame"] < hi)]
        drift, und = pdist.undrift_sharded(shard, info, int(g["segmentation"]), render_fn=_cpu_render,
                                           pair_shift_fn=_cpu_pair_shifts)
        q.put((rank, drift["x"].tolist(), drift["y"].tolist(), und["x"].tolist(), und.index.tolist()))
    except Exception as exc:
        q.put((rank, repr(exc)))
        raise
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_undrift_sharded_world2_matches_reference_goldens():
    g, locs, info = _undrift_inputs()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_undrift_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(len(r) == 5 for r in results), results
    und_x = np.full(len(locs), np.nan)
    for rank, dx, dy, ux, idx in results:
        # every rank ends with the same drift, and it is the reference's (segments differ only by the
        # float32-vs-float64 coordinate promotion and the split sums)
        assert np.max(np.abs(np.array(dx) - g["drift_x"])) < 2e-4 and np.max(np.abs(np.array(dy) - g["drift_y"])) < 2e-4
        und_x[np.array(idx, dtype=np.int64)] = ux
    assert not np.isnan(und_x).any()                      # the two shards cover every localization exactly once
    assert np.max(np.abs(und_x - g["undrifted_x"])) < 2e-4
    # single process == sharded, with the same stand-in functions
    d1, u1 = pdist.undrift_sharded(locs, info, int(g["segmentation"]), render_fn=_cpu_render, pair_shift_fn=_cpu_pair_shifts)
    assert np.max(np.abs(d1["x"].to_numpy() - np.array(results[0][1]))) < 1e-9


def _pipelined_worker(rank, world, port, piece_counts, caps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = pdist.PipelinedTableGather()
        for i, cap in enumerate(caps):
            n = piece_counts[rank][i]
            _, table = _fake_table(10 * rank + i, min(n, cap), cap)
            g.submit(table, torch.tensor([n], dtype=torch.int64))
        out = g.finish()
        q.put((rank, None if out is None else out.tolist()))
    except Exception as exc:
        q.put((rank, repr(exc)))
        raise
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("piece_counts,caps,overflow", [(((4, 0, 7), (2, 5, 1)), (8, 6, 9), False),
                                                        (((4, 3, 7), (2, 9, 1)), (8, 6, 9), True)])
def test_pipelined_table_gather_world2(piece_counts, caps, overflow):
    """Pieces gathered one by one (asynchronous collectives, as after each frame chunk of a shard) come back
    ordered by rank, then piece; a piece that overflowed its capacity on any rank makes every rank return None."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, 2, port, piece_counts, caps, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if overflow:
        assert results[0] is None and results[1] is None
        return
    want = torch.cat([_fake_table(10 * r + i, piece_counts[r][i], caps[i])[1][:, : piece_counts[r][i]]
                      for r in range(2) for i in range(len(caps))], dim=1)
    for r in range(2):
        assert isinstance(results[r], list), results[r]
        assert torch.equal(torch.tensor(results[r], dtype=torch.int32), want)


def test_pipelined_table_gather_single_process():
    g = pdist.PipelinedTableGather()
    pieces = [(_fake_table(i, n, 10)[1], n) for i, n in enumerate((3, 0, 10))]
    for t, n in pieces:
        g.submit(t, torch.tensor([n], dtype=torch.int64))
    assert torch.equal(g.finish(), torch.cat([t[:, :n] for t, n in pieces], dim=1))
    g = pdist.PipelinedTableGather()
    g.submit(pieces[0][0], torch.tensor([11], dtype=torch.int64))
    assert g.finish() is None


# ---------------------------------------------------------------------------
# the native communicator's set-up: one rank that cannot join must not leave the others waiting
# ---------------------------------------------------------------------------
def _comm_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 1 "cannot load RCCL"; rank 0 could.  No rank may reach pmi_comm_init (here it would fail anyway: no GPU),
        # and BOTH must learn that the native path is off — twice: the verdict is cached per group.
        pdist.NativeComm.available = staticmethod(lambda: rank == 0)
        pdist.NativeComm.unique_id = staticmethod(lambda: (_ for _ in ()).throw(AssertionError("id requested although a rank is out")))
        seen = []
        for _ in range(2):
            try:
                pdist.NativeComm.for_group(None, device=torch.device("cpu"))
                seen.append("communicator")
            except pdist.NativeCommUnavailable as exc:
                seen.append(str(exc))
        # second scenario on a fresh sub-group: every rank is able, rank 0 fails to make the id -> the same verdict everywhere
        sub = dist.new_group([0, 1])
        pdist.NativeComm.available = staticmethod(lambda: True)
        pdist.NativeComm.unique_id = staticmethod(lambda: (_ for _ in ()).throw(RuntimeError("no id today")))
        try:
            pdist.NativeComm.for_group(sub, device=torch.device("cpu"))
            seen.append("communicator")
        except pdist.NativeCommUnavailable as exc:
            seen.append(str(exc))
        q.put((rank, seen))
    except Exception as exc:
        q.put((rank, repr(exc)))
        raise
    finally:
        dist.barrier()
        pdist.NativeComm.close_all()
        dist.destroy_process_group()


def test_native_comm_setup_agrees_before_any_rank_enters_rccl():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        seen = results[r]
        assert isinstance(seen, list) and len(seen) == 3, seen
        assert "cannot be loaded" in seen[0] and seen[1] == seen[0]
        assert "no RCCL unique id from rank 0" in seen[2] and "no id today" in seen[2]
