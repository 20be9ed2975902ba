"""Frame sharding across the GPUs of one node and the all-gather of the result.

The reference has no distributed code: its workers split the movie frame by
frame (picasso/localize.py:438-454) inside one process.  Here every GPU (one
process per GPU, torch.distributed, backend "nccl" = RCCL over xGMI) owns a
contiguous frame range — a localization depends on one frame only, so there is
no halo and no collective on the data path.  The only exchange is the
all-gather of the localization table at the end: counts first,
then the rows padded to the largest count.  Contiguous shards keep the gathered
table frame-sorted, which is the order picasso/gaussmle.py:1036 produces.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .backend import LOC_COLUMNS


def shard_frames(n_frames: int, world: int, rank: int) -> Tuple[int, int]:
    """Half-open frame range [lo, hi) of `rank`; ranges are contiguous, disjoint,
    cover [0, n_frames) and differ in length by at most one frame."""
    base, rem = divmod(int(n_frames), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def allgather_table(table: torch.Tensor, n_rows, group=None) -> torch.Tensor:
    """All-gather a column-major table (C columns x capacity, 4-byte cells, first
    `n_rows` of each column valid) from every rank.  Returns a (C, total) tensor
    on every rank, rows in rank order."""
    world = dist.get_world_size(group)
    C = table.shape[0]
    n = torch.as_tensor([int(n_rows)], dtype=torch.int64, device=table.device)
    counts = torch.empty(world, dtype=torch.int64, device=table.device)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts_h = counts.cpu().tolist()
    pad = max(max(counts_h), 1)
    send = torch.zeros((C, pad), dtype=table.dtype, device=table.device)
    send[:, : int(n_rows)] = table[:, : int(n_rows)]
    recv = torch.empty((world * C, pad), dtype=table.dtype, device=table.device)   # dim-0 concatenation
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, C, pad)
    return torch.cat([recv[r, :, : counts_h[r]] for r in range(world)], dim=1)


def table_to_columns(table: torch.Tensor) -> Dict[str, np.ndarray]:
    """(17, n) int32 storage -> named numpy columns with the table dtypes."""
    host = table.cpu().numpy()
    return {name: host[c].view(dt).copy() for c, (name, dt) in enumerate(LOC_COLUMNS)}


def columns_to_table(cols: Dict[str, np.ndarray], device="cpu") -> torch.Tensor:
    n = len(cols["frame"])
    out = np.empty((len(LOC_COLUMNS), n), np.int32)
    for c, (name, dt) in enumerate(LOC_COLUMNS):
        out[c] = np.ascontiguousarray(cols[name], dtype=dt).view(np.int32)
    return torch.from_numpy(out).to(device)


def localize_sharded(movie_shard: torch.Tensor, first_frame: int, camera_info: dict, parameters: dict, *,
                     eps: float = 1e-3, max_it: int = 100, mle_method: str = "sigmaxy", group=None):
    """Each rank localizes its resident shard (uint16 CUDA tensor of its frames),
    then all ranks receive the whole table.  `first_frame` is the label of the
    shard's first frame (from shard_frames)."""
    import ctypes

    from . import _lib
    L = _lib.load()
    _lib.require_gpu()
    F, H, W = movie_shard.shape
    cap = max(4096, 400 * F)
    dev = movie_shard.device
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    while True:
        table = torch.empty((len(LOC_COLUMNS), cap), dtype=torch.int32, device=dev)
        d_n = torch.zeros(1, dtype=torch.int64, device=dev)
        rc = L.pmi_localize_mle_dev(ctypes.c_void_p(movie_shard.data_ptr()), 0, F, H, W,
                                    int(parameters["Box Size"]), float(parameters["Min. Net Gradient"]), None,
                                    0, F - 1, float(camera_info["Baseline"]), float(camera_info["Sensitivity"]),
                                    float(camera_info["Gain"]), float(eps), int(max_it),
                                    _lib.MLE_METHODS[mle_method], ctypes.c_void_p(table.data_ptr()), cap,
                                    ctypes.c_void_p(d_n.data_ptr()), stream)
        _lib.check(rc, "pmi_localize_mle_dev")
        n = int(d_n.item())
        if n <= cap:
            break
        cap = n
    table[0, :n] += int(first_frame)        # shard-local frame index -> movie frame label
    return table_to_columns(allgather_table(table, n, group))
