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
    on every rank, rows in rank order.  Without a process group (one process, one GPU) it is
    the identity on the valid rows."""
    if not (dist.is_available() and dist.is_initialized()):
        return table[:, : int(n_rows)].clone()
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


# ---------------------------------------------------------------------------
# RCC undrift over frame shards (SURVEY 8e "undrift at scale"; picasso/postprocess.py:2900-2961)
# ---------------------------------------------------------------------------
def _all_reduce_sum(a: np.ndarray, device, group=None) -> np.ndarray:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return a
    t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def rcc_sharded(segments: np.ndarray, max_shift=32, group=None, device="cpu", pair_shift_fn=None):
    """imageprocess.rcc with the n(n-1)/2 correlations split round-robin over the ranks.
    `segments` must be identical on every rank (see undrift_sharded).  Every rank returns the
    minimized (shift_y, shift_x) of all segments.  `pair_shift_fn(segments, box, roi, pairs)`
    defaults to the GPU path (imageprocess._shifts_of_pairs); the CPU tests inject a numpy one."""
    from . import imageprocess, lib
    fn = pair_shift_fn or imageprocess._shifts_of_pairs
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    n = len(segments)
    pairs = [(i, j) for i in range(n - 1) for j in range(i + 1, n)]
    mine = pairs[rank::world]
    shifts = np.zeros((2, n, n))
    if mine:
        for (i, j), (sy, sx) in zip(mine, fn(np.asarray(segments, np.float64), 5, max_shift, mine)):
            shifts[0, i, j], shifts[1, i, j] = sy, sx
    shifts = _all_reduce_sum(shifts, device, group)          # disjoint supports: the sum is the union
    return lib.minimize_shifts(shifts[1], shifts[0])


def undrift_sharded(locs_shard, info, segmentation: int, group=None, device="cpu", render_fn=None,
                    pair_shift_fn=None):
    """RCC undrift when every rank holds the localizations of its own frame range.
    Each rank renders its share of every temporal segment (Gaussian, min_blur_width 1); segment
    bounds come from the global frame count, so a segment that straddles two shards is the SUM of
    two partial images (all-reduce; the render is additive).  The correlations are split over the
    ranks (rcc_sharded); drift spline and its application are local.
    -> (drift DataFrame over all frames, undrifted localizations of this shard)."""
    import warnings

    import pandas as pd
    from scipy import interpolate

    from . import postprocess, render
    rfn = render_fn or render.render
    Y, X, n_frames = info[0]["Height"], info[0]["Width"], info[0]["Frames"]
    n_seg = postprocess.n_segments(info, segmentation)
    bounds = np.linspace(0, n_frames - 1, n_seg + 1, dtype=np.uint32)
    segments = np.zeros((n_seg, Y, X))
    locs_shard = locs_shard.copy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        for i in range(n_seg):
            part = locs_shard[(locs_shard["frame"] >= bounds[i]) & (locs_shard["frame"] < bounds[i + 1])]
            if len(part):
                _, segments[i] = rfn(part, info, blur_method="gaussian", min_blur_width=1)
    segments = _all_reduce_sum(segments, device, group)
    shift_y, shift_x = rcc_sharded(segments, 32, group, device, pair_shift_fn)
    t = (bounds[1:] + bounds[:-1]) / 2
    t_inter = np.arange(n_frames)
    drift = pd.DataFrame({"x": interpolate.InterpolatedUnivariateSpline(t, shift_x, k=3)(t_inter),
                          "y": interpolate.InterpolatedUnivariateSpline(t, shift_y, k=3)(t_inter)})
    return drift, postprocess.apply_drift(locs_shard, info, drift=drift)
