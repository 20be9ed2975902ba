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


class NativeCommUnavailable(RuntimeError):
    """Raised on EVERY rank of a group when some rank cannot take part in the library's RCCL communicator; callers
    fall back to torch.distributed's collectives."""


def _agree(ok: bool, device, group=None) -> bool:
    """True on every rank iff `ok` on every rank (all-reduce MIN over the group's own backend)."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


class NativeComm:
    """RCCL communicator owned by libpicasso_hip.so (pmi_comm_init): the all-gather of the localization tables
    runs inside the library (pmi_allgather_locs + pmi_compact_gathered_dev), torch.distributed only carries the
    128-byte id from rank 0 to the others — the part a host without torch does with a file or a socket."""

    # (group object, communicator or the exception that said why there is none): the group object itself is kept (a
    # strong reference), so its identity cannot be reused by a later group while the entry stands
    _cache = []

    def __init__(self, world: int, rank: int, id_bytes: bytes):
        import ctypes

        from . import _lib
        self.world, self.rank = int(world), int(rank)
        self._h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(id_bytes, 128)
        _lib.check(_lib.load().pmi_comm_init(buf, self.world, self.rank, ctypes.byref(self._h)), "pmi_comm_init")

    @staticmethod
    def available() -> bool:
        from . import _lib
        try:
            return _lib.load().pmi_comm_available() == 0
        except Exception:      # noqa: BLE001 - a library that does not load is "not available"
            return False

    @staticmethod
    def unique_id() -> bytes:
        import ctypes

        from . import _lib
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().pmi_comm_unique_id(buf), "pmi_comm_unique_id")
        return buf.raw

    @classmethod
    def for_group(cls, group=None, device=None):
        """The communicator that mirrors a torch.distributed group (made once per group).  Collective: every rank of
        the group calls it.  The ranks agree, over the group itself, that each of them can load RCCL BEFORE any of
        them enters ncclCommInitRank (a rank that cannot join would leave the others waiting inside it), that rank 0
        produced an id, and that every rank's communicator came up; on any "no" every rank raises
        NativeCommUnavailable and the callers take torch.distributed's collectives instead."""
        pg = group if group is not None else dist.distributed_c10d._get_default_group()
        for g, entry in cls._cache:
            if g is pg:
                if isinstance(entry, Exception):
                    raise entry
                return entry
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())

        def give_up(why):
            exc = NativeCommUnavailable(why)
            cls._cache.append((pg, exc))
            raise exc

        if not _agree(cls.available(), device, group):
            give_up("RCCL cannot be loaded from libpicasso_hip.so on some rank")
        box = [None]
        if rank == 0:
            try:
                box[0] = cls.unique_id()
            except Exception as exc:      # noqa: BLE001 - every rank must leave the broadcast below the same way
                box[0] = f"error: {exc}".encode()
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if not isinstance(box[0], (bytes, bytearray)) or len(box[0]) != 128:
            give_up(f"no RCCL unique id from rank 0 ({box[0]!r})")        # the same verdict on every rank: it was broadcast
        comm, err = None, None
        try:
            comm = cls(world, rank, box[0])
            w, r = comm.info()
            if (w, r) != (world, rank):
                raise RuntimeError(f"communicator reports world {w} rank {r}, the group has world {world} rank {rank}")
        except Exception as exc:      # noqa: BLE001
            err = exc
        if not _agree(err is None, device, group):
            if comm is not None:
                comm.close()
            give_up(f"pmi_comm_init failed on some rank ({err})" if err else "pmi_comm_init failed on another rank")
        cls._cache.append((pg, comm))
        return comm

    @classmethod
    def close_all(cls):
        """Destroys the cached communicators (call before dist.destroy_process_group)."""
        for _, entry in cls._cache:
            if isinstance(entry, NativeComm):
                entry.close()
        cls._cache.clear()

    def info(self):
        import ctypes

        from . import _lib
        w, r = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.load().pmi_comm_info(self._h, ctypes.byref(w), ctypes.byref(r)), "pmi_comm_info")
        return w.value, r.value

    @staticmethod
    def library_path() -> str:
        """The librccl shared object the library's collectives resolved to (dladdr of ncclAllGather)."""
        import ctypes

        from . import _lib
        buf = ctypes.create_string_buffer(1024)
        _lib.check(_lib.load().pmi_comm_library_path(buf, 1024), "pmi_comm_library_path")
        return buf.value.decode()

    def allgather_table(self, table: torch.Tensor, d_n: torch.Tensor, stream=None):
        """table: (C, cap) int32 on the GPU, the same cap on every rank; d_n: its device row count (int64).
        Returns (gathered (world, C, cap), counts (world,) int64), both on the device; asynchronous."""
        import ctypes

        from . import _lib
        C, cap = table.shape
        allt = torch.empty((self.world, C, cap), dtype=torch.int32, device=table.device)
        counts = torch.empty((self.world,), dtype=torch.int64, device=table.device)
        s = ctypes.c_void_p(stream if stream is not None else torch.cuda.current_stream(table.device).cuda_stream)
        _lib.check(_lib.load().pmi_allgather_locs(self._h, ctypes.c_void_p(table.data_ptr()), C, cap,
                                                  ctypes.c_void_p(d_n.data_ptr()), ctypes.c_void_p(allt.data_ptr()),
                                                  ctypes.c_void_p(counts.data_ptr()), s), "pmi_allgather_locs")
        return allt, counts

    def compact(self, allt: torch.Tensor, counts: torch.Tensor, stream=None):
        """(world, C, cap) padded tables -> ((C, world * cap) table, device total): rows in rank order."""
        import ctypes

        from . import _lib
        world, C, cap = allt.shape
        out = torch.empty((C, world * cap), dtype=torch.int32, device=allt.device)
        total = torch.zeros((1,), dtype=torch.int64, device=allt.device)
        s = ctypes.c_void_p(stream if stream is not None else torch.cuda.current_stream(allt.device).cuda_stream)
        _lib.check(_lib.load().pmi_compact_gathered_dev(ctypes.c_void_p(allt.data_ptr()), ctypes.c_void_p(counts.data_ptr()),
                                                        world, C, cap, ctypes.c_void_p(out.data_ptr()), world * cap,
                                                        ctypes.c_void_p(total.data_ptr()), s), "pmi_compact_gathered_dev")
        return out, total

    def close(self):
        from . import _lib
        if self._h:
            _lib.load().pmi_comm_destroy(self._h)
            self._h = None


def allgather_table(table: torch.Tensor, n_rows, group=None) -> torch.Tensor:
    """All-gather a column-major table (C columns x capacity, 4-byte cells, first
    `n_rows` of each column valid) from every rank.  Returns a (C, total) tensor
    on every rank, rows in rank order.  Without a process group (one process, one GPU) it is
    the identity on the valid rows.  On the GPU the exchange is the library's own (pmi_allgather_locs, RCCL called
    from C); host tensors (the gloo tests) go through torch.distributed."""
    if not (dist.is_available() and dist.is_initialized()):
        return table[:, : int(n_rows)].clone()
    world = dist.get_world_size(group)
    if table.element_size() != 4:
        raise TypeError(f"allgather_table moves 4-byte cells, got {table.dtype}")
    comm = None
    if table.is_cuda:
        try:
            comm = NativeComm.for_group(group, table.device)
        except NativeCommUnavailable:      # raised on every rank alike: all of them take torch.distributed's path below
            comm = None
    if comm is not None:
        cap = _all_reduce_max_int(max(int(n_rows), 1), table.device, group)      # the same padded width on every rank
        send = torch.zeros((table.shape[0], cap), dtype=torch.int32, device=table.device)
        send[:, : int(n_rows)] = table[:, : int(n_rows)].view(torch.int32) if table.dtype != torch.int32 else table[:, : int(n_rows)]
        d_n = torch.tensor([int(n_rows)], dtype=torch.int64, device=table.device)
        allt, counts = comm.allgather_table(send, d_n)
        out, total = comm.compact(allt, counts)
        res = out[:, : int(total.item())]
        return res if table.dtype == torch.int32 else res.view(table.dtype)
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


class PipelinedTableGather:
    """All-gather of a rank's table in pieces while the next piece is still being computed.

    ``submit(table, d_n)`` is called after each frame chunk has been queued on the current stream: the padded
    (C, cap) table and its device-side row count go out as asynchronous collectives (RCCL's own stream, ordered
    after the kernels queued so far), so the host never waits and the next chunk's kernels overlap the transfer.
    Every rank must submit the same number of pieces with the same capacities.  ``finish()`` waits for the
    collectives and returns the (C, total) table ordered by rank, then piece — i.e. by frame for contiguous frame
    shards — or None when some piece overflowed its capacity anywhere (the caller then falls back to the plain
    path).  Without a process group it concatenates the local pieces."""

    def __init__(self, group=None):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.pieces = []          # (table or gathered, counts tensor, cap, work handles)

    def submit(self, table: torch.Tensor, d_n: torch.Tensor):
        C, cap = table.shape
        if not self.on:
            self.pieces.append((table.view(1, C, cap), d_n.view(1), cap, ()))
            return
        recv = torch.empty((self.world * C, cap), dtype=table.dtype, device=table.device)
        counts = torch.empty((self.world,), dtype=d_n.dtype, device=d_n.device)
        works = (dist.all_gather_into_tensor(counts, d_n.view(1), group=self.group, async_op=True),
                 dist.all_gather_into_tensor(recv, table, group=self.group, async_op=True))
        self.pieces.append((recv.view(self.world, C, cap), counts, cap, works))

    def finish(self):
        for _, _, _, works in self.pieces:
            for w in works:
                w.wait()
        if not self.pieces:
            return None
        counts = torch.stack([c for _, c, _, _ in self.pieces]).cpu()          # (pieces, world): the one host sync
        if any(int(counts[i].max()) > self.pieces[i][2] for i in range(len(self.pieces))):
            return None
        parts = [self.pieces[i][0][r, :, : int(counts[i, r])] for r in range(self.world) for i in range(len(self.pieces))]
        return torch.cat(parts, dim=1)


def _all_reduce_max_int(value: int, device, group=None) -> int:
    if not (dist.is_available() and dist.is_initialized()):
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def localize_sharded(movie_shard: torch.Tensor, first_frame: int, camera_info: dict, parameters: dict, *,
                     eps: float = 1e-3, max_it: int = 100, mle_method: str = "sigmaxy", group=None, chunks: int = 1):
    """Each rank localizes its resident shard (uint16 CUDA tensor of its frames),
    then all ranks receive the whole table.  `first_frame` is the label of the
    shard's first frame (from shard_frames).  With ``chunks`` > 1 the shard is processed in that many frame
    ranges and the all-gather of each range overlaps the kernels of the next (a large shard's table is
    gigabytes: config 4 moves 2.7 GB per rank); the result is the same table."""
    import ctypes

    from . import _lib
    L = _lib.load()
    _lib.require_gpu()
    F, H, W = movie_shard.shape
    if chunks > 1 and F >= 2 * chunks:
        out = _localize_sharded_pipelined(L, movie_shard, first_frame, camera_info, parameters, eps, max_it,
                                          mle_method, group, chunks)
        if out is not None:
            return table_to_columns(out)
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


def _localize_sharded_pipelined(L, movie_shard, first_frame, camera_info, parameters, eps, max_it, mle_method, group, chunks):
    """Frame ranges of the shard, each followed by its asynchronous gather.  The first range is timed to learn the
    density of localizations (one host sync, maximum over the ranks) so that the later tables are tight; if a
    later range overflows its capacity anywhere the whole call returns None and the caller takes the plain path."""
    import ctypes

    from . import _lib
    F, H, W = movie_shard.shape
    dev = movie_shard.device
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    frames = _all_reduce_max_int(F, dev, group)                 # shards differ by at most one frame: same ranges everywhere
    cuts = [(frames * i + chunks - 1) // chunks for i in range(chunks + 1)]      # identical on every rank
    bounds = [min(F, c) for c in cuts]                                          # this rank's (a shorter shard ends early)
    C = len(LOC_COLUMNS)
    frame_bytes = H * W * movie_shard.element_size()

    def run(lo, hi, cap):
        table = torch.empty((C, cap), dtype=torch.int32, device=dev)
        d_n = torch.zeros(1, dtype=torch.int64, device=dev)
        nf = hi - lo
        if nf > 0:
            rc = L.pmi_localize_mle_dev(ctypes.c_void_p(movie_shard.data_ptr() + lo * frame_bytes), 0, nf, H, W,
                                        int(parameters["Box Size"]), float(parameters["Min. Net Gradient"]), None,
                                        0, nf - 1, float(camera_info["Baseline"]), float(camera_info["Sensitivity"]),
                                        float(camera_info["Gain"]), float(eps), int(max_it),
                                        _lib.MLE_METHODS[mle_method], ctypes.c_void_p(table.data_ptr()), cap,
                                        ctypes.c_void_p(d_n.data_ptr()), stream)
            _lib.check(rc, "pmi_localize_mle_dev")
            table[0] += int(first_frame) + lo              # frame labels; the padding beyond the count is never read
        return table, d_n

    width = max(1, cuts[1] - cuts[0])
    cap0 = _all_reduce_max_int(max(4096, 400 * width), dev, group)
    t0, n0 = run(bounds[0], bounds[1], cap0)
    found = int(n0.item())                                      # the one sync on the compute stream: density of this movie
    per_frame = _all_reduce_max_int(-(-found // width), dev, group)
    if _all_reduce_max_int(found, dev, group) > cap0:
        return None
    gather = PipelinedTableGather(group)
    tight = found if not gather.on else _all_reduce_max_int(found, dev, group)
    gather.submit(t0[:, : max(tight, 1)].contiguous(), n0)
    for i in range(1, chunks):
        lo, hi = bounds[i], bounds[i + 1]
        cap = int(per_frame * max(1, cuts[i + 1] - cuts[i]) * 1.15) + 1024          # the same on every rank
        table, d_n = run(lo, hi, cap)
        gather.submit(table, d_n)
    return gather.finish()


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
