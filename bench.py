#!/usr/bin/env python3
"""Headline benchmark: localizations/s, 7x7 ROI Poisson-MLE, on MI355X.

One step = one pass of the whole hot path (identify -> fused ROI cut + photon
conversion + MLE fit -> localization table) over the rank's resident synthetic
movie (BASELINE.json configs[1]: 10k frames, 512x512 uint16, ~1e6 spots).  With
N > 1 every rank owns its own movie shard (frames shard without a halo, weak
scaling) and each step ends with the RCCL all-gather of the localization table
the north star asks for.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FIT_BYTES_PER_SPOT_7 = 166.0   # 98 px + 12 id + 56 result


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: half a second of timed steps — long enough for an outside sampler of GPU activity to see the run)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=10000)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--emitters", type=int, default=116, help="emitters per frame (~86%% pass min_ng 5000)")
    ap.add_argument("--box", type=int, default=7)
    ap.add_argument("--min-ng", type=float, default=5000.0)
    ap.add_argument("--method", default="sigmaxy")
    ap.add_argument("--eps", type=float, default=1e-3,
                    help="convergence criterion of the MLE fit (the GUI's default; 1e-4 re-fits 11 %% of the spots: DESIGN.md section 7)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget (0 = skip)")
    ap.add_argument("--profile-steps", type=int, default=5, help="extra instrumented steps for per-kernel time")
    ap.add_argument("--strict-steps", type=int, default=2,
                    help="extra steps with every spot in the reference's arithmetic (value_strict; 0 = skip)")
    ap.add_argument("--ranges", type=int, default=2, choices=(1, 2),
                    help="frame ranges pmi_localize_mle_dev keeps in flight in the timed steps (2 = the library's default "
                         "schedule: the scan of the second half beside the fit of the first; 1 = one range, as the profiled passes)")
    ap.add_argument("--allow-env", action="store_true",
                    help="run although PMI_* / PICASSO_AMD_LIB tuning variables are set (they are echoed in the line)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="N > 1: weak = every rank owns a --frames movie (the default, what BASELINE's metric scales); "
                         "strong = --frames is the whole job, each rank takes frames / N of it")
    ap.add_argument("--serial-gather", action="store_true",
                    help="N > 1: wait for each step's all-gather before the next step computes (no overlap)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start N ranks (one per GPU) under
    torch.distributed.run as a CHILD process, relay rank 0's JSON line and return the children's exit status.
    Nothing here imports torch or touches HIP — a process that has initialised the GPU must never exec or fork
    into another program on this pool."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what this pool's driver supports (RCCL)
    proc = subprocess.run(cmd, env=env)
    if proc.returncode != 0:
        print(f"bench.py: the {args.gpus}-rank launch failed with status {proc.returncode}", file=sys.stderr)
    return proc.returncode


def main():
    args = parse()
    # the library reads tuning / debugging variables from the environment (PMI_MLE_MODE, PMI_IDENTIFY_GENERIC, ...,
    # PICASSO_AMD_LIB loads another build): a benchmark line measured under any of them is not the product's
    overrides = {k: v for k, v in sorted(os.environ.items()) if k.startswith("PMI_") or k == "PICASSO_AMD_LIB"}
    if overrides and not args.allow_env:
        raise SystemExit(f"bench.py: tuning variables are set ({overrides}); unset them or pass --allow-env")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    import torch
    import torch.distributed as dist

    from picasso_amd import _lib, backend, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    if local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible — one process per GPU")
    torch.cuda.set_device(local_rank)                # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run (even with one process) the collectives are part of the step
    grouped = world > 1 or ("MASTER_PORT" in os.environ and "RANK" in os.environ)
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")      # "nccl" is RCCL on ROCm
    L = _lib.load()
    _lib.require_gpu()
    _lib.check(L.pmi_set_device(local_rank), "pmi_set_device")

    _lib.check(L.pmi_localize_set_ranges(args.ranges), "pmi_localize_set_ranges")
    F, H, W, box = args.frames, args.size, args.size, args.box
    if args.scaling == "strong":
        # the whole job is --frames frames: this rank's contiguous share (picasso_amd.dist.shard_frames), its own movie
        from picasso_amd.dist import shard_frames
        f_lo, f_hi = shard_frames(args.frames, world, rank)
        F = f_hi - f_lo
        if F < 16:
            raise SystemExit(f"bench.py --scaling strong: {args.frames} frames over {world} ranks leaves {F} per rank")
    cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}
    movie = synth.simulate_movie(F, H, W, emitters_per_frame=args.emitters,
                                 seed=synth.DEFAULT_SEED + rank, device=dev)
    torch.cuda.synchronize()
    movie_bytes = movie.numel() * movie.element_size()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    method = _lib.MLE_METHODS[args.method]

    def run(table, d_n, cap):
        rc = L.pmi_localize_mle_dev(ctypes.c_void_p(movie.data_ptr()), 0, F, H, W, box, args.min_ng, None, 0, F - 1,
                                    cam["Baseline"], cam["Sensitivity"], cam["Gain"], args.eps, 100, method,
                                    ctypes.c_void_p(table.data_ptr()), cap, ctypes.c_void_p(d_n.data_ptr()), stream)
        _lib.check(rc, "pmi_localize_mle_dev")

    # sizing pass: learn the row count, then keep the table tight for the all-gather
    cap = max(4096, 400 * F)
    d_n = torch.zeros(1, dtype=torch.int64, device=dev)
    table = torch.empty((_lib.PMI_LOC_COLUMNS, cap), dtype=torch.int32, device=dev)
    run(table, d_n, cap)
    torch.cuda.synchronize()
    n_local = int(d_n.item())
    assert 0 < n_local <= cap, f"sizing pass found {n_local} rows (cap {cap})"
    cap = int(n_local * 1.02) + 1024
    if grouped:
        capt = torch.tensor([cap], dtype=torch.int64, device=dev)
        dist.all_reduce(capt, op=dist.ReduceOp.MAX)
        cap = int(capt.item())
    del table
    # Two sets of buffers: the all-gather of step i (RCCL's own stream, over xGMI) runs while step i+1
    # computes into the other set; a set is reused only after its gather has finished.  Every gather
    # completes inside the timed region (drained before the closing synchronize + barrier).
    nbuf = 2 if (grouped and not args.serial_gather) else 1
    tables = [torch.empty((_lib.PMI_LOC_COLUMNS, cap), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    d_ns = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(nbuf)]
    gathered = gathered_n = None
    if grouped:
        gathered = [torch.empty((world * _lib.PMI_LOC_COLUMNS, cap), dtype=torch.int32, device=dev) for _ in range(nbuf)]
        gathered_n = [torch.empty((world,), dtype=torch.int64, device=dev) for _ in range(nbuf)]
    pending = [None] * nbuf
    table, d_n = tables[0], d_ns[0]

    # The all-gather is the library's own (pmi_allgather_locs: RCCL called from C on a side stream of ours); if that
    # communicator cannot be made the step falls back to torch.distributed's collectives and says so in the line.
    gather_impl, comm, gstream = "none", None, None
    rccl_seen, rccl_lib = None, None
    if grouped:
        try:
            if os.environ.get("PMI_BENCH_TORCH_GATHER"):
                raise RuntimeError("PMI_BENCH_TORCH_GATHER set")
            from picasso_amd.dist import NativeComm
            comm = NativeComm.for_group(None, dev)
            rccl_seen = comm.info()          # ncclCommCount / ncclCommUserRank of the library's own communicator
            assert rccl_seen == (world, rank), f"communicator {rccl_seen} but the launch has world {world} rank {rank}"
            rccl_lib = NativeComm.library_path()
            gstream = torch.cuda.Stream(device=dev)
            gather_impl = "pmi_allgather_locs (RCCL from libpicasso_hip.so)"
        except Exception as exc:      # noqa: BLE001 - any failure to set up the native communicator
            comm = None
            gather_impl = f"torch.distributed all_gather_into_tensor (native communicator unavailable: {exc})"
        # every rank takes the same path: one rank without the native communicator puts all of them on torch's
        okt = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 0 and comm is not None:
            comm.close()
            comm, gstream = None, None
            rccl_seen = None
            gather_impl = "torch.distributed all_gather_into_tensor (native communicator unavailable on another rank)"
    g_stream_ptr = ctypes.c_void_p(gstream.cuda_stream) if gstream is not None else None

    timeline = []          # per timed step: (compute start, compute end, gather end or None) events

    def native_gather(k, ready):
        """gather set k on the side stream, after the kernels queued so far; returns the event that marks its end"""
        gstream.wait_event(ready)
        rc = L.pmi_allgather_locs(comm._h, ctypes.c_void_p(tables[k].data_ptr()), _lib.PMI_LOC_COLUMNS, cap,
                                  ctypes.c_void_p(d_ns[k].data_ptr()), ctypes.c_void_p(gathered[k].data_ptr()),
                                  ctypes.c_void_p(gathered_n[k].data_ptr()), g_stream_ptr)
        _lib.check(rc, "pmi_allgather_locs")
        done = torch.cuda.Event(enable_timing=True)
        done.record(gstream)
        return done

    def drain(k):
        if pending[k] is not None:
            if comm is not None:
                torch.cuda.current_stream(dev).wait_event(pending[k])
            else:
                for w in pending[k]:
                    w.wait()             # the compute stream waits for that gather; the host does not block
            pending[k] = None

    def step(i, timed=False):
        k = i % nbuf
        drain(k)
        cur = torch.cuda.current_stream(dev)
        ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev_a.record(cur)
        run(tables[k], d_ns[k], cap)
        ev_b.record(cur)
        if timed:
            timeline.append([ev_a, ev_b, None])
        if grouped:        # localization table of every shard on every GPU (RCCL over xGMI)
            if comm is not None:
                pending[k] = native_gather(k, ev_b)
                if timed:
                    timeline[-1][2] = pending[k]
                if nbuf == 1:
                    drain(k)
            elif nbuf == 1:
                dist.all_gather_into_tensor(gathered_n[0], d_ns[0])
                dist.all_gather_into_tensor(gathered[0], tables[0])
            else:
                pending[k] = (dist.all_gather_into_tensor(gathered_n[k], d_ns[k], async_op=True),
                              dist.all_gather_into_tensor(gathered[k], tables[k], async_op=True))

    def drain_all():
        for k in range(nbuf):
            drain(k)

    for i in range(args.warmup):
        step(i)
    drain_all()
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, timed=True)
    drain_all()
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if grouped:
        # the gathered table is what every rank would hand on: check rank order and counts once
        last = (args.steps - 1) % nbuf if args.steps > 0 else 0
        counts = gathered_n[last].cpu().tolist()
        assert counts[rank] == int(d_ns[last].item()) and all(0 < c <= cap for c in counts), counts
        mine = gathered[last].view(world, _lib.PMI_LOC_COLUMNS, cap)[rank, :, : counts[rank]]
        assert torch.equal(mine, tables[last][:, : counts[rank]]), "gathered table differs from the local one"
    # where a rank's step went: kernels of the path (compute stream) and, beside them, the all-gather on its side stream
    # (from the moment the step's table was ready to the end of the collective — it includes waiting for the slowest rank)
    compute_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in timeline])) if timeline else float("nan")
    # the device's own clock over the timed region: first timed step's start event to the last one's end event (rank 0's
    # stream).  The GPU-busy sampler of a harness cannot see a 50 ms region; these two numbers agreeing is the line's
    # own evidence that `elapsed` is device time, not host time spent queueing
    region_ms = float(timeline[0][0].elapsed_time(timeline[-1][1])) if timeline else float("nan")
    gather_ms = float(np.mean([b.elapsed_time(g) for _, b, g in timeline if g is not None])) if any(g is not None for _, _, g in timeline) else None
    per_rank = None
    if grouped:
        mine_t = torch.tensor([compute_ms, gather_ms if gather_ms is not None else float("nan")], dtype=torch.float64, device=dev)
        all_t = torch.empty((world, 2), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(all_t, mine_t)
        all_h = all_t.cpu().tolist()
        per_rank = {"compute_ms": [round(r[0], 4) for r in all_h],
                    "gather_ms": [None if r[1] != r[1] else round(r[1], 4) for r in all_h],
                    "gather_bytes_received_per_step": world * _lib.PMI_LOC_COLUMNS * cap * 4 + 8 * world}
    et = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    nt = torch.tensor([int(d_n.item())], dtype=torch.int64, device=dev)
    if grouped:
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        dist.all_reduce(nt, op=dist.ReduceOp.SUM)
    elapsed = float(et.item())
    n_total = int(nt.item())
    value = n_total * args.steps / elapsed

    # per-kernel durations: HIP events around the kernels on the launch stream (library-side)
    scan_ms = fit_ms = float("nan")
    if args.profile_steps > 0:
        # the roofline of a kernel is measured on launches of its own: one frame range over the whole movie, nothing
        # beside it (in the timed steps above each step launches the scan twice, half the movie each, the second beside
        # the fit of the first half)
        _lib.check(L.pmi_localize_set_ranges(1), "pmi_localize_set_ranges")
        L.pmi_set_kernel_timing(1)
        s_acc, f_acc = [], []
        a, b = ctypes.c_float(0), ctypes.c_float(0)
        for _ in range(args.profile_steps):
            run(table, d_n, cap)
            torch.cuda.synchronize()
            L.pmi_last_kernel_ms(ctypes.byref(a), ctypes.byref(b))
            s_acc.append(a.value); f_acc.append(b.value)
        L.pmi_set_kernel_timing(0)
        _lib.check(L.pmi_localize_set_ranges(args.ranges), "pmi_localize_set_ranges")
        scan_ms, fit_ms = float(np.mean(s_acc)), float(np.mean(f_acc))

    # the refit count of the timed configuration, then — beside `value` — the same step with EVERY spot fitted in the
    # reference's arithmetic (float64 intermediates, float32 stores, its summation order: pmi_mle_set_mode strict)
    mle_mode, mle_margin = backend.get_mle_mode()
    refit = backend.last_refit_count(stream)
    scan_kernel = last_scan_kernel(L)
    strict_ms = strict_dev_ms = None
    if args.strict_steps > 0 and mle_mode != "strict":
        def strict_pass():
            run(table, d_n, cap)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.strict_steps):
                run(table, d_n, cap)
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t1) / args.strict_steps
        libm = backend.get_mle_libm()
        backend.set_mle_mode("strict", mle_margin)
        try:
            strict_ms = strict_pass()
            # ... and with the device library's erf / exp instead of glibc's bits (pmi_mle_set_libm: the kernel of round 5's
            # arithmetic; the default pays for the reference's C library bit for bit, csrc/libm_glibc.h)
            backend.set_mle_libm("device")
            strict_dev_ms = strict_pass()
        finally:
            backend.set_mle_libm(libm)
            backend.set_mle_mode(mle_mode, mle_margin)
        if grouped:
            st = torch.tensor([strict_ms, strict_dev_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
            strict_ms, strict_dev_ms = float(st[0].item()), float(st[1].item())

    result = None
    if rank == 0:
        n_rank0 = int(d_n.item())
        kernels = {
            "identify_scan": {"ms": scan_ms, "algorithmic_bytes": movie_bytes,
                              "GB/s": movie_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms == scan_ms else None},
            "mle_fit": {"ms": fit_ms, "algorithmic_bytes": FIT_BYTES_PER_SPOT_7 * n_rank0,
                        "GB/s": FIT_BYTES_PER_SPOT_7 * n_rank0 / (fit_ms * 1e-3) / 1e9 if fit_ms == fit_ms else None,
                        "spots_per_s": n_rank0 / (fit_ms * 1e-3) if fit_ms == fit_ms else None},
        }
        # The roofline object describes the HBM-bound kernel of the path, the frame scan (it moves
        # 5.24 kB per localization; the fit moves 166 B and is FP32-ALU bound, listed beside it).
        dom = "identify_scan"
        ach = kernels[dom]["GB/s"]
        traffic, traffic_source = pmc_traffic_bytes(F, H, W, box, scan_kernel)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (ach / HBM_PEAK_GBS) if ach else None,
                    "traffic": traffic, "traffic_source": traffic_source,
                    "measured_on": f"{args.profile_steps} single-range passes over the whole movie after the timed steps (HIP events "
                                   "around the kernels on their launch stream, inside the library); the same launches as "
                                   "`bench.py --ranges 1`, profiles/r06_bench_ranges1_kernel_stats.txt",
                    "scan_kernel": scan_kernel,
                    "kernels": kernels}
        kernels["mle_fit"]["bound"] = "fp32 valu (no MFMA shape); algorithmic bytes are 166 B/spot"
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            cpu = cpu_baseline(movie, cam, box, args.min_ng, args.method, args.cpu_seconds, args.eps)
        result = {
            "metric": "localizations/sec (7x7 ROI, MLE)", "value": value, "unit": "localizations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None,
            # HIP events on the compute stream: start of the first timed step to the end of the last, per step (rank 0); and the
            # mean of the per-step event pairs.  `ms_per_step` above is the host clock between the two synchronisations.
            "ms_per_step_hip_events": region_ms / args.steps if args.steps else None,
            "ms_per_step_hip_events_mean_of_steps": compute_ms,
            # every spot in the reference's arithmetic (pmi_mle_set_mode strict), the same step, after the timed ones (no all-gather)
            "ms_per_step_strict": strict_ms,
            "value_strict": (n_total / (strict_ms * 1e-3)) if strict_ms else None,
            "ms_per_step_strict_device_libm": strict_dev_ms,      # the same with the device library's erf / exp (PMI_LIBM_DEVICE)
            "refit_fraction": (refit / n_rank0) if n_rank0 else None,
            # float32 Newton loop; spots whose convergence test falls within rounding distance of eps are fitted again
            # with the reference's float64 intermediates INSIDE the timed step (csrc/gaussmle_strict.hip)
            "dtype": {"fast": "f32", "refit": "f32+f64", "strict": "f64"}[mle_mode], "data": "synthetic",
            "config": {"workload": f"{F}-frame {H}x{W} uint16 simulated DNA-PAINT movie per GPU, "
                                   f"{n_total // world} spots per GPU, {box}x{box} ROI MLE ({args.method}), "
                                   "identify+cut+fit+table resident in HBM"
                                   + (", + RCCL all-gather of the table"
                                      + (" (double-buffered: overlaps the next step's compute)" if nbuf == 2 else "")
                                      if grouped else ""),
                       "frames": F, "height": H, "width": W, "box": box, "min_net_gradient": args.min_ng,
                       "eps": args.eps, "max_it": 100, "localizations_total": n_total,
                       "mle_mode": mle_mode, "refit_margin": mle_margin, "refit_spots_rank0": refit,
                       "frame_ranges_in_flight": args.ranges,
                       "all_gather": gather_impl,
                       # the collective's own evidence: world size and rank RCCL reports for the library's communicator on
                       # rank 0 (ncclCommCount / ncclCommUserRank), and the librccl its entry points resolved to (dladdr)
                       "rccl_world": rccl_seen[0] if rccl_seen else None,
                       "rccl_rank0_sees": list(rccl_seen) if rccl_seen else None,
                       "librccl": rccl_lib,
                       "sharding": f"frames x{world}", "per_rank": per_rank, "env_overrides": overrides},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(result), flush=True)
    if grouped:
        dist.barrier()
        if comm is not None:
            from picasso_amd.dist import NativeComm
            NativeComm.close_all()
        dist.destroy_process_group()
    return result


PMC_TRAFFIC_FILE = "profiles/r06_identify_pmc.json"


def last_scan_kernel(L):
    buf = ctypes.create_string_buffer(128)
    L.pmi_last_scan_kernel(buf, 128)
    return buf.value.decode()


def pmc_traffic_bytes(F, H, W, box, scan_kernel):
    """HBM-side bytes per launch of the scan kernel from the committed rocprofv3 PMC run of THIS kernel on THIS
    workload (TCC_EA0_RDREQ x 128 B = 2 x FETCH_SIZE x 1024, the gfx950 correction of MI355X_MICROARCH.md, plus
    WRITE_SIZE), and where the number comes from; (None, None) for any other workload.  Counters cannot be read
    inside an un-profiled run: the figure is a measurement of the committed profile, labelled as such — and the
    committed profile must be of the kernel (template instance, deferred exact stage or not) this run launched:
    anything else is a stale file and stops the benchmark."""
    path = os.path.join(ROOT, PMC_TRAFFIC_FILE)
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None, None
    if [rec.get("frames"), rec.get("height"), rec.get("width"), rec.get("box")] != [F, H, W, box]:
        return None, None
    profiled = rec["kernel"].split("(")[0].replace("void ", "").replace("pmi::", "") + (" defer" if rec.get("defer") else "")
    if profiled != scan_kernel:
        raise SystemExit(f"bench.py: {PMC_TRAFFIC_FILE} holds the counters of `{profiled}` but this run launched `{scan_kernel}`: "
                         "take the counters again (tools/pmc_scan.sh) before quoting roofline.traffic")
    return (rec["hbm_read_bytes_per_launch"] + rec["hbm_write_bytes_per_launch"],
            f"{PMC_TRAFFIC_FILE} (rocprofv3 --pmc TCC_EA0_RDREQ_sum / WRITE_SIZE passes of tools/pmc_scan.sh, kernel {profiled})")


def cpu_baseline(movie, cam, box, min_ng, method, budget_s, eps=1e-3):
    """The CPU oracle (C restatement of the reference algorithm) on this box's host
    cores, on a bounded sample of the same movie: identify + get_spots + gaussmle."""
    from oracle import oracle as orc
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except Exception:
        pass

    def run(nframes, nthreads):
        host = movie[:nframes].cpu().numpy()
        t0 = time.perf_counter()
        fr, y, x, ng = orc.identify(host, min_ng, box, threads=nthreads)
        spots = orc.get_spots(host, fr, y, x, box, cam)
        orc.gaussmle(spots, eps, 100, method, threads=nthreads)
        return len(fr), time.perf_counter() - t0

    # The visible CPU count may exceed what the container may actually use (CPU quota): probe a
    # few thread counts on a small sample and keep the fastest; `cores` reports the threads used.
    probe = min(48, movie.shape[0])
    run(probe, threads)                      # warms the OpenMP pool and the page cache
    best = None
    for t in sorted({threads, min(threads, 64), min(threads, 16), min(threads, 8)}, reverse=True):
        n, dt = run(probe, t)
        if best is None or n / dt > best[0]:
            best = (n / dt, t, dt)
    rate, threads, dt = best
    n1, dt1 = run(probe, 1)                  # the single-thread rate beside the all-thread one
    frames = int(min(movie.shape[0], max(probe, probe * budget_s / max(dt, 1e-3))))
    n, dt = run(frames, threads)
    return {"value": n / dt, "unit": "localizations/s", "cores": threads, "kind": "port", "value_1_thread": n1 / dt1,
            "sample": f"first {frames} frames of the same movie ({n} spots), identify+get_spots+gaussmle "
                      f"({method}, eps {eps:g}, max_it 100), C/OpenMP restatement of the reference algorithm, "
                      f"{dt:.1f} s"}


if __name__ == "__main__":
    main()
