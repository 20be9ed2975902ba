"""GPU tier: the process-level contracts — RCCL initialises under torch.distributed.run (one rank here; the
2/4/8-rank runs are the driver's), bench.py starts its own ranks for --gpus N and refuses a world it was not
asked for.  Every program runs as a CHILD process: a process that has initialised the GPU never execs."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _launch(args, tries=3):
    """`python -m torch.distributed.run --nproc-per-node=1 ... <args>` on a free port of the loopback; a port that was free
    when it was picked can be taken by the time the rendezvous store binds it (EADDRINUSE): another port, again."""
    for attempt in range(tries):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port())] + list(args)
        out = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
        if out.returncode == 0 or "EADDRINUSE" not in out.stderr:
            break
    return out


def test_rccl_probe_one_rank():
    """tools/rccl_probe.py: nccl (= RCCL) process group, the table all-gather of picasso_amd/dist.py, the
    all-reduce of the sharded undrift, the pipelined shard path — on one rank."""
    out = _launch([os.path.join(ROOT, "tools", "rccl_probe.py")])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "rccl probe ok: world 1" in out.stdout


def test_bench_one_gpu_line_and_world_checks():
    small = ["--steps", "2", "--warmup", "1", "--frames", "400", "--cpu-seconds", "0", "--profile-steps", "1", "--strict-steps", "1"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + small, cwd=ROOT, env=_env(),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 1e6
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
    assert line["roofline"]["scan_kernel"].startswith("identify_scan_u16_fast_kernel<3, 3, 1, 0, false>")
    # every spot in the reference's arithmetic beside the timed configuration, and how many of its spots were fitted twice
    assert line["value_strict"] and 0 < line["value_strict"] < line["value"] and 0 < line["refit_fraction"] < 0.05
    # a world the command line did not ask for is refused
    env = _env()
    env.update({"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + small, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "WORLD_SIZE is 2" in bad.stderr


def test_bench_starts_its_own_ranks():
    """--gpus 2 without a launcher: bench.py spawns torch.distributed.run itself.  On a one-GPU box the second rank
    finds no device and the whole launch fails loudly; on a box with two GPUs it prints a 2-GPU line."""
    import torch
    small = ["--steps", "2", "--warmup", "1", "--frames", "400", "--cpu-seconds", "0", "--profile-steps", "0"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + small, cwd=ROOT, env=_env(),
                         capture_output=True, text=True, timeout=900)
    if torch.cuda.device_count() >= 2:
        assert out.returncode == 0, out.stderr[-2000:]
        assert json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 2
    else:
        assert out.returncode != 0
        assert "only 1 GPU(s) visible" in out.stderr and "2-rank launch failed" in out.stderr


@pytest.mark.parametrize("serial", [False, True])
def test_bench_under_the_launcher_uses_the_native_all_gather(serial):
    """The N > 1 path of bench.py on the one GPU there is: one rank under torch.distributed.run, 50 timed steps that each
    end with pmi_allgather_locs — RCCL called from the library (its own communicator, its own librccl handle) beside
    torch.distributed's RCCL in the same process — double-buffered on a side stream, and unoverlapped (--serial-gather).
    bench.py checks the gathered table against the local one; the line carries the per-rank compute / gather split."""
    small = ["--steps", "50", "--warmup", "2", "--frames", "400", "--cpu-seconds", "0", "--profile-steps", "0", "--strict-steps", "0"]
    out = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1"] + small + (["--serial-gather"] if serial else []))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["steps"] == 50 and line["n_gpus"] == 1
    assert line["config"]["all_gather"].startswith("pmi_allgather_locs"), line["config"]["all_gather"]
    per = line["config"]["per_rank"]
    assert per and len(per["compute_ms"]) == 1 and per["compute_ms"][0] > 0
    assert per["gather_ms"][0] is not None and per["gather_ms"][0] >= 0 and per["gather_bytes_received_per_step"] > 0
    assert ("double-buffered" in line["config"]["workload"]) == (not serial)
    # the line proves its own collective: what RCCL reports for the library's communicator, and which librccl it called
    assert line["config"]["rccl_world"] == 1 and line["config"]["rccl_rank0_sees"] == [1, 0]
    assert os.path.basename(line["config"]["librccl"]).startswith("librccl.so") and os.path.exists(line["config"]["librccl"])
    assert line["scaling"] == "weak"


def test_bench_strong_scaling_shares_the_frames():
    """`bench.py --scaling strong`: --frames is the whole job, each rank localizes frames / N of it (one rank here: all of
    it), the line says "strong" and its localization count is the whole job's."""
    small = ["--steps", "3", "--warmup", "1", "--frames", "300", "--cpu-seconds", "0", "--profile-steps", "0", "--strict-steps", "0"]
    out = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--scaling", "strong"] + small)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["config"]["frames"] == 300 and line["config"]["localizations_total"] > 20000


def test_native_communicator_one_rank():
    """pmi_comm_unique_id / pmi_comm_init / pmi_allgather_locs / pmi_compact_gathered_dev through raw ctypes, no
    torch.distributed anywhere: what a torch-free host would do."""
    import ctypes

    import numpy as np
    import torch
    from picasso_amd import _lib
    L = _lib.load()
    _lib.require_gpu()
    idb = ctypes.create_string_buffer(128)
    _lib.check(L.pmi_comm_unique_id(idb), "pmi_comm_unique_id")
    comm = ctypes.c_void_p()
    _lib.check(L.pmi_comm_init(idb, 1, 0, ctypes.byref(comm)), "pmi_comm_init")
    w, r = ctypes.c_int(-1), ctypes.c_int(-1)
    _lib.check(L.pmi_comm_info(comm, ctypes.byref(w), ctypes.byref(r)))
    assert (w.value, r.value) == (1, 0)
    C, cap, n = _lib.PMI_LOC_COLUMNS, 1000, 617
    t = torch.arange(C * cap, dtype=torch.int32, device="cuda").view(C, cap)
    d_n = torch.tensor([n], dtype=torch.int64, device="cuda")
    allt = torch.empty((1, C, cap), dtype=torch.int32, device="cuda")
    counts = torch.zeros(1, dtype=torch.int64, device="cuda")
    _lib.check(L.pmi_allgather_locs(comm, ctypes.c_void_p(t.data_ptr()), C, cap, ctypes.c_void_p(d_n.data_ptr()),
                                    ctypes.c_void_p(allt.data_ptr()), ctypes.c_void_p(counts.data_ptr()), None))
    out = torch.full((C, cap), -1, dtype=torch.int32, device="cuda")
    total = torch.zeros(1, dtype=torch.int64, device="cuda")
    _lib.check(L.pmi_compact_gathered_dev(ctypes.c_void_p(allt.data_ptr()), ctypes.c_void_p(counts.data_ptr()), 1, C, cap,
                                          ctypes.c_void_p(out.data_ptr()), cap, ctypes.c_void_p(total.data_ptr()), None))
    torch.cuda.synchronize()
    assert int(counts.item()) == n and int(total.item()) == n
    assert np.array_equal(out[:, :n].cpu().numpy(), t[:, :n].cpu().numpy()) and int(out[:, n:].max().item()) == -1
    assert L.pmi_allgather_locs(comm, None, C, cap, None, None, None, None) != 0
    _lib.check(L.pmi_comm_destroy(comm))
