"""One-rank RCCL smoke: init the nccl (= RCCL) process group and run the collectives dist.py and bench.py use.
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_probe.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, ".")
from picasso_amd import dist as pdist  # noqa: E402
from picasso_amd.backend import LOC_COLUMNS  # noqa: E402

torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dist.init_process_group(backend="nccl")
dev = torch.device("cuda", torch.cuda.current_device())
t = torch.arange(len(LOC_COLUMNS) * 10, dtype=torch.int32, device=dev).view(len(LOC_COLUMNS), 10)
out = pdist.allgather_table(t, 7)
assert out.shape == (len(LOC_COLUMNS), 7) and torch.equal(out, t[:, :7])
# the library's own communicator (RCCL called from C): padded gather + device-side compaction
comm = pdist.NativeComm.for_group(None)
d_n = torch.tensor([7], dtype=torch.int64, device=dev)
allt, counts = comm.allgather_table(t.contiguous(), d_n)
packed, total = comm.compact(allt, counts)
torch.cuda.synchronize()
assert counts.tolist() == [7] * dist.get_world_size() and int(total.item()) == 7 * dist.get_world_size()
assert torch.equal(packed[:, :7], t[:, :7])
a = pdist._all_reduce_sum(np.ones((3, 4)), dev)
assert np.array_equal(a, np.ones((3, 4)))
g = torch.empty((1,), dtype=torch.int64, device=dev)
dist.all_gather_into_tensor(g, torch.tensor([5], dtype=torch.int64, device=dev))
# the pipelined shard path (frame ranges, asynchronous gathers) on a synthetic movie, against the plain one
from picasso_amd import synth  # noqa: E402
cam = {"Baseline": 100.0, "Sensitivity": 1.0, "Gain": 1.0}
params = {"Box Size": 7, "Min. Net Gradient": 5000}
mov = synth.simulate_movie(300, 256, 256, emitters_per_frame=30, device=dev)
plain = pdist.localize_sharded(mov, 50, cam, params)
piped = pdist.localize_sharded(mov, 50, cam, params, chunks=4)
assert len(plain["frame"]) > 1000 and all(np.array_equal(plain[c], piped[c], equal_nan=True) for c in plain)
dist.barrier()
print("rccl probe ok: world", dist.get_world_size(), "gathered", int(g.item()))
dist.destroy_process_group()
