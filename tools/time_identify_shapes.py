"""Scan-kernel rate over frame shapes at about the same total bytes (2 GB): small ROI-cropped frames, non-square,
wide, odd widths (generic path).  usage: python tools/time_identify_shapes.py"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from picasso_amd import _lib, synth  # noqa: E402

L = _lib.load()
for H, W in ((64, 64), (128, 128), (256, 256), (512, 512), (1024, 1024), (2048, 2048), (128, 1024), (1024, 128), (1024, 64), (512, 200), (300, 300), (512, 511)):
    F = max(8, int(2.0e9 / (H * W * 2)))
    mov = synth.simulate_movie(F, H, W, emitters_per_frame=max(1, H * W // 2300), device="cuda", chunk_frames=max(1, 2 ** 25 // (H * W)))
    torch.cuda.synchronize()
    cap = max(4096, int(F * H * W / 1500))
    out = [torch.empty(cap, dtype=torch.int32, device="cuda") for _ in range(3)] + [torch.empty(cap, dtype=torch.float32, device="cuda")]
    dn = torch.zeros(1, dtype=torch.int64, device="cuda")
    L.pmi_set_kernel_timing(1)
    a, b = ctypes.c_float(0), ctypes.c_float(0)
    ts = []
    for _ in range(4):
        _lib.check(L.pmi_identify_dev(ctypes.c_void_p(mov.data_ptr()), 0, F, H, W, 7, 5000.0, None, 0, F - 1,
                                      *[ctypes.c_void_p(t.data_ptr()) for t in out], cap, ctypes.c_void_p(dn.data_ptr()), None), "identify")
        torch.cuda.synchronize()
        L.pmi_last_kernel_ms(ctypes.byref(a), ctypes.byref(b))
        ts.append(a.value)
    gb = mov.numel() * 2 / 1e9
    print(f"{H:5d} x {W:5d}  frames {F:7d}  rows {int(dn.item()):9d}  scan {min(ts[1:]):8.3f} ms  {gb / (min(ts[1:]) * 1e-3):6.0f} GB/s", flush=True)
    del mov
