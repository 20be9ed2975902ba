"""Scan-kernel rate over frame shapes and pixel types at about the same number of pixels (1e9): small ROI-cropped
frames, non-square, wide, odd widths; uint8 / int16 movies (packed scan) and float32 (generic kernel).
usage: python tools/time_identify_shapes.py [box]"""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from picasso_amd import _lib, synth  # noqa: E402

L = _lib.load()
BOX = int(sys.argv[1]) if len(sys.argv) > 1 else 7
CASES = [(h, w, "uint16") for h, w in ((64, 64), (128, 128), (256, 256), (512, 512), (1024, 1024), (2048, 2048), (128, 1024),
                                         (1024, 128), (1024, 64), (512, 200), (300, 300), (512, 511), (511, 333))]
CASES += [(512, 512, "uint8"), (512, 511, "uint8"), (512, 512, "int16"), (512, 512, "uint16+20000"), (512, 512, "float32"),
          (512, 512, "float32 x1.37"), (1024, 1024, "float32 x1.37")]
if len(sys.argv) > 2:      # usage: ... box H W dtype
    CASES = [(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else "uint16")]
for H, W, dt in CASES:
    F = max(8, int(1.0e9 / (H * W)))
    mov = synth.simulate_movie(F, H, W, emitters_per_frame=max(1, H * W // 2300), device="cuda", chunk_frames=max(1, 2 ** 25 // (H * W)))
    if dt == "uint8":        # the same scene at an eighth of the counts (fits 8 bits), threshold scaled alike
        mov = (mov.to(torch.int32) // 8).clamp(max=255).to(torch.uint8)
    elif dt == "int16":
        mov = (mov.to(torch.int32) - 20000).to(torch.int16)
    elif dt == "uint16+20000":   # a large camera offset
        mov = (mov.to(torch.int32) + 20000).to(torch.uint16)
    elif dt == "float32":
        mov = mov.view(torch.int16).to(torch.float32)
    elif dt == "float32 x1.37":      # fractions: 16-bit keys + exact decisions on the float32 pixels
        mov = mov.view(torch.int16).to(torch.float32) * 1.37 + 0.25
    elif dt == "int32":              # a camera's counts saved as 32-bit integers
        mov = mov.view(torch.int16).to(torch.int32)
    elif dt == "int32 x70000":       # 32-bit integers beyond 16 bits (they compare as float32 in the reference, localize.py:332)
        mov = mov.view(torch.int16).to(torch.int32) * 70000
    code = {"uint16": 0, "uint16+20000": 0, "uint8": 1, "int16": 2, "float32": 5, "float32 x1.37": 5, "int32": 4, "int32 x70000": 4}[dt]
    min_ng = 5000.0 / 8 if dt == "uint8" else (5000.0 * 70000 if dt == "int32 x70000" else 5000.0)
    torch.cuda.synchronize()
    cap = max(4096, int(F * H * W / 1500))
    out = [torch.empty(cap, dtype=torch.int32, device="cuda") for _ in range(3)] + [torch.empty(cap, dtype=torch.float32, device="cuda")]
    dn = torch.zeros(1, dtype=torch.int64, device="cuda")
    L.pmi_set_kernel_timing(1)
    a, b = ctypes.c_float(0), ctypes.c_float(0)
    ts = []
    for _ in range(4):
        _lib.check(L.pmi_identify_dev(ctypes.c_void_p(mov.data_ptr()), code, F, H, W, BOX, min_ng, None, 0, F - 1,
                                      *[ctypes.c_void_p(t.data_ptr()) for t in out], cap, ctypes.c_void_p(dn.data_ptr()), None), "identify")
        torch.cuda.synchronize()
        L.pmi_last_kernel_ms(ctypes.byref(a), ctypes.byref(b))
        ts.append(a.value)
    gb = mov.numel() * mov.element_size() / 1e9
    print(f"{H:5d} x {W:5d} {dt:13s} frames {F:7d}  rows {int(dn.item()):9d}  scan {min(ts[1:]):8.3f} ms  {gb / (min(ts[1:]) * 1e-3):6.0f} GB/s  "
          f"{mov.numel() / (min(ts[1:]) * 1e-3) / 1e12:5.2f} Tpx/s", flush=True)
    del mov
