import ctypes, sys, torch
sys.path.insert(0, ".")
from picasso_amd import _lib, synth
L=_lib.load()
for W in (512, 510, 428):
    F=4000
    mov=synth.simulate_movie(F, 512, W, emitters_per_frame=100, device="cuda")
    torch.cuda.synchronize()
    cap=400*F
    out=[torch.empty(cap,dtype=torch.int32,device="cuda") for _ in range(3)]+[torch.empty(cap,dtype=torch.float32,device="cuda")]
    dn=torch.zeros(1,dtype=torch.int64,device="cuda")
    L.pmi_set_kernel_timing(1); a,b=ctypes.c_float(0),ctypes.c_float(0); ts=[]
    for _ in range(4):
        _lib.check(L.pmi_identify_dev(ctypes.c_void_p(mov.data_ptr()),0,F,512,W,7,5000.0,None,0,F-1,*[ctypes.c_void_p(t.data_ptr()) for t in out],cap,ctypes.c_void_p(dn.data_ptr()),None))
        torch.cuda.synchronize(); L.pmi_last_kernel_ms(ctypes.byref(a),ctypes.byref(b)); ts.append(a.value)
    gb=mov.numel()*2/1e9
    print(W, int(dn.item()), f"{min(ts[1:]):.3f} ms {gb/(min(ts[1:])*1e-3):.0f} GB/s")
