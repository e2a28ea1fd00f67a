#!/usr/bin/env python3
"""Kernel time per frame of part 0 of N of the 4K headline frame when consecutive frames are
launched on 1, 2, 3 or 4 streams (what the band pipeline's two streams buy on one GPU)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
import rt355
lib = rt355.init(0)
scene, cam = rt355.default_scene(), rt355.default_camera()
rt355.set_scene(scene)
W, H = 3840, 2160
S = [torch.cuda.Stream() for _ in range(4)]
bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(4)]
def t(ns, n=200, **kw):
    p = rt355.api.make_params(cam, W, H, max_depth=8, **kw)
    pp = ctypes.addressof(p)
    sp = [ctypes.c_void_p(s.cuda_stream) for s in S]
    for i in range(8): lib.rt_render_device(pp, bufs[i % ns].data_ptr(), bufs[0].numel(), sp[i % ns], 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(S[0])
    for j in range(1, ns): S[j].wait_event(e0)
    for i in range(n): assert lib.rt_render_device(pp, bufs[i % ns].data_ptr(), bufs[0].numel(), sp[i % ns], 0) == 0
    for j in range(1, ns): S[0].wait_stream(S[j])
    e1.record(S[0]); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
full = t(1, part=0, nparts=1)
for N in (1, 2, 4, 8):
    print("nparts %d, part 0, kernel ms per frame: 1 stream %.4f  2 streams %.4f  3 streams %.4f  4 streams %.4f   (whole frame / N = %.4f)" % (
        N, t(1, part=0, nparts=N), t(2, part=0, nparts=N), t(3, part=0, nparts=N), t(4, part=0, nparts=N), full / N))
