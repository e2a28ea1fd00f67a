#!/usr/bin/env python3
"""Kernel time of part r of N of the 4K headline frame on one stream, and of tiny whole frames:
the deepest tile sets a floor under every launch, however small."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
import rt355
lib = rt355.init(0)
scene, cam = rt355.default_scene(), rt355.default_camera()
rt355.set_scene(scene)
W, H = 3840, 2160
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
sptr = ctypes.c_void_p(stream.cuda_stream)
buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
def t(n=100, w=W, h=H, **kw):
    p = rt355.api.make_params(cam, w, h, max_depth=8, **kw)
    for _ in range(5): lib.rt_render_device(ctypes.addressof(p), buf.data_ptr(), buf.numel(), sptr, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(n): assert lib.rt_render_device(ctypes.addressof(p), buf.data_ptr(), buf.numel(), sptr, 0) == 0
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
full = t(part=0, nparts=1)
for N in (1, 2, 4, 8, 16, 32, 64, 135):
    ts = [t(part=r, nparts=N) for r in sorted(set([0, N // 2, N - 1]))]
    print("nparts %3d: kernel ms of parts %s: %s   (whole frame / N = %.4f)" % (
        N, sorted(set([0, N // 2, N - 1])), " ".join("%.4f" % x for x in ts), full / N))
# tiny frames: fixed cost of a launch and the longest single wave
for (w, h) in ((16, 16), (64, 64), (256, 256), (800, 800)):
    print("frame %dx%d: %.4f ms" % (w, h, t(w=w, h=h)))
