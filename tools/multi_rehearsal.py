#!/usr/bin/env python3
"""In-library multi-GPU mode on whatever devices there are: frame time of the 4K headline workload
with the frame split over N parts.  On a one-GPU machine all parts run on GPU 0 (a rehearsal of
the code path: it shows the cost of the split and of the strided copies, not a scaling curve).

  python tools/multi_rehearsal.py [--parts 1 2 4 8]
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--frames", type=int, default=200)
    args = ap.parse_args()
    import torch
    import rt355
    ndev = torch.cuda.device_count()
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "default_4k_d8s3.json")))
    scene, cam = rt355.default_scene(), rt355.default_camera()
    w, h = 3840, 2160
    for n in args.parts:
        devices = [i % ndev for i in range(n)]
        lib = rt355.init_multi(devices=devices)
        rt355.set_scene(scene)
        p = rt355.api.make_params(cam, w, h, max_depth=8, stack_size=3, flags=rt355.api.FLAG_BOTTOM_UP)
        buf = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0")
        st = torch.cuda.Stream()
        sp = ctypes.c_void_p(st.cuda_stream)
        for _ in range(10):
            assert lib.rt_render_device(ctypes.addressof(p), buf.data_ptr(), buf.numel(), sp, 0) == 0, lib.rt_last_error()
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.frames):
            lib.rt_render_device(ctypes.addressof(p), buf.data_ptr(), buf.numel(), sp, 0)
        t_issue = time.perf_counter() - t0
        st.synchronize()
        ms = (time.perf_counter() - t0) / args.frames * 1e3
        ok = hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest() == golden["sha256_rgba8"]
        print(json.dumps({"parts": n, "devices": devices, "ms_per_frame": round(ms, 4), "host_issue_ms_per_frame": round(t_issue / args.frames * 1e3, 4),
                          "frame_sha_ok": ok}), flush=True)
        rt355.shutdown()


if __name__ == "__main__":
    main()
