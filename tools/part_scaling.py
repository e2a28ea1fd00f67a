#!/usr/bin/env python3
"""What the band split over N GPUs looks like from one GPU: every part r of N = 1, 2, 4, 8 of a
benchmark frame is rendered alone (what GPU r of N would render) and its kernel time measured with
HIP events on the launch stream.  Printed per N: the parts' times, their maximum (the frame's kernel
time on N GPUs, no transfer), balance = sum of the parts / (N x slowest part), the predicted
kernel-only speed-up t(1) / slowest part and its efficiency, and the bytes each part sends to GPU 0
(packed RGB8).  The headline workload adds the floor under any launch (tiny whole frames).

  python tools/part_scaling.py [--workload default4k|tori64_4k|spheres256_4k|mixed1024_8k] [--iters N]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import rt355  # noqa: E402

WORKLOADS = {
    "default4k": ("default", 3840, 2160, dict(max_depth=8, stack_size=3, shadow_hops=2)),
    "tori64_4k": ("tori", 3840, 2160, dict(max_depth=8, stack_size=3, shadow_hops=2)),
    "spheres256_4k": ("spheres", 3840, 2160, dict(max_depth=16, stack_size=16, shadow_hops=4)),
    "mixed1024_8k": ("mixed", 7680, 4320, dict(max_depth=8, stack_size=8, shadow_hops=2, spp_grid=4)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="default4k", choices=sorted(WORKLOADS))
    ap.add_argument("--iters", type=int, default=0, help="launches per measurement (default: by the frame's cost)")
    args = ap.parse_args()
    which, W, H, o = WORKLOADS[args.workload]
    scene, cam = {"default": (rt355.default_scene, rt355.default_camera), "tori": (rt355.tori_stress_scene, rt355.tori_stress_camera),
                  "spheres": (rt355.spheres_deep_scene, rt355.spheres_deep_camera),
                  "mixed": (rt355.mixed_scene, rt355.mixed_camera)}[which]
    lib = rt355.init(0)
    rt355.set_scene(scene())
    cam = cam()
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    sptr = ctypes.c_void_p(stream.cuda_stream)
    buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")

    def t(n, w=W, h=H, **kw):
        p = rt355.api.make_params(cam, w, h, flags=rt355.api.FLAG_OUT_RGB8, **dict(o, **kw))
        for _ in range(max(1, n // 10)):
            assert lib.rt_render_device(ctypes.addressof(p), buf.data_ptr(), buf.numel(), sptr, 0) == 0, lib.rt_last_error()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            assert lib.rt_render_device(ctypes.addressof(p), buf.data_ptr(), buf.numel(), sptr, 0) == 0
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n, int(lib.rt_output_bytes(ctypes.addressof(p)))

    full, _ = t(3)
    iters = args.iters or max(2, min(200, int(300.0 / full)))          # about 0.3 s per measurement
    full, full_bytes = t(iters)
    print(f"{args.workload}: {W}x{H} {o}; whole frame {full:.4f} ms ({iters} launches per measurement), {full_bytes} bytes as RGB8")
    for N in (2, 4, 8):
        ts, nbytes = zip(*[t(iters, part=r, nparts=N) for r in range(N)])
        worst = max(ts)
        print(f"N = {N}: parts ms " + " ".join(f"{x:.4f}" for x in ts))
        print(f"        slowest part {worst:.4f} ms (whole frame / N = {full / N:.4f}); balance {sum(ts) / (N * worst):.3f}; "
              f"sum of the parts / whole frame {sum(ts) / full:.3f}; kernel-only speed-up {full / worst:.2f}x = {100 * full / worst / N:.0f} % of linear; "
              f"bytes per part to GPU 0: {max(nbytes)}")
    if args.workload == "default4k":
        # the floor under every launch: the longest single wave (deepest tile) of a tiny whole frame
        for (w, h) in ((16, 16), (64, 64), (256, 256), (800, 800)):
            print("whole frame %dx%d: %.4f ms" % (w, h, t(iters, w=w, h=h)[0]))


if __name__ == "__main__":
    main()
