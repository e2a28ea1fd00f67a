#!/usr/bin/env python3
"""Kernel A/B harness: time the render kernel of one or more builds of librt355 on the
headline workload (each build in its own process, interleaved rounds) and check that the
strict frame still hashes to the reference's golden SHA-256.

  python tools/kbench.py [--libs a.so b.so ...] [--rounds 5] [--iters 50] [--config 4k_d8s3]
"""
import argparse
import ctypes
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    "4k_d8s3": dict(scene="default", w=3840, h=2160, max_depth=8, stack_size=3, shadow_hops=2, golden="default_4k_d8s3"),
    "4k_d8s8": dict(scene="default", w=3840, h=2160, max_depth=8, stack_size=8, shadow_hops=2, golden="default_4k_d8s8"),
    "4k_d4s3": dict(scene="default", w=3840, h=2160, max_depth=4, stack_size=3, shadow_hops=2, golden="default_4k_d4s3"),
    "4k_d1h0": dict(scene="default", w=3840, h=2160, max_depth=1, stack_size=3, shadow_hops=0, golden=None),
    "4k_d1h2": dict(scene="default", w=3840, h=2160, max_depth=1, stack_size=3, shadow_hops=2, golden=None),
    "4k_d2h2": dict(scene="default", w=3840, h=2160, max_depth=2, stack_size=3, shadow_hops=2, golden=None),
    "4k_d8s3h0": dict(scene="default", w=3840, h=2160, max_depth=8, stack_size=3, shadow_hops=0, golden=None),
    "4k_empty": dict(scene="empty", w=3840, h=2160, max_depth=8, stack_size=3, shadow_hops=2, golden=None),
    "tori64_4k": dict(scene="tori", w=3840, h=2160, max_depth=8, stack_size=3, shadow_hops=2, golden=None),
    "spheres256_4k": dict(scene="spheres", w=3840, h=2160, max_depth=16, stack_size=16, shadow_hops=4, golden=None),
    "spheres256_1080": dict(scene="spheres", w=1920, h=1080, max_depth=16, stack_size=16, shadow_hops=4, golden=None),
    "mixed1024_1080": dict(scene="mixed", w=1920, h=1080, max_depth=8, stack_size=8, shadow_hops=2, golden=None),
    "mixed1024_4k_d4": dict(scene="mixed", w=3840, h=2160, max_depth=4, stack_size=3, shadow_hops=2, golden=None),
    **{f"mixed{n}_4k_d4": dict(scene=f"mixed:{n}", w=3840, h=2160, max_depth=4, stack_size=3, shadow_hops=2, golden=None)
       for n in (12, 16, 24, 32, 48, 64, 96, 128)},
    **{f"mixed{n}_4k_d8": dict(scene=f"mixed:{n}", w=3840, h=2160, max_depth=8, stack_size=8, shadow_hops=2, golden=None)
       for n in (48, 96, 256, 512)},
    "tori64_4k_d4": dict(scene="tori", w=3840, h=2160, max_depth=4, stack_size=3, shadow_hops=2, golden=None),
    "tori64_4k_d2": dict(scene="tori", w=3840, h=2160, max_depth=2, stack_size=3, shadow_hops=2, golden=None),
    "4k_d8s3_spp4": dict(scene="default", w=3840, h=2160, max_depth=8, stack_size=3, shadow_hops=2, spp_grid=4, golden=None),
    "1080_d8s3_spp4": dict(scene="default", w=1920, h=1080, max_depth=8, stack_size=3, shadow_hops=2, spp_grid=4, golden=None),
    "4k_d16s16": dict(scene="default", w=3840, h=2160, max_depth=16, stack_size=16, shadow_hops=2, golden=None),
    "tori64_4k_d8s8": dict(scene="tori", w=3840, h=2160, max_depth=8, stack_size=8, shadow_hops=2, golden=None),
    "spheres256_1080_d8s8": dict(scene="spheres", w=1920, h=1080, max_depth=8, stack_size=8, shadow_hops=2, golden=None),
    "mixed1024_1080_d16": dict(scene="mixed", w=1920, h=1080, max_depth=16, stack_size=16, shadow_hops=2, golden=None),
    "tori64_1080": dict(scene="tori", w=1920, h=1080, max_depth=8, stack_size=3, shadow_hops=2, golden=None),
    "mixed1024_8k_16spp": dict(scene="mixed", w=7680, h=4320, max_depth=8, stack_size=8, shadow_hops=2, spp_grid=4, golden=None),
}


def child(args):
    os.environ["RT355_LIB"] = args.lib
    import torch  # before librt355: one HIP runtime per process (torch ships its own copy)
    import numpy as np
    import rt355
    cfg = dict(CONFIGS[args.config])
    which = cfg.pop("scene")
    golden = cfg.pop("golden")
    w, h = cfg.pop("w"), cfg.pop("h")
    if which.startswith("mixed:"):
        nobj = int(which.split(":")[1])
        which = "mixedN"
    scene, cam = {"mixedN": (lambda: rt355.mixed_scene(nobj), rt355.mixed_camera), "default": (rt355.default_scene, rt355.default_camera), "empty": (lambda: [], rt355.default_camera), "tori": (rt355.tori_stress_scene, rt355.tori_stress_camera),
                  "spheres": (rt355.spheres_deep_scene, rt355.spheres_deep_camera),
                  "mixed": (rt355.mixed_scene, rt355.mixed_camera)}[which]
    scene, cam = scene(), cam()
    lib = rt355.init(0)
    rt355.set_scene(scene)
    flags = (rt355.api.FLAG_FAST_MATH if args.fast else 0) | rt355.api.FLAG_BOTTOM_UP
    if args.no_accel:
        flags |= rt355.api.FLAG_NO_ACCEL
    p = rt355.api.make_params(cam, w, h, flags=flags, **cfg)
    out = np.empty((h, w, 4), dtype=np.uint8)
    ok = None
    packed = rt355.api.pack_scene(scene)
    assert lib.rt_render(ctypes.addressof(packed), len(scene), ctypes.addressof(p), out.ctypes.data, out.nbytes) == 0, lib.rt_last_error()
    t_host = rt355.api.RtTiming()
    lib.rt_render(ctypes.addressof(packed), len(scene), ctypes.addressof(p), out.ctypes.data, out.nbytes)
    lib.rt_get_timing(ctypes.addressof(t_host))
    if golden and not args.fast:
        meta = json.load(open(os.path.join(ROOT, "tests", "golden", golden + ".json")))
        ok = hashlib.sha256(out.tobytes()).hexdigest() == meta["sha256_rgba8"]
    buf = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda:0")
    t = rt355.api.RtTiming()
    times = []
    for i in range(args.iters):
        assert lib.rt_render_device(ctypes.addressof(p), buf.data_ptr(), buf.numel(), None, 1) == 0, lib.rt_last_error()
        lib.rt_get_timing(ctypes.addressof(t))
        times.append(t.kernel_ms)
    times = np.array(times[2:])
    print(json.dumps({"lib": os.path.relpath(args.lib, ROOT), "config": args.config, "min_ms": float(times.min()),
                      "median_ms": float(np.median(times)), "vgprs": t.vgprs, "cap": t.pending_cap, "sha_ok": ok,
                      "sha": hashlib.sha256(out.tobytes()).hexdigest()[:16], "no_accel": bool(args.no_accel),
                      "host_roundtrip_ms": t_host.total_ms}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="*", default=[os.path.join(ROOT, "ray-tracing_amd", "librt355.so")])
    ap.add_argument("--lib")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--config", default="4k_d8s3")
    ap.add_argument("--fast", action="store_true")
    ap.add_argument("--no-accel", action="store_true")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    for r in range(args.rounds):
        for lib in args.libs:
            cmd = [sys.executable, __file__, "--child", "--lib", os.path.abspath(lib), "--iters", str(args.iters), "--config", args.config]
            if args.fast:
                cmd.append("--fast")
            if args.no_accel:
                cmd.append("--no-accel")
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            line = [l for l in res.stdout.splitlines() if l.startswith("{")]
            print(line[-1] if line else f"FAILED {lib}: {res.stderr[-500:]}", flush=True)


if __name__ == "__main__":
    main()
