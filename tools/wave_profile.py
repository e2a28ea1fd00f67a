#!/usr/bin/env python3
"""Wave-level block profile of the render loop (development aid; needs the development build:
`make -C ray-tracing_amd/csrc dev` -> build/dev/librt355.so).

The instrumented kernel counts, per code block of the loop, the waves that enter it with at least
one lane ("NAME") and the lanes that execute it ("NAME_LANES").  A wave issues a block's
instructions once however many of its lanes are in it, so sum(block cost x NAME) is what the
wave's instruction stream holds; tools/isa_census.py joins these counts with the per-block ISA.

  python tools/wave_profile.py [--w 3840 --h 2160 --max-depth 8 --stack-size 3] > profile.txt
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RT355_LIB", os.path.join(ROOT, "build", "dev", "librt355.so"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=3840)
    ap.add_argument("--h", type=int, default=2160)
    ap.add_argument("--max-depth", type=int, default=8)
    ap.add_argument("--stack-size", type=int, default=3)
    ap.add_argument("--shadow-hops", type=int, default=2)
    ap.add_argument("--scene", default="default", choices=["default", "tori", "spheres", "mixed"])
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401  (before librt355: one HIP runtime per process)
    import rt355
    scene, cam = {"default": (rt355.default_scene, rt355.default_camera), "tori": (rt355.tori_stress_scene, rt355.tori_stress_camera),
                  "spheres": (rt355.spheres_deep_scene, rt355.spheres_deep_camera),
                  "mixed": (rt355.mixed_scene, rt355.mixed_camera)}[args.scene]
    lib = rt355.init(0)
    if not hasattr(lib, "rt_wave_profile"):
        sys.exit(f"{rt355.api.LIB_PATH} is not a development build (make -C ray-tracing_amd/csrc dev)")
    lib.rt_wave_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.rt_wave_profile_name.restype = ctypes.c_char_p
    lib.rt_wave_profile_name.argtypes = [ctypes.c_int]
    rt355.set_scene(scene())
    p = rt355.api.make_params(cam(), args.w, args.h, max_depth=args.max_depth, stack_size=args.stack_size,
                              shadow_hops=args.shadow_hops)
    counts = np.zeros(64, dtype=np.uint64)
    rc = lib.rt_wave_profile(ctypes.addressof(p), counts.ctypes.data)
    assert rc == 0, lib.rt_last_error()
    waves = ((args.w + 15) // 16) * ((args.h + 15) // 16) * 4
    print(f"waves {waves}")
    for i in range(lib.rt_wave_profile_blocks()):
        print(f"{lib.rt_wave_profile_name(i).decode():16s} {int(counts[i]):12d}  per wave {int(counts[i]) / waves:8.3f}")


if __name__ == "__main__":
    main()
