#!/usr/bin/env python3
"""Wave-level block profile of the render loop (development aid).

RT_WAVE_PROFILE=1 turns the event counters of the instrumented kernel into counters of the
code blocks each wave executes: "w" = waves that entered the block with at least one lane,
"lanes" = lanes that executed it.  Time of the VALU-issue-bound kernel is proportional to
sum(block size x w), so this shows which blocks to shrink and how coherent the lanes are.

  RT_WAVE_PROFILE=1 python tools/wave_profile.py [--w 3840 --h 2160 --max-depth 8 --stack-size 3]
"""
import argparse
import os
import sys

os.environ["RT_WAVE_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ["iter", "iter_lanes", "pair", "cand", "cand_lanes", "tbound", "sphere", "sphere_lanes", "sphere_root2",
         "torus", "torus_lanes", "plane_div", "path", "path_lanes", "sky", "hit", "hit_lanes", "normal_sphere",
         "normal_torus", "fresnel_pow", "fresnel_div", "local", "shadow", "shadow_lanes", "spawn", "spawn_lanes",
         "phong", "phong_lanes", "children", "refr", "refr_push", "refl", "pop", "pop_load", "finish", "start"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=3840)
    ap.add_argument("--h", type=int, default=2160)
    ap.add_argument("--max-depth", type=int, default=8)
    ap.add_argument("--stack-size", type=int, default=3)
    ap.add_argument("--shadow-hops", type=int, default=2)
    ap.add_argument("--scene", default="default", choices=["default", "tori", "spheres", "mixed"])
    args = ap.parse_args()
    import torch  # noqa: F401  (before librt355: one HIP runtime per process)
    import rt355
    scene, cam = {"default": (rt355.default_scene, rt355.default_camera), "tori": (rt355.tori_stress_scene, rt355.tori_stress_camera),
                  "spheres": (rt355.spheres_deep_scene, rt355.spheres_deep_camera),
                  "mixed": (rt355.mixed_scene, rt355.mixed_camera)}[args.scene]
    ev = rt355.count_events(scene(), cam(), args.w, args.h, max_depth=args.max_depth,
                            stack_size=args.stack_size, shadow_hops=args.shadow_hops)
    waves = ((args.w + 15) // 16) * ((args.h + 15) // 16) * 4
    print(f"waves {waves}")
    for n, v in zip(NAMES, ev):
        print(f"{n:16s} {int(v):12d}  per wave {int(v) / waves:8.3f}")


if __name__ == "__main__":
    main()
