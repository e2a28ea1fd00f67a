#!/usr/bin/env python3
"""Soak run: many seeded large random scenes (100-700 objects: nested, duplicated, outlying ones,
cameras inside and far outside), accelerated kernels against the plain per-object loop; float32
frames must be identical.  Every third seed runs as a deep frame (max_depth 12-16, 8-16 pending
rays: pixels spread over waves, cooperative trace; every other one of those with the pixels handed
out in the cost order of a pilot launch).
usage: tools/soak_large.py first_seed last_seed"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (before librt355: one HIP runtime per process)
import numpy as np  # noqa: E402
import rt355  # noqa: E402

a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b):
    scene, cam, o = rt355.random_large_scene(seed)
    if seed % 3 == 2:
        o = dict(o, max_depth=12 + seed % 5, stack_size=8 + seed % 9)
    w, h = 160 + 8 * (seed % 7), 90 + 5 * (seed % 5)
    spp = 2 if seed % 4 == 0 else 1
    ordered = seed % 3 == 2 and seed % 2 == 0
    x = rt355.render_rgba(scene, cam, w, h, out_f32=True, spp_grid=spp, pixel_order=ordered, **o)
    y = rt355.render_rgba(scene, cam, w, h, out_f32=True, spp_grid=spp, no_accel=True, **o)
    same = np.array_equal(np.where(np.isnan(x), np.float32(7), x), np.where(np.isnan(y), np.float32(7), y))
    bad += not same
    print(seed, len(scene), o, w, h, spp, "cost order" if ordered else "", "ok" if same else "MISMATCH", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
