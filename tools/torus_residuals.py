#!/usr/bin/env python3
"""Accuracy of the torus solve on the 64-tori stress scene: the primary hit distance the
kernel reports (strict flavour = the reference's FP32 Ferrari arithmetic; optionally with the
Newton polish flag, or the fast flavour) against the roots of the same quartic computed in
float64 with a companion-matrix solver.

  python tools/torus_residuals.py [--w 960 --h 540]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from make_torus_f64 import exact_primary  # noqa: E402  (float64 companion-matrix roots)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=960)
    ap.add_argument("--h", type=int, default=540)
    args = ap.parse_args()
    import torch  # noqa: F401
    import numpy as np
    import rt355
    scene, cam = rt355.tori_stress_scene(), rt355.tori_stress_camera()
    t64, id64 = exact_primary(scene, cam, args.w, args.h)
    n = len(scene)
    for label, kw in [("strict (reference arithmetic)", {}), ("strict + Newton polish", dict(newton_polish=True)),
                      ("fast flavour", dict(fast_math=True)), ("fast + Newton polish", dict(fast_math=True, newton_polish=True))]:
        cap = rt355.capture_segment(scene, cam, args.w, args.h, 0, max_depth=8, **kw)
        kid, kt = cap[..., 0].astype(int), cap[..., 1].astype(np.float64)
        torus_k = (kid >= 0) & (kid < n)
        torus_e = id64 >= 0
        same = torus_k & torus_e & (kid == id64)
        rel = np.abs(kt[same] - t64[same]) / t64[same]
        print(f"{label}: kernel torus pixels {int(torus_k.sum())}, float64 torus pixels {int(torus_e.sum())}, same torus "
              f"{int(same.sum())}; kernel-only {int((torus_k & ~torus_e).sum())}, float64-only {int((torus_e & ~torus_k).sum())}, "
              f"different torus {int((torus_k & torus_e & (kid != id64)).sum())}")
        if rel.size:
            print(f"    relative error of t on the common pixels: median {np.median(rel):.2e}, p90 {np.percentile(rel, 90):.2e}, "
                  f"p99 {np.percentile(rel, 99):.2e}, max {rel.max():.2e}; above 1e-4: {float((rel > 1e-4).mean()) * 100:.1f} %, "
                  f"above 1e-3: {float((rel > 1e-3).mean()) * 100:.1f} %")


if __name__ == "__main__":
    main()
