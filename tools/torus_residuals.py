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


def exact_primary(scene, cam, w, h):
    """float64 nearest torus hit per pixel: (t, id) arrays, id -1 where no torus is hit."""
    import numpy as np
    m = cam.matrix().astype(np.float64)
    right, up, fwd, eye = m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3]
    ys, xs = np.mgrid[0:h, 0:w]
    uvx = ((xs + 0.5) * 2 - w) / h
    uvy = ((ys + 0.5) * 2 - h) / h
    d = uvx[..., None] * right + uvy[..., None] * up + cam.focal_length * fwd
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    best_t = np.full((h, w), np.inf)
    best_id = np.full((h, w), -1)
    for i, o in enumerate(scene):
        if o["type"] != 2:
            continue
        c = np.array(o["center"], dtype=np.float64)
        wv = np.array(o["normal"], dtype=np.float64)
        wv /= np.linalg.norm(wv)
        helper = np.array([1.0, 0, 0]) if abs(wv[1]) > 0.99 else np.array([0, 1.0, 0])
        u = np.cross(wv, helper); u /= np.linalg.norm(u)
        v = np.cross(wv, u)
        R, r = float(o["major_radius"]), float(o["minor_radius"])
        oc = eye - c
        b = d @ oc
        disc = b * b - (oc @ oc - (R + r) ** 2)
        idx = np.argwhere(disc >= 0)
        if len(idx) == 0:
            continue
        dd = d[idx[:, 0], idx[:, 1]]
        ro = np.array([oc @ u, oc @ wv, oc @ v])
        rd = np.stack([dd @ u, dd @ wv, dd @ v], axis=1)
        mm = ro @ ro
        n = rd @ ro
        k = mm + R * R - r * r
        rdxz = rd[:, 0] ** 2 + rd[:, 2] ** 2
        rordxz = ro[0] * rd[:, 0] + ro[2] * rd[:, 2]
        roxz = ro[0] ** 2 + ro[2] ** 2
        A = 4 * n
        B = 2 * k + 4 * n * n - 4 * R * R * rdxz
        C = 4 * n * k - 8 * R * R * rordxz
        D = np.full_like(A, k * k - 4 * R * R * roxz)
        for j in range(len(idx)):
            roots = np.roots([1.0, A[j], B[j], C[j], D[j]])
            real = roots[np.abs(roots.imag) < 1e-9 * (1 + np.abs(roots.real))].real
            real = real[real > 0.001]
            if len(real):
                t = real.min()
                y, x = idx[j]
                if t < best_t[y, x]:
                    best_t[y, x] = t
                    best_id[y, x] = i
    return best_t, best_id


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
