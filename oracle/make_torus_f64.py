"""float64 reference for the torus solve: nearest torus hit of every primary ray of the 64-tori
stress scene, from the roots of the same quartic (fragment_shader.glsl:325-339) computed in float64
with a companion-matrix solver (numpy.roots).  TEST INFRASTRUCTURE ONLY; needs neither the reference
checkout nor a GPU.  Writes tests/golden/tori64_f64_240.npz (t as float64, torus id as int16, -1 =
no torus hit; rows bottom-up like every fixture).

Usage:  python oracle/make_torus_f64.py [--w 240 --h 135]
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def exact_primary(scene, cam, w, h):
    """float64 nearest torus hit per pixel: (t, id) arrays, id -1 where no torus is hit."""
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
    ap.add_argument("--w", type=int, default=240)
    ap.add_argument("--h", type=int, default=135)
    args = ap.parse_args()
    import rt355
    scene, cam = rt355.tori_stress_scene(), rt355.tori_stress_camera()
    t64, id64 = exact_primary(scene, cam, args.w, args.h)
    out = os.path.join(ROOT, "tests", "golden", f"tori64_f64_{args.w}.npz")
    np.savez_compressed(out, t=t64, id=id64.astype(np.int16))
    print(f"{out}: {int((id64 >= 0).sum())} of {args.w * args.h} primary rays hit a torus")


if __name__ == "__main__":
    main()
