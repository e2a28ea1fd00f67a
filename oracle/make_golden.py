"""Generate tests/golden/* by running the reference's own shader on Mesa llvmpipe.

TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference is mounted and
oracle/_ref/liblp_harness.so is built (`make -C oracle ref`).  The fixtures it writes are
data (inputs + the reference's outputs); no reference source text is stored.

Every image is kept in GL read-back order: row 0 is the BOTTOM row of the frame.

Usage:  python oracle/make_golden.py [--only NAME ...] [--check-oracle]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import llvmpipe as lp          # noqa: E402
import rt_oracle as orc        # noqa: E402
import rt355                   # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def cam_dict(cam):
    return {"eye": [float(v) for v in cam.eye], "yaw": float(cam.yaw), "pitch": float(cam.pitch),
            "focal_length": float(cam.focal_length), "light_pos": [float(v) for v in cam.light_pos]}


# name -> (scene factory, camera, w, h, max_depth, stack_size, hops, store)
# store: "full" keeps RGBA8 + float32 + segment captures, "rgb" keeps RGB8 only,
#        "sha" keeps hashes and probe pixels only.
def cases():
    C = rt355.Camera
    d = rt355.default_scene
    dc = rt355.default_camera()
    out = {
        "default_800": (d, dc, 800, 800, 4, 3, 2, "rgb+seg"),
        "default_4k_d8s3": (d, dc, 3840, 2160, 8, 3, 2, "sha"),
        "default_4k_d8s8": (d, dc, 3840, 2160, 8, 8, 2, "sha"),
        "default_4k_d4s3": (d, dc, 3840, 2160, 4, 3, 2, "sha"),
        "default_320_d8s3": (d, dc, 320, 240, 8, 3, 2, "full"),
        "default_320_d8s8": (d, dc, 320, 240, 8, 8, 2, "full"),
        "default_320_d1": (d, dc, 320, 240, 1, 3, 2, "full"),
        "default_320_s1": (d, dc, 320, 240, 4, 1, 2, "full"),
        "default_320_hops0": (d, dc, 320, 240, 4, 3, 0, "full"),
        "default_320_hops4": (d, dc, 320, 240, 4, 3, 4, "full"),
        "cam_inside_glass": (d, C(eye=(0.0, 0.0, -0.6), yaw=-90.0, pitch=0.0), 320, 240, 4, 3, 2, "full"),
        "cam_down": (d, C(eye=(0.0, 4.0, -2.0), yaw=-90.0, pitch=-89.0), 320, 240, 4, 3, 2, "full"),
        "cam_up": (d, C(eye=(4.0, 1.2, 3.0), yaw=-135.0, pitch=60.0), 320, 240, 4, 3, 2, "full"),
        "cam_zoom10": (d, C(eye=(4.0, 1.2, 3.0), yaw=-135.0, pitch=-12.0, focal_length=10.0), 320, 240, 4, 3, 2, "full"),
        "cam_wide": (d, C(eye=(4.0, 1.2, 3.0), yaw=-135.0, pitch=-12.0, focal_length=0.5), 320, 240, 4, 3, 2, "full"),
        "cam_below_ground": (d, C(eye=(2.0, -2.0, 1.0), yaw=-135.0, pitch=20.0), 320, 240, 4, 3, 2, "full"),
        "cam_light_moved": (d, C(eye=(4.0, 1.2, 3.0), yaw=-135.0, pitch=-12.0, light_pos=(3.0, 2.5, -1.0)), 320, 240, 4, 3, 2, "full"),
        "aspect_400x200": (d, dc, 400, 200, 4, 3, 2, "full"),
        "aspect_150x300": (d, dc, 150, 300, 4, 3, 2, "full"),
        "tiny_1x1": (d, dc, 1, 1, 4, 3, 2, "full"),
        "tiny_7x5": (d, dc, 7, 5, 4, 3, 2, "full"),
        "empty_scene": (lambda: [], dc, 160, 120, 4, 3, 2, "full"),
        "tori64_480": (rt355.tori_stress_scene, rt355.tori_stress_camera(), 480, 270, 8, 3, 2, "full"),
        "tori64_480_d8s8": (rt355.tori_stress_scene, rt355.tori_stress_camera(), 480, 270, 8, 8, 2, "full"),
        # the 64-tori benchmark configuration at exactly bench.py's parameters (hashes only)
        "tori64_4k_d8s3": (rt355.tori_stress_scene, rt355.tori_stress_camera(), 3840, 2160, 8, 3, 2, "sha"),
        # llvmpipe stops every shader loop once 65535 loop iterations (all loops of one 8-pixel
        # vector together) have run, so it is only a valid reference while a frame stays under
        # that: deep-stack behaviour is pinned on a 12-sphere cut of the scene.
        "spheres12_d16s16h4": (lambda: rt355.spheres_deep_scene(12, seed=7), rt355.Camera(eye=(0.0, 1.0, 1.0), yaw=-90.0, pitch=-5.0), 240, 135, 16, 16, 4, "full"),
        "spheres12_d16s4h4": (lambda: rt355.spheres_deep_scene(12, seed=7), rt355.Camera(eye=(0.0, 1.0, 1.0), yaw=-90.0, pitch=-5.0), 240, 135, 16, 4, 4, "full"),
        "spheres256_320_d4": (rt355.spheres_deep_scene, rt355.spheres_deep_camera(), 320, 180, 4, 3, 2, "full"),
        "mixed1022_160_d2": (lambda: rt355.mixed_scene(1022), rt355.mixed_camera(), 160, 90, 2, 3, 2, "full"),
        "mixed200_240_d5s5": (lambda: rt355.mixed_scene(200), rt355.mixed_camera(), 240, 135, 5, 5, 2, "full"),
    }
    # random nasty scenes (rt355.random_scene): pin the restatement beyond the hand-picked cases
    for seed in range(8):
        sc, cam, o = rt355.random_scene(seed)
        out[f"random_{seed}"] = ((lambda sc=sc: sc), cam, 160, 120, o["max_depth"], o["stack_size"], o["shadow_hops"], "full")
    return out


PROBES_PER_CASE = 64
MAX_SEGS = 6


def run_case(name, spec, check_oracle):
    factory, cam, w, h, depth, stack, hops, store = spec
    scene = factory()
    M = rt355.camera_to_world(cam.eye, cam.yaw, cam.pitch)
    n = len(scene)
    t0 = time.time()
    prog = lp.Program(max(n, 1), depth, stack, hops)
    # NUM_OBJECTS cannot be 0 in GLSL: the empty scene is one sphere of radius 0 far away,
    # which no ray can hit (disc < 0) -> identical to an empty list.
    lp_scene = scene if n else [{"type": 1, "center": [0.0, -1e6, 0.0], "radius": 0.0, "color": [0, 0, 0],
                                 "reflectivity": 0.0, "transparency": 0.0, "refractive_index": 1.0}]
    prog.set_scene(lp_scene)
    prog.set_camera(M, cam.focal_length, w, h, cam.light_pos)
    rgba8 = prog.draw_rgba8(w, h)
    f32 = prog.draw_rgba32f(w, h)
    meta = {"name": name, "w": w, "h": h, "max_depth": depth, "stack_size": stack, "shadow_hops": hops,
            "camera": cam_dict(cam), "n_objects": n, "sha256_rgba8": sha(rgba8), "sha256_rgba32f": sha(f32),
            "row_order": "bottom-up", "gl": lp.gl_info()}
    arrays = {}
    if "seg" in store or store == "full":
        cap = lp.Program(max(n, 1), depth, stack, hops, capture=True)
        cap.set_scene(lp_scene)
        cap.set_camera(M, cam.focal_length, w, h, cam.light_pos)
        nseg_max = 0
        for k in range(MAX_SEGS):
            cap.set_capture_segment(k)
            c = cap.draw_rgba32f(w, h)
            ids = c[..., 0].astype(np.int16)
            if n == 0:
                ids[ids == 1] = 0          # plane id is NUM_OBJECTS of the stand-in scene
            arrays[f"seg{k}_id"] = ids
            if k == 0:
                arrays["seg0_t"] = c[..., 1].copy()
                arrays["nseg"] = c[..., 3].astype(np.int16)
                nseg_max = int(c[..., 3].max())
        meta["nseg_max"] = nseg_max
    if store == "full":
        arrays["rgba8"] = rgba8
        arrays["rgba32f"] = f32
    elif store.startswith("rgb"):
        arrays["rgb8"] = np.ascontiguousarray(rgba8[..., :3])
    rng = np.random.default_rng(int(hashlib.sha256(name.encode()).hexdigest()[:8], 16))
    k = min(PROBES_PER_CASE, w * h)
    px = rng.integers(0, w, size=k)
    py = rng.integers(0, h, size=k)
    meta["probes"] = [{"x": int(x), "y": int(y), "rgba8": rgba8[y, x].tolist(),
                       "rgba32f_bits": f32[y, x].view(np.uint32).tolist()} for x, y in zip(px, py)]
    dt = time.time() - t0
    msg = f"{name}: {w}x{h} n={n} d{depth} s{stack} h{hops}  llvmpipe {dt:.1f}s"
    if check_oracle:
        t1 = time.time()
        o = orc.render(scene, M, cam.focal_length, w, h, light=cam.light_pos, max_depth=depth,
                       stack_size=stack, shadow_hops=hops, want_f32=True)
        bad8 = int(np.count_nonzero(np.any(o["rgba8"] != rgba8, axis=2)))
        badf = int(np.count_nonzero(np.any(o["f32"].view(np.uint32)[..., :3] != f32.view(np.uint32)[..., :3], axis=2)))
        msg += f" | oracle {time.time() - t1:.2f}s: rgba8 mismatches {bad8}, f32-bit mismatches {badf}"
        meta["oracle_check"] = {"rgba8_mismatch_pixels": bad8, "f32_bit_mismatch_pixels": badf}
    print(msg, flush=True)
    os.makedirs(GOLDEN, exist_ok=True)
    with open(os.path.join(GOLDEN, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1)
    if arrays:
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **arrays)


def spp_case(name, N, check_oracle):
    """N x N stratified samples: the reference shader rendered N*N times with gl_FragCoord
    shifted by the sample's sub-pixel offset, accumulated in float32 in sample order
    (row-major over the grid) and scaled by 1/N^2 before the framebuffer clamp."""
    scene = rt355.default_scene()
    cam = rt355.default_camera()
    w, h = 200, 150
    M = rt355.camera_to_world(cam.eye, cam.yaw, cam.pitch)
    prog = lp.Program(len(scene), 4, 3, 2, jitter=True)
    prog.set_scene(scene)
    prog.set_camera(M, cam.focal_length, w, h, cam.light_pos)
    acc = np.zeros((h, w, 4), dtype=np.float32)
    offs = [np.float32((i + 0.5) / N - 0.5) for i in range(N)]
    for sj in range(N):
        for si in range(N):
            prog.set_subpixel(float(offs[si]), float(offs[sj]))
            acc = acc + prog.draw_rgba32f(w, h)
    mean = acc * np.float32(1.0 / (N * N))
    meta = {"name": name, "w": w, "h": h, "spp_grid": N, "max_depth": 4, "stack_size": 3, "shadow_hops": 2,
            "camera": cam_dict(cam), "sha256_rgb32f": sha(mean[..., :3]), "row_order": "bottom-up"}
    msg = f"{name}: spp {N}x{N}"
    if check_oracle:
        o = orc.render(scene, M, cam.focal_length, w, h, spp_grid=N, want_f32=True)
        badf = int(np.count_nonzero(np.any(o["f32"].view(np.uint32)[..., :3] != mean.view(np.uint32)[..., :3], axis=2)))
        msg += f" | oracle f32-bit mismatches {badf}"
        meta["oracle_check"] = {"f32_bit_mismatch_pixels": badf}
    print(msg, flush=True)
    with open(os.path.join(GOLDEN, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), rgb32f=np.ascontiguousarray(mean[..., :3]))


def light_case(name, G, size, check_oracle):
    """Square area light, G x G samples: the reference has one point light (main.py:278), but its
    ray paths, sky and ambient terms do not depend on the light and the colour is linear in the
    (diffuse + specular) * attenuation term (fragment_shader.glsl:513, :603).  The fixture is the
    mean (float64, stored as float32) of the G*G unclamped RGBA32F frames the reference shader
    renders with u_light_pos at the G*G sample positions; an area-light frame equals it up to
    rounding."""
    scene = rt355.default_scene()
    cam = rt355.default_camera()
    w, h = 160, 120
    M = rt355.camera_to_world(cam.eye, cam.yaw, cam.pitch)
    prog = lp.Program(len(scene), 4, 3, 2)
    prog.set_scene(scene)
    offs = [np.float32(np.float32((i + 0.5) / G - 0.5) * np.float32(size)) for i in range(G)]
    lx, ly, lz = (np.float32(v) for v in cam.light_pos)
    acc = np.zeros((h, w, 4), dtype=np.float64)
    for j in range(G):
        for i in range(G):
            light = (float(np.float32(lx + offs[i])), float(ly), float(np.float32(lz + offs[j])))
            prog.set_camera(M, cam.focal_length, w, h, light)
            acc += prog.draw_rgba32f(w, h)
    mean = (acc / (G * G)).astype(np.float32)
    meta = {"name": name, "w": w, "h": h, "light_grid": G, "light_size": size, "max_depth": 4, "stack_size": 3,
            "shadow_hops": 2, "camera": cam_dict(cam), "sha256_rgb32f": sha(mean[..., :3]), "row_order": "bottom-up"}
    msg = f"{name}: light {G}x{G}, edge {size}"
    if check_oracle:
        o = orc.render(scene, M, cam.focal_length, w, h, light=cam.light_pos, light_grid=G, light_size=size, want_f32=True)["f32"]
        err = np.abs(o[..., :3].astype(np.float64) - mean[..., :3])
        msg += f" | oracle max abs difference {err.max():.3g}, max relative {np.max(err / np.maximum(np.abs(mean[..., :3]), 1e-3)):.3g}"
        meta["oracle_check"] = {"max_abs_difference": float(err.max())}
    print(msg, flush=True)
    with open(os.path.join(GOLDEN, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), rgb32f=np.ascontiguousarray(mean[..., :3]))


def quartic_kats(check_oracle):
    """Known answers for solve_quartic (fragment_shader.glsl:250-305): the reference's solver
    functions with a test main() that evaluates them on uniform coefficient sets."""
    import re
    fs = lp.read_reference_shader("fragment_shader.glsl")
    vs = lp.read_reference_shader("vertex_shader.glsl")
    head = fs[:fs.index("void main()")]
    K = 256
    test_main = f"""
uniform vec4 u_coef[{K}];
void main() {{
    int i = int(gl_FragCoord.x);
    vec4 r = vec4(0.0);
    int n = solve_quartic(u_coef[i].x, u_coef[i].y, u_coef[i].z, u_coef[i].w, r);
    float t = 1e20;
    for (int k = 0; k < n; k++) if (r[k] > 0.001 && r[k] < t) t = r[k];
    FragColor = vec4(float(n), t, n > 0 ? r[0] : 0.0, n > 1 ? r[1] : 0.0);
}}
"""
    lib = lp._load()
    assert lib.lp_program(vs.encode(), (head + test_main).encode()) == 0, lib.lp_last_error().decode()
    rng = np.random.default_rng(355)
    coefs = []
    while len(coefs) < 4 * K:
        # ray vs torus in the torus frame, as intersect_torus builds the quartic (:325-339)
        R, r = rng.uniform(0.3, 1.0), rng.uniform(0.05, 0.3)
        ro = rng.normal(size=3) * 3.0
        tgt = rng.normal(size=3) * (R + r) * 0.7
        rd = tgt - ro
        rd /= np.linalg.norm(rd)
        if len(coefs) % 7 == 0:
            rd[1] = 0.0; ro[1] = 0.0; rd /= np.linalg.norm(rd)   # in-plane rays
        m, n = ro @ ro, ro @ rd
        k = m + R * R - r * r
        A = 4 * n
        B = 2 * k + 4 * n * n - 4 * R * R * (rd[0] ** 2 + rd[2] ** 2)
        C = 4 * n * k - 8 * R * R * (ro[0] * rd[0] + ro[2] * rd[2])
        D = k * k - 4 * R * R * (ro[0] ** 2 + ro[2] ** 2)
        coefs.append([A, B, C, D])
    coefs = np.array(coefs, dtype=np.float32)
    # a few hand-made edge cases: biquadratic (q == 0), no real roots, repeated roots
    coefs[0] = [0, -5, 0, 4]       # (x^2-1)(x^2-4)
    coefs[1] = [0, 5, 0, 4]        # no real roots
    coefs[2] = [0, -2, 0, 1]       # (x^2-1)^2
    coefs[3] = [-10, 35, -50, 24]  # roots 1,2,3,4
    coefs[4] = [0, 0, 0, 0]
    coefs[5] = [0, 0, 0, -1]
    out = np.zeros((4 * K, 4), dtype=np.float32)
    for blk in range(4):
        for i in range(K):
            c = coefs[blk * K + i]
            lib.lp_uniform_f(f"u_coef[{i}]".encode(), 4, np.ascontiguousarray(c).ctypes.data)
        img = np.empty((1, K, 4), dtype=np.float32)
        assert lib.lp_draw(K, 1, 1, img.ctypes.data) == 0
        out[blk * K:(blk + 1) * K] = img[0]
    msg = f"quartic_kats: {4 * K} coefficient sets"
    if check_oracle:
        bad = 0
        for c, o in zip(coefs, out):
            r = orc.solve_quartic(*[float(v) for v in c])
            t = np.float32(1e20)
            for v in r:
                if v > np.float32(0.001) and v < t:
                    t = v
            mine = np.array([len(r), t, r[0] if len(r) > 0 else 0, r[1] if len(r) > 1 else 0], dtype=np.float32)
            if not np.array_equal(mine.view(np.uint32), o.view(np.uint32)):
                bad += 1
        msg += f" | oracle mismatches {bad}"
    print(msg, flush=True)
    np.savez_compressed(os.path.join(GOLDEN, "quartic_kats.npz"), coefs=coefs, n_t_r0_r1=out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--check-oracle", action="store_true")
    args = ap.parse_args()
    if not lp.available():
        sys.exit("needs /root/reference and oracle/_ref/liblp_harness.so")
    print(lp.gl_info())
    todo = cases()
    for name, spec in todo.items():
        if args.only and name not in args.only:
            continue
        run_case(name, spec, args.check_oracle)
    if not args.only or "spp" in args.only:
        spp_case("spp2_200", 2, args.check_oracle)
        spp_case("spp4_200", 4, args.check_oracle)
    if not args.only or "light" in args.only:
        light_case("light2_160", 2, 1.5, args.check_oracle)
        light_case("light3_160", 3, 0.6, args.check_oracle)
    if not args.only or "quartic_kats" in args.only:
        quartic_kats(args.check_oracle)


if __name__ == "__main__":
    main()
