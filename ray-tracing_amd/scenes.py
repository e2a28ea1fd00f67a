"""Scene lists in the reference's own format, and the synthetic stress scenes.

A scene is the list of dicts the reference keeps in App.scene (main.py:125-132): every
object has type (1 sphere, 2 torus), center, color, reflectivity, transparency,
refractive_index; spheres add radius; tori add normal, major_radius, minor_radius.
`default_scene` / `default_camera` are the reference's shipped values.  The other
generators are seeded synthetic workloads of the same format (the reference ships only
the one scene).
"""
from __future__ import annotations

import numpy as np

from .camera import Camera

SHAPE_SPHERE = 1  # fragment_shader.glsl:19
SHAPE_TORUS = 2   # fragment_shader.glsl:20


def default_scene():
    """main.py:125-132, verbatim values."""
    return [
        {"type": 1, "center": [0.0, 0.0, -0.6], "radius": 1.0, "color": [1.0, 1.0, 1.0],
         "reflectivity": 0.1, "transparency": 0.9, "refractive_index": 1.5},
        {"type": 1, "center": [-0.5, -0.5, -3.0], "radius": 0.5, "color": [0.2, 1.0, 0.2],
         "reflectivity": 0.05, "transparency": 0.0, "refractive_index": 1.5},
        {"type": 1, "center": [0.5, -0.5, -3.0], "radius": 0.5, "color": [0.2, 0.2, 1.0],
         "reflectivity": 0.05, "transparency": 0.0, "refractive_index": 1.5},
        {"type": 1, "center": [0.0, 0.366, -3.0], "radius": 0.5, "color": [1.0, 0.2, 0.2],
         "reflectivity": 0.05, "transparency": 0.0, "refractive_index": 1.5},
        {"type": 2, "center": [0.0, 1.2, -3.0], "normal": [0.0, 1.0, 0.0], "major_radius": 0.8,
         "minor_radius": 0.2, "color": [1.0, 0.8, 0.2], "reflectivity": 0.4, "transparency": 0.0,
         "refractive_index": 1.0},
    ]


def default_camera() -> Camera:
    """main.py:103-109."""
    return Camera(eye=(4.0, 1.2, 3.0), yaw=-135.0, pitch=-12.0, focal_length=2.0)


def _f32(x):
    """Round through float32 so every consumer (ctypes, numpy, GL uniforms) sees one value."""
    return float(np.float32(x))


def _vec(v):
    return [_f32(c) for c in v]


def _sphere(center, radius, color, refl, transp, ior):
    return {"type": SHAPE_SPHERE, "center": _vec(center), "radius": _f32(radius), "color": _vec(color),
            "reflectivity": _f32(refl), "transparency": _f32(transp), "refractive_index": _f32(ior)}


def _torus(center, normal, major, minor, color, refl, transp, ior):
    return {"type": SHAPE_TORUS, "center": _vec(center), "normal": _vec(normal), "major_radius": _f32(major),
            "minor_radius": _f32(minor), "color": _vec(color), "reflectivity": _f32(refl),
            "transparency": _f32(transp), "refractive_index": _f32(ior)}


def tori_stress_scene(n: int = 64, seed: int = 355):
    """Quartic stress: n opaque, reflective tori with random orientations in a long slab."""
    rng = np.random.default_rng(seed)
    scene = []
    for _ in range(n):
        center = [rng.uniform(-4.0, 4.0), rng.uniform(-0.6, 1.6), rng.uniform(-14.0, 2.0)]
        normal = rng.normal(size=3)
        normal /= np.linalg.norm(normal)
        scene.append(_torus(center, normal, rng.uniform(0.3, 0.6), rng.uniform(0.05, 0.15),
                            rng.uniform(0.2, 1.0, size=3), 0.3, 0.0, 1.0))
    return scene


def tori_stress_camera() -> Camera:
    """Low camera looking along the slab so that most rays graze the tori."""
    return Camera(eye=(0.0, 0.2, 6.0), yaw=-90.0, pitch=-3.0, focal_length=2.0)


def spheres_deep_scene(n: int = 256, seed: int = 355):
    """Deep refraction: half glass spheres (transparency 0.6-0.9, ior 1.5), half mirrors
    (reflectivity 0.3-0.8)."""
    rng = np.random.default_rng(seed)
    scene = []
    for i in range(n):
        center = [rng.uniform(-5.0, 5.0), rng.uniform(-0.6, 3.0), rng.uniform(-12.0, -1.0)]
        radius = rng.uniform(0.2, 0.5)
        color = rng.uniform(0.2, 1.0, size=3)
        if i % 2 == 0:
            scene.append(_sphere(center, radius, color, 0.05, rng.uniform(0.6, 0.9), 1.5))
        else:
            scene.append(_sphere(center, radius, color, rng.uniform(0.3, 0.8), 0.0, 1.5))
    return scene


def spheres_deep_camera() -> Camera:
    return Camera(eye=(0.0, 1.5, 4.0), yaw=-90.0, pitch=-8.0, focal_length=2.0)


def mixed_scene(n: int = 1024, seed: int = 355):
    """Mixed objects: every fourth a torus, the rest spheres; a third of the spheres glass,
    a third mirrors, a third diffuse."""
    rng = np.random.default_rng(seed)
    scene = []
    for i in range(n):
        center = [rng.uniform(-10.0, 10.0), rng.uniform(-0.7, 5.0), rng.uniform(-30.0, -1.0)]
        color = rng.uniform(0.2, 1.0, size=3)
        if i % 4 == 3:
            normal = rng.normal(size=3)
            normal /= np.linalg.norm(normal)
            scene.append(_torus(center, normal, rng.uniform(0.25, 0.45), rng.uniform(0.05, 0.12),
                                color, 0.3, 0.0, 1.0))
        else:
            radius = rng.uniform(0.15, 0.4)
            kind = i % 3
            if kind == 0:
                scene.append(_sphere(center, radius, color, 0.05, rng.uniform(0.6, 0.9), 1.5))
            elif kind == 1:
                scene.append(_sphere(center, radius, color, rng.uniform(0.3, 0.8), 0.0, 1.5))
            else:
                scene.append(_sphere(center, radius, color, 0.05, 0.0, 1.5))
    return scene


def mixed_camera() -> Camera:
    return Camera(eye=(0.0, 2.5, 5.0), yaw=-90.0, pitch=-10.0, focal_length=2.0)


def random_scene(seed: int):
    """Small random mixed scene + camera + constants, deliberately nasty: overlapping glass
    objects (rays enter one medium from another), transparent tori, reflectivity 0 and 1,
    tilted tori, objects cutting the floor.  Returns (scene, camera, opts)."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 14))
    scene = []
    for _ in range(n):
        c = [rng.uniform(-2.5, 2.5), rng.uniform(-1.2, 2.0), rng.uniform(-6.0, -1.0)]
        col = rng.uniform(0.0, 1.0, size=3)
        kind = rng.integers(0, 4)
        refl = float(rng.choice([0.0, 0.05, 0.3, 0.8, 1.0]))
        transp = float(rng.choice([0.0, 0.0, 0.5, 0.9]))
        if refl + transp > 1.0:
            transp = round(1.0 - refl, 3)
        ior = float(rng.choice([1.0, 1.33, 1.5, 2.4]))
        if kind == 0:
            scene.append(_torus(c, rng.normal(size=3), rng.uniform(0.3, 0.9), rng.uniform(0.05, 0.3), col, refl, transp, ior))
        else:
            scene.append(_sphere(c, rng.uniform(0.2, 1.1), col, refl, transp, ior))
    cam = Camera(eye=(_f32(rng.uniform(-3, 3)), _f32(rng.uniform(-0.5, 3)), _f32(rng.uniform(0, 4))),
                 yaw=_f32(rng.uniform(-120, -60)), pitch=_f32(rng.uniform(-35, 15)),
                 focal_length=_f32(rng.uniform(0.8, 3.0)),
                 light_pos=(_f32(rng.uniform(-4, 4)), _f32(rng.uniform(1, 6)), _f32(rng.uniform(-3, 3))))
    opts = dict(max_depth=int(rng.integers(1, 10)), stack_size=int(rng.integers(1, 9)), shadow_hops=int(rng.integers(0, 5)))
    return scene, cam, opts


def cluster_scene(seed: int):
    """3-39 objects in one compact cluster, placed and viewed so that many rays pass close to the
    sphere around the whole cluster: the cluster may sit far from the origin (coordinates where
    binary32 spacing is coarse), the camera far outside it, just outside it, or inside it, with a
    wide or narrow field of view.  Returns (scene, camera, opts)."""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(3, 40))
    offset = np.array([0.0, 0.0, -4.0]) if seed % 4 else rng.uniform(-1, 1, size=3) * float(rng.choice([50.0, 600.0, 5000.0]))
    spread = float(rng.choice([0.3, 1.0, 3.0]))
    scene = []
    for _ in range(n):
        c = offset + rng.normal(size=3) * spread
        col = rng.uniform(0.1, 1.0, size=3)
        refl = float(rng.choice([0.0, 0.3, 0.9]))
        transp = float(rng.choice([0.0, 0.0, 0.6]))
        if refl + transp > 1.0:
            transp = 0.0
        if rng.uniform() < 0.3:
            scene.append(_torus(c, rng.normal(size=3), rng.uniform(0.1, 0.8) * spread, rng.uniform(0.02, 0.2) * spread, col, refl, transp, 1.5))
        else:
            scene.append(_sphere(c, rng.uniform(0.05, 0.7) * spread, col, refl, transp, 1.5))
    dist = float([0.0, 1.2, 3.0, 10.0][seed % 4]) * (3.0 * spread)
    direction = rng.normal(size=3)
    direction /= np.linalg.norm(direction)
    eye = offset + direction * dist + rng.normal(size=3) * 0.05
    # look at the cluster, or a little past its edge
    aim = offset + rng.normal(size=3) * spread * float(rng.choice([0.0, 0.7, 1.5])) - eye
    yaw = float(np.degrees(np.arctan2(aim[2], aim[0]))) if np.hypot(aim[0], aim[2]) > 1e-6 else -90.0
    pitch = float(np.clip(np.degrees(np.arctan2(aim[1], np.hypot(aim[0], aim[2]))), -89.0, 89.0))
    cam = Camera(eye=tuple(_f32(v) for v in eye), yaw=_f32(yaw), pitch=_f32(pitch),
                 focal_length=_f32(rng.choice([0.5, 2.0, 8.0]) if seed % 4 < 3 else rng.choice([2.0, 8.0])),
                 light_pos=tuple(_f32(v) for v in (offset + rng.normal(size=3) * 4.0 * spread + np.array([0.0, 3.0 * spread, 0.0]))))
    opts = dict(max_depth=int(rng.integers(2, 7)), stack_size=int(rng.integers(1, 5)), shadow_hops=int(rng.integers(0, 3)))
    return scene, cam, opts


def random_large_scene(seed: int):
    """Large random scene (100-700 objects) built to stress a spatial hierarchy: clusters of very
    different density, a few huge spheres that contain many others, objects far from the rest,
    duplicated centres (equal hit distances decided by index), tori of all sizes; the camera is
    inside the cloud, far outside it, or looking along a row.  Returns (scene, camera, opts)."""
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.integers(100, 700))
    scene = []
    ncl = int(rng.integers(1, 6))
    centres = rng.uniform([-8, 0, -25], [8, 6, -2], size=(ncl, 3))
    spread = rng.uniform(0.3, 6.0, size=ncl)
    while len(scene) < n:
        k = int(rng.integers(0, ncl))
        c = centres[k] + rng.normal(size=3) * spread[k]
        u = rng.uniform()
        col = rng.uniform(0.1, 1.0, size=3)
        refl = float(rng.choice([0.0, 0.05, 0.5, 0.9]))
        transp = float(rng.choice([0.0, 0.0, 0.0, 0.7]))
        if refl + transp > 1.0:
            transp = 0.0
        ior = float(rng.choice([1.0, 1.33, 1.5]))
        if u < 0.02:
            scene.append(_sphere(c, rng.uniform(3.0, 9.0), col, 0.05, 0.85, 1.1))       # huge glass shell
        elif u < 0.05:
            scene.append(_sphere(c + rng.normal(size=3) * 60.0, rng.uniform(0.5, 4.0), col, refl, 0.0, ior))  # outlier
        elif u < 0.10 and scene:
            twin = dict(scene[int(rng.integers(0, len(scene)))])                        # exact duplicate
            scene.append(twin)
        elif u < 0.35:
            scene.append(_torus(c, rng.normal(size=3), rng.uniform(0.1, 1.5), rng.uniform(0.02, 0.4), col, refl, transp, ior))
        else:
            scene.append(_sphere(c, rng.uniform(0.05, 0.9), col, refl, transp, ior))
    mode = seed % 3
    if mode == 0:       # inside the cloud
        eye = centres[0] + rng.normal(size=3) * 0.5
        cam = Camera(eye=_vec(eye), yaw=_f32(rng.uniform(-180, 180)), pitch=_f32(rng.uniform(-30, 30)),
                     focal_length=_f32(rng.uniform(0.7, 2.0)))
    elif mode == 1:     # far away, long lens
        cam = Camera(eye=(_f32(rng.uniform(-20, 20)), _f32(rng.uniform(5, 40)), _f32(rng.uniform(150, 400))),
                     yaw=-90.0, pitch=_f32(rng.uniform(-8, -2)), focal_length=_f32(rng.uniform(8.0, 20.0)))
    else:
        cam = Camera(eye=(_f32(rng.uniform(-3, 3)), _f32(rng.uniform(0.5, 4)), _f32(rng.uniform(2, 8))),
                     yaw=_f32(rng.uniform(-110, -70)), pitch=_f32(rng.uniform(-20, 5)),
                     focal_length=_f32(rng.uniform(1.0, 3.0)))
    opts = dict(max_depth=int(rng.integers(2, 9)), stack_size=int(rng.integers(1, 9)), shadow_hops=int(rng.integers(0, 4)))
    return scene, cam, opts


def sphere_row_scene(n: int = 120):
    """n equal glass spheres in a row along -z: a ray down the row passes every bounding sphere."""
    return [_sphere([0.0, 0.5, -2.0 - 1.5 * i], 0.6, [0.9, 0.9, 1.0], 0.05, 0.9, 1.02) for i in range(n)]
