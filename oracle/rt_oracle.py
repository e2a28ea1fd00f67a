"""ctypes binding of oracle/librt_oracle.so (the CPU restatement of the reference shader).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg as the checker / baseline.  The renderer package never imports it.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librt_oracle.so")


class OrcObject(ctypes.Structure):
    """struct SceneObject, fragment_shader.glsl:41-56."""
    _fields_ = [("type", ctypes.c_int32), ("center", ctypes.c_float * 3), ("radius", ctypes.c_float),
                ("normal", ctypes.c_float * 3), ("major_radius", ctypes.c_float), ("minor_radius", ctypes.c_float),
                ("color", ctypes.c_float * 3), ("reflectivity", ctypes.c_float),
                ("transparency", ctypes.c_float), ("refractive_index", ctypes.c_float)]


class OrcParams(ctypes.Structure):
    _fields_ = [("cam_to_world", ctypes.c_float * 16), ("focal", ctypes.c_float), ("light", ctypes.c_float * 3),
                ("w", ctypes.c_int32), ("h", ctypes.c_int32), ("max_depth", ctypes.c_int32),
                ("stack_size", ctypes.c_int32), ("shadow_hops", ctypes.c_int32), ("spp_grid", ctypes.c_int32),
                ("light_grid", ctypes.c_int32), ("light_size", ctypes.c_float), ("flags", ctypes.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is not built; run `make -C oracle`")
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_pow.restype = ctypes.c_float
        L.orc_pow.argtypes = [ctypes.c_float, ctypes.c_float]
        L.orc_cos.restype = ctypes.c_float
        L.orc_cos.argtypes = [ctypes.c_float]
        L.orc_acos.restype = ctypes.c_float
        L.orc_acos.argtypes = [ctypes.c_float]
        L.orc_solve_quartic.argtypes = [ctypes.c_float] * 4 + [ctypes.c_void_p]
        L.orc_render.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.orc_render_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.orc_capture.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                  ctypes.c_void_p, ctypes.c_int]
        _lib = L
    return _lib


def pack_scene(scene):
    """List of main.py-style dicts (main.py:125-132) -> array of OrcObject.  Fields a dict
    does not carry stay 0, as the reference leaves those uniforms unset (main.py:291-296)."""
    arr = (OrcObject * max(len(scene), 1))()
    for i, o in enumerate(scene):
        a = arr[i]
        a.type = int(o["type"])
        a.center[:] = [float(v) for v in o["center"]]
        a.color[:] = [float(v) for v in o["color"]]
        a.reflectivity = float(o["reflectivity"])
        a.transparency = float(o["transparency"])
        a.refractive_index = float(o["refractive_index"])
        if a.type == 1:
            a.radius = float(o["radius"])
        elif a.type == 2:
            a.normal[:] = [float(v) for v in o["normal"]]
            a.major_radius = float(o["major_radius"])
            a.minor_radius = float(o["minor_radius"])
    return arr


def make_params(cam_to_world, focal, w, h, light=(-2.0, 4.0, 1.0), max_depth=4, stack_size=3,
                shadow_hops=2, spp_grid=1, light_grid=1, light_size=0.0, newton_polish=False):
    p = OrcParams()
    p.flags = 1 if newton_polish else 0
    m = np.ascontiguousarray(cam_to_world, dtype=np.float32).ravel()
    p.cam_to_world[:] = m.tolist()
    p.focal = float(focal)
    p.light[:] = [float(v) for v in light]
    p.w, p.h = int(w), int(h)
    p.max_depth, p.stack_size, p.shadow_hops, p.spp_grid = int(max_depth), int(stack_size), int(shadow_hops), int(spp_grid)
    p.light_grid, p.light_size = int(light_grid), float(light_size)
    return p


def event_table():
    L = lib()
    n = L.orc_event_count()
    return np.array([L.orc_event_flops(i) for i in range(n)], dtype=np.int64)


def render(scene, cam_to_world, focal, w, h, *, light=(-2.0, 4.0, 1.0), max_depth=4, stack_size=3,
           shadow_hops=2, spp_grid=1, light_grid=1, light_size=0.0, want_f32=False, want_events=False, rows=None,
           nthreads=None, newton_polish=False):
    """Returns dict(rgba8=(h,w,4) uint8 bottom-up [, f32=(h,w,4) float32][, events=uint64[EV_COUNT]])."""
    L = lib()
    objs = pack_scene(scene)
    p = make_params(cam_to_world, focal, w, h, light, max_depth, stack_size, shadow_hops, spp_grid, light_grid, light_size,
                    newton_polish)
    y0, y1 = (0, h) if rows is None else rows
    nrows = y1 - y0
    u8 = np.empty((nrows, w, 4), dtype=np.uint8)
    f32 = np.empty((nrows, w, 4), dtype=np.float32) if want_f32 else None
    ev = np.zeros(L.orc_event_count(), dtype=np.uint64) if want_events else None
    if nthreads is None:
        nthreads = min(os.cpu_count() or 1, 16)
    rc = L.orc_render_rows(ctypes.addressof(objs), len(scene), ctypes.addressof(p), y0, y1,
                           f32.ctypes.data if want_f32 else None, u8.ctypes.data,
                           ev.ctypes.data if want_events else None, nthreads)
    if rc != 0:
        raise ValueError(f"orc_render_rows failed with code {rc}")
    out = {"rgba8": u8}
    if want_f32:
        out["f32"] = f32
    if want_events:
        out["events"] = ev
    return out


def capture(scene, cam_to_world, focal, w, h, seg, *, light=(-2.0, 4.0, 1.0), max_depth=4, stack_size=3,
            shadow_hops=2, nthreads=None, newton_polish=False):
    """(h,w,4) float32: id, t, depth, segment count for segment `seg` of each pixel's path."""
    L = lib()
    objs = pack_scene(scene)
    p = make_params(cam_to_world, focal, w, h, light, max_depth, stack_size, shadow_hops, 1, newton_polish=newton_polish)
    out = np.empty((h, w, 4), dtype=np.float32)
    if nthreads is None:
        nthreads = min(os.cpu_count() or 1, 16)
    rc = L.orc_capture(ctypes.addressof(objs), len(scene), ctypes.addressof(p), int(seg), out.ctypes.data, nthreads)
    if rc != 0:
        raise ValueError(f"orc_capture failed with code {rc}")
    return out


def solve_quartic(a, b, c, d):
    roots = np.zeros(4, dtype=np.float32)
    n = lib().orc_solve_quartic(a, b, c, d, roots.ctypes.data)
    return roots[:n].copy()
