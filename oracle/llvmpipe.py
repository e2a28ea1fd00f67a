"""Run the reference renderer's own GLSL on Mesa llvmpipe (CPU), headless.

TEST INFRASTRUCTURE ONLY -- part of the oracle.  Nothing in the shipped renderer imports
this module.  It only works where the reference checkout is mounted (the shader text is
read from there at run time and is never written into this repository) and where
oracle/_ref/liblp_harness.so has been built (oracle/Makefile).

It restates the reference's host side for one frame:
  * camera basis + float32 camera_to_world        -> main.py:253-274
  * uniform upload (u_resolution, u_light_pos, u_camera_to_world with transpose=GL_TRUE,
    u_focal_length, scene[i].*)                    -> main.py:277-296
  * one draw of the full-screen quad               -> main.py:300
The window back buffer is replaced by an off-screen RGBA8 (or RGBA32F) target that is
read back bottom-up.

Shader constants that the reference hard-codes (NUM_OBJECTS fragment_shader.glsl:68,
max_depth :653, STACK_SIZE :547, shadow hop bound :453) can be substituted textually in
memory, and an instrumentation patch can be applied that makes the shader output the
object id / t / depth of the k-th traced segment instead of the colour.
"""
from __future__ import annotations

import ctypes
import os
import re
import time

import numpy as np

REFERENCE_DIR = os.environ.get("RT_REFERENCE_DIR", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "liblp_harness.so")

DEFAULT_LIGHT = (-2.0, 4.0, 1.0)  # main.py:278


def available() -> bool:
    return os.path.exists(_LIB_PATH) and os.path.exists(os.path.join(REFERENCE_DIR, "fragment_shader.glsl"))


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    lib = ctypes.CDLL(_LIB_PATH)
    lib.lp_last_error.restype = ctypes.c_char_p
    lib.lp_gl_string.restype = ctypes.c_char_p
    lib.lp_gl_string.argtypes = [ctypes.c_int]
    lib.lp_program.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lib.lp_uniform_i.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.lp_uniform_f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]
    lib.lp_uniform_m4.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]
    lib.lp_draw.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    if lib.lp_open() != 0:
        raise RuntimeError("lp_open: " + lib.lp_last_error().decode())
    _lib = lib
    return lib


def gl_info() -> dict:
    lib = _load()
    return {k: lib.lp_gl_string(i).decode() for i, k in enumerate(["renderer", "version", "glsl", "vendor"])}


def read_reference_shader(name: str) -> str:
    with open(os.path.join(REFERENCE_DIR, name), "r", encoding="utf-8") as f:
        return f.read()


def _sub_once(text: str, pattern: str, repl: str) -> str:
    out, n = re.subn(pattern, repl, text, count=1)
    if n != 1:
        raise RuntimeError(f"shader patch did not apply: {pattern!r}")
    return out


def patch_constants(fs: str, num_objects=None, max_depth=None, stack_size=None, shadow_hops=None) -> str:
    """Substitute the reference's compile-time constants, in memory only."""
    if num_objects is not None:
        fs = _sub_once(fs, r"const int NUM_OBJECTS = \d+;", f"const int NUM_OBJECTS = {int(num_objects)};")
    if max_depth is not None:
        fs = _sub_once(fs, r"const int max_depth = \d+;", f"const int max_depth = {int(max_depth)};")
    if stack_size is not None:
        fs = _sub_once(fs, r"const int STACK_SIZE = \d+;", f"const int STACK_SIZE = {int(stack_size)};")
    if shadow_hops is not None:
        fs = _sub_once(fs, r"for\(int i = 0; i < 2; i\+\+\) \{", f"for(int i = 0; i < {int(shadow_hops)}; i++) {{")
    return fs


def patch_capture_segment(fs: str) -> str:
    """Instrument the shader so that it writes, for the u_capture_seg-th ray popped from the
    stack (visit order; 0 = primary), vec4(object id, t, depth, number of segments traced).
    id: -1 miss (sky), 0..NUM_OBJECTS-1 scene object, NUM_OBJECTS ground plane; -2 if the
    pixel's path has fewer segments than u_capture_seg+1."""
    fs = _sub_once(fs, r"(uniform float u_focal_length;[^\n]*\n)",
                   r"\1uniform int u_capture_seg;\nint g_last_id; int g_seg; vec4 g_cap;\n")
    fs = _sub_once(fs, r"(closest_hit\.t = 1e30;[^\n]*\n)", r"\1    g_last_id = -1;\n")
    fs = _sub_once(fs, r"(closest_hit = current_hit;)", r"\1 g_last_id = i;")
    fs = _sub_once(fs, r"(closest_hit = plane_hit;)", r"\1 g_last_id = NUM_OBJECTS;")
    fs = _sub_once(fs, r"(HitInfo hit = trace\(current_state\.ray\);)",
                   r"\1\n        if (g_seg == u_capture_seg) g_cap = vec4(float(g_last_id), hit.hit ? hit.t : 0.0, "
                   r"float(current_state.depth), 0.0);\n        g_seg++;")
    fs = _sub_once(fs, r"(vec3 final_color = RayTraceIterative\()", r"g_seg = 0; g_cap = vec4(-2.0, 0.0, 0.0, 0.0);\n    \1")
    fs = _sub_once(fs, r"FragColor = vec4\(final_color, 1\.0\);", "FragColor = vec4(g_cap.xyz, float(g_seg));")
    return fs


def patch_pixel_offset(fs: str) -> str:
    """Sub-pixel sampling: shift gl_FragCoord.xy by u_subpixel, the sample's offset from the
    pixel centre; (0,0) is the unmodified pixel-centre sample (fragment_shader.glsl:652)."""
    fs = _sub_once(fs, r"(uniform float u_focal_length;[^\n]*\n)", r"\1uniform vec2 u_subpixel;\n")
    fs = _sub_once(fs, r"\(gl_FragCoord\.xy \* 2\.0 - u_resolution\.xy\)",
                   "((gl_FragCoord.xy + u_subpixel) * 2.0 - u_resolution.xy)")
    return fs


def camera_to_world(eye, yaw, pitch, up=(0.0, 1.0, 0.0)) -> np.ndarray:
    """float32 4x4 with rows [x|y|view|eye], built the way main.py:255-274 does:
    float64 trig stored into a float32 vector, float32 normalise / cross."""
    def normalize(v):
        n = np.linalg.norm(v)
        return v if n == 0 else v / n
    d = np.empty(3, dtype=np.float32)
    d[0] = np.cos(np.radians(yaw)) * np.cos(np.radians(pitch))
    d[1] = np.sin(np.radians(pitch))
    d[2] = np.sin(np.radians(yaw)) * np.cos(np.radians(pitch))
    view = normalize(d)
    upv = np.array(up, dtype=np.float32)
    z = -view
    x = normalize(np.cross(upv, z))
    y = normalize(np.cross(z, x))
    e = np.array(eye, dtype=np.float32)
    return np.array([[x[0], y[0], view[0], e[0]],
                     [x[1], y[1], view[1], e[1]],
                     [x[2], y[2], view[2], e[2]],
                     [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)


class Program:
    """One compiled+linked variant of the reference shader."""

    def __init__(self, num_objects=5, max_depth=4, stack_size=3, shadow_hops=2,
                 capture=False, jitter=False):
        self.lib = _load()
        fs = read_reference_shader("fragment_shader.glsl")
        vs = read_reference_shader("vertex_shader.glsl")
        as_shipped = (num_objects, max_depth, stack_size, shadow_hops) == (5, 4, 3, 2)
        if not as_shipped:
            fs = patch_constants(fs, num_objects, max_depth, stack_size, shadow_hops)
        if capture:
            fs = patch_capture_segment(fs)
        if jitter:
            fs = patch_pixel_offset(fs)
        self.num_objects = num_objects
        self.capture = capture
        self.jitter = jitter
        t0 = time.perf_counter()
        if self.lib.lp_program(vs.encode(), fs.encode()) != 0:
            raise RuntimeError(self.lib.lp_last_error().decode())
        self.compile_s = time.perf_counter() - t0

    def _f(self, name, vals):
        a = np.asarray(vals, dtype=np.float32).ravel()
        return self.lib.lp_uniform_f(name.encode(), a.size, a.ctypes.data)

    def set_scene(self, scene):
        if len(scene) != self.num_objects:
            raise ValueError("scene length must equal the NUM_OBJECTS this program was built with")
        for i, obj in enumerate(scene):  # main.py:283-296
            p = f"scene[{i}]."
            self.lib.lp_uniform_i((p + "type").encode(), int(obj["type"]))
            self._f(p + "center", obj["center"])
            self._f(p + "color", obj["color"])
            self._f(p + "reflectivity", [obj["reflectivity"]])
            self._f(p + "transparency", [obj["transparency"]])
            self._f(p + "refractive_index", [obj["refractive_index"]])
            if obj["type"] == 1:
                self._f(p + "radius", [obj["radius"]])
                # a previous scene may have left torus fields behind in this program object
                self._f(p + "normal", [0, 0, 0]); self._f(p + "major_radius", [0]); self._f(p + "minor_radius", [0])
            elif obj["type"] == 2:
                self._f(p + "radius", [0])
                self._f(p + "normal", obj["normal"])
                self._f(p + "major_radius", [obj["major_radius"]])
                self._f(p + "minor_radius", [obj["minor_radius"]])

    def set_camera(self, cam_to_world, focal_length, w, h, light=DEFAULT_LIGHT):
        self._f("u_resolution", [float(w), float(h)])            # main.py:277
        self._f("u_light_pos", light)                            # main.py:278
        m = np.ascontiguousarray(cam_to_world, dtype=np.float32)
        self.lib.lp_uniform_m4(b"u_camera_to_world", 1, m.ctypes.data)  # main.py:279 (GL_TRUE)
        self._f("u_focal_length", [focal_length])                # main.py:280

    def draw_rgba8(self, w, h) -> np.ndarray:
        """(h, w, 4) uint8, row 0 = bottom row (GL order)."""
        out = np.empty((h, w, 4), dtype=np.uint8)
        if self.lib.lp_draw(w, h, 0, out.ctypes.data) != 0:
            raise RuntimeError(self.lib.lp_last_error().decode())
        return out

    def draw_rgba32f(self, w, h) -> np.ndarray:
        out = np.empty((h, w, 4), dtype=np.float32)
        if self.lib.lp_draw(w, h, 1, out.ctypes.data) != 0:
            raise RuntimeError(self.lib.lp_last_error().decode())
        return out

    def set_capture_segment(self, k):
        self.lib.lp_uniform_i(b"u_capture_seg", int(k))

    def set_subpixel(self, ox, oy):
        self._f("u_subpixel", [ox, oy])


def render_reference(scene, camera, w, h, max_depth=4, stack_size=3, shadow_hops=2,
                     float_target=False):
    """Convenience: one frame of the reference shader.  camera = dict(eye, yaw, pitch,
    focal_length[, light_pos]).  Returns the bottom-up RGBA image."""
    prog = Program(len(scene), max_depth, stack_size, shadow_hops)
    prog.set_scene(scene)
    m = camera_to_world(camera["eye"], camera["yaw"], camera["pitch"])
    prog.set_camera(m, camera["focal_length"], w, h, camera.get("light_pos", DEFAULT_LIGHT))
    return prog.draw_rgba32f(w, h) if float_target else prog.draw_rgba8(w, h)
