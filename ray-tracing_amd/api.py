"""Python host over the C ABI of librt355.so (include/rt355.h).

`render(scene, camera, w, h)` stands where the reference's App.render stood
(main.py:244-303): it takes the same scene list (main.py:125-132) and the same camera
state (main.py:102-109), computes the camera matrix the way main.py:253-274 does and hands
everything to the HIP kernels.  There is no CPU fallback: if the library or a GPU is
missing the call raises.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from .camera import Camera

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT355_LIB") or os.path.join(_PKG_DIR, "librt355.so")

RT_MAX_OBJECTS = 1024
RT_MAX_PENDING = 16
RT_BAND_ROWS = 16

FLAG_FAST_MATH = 1 << 0
FLAG_NEWTON_POLISH = 1 << 1
FLAG_BOTTOM_UP = 1 << 2
FLAG_OUT_F32 = 1 << 3
FLAG_NO_ACCEL = 1 << 4
FLAG_OUT_RGB8 = 1 << 5
FLAG_PIXEL_ORDER = 1 << 6


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librt355 error {code}: {msg}")
        self.code = code


class RtObject(ctypes.Structure):
    """struct rt_object == struct SceneObject (fragment_shader.glsl:41-56)."""
    _fields_ = [("type", ctypes.c_int32), ("center", ctypes.c_float * 3), ("radius", ctypes.c_float),
                ("normal", ctypes.c_float * 3), ("major_radius", ctypes.c_float), ("minor_radius", ctypes.c_float),
                ("color", ctypes.c_float * 3), ("reflectivity", ctypes.c_float),
                ("transparency", ctypes.c_float), ("refractive_index", ctypes.c_float)]


class RtParams(ctypes.Structure):
    _fields_ = [("cam_to_world", ctypes.c_float * 16), ("focal_length", ctypes.c_float),
                ("light_pos", ctypes.c_float * 3), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("max_depth", ctypes.c_int32), ("stack_size", ctypes.c_int32), ("shadow_hops", ctypes.c_int32),
                ("spp_grid", ctypes.c_int32), ("flags", ctypes.c_uint32), ("part", ctypes.c_int32),
                ("nparts", ctypes.c_int32), ("light_grid", ctypes.c_int32), ("light_size", ctypes.c_float)]


class RtTiming(ctypes.Structure):
    _fields_ = [("kernel_ms", ctypes.c_float), ("total_ms", ctypes.c_float), ("vgprs", ctypes.c_int32),
                ("lds_bytes", ctypes.c_int32), ("pending_cap", ctypes.c_int32)]


assert ctypes.sizeof(RtObject) == 64

_lib = None
_inited_device = None

EXPORTS = ["rt_init", "rt_shutdown", "rt_last_error", "rt_set_scene", "rt_output_bytes", "rt_render_device",
           "rt_render", "rt_assemble_device", "rt_get_timing", "rt_count_events", "rt_event_count",
           "rt_event_flops", "rt_event_name", "rt_capture_segment", "rt_selftest_math", "rt_stage_scene_host",
           "rt_render_device_noalloc", "rt_host_alloc", "rt_host_free", "rt_init_multi", "rt_init_multi_devices", "rt_device_count",
           "rt_part_row_map", "rt_get_part_timing"]


def load_library():
    """dlopen librt355.so and declare its prototypes.  No GPU is touched."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(-4, f"{LIB_PATH} is not built (python -c 'import __graft_entry__ as g; g.build()')")
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    L.rt_init.argtypes = [ci]
    L.rt_shutdown.restype = None
    L.rt_last_error.restype = ctypes.c_char_p
    L.rt_set_scene.argtypes = [vp, ci]
    L.rt_output_bytes.restype = sz
    L.rt_output_bytes.argtypes = [vp]
    L.rt_render_device.argtypes = [vp, vp, sz, vp, ci]
    L.rt_render_device_noalloc.argtypes = [vp, vp, sz, vp]
    L.rt_render.argtypes = [vp, ci, vp, vp, sz]
    L.rt_init_multi.argtypes = [ci]
    L.rt_init_multi_devices.argtypes = [vp, ci]
    L.rt_part_row_map.argtypes = [ci, ci, ci, vp, ci]
    L.rt_get_part_timing.argtypes = [ci, vp, vp]
    L.rt_host_alloc.argtypes = [sz]
    L.rt_host_alloc.restype = vp
    L.rt_host_free.argtypes = [vp]
    L.rt_assemble_device.argtypes = [vp, vp, ci, ci, ci, ci, vp]
    L.rt_get_timing.argtypes = [vp]
    L.rt_count_events.argtypes = [vp, vp]
    L.rt_event_flops.argtypes = [ci]
    L.rt_event_name.argtypes = [ci]
    L.rt_event_name.restype = ctypes.c_char_p
    L.rt_capture_segment.argtypes = [vp, ci, vp, sz]
    L.rt_selftest_math.argtypes = [vp]
    L.rt_stage_scene_host.argtypes = [vp, ci, vp, sz, vp]
    _lib = L
    return L


def library_fingerprint():
    """What a measurement must be tied to: SHA-256 of the loaded librt355.so and of the kernel
    sources it is built from (device code, headers, host side, Makefile).  Counter summaries
    under profiles/ carry the same two values; a summary whose source hash differs from the
    running library's describes other kernels."""
    import hashlib
    src = hashlib.sha256()
    csrc = os.path.join(_PKG_DIR, "csrc")
    for name in ("Makefile", "rt_api.hip", "rt_device.hip", "rt_intersect.h", "rt_numerics.h", "rt_types.h"):
        with open(os.path.join(csrc, name), "rb") as f:
            src.update(name.encode() + b"\0" + f.read())
    with open(LIB_PATH, "rb") as f:
        lib = hashlib.sha256(f.read()).hexdigest()
    return {"lib_sha256": lib, "kernel_src_sha256": src.hexdigest()}


def _check(rc):
    if rc != 0:
        raise RtError(rc, _lib.rt_last_error().decode(errors="replace"))


def init(device: int = 0):
    """rt_init: bind this process to one GPU (one process per GPU)."""
    global _inited_device
    L = load_library()
    _check(L.rt_init(int(device)))
    _inited_device = int(device)
    return L


def init_multi(n_devices=None, devices=None):
    """One process, several GPUs: frames of render()/render_rgba() are split into interleaved
    16-row bands over `devices` (default: 0..n_devices-1) and put together on the first one."""
    global _inited_device
    L = load_library()
    if devices is None:
        _check(L.rt_init_multi(int(n_devices)))
        devices = list(range(int(n_devices)))
    else:
        arr = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        _check(L.rt_init_multi_devices(arr, len(devices)))
    _inited_device = int(devices[0])
    return L


def part_timing(part):
    """(kernel_ms, copy_ms) of part `part` of the last timed frame in multi-device mode."""
    L = _ensure_init()
    k, c = ctypes.c_float(0.0), ctypes.c_float(0.0)
    _check(L.rt_get_part_timing(int(part), ctypes.byref(k), ctypes.byref(c)))
    return float(k.value), float(c.value)


def part_row_map(height, part, nparts):
    L = load_library()
    n = L.rt_part_row_map(int(height), int(part), int(nparts), None, 0)
    if n < 0:
        raise RtError(n, L.rt_last_error().decode(errors="replace"))
    rows = np.zeros(max(n, 1), dtype=np.int32)
    L.rt_part_row_map(int(height), int(part), int(nparts), rows.ctypes.data, n)
    return rows[:n]


def _ensure_init():
    if _inited_device is None:
        init(int(os.environ.get("LOCAL_RANK", "0")))
    return _lib


def shutdown():
    global _inited_device
    if _lib is not None and _inited_device is not None:
        for _, ptr in _pinned:
            _lib.rt_host_free(ptr)
        _pinned.clear()
        _lib.rt_shutdown()
    _inited_device = None


_pinned = []      # (array, address) pairs handed out by pinned_empty


def pinned_empty(shape, dtype=np.uint8):
    """A numpy array in page-locked host memory (rt_host_alloc), for `render(..., out=array)`: the
    frame then comes back in one DMA transfer (at 3840x2160 the read-back, not the kernel, is most
    of a frame's time).  The memory belongs to the library: it is gone after pinned_release(array)
    or shutdown(), and the array must not be used afterwards."""
    L = _ensure_init()
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dt.itemsize
    ptr = L.rt_host_alloc(nbytes)
    if not ptr:
        raise RtError(-5, L.rt_last_error().decode(errors="replace"))
    raw = (ctypes.c_uint8 * max(nbytes, 1)).from_address(ptr)
    a = np.frombuffer(raw, dtype=dt, count=int(np.prod(shape))).reshape(shape)
    _pinned.append((a, ptr))
    return a


def is_pinned(a) -> bool:
    return any(b is a for b, _ in _pinned)


def pinned_release(a):
    """Give an array from pinned_empty back (do not touch it afterwards)."""
    for i, (b, ptr) in enumerate(_pinned):
        if b is a:
            _check(_lib.rt_host_free(ptr))
            del _pinned[i]
            return


_SCENE_KEYS = {"type", "center", "radius", "normal", "major_radius", "minor_radius", "color", "reflectivity",
               "transparency", "refractive_index"}


def pack_scene(scene):
    """Scene list (dicts as in main.py:125-132) -> ctypes array of rt_object.  As in the
    reference's upload loop (main.py:283-296) a sphere only sets radius and a torus only sets
    normal/major_radius/minor_radius; the other geometric fields stay 0."""
    if len(scene) > RT_MAX_OBJECTS:
        raise RtError(-2, f"{len(scene)} objects exceed the limit of {RT_MAX_OBJECTS}")
    arr = (RtObject * max(len(scene), 1))()
    for i, o in enumerate(scene):
        unknown = set(o) - _SCENE_KEYS
        if unknown:
            raise RtError(-2, f"object {i}: unknown keys {sorted(unknown)}")
        a = arr[i]
        try:
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
        except KeyError as e:
            raise RtError(-2, f"object {i}: missing key {e}") from None
    return arr


def make_params(camera, w, h, *, max_depth=4, stack_size=3, shadow_hops=2, spp_grid=1, flags=0,
                part=0, nparts=1, light_grid=1, light_size=0.0) -> RtParams:
    cam = Camera.coerce(camera)
    p = RtParams()
    p.cam_to_world[:] = cam.matrix().ravel().tolist()
    p.focal_length = float(cam.focal_length)
    p.light_pos[:] = [float(v) for v in cam.light_pos]
    p.width, p.height = int(w), int(h)
    p.max_depth, p.stack_size, p.shadow_hops, p.spp_grid = int(max_depth), int(stack_size), int(shadow_hops), int(spp_grid)
    p.flags = int(flags)
    p.part, p.nparts = int(part), int(nparts)
    p.light_grid, p.light_size = int(light_grid), float(light_size)
    return p


def _flags(fast_math, newton_polish, bottom_up, out_f32, no_accel=False, out_rgb8=False, pixel_order=False):
    return ((FLAG_FAST_MATH if fast_math else 0) | (FLAG_NEWTON_POLISH if newton_polish else 0) |
            (FLAG_BOTTOM_UP if bottom_up else 0) | (FLAG_OUT_F32 if out_f32 else 0) |
            (FLAG_NO_ACCEL if no_accel else 0) | (FLAG_OUT_RGB8 if out_rgb8 else 0) |
            (FLAG_PIXEL_ORDER if pixel_order else 0))


def set_scene(scene):
    L = _ensure_init()
    arr = pack_scene(scene)
    _check(L.rt_set_scene(ctypes.addressof(arr), len(scene)))


def _render_host(scene, camera, w, h, channels, *, max_depth=4, stack_size=3, shadow_hops=2, spp_grid=1,
                 fast_math=False, newton_polish=False, bottom_up=False, out_f32=False, part=0, nparts=1,
                 no_accel=False, light_grid=1, light_size=0.0, out=None, pixel_order=False):
    L = _ensure_init()
    arr = pack_scene(scene)
    p = make_params(camera, w, h, max_depth=max_depth, stack_size=stack_size, shadow_hops=shadow_hops,
                    spp_grid=spp_grid,
                    flags=_flags(fast_math, newton_polish, bottom_up, out_f32, no_accel, out_rgb8=channels == 3, pixel_order=pixel_order),
                    part=part, nparts=nparts, light_grid=light_grid, light_size=light_size)
    nbytes = L.rt_output_bytes(ctypes.addressof(p))
    bpp = 16 if out_f32 else channels
    rows = nbytes // (bpp * int(w)) if w > 0 else 0
    want = ((rows, int(w), channels), np.dtype(np.float32 if out_f32 else np.uint8))
    if out is None:
        out = np.empty(*want)
    elif out.shape != want[0] or out.dtype != want[1] or not out.flags.c_contiguous:
        raise RtError(-1, f"out must be a C-contiguous {want[1]} array of shape {want[0]}")
    _check(L.rt_render(ctypes.addressof(arr), len(scene), ctypes.addressof(p), out.ctypes.data, out.nbytes))
    return out


def render_rgba(scene, camera, w, h, **opts):
    """Render and return (rows, w, 4): uint8 RGBA as the reference's RGBA8 back buffer would
    hold it (main.py:303), or float32 unclamped colour with out_f32.  Row 0 is the TOP row
    unless bottom_up (GL read-back order).  With nparts > 1 only this part's bands.
    opts: max_depth, stack_size, shadow_hops, spp_grid, fast_math, newton_polish, bottom_up,
    out_f32, part, nparts, no_accel, light_grid, light_size (G x G samples of a square area light
    of that edge; 1 = the reference's point light), pixel_order (deep frames: force the pilot launch and
    the cost order of the pixels at any frame size; the frame is the same)."""
    return _render_host(scene, camera, w, h, 4, **opts)


def render(scene, camera, w, h, **opts) -> np.ndarray:
    """render(scene, camera, w, h) -> (h, w, 3) uint8 RGB, row 0 = top of the image.

    scene : list of dicts exactly as the reference's App.scene (main.py:125-132)
    camera: Camera or dict(eye, yaw, pitch, focal_length[, light_pos]) (main.py:102-109, :278)
    opts  : max_depth=4, stack_size=3, shadow_hops=2 (the constants of fragment_shader.glsl:653,
            :547, :453), spp_grid=1, fast_math=False, newton_polish=False,
            out=array from pinned_empty((h, w, 3)) to receive the frame (a frame loop should pass one:
            page-locked memory halves the time of a 4K frame)

    The kernel writes packed RGB8 (the reference's alpha is the constant 1.0), so the read-back
    is three bytes per pixel.
    """
    if opts.get("out_f32"):
        raise RtError(-2, "render returns uint8 RGB; use render_rgba(out_f32=True) for float colour")
    return _render_host(scene, camera, w, h, 3, **opts)


def timing() -> dict:
    L = _ensure_init()
    t = RtTiming()
    _check(L.rt_get_timing(ctypes.addressof(t)))
    return {"kernel_ms": t.kernel_ms, "total_ms": t.total_ms, "vgprs": t.vgprs, "lds_bytes": t.lds_bytes,
            "pending_cap": t.pending_cap}


def event_table():
    L = load_library()
    n = L.rt_event_count()
    return [L.rt_event_name(i).decode() for i in range(n)], np.array([L.rt_event_flops(i) for i in range(n)], dtype=np.int64)


def count_events(scene, camera, w, h, *, max_depth=4, stack_size=3, shadow_hops=2, spp_grid=1, fast_math=False,
                 light_grid=1, light_size=0.0):
    """Per-frame work counters (uint64 per event class) from the instrumented kernel flavour."""
    L = _ensure_init()
    set_scene(scene)
    p = make_params(camera, w, h, max_depth=max_depth, stack_size=stack_size, shadow_hops=shadow_hops,
                    spp_grid=spp_grid, flags=_flags(fast_math, False, False, False), light_grid=light_grid,
                    light_size=light_size)
    ev = np.zeros(L.rt_event_count(), dtype=np.uint64)
    _check(L.rt_count_events(ctypes.addressof(p), ev.ctypes.data))
    return ev


def capture_segment(scene, camera, w, h, seg, *, max_depth=4, stack_size=3, shadow_hops=2, fast_math=False,
                    bottom_up=True, newton_polish=False):
    """(h, w, 4) float32: object id, t, depth, segment count for segment `seg` of each path."""
    L = _ensure_init()
    set_scene(scene)
    p = make_params(camera, w, h, max_depth=max_depth, stack_size=stack_size, shadow_hops=shadow_hops,
                    flags=_flags(fast_math, newton_polish, bottom_up, False))
    out = np.empty((int(h), int(w), 4), dtype=np.float32)
    _check(L.rt_capture_segment(ctypes.addressof(p), int(seg), out.ctypes.data, out.nbytes))
    return out


def stage_scene_host(scene):
    """What rt_set_scene would stage for `scene`, computed on the host only (no GPU needed):
    returns (layout dict, blob float32 array, scene sphere (cx, cy, cz, R))."""
    L = load_library()
    objs = pack_scene(scene)
    layout = np.zeros(16, dtype=np.int32)
    _check(L.rt_stage_scene_host(ctypes.addressof(objs) if len(scene) else None, len(scene), None, 0, layout.ctypes.data))
    total = int(layout[5])
    blob = np.zeros(total + 4, dtype=np.float32)
    _check(L.rt_stage_scene_host(ctypes.addressof(objs) if len(scene) else None, len(scene), blob.ctypes.data, blob.size,
                                 layout.ctypes.data))
    keys = ["n", "ntori", "off_material", "off_torus", "off_kind", "total_dwords", "n1", "n2", "off_top", "off_leaf",
            "off_member", "off_slot_id", "leaf", "fan"]
    return dict(zip(keys, (int(v) for v in layout))), blob[:total], tuple(float(v) for v in blob[total:])


def selftest_math():
    """Mismatch counts (sqrt, 1/sqrt, reciprocal, log2 quotient, x/3, x/27) of the strict
    flavour's range-restricted arithmetic against the general forms, over all operands; all 0."""
    L = _ensure_init()
    counts = np.zeros(6, dtype=np.uint64)
    _check(L.rt_selftest_math(counts.ctypes.data))
    return counts


def frame_flops(events) -> int:
    """Algorithmic FLOPs of a frame: sum over event classes of count x per-event FLOPs."""
    _, table = event_table()
    return int((np.asarray(events, dtype=np.int64) * table).sum())
