"""rt_render_device_noalloc (queue counters and pending-ray slots from a static device array, no
allocation inside the call) against rt_render_device, through the C ABI: equal bytes, on two calls
in a row per kernel family, the second with a smaller frame, so that whatever the first call left
in the shared array is there when the second one starts."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(gpu_rt, scene, cam, w, h, flags=0, **o):
    """The same frame through both entry points -> two (h, w, 4) uint8 arrays."""
    import torch
    lib = gpu_rt.api.load_library()
    gpu_rt.set_scene(scene)
    p = gpu_rt.api.make_params(cam, w, h, flags=flags | gpu_rt.api.FLAG_BOTTOM_UP, **o)
    nbytes = lib.rt_output_bytes(ctypes.addressof(p))
    assert nbytes == w * h * 4
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.full((nbytes,), 0x5a, dtype=torch.uint8, device="cuda:0")
    b = torch.full((nbytes,), 0xa5, dtype=torch.uint8, device="cuda:0")
    assert lib.rt_render_device(ctypes.addressof(p), a.data_ptr(), nbytes, stream, 0) == 0, lib.rt_last_error()
    assert lib.rt_render_device_noalloc(ctypes.addressof(p), b.data_ptr(), nbytes, stream) == 0, lib.rt_last_error()
    torch.cuda.synchronize()
    return a.cpu().numpy().reshape(h, w, 4), b.cpu().numpy().reshape(h, w, 4)


FAMILIES = {
    # small scene, four pending rays possible: persistent whole-tile waves, queue counters + slot 3 in the array
    "whole_tiles_hybrid": (lambda rt: (rt.default_scene(), rt.default_camera()), dict(max_depth=8, stack_size=4, shadow_hops=2)),
    # large scene: sphere hierarchy, per-lane refill, every pending ray in the array
    "hierarchy": (lambda rt: (rt.mixed_scene(96), rt.mixed_camera()), dict(max_depth=5, stack_size=4, shadow_hops=2)),
    # the headline family uses neither counters nor workspace: the entry point must pass it through
    "one_pixel_per_lane": (lambda rt: (rt.default_scene(), rt.default_camera()), dict(max_depth=8, stack_size=3, shadow_hops=2)),
}


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_noalloc_equals_render_device_two_calls(gpu_rt, family):
    make, o = FAMILIES[family]
    scene, cam = make(gpu_rt)
    for (w, h) in ((648, 360), (203, 117)):         # second call: smaller, ragged frame
        a, b = _pair(gpu_rt, scene, cam, w, h, **o)
        assert a[..., 3].min() == 255                  # the reference entry point did write the frame
        assert np.array_equal(a, b), f"{family} {w}x{h}: {int(np.count_nonzero(np.any(a != b, axis=2)))} pixels differ"


def test_noalloc_float_output_and_parts(gpu_rt):
    """Float output and a band-split part go through the same path."""
    import torch
    lib = gpu_rt.api.load_library()
    scene, cam = gpu_rt.mixed_scene(64), gpu_rt.mixed_camera()
    gpu_rt.set_scene(scene)
    p = gpu_rt.api.make_params(cam, 320, 200, flags=gpu_rt.api.FLAG_OUT_F32, max_depth=4, stack_size=3, part=1, nparts=3)
    nbytes = lib.rt_output_bytes(ctypes.addressof(p))
    a = torch.zeros((nbytes // 4,), dtype=torch.float32, device="cuda:0")
    b = torch.ones((nbytes // 4,), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.rt_render_device(ctypes.addressof(p), a.data_ptr(), nbytes, stream, 0) == 0
    assert lib.rt_render_device_noalloc(ctypes.addressof(p), b.data_ptr(), nbytes, stream) == 0
    torch.cuda.synchronize()
    assert np.array_equal(a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32))


def test_noalloc_rejects_what_the_array_cannot_hold(gpu_rt):
    import torch
    lib = gpu_rt.api.load_library()
    gpu_rt.set_scene(gpu_rt.default_scene())
    p = gpu_rt.api.make_params(gpu_rt.default_camera(), 64, 64, max_depth=16, stack_size=16)
    buf = torch.zeros((64 * 64 * 4,), dtype=torch.uint8, device="cuda:0")
    rc = lib.rt_render_device_noalloc(ctypes.addressof(p), buf.data_ptr(), buf.numel(), None)
    assert rc == -3 and b"pending-ray slots" in lib.rt_last_error()
    # and too small an output buffer, as the allocating entry point does
    p = gpu_rt.api.make_params(gpu_rt.default_camera(), 64, 64)
    assert lib.rt_render_device_noalloc(ctypes.addressof(p), buf.data_ptr(), 100, None) == -6
