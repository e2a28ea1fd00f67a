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


def test_noalloc_two_streams_share_the_scratch_in_order_and_allocate_nothing(gpu_rt):
    """Two calls of a persistent-wave family back to back on two different streams, different frame
    sizes, two output buffers: the library's event orders them on the shared array.  Free device
    memory is the same before and after the calls."""
    import torch
    lib = gpu_rt.api.load_library()
    scene, cam = gpu_rt.mixed_scene(96), gpu_rt.mixed_camera()
    gpu_rt.set_scene(scene)
    o = dict(max_depth=5, stack_size=4, shadow_hops=2)
    sizes = [(648, 360), (203, 117)]
    params = [gpu_rt.api.make_params(cam, w, h, flags=gpu_rt.api.FLAG_BOTTOM_UP, **o) for w, h in sizes]
    want, got = [], []
    for (w, h), p in zip(sizes, params):
        buf = torch.zeros((w * h * 4,), dtype=torch.uint8, device="cuda:0")
        assert lib.rt_render_device(ctypes.addressof(p), buf.data_ptr(), buf.numel(), None, 0) == 0, lib.rt_last_error()
        torch.cuda.synchronize()
        want.append(buf.cpu().numpy())
        got.append(torch.full((w * h * 4,), 0xa5, dtype=torch.uint8, device="cuda:0"))
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(2)]
    # warm both frames up on their streams (a variant's first launch sets a function attribute, a
    # stream's first use makes the runtime set up its queue), then measure the calls
    for p, b, st in zip(params, got, streams):
        assert lib.rt_render_device_noalloc(ctypes.addressof(p), b.data_ptr(), b.numel(), ctypes.c_void_p(st.cuda_stream)) == 0, lib.rt_last_error()
    torch.cuda.synchronize()
    for b in got:
        b.fill_(0x5a)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for rep in range(3):
        for p, b, st in zip(params, got, streams):
            assert lib.rt_render_device_noalloc(ctypes.addressof(p), b.data_ptr(), b.numel(), ctypes.c_void_p(st.cuda_stream)) == 0, lib.rt_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    for w_, g_ in zip(want, got):
        assert np.array_equal(w_, g_.cpu().numpy())


def test_noalloc_under_graph_capture_replays_the_frame(gpu_rt):
    """A call of the hierarchy family recorded into a HIP graph (stream capture) and replayed: the
    replay writes the frame rt_render_device writes, and plain calls keep working afterwards."""
    import torch
    lib = gpu_rt.api.load_library()
    scene, cam = gpu_rt.mixed_scene(96), gpu_rt.mixed_camera()
    gpu_rt.set_scene(scene)
    w, h = 320, 200
    p = gpu_rt.api.make_params(cam, w, h, flags=gpu_rt.api.FLAG_BOTTOM_UP, max_depth=5, stack_size=4, shadow_hops=2)
    ref = torch.zeros((w * h * 4,), dtype=torch.uint8, device="cuda:0")
    out = torch.zeros((w * h * 4,), dtype=torch.uint8, device="cuda:0")
    assert lib.rt_render_device(ctypes.addressof(p), ref.data_ptr(), ref.numel(), None, 0) == 0, lib.rt_last_error()
    # the variant's first launch on the device happens outside the capture
    assert lib.rt_render_device_noalloc(ctypes.addressof(p), out.data_ptr(), out.numel(), None) == 0, lib.rt_last_error()
    torch.cuda.synchronize()
    out.zero_()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device="cuda:0")
    with torch.cuda.graph(graph, stream=side):
        rc = lib.rt_render_device_noalloc(ctypes.addressof(p), out.data_ptr(), out.numel(), ctypes.c_void_p(side.cuda_stream))
    assert rc == 0, lib.rt_last_error()
    torch.cuda.synchronize()
    assert int(out.max()) == 0                      # captured, not executed
    for _ in range(2):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), ref.cpu().numpy())
    # a plain call after the capture still waits on / records the library's event without error
    out.zero_()
    assert lib.rt_render_device_noalloc(ctypes.addressof(p), out.data_ptr(), out.numel(), None) == 0, lib.rt_last_error()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref.cpu().numpy())
