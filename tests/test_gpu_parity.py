"""Parity of the HIP renderer (through the C ABI of librt355.so) with the reference.

Bar (strict flavour, the default): BIT-EXACT.  RGBA8 frames, unclamped float32 colours, the
object id of every traced segment, primary t and the per-frame work counters equal the
golden vectors rendered by the reference shader on llvmpipe and the CPU oracle, torus
pixels included.
Bar (fast flavour): primary hit/miss mask equal outside torus pixels (any pixel whose
primary ray passes the torus bounding sphere is a stated exclusion), and colour within
1/255 per channel on >= 99.9 % of the pixels outside the exclusion and >= 99 % of all.
"""
import ctypes
import hashlib
import os


import numpy as np
import pytest

from conftest import camera_for, golden_names, load_golden, scene_for

pytestmark = pytest.mark.gpu

FRAME_CASES = [n for n in golden_names() if not n.startswith(("spp", "light"))]
SMALL = [n for n in FRAME_CASES if "4k" not in n]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def f32_identical(a, b):
    """Bitwise equality, except that any NaN equals any NaN (the sign/payload of a NaN is
    not defined by GLSL and differs between x86 and gfx950; both write 0 to the RGBA8 target)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def opts(meta):
    return dict(max_depth=meta["max_depth"], stack_size=meta["stack_size"], shadow_hops=meta["shadow_hops"])


@pytest.mark.parametrize("name", FRAME_CASES)
def test_strict_frame_bit_exact_vs_reference(gpu_rt, orc, name):
    meta, arr = load_golden(name)
    cam, scene = camera_for(meta), scene_for(name)
    img = gpu_rt.render_rgba(scene, cam, meta["w"], meta["h"], bottom_up=True, **opts(meta))
    assert sha(img) == meta["sha256_rgba8"]
    if "rgba8" in arr:
        assert np.array_equal(img, arr["rgba8"])
    f32 = gpu_rt.render_rgba(scene, cam, meta["w"], meta["h"], bottom_up=True, out_f32=True, **opts(meta))
    if sha(f32) != meta["sha256_rgba32f"]:
        # only legitimate cause: NaN pixels (their bit pattern is platform-specific)
        if "rgba32f" in arr:
            ref = arr["rgba32f"]
        else:
            ref = orc.render(scene, cam.matrix(), cam.focal_length, meta["w"], meta["h"], light=cam.light_pos,
                             want_f32=True, **opts(meta))["f32"]
            assert sha(ref) == meta["sha256_rgba32f"]
        assert np.isnan(ref).any()
        assert f32_identical(f32, ref)


@pytest.mark.parametrize("name", SMALL)
def test_strict_segment_masks_bit_exact(gpu_rt, name):
    meta, arr = load_golden(name)
    if "seg0_id" not in arr:
        pytest.skip("no segment captures stored")
    cam, scene = camera_for(meta), scene_for(name)
    for k in range(6):
        c = gpu_rt.capture_segment(scene, cam, meta["w"], meta["h"], k, **opts(meta))
        assert np.array_equal(c[..., 0].astype(np.int16), arr[f"seg{k}_id"]), f"segment {k} ids"
        if k == 0:
            assert np.array_equal(c[..., 1].view(np.uint32), arr["seg0_t"].view(np.uint32))
            assert np.array_equal(c[..., 3].astype(np.int16), arr["nseg"])


@pytest.mark.parametrize("name", ["default_320_d8s3", "tori64_480", "spheres12_d16s16h4", "mixed200_240_d5s5",
                                  "cam_inside_glass", "default_320_hops4"])
def test_work_counters_equal_oracle(gpu_rt, orc, name):
    """The instrumented kernel flavour counts exactly the events the oracle counts."""
    meta, _ = load_golden(name)
    cam, scene = camera_for(meta), scene_for(name)
    ev = gpu_rt.count_events(scene, cam, meta["w"], meta["h"], **opts(meta))
    ref = orc.render(scene, cam.matrix(), cam.focal_length, meta["w"], meta["h"], light=cam.light_pos,
                     want_events=True, **opts(meta))["events"]
    names, _ = gpu_rt.event_table()
    assert dict(zip(names, ev.tolist())) == dict(zip(names, ref.tolist()))


@pytest.mark.parametrize("N", [2, 4])
def test_stratified_samples_bit_exact(gpu_rt, N):
    meta, arr = load_golden(f"spp{N}_200")
    cam = camera_for(meta)
    f32 = gpu_rt.render_rgba(gpu_rt.default_scene(), cam, meta["w"], meta["h"], spp_grid=N, bottom_up=True, out_f32=True)
    assert np.array_equal(f32[..., :3].view(np.uint32), arr["rgb32f"].view(np.uint32))


def test_deep_config_equals_oracle(gpu_rt, orc):
    """256 glass/mirror spheres, 16 bounces, stack 16, 4 shadow hops: beyond what llvmpipe can
    run (its shader loops stop after 65535 iterations), so the oracle is the reference here."""
    scene, cam = gpu_rt.spheres_deep_scene(), gpu_rt.spheres_deep_camera()
    w, h = 160, 90
    o = dict(max_depth=16, stack_size=16, shadow_hops=4)
    f32 = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, out_f32=True, **o)
    ref = orc.render(scene, cam.matrix(), cam.focal_length, w, h, want_f32=True, **o)["f32"]
    assert f32_identical(f32, ref)


@pytest.mark.parametrize("case", ["mixed200", "mixed1024", "spheres256", "tori64"])
@pytest.mark.parametrize("size", [(203, 117), (64, 40), (8, 8)])
def test_deep_frames_spread_and_cooperative_trace_equal_oracle(gpu_rt, orc, case, size):
    """max_depth >= 12 with >= 8 pending rays: the persistent hierarchy kernel hands a wave the
    pixels of 64 different tiles and traces the last few rays of a wave one at a time with all
    lanes.  Frame sizes that are not multiples of the tile, a frame of a single tile (where the
    spread falls back to whole tiles), spheres and tori, several samples per pixel."""
    scene, cam = {"mixed200": (lambda: gpu_rt.mixed_scene(200), gpu_rt.mixed_camera),
                  "mixed1024": (lambda: gpu_rt.mixed_scene(1024), gpu_rt.mixed_camera),
                  "spheres256": (gpu_rt.spheres_deep_scene, gpu_rt.spheres_deep_camera),
                  "tori64": (gpu_rt.tori_stress_scene, gpu_rt.tori_stress_camera)}[case]
    scene, cam = scene(), cam()
    w, h = size
    o = dict(max_depth=12, stack_size=9, shadow_hops=3)
    if case == "mixed200":
        o["spp_grid"] = 2
    f32 = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, out_f32=True, **o)
    ref = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, want_f32=True, **o)["f32"]
    assert f32_identical(f32, ref)


@pytest.mark.parametrize("case", ["mixed200", "mixed1024", "spheres256", "tori64"])
@pytest.mark.parametrize("size", [(203, 117), (64, 40), (9, 3)])
def test_deep_frames_in_cost_order_equal_oracle(gpu_rt, orc, case, size):
    """Deep frames with the pixels handed out in the order of a pilot launch (largest ray trees first,
    dealt over the waves; what large deep frames do by themselves, forced here at small sizes):
    ragged sizes, a frame smaller than one queue entry, tori, several samples per pixel, and a
    band-split part."""
    scene, cam = {"mixed200": (lambda: gpu_rt.mixed_scene(200), gpu_rt.mixed_camera),
                  "mixed1024": (lambda: gpu_rt.mixed_scene(1024), gpu_rt.mixed_camera),
                  "spheres256": (gpu_rt.spheres_deep_scene, gpu_rt.spheres_deep_camera),
                  "tori64": (gpu_rt.tori_stress_scene, gpu_rt.tori_stress_camera)}[case]
    scene, cam = scene(), cam()
    w, h = size
    o = dict(max_depth=12, stack_size=9, shadow_hops=3)
    if case == "mixed200":
        o["spp_grid"] = 2
    f32 = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, out_f32=True, pixel_order=True, **o)
    ref = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, want_f32=True, **o)["f32"]
    assert f32_identical(f32, ref)
    if size == (203, 117):
        from importlib import import_module
        tiling = import_module("ray-tracing_amd.tiling")
        full = gpu_rt.render_rgba(scene, cam, w, h, pixel_order=True, **o)
        for nparts in (3,):
            rows = tiling.padded_part_rows(h, nparts)
            gathered = np.zeros((nparts, rows, w, 4), dtype=np.uint8)
            for part in range(nparts):
                p = gpu_rt.render_rgba(scene, cam, w, h, part=part, nparts=nparts, pixel_order=True, **o)
                gathered[part, :p.shape[0]] = p
            assert np.array_equal(tiling.assemble_numpy(gathered, h, nparts), full)


def test_deep_frame_in_parts_equals_whole_frame(gpu_rt):
    """The spread of pixels over waves is per part (its own tile sequence): parts reassemble to the frame."""
    from importlib import import_module
    tiling = import_module("ray-tracing_amd.tiling")
    scene, cam = gpu_rt.spheres_deep_scene(), gpu_rt.spheres_deep_camera()
    w, h = 200, 150
    o = dict(max_depth=16, stack_size=16, shadow_hops=4)
    full = gpu_rt.render_rgba(scene, cam, w, h, **o)
    for nparts in (3, 7):
        rows = tiling.padded_part_rows(h, nparts)
        gathered = np.zeros((nparts, rows, w, 4), dtype=np.uint8)
        for part in range(nparts):
            p = gpu_rt.render_rgba(scene, cam, w, h, part=part, nparts=nparts, **o)
            gathered[part, :p.shape[0]] = p
        assert np.array_equal(tiling.assemble_numpy(gathered, h, nparts), full)


def test_1024_objects_16spp_equals_oracle(gpu_rt, orc):
    """The 1024-object scene (does not link on llvmpipe) with 4x4 samples per pixel."""
    scene, cam = gpu_rt.mixed_scene(1024), gpu_rt.mixed_camera()
    w, h = 64, 36
    o = dict(max_depth=8, stack_size=8, shadow_hops=2, spp_grid=4)
    f32 = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, out_f32=True, **o)
    ref = orc.render(scene, cam.matrix(), cam.focal_length, w, h, want_f32=True, **o)["f32"]
    assert f32_identical(f32, ref)


def test_row_order_and_rgb_entry_point(gpu_rt):
    scene, cam = gpu_rt.default_scene(), gpu_rt.default_camera()
    top_down = gpu_rt.render(scene, cam, 120, 90)
    bottom_up = gpu_rt.render_rgba(scene, cam, 120, 90, bottom_up=True)
    assert top_down.shape == (90, 120, 3) and top_down.dtype == np.uint8
    assert np.array_equal(top_down, bottom_up[::-1, :, :3])
    assert (bottom_up[..., 3] == 255).all()


@pytest.mark.parametrize("nparts", [2, 3, 8])
def test_band_split_reassembles_to_full_frame(gpu_rt, nparts):
    from importlib import import_module
    tiling = import_module("ray-tracing_amd.tiling")
    scene, cam = gpu_rt.default_scene(), gpu_rt.default_camera()
    w, h = 200, 150
    full = gpu_rt.render_rgba(scene, cam, w, h)
    rows = tiling.padded_part_rows(h, nparts)
    gathered = np.zeros((nparts, rows, w, 4), dtype=np.uint8)
    for part in range(nparts):
        p = gpu_rt.render_rgba(scene, cam, w, h, part=part, nparts=nparts)
        assert p.shape[0] == tiling.part_rows(h, part, nparts)
        gathered[part, :p.shape[0]] = p
    assert np.array_equal(tiling.assemble_numpy(gathered, h, nparts), full)


def test_device_assemble_kernel(gpu_rt):
    """rt_render_device + rt_assemble_device on caller-owned device memory (torch tensors)."""
    import ctypes
    import torch
    from importlib import import_module
    tiling = import_module("ray-tracing_amd.tiling")
    scene, cam = gpu_rt.default_scene(), gpu_rt.default_camera()
    w, h, nparts = 208, 150, 4
    lib = gpu_rt.load_library()
    gpu_rt.set_scene(scene)
    rows = tiling.padded_part_rows(h, nparts)
    gathered = torch.zeros((nparts, rows, w, 4), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for part in range(nparts):
        p = gpu_rt.api.make_params(cam, w, h, part=part, nparts=nparts)
        rc = lib.rt_render_device(ctypes.addressof(p), gathered[part].data_ptr(), gathered[part].numel(), stream, 0)
        assert rc == 0, lib.rt_last_error()
    image = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda:0")
    assert lib.rt_assemble_device(gathered.data_ptr(), image.data_ptr(), w, h, nparts, 4, stream) == 0
    torch.cuda.synchronize()
    assert np.array_equal(image.cpu().numpy(), gpu_rt.render_rgba(scene, cam, w, h))


@pytest.mark.parametrize("case", ["default", "default_deep_stack", "default_4spp", "large_scene"])
@pytest.mark.parametrize("size", [(200, 150), (37, 23), (1, 1), (203, 7)])
def test_packed_rgb8_output_equals_rgba8(gpu_rt, case, size):
    """RT_FLAG_OUT_RGB8 (what render() and the inter-GPU transport use) writes the RGB bytes of
    the RGBA8 frame, three per pixel, from every kernel variant and at odd widths."""
    w, h = size
    if case == "large_scene":
        scene, cam, o = gpu_rt.random_large_scene(5)
    else:
        scene, cam = gpu_rt.default_scene(), gpu_rt.default_camera()
        o = {"default": dict(max_depth=8), "default_deep_stack": dict(max_depth=8, stack_size=8),
             "default_4spp": dict(max_depth=4, spp_grid=2)}[case]
    rgba = gpu_rt.render_rgba(scene, cam, w, h, **o)
    rgb = gpu_rt.render(scene, cam, w, h, **o)
    assert rgb.shape == (h, w, 3) and rgb.dtype == np.uint8
    assert np.array_equal(rgb, rgba[..., :3])
    assert (rgba[..., 3] == 255).all()


@pytest.mark.parametrize("w,h,nparts", [(208, 150, 4), (201, 77, 3), (64, 16, 2)])
def test_device_assemble_kernel_rgb8_parts(gpu_rt, w, h, nparts):
    """Parts rendered as packed RGB8, assembled on the device into the RGBA8 frame (both the
    four-pixels-per-thread and the one-pixel-per-thread form of the kernel)."""
    import ctypes
    import torch
    from importlib import import_module
    tiling = import_module("ray-tracing_amd.tiling")
    scene, cam = gpu_rt.default_scene(), gpu_rt.default_camera()
    lib = gpu_rt.load_library()
    gpu_rt.set_scene(scene)
    rows = tiling.padded_part_rows(h, nparts)
    gathered = torch.zeros((nparts, rows, w, 3), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for part in range(nparts):
        p = gpu_rt.api.make_params(cam, w, h, part=part, nparts=nparts, flags=gpu_rt.api.FLAG_OUT_RGB8)
        assert lib.rt_output_bytes(ctypes.addressof(p)) == tiling.part_rows(h, part, nparts) * w * 3
        rc = lib.rt_render_device(ctypes.addressof(p), gathered[part].data_ptr(), gathered[part].numel(), stream, 0)
        assert rc == 0, lib.rt_last_error()
    image = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda:0")
    assert lib.rt_assemble_device(gathered.data_ptr(), image.data_ptr(), w, h, nparts, 3, stream) == 0
    torch.cuda.synchronize()
    assert np.array_equal(image.cpu().numpy(), gpu_rt.render_rgba(scene, cam, w, h))


def test_output_format_flags_exclude_each_other(gpu_rt):
    import ctypes
    lib = gpu_rt.load_library()
    gpu_rt.set_scene(gpu_rt.default_scene())
    p = gpu_rt.api.make_params(gpu_rt.default_camera(), 16, 16,
                               flags=gpu_rt.api.FLAG_OUT_RGB8 | gpu_rt.api.FLAG_OUT_F32)
    buf = np.zeros(16 * 16 * 16, dtype=np.uint8)
    assert lib.rt_render(None, 0, ctypes.addressof(p), buf.ctypes.data, buf.nbytes) != 0
    assert b"exclude" in lib.rt_last_error()


def test_repeated_frames_on_two_streams_are_identical(gpu_rt):
    """Repeated frames of interleaved geometries and parts, launched on two streams, always give
    the first frame (nothing carries over from launch to launch)."""
    import ctypes
    import torch
    scene, cam = gpu_rt.default_scene(), gpu_rt.default_camera()
    lib = gpu_rt.load_library()
    gpu_rt.set_scene(scene)
    geoms = [(200, 150, 0, 1), (37, 23, 0, 1), (640, 360, 0, 1), (640, 360, 1, 3), (203, 77, 2, 3), (16, 16, 0, 1),
             (1, 1, 0, 1)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    first = {}
    for rep in range(4):
        for gi, (w, h, part, nparts) in enumerate(geoms):
            p = gpu_rt.api.make_params(cam, w, h, max_depth=8, part=part, nparts=nparts)
            nbytes = lib.rt_output_bytes(ctypes.addressof(p))
            for si, st in enumerate(streams):
                if rep % 2 == 1 and si == 1:
                    continue                      # uneven use of the streams
                buf = torch.full((nbytes,), 7, dtype=torch.uint8, device="cuda:0")
                st.wait_stream(torch.cuda.current_stream())
                rc = lib.rt_render_device(ctypes.addressof(p), buf.data_ptr(), nbytes, ctypes.c_void_p(st.cuda_stream), 0)
                assert rc == 0, lib.rt_last_error()
                st.synchronize()
                got = buf.cpu().numpy()
                if gi not in first:
                    first[gi] = got
                    full = gpu_rt.render_rgba(scene, cam, w, h, max_depth=8, part=part, nparts=nparts)
                    assert np.array_equal(got, full.ravel())
                assert np.array_equal(got, first[gi]), (rep, gi, si)


@pytest.mark.parametrize("case", ["default", "default_hops0", "default_deep_stack", "default_2spp", "large_scene", "tori"])
def test_area_light_equals_oracle(gpu_rt, orc, case):
    """G x G light samples per shaded hit (every kernel family): float32 frame, RGBA8 frame and the
    event counters equal the oracle's, bit for bit."""
    w, h = 120, 90
    scene, cam = gpu_rt.default_scene(), gpu_rt.default_camera()
    o = dict(max_depth=4, stack_size=3, shadow_hops=2, light_grid=2, light_size=1.5)
    if case == "default_hops0":
        o.update(shadow_hops=0, light_grid=3)
    elif case == "default_deep_stack":
        o.update(max_depth=8, stack_size=8, light_grid=3, light_size=0.7)
    elif case == "default_2spp":
        o.update(spp_grid=2)
    elif case == "large_scene":
        scene, cam, lo = gpu_rt.random_large_scene(7)
        o.update(lo)
        w, h = 96, 64
    elif case == "tori":
        scene, cam = gpu_rt.tori_stress_scene(), gpu_rt.tori_stress_camera()
        o.update(max_depth=3)
        w, h = 96, 54
    f32 = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, out_f32=True, **o)
    ref = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, want_f32=True, want_events=True, **o)
    assert f32_identical(f32, ref["f32"])
    img = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, **o)
    assert np.array_equal(img, ref["rgba8"])
    ev = gpu_rt.count_events(scene, cam, w, h, **o)
    assert np.array_equal(ev, ref["events"])


def test_area_light_parameter_limits(gpu_rt):
    scene, cam = gpu_rt.default_scene(), gpu_rt.default_camera()
    with pytest.raises(gpu_rt.RtError):
        gpu_rt.render(scene, cam, 16, 16, light_grid=9, light_size=1.0)
    with pytest.raises(gpu_rt.RtError):
        gpu_rt.render(scene, cam, 16, 16, light_grid=2, light_size=-1.0)
    a = gpu_rt.render_rgba(scene, cam, 64, 48, out_f32=True)
    for G, size in ((0, 0.0), (1, 3.0)):
        b = gpu_rt.render_rgba(scene, cam, 64, 48, out_f32=True, light_grid=G, light_size=size)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_band_pipeline_streams_rccl_one_rank(gpu_rt):
    """The multi-GPU frame pipeline as far as one GPU can run it: a one-rank RCCL group, parts
    travelling as RGB8 through dist.gather, frames rotating over three streams, a different
    camera every frame.  Every delivered frame must be that camera's frame."""
    import ctypes
    import socket
    import torch
    import torch.distributed as dist
    from importlib import import_module
    tiling = import_module("ray-tracing_amd.tiling")
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        lib = gpu_rt.load_library()
        scene = gpu_rt.default_scene()
        gpu_rt.set_scene(scene)
        w, h = 208, 150
        cams = []
        for k in range(5):
            c = gpu_rt.default_camera()
            c.yaw += 3.0 * k            # degrees (main.py:102-103)
            c.pitch -= 1.5 * k
            cams.append(c)
        params = [gpu_rt.api.make_params(c, w, h, max_depth=6, flags=gpu_rt.api.FLAG_OUT_RGB8) for c in cams]
        streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
        sptrs = [ctypes.c_void_p(st.cuda_stream) for st in streams]
        frame = {"k": 0}

        def render_part(part, nparts, out):
            p = params[frame["k"]]
            assert lib.rt_render_device(ctypes.addressof(p), out.data_ptr(), out.numel(), sptrs[pipe.cur], 0) == 0

        def assemble(gathered, image):
            assert lib.rt_assemble_device(gathered.data_ptr(), image.data_ptr(), w, h, 1, 3, sptrs[pipe.cur]) == 0

        pipe = tiling.BandPipeline(h, w, 3, torch.uint8, dev, dist, render_part, assemble, always_gather=True,
                                   image_channels=4, streams=streams)
        want = [gpu_rt.render_rgba(scene, c, w, h, max_depth=6) for c in cams]
        assert all(not np.array_equal(want[k], want[k + 1]) for k in range(4))
        # frame by frame
        for k in range(3):
            frame["k"] = k
            pipe.step()
            pipe.flush()
            torch.cuda.synchronize()
            assert np.array_equal(pipe.image.cpu().numpy(), want[k]), k
        # several frames in flight: the last one delivered is the last one rendered
        for k in range(5):
            frame["k"] = k
            pipe.step()
        pipe.flush()
        torch.cuda.synchronize()
        assert np.array_equal(pipe.image.cpu().numpy(), want[4])
    finally:
        dist.destroy_process_group()


def test_fast_flavour_within_stated_tolerance(gpu_rt):
    meta, arr = load_golden("default_800")
    cam, scene = camera_for(meta), scene_for("default_800")
    fast = gpu_rt.render_rgba(scene, cam, 800, 800, bottom_up=True, fast_math=True)
    ref = arr["rgb8"]
    close = np.all(np.abs(fast[..., :3].astype(int) - ref.astype(int)) <= 1, axis=2)
    # exclusion: pixels whose primary ray passes the torus bounding sphere
    cap = gpu_rt.capture_segment(scene, cam, 800, 800, 0, fast_math=True)
    ids_fast = cap[..., 0].astype(np.int16)
    ids_ref = arr["seg0_id"]
    M = cam.matrix().astype(np.float64)
    ys, xs = np.mgrid[0:800, 0:800]
    uvx = ((xs + 0.5) * 2 - 800) / 800
    uvy = ((ys + 0.5) * 2 - 800) / 800
    d = uvx[..., None] * M[:3, 0] + uvy[..., None] * M[:3, 1] + 2.0 * M[:3, 2]
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    oc = M[:3, 3] - np.array([0.0, 1.2, -3.0])
    b = d @ oc
    torus_zone = b * b - (oc @ oc - 1.0 * 1.0) >= -1e-3
    mask_diff = (ids_fast != ids_ref) & ~torus_zone
    assert mask_diff.sum() <= 4, f"{mask_diff.sum()} primary-mask differences outside the torus zone"
    assert close[~torus_zone].mean() >= 0.999
    assert close.mean() >= 0.99


def test_errors_are_reported_not_fatal(gpu_rt):
    scene, cam = gpu_rt.default_scene(), gpu_rt.default_camera()
    bad = gpu_rt.default_scene()
    bad[2]["type"] = 7
    with pytest.raises(gpu_rt.RtError) as e:
        gpu_rt.render(bad, cam, 16, 16)
    assert e.value.code == -2 and "type 7" in str(e.value)
    bad = gpu_rt.default_scene()
    bad[0]["radius"] = float("nan")
    with pytest.raises(gpu_rt.RtError):
        gpu_rt.render(bad, cam, 16, 16)
    with pytest.raises(gpu_rt.RtError) as e:
        gpu_rt.render(scene, cam, 16, 16, max_depth=40, stack_size=40)
    assert e.value.code == -3
    with pytest.raises(gpu_rt.RtError):
        gpu_rt.render(scene, cam, 0, 16)
    # depth 40 with a 3-entry stack is fine: only min(stack, depth-1) rays are ever pending
    img = gpu_rt.render(scene, cam, 32, 24, max_depth=40, stack_size=3)
    assert img.shape == (24, 32, 3)
    # the library still works after the errors
    assert gpu_rt.render(scene, cam, 16, 16).shape == (16, 16, 3)


def test_4k_headline_properties(gpu_rt):
    """Full-size frame of the headline workload: SHA equals the reference's, and halves of the
    frame rendered as separate parts agree with it (size-independent consistency)."""
    meta, _ = load_golden("default_4k_d8s3")
    cam, scene = camera_for(meta), scene_for("default_4k_d8s3")
    img = gpu_rt.render_rgba(scene, cam, 3840, 2160, bottom_up=True, **opts(meta))
    assert sha(img) == meta["sha256_rgba8"]
    t = gpu_rt.timing()
    assert t["kernel_ms"] > 0 and t["pending_cap"] == 3


@pytest.mark.parametrize("seed", range(int(os.environ.get("RT_TEST_SEEDS", "12"))))     # RT_TEST_SEEDS=N: soak run
def test_random_scenes_bit_exact_vs_oracle(gpu_rt, orc, seed):
    scene, cam, o = gpu_rt.random_scene(seed)
    n = len(scene)
    w, h = 160, 120
    f32 = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, out_f32=True, **o)
    ref = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, want_f32=True, **o)["f32"]
    assert f32_identical(f32, ref), f"seed {seed}: n={n} {o}"
    img = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, **o)
    ref8 = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, **o)["rgba8"]
    assert np.array_equal(img, ref8)
    ev = gpu_rt.count_events(scene, cam, w, h, **o)
    ref_ev = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, want_events=True, **o)["events"]
    assert np.array_equal(ev, ref_ev)


@pytest.mark.parametrize("seed", range(int(os.environ.get("RT_TEST_SEEDS", "16"))))
def test_cluster_scenes_bit_exact_vs_oracle(gpu_rt, orc, seed):
    """Rays that miss the sphere around the whole scene skip the object loop: clusters far from the
    origin, seen from far away, from just outside and from inside, wide and narrow fields of view."""
    scene, cam, o = gpu_rt.cluster_scene(seed)
    w, h = 128, 96
    f32 = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, out_f32=True, **o)
    ref = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, want_f32=True, **o)["f32"]
    assert f32_identical(f32, ref), f"seed {seed}: n={len(scene)} {o}"
    img = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, **o)
    ref8 = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, **o)["rgba8"]
    assert np.array_equal(img, ref8)


@pytest.mark.parametrize("seed", range(6))
def test_random_scenes_with_area_light_and_samples_vs_oracle(gpu_rt, orc, seed):
    """The same nasty small scenes with an area light (and, on odd seeds, 2 x 2 pixel samples)."""
    scene, cam, o = gpu_rt.random_scene(seed)
    o = dict(o, light_grid=2 + seed % 2, light_size=0.5 + 0.75 * seed, spp_grid=1 + seed % 2)
    w, h = 96, 72
    f32 = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, out_f32=True, **o)
    ref = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, want_f32=True, **o)["f32"]
    assert f32_identical(f32, ref), f"seed {seed}: n={len(scene)} {o}"


# ---- the sphere hierarchy used on large scenes returns the per-object loop's hits, bit for bit

def _accel_vs_plain(gpu_rt, scene, cam, w, h, **o):
    fast = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, out_f32=True, **o)
    lds_fast = gpu_rt.timing()["lds_bytes"]
    plain = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, out_f32=True, no_accel=True, **o)
    # two different kernels ran: the hierarchy variant keeps the node records and a 16 KB
    # candidate queue in LDS, the per-object loop the whole scene
    assert lds_fast != gpu_rt.timing()["lds_bytes"]
    return fast, plain


@pytest.mark.parametrize("seed", range(9))
def test_hierarchy_equals_per_object_loop_random_large(gpu_rt, orc, seed):
    scene, cam, o = gpu_rt.random_large_scene(seed)
    w, h = 240, 136
    fast, plain = _accel_vs_plain(gpu_rt, scene, cam, w, h, **o)
    assert f32_identical(fast, plain), f"seed {seed}: n={len(scene)} {o}"
    if seed < 4:
        ws, hs = 96, 54
        small = gpu_rt.render_rgba(scene, cam, ws, hs, bottom_up=True, out_f32=True, **o)
        ref = orc.render(scene, cam.matrix(), cam.focal_length, ws, hs, light=cam.light_pos, want_f32=True, **o)["f32"]
        assert f32_identical(small, ref), f"seed {seed}: n={len(scene)} {o}"


@pytest.mark.parametrize("case", ["lists", "lists_spp2", "tori_pooled", "tori_pooled_spp2", "deep", "deep_256_spheres"])
def test_mid_sized_scenes_with_the_whole_scene_in_lds(gpu_rt, orc, case):
    """Hierarchy kernels of scenes whose blob fits next to the work lists in a quarter of a CU's LDS (a third
    for small deep frames) stage all of it there, not only the hierarchy sections: one instantiation each for
    per-lane lists / pooled trace, one / several samples per pixel, deep frames.  Frames equal the per-object
    loop's and the oracle's bit for bit, and the launch's LDS size shows that the variant ran."""
    scene, cam, w, h, o = {
        "lists": (gpu_rt.mixed_scene(120, seed=3), gpu_rt.mixed_camera(), 200, 112, dict(max_depth=6, stack_size=5, shadow_hops=2)),
        "lists_spp2": (gpu_rt.mixed_scene(120, seed=3), gpu_rt.mixed_camera(), 120, 68, dict(max_depth=5, stack_size=3, shadow_hops=2, spp_grid=2)),
        "tori_pooled": (gpu_rt.tori_stress_scene(), gpu_rt.tori_stress_camera(), 200, 112, dict(max_depth=8, stack_size=3, shadow_hops=2)),
        "tori_pooled_spp2": (gpu_rt.tori_stress_scene(), gpu_rt.tori_stress_camera(), 120, 68, dict(max_depth=4, stack_size=3, shadow_hops=2, spp_grid=2)),
        "deep": (gpu_rt.mixed_scene(150, seed=9), gpu_rt.mixed_camera(), 160, 90, dict(max_depth=12, stack_size=9, shadow_hops=2)),
        "deep_256_spheres": (gpu_rt.spheres_deep_scene(), gpu_rt.spheres_deep_camera(), 160, 90, dict(max_depth=16, stack_size=16, shadow_hops=4)),
    }[case]
    fast, plain = _accel_vs_plain(gpu_rt, scene, cam, w, h, **o)
    assert f32_identical(fast, plain), f"{case}: n={len(scene)} {o}"
    ref = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, want_f32=True, **o)["f32"]
    assert f32_identical(fast, ref), f"{case}: n={len(scene)} {o}"
    gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, out_f32=True, **o)
    lds = gpu_rt.timing()["lds_bytes"]
    blob_bytes = (4 * gpu_rt.stage_scene_host(scene)[0]["total_dwords"] + 15) & ~15
    per_lane_queue, pooled_lists, three_per_cu = 32 * 256 * 2, 4 * 1792 * 4, 41 * 1024
    assert lds in (blob_bytes + per_lane_queue, blob_bytes + pooled_lists, three_per_cu), (case, lds, blob_bytes)
    assert lds >= blob_bytes + per_lane_queue


def test_hierarchy_equals_per_object_loop_named_scenes(gpu_rt):
    for scene, cam, w, h, o in [
        (gpu_rt.spheres_deep_scene(), gpu_rt.spheres_deep_camera(), 480, 270, dict(max_depth=16, stack_size=16, shadow_hops=4)),
        (gpu_rt.mixed_scene(1024), gpu_rt.mixed_camera(), 480, 270, dict(max_depth=8, stack_size=8, shadow_hops=2)),
        (gpu_rt.mixed_scene(1024), gpu_rt.mixed_camera(), 160, 90, dict(max_depth=4, stack_size=3, shadow_hops=2, spp_grid=2)),
        (gpu_rt.mixed_scene(333, seed=7), gpu_rt.mixed_camera(), 320, 180, dict(max_depth=5, stack_size=2, shadow_hops=1)),
    ]:
        fast, plain = _accel_vs_plain(gpu_rt, scene, cam, w, h, **o)
        assert f32_identical(fast, plain), f"n={len(scene)} {o}"


def test_hierarchy_candidate_queue_overflow(gpu_rt, orc):
    """A ray down a row of 120 spheres passes more bounding spheres than the candidate queue
    holds; such rays walk all objects and must still agree."""
    scene = gpu_rt.sphere_row_scene(120)
    cam = gpu_rt.Camera(eye=(0.0, 0.5, 3.0), yaw=-90.0, pitch=0.0, focal_length=3.0)
    o = dict(max_depth=6, stack_size=4, shadow_hops=2)
    fast, plain = _accel_vs_plain(gpu_rt, scene, cam, 200, 120, **o)
    assert f32_identical(fast, plain)
    ref = orc.render(scene, cam.matrix(), cam.focal_length, 200, 120, want_f32=True, **o)["f32"]
    assert f32_identical(fast, ref)


def test_range_restricted_arithmetic_equals_general_forms_exhaustively(gpu_rt):
    """The strict kernels run sqrt / 1/sqrt / log2's quotient without the operand scaling and
    special-value steps when the operand is well inside the normal range; every binary32
    operand of those ranges gives the bits of the general (correctly rounded) forms."""
    counts = gpu_rt.selftest_math()
    assert counts.tolist() == [0, 0, 0, 0, 0, 0]


@pytest.mark.parametrize("w,h,nparts", [(203, 117, 3), (64, 40, 2), (333, 31, 5), (1, 1, 1), (7, 3, 2), (1, 40, 3), (9, 17, 8)])
def test_persistent_wave_kernel_ragged_sizes_and_band_split(gpu_rt, orc, w, h, nparts):
    """Large scene + deep stack runs the persistent-wave kernel (tile queue, pixel refill):
    sizes that are not multiples of the 8x8 tiles / 16-row bands, whole and split in parts."""
    scene, cam = gpu_rt.mixed_scene(200, seed=11), gpu_rt.mixed_camera()
    o = dict(max_depth=8, stack_size=8, shadow_hops=2)
    full = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=False, **o)
    assert gpu_rt.timing()["pending_cap"] == 16          # the workspace variant (capacity 16 whatever the stack size)
    ref = orc.render(scene, cam.matrix(), cam.focal_length, w, h, **o)["rgba8"][::-1]
    assert np.array_equal(full, ref)
    from importlib import import_module
    tiling = import_module("ray-tracing_amd.tiling")
    parts = []
    rows = tiling.padded_part_rows(h, nparts)
    for part in range(nparts):
        buf = gpu_rt.render_rgba(scene, cam, w, h, part=part, nparts=nparts, **o)
        pad = np.zeros((rows, w, 4), dtype=np.uint8)
        pad[:buf.shape[0]] = buf
        parts.append(pad)
    assert np.array_equal(tiling.assemble_numpy(np.stack(parts), h, nparts), full)


def test_large_scene_launches_on_two_streams_do_not_share_the_workspace_at_once(gpu_rt):
    """The persistent-wave kernels keep their pending rays in one workspace; launches on
    different streams are ordered on the GPU by the library, so un-synchronised callers still
    get the right frames."""
    import ctypes
    import torch
    scene, cam = gpu_rt.spheres_deep_scene(), gpu_rt.spheres_deep_camera()
    w, h = 320, 180
    o = dict(max_depth=10, stack_size=10, shadow_hops=2)
    ref = gpu_rt.render_rgba(scene, cam, w, h, no_accel=True, **o)
    lib = gpu_rt.load_library()
    gpu_rt.set_scene(scene)
    p = gpu_rt.api.make_params(cam, w, h, **o)
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(2)]
    bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0") for _ in range(6)]
    torch.cuda.synchronize()
    for i, b in enumerate(bufs):
        st = streams[i % 2]
        assert lib.rt_render_device(ctypes.addressof(p), b.data_ptr(), b.numel(), ctypes.c_void_p(st.cuda_stream), 0) == 0
    torch.cuda.synchronize()
    for b in bufs:
        assert np.array_equal(b.cpu().numpy(), ref)


def test_fast_flavour_runs_the_large_scene_kernels(gpu_rt):
    """The opt-in fast flavour shares the hierarchy / persistent-wave code; on a 200-object scene
    it must finish and stay close to the strict frame (its tolerance is loose by design)."""
    scene, cam = gpu_rt.mixed_scene(200, seed=3), gpu_rt.mixed_camera()
    o = dict(max_depth=6, stack_size=6, shadow_hops=2)
    strict = gpu_rt.render_rgba(scene, cam, 320, 180, **o).astype(int)
    fast = gpu_rt.render_rgba(scene, cam, 320, 180, fast_math=True, **o).astype(int)
    close = np.all(np.abs(fast - strict) <= 1, axis=2).mean()
    assert close >= 0.85, close        # 50 tori: the fast flavour moves torus pixels (0.936 measured)


def test_zz_shutdown_and_reinit_cycle(gpu_rt):
    """rt_shutdown releases everything and rt_init can be called again (the reference's
    cleanup + a new App); rendering without a scene after re-init is an error, not a crash."""
    import ctypes
    scene, cam = gpu_rt.default_scene(), gpu_rt.default_camera()
    before = gpu_rt.render_rgba(scene, cam, 96, 64)
    gpu_rt.shutdown()
    lib = gpu_rt.init(0)
    p = gpu_rt.api.make_params(cam, 96, 64)
    buf = np.zeros((64, 96, 4), dtype=np.uint8)
    assert lib.rt_render_device(ctypes.addressof(p), buf.ctypes.data, buf.nbytes, None, 0) == -4      # RT_E_STATE: no scene
    assert b"scene" in lib.rt_last_error()
    assert np.array_equal(gpu_rt.render_rgba(scene, cam, 96, 64), before)
    big = gpu_rt.mixed_scene(64)
    a = gpu_rt.render_rgba(big, gpu_rt.mixed_camera(), 96, 64, max_depth=6, stack_size=6)
    b = gpu_rt.render_rgba(big, gpu_rt.mixed_camera(), 96, 64, max_depth=6, stack_size=6, no_accel=True)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("which", ["spheres256_d16", "mixed1024_16spp", "tori64_d8s8"])
def test_heavy_configs_at_larger_sizes_equal_oracle(gpu_rt, orc, which):
    """The big configurations at larger sizes (a few seconds of CPU oracle time on the GPU box's
    cores)."""
    if which == "spheres256_d16":
        scene, cam, w, h, o = gpu_rt.spheres_deep_scene(), gpu_rt.spheres_deep_camera(), 640, 360, dict(max_depth=16, stack_size=16, shadow_hops=4)
    elif which == "mixed1024_16spp":
        scene, cam, w, h, o = gpu_rt.mixed_scene(1024), gpu_rt.mixed_camera(), 384, 216, dict(max_depth=8, stack_size=8, shadow_hops=2, spp_grid=4)
    else:
        scene, cam, w, h, o = gpu_rt.tori_stress_scene(), gpu_rt.tori_stress_camera(), 960, 540, dict(max_depth=8, stack_size=8, shadow_hops=2)
    if os.environ.get("RT_HEAVY_FULL") == "1":       # one-off: the full benchmark sizes (minutes of CPU time)
        w, h = (1920, 1080) if which == "mixed1024_16spp" else (3840, 2160)
    elif os.environ.get("RT_HEAVY_FULL") != "0":
        # the default run: the 64 tori at their full 3840x2160; the two configurations whose CPU
        # oracle needs minutes at full size stay at a quarter of it (RT_HEAVY_FULL=1 runs them whole)
        w, h = {"tori64_d8s8": (3840, 2160), "spheres256_d16": (1920, 1080), "mixed1024_16spp": (960, 540)}[which]
    f32 = gpu_rt.render_rgba(scene, cam, w, h, bottom_up=True, out_f32=True, **o)
    ref = orc.render(scene, cam.matrix(), cam.focal_length, w, h, want_f32=True, **o)["f32"]
    assert f32_identical(f32, ref)


@pytest.mark.parametrize("which", ["spheres256_4k_d16", "mixed1024_8k_16spp"])
def test_benchmark_sizes_hierarchy_kernels_equal_per_object_loop_and_oracle_bands(gpu_rt, orc, which):
    """The two heaviest benchmark configurations at their full sizes: the frame of the hierarchy
    kernels (pooled trace; for the deep one also the pilot launch and the cost order of the pixels)
    must equal the frame of the plain per-object loop, the reference's algorithm as it stands, and
    a seeded sample of its 16-row bands must equal the CPU oracle's rows bit for bit."""
    if which == "spheres256_4k_d16":
        scene, cam, w, h = gpu_rt.spheres_deep_scene(), gpu_rt.spheres_deep_camera(), 3840, 2160
        o = dict(max_depth=16, stack_size=16, shadow_hops=4, out_f32=True)
    else:
        scene, cam, w, h = gpu_rt.mixed_scene(1024), gpu_rt.mixed_camera(), 7680, 4320
        o = dict(max_depth=8, stack_size=8, shadow_hops=2, spp_grid=4)
    fast = gpu_rt.render_rgba(scene, cam, w, h, **o)
    plain = gpu_rt.render_rgba(scene, cam, w, h, no_accel=True, **o)
    if fast.dtype == np.float32:
        assert f32_identical(fast, plain)
    else:
        assert np.array_equal(fast, plain)
    # ... and against the CPU oracle itself on a seeded sample of 16-row bands of the full-size frame
    # (about half a megapixel: seconds of CPU time, where the whole frame would need many minutes)
    rng = np.random.default_rng(355)
    nbands = h // 16
    sample = sorted(rng.choice(nbands, size=max(1, 500000 // (16 * w)), replace=False).tolist())
    oo = {k: v for k, v in o.items() if k != "out_f32"}
    for band in sample:
        y0 = band * 16                                   # GL rows (0 = bottom): output row r is GL row h - 1 - r
        ref = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, rows=(y0, y0 + 16),
                         want_f32=fast.dtype == np.float32, **oo)
        got = fast[h - y0 - 16:h - y0][::-1]
        if fast.dtype == np.float32:
            assert f32_identical(got, ref["f32"]), f"{which}: band of GL rows {y0}..{y0 + 15}"
        else:
            assert np.array_equal(got, ref["rgba8"]), f"{which}: band of GL rows {y0}..{y0 + 15}"


def test_largest_scene_blob_with_and_without_the_hierarchy(gpu_rt):
    """1024 tori = the largest blob (139 KB): the forced per-object loop has no LDS left for
    pending slots and must still run (workspace slots) and agree with the hierarchy kernels."""
    scene = gpu_rt.tori_stress_scene(1024, seed=5)
    cam = gpu_rt.tori_stress_camera()
    for o in (dict(max_depth=3, stack_size=2, shadow_hops=1), dict(max_depth=6, stack_size=6, shadow_hops=1)):
        a = gpu_rt.render_rgba(scene, cam, 64, 36, out_f32=True, **o)
        b = gpu_rt.render_rgba(scene, cam, 64, 36, out_f32=True, no_accel=True, **o)
        assert f32_identical(a, b), o


def test_render_into_pinned_buffer_4k_is_exact(gpu_rt):
    """The drop-in call of a frame loop: render(scene, camera, w, h, out=page-locked array) with an
    unchanged scene.  The frame must be the reference's (golden SHA-256 of the RGBA8 frame, here as
    RGB).  The time per frame, read-back included, is printed (about 0.76 ms for 3840x2160 on an idle
    box; 1.84 ms through pageable memory with the scene re-staged on every call)."""
    import time
    meta, _ = load_golden("default_4k_d8s3")
    cam, scene = camera_for(meta), scene_for("default_4k_d8s3")
    out = gpu_rt.pinned_empty((2160, 3840, 3))
    o = opts(meta)
    gpu_rt.render(scene, cam, 3840, 2160, out=out, **o)
    rgba = np.concatenate([out[::-1], np.full((2160, 3840, 1), 255, np.uint8)], axis=2)      # golden rows are bottom-up
    assert sha(rgba) == meta["sha256_rgba8"]
    lib = gpu_rt.api.load_library()
    packed = gpu_rt.api.pack_scene(scene)
    p = gpu_rt.api.make_params(cam, 3840, 2160, flags=gpu_rt.api.FLAG_OUT_RGB8, **o)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            assert lib.rt_render(ctypes.addressof(packed), len(scene), ctypes.addressof(p), out.ctypes.data, out.nbytes) == 0
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    # wall-clock bounds depend on the box (PCIe, host load): asserted only on request (RT_ASSERT_TIMING=1)
    timing_bounds = os.environ.get("RT_ASSERT_TIMING") == "1"
    assert not timing_bounds or best < 0.9, f"{best:.3f} ms per frame"
    t0 = time.perf_counter()
    for _ in range(10):
        gpu_rt.render(scene, cam, 3840, 2160, out=out, **o)
    py_ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"rt_render into pinned memory: {best:.3f} ms per 4K frame; through render(): {py_ms:.3f} ms")
    assert not timing_bounds or py_ms < 1.5
    rgba = np.concatenate([out[::-1], np.full((2160, 3840, 1), 255, np.uint8)], axis=2)
    assert sha(rgba) == meta["sha256_rgba8"]
