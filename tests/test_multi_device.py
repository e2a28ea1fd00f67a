"""rt_init_multi: one process, several GPUs, the library splits the frame into interleaved 16-row
bands itself and every device copies its bands straight into the frame on the first device.
On a one-GPU machine the same code path runs with one device listed several times."""
import ctypes

import numpy as np
import pytest


@pytest.mark.parametrize("nparts", [1, 2, 3, 8])
def test_part_row_map_partitions_the_rows(rt, nparts):
    lib = rt.api.load_library()
    for height in (1, 15, 16, 17, 47, 48, 270, 2160):
        seen = []
        for part in range(nparts):
            rows = rt.api.part_row_map(height, part, nparts)
            # band order, 16 rows per band, bands part, part + nparts, ...
            assert all(((r // 16) % nparts) == part for r in rows)
            assert list(rows) == sorted(rows)
            p = rt.api.make_params(rt.default_camera(), 8, height, part=part, nparts=nparts)
            assert lib.rt_output_bytes(ctypes.addressof(p)) == len(rows) * 8 * 4
            seen += list(rows)
        assert sorted(seen) == list(range(height))
    assert lib.rt_part_row_map(10, 2, 2, None, 0) < 0


@pytest.mark.gpu
def test_multi_device_mode_equals_single_device_frames(gpu_rt):
    rt = gpu_rt
    cases = [
        (rt.default_scene(), rt.default_camera(), 648, 360, dict(max_depth=8, stack_size=3)),
        (rt.default_scene(), rt.default_camera(), 203, 117, dict(max_depth=8, stack_size=8)),            # ragged, deep stack
        (rt.mixed_scene(96), rt.mixed_camera(), 320, 200, dict(max_depth=5, stack_size=4)),              # hierarchy kernels
        (rt.default_scene(), rt.default_camera(), 64, 9, dict()),                                         # fewer bands than parts
    ]
    variants = [dict(), dict(bottom_up=True), dict(out_f32=True)]
    want = [[rt.render_rgba(sc, cam, w, h, **o, **v) for v in variants] + [rt.render(sc, cam, w, h, **o)]
            for sc, cam, w, h, o in cases]
    rt.shutdown()
    try:
        for devices in ([0, 0, 0], [0], [0, 0, 0, 0, 0, 0, 0, 0]):
            lib = rt.init_multi(devices=devices)
            assert lib.rt_device_count() == len(devices)
            for (sc, cam, w, h, o), ref in zip(cases, want):
                for v, r in zip(variants, ref):
                    got = rt.render_rgba(sc, cam, w, h, **o, **v)
                    assert np.array_equal(got.view(np.uint8), r.view(np.uint8)), (devices, w, h, o, v)
                assert np.array_equal(rt.render(sc, cam, w, h, **o), ref[-1])
            # caller-owned device memory and stream
            import torch
            sc, cam, w, h, o = cases[0]
            rt.set_scene(sc)
            p = rt.api.make_params(cam, w, h, **o)
            buf = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0")
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(3):      # back to back: a frame waits for the copies of the one before
                    assert lib.rt_render_device(ctypes.addressof(p), buf.data_ptr(), buf.numel(), ctypes.c_void_p(st.cuda_stream), 0) == 0
                st.synchronize()
            assert np.array_equal(buf.cpu().numpy(), want[0][0])
            # a timed frame reports every part's kernel and copy interval, and the frame is the same
            assert lib.rt_render_device(ctypes.addressof(p), buf.data_ptr(), buf.numel(), ctypes.c_void_p(st.cuda_stream), 1) == 0, lib.rt_last_error()
            st.synchronize()
            assert np.array_equal(buf.cpu().numpy(), want[0][0])
            times = [rt.api.part_timing(i) for i in range(len(devices))]
            assert all(k > 0.0 and c >= 0.0 for k, c in times), times
            assert abs(rt.timing()["kernel_ms"] - max(k for k, _ in times)) < 1e-6
            with pytest.raises(rt.RtError):
                rt.api.part_timing(len(devices))
            # the library splits the frame itself
            p2 = rt.api.make_params(cam, w, h, part=1, nparts=2, **o)
            assert lib.rt_render_device(ctypes.addressof(p2), buf.data_ptr(), buf.numel(), None, 0) == -1
            rt.shutdown()
        lib = rt.api.load_library()
        assert lib.rt_init_multi(99) != 0
        assert lib.rt_init_multi_devices((ctypes.c_int * 2)(0, 77), 2) != 0 and b"out of range" in lib.rt_last_error()
    finally:
        rt.shutdown()
        rt.init(0)
