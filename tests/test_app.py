"""Host-side application object (the reference's App without its ray tracer): frame loop
order, FPS counter, input routing, resize, PNG output.  A stand-in renderer records the calls;
the GPU test renders the default view through the real path."""
import struct
import zlib

import numpy as np
import pytest

from conftest import load_golden


class FakeClock:
    def __init__(self):
        self.t = 100.0

    def __call__(self):
        return self.t


def fake_renderer(calls):
    def render(scene, camera, w, h, **opts):
        calls.append((len(scene), camera.eye, camera.yaw, camera.pitch, camera.focal_length, w, h, dict(opts)))
        img = np.zeros((h, w, 4), dtype=np.uint8)
        img[..., 0] = np.arange(w, dtype=np.uint8)[None, :]
        img[..., 1] = np.arange(h, dtype=np.uint8)[:, None]
        img[..., 3] = 255
        return img
    return render


def read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        (n,), tag = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(tag + body) & 0xFFFFFFFF
        chunks.append((tag, body))
        pos += 12 + n
    w, h, depth, ctype, *_ = struct.unpack(">IIBBBBB", chunks[0][1])
    assert chunks[0][0] == b"IHDR" and chunks[-1][0] == b"IEND" and depth == 8
    c = {2: 3, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(b for t, b in chunks if t == b"IDAT")), dtype=np.uint8).reshape(h, 1 + w * c)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, c)


def test_png_round_trip(rt, tmp_path):
    rng = np.random.default_rng(1)
    for shape in [(7, 5, 4), (3, 9, 3), (1, 1, 4)]:
        img = rng.integers(0, 256, size=shape, dtype=np.uint8)
        rt.write_png(tmp_path / "a.png", img)
        assert np.array_equal(read_png(tmp_path / "a.png"), img)
    with pytest.raises(ValueError):
        rt.write_png(tmp_path / "b.png", np.zeros((4, 4), dtype=np.uint8))


def test_app_defaults_are_the_reference_ones(rt):
    app = rt.App(800, 800, renderer=fake_renderer([]))
    assert app.scene == rt.default_scene()
    cam = app.camera
    assert (cam.eye, cam.yaw, cam.pitch, cam.focal_length) == ((4.0, 1.2, 3.0), -135.0, -12.0, 2.0)   # main.py:103-109
    assert (app.controller.last_x, app.controller.last_y) == (400, 400)                              # main.py:110-111
    assert app.title == "Ray Tracer com Python e GLFW - FPS: 0"                                      # main.py:247


def test_fps_counter_refreshes_once_a_second(rt):
    clock = FakeClock()
    app = rt.App(64, 48, renderer=fake_renderer([]), clock=clock)
    # main.py:182-193: fps = frames counted since the last refresh, refreshed when >= 1 s passed
    for _ in range(7):
        clock.t += 0.125
        app.update_timing()
        assert app.fps == 0
    assert app.delta_time == 0.125
    clock.t += 0.125                     # 1.0 s since start: the 8th frame triggers the refresh
    app.update_timing()
    assert app.fps == 8 and app.frame_count == 0
    for _ in range(4):
        clock.t += 0.25
        app.update_timing()
    assert app.fps == 4


def test_frame_loop_order_and_input_routing(rt, tmp_path):
    calls, clock = [], FakeClock()
    app = rt.App(32, 24, renderer=fake_renderer(calls), clock=clock, render_opts=dict(max_depth=8))
    script = [
        ((), []),
        (("W",), [("cursor", 100.0, 100.0), ("cursor", 110.0, 95.0)]),       # first cursor event only latches
        ((), [("scroll", 0.0, 3.0), ("resize", 48, 40), ("resize", 0, 0)]),
        ((), [("key", "ESCAPE"), ("cursor", 500.0, 500.0)]),                 # released mouse: cursor ignored
        ((), [("mouse_button",), ("cursor", 0.0, 0.0), ("cursor", 10.0, 0.0)]),
    ]

    class StepClock(FakeClock):
        def __call__(self):
            self.t += 0.5
            return self.t
    app.clock = StepClock()
    app._t0 = 100.0
    last = app.run(frames=5, script=script, png_pattern=str(tmp_path / "f_%02d.png"))
    assert len(calls) == 5
    assert calls[0][5:7] == (32, 24) and calls[0][7] == {"max_depth": 8}
    # frame 1: yaw += 10 * 0.1, pitch += 5 * 0.1 (main.py:429-435), then W moves along the view's
    # horizontal projection by CAMERA_SPEED * delta_time (main.py:311-327)
    assert calls[1][2] == pytest.approx(-134.0) and calls[1][3] == pytest.approx(-11.5)
    assert calls[1][1] != calls[0][1] and calls[1][1][1] == pytest.approx(1.2)
    assert calls[2][4] == pytest.approx(2.6) and calls[2][5:7] == (48, 40)       # zoom main.py:406, resize :371
    assert calls[3][2] == calls[2][2]                                               # mouse released
    assert calls[4][2] == pytest.approx(calls[3][2] + 1.0)                          # re-captured, first event latches
    assert last.shape == (40, 48, 4)
    assert np.array_equal(read_png(tmp_path / "f_04.png"), last)
    assert read_png(tmp_path / "f_00.png").shape == (24, 32, 4)


def test_run_without_window_needs_a_frame_count(rt):
    app = rt.App(8, 8, renderer=fake_renderer([]))
    with pytest.raises(RuntimeError, match="headless"):
        app.run()


def test_unknown_event_is_an_error(rt):
    app = rt.App(8, 8, renderer=fake_renderer([]))
    with pytest.raises(ValueError):
        app.dispatch(("joystick", 1))


@pytest.mark.gpu
def test_app_renders_the_reference_frame(gpu_rt, tmp_path):
    meta, arr = load_golden("default_800")
    app = gpu_rt.App(800, 800)
    frame = app.run(frames=1, png_pattern=str(tmp_path / "frame_%04d.png"))
    assert np.array_equal(frame[::-1, :, :3], arr["rgb8"])    # golden rows are bottom-up
    assert (frame[..., 3] == 255).all()
    assert np.array_equal(read_png(tmp_path / "frame_0000.png"), frame)
