"""The reference's application object without its ray tracer.

The reference's `App` (main.py:95-438) owns a GLFW window, the camera/input state, the
scene list, an FPS counter, and draws every frame by running the fragment shader.  Here the
frame comes from `render(scene, camera, w, h)` (the HIP path); what is left of the window
is a blit of the returned RGBA8 buffer.  `App` keeps the reference's method names and
per-frame order (update_timing, process_input, render, poll: main.py:174-178) so that a
user of the reference finds the same object, and it runs in two ways:

* with a window backend (`GlfwBlitWindow`, needs the `glfw` and `PyOpenGL` packages and a
  display): the interactive fly-through of the reference;
* headless (the default when no backend is available): a scripted sequence of input events
  drives the same state machine and the frames are returned or written as PNG files.

Nothing here computes pixels: without the HIP library `render` raises.
"""
from __future__ import annotations

import struct
import time
import zlib

import numpy as np

from .camera import Camera, CameraController
from .scenes import default_scene

WINDOW_TITLE = "Ray Tracer com Python e GLFW"   # main.py:144


def write_png(path, image) -> None:
    """Writes an (h, w, 3|4) uint8 array (row 0 = top) as an 8-bit RGB/RGBA PNG."""
    img = np.ascontiguousarray(image)
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] not in (3, 4):
        raise ValueError("image must be (h, w, 3) or (h, w, 4) uint8")
    h, w, c = img.shape
    raw = np.empty((h, 1 + w * c), dtype=np.uint8)
    raw[:, 0] = 0                                   # filter type 0 (none) on every scanline
    raw[:, 1:] = img.reshape(h, w * c)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    ihdr = struct.pack(">IIBBBBB", w, h, 8, 6 if c == 4 else 2, 0, 0, 0)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)) +
                chunk(b"IEND", b""))


_frame_buffers = {}


def _gpu_renderer(scene, camera, w, h, **opts):
    """The HIP path into a page-locked frame buffer that is reused from frame to frame (like the
    reference's back buffer, main.py:303, the previous frame is gone once the next one is drawn)."""
    from . import api
    key = (int(h), int(w), bool(opts.get("out_f32")))
    if key not in _frame_buffers and not opts.get("nparts", 1) > 1:
        for old in _frame_buffers.values():         # a resized window: one buffer at a time
            api.pinned_release(old)
        _frame_buffers.clear()
        _frame_buffers[key] = api.pinned_empty((int(h), int(w), 4), np.float32 if key[2] else np.uint8)
    buf = _frame_buffers.get(key)
    if buf is not None and not api.is_pinned(buf):     # the library was shut down meanwhile: the memory is gone
        del _frame_buffers[key]
        return _gpu_renderer(scene, camera, w, h, **opts)
    return api.render_rgba(scene, camera, w, h, out=buf, **opts)


class App:
    """Scene + camera state + frame loop of the reference (main.py:95-193, :244-438).

    renderer(scene, camera, w, h, **render_opts) -> (h, w, 4) uint8, row 0 = top; defaults to
    the HIP path.  clock() -> seconds; defaults to time.perf_counter (the reference uses
    glfw.get_time, main.py:184).  window: a backend with the GlfwBlitWindow interface or None.
    """

    def __init__(self, width, height, *, renderer=None, clock=None, window=None, render_opts=None):
        self.width, self.height = int(width), int(height)
        self.scene = default_scene()                                   # main.py:125-132
        self.controller = CameraController(Camera(), last_x=width / 2, last_y=height / 2)   # main.py:103-113
        self.renderer = renderer or _gpu_renderer
        self.render_opts = dict(render_opts or {})
        self.clock = clock or time.perf_counter
        self.window = window
        self.last_frame_time = 0.0                                     # main.py:116-122
        self.delta_time = 0.0
        self.frame_count = 0
        self.last_fps_time = 0.0
        self.fps = 0
        self.frame = None
        self._t0 = self.clock()

    # ---- state the reference keeps directly on the App object
    @property
    def camera(self) -> Camera:
        return self.controller.camera

    @property
    def title(self) -> str:
        return f"{WINDOW_TITLE} - FPS: {self.fps}"                     # main.py:247

    # ---- per-frame steps, in the reference's order (main.py:174-178)
    def update_timing(self) -> None:
        """main.py:182-193: delta time, and an FPS value refreshed once a second."""
        now = self.clock() - self._t0
        self.delta_time = now - self.last_frame_time
        self.last_frame_time = now
        self.frame_count += 1
        if now - self.last_fps_time >= 1.0:
            self.fps = self.frame_count
            self.frame_count = 0
            self.last_fps_time = now

    def process_input(self, pressed=(), focused=True) -> None:
        """main.py:305-337 with the key states passed in."""
        self.controller.process_keys(pressed, self.delta_time, focused)

    def render(self):
        """main.py:244-303: one frame for the current camera at the current size."""
        self.frame = self.renderer(self.scene, self.camera, self.width, self.height, **self.render_opts)
        if self.window is not None:
            self.window.set_title(self.title)
            self.window.blit(self.frame)
        return self.frame

    # ---- event callbacks (main.py:369-438)
    def framebuffer_size_callback(self, width, height) -> None:
        if width > 0 and height > 0:                                   # a minimised window reports 0 x 0
            self.width, self.height = int(width), int(height)

    def key_callback(self, key, action="PRESS") -> None:
        if key == "ESCAPE" and action == "PRESS":
            self.controller.key_escape()
            if self.window is not None:
                self.window.capture_mouse(self.controller.mouse_captured)

    def mouse_button_callback(self, action="PRESS") -> None:
        if action == "PRESS":
            self.controller.mouse_button()
            if self.window is not None:
                self.window.capture_mouse(self.controller.mouse_captured)

    def scroll_callback(self, x_offset, y_offset) -> None:
        self.controller.scroll(y_offset)

    def cursor_pos_callback(self, xpos, ypos) -> None:
        self.controller.cursor_pos(xpos, ypos)

    def dispatch(self, event) -> None:
        """One scripted or polled event: ("key", name[, action]), ("mouse_button",),
        ("scroll", dx, dy), ("cursor", x, y), ("resize", w, h)."""
        kind, *a = event
        if kind == "key":
            self.key_callback(*a)
        elif kind == "mouse_button":
            self.mouse_button_callback(*a)
        elif kind == "scroll":
            self.scroll_callback(*a)
        elif kind == "cursor":
            self.cursor_pos_callback(*a)
        elif kind == "resize":
            self.framebuffer_size_callback(*a)
        else:
            raise ValueError(f"unknown event {kind!r}")

    # ---- main loop
    def run(self, frames=None, script=None, png_pattern=None):
        """With a window: the reference's loop until the window closes (main.py:163-180).
        Headless: `frames` iterations; script[i] = (pressed_keys, [events]) is applied before
        frame i; png_pattern (e.g. "out/frame_%04d.png") writes every frame.  Returns the last
        frame."""
        if self.window is None and frames is None:
            raise RuntimeError("no window backend (glfw/PyOpenGL or a display is missing): "
                               "run headless with run(frames=N[, script, png_pattern])")
        i = 0
        try:
            while frames is None or i < frames:
                if self.window is not None and self.window.should_close():
                    break
                pressed, events = (), ()
                if script is not None and i < len(script):
                    pressed, events = script[i]
                # the reference's iteration (main.py:174-178): update_timing, process_input, render,
                # poll_events (whose callbacks change the state the NEXT frame sees); scripted
                # events stand for what was polled at the end of the iteration before
                for ev in events:
                    self.dispatch(ev)
                self.update_timing()
                focused = True
                if self.window is not None:
                    pressed = set(pressed) | self.window.pressed_keys()     # glfw.get_key inside process_input
                    focused = self.window.focused()
                self.process_input(pressed, focused)
                self.render()
                if png_pattern:
                    write_png(png_pattern % i, self.frame)
                if self.window is not None:
                    for ev in self.window.poll():
                        self.dispatch(ev)
                i += 1
        finally:
            self.cleanup()
        return self.frame

    def cleanup(self) -> None:
        """main.py:350-355."""
        if self.window is not None:
            self.window.close()
            self.window = None


class GlfwBlitWindow:
    """Optional window backend: a GLFW window (OpenGL 3.3 core, v-sync off like main.py:138-150)
    that uploads each frame to a texture and draws it with one full-screen triangle.  Imports
    glfw / PyOpenGL on construction, so the rest of the module works without them."""

    _VS = """#version 330 core
out vec2 uv;
void main() {
    vec2 p = vec2((gl_VertexID << 1) & 2, gl_VertexID & 2);
    uv = vec2(p.x, 1.0 - p.y);
    gl_Position = vec4(p * 2.0 - 1.0, 0.0, 1.0);
}"""
    _FS = """#version 330 core
in vec2 uv;
out vec4 colour;
uniform sampler2D frame;
void main() { colour = texture(frame, uv); }"""
    _KEYS = ("W", "A", "S", "D", "SPACE", "LEFT_SHIFT")

    def __init__(self, width, height, title=WINDOW_TITLE):
        import glfw
        from OpenGL import GL
        self.glfw, self.GL = glfw, GL
        if not glfw.init():
            raise RuntimeError("glfw.init() failed")
        glfw.window_hint(glfw.CONTEXT_VERSION_MAJOR, 3)
        glfw.window_hint(glfw.CONTEXT_VERSION_MINOR, 3)
        glfw.window_hint(glfw.OPENGL_PROFILE, glfw.OPENGL_CORE_PROFILE)
        glfw.window_hint(glfw.OPENGL_FORWARD_COMPAT, GL.GL_TRUE)
        self.win = glfw.create_window(int(width), int(height), title, None, None)
        if not self.win:
            glfw.terminate()
            raise RuntimeError("could not create a GLFW window")
        glfw.make_context_current(self.win)
        glfw.swap_interval(0)
        self._events = []
        glfw.set_framebuffer_size_callback(self.win, lambda w, x, y: self._events.append(("resize", x, y)))
        glfw.set_key_callback(self.win, lambda w, key, sc, action, mods: self._events.append(
            ("key", "ESCAPE", "PRESS")) if key == glfw.KEY_ESCAPE and action == glfw.PRESS else None)
        glfw.set_cursor_pos_callback(self.win, lambda w, x, y: self._events.append(("cursor", x, y)))
        glfw.set_scroll_callback(self.win, lambda w, dx, dy: self._events.append(("scroll", dx, dy)))
        glfw.set_mouse_button_callback(self.win, lambda w, b, action, mods: self._events.append(
            ("mouse_button", "PRESS")) if action == glfw.PRESS else None)
        self.capture_mouse(True)
        from OpenGL.GL import shaders
        self.prog = shaders.compileProgram(shaders.compileShader(self._VS, GL.GL_VERTEX_SHADER),
                                           shaders.compileShader(self._FS, GL.GL_FRAGMENT_SHADER))
        self.vao = GL.glGenVertexArrays(1)
        self.tex = GL.glGenTextures(1)
        GL.glBindTexture(GL.GL_TEXTURE_2D, self.tex)
        GL.glTexParameteri(GL.GL_TEXTURE_2D, GL.GL_TEXTURE_MIN_FILTER, GL.GL_NEAREST)
        GL.glTexParameteri(GL.GL_TEXTURE_2D, GL.GL_TEXTURE_MAG_FILTER, GL.GL_NEAREST)
        self._tex_size = None

    def blit(self, frame) -> None:
        GL, glfw = self.GL, self.glfw
        h, w = frame.shape[:2]
        GL.glViewport(0, 0, *glfw.get_framebuffer_size(self.win))
        GL.glBindTexture(GL.GL_TEXTURE_2D, self.tex)
        GL.glPixelStorei(GL.GL_UNPACK_ALIGNMENT, 1)
        if self._tex_size != (w, h):
            GL.glTexImage2D(GL.GL_TEXTURE_2D, 0, GL.GL_RGBA8, w, h, 0, GL.GL_RGBA, GL.GL_UNSIGNED_BYTE, frame)
            self._tex_size = (w, h)
        else:
            GL.glTexSubImage2D(GL.GL_TEXTURE_2D, 0, 0, 0, w, h, GL.GL_RGBA, GL.GL_UNSIGNED_BYTE, frame)
        GL.glUseProgram(self.prog)
        GL.glBindVertexArray(self.vao)
        GL.glDrawArrays(GL.GL_TRIANGLES, 0, 3)
        glfw.swap_buffers(self.win)

    def poll(self):
        self.glfw.poll_events()
        ev, self._events = self._events, []
        return ev

    def pressed_keys(self):
        g = self.glfw
        names = {"LEFT_SHIFT": "LSHIFT"}
        return {names.get(k, k) for k in self._KEYS if g.get_key(self.win, getattr(g, "KEY_" + k)) == g.PRESS}

    def focused(self) -> bool:
        return bool(self.glfw.get_window_attrib(self.win, self.glfw.FOCUSED))

    def should_close(self) -> bool:
        return bool(self.glfw.window_should_close(self.win))

    def set_title(self, title) -> None:
        self.glfw.set_window_title(self.win, title)

    def capture_mouse(self, captured) -> None:
        g = self.glfw
        g.set_input_mode(self.win, g.CURSOR, g.CURSOR_DISABLED if captured else g.CURSOR_NORMAL)

    def close(self) -> None:
        self.glfw.terminate()


def open_window(width, height):
    """A GlfwBlitWindow, or None when glfw / PyOpenGL / a display is not available."""
    try:
        return GlfwBlitWindow(width, height)
    except Exception:       # ImportError, no display, no GL 3.3 context
        return None


def main(argv=None):
    """`python -m rt355_app`-style entry: interactive when a window can be opened, otherwise one
    PNG of the default view (the reference starts App(800, 800), main.py:444)."""
    import argparse
    ap = argparse.ArgumentParser(description="Ray tracer viewer: blits frames rendered by the HIP path")
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--headless", action="store_true", help="never open a window")
    ap.add_argument("--frames", type=int, default=1, help="headless: frames to render")
    ap.add_argument("--png", default="frame_%04d.png", help="headless: output pattern")
    # the shader's constants (fragment_shader.glsl:653, :547, :453) and the sampling extensions
    ap.add_argument("--max-depth", type=int, default=4)
    ap.add_argument("--stack-size", type=int, default=3)
    ap.add_argument("--shadow-hops", type=int, default=2)
    ap.add_argument("--spp-grid", type=int, default=1, help="N: N x N stratified samples per pixel")
    ap.add_argument("--light-grid", type=int, default=1, help="G: G x G samples of a square area light")
    ap.add_argument("--light-size", type=float, default=0.0, help="edge of the area light")
    args = ap.parse_args(argv)
    window = None if args.headless else open_window(args.width, args.height)
    opts = dict(max_depth=args.max_depth, stack_size=args.stack_size, shadow_hops=args.shadow_hops,
                spp_grid=args.spp_grid, light_grid=args.light_grid, light_size=args.light_size)
    app = App(args.width, args.height, window=window, render_opts=opts)
    if window is None:
        app.run(frames=args.frames, png_pattern=args.png)
        print(f"wrote {args.frames} frame(s) as {args.png}; {app.title}")
    else:
        print("WASD / Space / Shift move, mouse looks, scroll zooms, ESC releases the mouse")   # main.py:168-171
        app.run()
