"""Camera state and the camera -> uniform computation of the reference host.

`camera_to_world` restates the per-frame matrix block of the reference's App.render
(main.py:253-274): yaw/pitch -> view direction with float64 trig stored into a float32
vector, then float32 normalise / cross products, laid out as rows [x | y | view | eye].
`Camera` carries the state the reference keeps on its App object (main.py:102-109) and
`CameraController` restates its input handling (main.py:305-337, :404-408, :416-438) as a
pure state machine with no window attached.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

# main.py:61-65
CAMERA_SPEED = 3.0
MOUSE_SENSITIVITY = 0.1
ZOOM_SPEED = 0.2
MIN_FOCAL_LENGTH = 0.5
MAX_FOCAL_LENGTH = 10.0

DEFAULT_LIGHT_POS = (-2.0, 4.0, 1.0)  # main.py:278


def _normalize(v):
    """main.py:87-92."""
    n = np.linalg.norm(v)
    if n == 0:
        return v
    return v / n


def view_direction(yaw: float, pitch: float) -> np.ndarray:
    """main.py:255-259: float64 trig, stored into a float32 vector, then normalised."""
    d = np.empty(3, dtype=np.float32)
    d[0] = np.cos(np.radians(yaw)) * np.cos(np.radians(pitch))
    d[1] = np.sin(np.radians(pitch))
    d[2] = np.sin(np.radians(yaw)) * np.cos(np.radians(pitch))
    return _normalize(d)


def camera_to_world(eye, yaw: float, pitch: float, up=(0.0, 1.0, 0.0)) -> np.ndarray:
    """float32 4x4, rows [x_axis | y_axis | view_dir | eye] (main.py:262-274).  The shader
    reads its columns as right, up, forward, position (fragment_shader.glsl:656-659)."""
    view = view_direction(yaw, pitch)
    upv = np.asarray(up, dtype=np.float32)
    z_axis = -view
    x_axis = _normalize(np.cross(upv, z_axis))
    y_axis = _normalize(np.cross(z_axis, x_axis))
    e = np.asarray(eye, dtype=np.float32)
    return np.array([[x_axis[0], y_axis[0], view[0], e[0]],
                     [x_axis[1], y_axis[1], view[1], e[1]],
                     [x_axis[2], y_axis[2], view[2], e[2]],
                     [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)


@dataclass
class Camera:
    """Camera uniforms the reference host owns (main.py:102-109, :278)."""
    eye: tuple = (4.0, 1.2, 3.0)
    yaw: float = -135.0
    pitch: float = -12.0
    focal_length: float = 2.0
    up: tuple = (0.0, 1.0, 0.0)
    light_pos: tuple = DEFAULT_LIGHT_POS

    @classmethod
    def coerce(cls, cam) -> "Camera":
        if isinstance(cam, Camera):
            return cam
        if isinstance(cam, dict):
            unknown = set(cam) - {"eye", "yaw", "pitch", "focal_length", "up", "light_pos"}
            if unknown:
                raise ValueError(f"unknown camera keys: {sorted(unknown)}")
            return cls(**cam)
        raise TypeError("camera must be a Camera or a dict with eye/yaw/pitch/focal_length")

    def matrix(self) -> np.ndarray:
        return camera_to_world(self.eye, self.yaw, self.pitch, self.up)


@dataclass
class CameraController:
    """The reference's fly camera without the window: feed it key states, mouse positions and
    scroll offsets, read back a Camera.  Mirrors process_input (main.py:305-337),
    cursor_pos_callback (main.py:416-438), scroll_callback (main.py:404-408) and the mouse
    capture toggles (main.py:339-348, :380-396)."""
    camera: Camera = field(default_factory=Camera)
    last_x: float = 0.0
    last_y: float = 0.0
    first_mouse: bool = True
    mouse_captured: bool = True

    def process_keys(self, pressed, delta_time: float, focused: bool = True) -> None:
        """pressed: any container of 'W','A','S','D','SPACE','LSHIFT'."""
        if not focused:
            return
        speed = CAMERA_SPEED * delta_time
        # the camera's eye is the single state, as App.eye in the reference (main.py:103, :326-337):
        # read it every frame, so a position set from outside is not overwritten
        eye = np.array(self.camera.eye, dtype=np.float32)
        view = view_direction(self.camera.yaw, self.camera.pitch)
        up = np.asarray(self.camera.up, dtype=np.float32)
        forward = _normalize(np.array([view[0], 0, view[2]]))
        right = _normalize(np.cross(forward, up))
        if "W" in pressed:
            eye += forward * speed
        if "S" in pressed:
            eye -= forward * speed
        if "A" in pressed:
            eye -= right * speed
        if "D" in pressed:
            eye += right * speed
        if "SPACE" in pressed:
            eye += up * speed
        if "LSHIFT" in pressed:
            eye -= up * speed
        self.camera.eye = tuple(float(v) for v in eye)

    def capture_mouse(self, should_capture: bool) -> None:
        self.mouse_captured = should_capture
        if should_capture:
            self.first_mouse = True

    def key_escape(self) -> None:
        self.capture_mouse(not self.mouse_captured)

    def mouse_button(self) -> None:
        if not self.mouse_captured:
            self.capture_mouse(True)

    def cursor_pos(self, xpos: float, ypos: float) -> None:
        if not self.mouse_captured:
            return
        if self.first_mouse:
            self.last_x, self.last_y = xpos, ypos
            self.first_mouse = False
        xoffset = (xpos - self.last_x) * MOUSE_SENSITIVITY
        yoffset = (self.last_y - ypos) * MOUSE_SENSITIVITY
        self.last_x, self.last_y = xpos, ypos
        self.camera.yaw += xoffset
        self.camera.pitch = float(np.clip(self.camera.pitch + yoffset, -89.0, 89.0))

    def scroll(self, y_offset: float) -> None:
        f = self.camera.focal_length + y_offset * ZOOM_SPEED
        self.camera.focal_length = float(np.clip(f, MIN_FOCAL_LENGTH, MAX_FOCAL_LENGTH))
