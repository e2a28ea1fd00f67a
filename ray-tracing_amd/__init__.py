"""MI355X-native renderer for the reference's per-pixel ray-tracing path.

The directory name carries a hyphen; import it with
    importlib.import_module("ray-tracing_amd")
or through the `rt355` alias module at the repository root.

Public surface (mirrors what the reference host owns, main.py:96-132 and :244-303):
    render(scene, camera, w, h, ...) -> uint8 RGB image
    Camera, CameraController, camera_to_world
    App(width, height): the reference's application object; its window only blits
    default_scene, default_camera and the synthetic stress scenes
"""
from .camera import Camera, CameraController, camera_to_world, DEFAULT_LIGHT_POS
from .scenes import (default_scene, default_camera, tori_stress_scene, tori_stress_camera,
                     spheres_deep_scene, spheres_deep_camera, mixed_scene, mixed_camera, random_scene,
                     random_large_scene, sphere_row_scene, cluster_scene)

from . import api
from .api import (render, render_rgba, set_scene, init, shutdown, timing, count_events, capture_segment, selftest_math, stage_scene_host,
                  frame_flops, event_table, load_library, RtError, pinned_empty, init_multi)

from .app import App, write_png, open_window

__all__ = ["App", "write_png", "open_window", "api", "render", "render_rgba", "set_scene", "init", "shutdown", "timing", "count_events",
           "capture_segment", "selftest_math", "stage_scene_host", "frame_flops", "event_table", "load_library", "RtError", "pinned_empty", "init_multi",
           "Camera", "CameraController", "camera_to_world", "DEFAULT_LIGHT_POS",
           "default_scene", "default_camera", "tori_stress_scene", "tori_stress_camera",
           "spheres_deep_scene", "spheres_deep_camera", "mixed_scene", "mixed_camera", "random_scene",
           "random_large_scene", "sphere_row_scene", "cluster_scene"]
