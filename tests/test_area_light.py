"""Area light (G x G light samples per shaded hit): not a reference feature, but pinned by the
reference all the same -- ray paths, sky and ambient terms do not depend on the light and the
colour is linear in the light term, hence an area-light frame is, up to rounding, the mean of the
G*G point-light frames (before the framebuffer clamp).  tests/golden/light*_160 hold that mean of
frames the reference shader itself rendered on llvmpipe with u_light_pos at the sample positions
(oracle/make_golden.py: light_case); the oracle must agree with it to a few units in the last
place, and the HIP kernels equal the oracle bit for bit (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from conftest import camera_for, load_golden

# |area-light frame - mean of the reference's point-light frames|: both sum G*G light terms of
# magnitude <= ~2 in binary32, in different orders -> a few ulp of the largest term
LIGHT_ABS_TOL = 1.5e-6


@pytest.mark.parametrize("name", ["light2_160", "light3_160"])
def test_oracle_area_light_equals_mean_of_reference_frames(orc, rt, name):
    meta, arr = load_golden(name)
    cam = camera_for(meta)
    o = orc.render(rt.default_scene(), cam.matrix(), cam.focal_length, meta["w"], meta["h"], light=cam.light_pos,
                   light_grid=meta["light_grid"], light_size=meta["light_size"], max_depth=meta["max_depth"],
                   stack_size=meta["stack_size"], shadow_hops=meta["shadow_hops"], want_f32=True)["f32"]
    ref = arr["rgb32f"]
    assert np.isfinite(ref).all() and ref.max() > 0.5
    assert np.abs(o[..., :3].astype(np.float64) - ref).max() <= LIGHT_ABS_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["light2_160", "light3_160"])
def test_gpu_area_light_equals_mean_of_reference_frames(gpu_rt, name):
    meta, arr = load_golden(name)
    cam = camera_for(meta)
    f32 = gpu_rt.render_rgba(gpu_rt.default_scene(), cam, meta["w"], meta["h"], bottom_up=True, out_f32=True,
                             light_grid=meta["light_grid"], light_size=meta["light_size"], max_depth=meta["max_depth"],
                             stack_size=meta["stack_size"], shadow_hops=meta["shadow_hops"])
    assert np.abs(f32[..., :3].astype(np.float64) - arr["rgb32f"]).max() <= LIGHT_ABS_TOL


def _light_offsets(G, size):
    return [np.float32(np.float32((i + 0.5) / G - 0.5) * np.float32(size)) for i in range(G)]


@pytest.mark.parametrize("G,size", [(2, 1.5), (3, 0.6)])
def test_oracle_area_light_is_mean_of_point_lights(orc, rt, G, size):
    scene, cam = rt.default_scene(), rt.default_camera()
    w, h = 96, 72
    kw = dict(max_depth=4, stack_size=3, shadow_hops=2, want_f32=True)
    area = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, light_grid=G,
                      light_size=size, **kw)["f32"]
    offs = _light_offsets(G, size)
    lx, ly, lz = (np.float32(v) for v in cam.light_pos)
    acc = np.zeros((h, w, 4), dtype=np.float64)
    for j in range(G):
        for i in range(G):
            light = (float(np.float32(lx + offs[i])), float(ly), float(np.float32(lz + offs[j])))
            acc += orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=light, **kw)["f32"]
    mean = acc / (G * G)
    assert np.isfinite(area).all()
    np.testing.assert_allclose(area[..., :3], mean[..., :3], rtol=2e-5, atol=2e-6)
    # and it is not the point-light frame: the penumbra differs
    point = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, **kw)["f32"]
    assert np.abs(area - point).max() > 1e-2


def test_oracle_light_grid_one_is_the_point_light(orc, rt):
    scene, cam = rt.default_scene(), rt.default_camera()
    a = orc.render(scene, cam.matrix(), cam.focal_length, 64, 48, want_f32=True)["f32"]
    for G, size in ((0, 0.0), (1, 0.0), (1, 2.0)):
        b = orc.render(scene, cam.matrix(), cam.focal_length, 64, 48, want_f32=True, light_grid=G, light_size=size)["f32"]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    with pytest.raises(ValueError):
        orc.render(scene, cam.matrix(), cam.focal_length, 8, 8, light_grid=9, light_size=1.0)
