"""RT_FLAG_NEWTON_POLISH: Newton steps on the torus' implicit function after the reference's
closed-form (Ferrari) solve, and rejection of roots that do not lie on the torus.  Not the
reference's result (its solve ends at fragment_shader.glsl:345-352), so the bar is a float64
solve of the same quartic: tests/golden/tori64_f64_240.npz (oracle/make_torus_f64.py, numpy
companion-matrix roots for every primary ray of the 64-tori stress scene).  With the flag the
hit distances must be closer to it, the torus hit mask must move towards it, and the HIP
kernels must equal the CPU oracle's polished result bit for bit."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

W, H = 240, 135


def _stats(cap, n, t64, id64):
    kid, kt = cap[..., 0].astype(int), cap[..., 1].astype(np.float64)
    torus_k, torus_e = (kid >= 0) & (kid < n), id64 >= 0
    same = torus_k & torus_e & (kid == id64)
    rel = np.abs(kt[same] - t64[same]) / t64[same]
    mask_diff = int((torus_k != torus_e).sum() + (torus_k & torus_e & (kid != id64)).sum())
    return dict(median=float(np.median(rel)), p99=float(np.percentile(rel, 99)), max=float(rel.max()),
                kernel_only=int((torus_k & ~torus_e).sum()), mask_diff=mask_diff, common=int(same.sum()))


def _check_better(plain, polished):
    # distances: two orders of magnitude at the median and at the 99th percentile, and the worst
    # pixel (a root the closed form attributes to the wrong crossing) is not made worse
    assert polished["median"] < 1e-6 < 1e-4 < plain["median"]
    assert polished["p99"] < 1e-5 < 1e-3 < plain["p99"]
    assert polished["max"] <= plain["max"]
    # mask: far fewer hits that float64 does not have, and fewer differing pixels overall
    assert polished["kernel_only"] * 5 < plain["kernel_only"]
    assert polished["mask_diff"] < plain["mask_diff"]


@pytest.fixture(scope="module")
def f64():
    g = np.load(os.path.join(GOLDEN, f"tori64_f64_{W}.npz"))
    return g["t"], g["id"].astype(int)


def test_oracle_polish_moves_towards_float64(orc, rt, f64):
    scene, cam = rt.tori_stress_scene(), rt.tori_stress_camera()
    caps = [orc.capture(scene, cam.matrix(), cam.focal_length, W, H, 0, light=cam.light_pos, max_depth=8, newton_polish=p)
            for p in (False, True)]
    plain, polished = (_stats(c, len(scene), *f64) for c in caps)
    _check_better(plain, polished)


def test_oracle_polish_leaves_scenes_without_tori_alone(orc, rt):
    scene, cam = rt.spheres_deep_scene(12, seed=7), rt.default_camera()
    a = orc.render(scene, cam.matrix(), cam.focal_length, 96, 72, want_f32=True)["f32"]
    b = orc.render(scene, cam.matrix(), cam.focal_length, 96, 72, want_f32=True, newton_polish=True)["f32"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.gpu
def test_gpu_polish_equals_oracle_and_moves_towards_float64(gpu_rt, orc, f64):
    scene, cam = gpu_rt.tori_stress_scene(), gpu_rt.tori_stress_camera()
    n = len(scene)
    caps = [gpu_rt.capture_segment(scene, cam, W, H, 0, max_depth=8, newton_polish=p) for p in (False, True)]
    plain, polished = (_stats(c, n, *f64) for c in caps)
    _check_better(plain, polished)
    ref = orc.capture(scene, cam.matrix(), cam.focal_length, W, H, 0, light=cam.light_pos, max_depth=8, newton_polish=True)
    assert np.array_equal(caps[1][..., 0], ref[..., 0])
    assert np.array_equal(caps[1][..., 1].view(np.uint32), ref[..., 1].view(np.uint32))
    # whole frames (secondary rays and shadow rays polish too): every kernel family against the oracle
    for sc, cm, o in ((scene, cam, dict(max_depth=8, stack_size=3)), (scene, cam, dict(max_depth=6, stack_size=6)),
                      (gpu_rt.default_scene(), gpu_rt.default_camera(), dict(max_depth=8, stack_size=3))):
        f32 = gpu_rt.render_rgba(sc, cm, 200, 112, bottom_up=True, out_f32=True, newton_polish=True, **o)
        want = orc.render(sc, cm.matrix(), cm.focal_length, 200, 112, light=cm.light_pos, want_f32=True, newton_polish=True, **o)["f32"]
        same = (f32.view(np.uint32) == want.view(np.uint32)) | (np.isnan(f32) & np.isnan(want))
        assert same.all(), f"{int((~same).any(axis=2).sum())} pixels differ"
    # and the flag changes the picture (it is not a no-op): the default scene's torus
    a = gpu_rt.render_rgba(gpu_rt.default_scene(), gpu_rt.default_camera(), 200, 112, bottom_up=True, out_f32=True, max_depth=8)
    b = gpu_rt.render_rgba(gpu_rt.default_scene(), gpu_rt.default_camera(), 200, 112, bottom_up=True, out_f32=True, max_depth=8, newton_polish=True)
    assert not np.array_equal(a, b)
