"""Host-side scene staging (what rt_set_scene uploads), checked without a GPU through the
host-only entry point: per-object invariants against the oracle's, and the invariants the
sphere hierarchy must satisfy for the accelerated trace to be exact (every object inside its
leaf node, every leaf inside its top node, everything inside the scene sphere, slot ids a
permutation, empty slots impossible to hit)."""
import numpy as np
import pytest


def bound_records(blob, count, off):
    """Pair-interleaved {cx0 cx1 cy0 cy1}{cz0 cz1 r0 r1} records -> (count, 4) array."""
    rec = blob[off:off + 8 * ((count + 1) // 2)].reshape(-1, 8)
    out = np.empty((2 * rec.shape[0], 4), dtype=np.float32)
    out[0::2] = rec[:, [0, 2, 4, 6]]
    out[1::2] = rec[:, [1, 3, 5, 7]]
    return out[:count], out


def test_small_scene_has_no_hierarchy_and_the_shader_invariants(rt):
    scene = rt.default_scene()
    L, blob, sphere = rt.stage_scene_host(scene)
    assert (L["n"], L["ntori"], L["n1"], L["n2"]) == (5, 1, 0, 0)
    rec, padded = bound_records(blob, 5, 0)
    for i, o in enumerate(scene):
        assert rec[i, :3].tolist() == [np.float32(c) for c in o["center"]]
        br = np.float32(o["radius"]) if o["type"] == 1 else np.float32(o["major_radius"]) + np.float32(o["minor_radius"])
        assert rec[i, 3] == br * br                                   # fragment_shader.glsl:391-402
    assert padded[5, 3] == -np.inf                                    # odd count: the pad can never pass
    mat = blob[L["off_material"]:L["off_material"] + 40].reshape(5, 8)
    ior = np.float32(1.5)
    q = (np.float32(1.0) - ior) / (np.float32(1.0) + ior)
    assert mat[0, 6] == q * q and mat[0, 7] == np.float32(1.0) / ior  # Fresnel constants, :525-528
    kind = blob[L["off_kind"]:L["off_kind"] + 5].view(np.int32)
    assert kind.tolist() == [-1, -1, -1, -1, 0]
    tor = blob[L["off_torus"]:L["off_torus"] + 16]
    # normal (0,1,0): |w.y| > 0.99 selects the x axis as helper (:315-317)
    assert tor[3:6].tolist() == [0.0, 1.0, 0.0] and tor[9] == np.float32(0.8)
    assert tor[10] == np.float32(0.8) * np.float32(0.8) and tor[11] == np.float32(0.2) * np.float32(0.2)


@pytest.mark.parametrize("which,n", [("mixed", 1024), ("mixed", 1000), ("mixed", 41), ("spheres", 256), ("tori", 64),
                                     ("large", 0), ("large", 5), ("row", 120)])
def test_hierarchy_invariants(rt, which, n):
    if which == "mixed":
        scene = rt.mixed_scene(n)
    elif which == "spheres":
        scene = rt.spheres_deep_scene(n)
    elif which == "tori":
        scene = rt.tori_stress_scene(n)
    elif which == "row":
        scene = rt.sphere_row_scene(n)
    else:
        scene = rt.random_large_scene(n)[0]
    n = len(scene)
    L, blob, (sx, sy, sz, sR) = rt.stage_scene_host(scene)
    leaf, fan = L["leaf"], L["fan"]
    assert L["n1"] == -(-n // leaf) and L["n2"] == -(-L["n1"] // fan)
    obj, _ = bound_records(blob, n, 0)
    top, _ = bound_records(blob, L["n2"], L["off_top"])
    leafn, _ = bound_records(blob, L["n1"], L["off_leaf"])
    memb, _ = bound_records(blob, L["n1"] * leaf, L["off_member"])
    raw_ids = blob[L["off_slot_id"]:L["off_slot_id"] + L["n1"] * leaf // 2].view(np.uint16)
    ids = raw_ids & 0x7fff
    # bit 15 of a slot id marks a torus; a node of spheres only stores its radius negated
    is_sphere = np.array([o["type"] == 1 for o in scene])
    assert np.array_equal((raw_ids[:n] & 0x8000) == 0, is_sphere[ids[:n]])
    for nodes, span in ((leafn, leaf), (top, leaf * fan)):
        for j in range(len(nodes)):
            members = ids[j * span:min(n, (j + 1) * span)]
            assert np.signbit(nodes[j, 3]) == bool(is_sphere[members].all()), (span, j)
    leafn, top = leafn.copy(), top.copy()
    leafn[:, 3] = np.abs(leafn[:, 3])
    top[:, 3] = np.abs(top[:, 3])
    # the occupied slots are a permutation of the objects and carry their own bound records
    assert sorted(ids[:n].tolist()) == list(range(n))
    assert np.array_equal(memb[:n], obj[ids[:n]])
    assert (memb[n:, 3] == -np.inf).all()
    c = obj[:, :3].astype(np.float64)
    r = np.sqrt(obj[:, 3].astype(np.float64))
    for s in range(n):
        i = ids[s]
        lf, tp = s // leaf, s // (leaf * fan)
        for node in (leafn[lf], top[tp], np.array([sx, sy, sz, sR], dtype=np.float32)):
            d = np.linalg.norm(c[i] - node[:3].astype(np.float64))
            assert d + r[i] <= float(node[3]), (s, i)
    # nodes themselves lie inside the scene sphere (the ray-level slack is derived from it)
    for node in list(leafn) + list(top):
        d = np.linalg.norm(node[:3].astype(np.float64) - np.array([sx, sy, sz]))
        assert d + float(node[3]) <= sR * (1 + 1e-6)


def test_staging_rejects_what_set_scene_rejects(rt):
    bad = rt.default_scene()
    bad[2]["center"] = [0.0, float("nan"), 0.0]
    with pytest.raises(rt.RtError):
        rt.stage_scene_host(bad)
    L, blob, _ = rt.stage_scene_host([])
    assert L["n"] == 0 and L["n1"] == 0


def test_scene_sphere_polynomial_bounds_the_direct_form():
    """The per-object loop skips rays that miss the sphere around the scene.  The kernel evaluates the
    scene-only part of that test, T(|q|^2) = (hR + sf u)^2 + 2.02 sf2 u^2 + 1e-30 with
    u = 1.0001 (|q|^2 / 2 + 1/2 + hR), as a quadratic polynomial with three host-computed coefficients
    (scene_test_polynomial in csrc/rt_types.h: exact coefficients times 1 + 1e-5, rounded to binary32,
    times 1.000001) and two fused multiply-adds.  The test must stay conservative, i.e. the binary32
    polynomial value may never fall below the exact T: restated here and checked over the ranges of
    scene radius and origin distance the renderer meets."""
    rng = np.random.default_rng(5)
    sf, sf2 = 3.75e-3, 1.42e-5
    hR = np.exp(rng.uniform(np.log(1e-3), np.log(1e6), 20000))
    hR = np.float32(hR).astype(np.float64)                        # the kernel parameter is a binary32 value
    q2 = np.float32(np.exp(rng.uniform(np.log(1e-6), np.log(1e14), 20000))).astype(np.float64)
    g, k0, k1 = sf * sf + 2.02 * sf2, 1.0001 * (0.5 + hR), 1.0001 * 0.5
    exact = [hR * hR + 2 * hR * sf * k0 + g * k0 * k0 + 1e-30, 2 * hR * sf * k1 + 2 * g * k0 * k1, np.full_like(hR, g * k1 * k1)]
    c0, c1, c2 = (np.float32(np.float32(c * (1.0 + 1e-5)) * np.float32(1.000001)) for c in exact)
    # fused multiply-add in binary32 = the exact product-sum (representable in float64 for these magnitudes
    # up to a relative 1e-16) rounded once
    inner = np.float32(c2.astype(np.float64) * q2 + c1.astype(np.float64))
    t_kernel = np.float32(inner.astype(np.float64) * q2 + c0.astype(np.float64)).astype(np.float64)
    u = 1.0001 * (0.5 * q2 + 0.5 + hR)
    t_exact = (hR + sf * u) ** 2 + 2.02 * sf2 * u * u + 1e-30
    finite = np.isfinite(t_kernel)
    assert finite.mean() > 0.9
    assert np.all(t_kernel[finite] >= t_exact[finite])
    assert np.all(t_kernel[finite] <= t_exact[finite] * (1 + 3e-5))      # and no looser than the margin it was given
