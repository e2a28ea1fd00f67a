"""Kept leaf nodes per ray with sphere-shaped and with box-shaped leaf nodes, on the hierarchies the library
builds for the benchmark scenes (host only: rt_stage_scene_host), for rays that start on object surfaces
(half in random directions, half towards the light).  The data behind "box-shaped nodes: not built"."""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import rt355

def bound_records(blob, count, off):
    rec = blob[off:off + 8 * ((count + 1) // 2)].reshape(-1, 8)
    out = np.empty((2 * rec.shape[0], 4), dtype=np.float32)
    out[0::2] = rec[:, [0, 2, 4, 6]]; out[1::2] = rec[:, [1, 3, 5, 7]]
    return out[:count]

def sim(scene, nrays=20000, seed=1, light=(-2,4,1)):
    n = len(scene)
    L, blob, (sx, sy, sz, sR) = rt355.stage_scene_host(scene)
    leaf = L["leaf"]; n1 = L["n1"]
    obj = bound_records(blob, n, 0).astype(np.float64)
    leafn = bound_records(blob, n1, L["off_leaf"]).astype(np.float64)
    ids = blob[L["off_slot_id"]:L["off_slot_id"] + n1 * leaf // 2].view(np.uint16) & 0x7fff
    c = obj[:, :3]; r = np.sqrt(obj[:, 3])
    is_sphere = np.array([o["type"] == 1 for o in scene])
    rng = np.random.default_rng(seed)
    # leaf boxes
    lo = np.zeros((n1, 3)); hi = np.zeros((n1, 3)); sph_only = np.zeros(n1, bool)
    for j in range(n1):
        m = ids[j*leaf:min(n, (j+1)*leaf)]
        lo[j] = (c[m] - r[m, None]).min(0); hi[j] = (c[m] + r[m, None]).max(0)
        sph_only[j] = is_sphere[m].all()
    lc = leafn[:, :3]; lR = np.abs(leafn[:, 3])
    # rays: origin on a random object's bounding sphere surface, direction random (half), toward light (half)
    k = rng.integers(0, n, nrays)
    nrm = rng.normal(size=(nrays, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    o = c[k] + nrm * r[k, None] * 1.001
    d = rng.normal(size=(nrays, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    tl = np.array(light) - o; tl /= np.linalg.norm(tl, axis=1, keepdims=True)
    shadow = rng.uniform(size=nrays) < 0.5
    d[shadow] = tl[shadow]
    kept_s = kept_b = kept_b_ray = 0
    memb_pass = 0
    for i in range(nrays):
        oc = o[i] - lc
        b = oc @ d[i]
        disc = b*b - ((oc*oc).sum(1) - lR*lR)
        keep_s = (disc >= 0) & ~(sph_only & (b > lR))
        # slab test (line), and ray version for sphere-only nodes
        with np.errstate(divide='ignore', invalid='ignore'):
            inv = 1.0 / d[i]
            t1 = (lo - o[i]) * inv; t2 = (hi - o[i]) * inv
        tn = np.minimum(t1, t2).max(1); tf = np.maximum(t1, t2).min(1)
        keep_b = (tn <= tf) & ~(sph_only & (tf < 0))
        kept_s += keep_s.sum(); kept_b += keep_b.sum()
        kept_b_ray += (keep_b & keep_s).sum()
    return kept_s / nrays, kept_b / nrays, kept_b_ray / nrays

for name, scene in (("spheres256", rt355.spheres_deep_scene()), ("mixed1024", rt355.mixed_scene(1024)), ("tori64", rt355.tori_stress_scene())):
    print(name, "kept leaves per ray: sphere nodes %.2f, box nodes %.2f, both %.2f" % sim(scene, 6000))
