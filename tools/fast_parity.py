#!/usr/bin/env python3
"""Parity of the fast flavour against every golden fixture at the tolerance the project
states: primary hit mask equal except torus pixels; colour within 1/255 on >= 99.9 % of pixels."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, numpy as np
import rt355, rt_oracle as orc
from conftest import golden_names, load_golden, scene_for, camera_for
polish = "--polish" in sys.argv
rt355.init(0)
for name in golden_names():
    if name.startswith("spp"): continue
    meta, arr = load_golden(name)
    cam, scene = camera_for(meta), scene_for(name)
    o = dict(max_depth=meta["max_depth"], stack_size=meta["stack_size"], shadow_hops=meta["shadow_hops"])
    w, h = meta["w"], meta["h"]
    img = rt355.render_rgba(scene, cam, w, h, bottom_up=True, fast_math=True, newton_polish=polish, **o)
    if "rgba8" in arr: ref = arr["rgba8"]
    else: ref = orc.render(scene, cam.matrix(), cam.focal_length, w, h, light=cam.light_pos, **o)["rgba8"]
    close = np.all(np.abs(img[..., :3].astype(int) - ref[..., :3].astype(int)) <= 1, axis=2)
    line = f"{name:24s} {w}x{h}: within 1/255: {close.mean()*100:.4f}%  max diff {np.abs(img[...,:3].astype(int)-ref[...,:3].astype(int)).max()}"
    if "seg0_id" in arr:
        cap = rt355.capture_segment(scene, cam, w, h, 0, fast_math=True, **o)
        ids = cap[..., 0].astype(np.int16)
        diff = ids != arr["seg0_id"]
        ntor = [i for i, ob in enumerate(scene) if ob["type"] == 2]
        tor = np.isin(ids, ntor) | np.isin(arr["seg0_id"], ntor)
        line += f" | primary-mask diffs {int(diff.sum())} (non-torus {int((diff & ~tor).sum())})"
    print(line, flush=True)
