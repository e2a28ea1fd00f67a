import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, rt355
scene, cam = rt355.spheres_deep_scene(), rt355.spheres_deep_camera()
rt355.init(0)
w, h = 480, 270
cap = rt355.capture_segment(scene, cam, w, h, 0, max_depth=16, stack_size=16, shadow_hops=4)
nseg = cap[..., 3].astype(np.int64)
print("pixels", nseg.size, "mean", nseg.mean(), "max", nseg.max(), "sum", nseg.sum())
for q in (50, 90, 99, 99.9, 99.99):
    print("p%s" % q, np.percentile(nseg, q))
s = np.sort(nseg.ravel())[::-1]
print("top 10", s[:10])
print("share of total in top 0.1%% of pixels: %.3f, top 1%%: %.3f, top 10%%: %.3f" % (s[:len(s)//1000].sum()/s.sum(), s[:len(s)//100].sum()/s.sum(), s[:len(s)//10].sum()/s.sum()))
np.save("/root/repo/gpurun_out/nseg_s256_480.npy", nseg.astype(np.int32))
