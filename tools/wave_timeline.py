#!/usr/bin/env python3
"""When do the persistent waves of a frame end, and how much did each do (development build:
`make -C ray-tracing_amd/csrc dev`)?  Prints the distribution of wave end times relative to the
kernel's span, the resident-wave count over time and the spread of loop trips per wave: a long thin
tail means the frame waits for a few deep pixels.

  python tools/wave_timeline.py --config spheres256_1080
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RT355_LIB", os.path.join(ROOT, "build", "dev", "librt355.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="spheres256_1080")
    args = ap.parse_args()
    import numpy as np
    import torch
    import rt355
    from kbench import CONFIGS
    cfg = dict(CONFIGS[args.config])
    which = cfg.pop("scene"); cfg.pop("golden"); w, h = cfg.pop("w"), cfg.pop("h")
    scene, cam = {"default": (rt355.default_scene, rt355.default_camera), "tori": (rt355.tori_stress_scene, rt355.tori_stress_camera),
                  "spheres": (rt355.spheres_deep_scene, rt355.spheres_deep_camera),
                  "mixed": (rt355.mixed_scene, rt355.mixed_camera)}[which]
    lib = rt355.init(0)
    if not hasattr(lib, "rt_dev_wave_log"):
        sys.exit(f"{rt355.api.LIB_PATH} is not a development build (make -C ray-tracing_amd/csrc dev)")
    lib.rt_dev_wave_log.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rt355.set_scene(scene())
    p = rt355.api.make_params(cam(), w, h, **cfg)
    buf = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda:0")
    t = rt355.api.RtTiming()
    for _ in range(2):
        lib.rt_dev_wave_log_clear()
        assert lib.rt_render_device(ctypes.addressof(p), buf.data_ptr(), buf.numel(), None, 1) == 0, lib.rt_last_error()
        lib.rt_get_timing(ctypes.addressof(t))
    WORDS = 64      # RT_WAVE_LOG_WORDS
    log = np.zeros((8192, WORDS), dtype=np.uint64)
    n = lib.rt_dev_wave_log(log.ctypes.data, 8192)
    log = log[:n]
    log = log[log[:, 1] > 0].astype(np.float64)
    t0, t1 = log[:, 0].min(), log[:, 1].max()
    span = (t1 - t0) / 100.0          # microseconds (100 MHz ticks)
    print(f"{args.config}: kernel {t.kernel_ms:.3f} ms, {len(log)} waves, span of the waves {span / 1000:.3f} ms")
    end = (log[:, 1] - t0) / (t1 - t0)
    print("wave end time / span: " + "  ".join(f"p{q}={np.percentile(end, q):.3f}" for q in (1, 10, 25, 50, 75, 90, 99, 100)))
    grid = np.linspace(0, 1, 21)[1:]
    alive = [(int(((log[:, 0] - t0) / (t1 - t0) <= g).sum() - (end < g).sum())) for g in grid]
    print("waves still running at 5% steps of the span: " + " ".join(str(a) for a in alive))
    print(f"mean fraction of the span a wave is alive: {((log[:, 1] - log[:, 0]) / (t1 - t0)).mean():.3f}")
    trips, px = log[:, 2], log[:, 3]
    print(f"loop trips per wave: mean {trips.mean():.0f}  p50 {np.percentile(trips, 50):.0f}  max {trips.max():.0f};  "
          f"pixels per wave: mean {px.mean():.0f} min {px.min():.0f} max {px.max():.0f}")
    us_per_trip = (log[:, 1] - log[:, 0]) / 100.0 / np.maximum(trips, 1)
    print(f"microseconds per loop trip: p10 {np.percentile(us_per_trip, 10):.2f}  p50 {np.percentile(us_per_trip, 50):.2f}  p90 {np.percentile(us_per_trip, 90):.2f}")
    names = ["top-node loop", "leaf loop", "candidate loop (step 2)", "queue scan reads", "cooperative rays", "cooperative rounds"]
    HS = 23         # HS_COUNT (rt_intersect.h)
    tot_trips = trips.sum()
    print("hierarchy walk, loop iterations per loop trip of a wave (wave-level = instructions issued; lanes = mean lanes in it):")
    for k, nm in enumerate(names):
        wv, ln = log[:, 4 + k].sum(), log[:, 4 + HS + k].sum()
        print(f"  {nm:26s} {wv / max(tot_trips, 1):8.2f} per trip   lanes per iteration {ln / max(wv, 1):6.1f}")
    print(f"  {'torus rounds':26s} {log[:, 4 + 14].sum() / max(tot_trips, 1):8.2f} per trip")
    tnames = ["loop head + refill", "cooperative trace", "node tests + item lists", "leaf rounds", "exact rounds", "owner + replay", "shading + state machine"]
    tt = [log[:, 4 + 16 + k].sum() for k in range(7)]
    if sum(tt) > 0:
        print("cycles of a wave by phase (s_memtime deltas; other waves of the SIMD run in between):")
        for nm, v in zip(tnames, tt):
            print(f"  {nm:34s} {100 * v / sum(tt):6.1f} %   {v / max(tot_trips, 1):9.0f} cycles per trip")
    stats = ["rays", "shadow rays", "rays that hit an object", "kept leaf nodes", "kept leaf nodes in front of the hit",
             "the same, shadow rays also cut at the light", "candidates", "candidates in front of the hit", "-", "rays with a hit closer than 1"]
    tot = [log[:, 4 + HS + 6 + k].sum() for k in range(len(stats))]
    if tot[0] > 0:
        print("cut-off statistics of the pooled trace (RT_DEV_STATS=1), per ray:")
        for nm, v in zip(stats, tot):
            print(f"  {nm:44s} {v / tot[0]:8.3f}   ({v / max(tot_trips, 1):7.2f} per trip)")
    last = np.argsort(log[:, 1])[-5:]
    print("the five last waves: end/span, trips, pixels: " + "; ".join(f"{end[i]:.3f} {int(trips[i])} {int(px[i])}" for i in last))


if __name__ == "__main__":
    main()
