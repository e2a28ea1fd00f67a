#!/usr/bin/env python3
"""Headline benchmark: Mpixel-paths/s (and fps) of the per-pixel ray-tracing path on the
default scene at 3840x2160, max bounce depth 8, on 1/2/4/8 MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one full 3840x2160 frame through the HIP path (strict flavour: the frame is
bit-identical to the reference shader's).  Scene and camera are resident on the GPU before
the timed region; the frame stays in HBM on GPU 0.  With N GPUs the frame is cut into
interleaved 16-row bands, each rank renders its bands, rank 0 gathers them with RCCL
(torch.distributed "nccl") and scatters them into place; total work per step is fixed, so
scaling is "strong".

Extra objects in the JSON line:
  roofline     FP32-VALU roofline of the render kernel: algorithmic FLOPs of the frame
               (event counters of the instrumented kernel x the shader-source FLOP table)
               / average kernel time measured with HIP events on the launch stream,
               against the 157.3 TFLOP/s FP32 vector peak (78.6 without packed FMA).
  cpu_baseline the CPU oracle (the reference shader restated in C, OpenMP) timed on this
               box's host cores on a bounded sample of the same workload, rank 0 at N=1.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

W, H = 3840, 2160
MAX_DEPTH, STACK_SIZE, SHADOW_HOPS = 8, 3, 2

# --workload: the headline configuration (default) and the other configurations of the
# benchmark list, which are parity-test cases first and are timed here for the record
WORKLOADS = {
    "default4k": dict(scene="default", w=3840, h=2160, max_depth=8, stack_size=3, shadow_hops=2, spp_grid=1,
                      text="default main.py scene, 3840x2160, max_depth 8, stack_size %(stack_size)d, 2 shadow hops, 1 spp"),
    "tori64_4k": dict(scene="tori", w=3840, h=2160, max_depth=8, stack_size=3, shadow_hops=2, spp_grid=1,
                      text="64 tori at grazing angles, 3840x2160, max_depth 8, stack_size %(stack_size)d, 2 shadow hops"),
    "spheres256_4k": dict(scene="spheres", w=3840, h=2160, max_depth=16, stack_size=16, shadow_hops=4, spp_grid=1,
                          text="256 glass/mirror spheres, 3840x2160, max_depth 16, stack_size %(stack_size)d, 4 shadow hops"),
    "mixed1024_8k": dict(scene="mixed", w=7680, h=4320, max_depth=8, stack_size=8, shadow_hops=2, spp_grid=4,
                         text="1024 mixed objects, 7680x4320, 16 spp, max_depth 8, stack_size %(stack_size)d, 2 shadow hops"),
}
# 256 CU x 4 SIMD x 32 lanes per clock x 2 (FMA) x 2.4 GHz: a SIMD issues one wave64 v_fma_f32 every
# 2 cycles; v_pk_fma_f32 takes 4, so packed math reaches the same peak, not twice it
# (tools/microbench/valu_rate; its output is kept with the committed profiles as valu_issue_rates_mi355x.txt)
PEAK_FP32_VECTOR_TFLOPS = 157.3
# the same lanes executing one operation per instruction (add, mul; the reference's arithmetic is
# separately rounded, so the strict flavour cannot fuse): 78.6 T lane-operations/s
PEAK_FP32_LANE_OPS_T = 78.6


def hbm_traffic_bytes(args):
    """HBM bytes per launch of the render kernel as measured with rocprofv3 PMC passes
    (FETCH_SIZE doubled as gfx950 requires, plus WRITE_SIZE; tools/prof_hbm.sh); the summary
    of the latest round under profiles/ is read back, None when there is none for this config."""
    if args.fast_math or args.stack_size != STACK_SIZE or args.workload != "default4k":
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_4k_d8s3_strict.json")))
    if not files:
        return None
    try:
        return float(json.load(open(files[-1]))["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        return None


def executed_work(workload, fast_math, kernel_ms):
    """What the hardware executed, from the committed rocprofv3 counter summary of this workload's
    render kernel (profiles/r*/pmc_<workload>.json, tools/prof_workload.sh; the summary names the kernel
    sources it was collected with, and one that belongs to other sources is reported as stale): VALU lane-operations
    per launch = SQ_INSTS_VALU x 64 x VALUUtilization (instruction counts do not change from launch
    to launch), their rate over the kernel time measured in this run (launches under the counter
    collection itself run 1.0-1.4x longer) against the 78.6 T lane-operations/s of the chip, VALUBusy
    (gfx94x formula: 4 cycles per instruction, so it passes 100 % on gfx950) and the active-lane fraction.
    Unlike the algorithmic figure it cannot exceed 1 and does not credit work a spatial hierarchy
    avoids.  None when no summary is committed for the workload."""
    if fast_math:
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_%s.json" % workload)))
    if not files:
        return None
    try:
        c = json.load(open(files[-1]))
        import rt355
        fp = rt355.api.library_fingerprint()
        if c.get("kernel_src_sha256") != fp["kernel_src_sha256"]:
            # the counters were collected with other kernels than the ones that just ran: no rate
            return {"stale": True, "source": os.path.relpath(files[-1], ROOT),
                    "summary_kernel_src_sha256": c.get("kernel_src_sha256"), "running_kernel_src_sha256": fp["kernel_src_sha256"]}
        lane_ops = c["SQ_INSTS_VALU"] * 64.0 * c["VALUUtilization"] / 100.0
        secs = kernel_ms * 1e-3
        return {"valu_lane_ops_per_launch": lane_ops, "rate_T_per_s": lane_ops / secs / 1e12,
                "frac_of_lane_rate": lane_ops / secs / 1e12 / PEAK_FP32_LANE_OPS_T,
                "valu_busy_pct": c["VALUBusy"], "valu_utilization_pct": c["VALUUtilization"],
                "kernel_ms_profiled": c["kernel_ms"], "source": os.path.relpath(files[-1], ROOT), "stale": False,
                "lib_sha256_matches": c.get("lib_sha256") == fp["lib_sha256"]}
    except (OSError, ValueError, KeyError):
        return None


def main_inlib(args, wl):
    """--multi inlib: ONE process drives all N GPUs through rt_init_multi (each device renders its
    bands and copies them straight into the frame on device 0; no RCCL).  Launched directly, or under
    torch.distributed.run, where rank 0 does the work and the other ranks leave at once.  On a machine
    with fewer than N GPUs the parts share the devices there are (a rehearsal of the code path; the
    line says so)."""
    if int(os.environ.get("RANK", "0")) != 0:
        return
    import hashlib
    import torch
    import rt355
    n = args.gpus
    ndev = torch.cuda.device_count()
    devices = [i % ndev for i in range(n)]
    lib = rt355.init_multi(devices=devices)
    scene, cam = {"default": (rt355.default_scene, rt355.default_camera),
                  "tori": (rt355.tori_stress_scene, rt355.tori_stress_camera),
                  "spheres": (rt355.spheres_deep_scene, rt355.spheres_deep_camera),
                  "mixed": (rt355.mixed_scene, rt355.mixed_camera)}[wl["scene"]]
    scene, cam = scene(), cam()
    rt355.set_scene(scene)
    flags = rt355.api.FLAG_FAST_MATH if args.fast_math else 0
    params = rt355.api.make_params(cam, W, H, max_depth=MAX_DEPTH, stack_size=args.stack_size, shadow_hops=SHADOW_HOPS,
                                   spp_grid=wl["spp_grid"], flags=flags)
    pptr = ctypes.addressof(params)
    dev0 = torch.device("cuda", devices[0])
    torch.cuda.set_device(dev0)
    nbuf = 4
    streams = [torch.cuda.Stream(device=dev0) for _ in range(nbuf)]
    frames = [torch.zeros((H, W, 4), dtype=torch.uint8, device=dev0) for _ in range(nbuf)]
    sptrs = [ctypes.c_void_p(s_.cuda_stream) for s_ in streams]

    def check(rc):
        if rc != 0:
            raise RuntimeError(lib.rt_last_error().decode())

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    def run(steps, rotate):
        for i in range(steps):
            k = i % nbuf if rotate else 0
            check(lib.rt_render_device(pptr, frames[k].data_ptr(), frames[k].numel(), sptrs[k], 0))

    run(args.warmup, False)
    sync_all()
    t0 = time.perf_counter()
    run(args.steps, False)
    host_issue_s = time.perf_counter() - t0
    sync_all()
    ms_per_step = (time.perf_counter() - t0) * 1e3 / args.steps
    # informational: consecutive frames on rotating streams / frame buffers (the last waves of a frame overlap the next one)
    run(nbuf, True)
    sync_all()
    t0 = time.perf_counter()
    run(args.steps, True)
    sync_all()
    ms_rotating = (time.perf_counter() - t0) * 1e3 / args.steps
    # one timed frame: kernel alone and copy alone, per part
    check(lib.rt_render_device(pptr, frames[0].data_ptr(), frames[0].numel(), sptrs[0], 1))
    sync_all()
    parts = []
    for i in range(n):
        try:
            k, c = rt355.api.part_timing(i)
            parts.append({"part": i, "device": devices[i], "kernel_ms": k, "copy_ms": c})
        except rt355.RtError:
            parts.append({"part": i, "device": devices[i], "kernel_ms": None, "copy_ms": None})
    headline = args.workload == "default4k"
    mpix = W * H * wl["spp_grid"] ** 2 / 1e6
    result = {
        "metric": "Mpixel-paths/s at 3840x2160 (default scene, max bounce depth 8)" if headline else "Mpixel-paths/s (%s)" % args.workload,
        "value": mpix / (ms_per_step * 1e-3), "unit": "Mpixel-paths/s", "fps": 1e3 / ms_per_step, "n_gpus": n, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (the reference's default scene and camera; no dataset exists)" if headline
                                else "synthetic (seeded stress scene of the benchmark list)",
        "config": {"workload": wl["text"] % wl, "numerics": "fast" if args.fast_math else "strict (bit-identical to the reference shader)",
                   "tiling": "in-library: one process, %d parts of interleaved 16-row bands on devices %s, strided peer copies into the frame on device %d"
                             % (n, devices, devices[0])},
        "host_issue_ms_per_step": host_issue_s * 1e3 / args.steps,
        "rotating_streams": {"streams": nbuf, "ms_per_step": ms_rotating, "note": "not the reported value"},
        "parts": parts,
        "kernel_only_ms_per_step": max((p_["kernel_ms"] or 0.0) for p_ in parts),
        "kernel_plus_copy_ms_per_step": max(((p_["kernel_ms"] or 0.0) + (p_["copy_ms"] or 0.0)) for p_ in parts),
    }
    if len(set(devices)) < n:
        result["rehearsal"] = "%d parts on %d GPU(s): the code path, not a scaling measurement" % (n, len(set(devices)))
    golden_name = {"default4k": "default_4k_d8s3", "tori64_4k": "tori64_4k_d8s3"}.get(args.workload)
    golden_file = os.path.join(ROOT, "tests", "golden", "%s.json" % golden_name) if golden_name else None
    if golden_file and os.path.exists(golden_file) and not args.fast_math and args.stack_size == WORKLOADS[args.workload]["stack_size"]:
        golden = json.load(open(golden_file))
        got = hashlib.sha256(frames[0].flip(0).contiguous().cpu().numpy().tobytes()).hexdigest()
        result["frame_sha_ok"] = got == golden["sha256_rgba8"]
    else:
        result["frame_sha_ok"] = None
    result["library"] = rt355.api.library_fingerprint()
    print(json.dumps(result))
    rt355.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--fast-math", action="store_true", help="bench the fast flavour instead of the strict one")
    ap.add_argument("--stack-size", type=int, default=None)
    ap.add_argument("--workload", default="default4k", choices=sorted(WORKLOADS))
    ap.add_argument("--roofline", action="store_true",
                    help="also for the other workloads: count the frame's events with the instrumented kernel (slow)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inflight", action="store_true",
                    help="skip the informational four-frames-in-flight figure (profiler runs: its overlapping launches would "
                         "stretch the kernel's average duration in the trace)")
    ap.add_argument("--multi", default="rccl", choices=["rccl", "inlib"],
                    help="N > 1: one process per GPU with an RCCL gather (default), or one process driving all GPUs inside the library")
    args = ap.parse_args()
    wl = dict(WORKLOADS[args.workload])
    wl_default_stack = wl["stack_size"]
    if args.stack_size is None:
        args.stack_size = wl["stack_size"]
    wl["stack_size"] = args.stack_size
    headline = args.workload == "default4k"
    global W, H, MAX_DEPTH, SHADOW_HOPS
    W, H, MAX_DEPTH, SHADOW_HOPS = wl["w"], wl["h"], wl["max_depth"], wl["shadow_hops"]
    if args.multi == "inlib":
        return main_inlib(args, wl)

    # Multi-GPU runs keep several frames in flight on separate streams; HIP gives a process four
    # hardware queues by default and streams that share a queue do not overlap, so ask for eight
    # (read when the HIP runtime starts, i.e. before torch is imported)
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"

    import torch
    import torch.distributed as dist
    import rt355
    from importlib import import_module
    tiling = import_module("ray-tracing_amd.tiling")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RT_BENCH_FORCE_GATHER=1 (under torch.distributed.run, also with one rank) sends every frame
    # through the RCCL gather + assemble path: a rehearsal of the N > 1 code on a one-GPU box
    force_gather = os.environ.get("RT_BENCH_FORCE_GATHER") == "1" and "RANK" in os.environ
    use_dist = world > 1 or force_gather
    # The streams of the frame pipeline are created first, back to back: the runtime deals
    # hardware queues to streams in creation order, and streams created later (after RCCL has
    # made its own) ended up sharing queues, i.e. not overlapping.
    nstreams = int(os.environ.get("RT_BENCH_STREAMS", "4")) if use_dist else 1
    all_streams = [torch.cuda.Stream(device=dev) for _ in range(max(nstreams, 4))]      # (N = 1 uses the extra ones after the timed region)
    for s_ in all_streams:                      # first use binds a stream to its queue
        with torch.cuda.stream(s_):
            torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    lib = rt355.init(local_rank)
    scene, cam = {"default": (rt355.default_scene, rt355.default_camera),
                  "tori": (rt355.tori_stress_scene, rt355.tori_stress_camera),
                  "spheres": (rt355.spheres_deep_scene, rt355.spheres_deep_camera),
                  "mixed": (rt355.mixed_scene, rt355.mixed_camera)}[wl["scene"]]
    scene, cam = scene(), cam()
    rt355.set_scene(scene)
    flags = rt355.api.FLAG_FAST_MATH if args.fast_math else 0
    # between GPUs the parts travel as packed RGB8 (alpha is the constant 255 and is put back by
    # the assemble kernel): a quarter less through the xGMI links, which bound this cheap frame
    # (RT_BENCH_RGBA_TRANSPORT=1 sends RGBA8 instead, for an A/B)
    tch = 4 if (not use_dist or os.environ.get("RT_BENCH_RGBA_TRANSPORT") == "1") else 3
    if tch == 3:
        flags |= rt355.api.FLAG_OUT_RGB8
    # RT_BENCH_REHEARSE_NPARTS=N (one-rank rehearsal only): every step renders part 0 of N, i.e.
    # the kernel load of one GPU of N, through the same pipeline (the assembled frame is then
    # meaningless and not checked)
    rehearse_nparts = int(os.environ.get("RT_BENCH_REHEARSE_NPARTS", "0")) if (force_gather and world == 1) else 0
    params = rt355.api.make_params(cam, W, H, max_depth=MAX_DEPTH, stack_size=args.stack_size,
                                   shadow_hops=SHADOW_HOPS, spp_grid=wl["spp_grid"], flags=flags,
                                   part=rank, nparts=rehearse_nparts or world)
    # a real (non-null) stream: the kernels, the RCCL hand-off and the timing events all go
    # through it (a NULL stream argument would select the library's private stream, which
    # torch events do not see)
    stream = all_streams[0]
    torch.cuda.set_stream(stream)
    sptr = ctypes.c_void_p(stream.cuda_stream)
    assert stream.cuda_stream != 0
    pptr = ctypes.addressof(params)

    def check(rc):
        if rc != 0:
            raise RuntimeError(lib.rt_last_error().decode())

    # Split frames rotate over four streams, so that the last waves of a frame's part (the
    # deepest pixels) overlap the parts of the next frames (RT_BENCH_STREAMS=1 keeps them on one
    # stream, for an A/B).  Four streams only overlap when the process has more than HIP's
    # default of four hardware queues: see the top of main().
    streams = all_streams[:nstreams] if (use_dist and nstreams > 1) else None

    sptrs = [ctypes.c_void_p(s_.cuda_stream) for s_ in streams] if streams else [sptr] * 2

    def cur_stream():
        return sptrs[pipe.cur]

    def render_part(part, nparts, out):
        check(lib.rt_render_device(pptr, out.data_ptr(), out.numel(), cur_stream(), 0))

    def assemble(gathered, image):
        check(lib.rt_assemble_device(gathered.data_ptr(), image.data_ptr(), W, H, world, tch, cur_stream()))

    pipe = tiling.BandPipeline(H, W, tch, torch.uint8, dev, dist if use_dist else None, render_part, assemble,
                               always_gather=force_gather, image_channels=4, streams=streams)

    def run(steps):
        for _ in range(steps):
            pipe.step()
        pipe.flush()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The frame's work (event counters of the instrumented kernel, for the roofline) is counted before the
    # timed steps, not after them: it is set-up like the scene upload, and a GPU that has just run a
    # kernel of several milliseconds has left its idle clocks (measured: the first ~30 launches after an
    # idle period take 0.30-0.32 ms each, later ones 0.28).
    frame_events = None
    if world == 1 and rank == 0 and not force_gather and (headline or args.roofline):
        import numpy as np
        full = rt355.api.make_params(cam, W, H, max_depth=MAX_DEPTH, stack_size=args.stack_size,
                                     shadow_hops=SHADOW_HOPS, spp_grid=wl["spp_grid"], flags=flags)
        frame_events = np.zeros(lib.rt_event_count(), dtype=np.uint64)
        check(lib.rt_count_events(ctypes.addressof(full), frame_events.ctypes.data))

    # Every rank first has the library check its arithmetic cores on its own GPU (exhaustive over the
    # operand ranges the kernels use: a device whose square root or division disagreed with IEEE 754
    # would render, and so time, a different frame).  These are set-up kernels like the scene upload;
    # they also leave the GPU at its working clocks when the warm-up steps start.
    selftest_ok = None
    if not args.fast_math:
        s0 = time.perf_counter()
        selftest_ok = not any(int(c) for c in rt355.api.selftest_math())
        selftest_s = time.perf_counter() - s0

    # ---- single GPU, informational, ahead of the timed steps (the buffers and streams it needs are
    # set-up, and the timed steps then start on a GPU that is already at its working clocks: the first
    # launches after an idle stretch run 5-10 % slow): the same frames rotating over four streams
    # (what the N > 1 runs do with their parts): the last waves of a frame, its deepest tiles, overlap
    # the next frames
    in_flight = None
    if world == 1 and rank == 0 and not force_gather and not args.no_inflight:
        k_steps = 200
        bufs = [torch.empty_like(pipe.image) for _ in range(4)]
        s4 = [ctypes.c_void_p(s_.cuda_stream) for s_ in all_streams[:4]]
        for i in range(8):
            check(lib.rt_render_device(pptr, bufs[i % 4].data_ptr(), bufs[i % 4].numel(), s4[i % 4], 0))
        torch.cuda.synchronize()
        f0 = time.perf_counter()
        for i in range(k_steps):
            check(lib.rt_render_device(pptr, bufs[i % 4].data_ptr(), bufs[i % 4].numel(), s4[i % 4], 0))
        torch.cuda.synchronize()
        in_flight = {"streams": 4, "steps": k_steps, "ms_per_step": (time.perf_counter() - f0) * 1e3 / k_steps,
                     "note": "not the reported value: consecutive frames on rotating streams, host clock"}
        del bufs

    # N > 1 (no four-streams figure there): the same number of untimed frames through the pipeline, for the
    # same reason -- working clocks, and RCCL's first collectives (connection set-up) out of the way
    preroll = 0
    if use_dist:
        preroll = 200
        run(preroll)

    run(args.warmup)
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    run(args.steps)
    host_issue_s = time.perf_counter() - t0          # host time to enqueue the steps (before any sync)
    e1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    stream_ms = e0.elapsed_time(e1)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # kernel-only view of the same split (no gather, no assemble): max over ranks of the average
    # render time of this rank's bands, HIP events on the launch stream
    kernel_only_ms = None
    if use_dist:
        k_steps = min(args.steps, 100)
        barrier()
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record(stream)
        for _ in range(k_steps):
            render_part(rank, world, pipe.mine[0])
        k1.record(stream)
        torch.cuda.synchronize()
        t = torch.tensor([k0.elapsed_time(k1) / k_steps], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kernel_only_ms = float(t.item())
        kernel_only_2s_ms = None
        if streams:
            # the same launches rotating over the streams
            barrier()
            k0.record(stream)
            for other in streams[1:]:
                other.wait_event(k0)
            for i in range(k_steps):
                pipe.use_stream(i)
                render_part(rank, world, pipe.mine[i % len(streams)])
            pipe.use_stream(0)
            for other in streams[1:]:
                stream.wait_stream(other)
            k1.record(stream)
            torch.cuda.synchronize()
            t = torch.tensor([k0.elapsed_time(k1) / k_steps], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            kernel_only_2s_ms = float(t.item())

    # the transfer alone (no render, no assemble): k_steps gathers of this rank's part buffer
    gather_only_ms = None
    if use_dist:
        barrier()
        g0 = time.perf_counter()
        for i in range(k_steps):
            dist.gather(pipe.mine[i % pipe.nbuf], gather_list=pipe.chunks[i % pipe.nbuf] if rank == 0 else None, dst=0)
        barrier()
        t = torch.tensor([(time.perf_counter() - g0) * 1e3 / k_steps], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gather_only_ms = float(t.item())

    result = None
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        mpix = W * H * wl["spp_grid"] ** 2 / 1e6
        result = {
            "metric": "Mpixel-paths/s at 3840x2160 (default scene, max bounce depth 8)" if headline
                      else "Mpixel-paths/s (%s)" % args.workload,
            "value": mpix / (ms_per_step * 1e-3), "unit": "Mpixel-paths/s", "fps": 1e3 / ms_per_step,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (the reference's default scene and camera; no dataset exists)" if headline
                    else "synthetic (seeded stress scene of the benchmark list)",
            "config": {"workload": wl["text"] % wl,
                       "numerics": "fast" if args.fast_math else "strict (bit-identical to the reference shader)",
                       "tiling": "%d x interleaved 16-row bands, RCCL gather to GPU 0 (parts travel as %s)" % (world, "RGB8" if tch == 3 else "RGBA8")
                                 if use_dist else "single GPU"},
        }
        if kernel_only_ms is not None:
            result["kernel_only_ms_per_step"] = kernel_only_ms      # value above includes gather + assemble
            if kernel_only_2s_ms is not None:
                result["kernel_only_all_streams_ms_per_step"] = kernel_only_2s_ms
                result["streams"] = len(streams)
            result["host_issue_ms_per_step"] = host_issue_s * 1e3 / args.steps   # rank 0's enqueue cost
            result["gather_only_ms_per_step"] = gather_only_ms                   # transfer alone, max over ranks
            result["part_bytes"] = int(pipe.mine[0].numel())
            result["untimed_frames_before_warmup"] = preroll

    if in_flight is not None:
        result["four_frames_in_flight"] = in_flight

    # ---- kernel-only measurements, single GPU: roofline + cpu baseline
    if frame_events is not None:
        flops = rt355.frame_flops(frame_events)
        kernel_ms = stream_ms / args.steps          # back-to-back launches of the one kernel on this stream
        tim = rt355.api.RtTiming()
        image = pipe.image
        check(lib.rt_render_device(ctypes.addressof(full), image.data_ptr(), image.numel(), sptr, 1))
        torch.cuda.synchronize()
        check(lib.rt_get_timing(ctypes.addressof(tim)))
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        result["roofline"] = {
            "bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_VECTOR_TFLOPS, "frac_of_lane_op_rate": achieved / PEAK_FP32_LANE_OPS_T,
            "traffic": hbm_traffic_bytes(args),
            "kernel": "rt_kernel<%d,false> (%s)" % (tim.pending_cap, "fast" if args.fast_math else "strict"),
            "kernel_ms_avg": kernel_ms, "kernel_ms_single_launch": tim.kernel_ms,
            "algorithmic_flops_per_frame": flops, "flops_per_pixel_path": flops / (W * H * wl["spp_grid"] ** 2),
            "vgprs": tim.vgprs, "lds_bytes": tim.lds_bytes,
        }
        if wl["scene"] != "default":
            result["roofline"]["note"] = ("algorithmic FLOPs count the reference's per-object loop (every object's bounding test "
                                          "per ray); the sphere hierarchy does not execute most of them: see `executed`")
        ex = executed_work(args.workload, args.fast_math, kernel_ms)
        if ex:
            result["roofline"]["executed"] = ex
        traffic = result["roofline"]["traffic"]
        if traffic:
            # the memory side of the same launch, to show which roof is the near one
            gbps = traffic / (kernel_ms * 1e-3) / 1e9
            result["roofline"]["hbm"] = {"achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0}
        if not args.no_cpu_baseline and headline:
            import rt_oracle as orc
            # the host cores this job may use: the box sets OMP_NUM_THREADS to its share of the machine
            host_cpus = os.cpu_count() or 1
            threads = min(int(os.environ.get("OMP_NUM_THREADS", host_cpus)), host_cpus)
            frames = 0
            orc.render(scene, cam.matrix(), cam.focal_length, W, H // 8, max_depth=MAX_DEPTH,
                       stack_size=args.stack_size, shadow_hops=SHADOW_HOPS, nthreads=threads)   # warm up
            c0 = time.perf_counter()
            while frames < 3 or (time.perf_counter() - c0 < 2.0 and frames < 20):
                orc.render(scene, cam.matrix(), cam.focal_length, W, H, max_depth=MAX_DEPTH,
                           stack_size=args.stack_size, shadow_hops=SHADOW_HOPS, nthreads=threads)
                frames += 1
            cpu_s = (time.perf_counter() - c0) / frames
            result["cpu_baseline"] = {
                "value": W * H / 1e6 / cpu_s, "unit": "Mpixel-paths/s", "cores": threads, "host_cpus": host_cpus, "kind": "port",
                "sample": "%d full 3840x2160 frames of the same workload (%.2f s each) on %d OpenMP threads "
                          "(OMP_NUM_THREADS; the machine has %d CPUs)" % (frames, cpu_s, threads, host_cpus),
            }

    if rank == 0:
        # The frame the timed steps left in HBM must be the reference's frame: every run checks itself
        # against the golden SHA-256 of the workload (llvmpipe fixtures; rows there are bottom-up).
        import hashlib
        golden_name = {"default4k": "default_4k_d8s3", "tori64_4k": "tori64_4k_d8s3"}.get(args.workload)
        golden_file = os.path.join(ROOT, "tests", "golden", "%s.json" % golden_name) if golden_name else None
        if rehearse_nparts:
            result["rehearsed_part"] = "0 of %d" % rehearse_nparts
        elif golden_file and os.path.exists(golden_file) and not args.fast_math and args.stack_size == wl_default_stack:
            golden = json.load(open(golden_file))
            torch.cuda.synchronize()
            got = hashlib.sha256(pipe.image.flip(0).contiguous().cpu().numpy().tobytes()).hexdigest()
            result["frame_sha_ok"] = got == golden["sha256_rgba8"]
            if use_dist:
                result["gathered_frame_sha_ok"] = result["frame_sha_ok"]
        else:
            result["frame_sha_ok"] = None           # no llvmpipe fixture at this workload's parameters
        if selftest_ok is not None:
            result["arithmetic_selftest"] = {"ok": selftest_ok, "seconds": selftest_s}
        result["library"] = rt355.api.library_fingerprint()
        print(json.dumps(result))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
