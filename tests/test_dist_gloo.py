"""The N>1 path on CPU: two gloo ranks render interleaved bands (the oracle stands in for
the GPU kernel), gather to rank 0, assemble, and must reproduce the single-rank frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, w, h, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import rt355
    import rt_oracle as orc
    from importlib import import_module
    tiling = import_module("ray-tracing_amd.tiling")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, cam = rt355.default_scene(), rt355.default_camera()

    def render_part(part, nparts, out):
        # top-down output rows r <-> GL rows h-1-r; the oracle renders GL rows
        a = out.numpy()
        src = 0
        for r0, r1 in tiling.part_row_ranges(h, part, nparts):
            gl = orc.render(scene, cam.matrix(), cam.focal_length, w, h, rows=(h - r1, h - r0), nthreads=2)["rgba8"]
            a[src:src + (r1 - r0)] = gl[::-1][..., :a.shape[-1]]
            src += r1 - r0
        a[src:] = 0

    g = tiling.render_distributed(render_part, h, w, 4, torch.uint8, "cpu", dist)
    if rank == 0:
        single = tiling.assemble_numpy(g.numpy(), h, world)

    # the pipelined form bench.py drives: several frames in flight, gather overlapped with render
    def assemble(gathered, image):
        image.copy_(torch.from_numpy(tiling.assemble_numpy(gathered.numpy(), h, world)))

    pipe = tiling.BandPipeline(h, w, 4, torch.uint8, "cpu", dist, render_part, assemble)
    for _ in range(3):
        pipe.step()
    pipe.flush()
    # parts travelling as packed RGB (three channels), assembled into the RGBA frame
    def assemble_rgb(gathered, image):
        image[..., :3] = torch.from_numpy(tiling.assemble_numpy(gathered.numpy(), h, world))
        image[..., 3] = 255

    pipe3 = tiling.BandPipeline(h, w, 3, torch.uint8, "cpu", dist, render_part, assemble_rgb, image_channels=4)
    for _ in range(2):
        pipe3.step()
    pipe3.flush()
    if rank == 0:
        assert np.array_equal(pipe.image.numpy(), single)
        assert pipe3.image.shape == (h, w, 4) and np.array_equal(pipe3.image.numpy(), single)
        q.put(single)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h", [(96, 70), (64, 33)])
def test_two_rank_band_split_reassembles(orc, rt, w, h):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, q)) for r in range(2)]
    for p in procs:
        p.start()
    img = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cam = rt.default_camera()
    full = orc.render(rt.default_scene(), cam.matrix(), cam.focal_length, w, h)["rgba8"][::-1]
    assert np.array_equal(img, full)


def test_band_partition_covers_every_row_once(rt):
    from importlib import import_module
    tiling = import_module("ray-tracing_amd.tiling")
    for h in (1, 15, 16, 17, 270, 2160, 4320):
        for n in (1, 2, 3, 4, 8):
            seen = np.zeros(h, dtype=int)
            for part in range(n):
                for r0, r1 in tiling.part_row_ranges(h, part, n):
                    seen[r0:r1] += 1
                assert tiling.part_rows(h, part, n) <= tiling.padded_part_rows(h, n)
            assert (seen == 1).all()
