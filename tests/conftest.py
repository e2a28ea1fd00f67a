import json
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (loaded before librt355 so that the process holds ONE HIP runtime: torch ships its own copy)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(f[:-5] for f in os.listdir(GOLDEN) if f.endswith(".json"))


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    npz = os.path.join(GOLDEN, name + ".npz")
    arrays = dict(np.load(npz)) if os.path.exists(npz) else {}
    return meta, arrays


def scene_for(name):
    """The scene each fixture was rendered with (mirrors oracle/make_golden.py:cases)."""
    import rt355
    if name.startswith("tori64"):
        return rt355.tori_stress_scene()
    if name.startswith("spheres256"):
        return rt355.spheres_deep_scene()
    if name.startswith("spheres12"):
        return rt355.spheres_deep_scene(12, seed=7)
    if name.startswith("mixed1022"):
        return rt355.mixed_scene(1022)
    if name.startswith("mixed200"):
        return rt355.mixed_scene(200)
    if name.startswith("empty"):
        return []
    if name.startswith("random_"):
        return rt355.random_scene(int(name.split("_")[1]))[0]
    return rt355.default_scene()


def camera_for(meta):
    import rt355
    c = meta["camera"]
    return rt355.Camera(eye=tuple(c["eye"]), yaw=c["yaw"], pitch=c["pitch"], focal_length=c["focal_length"],
                        light_pos=tuple(c["light_pos"]))


@pytest.fixture(scope="session")
def rt():
    import rt355
    return rt355


@pytest.fixture(scope="session")
def orc():
    import rt_oracle
    rt_oracle.lib()
    return rt_oracle


@pytest.fixture(scope="session")
def gpu_rt(rt):
    """The renderer bound to GPU 0; fails loudly if the HIP library or the GPU is missing."""
    rt.init(0)
    yield rt
    rt.shutdown()
