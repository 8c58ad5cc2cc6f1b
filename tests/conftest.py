import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference and Mesa llvmpipe (this container only)")


def load_golden(name):
    from raytracing_amd import abi
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: z[k] for k in z.files}
    g["trans"] = np.frombuffer(g["trans"].tobytes(), dtype=abi.TRANS)
    g["ids"] = np.stack([g["hit_type"].astype(np.uint32), g["hit_idx"].astype(np.uint32)], axis=-1)
    for k in ("width", "height", "depth", "variant"):
        g[k] = int(g[k])
    return g


def image_stats(a, b):
    """a, b: [H, W, >=3] float images -> dict(rmse, max, frac_gt)"""
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    return {"rmse": float(np.sqrt(np.mean(d * d))), "max": float(np.abs(d).max()),
            "frac_gt_1e-3": float((np.abs(d).max(axis=-1) > 1e-3).mean())}


@pytest.fixture(scope="session")
def default_model():
    from raytracing_amd import scenes
    return scenes.bvh_default()


@pytest.fixture(scope="session")
def bunny0_model():
    from raytracing_amd import scenes
    return scenes.mesh_scene(subdivisions=0, builder="median")


@pytest.fixture(scope="session")
def tracer():
    from raytracing_amd import tracer as T
    t = T.Tracer(0)
    yield t
    t.close()
