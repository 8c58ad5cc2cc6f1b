import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- first: torch ships its own ROCm runtime libraries; if liblrltrace.so pulls in /opt/rocm's copies before torch is
#                              imported (a run of one test file that loads the tracer first), torch later finds "No HIP GPUs"

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
    """a, b: [H, W, >=3] float images -> dict(rmse, max, frac_gt, nonfinite_mismatch).
    Pixels that are non-finite in BOTH images are skipped (the reference's sphere uv is acos() of a ratio that
    rounding can push past 1, which is NaN in GLSL, in the oracle and on the GPU alike); a pixel that is
    non-finite in only one of them counts in `nonfinite_mismatch`, which callers bound."""
    a3, b3 = a[..., :3].astype(np.float64), b[..., :3].astype(np.float64)
    fa, fb = np.isfinite(a3).all(axis=-1), np.isfinite(b3).all(axis=-1)
    both = fa & fb
    d = (a3 - b3)[both]
    st = {"rmse": float(np.sqrt(np.mean(d * d))) if d.size else 0.0, "max": float(np.abs(d).max()) if d.size else 0.0,
          "frac_gt_1e-3": float((np.abs(d).max(axis=-1) > 1e-3).mean()) if d.size else 0.0,
          "nonfinite_mismatch": float((fa != fb).mean()), "nonfinite_both": float((~fa & ~fb).mean())}
    assert st["nonfinite_mismatch"] <= 1e-3, st
    return st


@pytest.fixture(scope="session")
def default_model():
    from raytracing_amd import scenes
    return scenes.bvh_default()


@pytest.fixture(scope="session")
def bunny0_model():
    from raytracing_amd import scenes
    return scenes.mesh_scene(subdivisions=0, builder="median")


@pytest.fixture(scope="session")
def pool_model():
    from raytracing_amd import scenes
    return scenes.pool()


@pytest.fixture(scope="session")
def tracer():
    from raytracing_amd import tracer as T
    t = T.Tracer(0)
    yield t
    t.close()
