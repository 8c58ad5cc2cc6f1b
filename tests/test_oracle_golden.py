"""The CPU oracle against the golden vectors: llvmpipe renders of the REFERENCE'S OWN shaders
(tests/golden/*.npz, made by tools/make_golden.py).  This is the pin that lets the oracle stand in
for the reference on the GPU box.  Bars (north_star): >= 99.99 % identical primary-hit primitive
ids; per-pixel RMSE <= 1e-5 (observed ~3e-8: the only differences are libm vs polynomial exp)."""
import numpy as np
import pytest

from conftest import image_stats, load_golden
from oracle import oracle_api as O
from raytracing_amd import abi, host, scenes

ID_BAR = 0.9999
RMSE_BAR = 1e-5


def check(model, name, jitter=None, decay=None):
    g = load_golden(name)
    trans = g["trans"]
    if jitter is not None:
        trans = host.sample_trans(trans, jitter)
    img, hits, cn = O.render(model.scene_view(), trans, g["width"], g["height"], max_depth=g["depth"], variant=g["variant"],
                             decay_origin=decay if decay is not None else (0, 0, 0, 1), threads=8)
    same = (hits == g["ids"]).all(axis=-1)
    st = image_stats(img, g["rgb"])
    assert same.mean() >= ID_BAR, (name, same.mean())
    assert st["rmse"] <= RMSE_BAR, (name, st)
    assert cn.raysPrimary == g["width"] * g["height"] * trans.shape[0]
    return st, same.mean()


def test_view_a(default_model):
    check(default_model, "bvh_viewA_256")


def test_view_b(default_model):
    check(default_model, "bvh_viewB_256")


def test_view_b_depth4(default_model):
    check(default_model, "bvh_viewB_192_depth4")


def test_view_b_spp4(default_model):
    g = load_golden("bvh_viewB_128_spp4")
    check(default_model, "bvh_viewB_128_spp4", jitter=g["jitter"])


def test_inside_glass_with_decay_origin(default_model):
    g = load_golden("bvh_inside_glass_128")
    d = O.decay_origin(default_model.scene_view(), g["trans"])
    gold = g["decay"].reshape(9, 4)[8]
    # the eye is inside box1 (n = 1.5, decay (0,0,-0.03)); probe directions come from a sin() hash of huge
    # arguments, so only the vote (not the 8 samples) is comparable between llvmpipe and libm
    np.testing.assert_allclose(d["decayOrigin"], gold, rtol=1e-6, atol=1e-7)
    check(default_model, "bvh_inside_glass_128", decay=tuple(float(x) for x in gold))


def test_decay_origin_in_air(default_model):
    z = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "bvh_decay_air.npz"))
    gold = z["decay"].reshape(9, 4)[8]
    scenes.overview_camera(default_model)
    d = O.decay_origin(default_model.scene_view(), default_model.camera(64))
    np.testing.assert_array_equal(gold, [0, 0, 0, 1])
    np.testing.assert_array_equal(d["decayOrigin"], [0, 0, 0, 1])


def test_mesh_bunny(bunny0_model):
    check(bunny0_model, "mesh_bunny0_240x135_depth4")


def test_triangle_and_circle_pre_bit_exact(default_model):
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "bvh_default_pre.npz"))
    tri_g = np.frombuffer(z["tri"].tobytes(), dtype=abi.TRIANGLE_GPU)
    cir_g = np.frombuffer(z["circles"].tobytes(), dtype=abi.CIRCLE)
    a = default_model.arrays()
    tri = O.triangle_pre(a["triangles"])
    for f in ("plane", "p1", "k1", "k2", "uv1", "uv2", "uv3"):
        assert np.array_equal(tri[f].view(np.uint32), tri_g[f].view(np.uint32)), f
    assert tri["color"].tobytes() == tri_g["color"].tobytes()
    cir = O.circle_pre(a["circles"])
    for f in ("plane", "sphere", "e1"):
        assert np.array_equal(cir[f].view(np.uint32), cir_g[f].view(np.uint32)), f


def test_triangle_pre_identities():
    # k1.e1 = 1, k1.e2 = 0, k2.e2 = 1, k2.e1 = 0, |n| = 1, plane contains p1
    rng = np.random.default_rng(7)
    t = np.zeros(500, dtype=abi.TRIANGLE)
    for f in ("p1", "p2", "p3"):
        t[f] = rng.uniform(-10, 10, size=(500, 3))
    g = O.triangle_pre(t)
    e1 = (t["p2"] - t["p1"]).astype(np.float64)
    e2 = (t["p3"] - t["p1"]).astype(np.float64)
    k1, k2 = g["k1"].astype(np.float64), g["k2"].astype(np.float64)
    scale = 1e-3
    assert np.abs((k1 * e1).sum(1) - 1).max() < scale and np.abs((k1 * e2).sum(1)).max() < scale
    assert np.abs((k2 * e2).sum(1) - 1).max() < scale and np.abs((k2 * e1).sum(1)).max() < scale
    n = g["plane"][:, :3].astype(np.float64)
    assert np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-5
    assert np.abs((n * t["p1"]).sum(1) + g["plane"][:, 3]).max() < 1e-4
