"""The CPU oracle against the golden vectors: llvmpipe renders of the REFERENCE'S OWN shaders
(tests/golden/*.npz, made by tools/make_golden.py).  This is the pin that lets the oracle stand in
for the reference on the GPU box.  Bars (north_star): >= 99.99 % identical primary-hit primitive
ids; per-pixel RMSE <= 1e-5 (observed ~3e-8: the only differences are libm vs polynomial exp)."""
import numpy as np
import pytest

import os

from conftest import GOLDEN, image_stats, load_golden
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


# ---- Pool variant of the shader (no checker, RayTraceDepth 8) on the Pool demo scene: water, textured walls ----
@pytest.mark.parametrize("name", ["pool_viewA_160", "pool_viewB_192"])
def test_pool_scene(pool_model, name):
    check(pool_model, name)


def test_pool_scene_counts(pool_model):
    v = pool_model.scene_view()
    assert v.num.triangleNum == 2 * 47 * 47 + 28 and v.num.pointLightNum == 1 and v.texLayers == 4


def test_pool_underwater_view_and_decay_origin(pool_model):
    """Eye under the water surface.  The image is pinned by llvmpipe with the decay origin llvmpipe computed, (0,0,0,1):
    on llvmpipe the Pool probe shader (direction exactly +z, Pool/shaders/DecayOriginPreCompute0.cpp:188-190) reports no
    hit at all, although the fragment shader's identical traversal finds the water triangle straight above the eye (checked
    with a 3x3 render).  The oracle's probe follows the shader text and finds water above the eye:
    decay -1 and n = 1.33 (triangle material of Pool/Pool.cpp:505-513), the analytic answer; that value is not pinned by
    llvmpipe."""
    g = load_golden("pool_underwater_128")
    gold = g["decay"].reshape(9, 4)[8]
    np.testing.assert_array_equal(gold, [0, 0, 0, 1])
    check(pool_model, "pool_underwater_128", decay=(0, 0, 0, 1))
    d = O.decay_origin(pool_model.scene_view(), g["trans"], variant=abi.VARIANT_POOL)
    np.testing.assert_allclose(d["decayOrigin"], [-1, -1, -1, 1.33], rtol=1e-6)


# ---- Pool water simulation (next row f2): the oracle against the reference's compute shaders on llvmpipe ----
def test_water_simulation_bit_exact(pool_model):
    """n x (WaterAcceleration, WaterPosition) + WaterPositionSync: z, v, a of every grid point and the z of every water
    triangle corner are bit-identical to llvmpipe's run of the reference shaders, after 1 and after 50 substeps."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "pool_water_steps.npz"))
    gx, gy, k, dt = int(g["gx"]), int(g["gy"]), float(g["k"]), float(g["dt"])
    w0 = scenes.pool_water_points(pool_model)
    np.testing.assert_array_equal(w0["z"].view(np.uint32), g["z0"].view(np.uint32))      # the scene's rest state is the fixture's input
    tris = pool_model.arrays()["triangles"]
    nw = 2 * (gx - 1) * (gy - 1)
    for n in (1, 50):
        w = O.water_run(w0, gx, gy, k, dt, n)
        zva = np.stack([w["z"], w["v"], w["a"]], axis=1)
        assert np.array_equal(zva.view(np.uint32), g[f"zva_{n}"].view(np.uint32)), n
        t = O.water_sync(w, gx, gy, tris)
        tz = np.stack([t["p1"][:nw, 2], t["p2"][:nw, 2], t["p3"][:nw, 2]], axis=1)
        assert np.array_equal(tz.view(np.uint32), g[f"tri_z_{n}"].view(np.uint32)), n
        assert np.array_equal(t[nw:].view(np.uint8), tris[nw:].view(np.uint8))
    # split runs compose: 20 + 30 substeps = 50
    w = O.water_run(O.water_run(w0, gx, gy, k, dt, 20), gx, gy, k, dt, 30)
    assert np.array_equal(w["z"].view(np.uint32), g["zva_50"][:, 0].view(np.uint32))


def test_c1_at_1024_ids_and_tile_sums():
    """BASELINE configs[0] at its own size (SURVEY 8c asks for the 1024^2 checksums beside the 256^2 golden): primary-hit ids of
    every pixel and the sums of the image over 16 x 16 tiles, from the reference's shaders on llvmpipe
    (tools/baseline_llvmpipe.py).  The oracle renders the frame on the CPU in a few seconds."""
    z = np.load(os.path.join(GOLDEN, "bvh_viewA_1024.npz"))
    from raytracing_amd import abi, scenes
    m = scenes.bvh_default()
    W = H = int(z["width"])
    tr = np.frombuffer(z["trans"].tobytes(), dtype=abi.TRANS)
    img, hits, _ = O.render(m.scene_view(), tr, W, H, max_depth=int(z["depth"]), threads=8)
    assert np.array_equal(hits[..., 0], z["hit_type"]) and np.array_equal(hits[..., 1], z["hit_idx"])
    T = int(z["tile"])
    sums = np.nan_to_num(img[..., :3].astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0).reshape(H // T, T, W // T, T, 3).sum(axis=(1, 3))
    err = np.abs(sums - z["tile_sums"])
    assert err.max() <= 1e-4 + 1e-6 * np.abs(z["tile_sums"]).max(), err.max()


# ---- the tree-free brute-force oracle (checks scenes too large for a host tree: C5) against the reference's own ids ----
@pytest.mark.parametrize("fixture,name", [("default_model", "bvh_viewB_256"), ("default_model", "bvh_viewA_256"), ("bunny0_model", "mesh_bunny0_240x135_depth4"),
                                          ("pool_model", "pool_viewB_192")])
def test_bruteforce_oracle_agrees_with_llvmpipe_ids(request, fixture, name):
    """oracle_bruteforce_primary tests every primitive against the primary rays of 512 scattered pixels, with no tree: its ids equal the
    ids llvmpipe's run of the reference's shader wrote for those pixels, except where two primitives are hit at exactly the same
    distance (the reference keeps the first one its traversal meets, the brute force the lowest index; the default scene has coincident
    discs and box faces), and its distances equal the tree-walking oracle's bit for bit."""
    model = request.getfixturevalue(fixture)
    g = load_golden(name)
    W, H = g["width"], g["height"]
    px = O.scattered_pixels(W, H, 512)
    assert len({(int(x), int(y)) for x, y in px}) == 512 and px[:, 0].max() < W and px[:, 1].max() < H
    hits, t = O.render_bruteforce(model.scene_view(), g["trans"], W, H, px, threads=8)
    gold = g["ids"][px[:, 1], px[:, 0]]
    same = (hits == gold).all(axis=-1)
    # the tree-walking oracle's primary hit distance for the same pixels (golden "t" where the fixture holds it)
    if "t" in g:
        gt = g["t"][px[:, 1], px[:, 0]]
        hit = gold[:, 0] != 0
        assert np.array_equal(t[hit].view(np.uint32), gt[hit].view(np.uint32)), np.abs(t[hit] - gt[hit]).max()
        assert same.mean() >= 0.97, same.mean()          # the rest: equal distances (checked by the line above), other primitive of the tie
    else:
        assert same.mean() >= ID_BAR, (name, same.mean())
    assert (hits[:, 0] != 0).sum() >= 64                 # the sample does look at the scene


def test_tie_prover_known_answers():
    """oracle_api.tie_adjusted_match on a scene made for it: two triangles in the plane z = -5 whose shared edge is laid through the point
    where the primary ray of pixel (40, 20) meets that plane, and whose outer corner is laid through the ray of pixel (12, 50).  {left,
    right} at the first pixel is a tie; {triangle, miss} at the second (a silhouette) is a tie; the same claims at a pixel deep inside a
    triangle are not, nor is a triangle claimed where the ray passes far outside it."""
    W = H = 64
    m = host.HostModel()
    m.add_triangles(np.zeros(1, dtype=abi.TRIANGLE))      # (a camera needs a model)
    m.camera_set(60.0, (0, 0, 0), np.eye(3))
    tr = m.camera(H)
    z0 = float(tr["z0"][0])

    def on_plane(x, y):                                   # where the ray of pixel (x, y) meets z = -5
        d = np.array([2 * (x + 0.5) - W, 2 * (y + 0.5) - H, z0])
        return d * (-5.0 / d[2])
    Q, S = on_plane(40, 20), on_plane(12, 50)
    e = np.array([0.3, 1.0, 0.0])
    a, b = Q - 1.5 * e, Q + 2.0 * e                       # the shared edge, through Q
    tri = np.zeros(2, dtype=abi.TRIANGLE)
    tri["p1"] = [a, a]; tri["p2"] = [b, S]; tri["p3"] = [S, b + (4, 0, 0)]      # triangle 0 = (a, b, S) with its corner at S; triangle 1 on the other side of a-b
    tri["p2"][1], tri["p3"][1] = b + np.array([4.0, 0, 0]), b
    for f in ("texR", "texT", "texD", "texG"):
        tri["color"][f] = -1
    tri["color"]["n"] = 1.0
    m2 = host.HostModel()
    m2.add_triangles(tri)
    m2.data_init("reference")
    view = m2.scene_view()
    t, sd, tol = O.tie_probe(view, tr, W, H, [(40, 20), (40, 20), (12, 50)], [(2, 0), (2, 1), (2, 0)])
    assert t[0] == t[1] and np.all(t > 5) and np.all(np.abs(sd) <= tol), (t, sd, tol)      # all three rays graze a boundary
    c0 = (np.asarray(tri["p1"][0]) + tri["p2"][0] + tri["p3"][0]) / 3                       # centroid of triangle 0 -> its pixel
    cx, cy = int((c0[0] / c0[2] * z0 + W) / 2), int((c0[1] / c0[2] * z0 + H) / 2)
    ti, sdi, toli = O.tie_probe(view, tr, W, H, [(cx, cy), (cx, cy)], [(2, 0), (2, 1)])
    assert sdi[0] < -0.1 and sdi[1] > 0.1                                                   # deep inside triangle 0, far outside triangle 1
    px = [(40, 20), (12, 50), (cx, cy), (cx, cy)]
    got = np.array([(2, 0), (2, 0), (2, 0), (2, 0)], dtype=np.uint32)
    want = np.array([(2, 1), (0, 0), (2, 1), (0, 0)], dtype=np.uint32)
    share, exact, bad = O.tie_adjusted_match(got, want, view, tr, W, H, pixels=px)
    assert exact == 0.0 and share == 0.5 and bad.tolist() == [[cx, cy], [cx, cy]], (share, exact, bad)


def test_tie_prover_on_the_default_scenes_coincident_primitives(default_model):
    """The default scene holds coincident discs and box faces that lie in the ground disc.  The oracle on a median-split tree meets them in
    another order than the reference's tree does, so its ids differ from llvmpipe's exactly there: every such pixel is proven a tie
    (equal distance), and the images agree."""
    g = load_golden("bvh_viewB_256")
    m = scenes.bvh_default(builder="median")
    img, hits, _ = O.render(m.scene_view(), g["trans"], g["width"], g["height"], max_depth=g["depth"], threads=8)
    share, exact, bad = O.tie_adjusted_match(hits, g["ids"], default_model.scene_view(), g["trans"], g["width"], g["height"])
    assert share >= ID_BAR and len(bad) == 0, (share, exact, bad[:8])
