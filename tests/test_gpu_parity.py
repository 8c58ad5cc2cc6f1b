"""GPU parity: the HIP path, called through the C ABI (liblrltrace.so), against
  (1) the golden vectors rendered by the reference's own shaders on llvmpipe, and
  (2) the CPU oracle on the same seeded inputs (random scenes, edge cases).
Bars (tests/conftest.py): primary-hit ids identical on >= 99.99 % of pixels; per-pixel RMSE <= 1e-4 in PARITY mode
(expression order of the reference on llvmpipe; differences come from expf only); in FAST mode (hoisted reciprocals,
fused multiply-add) at most 0.2 % of the pixels off by more than 1e-3 (floating-point tie edges) and RMSE <= 2e-5 over the rest.
Ray / node-visit / primitive-test counters must equal the oracle's when the images agree.
"""
import os

import numpy as np
import pytest

from conftest import FAST_RMSE, GOLDEN, ID_BAR, RMSE_PARITY, assert_ids, check_image, image_stats, load_golden
from oracle import oracle_api as O
from raytracing_amd import abi, host, scenes

pytestmark = pytest.mark.gpu



def sample0(trans, jitter):
    """the camera of sample 0 (the sample the hit ids belong to)"""
    return trans if jitter is None else host.sample_trans(trans, np.asarray(jitter, dtype=np.float32).reshape(-1, 2)[:1])


def run_golden(tracer, model, name, mode, jitter=None, decay=(0, 0, 0, 1)):
    g = load_golden(name)
    tracer.scene_upload(model.scene_view())
    tracer.set_camera(g["trans"], g["width"], g["height"])
    tracer.set_decay_origin(decay)
    img, hits = tracer.render(max_depth=g["depth"], variant=g["variant"], mode=mode, jitter=jitter)
    _, exact = assert_ids(hits, g["ids"], model.scene_view(), sample0(g["trans"], jitter), g["width"], g["height"], (name, mode))
    if mode == abi.MODE_PARITY:
        assert exact >= ID_BAR, (name, exact)          # the reference's arithmetic in the reference's visit order: no tie is resolved differently
    st = check_image(img, g["rgb"], mode, name)
    assert np.all(img[..., 3] == 1.0)
    return st


@pytest.mark.parametrize("mode", [abi.MODE_PARITY, abi.MODE_FAST])
@pytest.mark.parametrize("name", ["bvh_viewA_256", "bvh_viewB_256", "bvh_viewB_192_depth4"])
def test_default_scene_vs_llvmpipe_golden(tracer, default_model, name, mode):
    run_golden(tracer, default_model, name, mode)


@pytest.mark.parametrize("mode", [abi.MODE_PARITY, abi.MODE_FAST])
def test_spp4_jitter_vs_llvmpipe_golden(tracer, default_model, mode):
    g = load_golden("bvh_viewB_128_spp4")
    run_golden(tracer, default_model, "bvh_viewB_128_spp4", mode, jitter=g["jitter"])


@pytest.mark.parametrize("mode", [abi.MODE_PARITY, abi.MODE_FAST])
def test_bunny_mesh_vs_llvmpipe_golden(tracer, bunny0_model, mode):
    run_golden(tracer, bunny0_model, "mesh_bunny0_240x135_depth4", mode)


def test_inside_glass_decay_origin(tracer, default_model):
    g = load_golden("bvh_inside_glass_128")
    gold = g["decay"].reshape(9, 4)[8]
    tracer.scene_upload(default_model.scene_view())
    tracer.set_camera(g["trans"], g["width"], g["height"])
    d = tracer.compute_decay_origin(abi.VARIANT_BVH)
    np.testing.assert_allclose(d["decayOrigin"], gold, rtol=1e-6, atol=1e-7)
    img, hits = tracer.render(max_depth=10)          # uses the decay origin just computed
    st = image_stats(img, g["rgb"])
    assert (hits == g["ids"]).all(axis=-1).mean() >= ID_BAR and st["rmse"] <= RMSE_PARITY, st


def test_precompute_kernels_bit_exact(tracer, default_model):
    z = np.load(os.path.join(GOLDEN, "bvh_default_pre.npz"))
    tri_g = np.frombuffer(z["tri"].tobytes(), dtype=abi.TRIANGLE_GPU)
    cir_g = np.frombuffer(z["circles"].tobytes(), dtype=abi.CIRCLE)
    tracer.scene_upload(default_model.scene_view())
    tri = tracer.read_triangles_gpu(0, len(tri_g))
    for f in ("plane", "p1", "k1", "k2", "uv1", "uv2", "uv3"):
        assert np.array_equal(tri[f].view(np.uint32), tri_g[f].view(np.uint32)), f
    assert tri["color"].tobytes() == tri_g["color"].tobytes()
    cir = tracer.read_circles(0, len(cir_g))
    for f in ("plane", "sphere", "e1"):
        assert np.array_equal(cir[f].view(np.uint32), cir_g[f].view(np.uint32)), f


@pytest.mark.parametrize("scene", ["default", "pool"])
def test_scene_given_as_precomputed_triangles(tracer, default_model, pool_model, scene):
    """LrlSceneView.trianglesGPU: the caller hands over TrianglePre's OUTPUT (the reference's SSBO 2 as its compute pass leaves it,
    BVH/shaders/TrianglePreCompute0.cpp:55-80) instead of the origin records.  The frames are the frames of the origin upload bit for bit,
    in both arithmetic modes; without the corners nothing tells the library where a triangle's own box is, so a caller's tree is walked
    with the caller's boxes (ownBoxes 0) where the origin upload fits it."""
    m, gname, variant, depth = ((default_model, "bvh_viewB_256", abi.VARIANT_BVH, 6) if scene == "default" else (pool_model, "pool_viewB_192", abi.VARIANT_POOL, 8))
    g = load_golden(gname)
    W, H = 160, 160
    view = m.scene_view()
    out = {}
    tracer.scene_upload(view)
    fitted = tracer.scene_info()["ownBoxes"]
    tri_gpu = tracer.read_triangles_gpu(0, view.num.triangleNum)
    for how in ("origin", "gpu"):
        v = m.scene_view()
        if how == "gpu":
            v.trianglesOrigin = None
            v.trianglesGPU = tri_gpu.ctypes.data
        tracer.scene_upload(v)
        tracer.set_camera(g["trans"], W, H)
        tracer.set_decay_origin((0, 0, 0, 1))
        out[how] = (tracer.render(max_depth=depth, variant=variant), tracer.render(max_depth=depth, variant=variant, mode=abi.MODE_FAST), tracer.scene_info()["ownBoxes"])
        back = tracer.read_triangles_gpu(0, view.num.triangleNum)
        assert back.tobytes() == tri_gpu.tobytes()
    assert out["origin"][2] == fitted and out["gpu"][2] == 0
    for k in (0, 1):
        assert np.array_equal(out["origin"][k][1], out["gpu"][k][1])
        assert np.array_equal(out["origin"][k][0].view(np.uint32), out["gpu"][k][0].view(np.uint32))
    assert (out["gpu"][0][1][..., 0] != 0).mean() > 0.2                       # the scene is in view


def test_counters_equal_oracle(tracer, default_model):
    g = load_golden("bvh_viewB_192_depth4")
    tracer.scene_upload(default_model.scene_view())
    tracer.set_camera(g["trans"], g["width"], g["height"])
    tracer.set_decay_origin((0, 0, 0, 1))
    img, hits = tracer.render(max_depth=4, counters=True)
    dev = tracer.counters().as_dict()
    oimg, ohits, cn = O.render(default_model.scene_view(), g["trans"], g["width"], g["height"], max_depth=4, threads=8)
    ref = cn.as_dict()
    assert dev["raysPrimary"] == ref["raysPrimary"] == g["width"] * g["height"]
    assert dev["primaryHits"] == ref["primaryHits"]
    # expf differs in the last bit between libm and the device library; a handful of paths may cross the 0.01 cut-off
    for k in ref:
        assert abs(dev[k] - ref[k]) <= max(8, 2e-4 * ref[k]), (k, dev[k], ref[k])
    # images with and without the counting kernel are the same kernel otherwise
    img2, _ = tracer.render(max_depth=4, counters=False)
    assert np.array_equal(img, img2)


def random_scene(seed, n_tri=300, n_sph=12):
    rng = np.random.default_rng(seed)
    m = host.HostModel()
    t = np.zeros(n_tri, dtype=abi.TRIANGLE)
    c = rng.uniform(-12, 12, size=(n_tri, 3)) * (1, 1, 0.4) + (0, 0, 5)
    t["p1"] = c + rng.uniform(-2, 2, size=(n_tri, 3))
    t["p2"] = c + rng.uniform(-2, 2, size=(n_tri, 3))
    t["p3"] = c + rng.uniform(-2, 2, size=(n_tri, 3))
    t["uv2"], t["uv3"] = (1, 0), (0, 1)
    kinds = rng.integers(0, 3, size=n_tri)
    for i in range(n_tri):
        t["color"][i] = (abi.color(d=rng.uniform(0.2, 1, 3)), abi.color(r=0.7), abi.color(r=1, t=1, decay=(0, 0, -0.05), n=1.4))[kinds[i]]
    m.add_triangles(t)
    s = np.zeros(n_sph, dtype=abi.SPHERE)
    s["sphere"][:, :3] = rng.uniform(-10, 10, size=(n_sph, 3)) * (1, 1, 0.3) + (0, 0, 4)
    s["sphere"][:, 3] = rng.uniform(0.5, 4, size=n_sph)
    s["e1"], s["e2"] = (0, 0, 1), (1, 0, 0)
    for i in range(n_sph):
        s["color"][i] = (abi.color(d=rng.uniform(0.2, 1, 3)), abi.color(r=0.8), abi.color(r=1, t=1, decay=(-0.1, -0.1, 0), n=1.5))[i % 3]
    m.add_spheres(s)
    g = np.zeros(1, dtype=abi.CIRCLE)
    g["plane"], g["sphere"], g["e1"] = (0, 0, 2, 0), (0, 0, 0, 900), (3, 0, 0)      # un-normalised on purpose: CirclePre
    g["color"] = abi.color(d=1.0, g=0.05, texD=-2)
    m.add_circles(g)
    cy = np.zeros(1, dtype=abi.CYLINDER)
    cy["c"], cy["r2"], cy["n"], cy["l"], cy["e1"], cy["color"] = (6, -6, 0.5), 2.0, (0, 0, 1), 3.0, (1, 0, 0), abi.color(r=1, t=1, n=1.5)
    m.add_cylinder(cy)
    co = np.zeros(1, dtype=abi.CONE)
    co["c"], co["c2"], co["n"], co["l2"], co["e1"], co["color"] = (-6, -6, 5), 0.75, (0, 0, -1), 16.0, (1, 0, 0), abi.color(d=(0.9, 0.3, 0.2))
    m.add_cone(co)
    L = np.zeros(2, dtype=abi.POINTLIGHT)
    L["color"], L["p"] = [(300, 300, 300), (60, 50, 40)], [(5, -10, 40), (-15, 5, 12)]
    m.add_point_lights(L)
    faces = rng.integers(0, 256, size=(6, 16, 16, 3), dtype=np.uint8)
    m.set_cube_raw(faces)
    tex = rng.integers(0, 256, size=(2, 8, 8, 3), dtype=np.uint8)
    m.set_tex_raw(tex)
    return m


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("builder", ["reference", "median", "sah"])
def test_random_scenes_vs_oracle(tracer, seed, builder):
    m = random_scene(seed)
    m.data_init(builder)
    m.camera_look_at(55.0, (18, -22, 16), (0, 0, 3))
    W, H = 160, 120
    tr = m.camera(H)
    v = m.scene_view()
    tracer.scene_upload(v)
    tracer.set_camera(tr, W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    oimg, ohits, cn = O.render(v, tr, W, H, max_depth=6, threads=8)
    for mode in (abi.MODE_PARITY, abi.MODE_FAST):
        img, hits = tracer.render(max_depth=6, mode=mode)
        same = (hits == ohits).all(axis=-1).mean()
        assert same >= ID_BAR, (seed, builder, mode, same)
        check_image(img, oimg, mode, ("random scene", seed, builder))


@pytest.mark.parametrize("n", [1, 2, 3, 5])
def test_tiny_scenes_vs_oracle(tracer, n):
    """A caller's tree of one node (its root is a leaf: callerLeafRootKernel), of three, of a few: the smallest inputs of the device-side
    repack / fit / layout, in both modes, against the oracle."""
    m = host.HostModel()
    t = np.zeros(n, dtype=abi.TRIANGLE)
    for i in range(n):
        x = 3.0 * i - 1.5 * (n - 1)
        t["p1"][i], t["p2"][i], t["p3"][i] = (x - 1.2, -1.0, 0.2 * i), (x + 1.2, -1.0, 0.2 * i), (x, 1.4, 0.5 + 0.2 * i)
        t["color"][i] = abi.color(d=(0.9, 0.5 + 0.1 * (i % 3), 0.3)) if i % 2 == 0 else abi.color(r=0.6, d=(0.2, 0.2, 0.6))
    t["uv2"], t["uv3"] = (1, 0), (0, 1)
    m.add_triangles(t)
    L = np.zeros(1, dtype=abi.POINTLIGHT)
    L["color"], L["p"] = [(200, 200, 200)], [(2, -3, 12)]
    m.add_point_lights(L)
    m.data_init("reference")
    m.camera_look_at(50.0, (1.0, -2.0, 14.0), (0, 0, 0))
    W, H = 96, 64
    tr = m.camera(H)
    v = m.scene_view()
    tracer.scene_upload(v)
    tracer.set_camera(tr, W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    oimg, ohits, cn = O.render(v, tr, W, H, max_depth=4, threads=4)
    assert (ohits[..., 0] != 0).any()                    # the camera does see them
    for mode in (abi.MODE_PARITY, abi.MODE_FAST):
        img, hits = tracer.render(max_depth=4, mode=mode)
        assert np.array_equal(hits, ohits), (n, mode, int((hits != ohits).any(axis=-1).sum()))
        check_image(img, oimg, mode, ("tiny scene", n))
        img1, hits1 = tracer.render(max_depth=4, mode=mode, jitter=host.jitter_table(3))      # a batch of samples over the same tree
        assert np.isfinite(img1).all()
    if n == 3:
        # no material of these triangles asks for texture coordinates, so the shade kernels take normal and material from the hot record
        # alone (KScene::triNeedUV); an update that gives one of them a texture layer must bring the coordinates back
        rng = np.random.default_rng(5)
        tex = rng.integers(0, 256, size=(1, 8, 8, 3), dtype=np.uint8)
        t2 = m.arrays()["triangles"].copy()
        t2["color"][0] = abi.color(d=(1.0, 1.0, 1.0), texD=0)
        models = []
        for tris in (t, t2):
            mm = host.HostModel()
            mm.add_triangles(tris)
            mm.add_point_lights(L)
            mm.set_tex_raw(tex)
            mm.data_init("reference")
            models.append(mm)
        v0, v2 = models[0].scene_view(), models[1].scene_view()
        o2, oh2, _ = O.render(v2, tr, W, H, max_depth=4, threads=4)
        assert not np.array_equal(o2, oimg)
        tracer.scene_upload(v0)                           # the untextured triangles in a scene that has a texture layer
        tracer.set_decay_origin((0, 0, 0, 1))
        for tris, want, what in ((None, oimg, "untextured"), (t2, o2, "textured by an update"), (t, oimg, "untextured again"), (t2, o2, "textured again")):
            if tris is not None:
                tracer.update_triangles(0, tris)
            for mode in (abi.MODE_PARITY, abi.MODE_FAST):
                img, hits = tracer.render(max_depth=4, mode=mode)
                assert np.array_equal(hits, ohits)
                check_image(img, want, mode, ("tiny scene", what))


def test_textured_triangles_pool_variant_vs_oracle(tracer):
    # texture array on glow / diffuse / reflect / transmit slots, uv beyond 1 (REPEAT), Pool variant (no checker)
    rng = np.random.default_rng(11)
    m = host.HostModel()
    t = np.zeros(4, dtype=abi.TRIANGLE)
    quad = [(-8, -8, 0), (8, -8, 0), (8, 8, 0), (-8, 8, 0)]
    wall = [(-8, 8, 0), (8, 8, 0), (8, 8, 9), (-8, 8, 9)]
    for k, q in enumerate((quad, wall)):
        t["p1"][2 * k], t["p2"][2 * k], t["p3"][2 * k] = q[0], q[1], q[2]
        t["uv1"][2 * k], t["uv2"][2 * k], t["uv3"][2 * k] = (0, 0), (5, 0), (5, 5)
        t["p1"][2 * k + 1], t["p2"][2 * k + 1], t["p3"][2 * k + 1] = q[0], q[2], q[3]
        t["uv1"][2 * k + 1], t["uv2"][2 * k + 1], t["uv3"][2 * k + 1] = (0, 0), (5, 5), (0, 5)
    t["color"][0:2] = abi.color(d=1.0, texD=1, g=0.2, texG=0, r=0.3, texR=2)
    t["color"][2:4] = abi.color(d=0.8, texD=-2, g=0.3, texG=-2)      # checker codes must be ignored by the Pool variant
    m.add_triangles(t)
    s = np.zeros(1, dtype=abi.SPHERE)
    s["sphere"], s["e1"], s["e2"] = (0, 0, 3, 4), (0, 0, 1), (1, 0, 0)
    s["color"] = abi.color(r=1, t=1, texT=1, n=1.33, d=0.2, texD=0)
    m.add_spheres(s)
    L = np.zeros(1, dtype=abi.POINTLIGHT)
    L["color"], L["p"] = (80, 80, 80), (2, -6, 12)
    m.add_point_lights(L)
    m.set_tex_raw(rng.integers(0, 256, size=(3, 16, 32, 3), dtype=np.uint8))
    m.set_cube_raw(rng.integers(0, 256, size=(6, 8, 8, 3), dtype=np.uint8))
    m.data_init()
    m.camera_look_at(60.0, (3, -14, 7), (0, 2, 2))
    W, H = 128, 96
    tr = m.camera(H)
    v = m.scene_view()
    tracer.scene_upload(v)
    tracer.set_camera(tr, W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    for variant in (abi.VARIANT_POOL, abi.VARIANT_BVH):
        oimg, ohits, _ = O.render(v, tr, W, H, max_depth=8, variant=variant, threads=8)
        img, hits = tracer.render(max_depth=8, variant=variant)
        assert (hits == ohits).all(axis=-1).mean() >= ID_BAR
        st = image_stats(img, oimg)
        assert st["rmse"] <= RMSE_PARITY, (variant, st)


def test_row_stripes_and_sample_ranges_compose(tracer, default_model):
    # multi-GPU sharding units: row stripes are bit-identical to the full frame; sample ranges sum to the mean
    g = load_golden("bvh_viewB_128_spp4")
    tracer.scene_upload(default_model.scene_view())
    tracer.set_camera(g["trans"], g["width"], g["height"])
    tracer.set_decay_origin((0, 0, 0, 1))
    full, fh = tracer.render(max_depth=10)
    parts = [tracer.render(max_depth=10, rows=r) for r in ((0, 40), (40, 41), (41, 128))]
    assert np.array_equal(np.concatenate([p[0] for p in parts]), full)
    assert np.array_equal(np.concatenate([p[1] for p in parts]), fh)
    jit = g["jitter"]
    mean, _ = tracer.render(max_depth=10, jitter=jit)
    a, _ = tracer.render(max_depth=10, jitter=jit, samples=(0, 3))
    b, _ = tracer.render(max_depth=10, jitter=jit, samples=(3, 4))
    np.testing.assert_allclose(a[..., :3] + b[..., :3], mean[..., :3], rtol=0, atol=2e-6)


def test_edge_cases(tracer):
    from raytracing_amd.tracer import TraceError
    # empty scene: every pixel is the (absent) environment = 0
    m = host.HostModel()
    m.data_init()
    tracer.scene_upload(m.scene_view())
    tracer.set_camera(m.camera(20), 33, 20)           # ragged: not a multiple of the 16x16 tile
    tracer.set_decay_origin((0, 0, 0, 1))
    img, hits = tracer.render(max_depth=10)
    assert img.shape == (20, 33, 4) and np.all(img[..., :3] == 0) and np.all(hits == 0)
    # limits are refused with an error, not a fault
    with pytest.raises(TraceError):
        tracer.render(max_depth=11)
    with pytest.raises(TraceError):
        tracer.render(max_depth=4, rows=(10, 40))
    # malformed BVH is refused at upload
    m2 = host.HostModel()
    t = np.zeros(2, dtype=abi.TRIANGLE)
    t["p2"], t["p3"] = (1, 0, 0), (0, 1, 0)
    t["p1"][1], t["p2"][1], t["p3"][1] = (5, 5, 5), (6, 5, 5), (5, 6, 5)
    m2.add_triangles(t)
    m2.data_init()
    v = m2.scene_view()
    arr = m2.arrays()["bvh"].copy()
    arr["rightChild"][0] = 99
    v.bvh = arr.ctypes.data
    with pytest.raises(TraceError):
        tracer.scene_upload(v)


def test_full_size_properties_1080p(tracer, bunny0_model):
    # BASELINE config size (1920x1080, 4 bounces) on the base bunny: size-independent properties only
    m = bunny0_model
    W, H = 1920, 1080
    tr = m.camera(H)
    tracer.scene_upload(m.scene_view())
    tracer.set_camera(tr, W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    img, hits = tracer.render(max_depth=4, counters=True)
    c = tracer.counters()
    assert c.raysPrimary == W * H and np.isfinite(img).all() and (img[..., :3] >= 0).all()
    assert c.primaryHits == int((hits[..., 0] != 0).sum())
    # idempotence and stripe-invariance at full size
    img2, _ = tracer.render(max_depth=4)
    assert np.array_equal(img, img2)
    stripe, _ = tracer.render(max_depth=4, rows=(512, 560))
    assert np.array_equal(stripe, img[512:560])
    # oracle on a 48-row stripe of the same frame
    oimg, ohits, _ = O.render(m.scene_view(), tr, W, H, max_depth=4, rows=(512, 560), threads=8)
    assert (hits[512:560] == ohits).all(axis=-1).mean() >= ID_BAR
    assert image_stats(stripe, oimg)["rmse"] <= RMSE_PARITY


def test_headline_workload_properties(tracer):
    """bench.py's workload itself (C2: 1 240 960 triangles, 1920x1080, 4 bounces), through properties that need no oracle
    run of the whole frame: FAST against PARITY (ids, image, ray count), the reference builder's tree against the tree built
    on the device, idempotence, and the oracle on a 16-row stripe through the middle of the bunny."""
    m = scenes.mesh_scene(subdivisions=2, builder="reference")
    W, H = 1920, 1080
    tr = m.camera(H)
    view = m.scene_view()
    res = {}
    for dev in (False, True):
        tracer.scene_upload(without_tree(m) if dev else view)
        tracer.set_camera(tr, W, H)
        tracer.set_decay_origin((0, 0, 0, 1))
        for mode in (abi.MODE_PARITY, abi.MODE_FAST):
            img, hits = tracer.render(max_depth=4, mode=mode, counters=(mode == abi.MODE_PARITY))
            res[dev, mode] = (img, hits, tracer.counters().rays if mode == abi.MODE_PARITY else None)
            assert np.isfinite(img).all() and (img[..., :3] >= 0).all() and np.all(img[..., 3] == 1.0)
        again, _ = tracer.render(max_depth=4, mode=abi.MODE_FAST)
        assert np.array_equal(again, res[dev, abi.MODE_FAST][0])
    base_img, base_hits, base_rays = res[False, abi.MODE_PARITY]
    assert base_rays > 3 * W * H                      # reflections and shadow rays are there
    for key, (img, hits, rays) in res.items():
        if key == (False, abi.MODE_PARITY):
            continue
        same = (hits == base_hits).all(axis=-1).mean()
        assert same >= ID_BAR, (key, same)
        check_image(img, base_img, abi.MODE_FAST, ("headline workload vs PARITY on the reference tree", key))
        if rays is not None:
            assert abs(rays - base_rays) <= 1e-4 * base_rays, key
    r0 = 470
    oimg, ohits, _ = O.render(view, tr, W, H, max_depth=4, rows=(r0, r0 + 16), threads=8)
    assert (base_hits[r0:r0 + 16] == ohits).all(axis=-1).mean() >= ID_BAR
    assert image_stats(base_img[r0:r0 + 16], oimg)["rmse"] <= RMSE_PARITY


def test_diffuse_bounce_vs_oracle(tracer, default_model, bunny0_model):
    """LRL_FLAG_DIFFUSE_BOUNCE (incoherent secondary rays; no reference counterpart, so the oracle's own
    restatement is the truth, pinned on the CPU by Philox's known answer and the sampler's distribution).  PARITY computes
    the same directions bit for bit, so image and ray counts meet the bars of the mirror mode; FAST contracts
    multiply-adds, so a direction may move by an ulp and a bounce near an edge may land elsewhere: image bar only."""
    for m, (W, H), depth in ((default_model, (96, 96), 4), (bunny0_model, (160, 90), 4)):
        tr = m.camera(H)
        view = m.scene_view()
        tracer.scene_upload(view)
        tracer.set_camera(tr, W, H)
        tracer.set_decay_origin((0, 0, 0, 1))
        oimg, ohits, oc = O.render(view, tr, W, H, max_depth=depth, threads=8, diffuse=True)
        img, hits = tracer.render(max_depth=depth, mode=abi.MODE_PARITY, counters=True, diffuse=True)
        c = tracer.counters()
        assert (hits == ohits).all(axis=-1).mean() >= ID_BAR
        st = image_stats(img, oimg)
        assert st["rmse"] <= RMSE_PARITY and st["frac_gt_1e-3"] <= 2e-3, st
        assert abs(c.rays - oc.rays) <= 2e-3 * oc.rays and c.raysPrimary == oc.raysPrimary
        mirror, _ = tracer.render(max_depth=depth, mode=abi.MODE_PARITY)
        assert not np.array_equal(mirror, img)
        fast, fhits = tracer.render(max_depth=depth, mode=abi.MODE_FAST, diffuse=True)
        assert (fhits == ohits).all(axis=-1).mean() >= ID_BAR
        # a diffuse direction that moves by an ulp may send the path elsewhere: more pixels move than at mirror ties
        check_image(fast, oimg, abi.MODE_FAST, ("diffuse bounces", W, H), frac_bar=1e-2)
        # stripes and sample ranges compose (the stream is keyed by pixel and sample of the full frame)
        stripe, _ = tracer.render(max_depth=depth, mode=abi.MODE_PARITY, rows=(32, 48), diffuse=True)
        assert np.array_equal(stripe, img[32:48])
    with pytest.raises(Exception):
        tracer.render(max_depth=4, diffuse=True, megakernel=True)


def chain_scene(n):
    """n mirror triangles in a row under a degenerate caller-made tree: inner node k holds triangle k on the left and the
    rest of the row on the right, so the tree is n levels deep (the reference's mean split makes such chains from
    geometric progressions of sizes; here it is made by hand in the reference's 64-B pre-order layout)."""
    m = host.HostModel()
    t = np.zeros(n, dtype=abi.TRIANGLE)
    x = np.arange(n, dtype=np.float32) * 0.5
    zero = np.zeros(n, dtype=np.float32)
    h = np.float32(0.15 * n)               # tall slats: the row fills a good part of the view
    t["p1"] = np.stack([x, zero - h, zero], axis=1)
    t["p2"] = np.stack([x + 0.45, zero - h, zero + 0.1], axis=1)
    t["p3"] = np.stack([x + 0.2, zero + h, zero], axis=1)
    t["uv2"], t["uv3"] = (1, 0), (0, 1)
    for i in range(n):
        t["color"][i] = abi.color(d=(0.3 + 0.7 * (i % 3 == 0), 0.3 + 0.7 * (i % 3 == 1), 0.3 + 0.7 * (i % 3 == 2)), r=0.5)
    m.add_triangles(t)
    g = np.zeros(1, dtype=abi.CIRCLE)      # a mirror floor behind the row, so that reflected rays come back through the chain
    g["plane"], g["sphere"], g["e1"] = (0, 0, 1, 2), (n * 0.25, 0, -2, (n * 0.5) ** 2), (1, 0, 0)
    g["color"] = abi.color(d=0.5, r=0.8)
    m.add_circles(g)
    L = np.zeros(1, dtype=abi.POINTLIGHT)
    L["color"], L["p"] = (400, 400, 400), (n * 0.25, -3, 25)
    m.add_point_lights(L)
    m.set_cube_raw(np.full((6, 4, 4, 3), 90, dtype=np.uint8))
    m.data_init("reference")               # a valid view (counts, leaf boxes); its tree is replaced below
    m.camera_look_at(50.0, (n * 0.25, -0.5 * n - 4, 0.45 * n + 4), (n * 0.25, 0, 0))
    v = m.scene_view()
    lo = np.minimum(np.minimum(t["p1"], t["p2"]), t["p3"])
    hi = np.maximum(np.maximum(t["p1"], t["p2"]), t["p3"])
    ref = abi.scene_view_arrays(v)["bvh"]
    disc = ref[ref["geometry"] == 4][0]                     # the floor's leaf box, as the host builder made it
    nodes = np.zeros(2 * (n + 1) - 1, dtype=abi.BVHNODE)
    # root: left = the floor disc, right = the chain over the triangles
    suffix_lo = np.minimum.accumulate(lo[::-1], axis=0)[::-1]
    suffix_hi = np.maximum.accumulate(hi[::-1], axis=0)[::-1]
    nodes[0]["min"], nodes[0]["max"] = np.minimum(disc["min"], suffix_lo[0]), np.maximum(disc["max"], suffix_hi[0])
    nodes[0]["leftChild"], nodes[0]["rightChild"], nodes[0]["axis"] = 1, 2, 2
    nodes[1] = disc
    nodes[1]["father"], nodes[1]["leftChild"], nodes[1]["rightChild"] = 0, 0, 0
    at = 2
    father = 0
    for k in range(n - 1):
        nodes[at]["min"], nodes[at]["max"] = suffix_lo[k], suffix_hi[k]
        nodes[at]["leftChild"], nodes[at]["rightChild"], nodes[at]["father"], nodes[at]["axis"] = 1, at + 2, father, 0
        leaf = nodes[at + 1]
        leaf["min"], leaf["max"], leaf["father"], leaf["geometry"], leaf["geometryNum"] = lo[k], hi[k], at, 2, k
        father = at
        at += 2
    last = nodes[at]
    last["min"], last["max"], last["father"], last["geometry"], last["geometryNum"] = lo[n - 1], hi[n - 1], father, 2, n - 1
    nodes = nodes[:at + 1]
    chain_scene.keep = nodes
    v.bvh = nodes.ctypes.data
    v.bvhNodeNum = len(nodes)
    return m, v


def test_malformed_trees_are_refused(tracer, pool_model):
    """lrl_scene_upload validates a caller's tree on the host before any kernel indexes with it: a child index out of range or not later
    in pre-order, a right child without a left one, an unknown primitive class, a primitive number beyond its array, and a node with two
    parents (the device passes give every node one place in the packed tree) are all refused with LRL_ERR_INVALID_ARG, and the context
    renders the next valid scene as if nothing had happened."""
    from raytracing_amd.tracer import TraceError
    g = load_golden("pool_viewB_192")
    view = pool_model.scene_view()
    good = pool_model.arrays()["bvh"]
    inner = np.flatnonzero((good["leftChild"] != 0) & (good["rightChild"] != 0))
    leaves = np.flatnonzero((good["leftChild"] == 0) & (good["geometry"] == 2))

    def broken(edit):
        b = good.copy()
        edit(b)
        return b

    def two_parents(b):                                   # two inner nodes name the same right child
        c = int(inner[40])
        d = next(int(x) for x in inner if int(x) != c and int(b["rightChild"][x]) > c + 1)
        b["rightChild"][c] = b["rightChild"][d]

    cases = {
        "right child past the end": lambda b: b["rightChild"].__setitem__(int(inner[5]), len(b) + 7),
        "right child not later in pre-order": lambda b: b["rightChild"].__setitem__(int(inner[5]), int(inner[5])),
        "right child without left": lambda b: (b["leftChild"].__setitem__(int(leaves[9]), 0), b["rightChild"].__setitem__(int(leaves[9]), len(b) - 1)),
        "left child past the end": lambda b: b["leftChild"].__setitem__(len(b) - 1, 1),
        "unknown primitive class": lambda b: b["geometry"].__setitem__(int(leaves[3]), 9),
        "primitive number beyond its array": lambda b: b["geometryNum"].__setitem__(int(leaves[3]), view.num.triangleNum),
        "two parents": two_parents,
        "axis": lambda b: b["axis"].__setitem__(int(inner[2]), 3),
    }
    for what, edit in cases.items():
        bad = broken(edit)
        v = pool_model.scene_view()
        v.bvh = bad.ctypes.data
        with pytest.raises(TraceError) as e:
            tracer.scene_upload(v)
        assert e.value.code == 1, (what, e.value.code)     # LRL_ERR_INVALID_ARG
    tracer.scene_upload(view)
    tracer.set_camera(g["trans"], g["width"], g["height"])
    tracer.set_decay_origin((0, 0, 0, 1))
    img, hits = tracer.render(max_depth=8, variant=abi.VARIANT_POOL)
    assert (hits == g["ids"]).all(axis=-1).mean() >= ID_BAR


@pytest.mark.parametrize("n", [60, 92, 140])
def test_deep_chain_tree(tracer, n):
    """Trees far deeper than the LDS stacks: 61 and 93 levels (the traversal stacks spill to their global overflow, the
    hand-over records are 62 and 95 refs long: the deeper one nearly fills a 16-lane group's share of the frontier, so its
    rounds run depth-first), and 141 levels: beyond LRL_MAX_BVH_DEPTH, refused with its own error code.  PARITY against the oracle, FAST against both; then with a
    hand-over after 4 trips so that the continuation kernels do most of the work."""
    m, v = chain_scene(n)
    if n > 95:
        from raytracing_amd.tracer import TraceError
        with pytest.raises(TraceError) as e:
            tracer.scene_upload(v)
        assert e.value.code == 4                           # LRL_ERR_BVH_DEPTH
        return
    W, H = 128, 96
    tr = m.camera(H)
    tracer.scene_upload(v)
    tracer.set_camera(tr, W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    # the oracle keeps the shader's trail in 64-bit masks (the shader: 32), so beyond 64 levels it walks the same scene
    # under the host builder's own shallow tree: the picture does not depend on the tree
    oimg, ohits, oc = O.render(v if n <= 60 else m.scene_view(), tr, W, H, max_depth=4, threads=8)
    assert (ohits[..., 0] == 2).mean() > 0.05              # the row is in view
    img, hits = tracer.render(max_depth=4, mode=abi.MODE_PARITY, counters=True)
    assert (hits == ohits).all(axis=-1).mean() >= ID_BAR and image_stats(img, oimg)["rmse"] <= RMSE_PARITY
    if n <= 60:
        assert tracer.counters().nodeVisits == oc.nodeVisits
    assert tracer.counters().rays == oc.rays
    base = tracer.get_tuning()
    try:
        for early in (False, True):
            if early:
                tracer.set_tuning(adapt=0, handoverList=[4])
            fast, fhits = tracer.render(max_depth=4, mode=abi.MODE_FAST)
            assert (fhits == ohits).all(axis=-1).mean() >= ID_BAR, early
            check_image(fast, oimg, abi.MODE_FAST, ("deep chain", n, early))
    finally:
        tracer.set_tuning(base)                             # thresholds back to their defaults for the tests that follow


def test_compacted_ray_lists_change_nothing(tracer, default_model, bunny0_model):
    """FAST mode may trace the late segments from compacted lists of the live rays (on by default after diffuse bounces and
    for scenes beyond the caches; LrlTuning.compactFrom forces it).  A ray's result does not depend on the lane that carries it:
    the frames with and without compaction are the same bit for bit, mirror and diffuse."""
    base = tracer.get_tuning()
    try:
        for m, (W, H) in ((default_model, (200, 160)), (bunny0_model, (320, 180))):
            tr = m.camera(H)
            tracer.scene_upload(m.scene_view())
            tracer.set_camera(tr, W, H)
            tracer.set_decay_origin((0, 0, 0, 1))
            for diffuse in (False, True):
                frames = {}
                for start in (0, 1, 3):
                    tracer.set_tuning(compactFrom=start)
                    frames[start] = tracer.render(max_depth=6, mode=abi.MODE_FAST, diffuse=diffuse)
                for start in (1, 3):
                    assert np.array_equal(frames[start][0], frames[0][0]), (diffuse, start)
                    assert np.array_equal(frames[start][1], frames[0][1]), (diffuse, start)
    finally:
        tracer.set_tuning(base)


def test_the_tail_of_a_pass_in_one_launch_changes_nothing(tracer, default_model, bunny0_model, pool_model):
    """LrlTuning.tailFrom: from that iteration on one launch finishes the pass, every workgroup tracing and shading its own 64 path slots
    (wfTailKernel; on by default from the sixth segment where a material transmits).  The same bodies run in the same order per path:
    the frames with the tail starting anywhere, and without it, are the same bit for bit -- glass in glass (the default scene, depth 10),
    the Pool's water with a batch of samples, a scene without transmission (exact bound: no look at the device), diffuse bounces."""
    TAIL_NEVER = 0xffffffff
    base = tracer.get_tuning()
    try:
        for m, (W, H), variant, depth, spp, diffuse in ((default_model, (200, 160), abi.VARIANT_BVH, 10, 1, False),
                                                        (pool_model, (192, 108), abi.VARIANT_POOL, 8, 4, False),
                                                        (bunny0_model, (320, 180), abi.VARIANT_BVH, 4, 1, False),
                                                        (default_model, (160, 120), abi.VARIANT_BVH, 6, 2, True)):
            tr = m.camera(H)
            tracer.scene_upload(m.scene_view())
            tracer.set_camera(tr, W, H)
            tracer.set_decay_origin((0, 0, 0, 1))
            jit = host.jitter_table(spp) if spp > 1 else None
            frames = {}
            for start in (TAIL_NEVER, 0, 1, 2, 5):
                tracer.set_tuning(tailFrom=start)
                frames[start] = tracer.render(max_depth=depth, variant=variant, mode=abi.MODE_FAST, jitter=jit, diffuse=diffuse)
                again = tracer.render(max_depth=depth, variant=variant, mode=abi.MODE_FAST, jitter=jit, diffuse=diffuse)      # (the second frame: after the feedback of the first)
                assert np.array_equal(again[0], frames[start][0]), (W, start)
            for start in (0, 1, 2, 5):
                assert np.array_equal(frames[start][0], frames[TAIL_NEVER][0]), (W, H, start)
                assert np.array_equal(frames[start][1], frames[TAIL_NEVER][1]), (W, H, start)
    finally:
        tracer.set_tuning(base)


def test_cooperative_record_fetches_change_nothing(tracer, default_model, bunny0_model, pool_model):
    """LrlTuning.coopFetch: the FAST trace kernels fetch a 64-B record with four lanes (lanes l, l+16, l+32, l+48 read its four parts,
    a transpose of v_permlane swaps brings them home; a remedy for gathers bound by address translation, off by default).  The record a
    lane ends up with is the same 64 bytes, so the frame must be the same bit for bit: forced on here for the lean kernel over
    mask words and over compacted lists, and for the wave-per-ray hand-over kernel (threshold 4: nearly every ray goes through it),
    on scenes with every primitive class, exact ties and partly filled waves."""
    base = tracer.get_tuning()
    try:
        gP = load_golden("pool_viewA_160")
        for m, (W, H), tr, variant in ((default_model, (200, 168), default_model.camera(168), abi.VARIANT_BVH),
                                       (bunny0_model, (320, 184), bunny0_model.camera(184), abi.VARIANT_BVH),
                                       (pool_model, (160, 160), gP["trans"], abi.VARIANT_POOL)):
            tracer.scene_upload(m.scene_view())
            tracer.set_camera(tr, W, H)
            tracer.set_decay_origin((0, 0, 0, 1))
            for extra in ({}, {"compactFrom": 1}, {"adapt": 0, "handoverList": [4], "groupLanes": 64}):
                for diffuse in (False, True):
                    frames = {}
                    for coop in (0, 1):
                        tracer.set_tuning(base)
                        tracer.set_tuning(coopFetch=coop, **extra)
                        frames[coop] = tracer.render(max_depth=5, mode=abi.MODE_FAST, variant=variant, diffuse=diffuse)
                        assert tracer.last_render_info()["coopFetch"] == coop
                    assert np.array_equal(frames[1][1], frames[0][1]), (extra, diffuse, int((frames[1][1] != frames[0][1]).any(axis=-1).sum()))
                    assert np.array_equal(frames[1][0].view(np.uint32), frames[0][0].view(np.uint32)), (extra, diffuse, image_stats(frames[1][0], frames[0][0]))
    finally:
        tracer.set_tuning(base)


def test_lane_refill_changes_nothing(tracer, default_model, bunny0_model, pool_model):
    """LrlTuning.streamWords: the lean trace kernel as waves that own several 64-ray words and hand the next rays to idle lanes
    (wfTraceLeanStreamKernel).  A ray's result does not depend on the lane that carries it or on when it starts: frames are the same
    bit for bit as with one-word waves, over mask words and compacted lists, with the hand-over counting each ray's own trips, with
    samples in batches (words of several samples in one wave), every primitive class, exact ties, and refills of one lane at a time."""
    base = tracer.get_tuning()
    try:
        gP = load_golden("pool_viewA_160")
        jit = host.jitter_table(4)
        for m, (W, H), tr, variant in ((default_model, (200, 168), default_model.camera(168), abi.VARIANT_BVH),
                                       (bunny0_model, (320, 184), bunny0_model.camera(184), abi.VARIANT_BVH),
                                       (pool_model, (160, 160), gP["trans"], abi.VARIANT_POOL)):
            tracer.scene_upload(m.scene_view())
            tracer.set_camera(tr, W, H)
            tracer.set_decay_origin((0, 0, 0, 1))
            for extra in ({}, {"compactFrom": 1}, {"adapt": 0, "handoverList": [6]}):
                for diffuse in (False, True):
                    tracer.set_tuning(base)
                    tracer.set_tuning(**extra)
                    ref = tracer.render(max_depth=5, mode=abi.MODE_FAST, variant=variant, diffuse=diffuse, jitter=jit)
                    for words, refill in ((2, 16), (8, 16), (64, 1), (5, 40)):
                        tracer.set_tuning(base)
                        tracer.set_tuning(streamWords=words, refillMin=refill, **extra)
                        got = tracer.render(max_depth=5, mode=abi.MODE_FAST, variant=variant, diffuse=diffuse, jitter=jit)
                        assert tracer.last_render_info()["streamWords"] == words
                        assert np.array_equal(got[1], ref[1]), (extra, diffuse, words, refill, int((got[1] != ref[1]).any(axis=-1).sum()))
                        assert np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32)), (extra, diffuse, words, refill, image_stats(got[0], ref[0]))
    finally:
        tracer.set_tuning(base)


def test_one_block_a_trip_changes_nothing(tracer, default_model, bunny0_model, pool_model):
    """LrlTuning.phasedTrips: on trips where many lanes hold a triangle the lean kernel runs the triangle block alone and the node lanes
    wait with their record.  A ray's own sequence of items does not change, so the frame is the same bit for bit: forced on and off here
    over mask words, compacted lists and early hand-over, mirror and diffuse; then left to the library's own timed trial."""
    base = tracer.get_tuning()
    try:
        gP = load_golden("pool_viewA_160")
        for m, (W, H), tr, variant in ((default_model, (200, 168), default_model.camera(168), abi.VARIANT_BVH),
                                       (bunny0_model, (320, 184), bunny0_model.camera(184), abi.VARIANT_BVH),
                                       (pool_model, (160, 160), gP["trans"], abi.VARIANT_POOL)):
            tracer.scene_upload(m.scene_view())
            tracer.set_camera(tr, W, H)
            tracer.set_decay_origin((0, 0, 0, 1))
            for extra in ({}, {"compactFrom": 1}, {"adapt": 0, "handoverList": [6]}, {"sparseFrom": 1}):
                for diffuse in (False, True):
                    frames = {}
                    for ph in (2, 1):
                        tracer.set_tuning(base)
                        tracer.set_tuning(phasedTrips=ph, **extra)
                        frames[ph] = tracer.render(max_depth=5, mode=abi.MODE_FAST, variant=variant, diffuse=diffuse)
                        assert tracer.last_render_info()["phasedTrips"] == (1 if ph == 1 else 0)
                    assert np.array_equal(frames[1][1], frames[2][1]), (extra, diffuse, int((frames[1][1] != frames[2][1]).any(axis=-1).sum()))
                    assert np.array_equal(frames[1][0].view(np.uint32), frames[2][0].view(np.uint32)), (extra, diffuse, image_stats(frames[1][0], frames[2][0]))
            # left to itself (phasedTrips = 0) the library times one frame each way and settles: the same frame throughout
            tracer.set_tuning(base)
            assert tracer.get_tuning().phasedTrips == 0
            seen = set()
            for k in range(10):
                got = tracer.render(max_depth=5, mode=abi.MODE_FAST, variant=variant, diffuse=True)
                seen.add(tracer.last_render_info()["phasedTrips"])
                assert np.array_equal(got[0].view(np.uint32), frames[2][0].view(np.uint32)) and np.array_equal(got[1], frames[2][1]), k
            assert seen == {0, 1}                      # the trial ran both ways
    finally:
        tracer.set_tuning(base)


def test_hand_over_kernels_agree(default_model, bunny0_model, pool_model):
    """FAST mode's continuation kernels (16 lanes per ray, and one wave per ray for deep trees) keep no visit order; the
    closest hit must come out all the same.  With a threshold of 4 trips nearly every ray goes through them: both must
    agree with each other and with the default schedule bit for bit (the default scene has coincident discs and the Pool
    view along the water grid's diagonal has rays through shared edges: exact ties, decided by the primitive code)."""
    from raytracing_amd import tracer as T
    t = T.Tracer(0)
    base = t.get_tuning()
    try:
        gB, gP = load_golden("bvh_viewB_256"), load_golden("pool_viewA_160")
        for m, (W, H), tr, variant in ((default_model, (256, 256), gB["trans"], abi.VARIANT_BVH), (bunny0_model, (640, 360), bunny0_model.camera(360), abi.VARIANT_BVH),
                                       (pool_model, (160, 160), gP["trans"], abi.VARIANT_POOL)):
            res = {}
            for group in (None, 64, 16):
                t.set_tuning(base)
                if group:
                    t.set_tuning(adapt=0, handoverList=[4], groupLanes=group)
                t.scene_upload(m.scene_view())
                t.set_camera(tr, W, H)
                t.set_decay_origin((0, 0, 0, 1))
                res[group] = t.render(max_depth=8 if variant == abi.VARIANT_POOL else 4, mode=abi.MODE_FAST, variant=variant)
            # equal distances are decided by the primitive code and the triangle arithmetic is the same in the three kernels: the
            # frame does not depend on which kernel finishes a ray
            for group in (64, 16):
                assert np.array_equal(res[group][1], res[None][1]), (group, int((res[group][1] != res[None][1]).any(axis=-1).sum()))
                assert np.array_equal(res[group][0].view(np.uint32), res[None][0].view(np.uint32)), (group, image_stats(res[group][0], res[None][0]))
    finally:
        t.close()


@pytest.mark.parametrize("name", ["bvh_viewB_256", "bvh_viewB_128_spp4", "bvh_inside_glass_128"])
def test_wavefront_and_megakernel_schedules_agree_bit_for_bit(tracer, default_model, name):
    """The wavefront pipeline (default) and the one-launch kernel run the same PARITY arithmetic in a different order of
    launches: images, primary ids and every counter must be identical, glass (refraction stacks) and spp included."""
    g = load_golden(name)
    tracer.scene_upload(default_model.scene_view())
    tracer.set_camera(g["trans"], g["width"], g["height"])
    tracer.set_decay_origin(g["decay"].reshape(9, 4)[8] if name == "bvh_inside_glass_128" else (0, 0, 0, 1))
    jitter = g["jitter"] if "jitter" in g else None
    out = {}
    for mega in (False, True):
        img, hits = tracer.render(max_depth=g["depth"], mode=abi.MODE_PARITY, jitter=jitter, counters=True, megakernel=mega)
        out[mega] = (img, hits, tracer.counters().as_dict())
    assert np.array_equal(out[False][0].view(np.uint32), out[True][0].view(np.uint32))
    assert np.array_equal(out[False][1], out[True][1])
    assert out[False][2] == out[True][2]


# ---- Pool variant on the Pool demo scene (water surface refraction / reflection, four texture layers) ----
@pytest.mark.parametrize("mode", [abi.MODE_PARITY, abi.MODE_FAST])
@pytest.mark.parametrize("name", ["pool_viewA_160", "pool_viewB_192", "pool_underwater_128"])
def test_pool_scene_vs_llvmpipe_golden(tracer, pool_model, name, mode):
    """PARITY: 100 % of the ids.  View A looks along the diagonal of the water grid, so the pixels on the image diagonal
    x + y = 159 shoot rays exactly through edges shared by two water triangles (ids 47 apart): a true tie that FAST
    arithmetic (fma, rcp) may give to the neighbour.  7 of 25600 pixels do (tools/diag_pool.py lists them), all on that
    diagonal: each is proven a tie (assert_ids), the bar stays 99.99 %."""
    run_golden(tracer, pool_model, name, mode)


def test_pool_decay_origin_and_water_update(tracer, pool_model):
    """DecayOriginPre / DecayOrigin, Pool probe (+z), eye under water; then lrl_scene_update_triangles on the water
    triangles (what Pool does every frame, Pool/Pool.cpp:654-655): writing the same triangles back changes nothing,
    a lowered surface changes the image, in PARITY and in FAST mode (whose wide / lean kernels keep copies of the
    triangle records)."""
    g = load_golden("pool_underwater_128")
    view = pool_model.scene_view()
    tracer.scene_upload(view)
    tracer.set_camera(g["trans"], g["width"], g["height"])
    d = tracer.compute_decay_origin(abi.VARIANT_POOL)
    np.testing.assert_allclose(d["decayOrigin"], [-1, -1, -1, 1.33], rtol=1e-6)
    od = O.decay_origin(view, g["trans"], variant=abi.VARIANT_POOL)
    np.testing.assert_array_equal(d["decayOrigin"], od["decayOrigin"])
    g2 = load_golden("pool_viewB_192")
    tracer.set_camera(g2["trans"], g2["width"], g2["height"])
    tracer.set_decay_origin((0, 0, 0, 1))
    base, _ = tracer.render(max_depth=8, variant=abi.VARIANT_POOL)
    arr = pool_model.arrays()
    water = arr["triangles"][: 2 * 47 * 47].copy()
    tracer.update_triangles(0, water)
    same, _ = tracer.render(max_depth=8, variant=abi.VARIANT_POOL)
    assert np.array_equal(base, same)
    moved = water.copy()
    for f in ("p1", "p2", "p3"):
        moved[f][:, 2] -= 0.05
    tracer.update_triangles(0, moved)
    low, _ = tracer.render(max_depth=8, variant=abi.VARIANT_POOL)
    assert not np.array_equal(base, low)
    tracer.update_triangles(0, water)                       # and back: TrianglePre is a pure function of the triangles
    again, _ = tracer.render(max_depth=8, variant=abi.VARIANT_POOL)
    assert np.array_equal(base, again)
    fast0, _ = tracer.render(max_depth=8, variant=abi.VARIANT_POOL, mode=abi.MODE_FAST)   # the FAST kernels see the update too
    tracer.update_triangles(0, moved)
    fast1, _ = tracer.render(max_depth=8, variant=abi.VARIANT_POOL, mode=abi.MODE_FAST)
    check_image(fast0, base, abi.MODE_FAST, "pool water as uploaded, FAST vs PARITY")
    check_image(fast1, low, abi.MODE_FAST, "pool water lowered, FAST vs PARITY")
    tracer.update_triangles(0, water)


# ---- Pool water simulation on the device ----
def test_water_simulation_bit_exact_on_device(tracer, pool_model):
    """lrl_water_step against the llvmpipe fixture of the reference's three water shaders (z, v, a of every point, z of every
    water-triangle corner: bit-identical after 1 and after 50 substeps, in one go or in pieces), against the oracle for a
    longer run, and TrianglePre of the moved triangles against lrl_scene_update_triangles with the same triangles."""
    g = np.load(os.path.join(GOLDEN, "pool_water_steps.npz"))
    gx, gy, k, dt = int(g["gx"]), int(g["gy"]), float(g["k"]), float(g["dt"])
    nw = 2 * (gx - 1) * (gy - 1)
    view = pool_model.scene_view()
    tris = pool_model.arrays()["triangles"]
    w0 = scenes.pool_water_points(pool_model)

    def state():
        w, t = tracer.water_read()
        return np.stack([w["z"], w["v"], w["a"]], axis=1), np.stack([t["p1"][:, 2], t["p2"][:, 2], t["p3"][:, 2]], axis=1), w, t

    for pieces in ((1,), (50,), (20, 30)):
        tracer.scene_upload(view)
        tracer.water_init(gx, gy, k, dt, w0, 0, tris[:nw])
        for n in pieces:
            tracer.water_step(n)
        zva, tz, _, _ = state()
        n = sum(pieces)
        assert np.array_equal(zva.view(np.uint32), g[f"zva_{n}"].view(np.uint32)), pieces
        assert np.array_equal(tz.view(np.uint32), g[f"tri_z_{n}"].view(np.uint32)), pieces
    # the reference's default frame: 200 substeps (Pool/Pool.cpp:484), against the oracle
    tracer.water_step(150)
    zva, tz, w, t = state()
    ow = O.water_run(w0, gx, gy, k, dt, 200)
    ot = O.water_sync(ow, gx, gy, tris)
    assert np.array_equal(w["z"].view(np.uint32), ow["z"].view(np.uint32)) and np.array_equal(w["v"].view(np.uint32), ow["v"].view(np.uint32))
    assert np.array_equal(t.view(np.uint8), ot[:nw].view(np.uint8))
    # the tracer sees the moved water: same records as handing the triangles over from the host
    dev = tracer.read_triangles_gpu(0, nw)
    g2 = load_golden("pool_viewB_192")
    tracer.set_camera(g2["trans"], g2["width"], g2["height"])
    tracer.set_decay_origin((0, 0, 0, 1))
    img_dev, hits_dev = tracer.render(max_depth=8, variant=abi.VARIANT_POOL)
    tracer.scene_upload(view)
    tracer.update_triangles(0, ot[:nw])
    ref = tracer.read_triangles_gpu(0, nw)
    assert np.array_equal(dev.view(np.uint8), ref.view(np.uint8))
    img_ref, hits_ref = tracer.render(max_depth=8, variant=abi.VARIANT_POOL)
    assert np.array_equal(img_dev.view(np.uint32), img_ref.view(np.uint32)) and np.array_equal(hits_dev, hits_ref)
    oimg, ohits, _ = O.render(view_with(pool_model, ot), g2["trans"], g2["width"], g2["height"], max_depth=8, variant=abi.VARIANT_POOL, threads=8)
    assert (ohits == hits_dev).all(axis=-1).mean() >= ID_BAR and image_stats(img_dev, oimg)["rmse"] <= RMSE_PARITY


def view_with(model, triangles):
    """The model's scene view with another triangle array (same count, same BVH: what the Pool does every frame)."""
    v = model.scene_view()
    keep = np.ascontiguousarray(triangles, dtype=abi.TRIANGLE)
    view_with.keep = keep
    v.trianglesOrigin = keep.ctypes.data
    return v


# ---- BVH built on the device: LrlSceneView.bvh == NULL ----
def without_tree(model):
    v = model.scene_view()
    v.bvh = None
    v.bvhNodeNum = 0
    return v


@pytest.mark.parametrize("builder", [1, 2])
@pytest.mark.parametrize("mode", [abi.MODE_PARITY, abi.MODE_FAST])
@pytest.mark.parametrize("scene,name", [("default", "bvh_viewB_256"), ("default", "bvh_viewA_256"), ("bunny", "mesh_bunny0_240x135_depth4"),
                                        ("pool", "pool_viewB_192")])
def test_device_built_bvh_vs_llvmpipe_golden(default_model, bunny0_model, pool_model, scene, name, mode, builder):
    """A tree the library builds itself (Morton codes, radix sort, then Karras' radix tree (builder 1) or nearest-neighbour merging of
    clusters in Morton order (builder 2: in one workgroup for the small scenes, in rounds of kernels for the bunny), bottom-up fit,
    four-wide collapse: all on
    the device) gives the images the reference's shaders give with the host-built tree.  Another tree visits in another
    order, and a ray through an edge shared by two triangles (or through coplanar faces) hits both at the same distance:
    the first one visited keeps the pixel.  Ids may therefore differ on such ties (observed 8 of 65536 pixels on the default scene's
    view B, none elsewhere): each mismatch is proven a tie (assert_ids), the bar stays 99.99 %; the image bars are those of the host-built path."""
    from raytracing_amd import tracer as T
    model = {"default": default_model, "bunny": bunny0_model, "pool": pool_model}[scene]
    g = load_golden(name)
    tracer = T.Tracer(0)
    try:
        tracer.set_tuning(deviceBuilder=builder)
        tracer.scene_upload(without_tree(model))
        tracer.set_camera(g["trans"], g["width"], g["height"])
        tracer.set_decay_origin((0, 0, 0, 1))
        img, hits = tracer.render(max_depth=g["depth"], variant=g["variant"], mode=mode)
        tracer.rebuild_bvh()                     # the same tree again, in place
        img2, hits2 = tracer.render(max_depth=g["depth"], variant=g["variant"], mode=mode)
        assert np.array_equal(hits, hits2) and np.array_equal(img.view(np.uint32), img2.view(np.uint32))
    finally:
        tracer.close()
    st = image_stats(img, g["rgb"])
    assert_ids(hits, g["ids"], model.scene_view(), g["trans"], g["width"], g["height"], ("device-built tree", name, mode, builder))
    if scene == "default":
        # the default scene has coincident discs of different materials (circles 8 and 10: a 3 x 3 pixel patch of view B,
        # tools/diag_devbuild.py): those few pixels take the other disc's colour, all others meet the bar
        assert st["frac_gt_1e-3"] <= 5e-4, (name, mode, st)
    else:
        check_image(img, g["rgb"], mode, ("device-built tree", name))


def test_device_built_bvh_counts_and_edge_cases(tracer, default_model):
    """Ray counts are a property of the image, not of the tree: equal to the host-built tree's up to the tie pixels; the schedules agree bit for
    bit on the device-built tree too; a scene of one primitive and of two primitives builds and renders."""
    g = load_golden("bvh_viewB_256")
    out = {}
    for dev in (False, True):
        tracer.scene_upload(without_tree(default_model) if dev else default_model.scene_view())
        tracer.set_camera(g["trans"], g["width"], g["height"])
        tracer.set_decay_origin((0, 0, 0, 1))
        img, hits = tracer.render(max_depth=g["depth"], mode=abi.MODE_PARITY, counters=True)
        c = tracer.counters()
        out[dev] = (img, hits, c.rays, c.raysPrimary)
        if dev:
            img2, hits2 = tracer.render(max_depth=g["depth"], mode=abi.MODE_PARITY, megakernel=True)
            assert np.array_equal(img.view(np.uint32), img2.view(np.uint32)) and np.array_equal(hits, hits2)
    assert_ids(out[True][1], out[False][1], default_model.scene_view(), g["trans"], g["width"], g["height"], "device-built against host-built tree")
    assert abs(out[False][2] - out[True][2]) <= 2e-3 * out[False][2] and out[False][3] == out[True][3]      # (the tie pixels' paths differ)
    for nprim in (1, 2):
        m = host.HostModel()
        tri = np.zeros(nprim, dtype=abi.TRIANGLE)
        for i in range(nprim):
            tri["p1"][i] = (-1 + 3 * i, -1, -5); tri["p2"][i] = (1 + 3 * i, -1, -5); tri["p3"][i] = (0 + 3 * i, 1, -5)
            tri["color"]["g"][i] = (0.25 + 0.5 * i, 0.5, 0.75)
        for f in ("texR", "texT", "texD", "texG"):
            tri["color"][f] = -1
        tri["color"]["n"] = 1.0
        m.add_triangles(tri)
        m.data_init("reference")
        trans = m.camera(64)
        imgs = []
        for dev in (False, True):
            tracer.scene_upload(without_tree(m) if dev else m.scene_view())
            tracer.set_camera(trans, 64, 64)
            for mode in (abi.MODE_PARITY, abi.MODE_FAST):
                imgs.append(tracer.render(max_depth=4, mode=mode)[0])
        assert imgs[0][..., :3].max() > 0
        for im in imgs[1:]:
            check_image(im, imgs[0], abi.MODE_FAST, ("one / two primitives", nprim))


@pytest.mark.parametrize("count", [1000, 70000])
def test_device_builders_on_coincident_triangles(count):
    """Thousands of identical triangles (equal Morton codes, equal boxes: every merge is a tie) and a few others: both builders finish (PLOC
    pairs buddies off, in one workgroup below 65 536 primitives and in rounds of kernels above) and the frames agree: the nearest of equal
    hits is decided by the primitive code, whatever the tree."""
    from raytracing_amd import tracer as T
    m = host.HostModel()
    tri = np.zeros(count + 2, dtype=abi.TRIANGLE)
    tri["p1"][:count] = (-1, -1, -5); tri["p2"][:count] = (1, -1, -5); tri["p3"][:count] = (0, 1, -5)
    tri["p1"][count:] = [(-3, -1, -6), (2, -1, -4)]; tri["p2"][count:] = [(-1.5, -1, -6), (3.5, -1, -4)]; tri["p3"][count:] = [(-2, 1, -6), (3, 1, -4)]
    tri["color"]["g"] = (0.25, 0.5, 0.75)
    tri["color"]["g"][count:] = (0.75, 0.5, 0.25)
    for f in ("texR", "texT", "texD", "texG"):
        tri["color"][f] = -1
    tri["color"]["n"] = 1.0
    m.add_triangles(tri)
    m.data_init("median")
    trans = m.camera(64)
    out = {}
    for builder in (1, 2):
        t = T.Tracer(0)
        try:
            t.set_tuning(deviceBuilder=builder)
            t.scene_upload(without_tree(m))
            t.set_camera(trans, 96, 64)
            t.set_decay_origin((0, 0, 0, 1))
            out[builder] = t.render(max_depth=2, mode=abi.MODE_FAST)
        finally:
            t.close()
    assert (out[1][1][..., 0] == 2).sum() > 10                                    # the triangles are in view
    assert np.array_equal(out[1][1], out[2][1]) and np.array_equal(out[1][0].view(np.uint32), out[2][0].view(np.uint32))
    assert set(np.unique(out[2][1][..., 1][out[2][1][..., 0] == 2])) <= {0, count, count + 1}      # of the coincident ones, the lowest code


def test_rebuild_of_a_device_built_tree_follows_the_water(tracer, pool_model):
    """Device-built tree over the Pool scene: after lrl_water_step the triangles have left the boxes the tree was built
    with; lrl_scene_rebuild_bvh builds it again from the current triangles and the image is the oracle's image of the moved
    scene (the oracle walks the host tree, whose tall boxes hold every height).  A caller-supplied tree is never rebuilt."""
    from raytracing_amd.tracer import TraceError
    gx, gy = scenes.POOL_GRID
    k, dt = scenes.POOL_WATER
    nw = 2 * (gx - 1) * (gy - 1)
    tris = pool_model.arrays()["triangles"]
    w0 = scenes.pool_water_points(pool_model)
    g = load_golden("pool_viewB_192")
    tracer.scene_upload(pool_model.scene_view())
    with pytest.raises(TraceError):
        tracer.rebuild_bvh()
    tracer.scene_upload(without_tree(pool_model))
    tracer.set_camera(g["trans"], g["width"], g["height"])
    tracer.set_decay_origin((0, 0, 0, 1))
    tracer.water_init(gx, gy, k, dt, w0, 0, tris[:nw])
    tracer.water_step(200)
    tracer.rebuild_bvh()
    ow = O.water_run(w0, gx, gy, k, dt, 200)
    ot = O.water_sync(ow, gx, gy, tris)
    oimg, ohits, _ = O.render(view_with(pool_model, ot), g["trans"], g["width"], g["height"], max_depth=8, variant=abi.VARIANT_POOL, threads=8)
    for mode in (abi.MODE_PARITY, abi.MODE_FAST):
        img, hits = tracer.render(max_depth=8, variant=abi.VARIANT_POOL, mode=mode)
        assert_ids(hits, ohits, view_with(pool_model, ot), g["trans"], g["width"], g["height"], ("rebuilt tree over the moved water", mode))
        check_image(img, oimg, mode, "rebuilt tree over the moved water")
    # and again after more steps and a host-side triangle update
    tracer.water_step(100)
    tracer.update_triangles(0, ot[:nw])          # back to the state after 200 steps, through the host path
    tracer.rebuild_bvh()
    img, hits = tracer.render(max_depth=8, variant=abi.VARIANT_POOL, mode=abi.MODE_PARITY)
    assert_ids(hits, ohits, view_with(pool_model, ot), g["trans"], g["width"], g["height"], "rebuilt tree after a host-side update")
    assert image_stats(img, oimg)["rmse"] <= RMSE_PARITY


def test_quantize_rgba8_is_opengl_float_to_unorm(tracer):
    """lrl_quantize_rgba8 = round(clamp(v, 0, 1) * 255), ties to even, NaN -> 0: the conversion the reference's colours
    undergo on their way into its RGBA8 window."""
    import torch
    rng = np.random.default_rng(5)
    v = rng.uniform(-0.5, 1.5, size=(1000, 4)).astype(np.float32)
    v[:8] = [[0, 1, 0.5, 0.25], [np.nan, np.inf, -np.inf, -0.0], [0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255],
             [1e-9, 1 - 1e-9, 0.999999, 0.0019607], [2, -2, 1.0000001, 0.4999], [127.5 / 255, 128.5 / 255, 0.1, 0.9],
             [0.3, 0.6, 0.7, 1.0], [0.2, 0.4, 0.8, 0.0]]
    src = torch.from_numpy(v).cuda()
    dst = torch.zeros((1000, 4), dtype=torch.uint8, device="cuda")
    tracer.quantize_rgba8(src.data_ptr(), dst.data_ptr(), 1000)
    tracer.sync()
    with np.errstate(invalid="ignore"):
        want = np.rint(np.clip(np.nan_to_num(v, nan=0.0, posinf=1.0, neginf=0.0), 0, 1).astype(np.float32) * np.float32(255)).astype(np.uint8)
    assert np.array_equal(dst.cpu().numpy(), want)


# ---- samples traced in batches (spp > 1) ----
@pytest.mark.parametrize("name", ["bvh_viewB_128_spp4", "bvh_inside_glass_128"])
def test_sample_batches_change_nothing(tracer, default_model, name):
    """A frame of several samples is traced batch by batch (LrlTuning.batchSlots path slots per launch); the sum over the
    samples is taken in sample order whatever the batching, so PARITY images and primary ids are bit-identical for batches of
    1 (the per-sample pipeline), 2 + 2 + 1 and 5 samples, without and with glass (refraction stacks, polled iterations), and
    with a sample range; so are FAST's."""
    g = load_golden(name)
    W, H = g["width"], g["height"]
    tracer.scene_upload(default_model.scene_view())
    tracer.set_camera(g["trans"], W, H)
    tracer.set_decay_origin(g["decay"].reshape(9, 4)[8] if name == "bvh_inside_glass_128" else (0, 0, 0, 1))
    jit = host.jitter_table(5)
    slots = ((W + 7) // 8) * ((H + 7) // 8) * 64
    slots = (slots + 255) // 256 * 256
    base = tracer.get_tuning()
    try:
        out = {}
        for mode in (abi.MODE_PARITY, abi.MODE_FAST):
            for b in (1, 2, 5):
                tracer.set_tuning(batchSlots=b * slots)
                out[mode, b] = tracer.render(max_depth=g["depth"], mode=mode, jitter=jit, diffuse=(name == "bvh_viewB_128_spp4"))
                if b == 2:
                    part = tracer.render(max_depth=g["depth"], mode=mode, jitter=jit, samples=(1, 4), diffuse=(name == "bvh_viewB_128_spp4"))
                    tracer.set_tuning(batchSlots=slots)
                    part1 = tracer.render(max_depth=g["depth"], mode=mode, jitter=jit, samples=(1, 4), diffuse=(name == "bvh_viewB_128_spp4"))
                    if mode == abi.MODE_PARITY:
                        assert np.array_equal(part[0].view(np.uint32), part1[0].view(np.uint32)) and np.array_equal(part[1], part1[1])
        for b in (2, 5):
            assert np.array_equal(out[abi.MODE_PARITY, b][0].view(np.uint32), out[abi.MODE_PARITY, 1][0].view(np.uint32)), b
            assert np.array_equal(out[abi.MODE_PARITY, b][1], out[abi.MODE_PARITY, 1][1]), b
            # FAST frames do not depend on the schedule either (ties are decided by the primitive code)
            assert np.array_equal(out[abi.MODE_FAST, b][0].view(np.uint32), out[abi.MODE_FAST, 1][0].view(np.uint32)), (b, image_stats(out[abi.MODE_FAST, b][0], out[abi.MODE_FAST, 1][0]))
            assert np.array_equal(out[abi.MODE_FAST, b][1], out[abi.MODE_FAST, 1][1]), b
        st = image_stats(out[abi.MODE_FAST, 5][0], out[abi.MODE_PARITY, 5][0])
        if name == "bvh_viewB_128_spp4":      # diffuse bounces: a direction that moves by an ulp in FAST may land its path on a light's hot spot
            assert st["frac_gt_1e-3"] <= 2e-3, st
        else:
            check_image(out[abi.MODE_FAST, 5][0], out[abi.MODE_PARITY, 5][0], abi.MODE_FAST, ("batched samples, glass", name))
    finally:
        tracer.set_tuning(base)


def test_hit_ids_of_a_sample_range(tracer, default_model):
    """hit_out holds the primary hits of the FIRST sample a render traces: a render of samples [1, 2) alone (what rank 1 of a
    sample-sharded frame does) returns the ids a one-sample render with that sample's jitter returns."""
    g = load_golden("bvh_viewB_128_spp4")
    tracer.scene_upload(default_model.scene_view())
    tracer.set_camera(g["trans"], g["width"], g["height"])
    tracer.set_decay_origin((0, 0, 0, 1))
    jit = np.asarray(g["jitter"], dtype=np.float32).reshape(-1, 2)
    for mega in (False, True):
        _, h_range = tracer.render(max_depth=6, jitter=jit, samples=(1, 2), megakernel=mega)
        _, h_single = tracer.render(max_depth=6, jitter=jit[1:2], megakernel=mega)
        _, h_zero = tracer.render(max_depth=6, jitter=jit[0:1], megakernel=mega)
        assert np.array_equal(h_range, h_single)
        assert not np.array_equal(h_range, h_zero)           # the jitter moves some edge pixels: the two samples differ


@pytest.mark.parametrize("mode", [abi.MODE_PARITY, abi.MODE_FAST])
def test_c1_at_1024_vs_llvmpipe_golden(tracer, default_model, mode):
    """BASELINE configs[0] at its own size, 1024 x 1024, RayTraceDepth 10: primary-hit ids of every pixel and 16 x 16 tile sums of
    the image against the reference's shaders on llvmpipe (tests/golden/bvh_viewA_1024.npz, tools/baseline_llvmpipe.py)."""
    z = np.load(os.path.join(GOLDEN, "bvh_viewA_1024.npz"))
    W = H = int(z["width"])
    tr = np.frombuffer(z["trans"].tobytes(), dtype=abi.TRANS)
    tracer.scene_upload(default_model.scene_view())
    tracer.set_camera(tr, W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    img, hits = tracer.render(max_depth=int(z["depth"]), mode=mode)
    same = ((hits[..., 0] == z["hit_type"]) & (hits[..., 1] == z["hit_idx"])).mean()
    assert same >= ID_BAR, (mode, same)
    T = int(z["tile"])
    sums = np.nan_to_num(img[..., :3].astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0).reshape(H // T, T, W // T, T, 3).sum(axis=(1, 3))
    err = np.abs(sums - z["tile_sums"])
    # a tile sum is 256 pixels: the per-pixel RMSE bars (1e-4 PARITY, 5e-3 FAST) times sqrt(256) for independent errors
    # a tile sum is 256 pixels: PARITY's per-pixel RMSE bar times sqrt(256) for independent errors; FAST: the gross-error guard (a tie pixel
    # moves its tile's sum by up to 0.5)
    bar = (RMSE_PARITY if mode == abi.MODE_PARITY else FAST_RMSE) * 16
    assert np.sqrt((err ** 2).mean()) <= bar, (mode, float(np.sqrt((err ** 2).mean())), float(err.max()), same)


def _host_records(ptr, n, dt):
    """A writable numpy view of one of the host model's own arrays (no copy): the test moves primitives the way Movement::run does."""
    import ctypes as C
    return np.frombuffer((C.c_uint8 * (n * dt.itemsize)).from_address(ptr), dtype=dt)


def test_moving_spheres_and_lights_like_movement_run(tracer):
    """Movement::run (/root/reference/BVH/BVH.cpp:187-205): two spheres and a point light move, then model.dataInit() uploads the
    scene again.  Here: lrl_scene_update_primitives on the moved records + lrl_scene_update_bvh with the tree the host builder makes
    for the new positions (caller's tree), or + lrl_scene_rebuild_bvh (library's tree).  The frame must be the one a fresh upload of
    the moved scene gives, bit for bit, in PARITY and FAST; the caller's-tree frame is also checked against the oracle."""
    from raytracing_amd import tracer as T
    g = load_golden("bvh_viewB_256")
    W, H = 192, 192
    m = scenes.bvh_default(builder="reference")
    v = m.scene_view()
    tracer.scene_upload(v)
    tracer.set_camera(g["trans"], W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    before = tracer.render(max_depth=6)
    dev = T.Tracer(0)                                     # the same scene on a tree of the library's own
    fresh = T.Tracer(0)
    try:
        vd = m.scene_view(); vd.bvh = None; vd.bvhNodeNum = 0
        dev.scene_upload(vd)
        dev.set_camera(g["trans"], W, H); dev.set_decay_origin((0, 0, 0, 1))
        for step in range(2):
            t = 0.1 * (step + 1) * 7                      # (a visible move: seven of the reference's 0.1 steps at once)
            sph = _host_records(v.spheres, v.num.sphereNum, abi.SPHERE)
            lights = _host_records(v.pointLights, v.num.pointLightNum, abi.POINTLIGHT)
            sph["sphere"][0, 1] += np.float32(np.sin(t)); sph["sphere"][1, 1] += np.float32(np.sin(t))      # sphereRun
            lights["p"][0, 0] = np.float32(3 * np.sin(t)); lights["p"][0, 2] += np.float32(0.2 * np.sin(t))  # pointLightRun
            m.data_init("reference")                      # model.dataInit(): the host builder's tree for the new positions
            v = m.scene_view()
            sph = _host_records(v.spheres, v.num.sphereNum, abi.SPHERE)
            lights = _host_records(v.pointLights, v.num.pointLightNum, abi.POINTLIGHT)
            tracer.update_primitives(abi.PRIM_SPHERE, 0, sph[:2])
            tracer.update_primitives(abi.PRIM_LIGHT, 0, lights[:1])
            tracer.update_bvh(v.bvh, v.bvhNodeNum)
            dev.update_primitives(abi.PRIM_SPHERE, 0, sph[:2])
            dev.update_primitives(abi.PRIM_LIGHT, 0, lights[:1])
            dev.rebuild_bvh()
            fresh.scene_upload(v)
            fresh.set_camera(g["trans"], W, H); fresh.set_decay_origin((0, 0, 0, 1))
            for mode in (abi.MODE_PARITY, abi.MODE_FAST):
                got, want = tracer.render(max_depth=6, mode=mode), fresh.render(max_depth=6, mode=mode)
                assert np.array_equal(got[1], want[1]) and np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), (step, mode)
            assert not np.array_equal(got[1], before[1])                      # something did move
            oimg, ohits, _ = O.render(v, g["trans"], W, H, max_depth=6, threads=8)
            par = tracer.render(max_depth=6)
            assert (par[1] == ohits).all(axis=-1).mean() >= ID_BAR
            check_image(par[0], oimg, abi.MODE_PARITY, ("moved spheres", step))
            # the library's own tree: against a fresh upload without a tree
            vf = m.scene_view(); vf.bvh = None; vf.bvhNodeNum = 0
            fresh.scene_upload(vf)
            for mode in (abi.MODE_PARITY, abi.MODE_FAST):
                got, want = dev.render(max_depth=6, mode=mode), fresh.render(max_depth=6, mode=mode)
                assert np.array_equal(got[1], want[1]) and np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), ("device", step, mode)
            # (another tree visits the scene's coincident discs in another order: exact ties, as in test_device_built_bvh_vs_llvmpipe_golden)
            assert_ids(dev.render(max_depth=6)[1], ohits, v, g["trans"], W, H, ("moved spheres, the library's own tree", step))
        # what is refused
        for bad in (lambda: tracer.update_primitives(abi.PRIM_SPHERE, v.num.sphereNum, sph[:1]),
                    lambda: tracer.update_primitives(2, 0, sph[:1]),
                    lambda: dev.update_bvh(v.bvh, v.bvhNodeNum),
                    lambda: tracer.update_bvh(v.bvh, 1)):
            with pytest.raises(T.TraceError):
                bad()
        assert np.array_equal(tracer.render(max_depth=6)[1], par[1])          # a refused update leaves the scene as it was
    finally:
        dev.close(); fresh.close()


def test_parked_primitives_are_bounded_by_their_own_boxes(tracer, pool_model):
    """The reference's builder parks a primitive in most inner nodes, and its shader tests it whenever the ray enters that node's box
    (a whole subtree's); the Pool's leaf boxes are as tall as the highest wave.  The four-wide tree of the FAST kernels is fitted to
    the primitives' own boxes, and refitted on the device whenever triangles change (lrl_refit_node4): the same frame, bit for bit, as
    with the caller's boxes alone.  The caller's boxes stay the contract (the Pool's water moves under a fixed tree whose leaf boxes were
    made tall enough, Pool/Pool.cpp:250-263): a parked triangle moved inside its node's box, without a new tree, must be found where the
    oracle finds it on that tree; one moved out of its node's box is bounded by that box, as for the reference, and is seen only where
    the oracle sees it through the old tree."""
    g = load_golden("pool_viewB_192")
    W, H = g["width"], g["height"]
    view = pool_model.scene_view()
    tris = pool_model.arrays()["triangles"]
    bvh = pool_model.arrays()["bvh"]
    parked = np.flatnonzero((bvh["leftChild"] != 0) & (bvh["geometry"] == 2))
    assert len(parked) > 500                                  # (the Pool's tree: 2220 of 4446 triangles sit in inner nodes)
    tracer.scene_upload(view)
    tracer.set_camera(g["trans"], W, H)
    tracer.set_decay_origin((0, 0, 0, 1))
    own = tracer.scene_info()["ownBoxes"]
    assert own >= len(parked)
    kw = dict(max_depth=8, variant=abi.VARIANT_POOL, mode=abi.MODE_FAST)
    tight = tracer.render(**kw)
    tracer.update_triangles(0, tris)                          # the same triangles again, as Pool sends them every frame: nothing moved
    assert tracer.scene_info()["ownBoxes"] == own             # (now the refit's count: the same rule, the same boxes)
    same = tracer.render(**kw)
    assert np.array_equal(same[1], tight[1]) and np.array_equal(same[0].view(np.uint32), tight[0].view(np.uint32))
    # a parked water triangle near the middle of the view, lowered by less than its node's box allows
    nodes = parked[bvh["geometryNum"][parked] < 2 * 47 * 47]
    pick = nodes[np.argmax(bvh["max"][nodes, 2] - bvh["min"][nodes, 2] > 0.05)]
    ti = int(bvh["geometryNum"][pick])

    def oracle_with(moved):
        all_tris = tris.copy(); all_tris[ti] = moved[0]
        m2 = scenes.pool()
        _host_records(m2.scene_view().trianglesOrigin, len(all_tris), abi.TRIANGLE)[:] = all_tris      # the oracle: moved triangle, the old tree
        return O.render(m2.scene_view(), g["trans"], W, H, max_depth=8, variant=abi.VARIANT_POOL, threads=8)

    moved = tris[ti:ti + 1].copy()
    for f in ("p1", "p2", "p3"):
        moved[f][:, 2] -= 0.04
    assert all((moved[f][0] >= bvh["min"][pick]).all() and (moved[f][0] <= bvh["max"][pick]).all() for f in ("p1", "p2", "p3"))
    tracer.update_triangles(ti, moved)
    assert tracer.scene_info()["ownBoxes"] == own             # still bounded by its own box, where it is now
    got = tracer.render(**kw)
    par = tracer.render(max_depth=8, variant=abi.VARIANT_POOL)
    oimg, ohits, _ = oracle_with(moved)
    assert (par[1] == ohits).all(axis=-1).mean() >= ID_BAR and (got[1] == ohits).all(axis=-1).mean() >= ID_BAR
    check_image(got[0], oimg, abi.MODE_FAST, "parked triangle moved under the caller's tree")
    assert not np.array_equal(got[0], tight[0])
    # out of its node's box (raised above it): the caller's box bounds it again, and decides which rays meet it
    out = tris[ti:ti + 1].copy()
    lift = np.float32(bvh["max"][pick, 2] - min(out[f][0, 2] for f in ("p1", "p2", "p3")) + 0.02)
    for f in ("p1", "p2", "p3"):
        out[f][:, 2] += lift
    assert all(out[f][0, 2] > bvh["max"][pick, 2] for f in ("p1", "p2", "p3"))
    tracer.update_triangles(ti, out)
    assert tracer.scene_info()["ownBoxes"] == own - 1
    got = tracer.render(**kw)
    par = tracer.render(max_depth=8, variant=abi.VARIANT_POOL)
    oimg, ohits, _ = oracle_with(out)
    assert (par[1] == ohits).all(axis=-1).mean() >= ID_BAR and (got[1] == ohits).all(axis=-1).mean() >= ID_BAR
    check_image(got[0], oimg, abi.MODE_FAST, "parked triangle moved out of its node's box")
    tracer.update_triangles(ti, tris[ti:ti + 1])              # back where it was: the same boxes, the same frame
    assert tracer.scene_info()["ownBoxes"] == own
    back = tracer.render(**kw)
    assert np.array_equal(back[1], tight[1]) and np.array_equal(back[0].view(np.uint32), tight[0].view(np.uint32))
    tracer.update_bvh(view.bvh, view.bvhNodeNum)              # a new tree: own boxes again
    assert tracer.scene_info()["ownBoxes"] == own
    again = tracer.render(**kw)
    assert np.array_equal(again[0].view(np.uint32), tight[0].view(np.uint32))
    # the other kinds keep the caller's boxes (the kernels' cold path is cheapest taken by a wave's lanes together, on arrival at the node):
    # a sphere may move inside its leaf box under a fitted tree
    gb = load_golden("bvh_viewB_256")
    m = scenes.bvh_default(builder="reference")
    v = m.scene_view()
    tracer.scene_upload(v)
    tracer.set_camera(gb["trans"], 192, 192)
    own = tracer.scene_info()["ownBoxes"]
    assert 0 < own <= v.num.triangleNum
    a = tracer.render(max_depth=6, mode=abi.MODE_FAST)
    sph = _host_records(v.spheres, v.num.sphereNum, abi.SPHERE).copy()
    small = sph.copy(); small["sphere"][:, 3] *= np.float32(0.25)        # half the radius: inside the leaf boxes
    tracer.update_primitives(abi.PRIM_SPHERE, 0, small)
    assert tracer.scene_info()["ownBoxes"] == own
    b = tracer.render(max_depth=6, mode=abi.MODE_FAST)
    m2 = scenes.bvh_default(builder="reference")
    _host_records(m2.scene_view().spheres, v.num.sphereNum, abi.SPHERE)[:] = small
    oimg, ohits, _ = O.render(m2.scene_view(), gb["trans"], 192, 192, max_depth=6, threads=8)
    assert (b[1] == ohits).all(axis=-1).mean() >= ID_BAR and not np.array_equal(a[1], b[1])
    check_image(b[0], oimg, abi.MODE_FAST, "spheres shrunk under a fitted tree")
    tracer.update_primitives(abi.PRIM_SPHERE, 0, sph)
    c = tracer.render(max_depth=6, mode=abi.MODE_FAST)
    assert np.array_equal(a[1], c[1]) and np.array_equal(a[0].view(np.uint32), c[0].view(np.uint32))


def test_oversized_boxes_of_a_callers_tree_are_fitted(tracer, bunny0_model):
    """Any caller's tree whose boxes are larger than what they hold gains from the fit, not only the Pool's: the bunny's median-split tree
    with every box grown by 4 % of the scene (still nested, still valid for the reference's traversal) renders the oracle's image of
    that tree, the same hits as with the tight tree, and `ownBoxes` says every triangle is bounded by its own box; a tree that is not
    nested (one leaf box poking out of its parent's) is walked with the caller's boxes as they are."""
    m = bunny0_model
    W, H = 160, 92                                  # (the oracle pays for the grown boxes: 10 s)
    tr = m.camera(H)
    v = m.scene_view()
    bvh = m.arrays()["bvh"].copy()
    grow = np.float32(0.04 * float((bvh["max"][0] - bvh["min"][0]).max()))
    big = bvh.copy()
    big["min"] -= grow; big["max"] += grow
    out = {}
    for name, tree in (("tight", bvh), ("grown", big)):
        v.bvh = tree.ctypes.data
        tracer.scene_upload(v)
        tracer.set_camera(tr, W, H)
        tracer.set_decay_origin((0, 0, 0, 1))
        out[name] = (tracer.render(max_depth=4, mode=abi.MODE_FAST), tracer.scene_info()["ownBoxes"])
    assert out["tight"][1] == 0 and out["grown"][1] == v.num.triangleNum        # (the tight tree's leaf boxes are the triangles' own)
    oimg, ohits, _ = O.render(v, tr, W, H, max_depth=4, threads=8)              # the oracle on the grown tree
    assert (out["grown"][0][1] == ohits).all(axis=-1).mean() >= ID_BAR
    check_image(out["grown"][0][0], oimg, abi.MODE_FAST, "bunny, grown boxes, fitted")
    assert (out["grown"][0][1] == out["tight"][0][1]).all(axis=-1).mean() >= ID_BAR
    moved = (out["grown"][0][0].view(np.uint32) != out["tight"][0][0].view(np.uint32)).any(axis=-1).mean()
    assert moved <= 1e-4, moved                                                 # (floating-point tie edges only)
    # not nested: no fit
    bad = big.copy()
    leaf = int(np.flatnonzero((bad["leftChild"] == 0) & (bad["geometry"] == 2))[1000])
    bad["max"][leaf, 2] = bad["max"][0, 2] + 1.0
    v.bvh = bad.ctypes.data
    tracer.scene_upload(v)
    assert tracer.scene_info()["ownBoxes"] == 0
    tracer.set_camera(tr, W, H)
    got = tracer.render(max_depth=4, mode=abi.MODE_FAST)
    assert (got[1] == ohits).all(axis=-1).mean() >= ID_BAR


def test_a_callers_tree_follows_the_water(pool_model):
    """The Pool's binding syncs the water's triangles on the device once at start-up (lrl_water_init + lrl_water_step(0),
    Pool/Pool.cpp:648-650) and steps them every frame: they move under the caller's tree.  The four-wide tree is refitted to where they
    are by a kernel queued behind the step (lrl_refit_node4; nothing is read back, no frame waits for the host): the frames are the
    oracle's on the moved triangles, at rest and after real waves, and the same as the PARITY kernels' walk of the caller's own boxes.
    "The same" is bit for bit except at floating-point tie edges: of this view's 36 864 x 64 primary rays one passes 8e-7 (barycentric) from
    the edge shared by triangles 1120 and 1167; FAST arithmetic finds it inside both, 1167 an ulp nearer.  The caller's leaf box of 1167
    ends exactly on that edge and the slab arithmetic's rounding turns the ray away, so the caller's boxes give 1120 (as the oracle does);
    the fitted boxes have a margin and give 1167 (tools/diag/fit_diff2.py)."""
    from raytracing_amd import tracer as T
    gx, gy = scenes.POOL_GRID
    k, dt = scenes.POOL_WATER
    nw = 2 * (gx - 1) * (gy - 1)
    tris = pool_model.arrays()["triangles"]
    w0 = scenes.pool_water_points(pool_model)
    g = load_golden("pool_viewB_192")
    W, H, spp = g["width"], g["height"], 64
    jit = host.jitter_table(spp)
    kw = dict(max_depth=8, variant=abi.VARIANT_POOL, mode=abi.MODE_FAST, jitter=jit)
    t = T.Tracer(0)
    try:
        t.scene_upload(pool_model.scene_view())
        t.set_camera(g["trans"], W, H)
        t.set_decay_origin((0, 0, 0, 1))
        fitted = t.scene_info()["ownBoxes"]
        assert fitted > 2000
        t.water_init(gx, gy, k, dt, w0, 0, tris[:nw])
        for steps in (0, 200):
            t.water_step(steps)
            first = t.render(**kw)
            assert t.scene_info()["ownBoxes"] >= fitted - 50              # (a wave may lift a corner out of a box made for the water at rest)
            again = t.render(**kw)
            assert np.array_equal(again[1], first[1]) and np.array_equal(again[0].view(np.uint32), first[0].view(np.uint32)), steps
            ow = O.water_run(w0, gx, gy, k, dt, steps) if steps else w0
            ot = O.water_sync(ow, gx, gy, tris)
            oimg, ohits, _ = O.render(view_with(pool_model, ot), host.sample_trans(g["trans"], jit), W, H, max_depth=8, variant=abi.VARIANT_POOL, threads=8)
            assert_ids(again[1], ohits, view_with(pool_model, ot), sample0(g["trans"], jit), W, H, ("the Pool's tree follows the water", steps))
            check_image(again[0], oimg, abi.MODE_FAST, ("the Pool's tree follows the water", steps))
        # many small steps, a frame each, nothing waited for in between: the frame after the last step is the frame of one long run
        t.water_init(gx, gy, k, dt, w0, 0, tris[:nw])
        for _ in range(20):
            t.water_step(10)
            last = t.render(**kw)
        t.water_init(gx, gy, k, dt, w0, 0, tris[:nw])
        t.water_step(200)
        once = t.render(**kw)
        assert np.array_equal(last[1], once[1]) and np.array_equal(last[0].view(np.uint32), once[0].view(np.uint32))
    finally:
        t.close()


def test_scene_changes_and_renders_on_a_callers_stream_are_ordered(pool_model):
    """lrl_water_step only queues its kernels on the context's stream, and lrl_render_async may run on a stream of the caller's: a render
    must see the water step queued before it, and the next water step (or triangle update) must not overwrite triangles the render in
    flight still reads.  Frame loop water step -> render on a foreign stream, nothing waited for in between, against the same loop with
    blocking renders on the context's own stream: the same frames bit for bit, with the library's tree rebuilt every frame."""
    import torch
    from raytracing_amd import tracer as T
    g = np.load(os.path.join(GOLDEN, "pool_water_steps.npz"))
    gx, gy, k, dt = int(g["gx"]), int(g["gy"]), float(g["k"]), float(g["dt"])
    nw = 2 * (gx - 1) * (gy - 1)
    gP = load_golden("pool_viewA_160")
    W = H = 160
    tris = pool_model.arrays()["triangles"]
    w0 = scenes.pool_water_points(pool_model)
    view = pool_model.scene_view()
    view.bvh = None; view.bvhNodeNum = 0
    frames = {}
    for mode in ("blocking", "foreign stream"):
        t = T.Tracer(0)
        try:
            t.scene_upload(view)
            t.water_init(gx, gy, k, dt, w0, 0, tris[:nw])
            t.set_camera(gP["trans"], W, H)
            t.set_decay_origin((0, 0, 0, 1))
            outs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(4)]
            stream = torch.cuda.Stream(device="cuda:0")
            got = []
            for i in range(4):
                t.water_step(25)
                t.rebuild_bvh()
                if mode == "blocking":
                    got.append(t.render(max_depth=8, variant=abi.VARIANT_POOL, mode=abi.MODE_FAST, want_hits=False)[0])
                else:
                    t.render_device(outs[i].data_ptr(), stream=stream.cuda_stream, max_depth=8, variant=abi.VARIANT_POOL, mode=abi.MODE_FAST)
            if mode != "blocking":
                stream.synchronize()
                t.sync()
                got = [o.cpu().numpy() for o in outs]
            frames[mode] = got
        finally:
            t.close()
    for i in range(4):
        assert np.array_equal(frames["blocking"][i].view(np.uint32), frames["foreign stream"][i].view(np.uint32)), i
    assert not np.array_equal(frames["blocking"][0], frames["blocking"][3])      # the water did move


def test_tree_layout_changes_nothing(default_model, bunny0_model, pool_model):
    """The four-wide tree is stored in the order the rays walk it (LrlTuning.treeLayout = 0: reachable nodes only, a node's children side by
    side, subtree after subtree; the triangles' hot records in the order of the leaves, their numbers inside the records) or as built,
    addressed by binary record index (1).  Where a record lies decides nothing: FAST frames and hit ids are the same bit for bit, for a
    caller's tree (every primitive class, the Pool's fitted + loose pair, the bunny) and for a tree the library built (the bunny: 77 560
    triangles, laid out from 65 536 on), also after triangles were updated in place (both copies of their records follow), after a new
    caller's tree (lrl_scene_update_bvh) and after a rebuild.  SceneInfo says what is in use."""
    from raytracing_amd import tracer as T
    cases = [("default", default_model, "bvh_viewB_256", False), ("bunny", bunny0_model, "mesh_bunny0_240x135_depth4", False),
             ("pool", pool_model, "pool_viewB_192", False), ("bunny, library's tree", bunny0_model, "mesh_bunny0_240x135_depth4", True)]
    for what, model, name, device in cases:
        g = load_golden(name)
        out = {}
        for layout in (0, 1):
            t = T.Tracer(0)
            try:
                t.set_tuning(treeLayout=layout)
                t.scene_upload(without_tree(model) if device else model.scene_view())
                t.set_camera(g["trans"], g["width"], g["height"])
                t.set_decay_origin((0, 0, 0, 1))
                info = t.scene_info()
                assert info["leafOrder"] == (1 - layout) and (info["wideNodes"] < info["bvhRecords"]) == (layout == 0), (what, layout, info)
                frames = [t.render(max_depth=g["depth"], variant=g["variant"], mode=abi.MODE_FAST)]
                tris = model.arrays()["triangles"]
                n = min(len(tris), 3000)
                moved = tris[:n].copy()
                for f in ("p1", "p2", "p3"):
                    moved[f][:, 2] += np.float32(0.01)
                t.update_triangles(0, moved)                               # in place: both copies of the hot records follow
                if device:
                    t.rebuild_bvh()
                frames.append(t.render(max_depth=g["depth"], variant=g["variant"], mode=abi.MODE_FAST))
                t.update_triangles(0, tris[:n])
                if device:
                    t.rebuild_bvh()
                else:
                    v = model.scene_view()
                    t.update_bvh(v.bvh, v.bvhNodeNum)                      # the same tree again, through the update path
                frames.append(t.render(max_depth=g["depth"], variant=g["variant"], mode=abi.MODE_FAST))
                out[layout] = frames
            finally:
                t.close()
        for i, (a, b) in enumerate(zip(out[0], out[1])):
            assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1]), (what, i)
        assert np.array_equal(out[0][0][0].view(np.uint32), out[0][2][0].view(np.uint32)), what          # back where it started
        if what == "pool":                                                                                # (the water is in view: the move was seen)
            assert not np.array_equal(out[0][0][0], out[0][1][0]), what


def test_deterministic_mode_decides_nothing_from_measurements(pool_model):
    """LrlTuning.deterministic = 1: no timed trial of triangle trips, no adaptation of the hand-over thresholds: every frame of a scene
    issues the same launches.  Twelve frames in a row report the base thresholds, trialState 0 and triangle trips off, and give the
    adaptive mode's frame bit for bit.  (The refit of a caller's tree after a water step is a fixed sequence of launches in either mode.)"""
    from raytracing_amd import tracer as T
    g = load_golden("pool_viewB_192")
    gx, gy = scenes.POOL_GRID
    k, dt = scenes.POOL_WATER
    nw = 2 * (gx - 1) * (gy - 1)
    tris = pool_model.arrays()["triangles"]
    w0 = scenes.pool_water_points(pool_model)
    jit = host.jitter_table(16)
    kw = dict(max_depth=8, variant=abi.VARIANT_POOL, mode=abi.MODE_FAST, jitter=jit)
    frames = {}
    for det in (1, 0):
        t = T.Tracer(0)
        try:
            t.set_tuning(deterministic=det)
            t.scene_upload(pool_model.scene_view())
            t.set_camera(g["trans"], g["width"], g["height"])
            t.set_decay_origin((0, 0, 0, 1))
            infos = []
            for i in range(12):
                img = t.render(**kw)
                infos.append(t.last_render_info())
            frames[det] = img
            if det:
                assert all(i["trialState"] == 0 and i["phasedTrips"] == 0 and i["handoverTrips"] == infos[0]["handoverTrips"] for i in infos), infos
                assert infos[0]["handoverTrips"][:4] == [48, 48, 32, 16]
                t.water_init(gx, gy, k, dt, w0, 0, tris[:nw])
                t.water_step(100)
                for i in range(6):
                    t.render(**kw)
                    info = t.last_render_info()
                    assert info["trialState"] == 0 and info["phasedTrips"] == 0 and info["handoverTrips"] == infos[0]["handoverTrips"], info
            else:
                assert any(i["trialState"] != 0 for i in infos)
        finally:
            t.close()
    assert np.array_equal(frames[0][0].view(np.uint32), frames[1][0].view(np.uint32)) and np.array_equal(frames[0][1], frames[1][1])
