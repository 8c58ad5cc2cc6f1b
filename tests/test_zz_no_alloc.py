"""lrl_scene_update_triangles_noalloc (the C-ABI entry point for callers that cannot allocate: it stages the caller's triangles
in a static device array of the library instead of buffers from hipMalloc) against lrl_scene_update_triangles, through the C ABI:
the same device records bit for bit and the same image, on two calls in a row, the second with a smaller input; and its limits."""
import numpy as np
import pytest

from conftest import load_golden
from raytracing_amd import abi, scenes
from raytracing_amd import tracer as T

pytestmark = pytest.mark.gpu

NOALLOC_MAX = 65536          # LRL_NOALLOC_MAX_TRIANGLES, include/lrl_trace.h


def moved_water(tris, n, dz, seed):
    rng = np.random.default_rng(seed)
    t = tris[:n].copy()
    for f in ("p1", "p2", "p3"):
        t[f][:, 2] += (dz * rng.uniform(-1, 1, size=n)).astype(np.float32)
    return t


@pytest.mark.parametrize("device_built", [False, True])
def test_noalloc_update_equals_the_allocating_update(pool_model, device_built):
    g = load_golden("pool_viewB_192")
    view = pool_model.scene_view()
    if device_built:
        view.bvh = None
        view.bvhNodeNum = 0
    tris = pool_model.arrays()["triangles"]
    nw = 2 * 47 * 47
    a, b = T.Tracer(0), T.Tracer(0)          # a: the existing entry point; b: the new one
    try:
        for t in (a, b):
            t.scene_upload(view)
            t.set_camera(g["trans"], g["width"], g["height"])
            t.set_decay_origin((0, 0, 0, 1))
        # two calls in a row, the second smaller (and elsewhere in the array): the staging area keeps the first call's records
        # beyond the second call's count, which must not matter
        for first, n, dz, seed in ((0, nw, 0.04, 1), (700, 1000, 0.03, 2), (nw - 1, 1, 0.02, 3)):
            moved = moved_water(tris[first:], n, dz, seed)
            a.update_triangles(first, moved)
            b.update_triangles_noalloc(first, moved)
            ra, rb = a.read_triangles_gpu(0, len(tris)), b.read_triangles_gpu(0, len(tris))
            assert np.array_equal(ra.view(np.uint8), rb.view(np.uint8)), (first, n)
            if device_built:                 # the primitive boxes follow too: the rebuilt trees render the same image
                a.rebuild_bvh()
                b.rebuild_bvh()
            for mode in (abi.MODE_PARITY, abi.MODE_FAST):
                ia, ha = a.render(max_depth=8, variant=abi.VARIANT_POOL, mode=mode)
                ib, hb = b.render(max_depth=8, variant=abi.VARIANT_POOL, mode=mode)
                assert np.array_equal(ia.view(np.uint32), ib.view(np.uint32)) and np.array_equal(ha, hb), (first, n, mode)
        # nothing is allocated: free device memory is the same before and after a run of calls of every size (hipMemGetInfo)
        import torch
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        for first, n, dz, seed in ((0, nw, 0.01, 4), (5, 4000, 0.01, 5), (0, 1, 0.01, 6), (0, nw, 0.0, 7)):
            b.update_triangles_noalloc(first, moved_water(tris[first:], n, dz, seed))
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(0)[0] == free0
        a.update_triangles(0, tris[:nw])
        b.update_triangles_noalloc(0, tris[:nw])
        if device_built:
            a.rebuild_bvh()
            b.rebuild_bvh()
        base, _ = a.render(max_depth=8, variant=abi.VARIANT_POOL)
        b.update_triangles_noalloc(0, tris[:0])          # empty input: nothing happens
        same, _ = b.render(max_depth=8, variant=abi.VARIANT_POOL)
        assert np.array_equal(base, same)
    finally:
        a.close()
        b.close()


def test_noalloc_limits_are_refused(pool_model):
    t = T.Tracer(0)
    try:
        m = scenes.mesh_scene(subdivisions=0, builder="median")          # 77 560 triangles: more than the staging area holds
        t.scene_upload(m.scene_view())
        tris = m.arrays()["triangles"]
        with pytest.raises(T.TraceError) as e:
            t.update_triangles_noalloc(0, tris[:NOALLOC_MAX + 1])
        assert e.value.code == 6 and "LRL_NOALLOC_MAX_TRIANGLES" in str(e.value)          # LRL_ERR_LIMIT
        before = t.read_triangles_gpu(0, 8)
        t.update_triangles_noalloc(0, tris[:NOALLOC_MAX])                 # the maximum itself is served
        strange = tris[:4].copy()
        strange["color"]["d"] = (0.123, 0.456, 0.789)                    # a material the scene does not have
        with pytest.raises(T.TraceError) as e:
            t.update_triangles_noalloc(0, strange)
        assert e.value.code == 6 and "material" in str(e.value)
        assert np.array_equal(before.view(np.uint8), t.read_triangles_gpu(0, 8).view(np.uint8))      # a refused call changes nothing
        with pytest.raises(T.TraceError):
            t.update_triangles_noalloc(len(tris) - 2, tris[:4])           # out of range
    finally:
        t.close()
