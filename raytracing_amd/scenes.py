"""Scene fixtures shared by tests, bench.py and tools: thin Python drivers over the C++ host
library (raytracing_amd/host/lrl/Scenes.hpp) plus the location of the committed data assets.

Assets under assets/ (repo root) are DATA taken from the reference's resources by
tools/make_golden.py (small binary STL files; the `room` cube map box-filtered to 128^2; the
Stanford bunny as a de-duplicated vertex/index table).  No reference source text is stored.
"""
import os
import struct
import tempfile

import numpy as np

from . import host

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(REPO, "assets")
BVH_RESOURCES = os.path.join(ASSETS, "bvh_resources")
POOL_RESOURCES = os.path.join(ASSETS, "pool_resources")
BUNNY = os.path.join(ASSETS, "bunny_indexed.npz")


def bvh_default(resdir=BVH_RESOURCES, load_cube=True, builder="reference"):
    m = host.HostModel()
    m.scene_bvh_default(resdir, load_cube)
    m.data_init(builder)
    return m


def overview_camera(m):
    """View B of the fixtures: an oblique look at the whole default scene (view A = the reference's start pose)."""
    m.camera_look_at(60.0, (0, -45, 50), (-2, 12, 4))


def write_binary_stl(path, tris):
    """tris: float32 [n, 3, 3]; the stored normal is the geometric one, so Model::addSTL keeps the vertex order."""
    tris = np.ascontiguousarray(tris, dtype=np.float32)
    n = tris.shape[0]
    e1, e2 = tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    nrm = np.cross(e1, e2).astype(np.float64)
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = (nrm / np.where(ln > 0, ln, 1)).astype(np.float32)
    rec = np.zeros(n, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    rec["n"], rec["v"] = nrm, tris
    with open(path, "wb") as f:
        f.write(b"lrltrace fixture".ljust(80, b"\0"))
        f.write(struct.pack("<I", n))
        f.write(rec.tobytes())


def bunny_triangles():
    z = np.load(BUNNY)
    return z["verts"][z["idx"]]          # [n, 3, 3] float32, winding already fixed against the STL normals


def mesh_scene(subdivisions=2, builder="median", stl_path=None, grid=1):
    """the Stanford bunny (77 560 triangles) x `subdivisions` 1->4 midpoint levels;
    grid > 1: grid x grid replicated copies (towards C5)."""
    m = host.HostModel()
    if stl_path is None:
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "bunny.stl")
            write_binary_stl(p, bunny_triangles())
            m.scene_mesh(p, subdivisions, grid)
    else:
        m.scene_mesh(stl_path, subdivisions, grid)
    m.data_init(builder)             # "none": counts only, no host tree (LrlSceneView.bvh = NULL: the library builds one on the device)
    return m


def pool(resdir=POOL_RESOURCES, load_textures=True, builder="reference"):
    m = host.HostModel()
    m.scene_pool(resdir, load_textures)
    m.data_init(builder)
    return m


POOL_GRID = (48, 48)            # groupSize * groupNum, Pool/Pool.cpp:484
POOL_WATER = (0.0005, 0.05)     # k, dt (same line)


def pool_water_points(model, gx=POOL_GRID[0], gy=POOL_GRID[1]):
    """The water grid {z, v = 0, a = 0} of a Pool model, read back from the triangle corners the reference's
    WaterPositionSync writes (Pool/shaders/WaterPositionSyncCompute0.cpp:53-74): the state WaterData starts from."""
    from . import abi
    t = model.arrays()["triangles"]
    w = np.zeros(gx * gy, dtype=abi.WATER_POINT)
    for y in range(gy):
        for x in range(gx):
            ti = x + y * 2 * (gx - 1)
            if x != gx - 1 and y != gy - 1:
                z = t["p1"][ti][2]
            elif x != 0 and y != 0:
                z = t["p1"][ti - gx][2]
            elif x != 0:
                z = t["p2"][ti - 1][2]
            else:
                z = t["p3"][ti - 2 * (gx - 1)][2]
            w["z"][x + gx * y] = z
    return w
