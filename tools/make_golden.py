#!/usr/bin/env python3
"""Generates assets/ (data assets taken from the reference's resources) and tests/golden/ (golden vectors
rendered by the REFERENCE'S OWN GLSL on Mesa llvmpipe (oracle/_ref/gl_harness).

Runs only where /root/reference is mounted (this container); the outputs are committed so the GPU box,
which has no /root/reference, can check the oracle and the HIP path against them.

  python tools/make_golden.py [--reference /root/reference]
"""
import argparse
import os
import shutil
import struct
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import oracle_api as O  # noqa: E402
from raytracing_amd import abi, host, scenes  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")


def read_bmp(path):
    b = open(path, "rb").read()
    off, = struct.unpack_from("<I", b, 10)
    w, h = struct.unpack_from("<ii", b, 18)
    bpp, = struct.unpack_from("<H", b, 28)
    assert bpp == 24 and h > 0
    stride = (w * 3 + 3) & ~3
    a = np.frombuffer(b, dtype=np.uint8, count=stride * h, offset=off).reshape(h, stride)[:, :w * 3].reshape(h, w, 3)
    return a.copy()   # BGR, bottom-up


def write_bmp(path, a):
    h, w, _ = a.shape
    stride = (w * 3 + 3) & ~3
    rows = np.zeros((h, stride), dtype=np.uint8)
    rows[:, :w * 3] = a.reshape(h, w * 3)
    hdr = b"BM" + struct.pack("<IHHI", 54 + stride * h, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, stride * h, 2835, 2835, 0, 0)
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(rows.tobytes())


def box_filter(a, k):
    h, w, c = a.shape
    return np.round(a.reshape(h // k, k, w // k, k, c).astype(np.float64).mean(axis=(1, 3))).astype(np.uint8)


def make_assets(ref):
    res = os.path.join(ref, "BVH", "resources")
    dst = scenes.BVH_RESOURCES
    for sub in ("box", "three", "room"):
        os.makedirs(os.path.join(dst, sub), exist_ok=True)
    for rel in ["box/box1.stl", "box/box2.stl", "box/box3.stl", "box/boxtransparent1.stl", "box/boxtransparent2.stl", "mirror.stl",
                "three/three1.stl", "three/three2.stl", "three/three3.stl", "three/three transparent1.stl", "three/three transparent2.stl"]:
        shutil.copyfile(os.path.join(res, rel), os.path.join(dst, rel))
    for face in ("front", "back", "down", "up", "right", "left"):
        a = read_bmp(os.path.join(res, "room", face + ".bmp"))
        write_bmp(os.path.join(dst, "room", face + ".bmp"), box_filter(a, a.shape[0] // 128))
    # Stanford bunny: vertex table + indices, winding fixed the way Model::addSTL does it
    b = open(os.path.join(res, "Stanford_bunny.stl"), "rb").read()
    n, = struct.unpack_from("<I", b, 80)
    rec = np.frombuffer(b, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
    v = rec["v"].copy()
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    cr = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2], e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], axis=1)
    nr = rec["n"]
    s = (cr[:, 0] * nr[:, 0] + cr[:, 1] * nr[:, 1]) + cr[:, 2] * nr[:, 2]
    flip = ~(s > 0)
    v[flip] = v[flip][:, [0, 2, 1]]
    verts, inv = np.unique(v.reshape(-1, 3), axis=0, return_inverse=True)
    np.savez_compressed(scenes.BUNNY, verts=verts.astype(np.float32), idx=inv.reshape(-1, 3).astype(np.uint32))
    print(f"bunny: {n} triangles, {verts.shape[0]} vertices, {int(flip.sum())} flipped")


def make_pool_assets(ref):
    """Pool textures box-filtered 1024^2 -> 128^2 (floor.bmp 1016x1024 -> 127x128: it stays narrower than its layer, as in
    the reference, Pool/Pool.cpp:491-492), the room cube 512^2 -> 128^2, pool.stl as it is."""
    res = os.path.join(ref, "Pool", "resources")
    dst = scenes.POOL_RESOURCES
    os.makedirs(os.path.join(dst, "room"), exist_ok=True)
    shutil.copyfile(os.path.join(res, "pool.stl"), os.path.join(dst, "pool.stl"))
    for name in ("pool", "brick", "ceiling", "floor"):
        a = read_bmp(os.path.join(res, name + ".bmp"))
        write_bmp(os.path.join(dst, name + ".bmp"), box_filter(a, 8))
    for face in ("front", "back", "down", "up", "right", "left"):
        a = read_bmp(os.path.join(res, "room", face + ".bmp"))
        write_bmp(os.path.join(dst, "room", face + ".bmp"), box_filter(a, a.shape[0] // 128))


def make_pool_goldens(ref):
    """Pool variant of the fragment shader (no checker, RayTraceDepth 8) on the Pool demo scene."""
    shaders = os.path.join(ref, "Pool", "shaders")
    m = scenes.pool()
    tA = m.camera(160)                     # the reference's start pose: (0,0,1) looking down at the water
    save("pool_viewA_160", O.run_reference(m, tA, 160, 160, shaders), tA, dict(width=160, height=160, depth=8, variant=1))
    m.camera_look_at(60.0, (0.42, -0.44, 0.9), (0.0, 0.1, 0.35))      # oblique: walls, ceiling, water, pool bottom
    tB = m.camera(192)
    save("pool_viewB_192", O.run_reference(m, tB, 192, 192, shaders), tB, dict(width=192, height=192, depth=8, variant=1))
    m.camera_look_at(60.0, (0.2, -0.2, 0.2), (0.0, 0.3, 0.45))        # camera under water: DecayOriginPre (+z probe) decides the medium
    tC = m.camera(128)
    save("pool_underwater_128", O.run_reference(m, tC, 128, 128, shaders, decay_compute=True), tC, dict(width=128, height=128, depth=8, variant=1))


def make_water_golden(ref):
    """The reference's water simulation on llvmpipe (WaterAcceleration / WaterPosition / WaterPositionSync) from the Pool
    scene's rest state: the grid after 1 and after 50 substeps, and the z of the three corners of every water triangle after
    the sync that follows them (Pool/Pool.cpp:224-235, n = 50 instead of 200 to keep the fixture's run short)."""
    shaders = os.path.join(ref, "Pool", "shaders")
    m = scenes.pool()
    tris = m.arrays()["triangles"]
    gx, gy = scenes.POOL_GRID
    k, dt = scenes.POOL_WATER
    w0 = scenes.pool_water_points(m)
    out = {"gx": gx, "gy": gy, "k": np.float32(k), "dt": np.float32(dt), "z0": w0["z"].copy()}
    for n in (1, 50):
        w, t = O.run_reference_water(w0, gx, gy, k, dt, n, tris, shaders)
        nw = 2 * (gx - 1) * (gy - 1)
        out[f"zva_{n}"] = np.stack([w["z"], w["v"], w["a"]], axis=1)
        out[f"tri_z_{n}"] = np.stack([t["p1"][:nw, 2], t["p2"][:nw, 2], t["p3"][:nw, 2]], axis=1)
        assert np.array_equal(t[nw:].view(np.uint8), tris[nw:].view(np.uint8))      # only the water triangles move
    np.savez_compressed(os.path.join(GOLDEN, "pool_water_steps.npz"), **out)
    print("pool_water_steps", {k: getattr(v, "shape", v) for k, v in out.items()})


def save(name, ref, trans, meta):
    ids = ref["ids"]
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), rgb=ref["img"][..., :3].copy(), hit_type=ids[..., 0].astype(np.uint8),
                        hit_idx=ids[..., 1].astype(np.uint32), t=ref["t"], trans=np.ascontiguousarray(trans).view(np.uint8),
                        decay=np.frombuffer(ref["decay"].tobytes(), dtype=np.float32), **meta)
    print(name, ref["timing"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--only", default="", help="'pool': only the Pool assets and goldens; 'water': only the water-simulation fixture")
    a = ap.parse_args()
    ref = a.reference
    os.makedirs(GOLDEN, exist_ok=True)
    if a.only == "pool":
        make_pool_assets(ref)
        make_pool_goldens(ref)
        make_water_golden(ref)
        return
    if a.only == "water":
        make_water_golden(ref)
        return
    make_assets(ref)
    bvh_shaders = os.path.join(ref, "BVH", "shaders")

    m = scenes.bvh_default()
    info = m.bvh_info()
    arrays = m.arrays()
    np.savez_compressed(os.path.join(GOLDEN, "bvh_default_scene.npz"), bvh=arrays["bvh"].view(np.uint8), nodes=info["nodes"], max_depth=info["max_depth"])

    # view A: the reference's start pose (r0 = (0,0,30), trans = I), as shipped (RayTraceDepth 10)
    tA = m.camera(256)
    r = O.run_reference(m, tA, 256, 256, bvh_shaders, want_pre=True)
    save("bvh_viewA_256", r, tA, dict(width=256, height=256, depth=10, variant=0))
    np.savez_compressed(os.path.join(GOLDEN, "bvh_default_pre.npz"), tri=r["tri"].view(np.uint8), circles=r["circles"].view(np.uint8))

    # view B: oblique overview that sees every primitive class
    scenes.overview_camera(m)
    tB = m.camera(256)
    save("bvh_viewB_256", O.run_reference(m, tB, 256, 256, bvh_shaders), tB, dict(width=256, height=256, depth=10, variant=0))
    tB192 = m.camera(192)
    save("bvh_viewB_192_depth4", O.run_reference(m, tB192, 192, 192, bvh_shaders, max_depth=4), tB192, dict(width=192, height=192, depth=4, variant=0))
    # 4 jittered samples (Halton 2,3) through the `trans` shear
    tB128 = m.camera(128)
    jit = host.jitter_table(4)
    tS = host.sample_trans(tB128, jit)
    save("bvh_viewB_128_spp4", O.run_reference(m, tS, 128, 128, bvh_shaders), tB128, dict(width=128, height=128, depth=10, variant=0, jitter=jit))

    # camera inside the first glass box: DecayOriginPre / DecayOrigin decide the medium at the eye
    m.camera_look_at(60.0, (22.0, 34.0, 3.0), (0.0, 10.0, 8.0))
    tG = m.camera(128)
    save("bvh_inside_glass_128", O.run_reference(m, tG, 128, 128, bvh_shaders, decay_compute=True), tG, dict(width=128, height=128, depth=10, variant=0))
    scenes.overview_camera(m)
    rD = O.run_reference(m, m.camera(64), 64, 64, bvh_shaders, decay_compute=True, ids=False)
    np.savez_compressed(os.path.join(GOLDEN, "bvh_decay_air.npz"), decay=np.frombuffer(rD["decay"].tobytes(), dtype=np.float32))

    # C2-shaped scene at its base resolution: bunny (no subdivision) on the ground disc, 4 bounces
    mm = scenes.mesh_scene(subdivisions=0, builder="median")
    tM = mm.camera(135)
    save("mesh_bunny0_240x135_depth4", O.run_reference(mm, tM, 240, 135, bvh_shaders, max_depth=4), tM, dict(width=240, height=135, depth=4, variant=0))

    make_pool_assets(ref)
    make_pool_goldens(ref)
    make_water_golden(ref)


if __name__ == "__main__":
    main()
