"""TEST INFRASTRUCTURE — ctypes front end of oracle/liboracle.so (the CPU restatement) and a runner
for oracle/_ref/gl_harness (the reference's own GLSL on llvmpipe).  Only tests/, smoke() and
bench.py's cpu_baseline leg import this module; nothing under raytracing_amd/ does.
"""
import ctypes as C
import json
import os
import subprocess
import tempfile

import numpy as np

from raytracing_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
HARNESS = os.path.join(_HERE, "_ref", "gl_harness")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle` or __graft_entry__.build()")
        _lib = C.CDLL(LIB_PATH)
    return _lib


def render(view, trans, width, height, max_depth=10, variant=abi.VARIANT_BVH, decay_origin=(0, 0, 0, 1), rows=None, threads=1,
           want_hits=True, diffuse=False):
    """-> (rgba [rows, W, 4] float32, hits [rows, W, 2] uint32 or None, abi.Counters); diffuse: LRL_FLAG_DIFFUSE_BOUNCE"""
    trans = np.ascontiguousarray(trans, dtype=abi.TRANS).reshape(-1)
    r0, r1 = rows if rows else (0, height)
    p = abi.OracleParams(width, height, max_depth, variant, trans.shape[0], r0, r1, threads, (C.c_float * 4)(*decay_origin), 1 if diffuse else 0)
    img = np.zeros((r1 - r0, width, 4), dtype=np.float32)
    hits = np.zeros((r1 - r0, width, 2), dtype=np.uint32) if want_hits else None
    cn = abi.Counters()
    rc = lib().oracle_render(C.byref(view), trans.ctypes.data_as(C.c_void_p), C.byref(p), img.ctypes.data_as(C.c_void_p),
                             hits.ctypes.data_as(C.c_void_p) if want_hits else None, C.byref(cn))
    if rc != 0:
        raise RuntimeError(f"oracle_render failed ({rc})")
    return img, hits, cn


def render_bruteforce(view, trans, width, height, pixels, threads=1):
    """Tree-free closest hit of the primary rays of `pixels` ([n, 2] of x, y): every primitive against every ray, the oracle's own
    test arithmetic, no BVH (oracle_bruteforce_primary).  -> (hits [n, 2] uint32 {type, idx}, t [n] float32, -1 = miss)"""
    trans = np.ascontiguousarray(trans, dtype=abi.TRANS).reshape(-1)
    px = np.ascontiguousarray(pixels, dtype=np.uint32).reshape(-1, 2)
    n = px.shape[0]
    hits = np.zeros((n, 2), dtype=np.uint32)
    t = np.zeros(n, dtype=np.float32)
    f = lib().oracle_bruteforce_primary
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    rc = f(C.byref(view), trans.ctypes.data_as(C.c_void_p), width, height, n, px.ctypes.data_as(C.c_void_p), threads,
           hits.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(f"oracle_bruteforce_primary failed ({rc})")
    return hits, t


def tie_probe(view, trans, width, height, pixels, prims):
    """For n {pixel, primitive} pairs: (t, signed distance to the primitive's boundary, rounding band), see oracle_tie_probe."""
    trans = np.ascontiguousarray(trans, dtype=abi.TRANS).reshape(-1)
    px = np.ascontiguousarray(pixels, dtype=np.uint32).reshape(-1, 2)
    pr = np.ascontiguousarray(prims, dtype=np.uint32).reshape(-1, 2)
    n = px.shape[0]
    t, sd, tol = (np.zeros(n, dtype=np.float32) for _ in range(3))
    if n == 0:
        return t, sd, tol
    f = lib().oracle_tie_probe
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = f(C.byref(view), trans.ctypes.data_as(C.c_void_p), width, height, n, px.ctypes.data_as(C.c_void_p), pr.ctypes.data_as(C.c_void_p),
           t.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p), tol.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(f"oracle_tie_probe failed ({rc})")
    return t, sd, tol


def tie_adjusted_match(a, b, view, trans, width, height, pixels=None, row0=0):
    """Share of pixels whose primary-hit ids {type, idx} agree or are a PROVEN floating-point tie (north_star: "mismatches allowed only
    at floating-point tie edges").  a, b: [rows, W, 2] id images (rows row0.. of a width x height frame), or [n, 2] lists with
    `pixels` [n, 2] = their {x, y}.  For every mismatching pixel the primary ray is intersected, in double precision, with both
    primitives (oracle_tie_probe).  The pixel is a tie when
      * both are met at the same distance, each within the float32 rounding band of its own patch (two triangles at a shared edge or
        vertex, coincident discs, a box face lying in the ground disc), or
      * the NEARER of the two is met within the rounding band of its boundary (a silhouette edge: one arithmetic sees the primitive,
        the other sees past it; a miss counts as infinitely far).
    -> (tie-adjusted share, exact share, array of the pixels {x, y} that are neither)"""
    a, b = np.asarray(a), np.asarray(b)
    if pixels is None:
        same = (a == b).all(axis=-1)
        ys, xs = np.nonzero(~same)
        pa, pb = a[ys, xs], b[ys, xs]
        px = np.stack([xs, ys + row0], axis=1).astype(np.uint32)
    else:
        same = (a == b).all(axis=-1)
        bad = np.nonzero(~same)[0]
        pa, pb = a[bad], b[bad]
        px = np.asarray(pixels, dtype=np.uint32).reshape(-1, 2)[bad]
    total = same.size
    if px.shape[0] == 0:
        return 1.0, 1.0, px
    ta, sda, tola = tie_probe(view, trans, width, height, px, pa)
    tb, sdb, tolb = tie_probe(view, trans, width, height, px, pb)
    inf = np.float32(np.inf)
    ta = np.where(ta > 0, ta, inf)
    tb = np.where(tb > 0, tb, inf)
    tol = np.maximum(tola, tolb)
    with np.errstate(invalid="ignore"):
        equal = (np.abs(ta - tb) <= tol) & (sda <= tola) & (sdb <= tolb)
        a_near = ta <= tb
        edge = np.where(a_near, np.abs(sda) <= tola, np.abs(sdb) <= tolb) & np.isfinite(np.minimum(ta, tb))
    tie = equal | edge
    return float((same.sum() + tie.sum()) / total), float(same.mean()), px[~tie]


def scattered_pixels(width, height, n, seed=1):
    """n distinct pixels {x, y} spread over the frame: a fixed low-discrepancy set (Halton 2, 3 with a seed offset), no RNG state."""
    from raytracing_amd import host
    k = n * 2 + seed
    xs = (host.halton(k, 2)[seed:] * width).astype(np.uint32)
    ys = (host.halton(k, 3)[seed:] * height).astype(np.uint32)
    seen, out = set(), []
    for x, y in zip(xs, ys):
        if (int(x), int(y)) not in seen:
            seen.add((int(x), int(y)))
            out.append((int(x), int(y)))
        if len(out) == n:
            break
    return np.array(out, dtype=np.uint32)


def philox4x32_10(counter):
    ctr = (C.c_uint32 * 4)(*counter)
    out = (C.c_uint32 * 4)()
    lib().oracle_philox4x32_10(ctr, out)
    return tuple(out)


def diffuse_direction(normal, cosi1, pixel, sample, depth):
    n = (C.c_float * 3)(*normal)
    out = (C.c_float * 3)()
    lib().oracle_diffuse_direction(n, C.c_float(cosi1), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(depth), out)
    return np.array(out, dtype=np.float32)


def triangle_pre(tris):
    tris = np.ascontiguousarray(tris, dtype=abi.TRIANGLE)
    out = np.zeros(tris.shape[0], dtype=abi.TRIANGLE_GPU)
    lib().oracle_triangle_pre(tris.ctypes.data_as(C.c_void_p), C.c_uint32(tris.shape[0]), out.ctypes.data_as(C.c_void_p))
    return out


def circle_pre(circles):
    out = np.ascontiguousarray(circles, dtype=abi.CIRCLE).copy()
    lib().oracle_circle_pre(out.ctypes.data_as(C.c_void_p), C.c_uint32(out.shape[0]))
    return out


def decay_origin(view, trans, variant=abi.VARIANT_BVH):
    trans = np.ascontiguousarray(trans, dtype=abi.TRANS).reshape(-1)
    out = np.zeros(1, dtype=abi.DECAY_ORIGIN)
    lib().oracle_decay_origin(C.byref(view), trans.ctypes.data_as(C.c_void_p), C.c_uint32(variant), out.ctypes.data_as(C.c_void_p))
    return out[0]


def water_run(points, gx, gy, k, dt, substeps):
    """n x (WaterAcceleration, WaterPosition) on a copy of `points` (abi.WATER_POINT[gx*gy]) -> new points"""
    w = np.ascontiguousarray(points, dtype=abi.WATER_POINT).copy()
    assert w.shape[0] == gx * gy
    lib().oracle_water_run(w.ctypes.data_as(C.c_void_p), C.c_uint32(gx), C.c_uint32(gy), C.c_float(k), C.c_float(dt), C.c_uint32(substeps))
    return w


def water_sync(points, gx, gy, tris):
    """WaterPositionSync on a copy of `tris` (abi.TRIANGLE[>= 2*(gx-1)*(gy-1)]) -> new triangles"""
    w = np.ascontiguousarray(points, dtype=abi.WATER_POINT)
    t = np.ascontiguousarray(tris, dtype=abi.TRIANGLE).copy()
    assert w.shape[0] == gx * gy and t.shape[0] >= 2 * (gx - 1) * (gy - 1)
    lib().oracle_water_sync(w.ctypes.data_as(C.c_void_p), C.c_uint32(gx), C.c_uint32(gy), t.ctypes.data_as(C.c_void_p))
    return t


# ---- the reference itself, on llvmpipe -----------------------------------------------------------
def harness_available(shader_dir):
    return os.path.exists(HARNESS) and os.path.isdir(shader_dir) and os.path.exists("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so")


def run_reference(host_model, trans, width, height, shader_dir, max_depth=None, ids=True, min_linear=True, decay=None,
                  decay_compute=False, want_pre=False, threads=None):
    """Render `host_model` (raytracing_amd.host.HostModel, after data_init) with the reference's GLSL
    from `shader_dir`.  -> dict(img, ids, t, tri, circles, decay, timing)"""
    trans = np.ascontiguousarray(trans, dtype=abi.TRANS).reshape(-1)
    with tempfile.TemporaryDirectory() as td:
        scene = os.path.join(td, "scene.lrls")
        host_model.scene_dump(scene)
        tf = os.path.join(td, "trans.bin")
        trans.tofile(tf)
        cmd = [HARNESS, "render", "--shaders", shader_dir, "--scene", scene, "--trans", tf, "--size", str(width), str(height),
               "--out", os.path.join(td, "img.bin"), "--decay-out", os.path.join(td, "decay.bin")]
        if max_depth is not None:
            cmd += ["--depth", str(max_depth)]
        if ids:
            cmd += ["--ids", os.path.join(td, "ids.bin")]
        if min_linear:
            cmd += ["--min-linear"]
        if decay is not None:
            cmd += ["--decay"] + [repr(float(x)) for x in decay]
        if decay_compute:
            cmd += ["--decay-compute"]
        if want_pre:
            cmd += ["--tri-out", os.path.join(td, "tri.bin"), "--circle-out", os.path.join(td, "circle.bin")]
        env = dict(os.environ)
        if threads:
            env["LP_NUM_THREADS"] = str(threads)
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        if r.returncode != 0:
            raise RuntimeError("gl_harness failed:\n" + r.stderr[-4000:])
        out = {"timing": json.loads(r.stdout.strip().splitlines()[-1]), "log": r.stderr}
        out["img"] = np.fromfile(os.path.join(td, "img.bin"), dtype=np.float32).reshape(height, width, 4)
        if ids:
            raw = np.fromfile(os.path.join(td, "ids.bin"), dtype=np.uint32).reshape(height, width, 4)
            out["ids"] = raw[..., :2].copy()
            out["t"] = raw[..., 2].copy().view(np.float32)
        out["decay"] = np.fromfile(os.path.join(td, "decay.bin"), dtype=abi.DECAY_ORIGIN)[0]
        if want_pre:
            out["tri"] = np.fromfile(os.path.join(td, "tri.bin"), dtype=abi.TRIANGLE_GPU)
            out["circles"] = np.fromfile(os.path.join(td, "circle.bin"), dtype=abi.CIRCLE)
        return out


def run_reference_water(points, gx, gy, k, dt, substeps, tris, shader_dir):
    """The reference's three water compute shaders on llvmpipe: substeps x (acceleration, position), then sync.
    -> (points, triangles) after the run"""
    w = np.ascontiguousarray(points, dtype=abi.WATER_POINT)
    t = np.ascontiguousarray(tris, dtype=abi.TRIANGLE)
    with tempfile.TemporaryDirectory() as td:
        wf, tf, wo, to = (os.path.join(td, n) for n in ("w.bin", "t.bin", "wo.bin", "to.bin"))
        w.tofile(wf); t.tofile(tf)
        cmd = [HARNESS, "water", "--shaders", shader_dir, "--water", wf, "--tris", tf, "--groups", str(gx // 8), str(gy // 8),
               "--steps", str(substeps), "--k", repr(float(k)), "--dt", repr(float(dt)), "--water-out", wo, "--tri-out", to]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("gl_harness water failed:\n" + r.stderr[-4000:])
        return np.fromfile(wo, dtype=abi.WATER_POINT), np.fromfile(to, dtype=abi.TRIANGLE)
