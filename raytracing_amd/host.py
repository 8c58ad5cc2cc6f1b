"""ctypes front end of liblrlhost.so (raytracing_amd/host): the C++ scene / BVH / camera library
that restates the reference's RayTracing::Model, Transform, STL and BMP types.  Used by tests,
bench.py and tools; the device ABI (tracer.py) takes the SceneView this produces.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "host", "liblrlhost.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(_LIB_PATH)
        L.lrlh_model_create.restype = C.c_void_p
        L.lrlh_last_error.restype = C.c_char_p
        L.lrlh_model_destroy.argtypes = [C.c_void_p]
        L.lrlh_model_destroy.restype = None
        L.lrlh_sheared.restype = None
        _lib = L
    return _lib


class HostError(RuntimeError):
    pass


class HostModel:
    """Owns one lrl::Model (+ its fixture camera) on the C++ side."""

    def __init__(self):
        self._L = lib()
        self._h = C.c_void_p(self._L.lrlh_model_create())
        if not self._h:
            raise MemoryError("lrlh_model_create")
        self._keep = []

    def close(self):
        if self._h:
            self._L.lrlh_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise HostError(self._L.lrlh_last_error().decode())

    # ---- fixtures ----
    def scene_bvh_default(self, resdir, load_cube=True):
        self._ck(self._L.lrlh_scene_bvh_default(self._h, os.fsencode(resdir), int(load_cube)))

    def scene_mesh(self, stl_path, subdivisions=0, grid=1):
        self._ck(self._L.lrlh_scene_mesh_grid(self._h, os.fsencode(stl_path), C.c_uint(subdivisions), C.c_uint(grid)))

    def scene_pool(self, resdir, load_textures=True):
        self._ck(self._L.lrlh_scene_pool(self._h, os.fsencode(resdir), int(load_textures)))

    # ---- generic adders ----
    def _add(self, fn, arr, dt):
        arr = np.ascontiguousarray(np.atleast_1d(arr), dtype=dt)
        self._ck(fn(self._h, arr.ctypes.data_as(C.c_void_p), C.c_uint(arr.shape[0])))

    def add_triangles(self, a):
        self._add(self._L.lrlh_add_triangles, a, abi.TRIANGLE)

    def add_spheres(self, a):
        self._add(self._L.lrlh_add_spheres, a, abi.SPHERE)

    def add_circles(self, a):
        self._add(self._L.lrlh_add_circles, a, abi.CIRCLE)

    def add_point_lights(self, a):
        self._add(self._L.lrlh_add_point_lights, a, abi.POINTLIGHT)

    def add_cylinder(self, rec):
        rec = np.ascontiguousarray(np.atleast_1d(rec), dtype=abi.CYLINDER)
        for i in range(rec.shape[0]):
            self._ck(self._L.lrlh_add_cylinder(self._h, rec[i:i + 1].ctypes.data_as(C.c_void_p)))

    def add_cone(self, rec):
        rec = np.ascontiguousarray(np.atleast_1d(rec), dtype=abi.CONE)
        for i in range(rec.shape[0]):
            self._ck(self._L.lrlh_add_cone(self._h, rec[i:i + 1].ctypes.data_as(C.c_void_p)))

    def add_stl(self, path, color, n=0):
        c = np.ascontiguousarray(np.atleast_1d(color), dtype=abi.COLOR)
        self._ck(self._L.lrlh_add_stl(self._h, os.fsencode(path), c.ctypes.data_as(C.c_void_p), C.c_uint(n)))

    def set_cube_dir(self, d):
        d = d if d.endswith("/") else d + "/"
        self._ck(self._L.lrlh_set_cube_dir(self._h, os.fsencode(d)))

    def set_cube_raw(self, faces_bgr):
        """faces_bgr: uint8 array [6, N, N, 3] (BGR, rows bottom-up), order +X,-X,+Y,-Y,+Z,-Z."""
        f = np.ascontiguousarray(faces_bgr, dtype=np.uint8)
        assert f.ndim == 4 and f.shape[0] == 6 and f.shape[1] == f.shape[2] and f.shape[3] == 3
        ptrs = (C.c_void_p * 6)(*[f[i].ctypes.data for i in range(6)])
        self._ck(self._L.lrlh_set_cube_raw(self._h, C.c_uint(f.shape[1]), ptrs))

    def set_tex_raw(self, layers_bgr):
        """layers_bgr: uint8 array [L, H, W, 3] (BGR, rows bottom-up)."""
        t = np.ascontiguousarray(layers_bgr, dtype=np.uint8)
        assert t.ndim == 4 and t.shape[3] == 3
        self._ck(self._L.lrlh_set_tex_raw(self._h, C.c_uint(t.shape[2]), C.c_uint(t.shape[1]), C.c_uint(t.shape[0]),
                                          t.ctypes.data_as(C.c_void_p)))

    # ---- model.dataInit() ----
    def data_init(self, builder="reference"):
        self._ck(self._L.lrlh_data_init(self._h, {"reference": 0, "median": 1, "none": 2, "sah": 3}[builder]))      # "none": counts only, no tree

    def scene_view(self):
        v = abi.SceneView()
        self._ck(self._L.lrlh_scene_view(self._h, C.byref(v)))
        return v

    def bvh_info(self):
        n, d, f = C.c_uint(), C.c_uint(), C.c_uint()
        self._L.lrlh_bvh_info(self._h, C.byref(n), C.byref(d), C.byref(f))
        return {"nodes": n.value, "max_depth": d.value, "fallback_splits": f.value}

    def arrays(self):
        return abi.scene_view_arrays(self.scene_view())

    # ---- camera ----
    def camera(self, height):
        t = np.zeros(1, dtype=abi.TRANS)
        self._ck(self._L.lrlh_camera(self._h, C.c_uint(height), t.ctypes.data_as(C.c_void_p)))
        return t

    def camera_set(self, fovy, r0, trans3x3):
        r = (C.c_double * 3)(*[float(x) for x in r0])
        m = (C.c_double * 9)(*[float(x) for x in np.asarray(trans3x3, dtype=np.float64).reshape(9)])
        self._ck(self._L.lrlh_camera_set(self._h, C.c_double(fovy), r, m))

    def camera_look_at(self, fovy, eye, target, up=(0, 0, 1)):
        eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
        f = target - eye
        f /= np.linalg.norm(f)
        r = np.cross(f, up)
        r /= np.linalg.norm(r)
        u = np.cross(r, f)
        self.camera_set(fovy, eye, np.stack([r, u, -f], axis=1))

    def scene_dump(self, path):
        self._ck(self._L.lrlh_scene_dump(self._h, os.fsencode(path)))


def sheared(trans, dx, dy):
    """Sub-pixel jitter (dx, dy) in pixels as a shear of `trans`."""
    t = np.ascontiguousarray(trans, dtype=abi.TRANS).reshape(1)
    o = np.zeros(1, dtype=abi.TRANS)
    lib().lrlh_sheared(t.ctypes.data_as(C.c_void_p), C.c_float(dx), C.c_float(dy), o.ctypes.data_as(C.c_void_p))
    return o


def halton(n, base):
    out = np.zeros(n, dtype=np.float64)
    for i in range(n):
        f, r, k = 1.0, 0.0, i + 1
        while k > 0:
            f /= base
            r += f * (k % base)
            k //= base
        out[i] = r
    return out


def jitter_table(spp):
    """Halton(2,3) sub-pixel offsets in [-0.5, 0.5)^2; sample 0 of a 1-spp render is the pixel centre."""
    if spp == 1:
        return np.zeros((1, 2), dtype=np.float32)
    return np.stack([halton(spp, 2) - 0.5, halton(spp, 3) - 0.5], axis=1).astype(np.float32)


def sample_trans(trans, jitter):
    return np.concatenate([sheared(trans, float(j[0]), float(j[1])) if (j[0] != 0 or j[1] != 0) else
                           np.ascontiguousarray(trans, dtype=abi.TRANS).reshape(1) for j in jitter])
