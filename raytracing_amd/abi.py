"""ctypes / numpy mirrors of include/lrl_types.h and include/lrl_trace.h.

Plumbing only: the Python side never computes anything on the hot path, it hands numpy-owned
host arrays (in the reference's record layouts) to the C ABI.
"""
import ctypes as C

import numpy as np

# ---- numpy record dtypes (sizes asserted below) -------------------------------------------------
COLOR = np.dtype([("r", "<f4", 3), ("texR", "<i4"), ("t", "<f4", 3), ("texT", "<i4"), ("d", "<f4", 3), ("texD", "<i4"),
                  ("g", "<f4", 3), ("texG", "<i4"), ("decayFactor", "<f4", 3), ("n", "<f4")])
BVHNODE = np.dtype([("min", "<f4", 3), ("leftChild", "<i4"), ("max", "<f4", 3), ("rightChild", "<i4"), ("father", "<u4"),
                    ("axis", "<u4"), ("geometry", "<u4"), ("geometryNum", "<u4"), ("blank", "<f4", 4)])
TRIANGLE = np.dtype([("p1", "<f4", 3), ("_0", "<f4"), ("p2", "<f4", 3), ("_1", "<f4"), ("p3", "<f4", 3), ("_2", "<f4"),
                     ("uv1", "<f4", 2), ("uv2", "<f4", 2), ("uv3", "<f4", 2), ("changed", "<i4"), ("_3", "<i4"), ("color", COLOR)])
TRIANGLE_GPU = np.dtype([("plane", "<f4", 4), ("p1", "<f4", 3), ("_0", "<f4"), ("k1", "<f4", 3), ("_1", "<f4"), ("k2", "<f4", 3),
                         ("_2", "<f4"), ("uv1", "<f4", 2), ("uv2", "<f4", 2), ("uv3", "<f4", 2), ("_3", "<f4", 2), ("color", COLOR)])
SPHERE = np.dtype([("sphere", "<f4", 4), ("e1", "<f4", 3), ("_0", "<f4"), ("e2", "<f4", 3), ("_1", "<f4"), ("color", COLOR)])
CIRCLE = np.dtype([("plane", "<f4", 4), ("sphere", "<f4", 4), ("e1", "<f4", 3), ("tex", "<i4"), ("color", COLOR)])
CYLINDER = np.dtype([("c", "<f4", 3), ("r2", "<f4"), ("n", "<f4", 3), ("l", "<f4"), ("e1", "<f4", 3), ("_0", "<f4"), ("color", COLOR)])
CONE = np.dtype([("c", "<f4", 3), ("c2", "<f4"), ("n", "<f4", 3), ("l2", "<f4"), ("e1", "<f4", 3), ("_0", "<f4"), ("color", COLOR)])
POINTLIGHT = np.dtype([("color", "<f4", 3), ("_0", "<f4"), ("p", "<f4", 3), ("_1", "<f4")])
TRANS = np.dtype([("trans", "<f4", (3, 4)), ("r0", "<f4", 3), ("z0", "<f4")])
DECAY_ORIGIN = np.dtype([("decayOrigins", "<f4", (8, 4)), ("decayOrigin", "<f4", 4)])

for _dt, _sz in ((COLOR, 80), (BVHNODE, 64), (TRIANGLE, 160), (TRIANGLE_GPU, 176), (SPHERE, 128), (CIRCLE, 128), (CYLINDER, 128),
                 (CONE, 128), (POINTLIGHT, 32), (TRANS, 64), (DECAY_ORIGIN, 144)):
    assert _dt.itemsize == _sz, (_dt, _dt.itemsize, _sz)

STRIPE_ROWS = 16          # LRL_STRIPE_ROWS
GEOM_TRIANGLE, GEOM_SPHERE, GEOM_CIRCLE, GEOM_CYLINDER, GEOM_CONE = 2, 3, 4, 5, 6
VARIANT_BVH, VARIANT_POOL = 0, 1
MODE_PARITY, MODE_FAST = 0, 1
FLAG_COUNTERS, FLAG_OUT_DEVICE, FLAG_MEGAKERNEL, FLAG_TIME_KERNELS, FLAG_DIFFUSE_BOUNCE, FLAG_COUNT_FETCHES = 1, 2, 4, 8, 16, 32


def color(r=0.0, t=0.0, d=0.0, g=0.0, decay=0.0, n=1.0, texR=-1, texT=-1, texD=-1, texG=-1):
    c = np.zeros((), dtype=COLOR)
    c["r"], c["t"], c["d"], c["g"], c["decayFactor"], c["n"] = r, t, d, g, decay, n
    c["texR"], c["texT"], c["texD"], c["texG"] = texR, texT, texD, texG
    return c


# ---- ctypes structures --------------------------------------------------------------------------
class GeometryNum(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("planeNum", "triangleNum", "sphereNum", "circleNum", "cylinderNum", "coneNum",
                                          "pointLightNum", "_pad")]


class SceneView(C.Structure):
    _fields_ = [("num", GeometryNum), ("trianglesOrigin", C.c_void_p), ("trianglesGPU", C.c_void_p), ("spheres", C.c_void_p),
                ("circles", C.c_void_p), ("cylinders", C.c_void_p), ("cones", C.c_void_p), ("pointLights", C.c_void_p),
                ("bvh", C.c_void_p), ("bvhNodeNum", C.c_uint32), ("texWidth", C.c_uint32), ("texHeight", C.c_uint32),
                ("texLayers", C.c_uint32), ("texBGR", C.c_void_p), ("cubeSize", C.c_uint32), ("cubeBGR", C.c_void_p * 6)]


class RenderParams(C.Structure):
    _fields_ = [("maxDepth", C.c_uint32), ("variant", C.c_uint32), ("mode", C.c_uint32), ("flags", C.c_uint32), ("spp", C.c_uint32),
                ("jitter", C.POINTER(C.c_float)), ("rowBegin", C.c_uint32), ("rowEnd", C.c_uint32), ("sampleBegin", C.c_uint32),
                ("sampleEnd", C.c_uint32), ("rowInterleave", C.c_uint32), ("rowShard", C.c_uint32)]


class Tuning(C.Structure):
    """LrlTuning (include/lrl_trace.h): scheduling knobs of the FAST pipeline."""
    _fields_ = [("handoverEarly", C.c_uint32), ("handoverLate", C.c_uint32), ("handoverListLen", C.c_uint32), ("handoverList", C.c_uint32 * 8),
                ("adapt", C.c_uint32), ("groupLanes", C.c_uint32), ("contSlots", C.c_uint32), ("groupCapX", C.c_uint32), ("sparseFrom", C.c_uint32),
                ("compactFrom", C.c_int32), ("workLow", C.c_float), ("pollLag", C.c_uint32), ("traceRef", C.c_uint32), ("batchSlots", C.c_uint32),
                ("coopFetch", C.c_uint32), ("groupQuad", C.c_uint32), ("streamWords", C.c_uint32), ("refillMin", C.c_uint32), ("phasedTrips", C.c_uint32), ("deviceBuilder", C.c_uint32), ("debug", C.c_uint32), ("treeLayout", C.c_uint32), ("deterministic", C.c_uint32), ("tailFrom", C.c_uint32)]


class SceneInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("bvhNodes", "bvhRecords", "bvhDepth", "wideDepth", "deviceBuilt", "materials", "ownBoxes", "wideNodes", "leafOrder")]


class RenderInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("iterations", "passes", "batch", "compactFrom", "groupLanes", "frontierEntries", "contSlots",
                                          "contStride", "coopFetch", "streamWords", "phasedTrips")] + [("handoverTrips", C.c_uint32 * 8), ("trialState", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("raysPrimary", "raysReflect", "raysRefract", "raysShadow", "nodeVisits", "triTests",
                                          "sphereTests", "otherTests", "primaryHits", "recordFetches", "laneTrips", "waveTrips", "groupLanes", "groupRounds")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    @property
    def rays(self):
        return int(self.raysPrimary + self.raysReflect + self.raysRefract + self.raysShadow)

    def algorithmic_bytes(self):
        """32 B per node visit, 52 B per triangle test, 32 B per sphere test, 128 B per other test,
        32 B ray in + 16 B result out per ray."""
        return int(32 * self.nodeVisits + 52 * self.triTests + 32 * self.sphereTests + 128 * self.otherTests + 48 * self.rays)


class OracleParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("maxDepth", C.c_uint32), ("variant", C.c_uint32),
                ("nSamples", C.c_uint32), ("rowBegin", C.c_uint32), ("rowEnd", C.c_uint32), ("threads", C.c_uint32),
                ("decayOrigin", C.c_float * 4), ("diffuse", C.c_uint32)]


PRIM_SPHERE, PRIM_CIRCLE, PRIM_CYLINDER, PRIM_CONE, PRIM_LIGHT = 3, 4, 5, 6, 0x100      # LRL_PRIM_* (include/lrl_trace.h)


def scene_view_arrays(view):
    """Copy the arrays a SceneView points at into numpy records (for fixtures and tests)."""
    def arr(ptr, n, dt):
        if not ptr or n == 0:
            return np.zeros(0, dtype=dt)
        buf = (C.c_uint8 * (n * dt.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).copy()
    n = view.num
    out = {
        "triangles": arr(view.trianglesOrigin, n.triangleNum, TRIANGLE),
        "spheres": arr(view.spheres, n.sphereNum, SPHERE),
        "circles": arr(view.circles, n.circleNum, CIRCLE),
        "cylinders": arr(view.cylinders, n.cylinderNum, CYLINDER),
        "cones": arr(view.cones, n.coneNum, CONE),
        "pointLights": arr(view.pointLights, n.pointLightNum, POINTLIGHT),
        "bvh": arr(view.bvh, view.bvhNodeNum, BVHNODE),
    }
    return out

# Pool water simulation point (Pool/Pool.cpp:84-90)
WATER_POINT = np.dtype([("z", "<f4"), ("v", "<f4"), ("a", "<f4"), ("blank", "<f4")])
assert WATER_POINT.itemsize == 16
