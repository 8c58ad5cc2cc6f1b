"""The compiled C++ caller (examples/bvh_headless.cpp): the reference's RayTrace::init / run (BVH/BVH.cpp:712-743, Pool/Pool.cpp:637-672)
restated over lrl::Model / lrl::Transform and the C ABI, built by raytracing_amd/csrc/Makefile (and CMakeLists.txt).  On the GPU: the BMP
it writes (RGBA8, what the reference's window shows) against the llvmpipe goldens quantised the same way.  On the CPU: the CMake build."""
import os
import shutil
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import REPO, load_golden
from raytracing_amd import scenes

EXE = os.path.join(REPO, "examples", "bvh_headless")


def read_bmp(path):
    b = open(path, "rb").read()
    off, = struct.unpack_from("<I", b, 10)
    w, h = struct.unpack_from("<ii", b, 18)
    stride = (w * 3 + 3) & ~3
    a = np.frombuffer(b, dtype=np.uint8, count=stride * h, offset=off).reshape(h, stride)[:, :w * 3].reshape(h, w, 3)
    return a[..., ::-1].copy()           # RGB, row 0 = bottom (as the tracer's image)


def quantised(rgb):
    with np.errstate(invalid="ignore"):
        return np.rint(np.clip(np.nan_to_num(rgb, nan=0.0, posinf=1.0, neginf=0.0), 0, 1).astype(np.float32) * np.float32(255)).astype(np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("name,args,resdir", [("bvh_viewA_256", ["--size", "256", "256"], scenes.BVH_RESOURCES),
                                              ("bvh_viewA_256", ["--size", "256", "256", "--fast", "--device-bvh", "--frames", "12"], scenes.BVH_RESOURCES),
                                              ("pool_viewA_160", ["--pool", "--size", "160", "160"], scenes.POOL_RESOURCES)])
def test_headless_example_writes_the_reference_frame(name, args, resdir):
    assert os.path.exists(EXE), "examples/bvh_headless is missing: run __graft_entry__.build()"
    g = load_golden(name)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "frame.bmp")
        r = subprocess.run([EXE, resdir, out] + args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr + r.stdout
        img = read_bmp(out)
    want = quantised(g["rgb"])
    assert img.shape == want.shape
    d = np.abs(img.astype(np.int16) - want.astype(np.int16))
    fast = "--fast" in args
    # PARITY: the float images agree to 1e-7, so a byte differs only where a value sits on a rounding boundary; FAST: tie-edge pixels too
    assert (d > 1).any(axis=-1).mean() <= (2e-3 if fast else 1e-5), (name, args, float((d > 1).any(axis=-1).mean()), int(d.max()))
    assert (d > 0).mean() <= (1e-2 if fast else 1e-3), (name, args, float((d > 0).mean()))
    assert "decayOrigin (0, 0, 0, 1)" in r.stdout          # the eye is in air: DecayOriginPre / DecayOrigin ran


def read_ppm(path):
    b = open(path, "rb").read()
    magic, dims, maxv, rest = b.split(b"\n", 3)
    w, h = (int(x) for x in dims.split())
    assert magic == b"P6" and maxv == b"255"
    return np.frombuffer(rest, dtype=np.uint8, count=w * h * 3).reshape(h, w, 3)[::-1].copy()      # row 0 = bottom


def read_pfm(path):
    b = open(path, "rb").read()
    magic, dims, scale, rest = b.split(b"\n", 3)
    w, h = (int(x) for x in dims.split())
    assert magic == b"PF" and float(scale) < 0           # little-endian
    return np.frombuffer(rest, dtype="<f4", count=w * h * 3).reshape(h, w, 3).copy()                # rows bottom-up already


def read_exr(path):
    """A reader for what lrl::writeEXR writes and nothing more: OpenEXR 2, single part, scan lines, uncompressed, FLOAT channels
    (checked, not assumed: magic, version, every header attribute the format requires, the line offset table)."""
    b = open(path, "rb").read()
    assert b[:4] == bytes([0x76, 0x2f, 0x31, 0x01]) and struct.unpack_from("<I", b, 4)[0] == 2
    at, attrs = 8, {}
    while b[at] != 0:
        name_end = b.index(b"\0", at); type_end = b.index(b"\0", name_end + 1)
        size, = struct.unpack_from("<i", b, type_end + 1)
        attrs[b[at:name_end].decode()] = (b[name_end + 1:type_end].decode(), b[type_end + 5:type_end + 5 + size])
        at = type_end + 5 + size
    at += 1
    for need in ("channels", "compression", "dataWindow", "displayWindow", "lineOrder", "pixelAspectRatio", "screenWindowCenter", "screenWindowWidth"):
        assert need in attrs, need
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    ch, c = [], attrs["channels"][1]
    k = 0
    while c[k] != 0:
        e = c.index(b"\0", k)
        ptype, _lin, xs, ys = struct.unpack_from("<iiii", c, e + 1)
        assert ptype == 2 and xs == 1 and ys == 1          # FLOAT, full resolution
        ch.append(c[k:e].decode()); k = e + 17
    assert ch == sorted(ch) == ["A", "B", "G", "R"]
    offsets = struct.unpack_from(f"<{h}Q", b, at)
    img = np.zeros((h, w, 4), dtype=np.float32)
    for y, off in enumerate(offsets):
        yy, size = struct.unpack_from("<ii", b, off)
        assert yy == y0 + y and size == w * 4 * len(ch)
        line = np.frombuffer(b, dtype="<f4", count=w * len(ch), offset=off + 8).reshape(len(ch), w)
        for i, name in enumerate(ch):
            img[h - 1 - y, :, "RGBA".index(name)] = line[i]      # row 0 = bottom, as the tracer's image
    assert offsets[-1] + 8 + w * 4 * len(ch) == len(b)
    return img


def test_exr_writer_round_trip():
    """lrl::writeEXR (lrl/Image.hpp; SURVEY 8f rank 4: "PPM/EXR dump"): a float frame with values beyond [0, 1], negative zero, an
    infinity and a NaN comes back bit for bit through a reader that checks the OpenEXR structure."""
    W, H = 37, 11
    rng = np.random.default_rng(5)
    img = (rng.standard_normal((H, W, 4)) * 3).astype(np.float32)
    img[0, 0] = (np.float32("inf"), np.float32("-0.0"), np.float32("nan"), 1.0)
    src = r'''
#include <lrl/Image.hpp>
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    const int w = std::atoi(argv[1]), h = std::atoi(argv[2]);
    std::vector<float> px(size_t(w) * h * 4);
    FILE* f = std::fopen(argv[3], "rb");
    if (!f || std::fread(px.data(), 4, px.size(), f) != px.size()) return 1;
    std::fclose(f);
    lrl::writeEXR(argv[4], w, h, px.data());
    return 0;
}
'''
    with tempfile.TemporaryDirectory() as td:
        cpp, exe, raw, out = (os.path.join(td, n) for n in ("w.cpp", "w", "in.raw", "out.exr"))
        open(cpp, "w").write(src)
        img.tofile(raw)
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(REPO, "raytracing_amd", "host"), "-o", exe, cpp], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert subprocess.run([exe, str(W), str(H), raw, out]).returncode == 0
        back = read_exr(out)
    assert np.array_equal(back.view(np.uint32), img.view(np.uint32))


@pytest.mark.gpu
def test_headless_example_moves_spheres_and_dumps_ppm_pfm():
    """--move: Movement::run before every frame through lrl_scene_update_primitives + lrl_scene_update_bvh (host tree) or
    + lrl_scene_rebuild_bvh (library's tree), from the compiled C++ caller.  After three frames the spheres have moved by
    sin 0.1 + sin 0.2 + sin 0.3: the frame must equal the tracer's frame of a scene assembled with the spheres there.  The PPM and
    PFM dumps (lrl/Image.hpp) hold the same frame as the BMP: byte for byte, and float values that quantise to those bytes."""
    from raytracing_amd import abi, tracer as T
    import ctypes as C
    W = H = 192
    m = scenes.bvh_default(builder="reference")
    v = m.scene_view()
    sph = np.frombuffer((C.c_uint8 * (v.num.sphereNum * 128)).from_address(v.spheres), dtype=abi.SPHERE)
    for k in (1, 2, 3):                                   # float arithmetic as the example's: t += 0.1f; y += sin(t)
        t = np.float32(0)
        for _ in range(k):
            t = np.float32(t + np.float32(0.1))
        s = np.float32(np.sin(np.float32(t)))
        sph["sphere"][0, 1] = np.float32(sph["sphere"][0, 1] + s)
        sph["sphere"][1, 1] = np.float32(sph["sphere"][1, 1] + s)
    m.data_init("reference")
    tr = T.Tracer(0)
    try:
        tr.scene_upload(m.scene_view())
        tr.set_camera(m.camera(H), W, H)
        tr.set_decay_origin((0, 0, 0, 1))
        want = quantised(tr.render(max_depth=10)[0][..., :3])
    finally:
        tr.close()
    for extra in ([], ["--device-bvh"]):
        with tempfile.TemporaryDirectory() as td:
            out, ppm, pfm, exr = (os.path.join(td, n) for n in ("f.bmp", "f.ppm", "f.pfm", "f.exr"))
            r = subprocess.run([EXE, scenes.BVH_RESOURCES, out, "--size", str(W), str(H), "--frames", "3", "--move", "--ppm", ppm, "--pfm", pfm, "--exr", exr] + extra,
                               capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr + r.stdout
            img, img_ppm, img_pfm, img_exr = read_bmp(out), read_ppm(ppm), read_pfm(pfm), read_exr(exr)
        assert np.array_equal(img, img_ppm)
        assert np.array_equal(img_exr[..., :3].view(np.uint32), img_pfm.view(np.uint32)) and np.all(img_exr[..., 3] == 1.0)
        assert np.array_equal(quantised(img_pfm), img)
        assert np.isfinite(img_pfm).mean() > 0.999 and img_pfm.max() > 1.0       # unclamped values survive in the float dump
        d = np.abs(img.astype(np.int16) - want.astype(np.int16))
        # sinf of the C library against numpy's: the sphere centres may differ in the last bit, a byte here and there with them
        assert (d > 1).any(axis=-1).mean() <= 1e-3, (extra, float((d > 1).any(axis=-1).mean()), int(d.max()))


def test_cmake_builds_libraries_and_example():
    """north_star's "headless Linux CMake build": configure + build of CMakeLists.txt (about 20 s on 8 cores), then the C-ABI library it
    produced exports what include/lrl_trace.h declares."""
    cmake = shutil.which("cmake")
    hipcxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not cmake or not os.path.exists(hipcxx):
        pytest.skip("cmake or the ROCm clang++ is not installed")
    import ctypes
    from test_abi_symbols import declared_symbols
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ, TMPDIR=td)
        r = subprocess.run([cmake, "-S", REPO, "-B", td, f"-DCMAKE_HIP_COMPILER={hipcxx}"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        r = subprocess.run([cmake, "--build", td, "-j", str(min(8, os.cpu_count() or 2))], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        for f in ("liblrltrace.so", "liblrlhost.so", "liboracle.so", "bvh_headless"):
            assert os.path.exists(os.path.join(td, f)), f
        L = ctypes.CDLL(os.path.join(td, "liblrltrace.so"))
        for s in declared_symbols():
            assert hasattr(L, s), s
