"""CPU tests of the oracle's restatement of LRL_FLAG_DIFFUSE_BOUNCE (include/lrl_trace.h).  The mode has no
counterpart in the reference, so there is no llvmpipe golden for it: the random generator is pinned by Random123's published
known-answer vector for Philox4x32-10, the sampler by its distribution, and the device by the oracle (tests/test_gpu_parity.py)."""
import numpy as np

from oracle import oracle_api as O
from raytracing_amd import abi, scenes


def test_philox4x32_10_known_answer():
    # Random123 kat_vectors: philox4x32 10 rounds, counter 0, key 0
    assert O.philox4x32_10((0, 0, 0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert O.philox4x32_10((1, 0, 0, 0)) != O.philox4x32_10((0, 1, 0, 0)) != O.philox4x32_10((0, 0, 1, 0))


def test_diffuse_direction_is_cosine_weighted_about_the_incoming_side():
    for normal, cosi1 in (((0.0, 0.0, 1.0), -0.3), ((0.6, 0.0, -0.8), 0.4), ((0.0, -1.0, 0.0), -1.0)):
        n = np.array(normal, dtype=np.float64) * (-1.0 if cosi1 > 0 else 1.0)      # the side the ray came from
        d = np.array([O.diffuse_direction(normal, cosi1, 12345 + i, i % 7, i % 5) for i in range(20000)], dtype=np.float64)
        assert np.abs(np.linalg.norm(d, axis=1) - 1.0).max() < 1e-6
        c = d @ n
        assert c.min() >= 0.0                                   # never below the surface
        assert abs(c.mean() - 2.0 / 3.0) < 0.01                 # pdf cos/pi: E[cos] = 2/3, E[cos^2] = 1/2
        assert abs((c * c).mean() - 0.5) < 0.01
        t = d - np.outer(c, n)                                  # tangential part: no preferred azimuth
        assert np.abs(t.mean(axis=0)).max() < 0.01
    a = O.diffuse_direction((0.0, 0.0, 1.0), -1.0, 5, 6, 7)
    assert np.array_equal(a, O.diffuse_direction((0.0, 0.0, 1.0), -1.0, 5, 6, 7))          # a pure function of its key
    assert not np.array_equal(a, O.diffuse_direction((0.0, 0.0, 1.0), -1.0, 5, 6, 8))


def test_diffuse_render_differs_only_after_the_first_hit():
    m = scenes.bvh_default()
    W = H = 48
    tr = m.camera(H)
    view = m.scene_view()
    mirror, hits_m, cm = O.render(view, tr, W, H, max_depth=3, threads=4)
    diffuse, hits_d, cd = O.render(view, tr, W, H, max_depth=3, threads=4, diffuse=True)
    assert np.array_equal(hits_m, hits_d) and cm.raysPrimary == cd.raysPrimary
    assert np.isfinite(diffuse).all() and not np.array_equal(mirror, diffuse)
    # stripes compose: the random stream is keyed by the pixel of the full frame
    stripe, _, _ = O.render(view, tr, W, H, max_depth=3, rows=(16, 32), threads=2, diffuse=True)
    assert np.array_equal(stripe, diffuse[16:32])
