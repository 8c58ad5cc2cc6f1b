#!/usr/bin/env python3
"""Baseline B1 (BASELINE.json configs[0]): the reference's own GLSL on Mesa llvmpipe, default BVH scene exactly as
BVH/BVH.cpp:280-703 builds it, camera r0 = (0,0,30), trans = I, fovy 60, 1024 x 1024, timed in THIS container (the reference cannot
travel to the GPU box), and the 1024^2 golden of the same frame: primary-hit ids + per-tile sums of the image.

  python tools/baseline_llvmpipe.py [--reference /root/reference] [--threads N]

Writes profiles/r02_baseline_llvmpipe_c1.json (what bench.py reports as cpu_baseline.llvmpipe) and
tests/golden/bvh_viewA_1024.npz.  Ray counts come from the CPU oracle's counters on the same frame (the shader has none).
RayTraceDepth 10 is the shader as shipped; RayTraceDepth 1 is the nearest the unmodified text gets to "primary + shadow rays"
(primary ray, its shadow rays, one reflected / refracted segment without further shading): a depth of 0 does not compile
(`Stack stack[RayTraceDepth]`)."""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import oracle_api as O  # noqa: E402
from raytracing_amd import scenes  # noqa: E402

TILE = 16


def tile_sums(img):
    h, w, _ = img.shape
    a = np.nan_to_num(img[..., :3].astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
    return a.reshape(h // TILE, TILE, w // TILE, TILE, 3).sum(axis=(1, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    shaders = os.path.join(a.reference, "BVH", "shaders")
    m = scenes.bvh_default()
    S = a.size
    tr = m.camera(S)
    out = {"config": f"C1: default BVH scene (94 triangles, 5 spheres, 16 discs, 5 cylinders, 5 cones, 3 lights), {S}x{S}, camera as shipped",
           "renderer": "the reference's BVH/shaders on Mesa llvmpipe via oracle/_ref/gl_harness (headless DRI swrast)",
           "host": f"build container, {os.cpu_count()} CPUs", "LP_NUM_THREADS": a.threads, "runs": {}}
    gold = None
    for depth in (10, 1):
        best, rr = None, None
        for _ in range(a.reps):
            r = O.run_reference(m, tr, S, S, shaders, max_depth=depth, threads=a.threads, ids=(depth == 10))
            ms = r["timing"]["render_ms"]
            if best is None or ms < best:
                best, rr = ms, r
        _, _, cn = O.render(m.scene_view(), tr, S, S, max_depth=depth, threads=a.threads, want_hits=False)
        out["runs"][f"RayTraceDepth {depth}"] = {"render_ms": round(best, 2), "rays": cn.rays, "rays_primary": int(cn.raysPrimary),
                                                 "rays_shadow": int(cn.raysShadow), "Mrays_per_s": round(cn.rays / best / 1e3, 3)}
        print(depth, out["runs"][f"RayTraceDepth {depth}"], flush=True)
        if depth == 10:
            gold = rr
    os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
    with open(os.path.join(REPO, "profiles", "r02_baseline_llvmpipe_c1.json"), "w") as f:
        json.dump(out, f, indent=1)
    if S == 1024:
        ids = gold["ids"]
        np.savez_compressed(os.path.join(REPO, "tests", "golden", "bvh_viewA_1024.npz"), hit_type=ids[..., 0].astype(np.uint8),
                            hit_idx=ids[..., 1].astype(np.uint16), tile_sums=tile_sums(gold["img"]), tile=TILE,
                            trans=np.ascontiguousarray(tr).view(np.uint8), width=S, height=S, depth=10, variant=0)
        print("golden bvh_viewA_1024 written", os.path.getsize(os.path.join(REPO, "tests", "golden", "bvh_viewA_1024.npz")), "bytes")


if __name__ == "__main__":
    main()
