#!/usr/bin/env python3
"""One-off parity sweep on the GPU box: seeds [lo, hi) of scenes.write_random_scene (plain and hard), every kernel
organisation against the oracle, bit for bit.  Prints the seeds that differ (none expected).

    python tools/fuzz/parity_sweep.py lo hi [width height frames]
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import logipathtracer_amd as L  # noqa: E402
from logipathtracer_amd import scenes  # noqa: E402
from oracle import binding as O  # noqa: E402  (the checker, as in the tests)


def same(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a, nan=-1.0).view(np.uint32),
                                                                         np.nan_to_num(b, nan=-1.0).view(np.uint32))


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    big = tuple(int(a) for a in sys.argv[3:6]) if len(sys.argv) > 5 else None
    d = tempfile.mkdtemp()
    dev = L.Device(0)
    bad = 0
    for seed in range(lo, hi):
        for hard in (False, True):
            path = os.path.join(d, "r.gltf")
            scenes.write_random_scene(path, seed, hard)
            builder = (L.BUILDER_SPLIT_BVH, L.BUILDER_BINNED_SAH, L.BUILDER_LBVH, L.BUILDER_PLOC)[seed % 4]
            s = L.Scene(path, builder=builder, device=dev if builder in L.GPU_BUILDERS else None)
            osc = O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh)
            w, h = ((96, 64), (61, 47), (128, 40))[seed % 3]
            bounces = (12, 5, 16)[seed % 3]
            bsdf = L.BSDF_BASIC if seed % 6 == 5 else L.BSDF_HEITZ
            frames = 2
            if big:
                w, h, frames = big
            ubos = L.frame_ubos(s.camera_world, s.fov_y, frames, seed=1000 + seed)
            ref, st = O.render_frames(osc, ubos, w, h, bounces, bsdf_mode=bsdf)
            dev.upload_scene(s)
            for pipeline in (L.PIPELINE_WAVEFRONT, L.PIPELINE_MEGAKERNEL, L.PIPELINE_PERSISTENT, L.PIPELINE_HYBRID):
                dev.resize(w, h)
                dev.set_options(bounces, bsdf, pipeline, 0)
                dev.render_frames(ubos)
                img = dev.read_accum()
                if dev.stats().segments != st["segments"] or not same(img, ref):
                    bad += 1
                    print("MISMATCH seed", seed, "hard", hard, "pipeline", pipeline, flush=True)
        if seed % 100 == 0:
            print("seed", seed, "mismatches so far", bad, flush=True)
    print("done", lo, hi, "mismatches", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
