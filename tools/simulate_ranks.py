#!/usr/bin/env python3
"""Predicts the tile-sharded scaling on ONE GPU: renders every rank's share of an N-way partition in turn
(64*N frames of 1/N of the pixels, as bench.py does per rank) and reports the per-rank step times and the
aggregate rate N ranks running in parallel would reach (total segments / slowest rank).

    python tools/simulate_ranks.py [tile ...]      (default tile sizes: 64)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import logipathtracer_amd as L  # noqa: E402

scene = L.Scene(os.path.join(ROOT, 'tests', 'golden', 'cornell_box.gltf'))
dev = L.Device(0)
dev.upload_scene(scene)
tiles = [int(a) for a in sys.argv[1:]] or [64]
for tile in tiles:
    for n in (1, 2, 4, 8):
        if n == 1 and tile != tiles[0]:
            continue
        times = []
        for r in range(n):
            dev.resize(1920, 1080)
            if n > 1:
                dev.set_partition(tile, tile, r, n)
            dev.set_options(8, L.BSDF_HEITZ, L.PIPELINE_AUTO, 0)
            ubos = L.frame_ubos(scene.camera_world, scene.fov_y, 64 * n * 3)
            dev.render_frames(ubos[:64 * n])
            dev.synchronize()
            t0 = time.perf_counter()
            seg = 0
            for i in (1, 2):
                dev.render_frames(ubos[i * 64 * n:(i + 1) * 64 * n])
                seg += dev.stats().segments
            dt = (time.perf_counter() - t0) / 2
            times.append((dt * 1e3, seg / 2))
        tmax = max(t for t, _ in times)
        total = sum(s for _, s in times)
        print('tile', tile, n, 'ranks: per-rank ms', [round(t, 1) for t, _ in times], 'aggregate Msamples/s if parallel',
              round(total / tmax / 1e3, 1), flush=True)
