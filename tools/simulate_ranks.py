#!/usr/bin/env python3
"""Predicts the tile-sharded scaling on ONE GPU: renders every rank's share of an N-way partition in turn
(64*N frames of 1/N of the pixels, as bench.py does per rank) and reports the per-rank step times and the
aggregate rate N ranks running in parallel would reach (total segments / slowest rank)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import logipathtracer_amd as L
scene = L.Scene('/root/repo/tests/golden/cornell_box.gltf')
dev = L.Device(0)
dev.upload_scene(scene)
for n in (1, 2, 4, 8):
    times = []
    for r in range(n):
        dev.resize(1920, 1080)
        if n > 1:
            dev.set_partition(64, 64, r, n)
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
    print(n, 'ranks: per-rank ms', [round(t, 1) for t, _ in times], 'aggregate Msamples/s if parallel', round(total / tmax / 1e3, 1))
