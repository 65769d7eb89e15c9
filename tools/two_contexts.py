#!/usr/bin/env python3
"""Experiment: two contexts on one GPU, each rendering half of the frames of a batch from its own host
thread and stream, against one context rendering all of them.  Shows how far kernels bound by different
units (closest-hit search: L1 gathers; shading: vector ALU) overlap when they share the chip.
usage: two_contexts.py [frames] [stagger_ms]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import logipathtracer_amd as L  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    stagger = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    scene = L.Scene(os.path.join(ROOT, "tests", "golden", "cornell_box.gltf"))
    ubos = L.frame_ubos(scene.camera_world, scene.fov_y, frames)
    devs = []
    for _ in range(2):
        d = L.Device(0)
        d.upload_scene(scene)
        d.resize(1920, 1080)
        d.set_options(8, L.BSDF_HEITZ, L.PIPELINE_WAVEFRONT, 0)
        devs.append(d)
    half = frames // 2

    def run(d, u, delay=0.0):
        if delay:
            time.sleep(delay)
        d.render_frames(u)
        d.synchronize()

    for rep in range(3):
        t0 = time.perf_counter()
        run(devs[0], ubos)
        one = time.perf_counter() - t0
        seg_one = devs[0].stats().segments
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(devs[0], ubos[:half])),
              threading.Thread(target=run, args=(devs[1], ubos[half:], stagger * 1e-3))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        two = time.perf_counter() - t0
        seg_two = devs[0].stats().segments + devs[1].stats().segments
        print(f"rep {rep}: one context {one * 1e3:.2f} ms ({seg_one / one / 1e6:.0f} Msamples/s), "
              f"two contexts {two * 1e3:.2f} ms ({seg_two / two / 1e6:.0f} Msamples/s)", flush=True)


if __name__ == "__main__":
    main()
