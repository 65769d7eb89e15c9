#!/usr/bin/env python3
"""Predicts the tile-sharded scaling on ONE GPU: renders every rank's share of an N-way partition in turn, as
bench.py does per rank, and reports the per-rank step times and the aggregate rate N ranks running in parallel
would reach (total segments / slowest rank).  No collective is involved in the timed region of the real run
either (tiles are independent), so what the prediction leaves out is only what differs between one GPU used N
times and N GPUs used once: nothing in the kernels, the host launching N processes, and clock/thermal spread.

  weak   : spp per step = spp * N   (each rank renders spp * N samples of 1/N of the pixels: constant batch)
  strong : spp per step = spp       (each rank renders spp samples of 1/N of the pixels: the batch shrinks)

    python tools/simulate_ranks.py [--out file.json] [--tile 64] [--configs c2,c4,c5]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import logipathtracer_amd as L  # noqa: E402
from logipathtracer_amd import scenes  # noqa: E402

CONFIGS = {
    # name: scene, width, height, bounces, spp per step at N=1 (the single-GPU batch of bench.py / of one step of the config)
    "c2": ("cornell", 1920, 1080, 8, 64),
    "c4": ("cornell", 3840, 2160, 8, 32),
    "c5": ("outside", 3840, 2160, 16, 32),
}


def run(dev, scene, w, h, bounces, tile, n, spp_step, reps=2):
    times = []
    for r in range(n):
        dev.resize(w, h)
        if n > 1:
            dev.set_partition(tile, tile, r, n)
        dev.set_options(bounces, L.BSDF_HEITZ, L.PIPELINE_AUTO, 0)
        ubos = L.frame_ubos(scene.camera_world, scene.fov_y, spp_step * (reps + 1))
        dev.render_frames(ubos[:spp_step])
        dev.synchronize()
        t0 = time.perf_counter()
        seg = 0
        pipe = None
        for i in range(1, reps + 1):
            dev.render_frames(ubos[i * spp_step:(i + 1) * spp_step])
            st = dev.stats()
            seg += st.segments
            pipe = st.pipeline
        dt = (time.perf_counter() - t0) / reps
        times.append((dt * 1e3, seg / reps, pipe))
    tmax = max(t for t, _, _ in times)
    total = sum(s for _, s, _ in times)
    return {"ranks": n, "spp_per_step": spp_step, "per_rank_ms": [round(t, 2) for t, _, _ in times],
            "pipeline": ["wavefront", "megakernel", "persistent", "auto", "hybrid"][times[0][2]],
            "slowest_ms": round(tmax, 2), "imbalance": round(tmax / min(t for t, _, _ in times), 4),
            "aggregate_msamples_per_s": round(total / tmax / 1e3, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--configs", default="c2,c4,c5")
    args = ap.parse_args()
    dev = L.Device(0)
    loaded = {}
    out = {"what": "one MI355X rendering every rank's share in turn (tools/simulate_ranks.py); aggregate = total segments / "
                   "slowest rank; speedup = aggregate at N / aggregate at 1", "tile": args.tile, "configs": {}}
    for name in args.configs.split(","):
        kind, w, h, bounces, spp = CONFIGS[name]
        if kind not in loaded:
            if kind == "cornell":
                loaded[kind] = L.Scene(os.path.join(ROOT, "tests", "golden", "cornell_box.gltf"))
            else:
                path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "outside_synth_sim.gltf")
                scenes.write_outside_synth(path, 1.0)
                loaded[kind] = L.Scene(path)
        scene = loaded[kind]
        dev.upload_scene(scene)
        rows = {"weak": [], "strong": []}
        base = run(dev, scene, w, h, bounces, args.tile, 1, spp)
        for mode in ("weak", "strong"):
            rows[mode].append(dict(base, speedup=1.0))
            for n in (2, 4, 8):
                r = run(dev, scene, w, h, bounces, args.tile, n, spp * n if mode == "weak" else spp)
                r["speedup"] = round(r["aggregate_msamples_per_s"] / base["aggregate_msamples_per_s"], 3)
                rows[mode].append(r)
                print(name, mode, r, flush=True)
        out["configs"][name] = {"scene": kind, "width": w, "height": h, "bounces": bounces, "spp_per_step_at_1": spp, **rows}
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)
    dev.close()


if __name__ == "__main__":
    main()
