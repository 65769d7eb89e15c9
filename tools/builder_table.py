#!/usr/bin/env python3
"""Mesh-BVH builder comparison: build time, tree size, traversal work per segment and traced throughput for
the host builders (SAH with spatial splits, binned SAH) and the GPU LBVH on the two benchmark scenes.
Run on the GPU box; writes JSON lines (one per scene and builder) to stdout."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import logipathtracer_amd as L  # noqa: E402
from logipathtracer_amd import scenes  # noqa: E402

NAMES = ["split_bvh", "binned_sah", "lbvh_gpu", "ploc_gpu"]


def main():
    tmp = os.environ.get("TMPDIR", "/tmp")
    outside = os.path.join(tmp, "builder_table_outside.gltf")
    scenes.write_outside_synth(outside, 1.0)
    cornell = os.path.join(ROOT, "tests", "golden", "cornell_box.gltf")
    dev = L.Device(0)
    for label, path, bench_scene, spp in (("cornell_box", cornell, cornell, 64), ("outside_synth", outside, "outside_synth", 16)):
        for b in range(4):
            L.Scene(path, builder=b, device=dev if b in L.GPU_BUILDERS else None)  # warm (first GPU build compiles nothing, but pages in)
            t0 = time.perf_counter()
            s = L.Scene(path, builder=b, device=dev if b in L.GPU_BUILDERS else None)
            build_s = time.perf_counter() - t0
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scene", bench_scene, "--builder", str(b),
                                "--steps", "3", "--warmup", "1", "--spp-per-step", str(spp), "--no-cpu-baseline", "--no-secondary"],
                               capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                print(json.dumps({"scene": label, "builder": NAMES[b], "error": r.stderr[-400:]}), flush=True)
                continue
            out = json.loads(line[-1])
            rf = out.get("roofline") or {}
            print(json.dumps({
                "scene": label, "builder": NAMES[b], "load_and_build_s": build_s, "triangle_refs": len(s.vertices) // 3,
                "mesh_nodes": len(s.mesh_bvh), "msamples_per_s": out["value"], "ms_per_step": out["ms_per_step"],
                "spp_per_step": spp, "pipeline": out["config"]["pipeline"],
                "inner_nodes_per_segment": rf.get("inner_nodes_per_segment"), "tris_per_segment": rf.get("tris_per_segment"),
                "kernel": rf.get("kernel"), "kernel_avg_launch_ms": rf.get("avg_launch_ms")}), flush=True)
    dev.close()


if __name__ == "__main__":
    main()
