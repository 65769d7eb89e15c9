"""Renders the open stand-in scene frame by frame on the GPU, lists the pixels whose sample is not finite and
checks what the CPU oracle computes for the same (frame, pixel)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import logipathtracer_amd as L
from logipathtracer_amd import scenes
from oracle import binding as O

W, H, B, FRAMES = 1920, 1080, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 48
path = "/tmp/outside_synth.gltf"
if not os.path.exists(path):
    scenes.write_outside_synth(path, 1.0)
sc = L.Scene(path)
ubos = L.frame_ubos(sc.camera_world, sc.fov_y, FRAMES, seed=0x5EED)
ubos["reset"] = 1  # every frame stores: the image after frame f is frame f's sample
found = []
with L.Device(0) as dev:
    dev.upload_scene(sc)
    dev.resize(W, H)
    dev.set_options(B, L.BSDF_HEITZ, L.PIPELINE_AUTO, 0)
    for f in range(FRAMES):
        dev.render_frames(ubos[f:f + 1])
        img = dev.read_accum()
        bad = np.argwhere(~np.isfinite(img[..., :3]).all(axis=2))
        for y, x in bad:
            found.append((f, int(x), int(y), img[y, x].tolist()))
print("non-finite samples:", len(found), "in", FRAMES, "frames")
osc = O.OracleScene(sc.objects, sc.object_bvh, sc.vertices, sc.mesh_bvh)
out = []
for f, x, y, px in found[:40]:
    ref, _ = O.render_frames(osc, ubos[f:f + 1], W, H, B, tile=(x, y, 1, 1), threads=1)
    out.append({"frame": f, "x": x, "y": y, "gpu": px, "oracle": ref[y, x].tolist(), "seed": ubos["seed"][f].tolist()})
    print(out[-1])
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r3a", "nan_samples.json"), "w"), indent=1)
