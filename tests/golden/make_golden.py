#!/usr/bin/env python3
"""Generates the committed fixtures of tests/golden/.

* rng_kat.json, primitive_kat.json: known-answer vectors derived here in pure Python from the
  formulas of the reference shaders (shaders/common/random.glsl:9-15, ray.glsl, util.glsl,
  heitz/BSDF.glsl) — independent of the oracle's C code, using exact integer arithmetic for the RNG
  and closed-form cases for the primitives.
* cornell_c1_*.npy / .json: the oracle's own 256x256, 1 spp, 2-bounce Cornell image and work counters
  (config 1 of the baseline), kept as a regression pin for the oracle and as a second reference for
  the GPU path.  The reference renderer itself cannot run here, so these are NOT outputs of the
  reference binary.
* cornell_box.gltf / cornell_box.bin are the reference's own scene asset (resources/), copied as data.

Run from the repository root after `make`:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

M32 = 0xFFFFFFFF


def glsl_rand(seed):
    """shaders/common/random.glsl:9-15 with Python integers; returns (n, new_seed)."""
    sx, sy = (seed[0] + 1) & M32, (seed[1] + 1) & M32
    qx = (1103515245 * ((sx >> 1) ^ sy)) & M32
    qy = (1103515245 * ((sy >> 1) ^ sx)) & M32
    n = (1103515245 * (qx ^ (qy >> 3))) & M32
    return n, (sx, sy)


def rng_kat():
    cases = []
    for seed in [(0, 0), (1, 1), (0x5EED, 0xC0FFEE), (M32, M32), (123456789, 987654321), (0, 77), (1804289383 * 5 & M32, 846930886 * 9 & M32)]:
        s = seed
        ns = []
        for _ in range(8):
            n, s = glsl_rand(s)
            ns.append(n)
        # float(n) * 2^-32 with float32 rounding of the conversion
        fl = [float(np.float32(n) * np.float32(2.0 ** -32)) for n in ns]
        cases.append({"seed": list(seed), "n": ns, "value": fl, "end_seed": list(s)})
    return cases


def primitive_kat():
    k = {}
    # Moeller-Trumbore: unit right triangle in z=0, rays straight down from z=+h
    tri = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    k["ray_triangle"] = [
        {"o": [0.25, 0.25, 2.0], "d": [0, 0, -1], "tri": tri, "t": 2.0},
        {"o": [0.25, 0.25, -3.0], "d": [0, 0, 1], "tri": tri, "t": 3.0},       # no back-face culling
        {"o": [0.75, 0.75, 2.0], "d": [0, 0, -1], "tri": tri, "t": 3.4e38},    # u+v > 1
        {"o": [-0.1, 0.2, 2.0], "d": [0, 0, -1], "tri": tri, "t": 3.4e38},     # u < 0
        {"o": [0.25, 0.25, 2.0], "d": [0, 0, 1], "tri": tri, "t": -2.0},       # behind: negative t is returned
        {"o": [0.5, 0.5, 1.0], "d": [0, 0, -1], "tri": tri, "t": 1.0},         # on the hypotenuse: u+v == 1 accepted
    ]
    box = [[-1, -1, -1], [1, 1, 1]]
    k["ray_aabb"] = [
        {"o": [0, 0, 5], "d": [0, 0, -1], "box": box, "dist": 3.4e38, "hit": 1},
        {"o": [0, 0, 5], "d": [0, 0, -1], "box": box, "dist": 4.0, "hit": 0},   # entry at t=4 is not < 4
        {"o": [0, 0, 5], "d": [0, 0, -1], "box": box, "dist": 4.5, "hit": 1},
        {"o": [0, 0, 5], "d": [0, 0, 1], "box": box, "dist": 3.4e38, "hit": 0},  # behind the ray
        {"o": [0, 0, 0], "d": [0, 0, 1], "box": box, "dist": 0.001, "hit": 1},   # inside: distance not consulted
        {"o": [3, 0, 5], "d": [0, 0, -1], "box": box, "dist": 3.4e38, "hit": 0},
    ]
    k["barycentric"] = [
        {"p": [0.25, 0.5, 0], "tri": tri, "uvw": [0.25, 0.25, 0.5]},
        {"p": [0, 0, 0], "tri": tri, "uvw": [1, 0, 0]},
        {"p": [1, 0, 0], "tri": tri, "uvw": [0, 1, 0]},
    ]
    k["srgb_to_linear"] = [
        {"x": 0.0, "y": 0.0}, {"x": 0.04045, "y": 0.04045 / 12.92}, {"x": 1.0, "y": 1.0},
        {"x": 0.5, "y": ((0.5 + 0.055) / 1.055) ** 2.4}, {"x": 0.8, "y": ((0.8 + 0.055) / 1.055) ** 2.4},
    ]
    # dielectric Fresnel: normal incidence ((1-eta)/(1+eta))^2, total internal reflection -> 1
    k["fresnel"] = [
        {"vdoth": 1.0, "eta": 1.5, "F": ((1 - 1.5) / (1 + 1.5)) ** 2},
        {"vdoth": 1.0, "eta": 1.0 / 1.5, "F": ((1 - 1.5) / (1 + 1.5)) ** 2},
        {"vdoth": 0.2, "eta": 1.0 / 1.5, "F": 1.0},
        {"vdoth": 1.0, "eta": 1.0, "F": 0.0},
    ]
    # refractEta obeys Snell's law: sin_t = sin_i / eta, result on the far side of the micro-normal
    th = np.deg2rad(30.0)
    k["refract_eta"] = [{"wi": [float(np.sin(th)), 0.0, float(np.cos(th))], "wm": [0, 0, 1], "eta": 1.5,
                         "out": [float(-np.sin(th) / 1.5), 0.0, float(-np.sqrt(1 - (np.sin(th) / 1.5) ** 2))]}]
    return k


def main():
    json.dump(rng_kat(), open(os.path.join(HERE, "rng_kat.json"), "w"), indent=1)
    json.dump(primitive_kat(), open(os.path.join(HERE, "primitive_kat.json"), "w"), indent=1)

    import logipathtracer_amd as L
    from oracle import binding as O
    scene = L.Scene(os.path.join(HERE, "cornell_box.gltf"), builder=L.BUILDER_BINNED_SAH)
    # the converted buffers themselves are part of the fixture: the pinned image is only meaningful
    # for exactly these node arrays (tie-breaks between coplanar surfaces depend on visit order)
    np.savez_compressed(os.path.join(HERE, "cornell_scene_binned_sah.npz"), objects=scene.objects,
                        object_bvh=scene.object_bvh, vertices=scene.vertices, mesh_bvh=scene.mesh_bvh,
                        camera_world=scene.camera_world, fov_y=np.float32(scene.fov_y))
    osc = O.OracleScene(scene.objects, scene.object_bvh, scene.vertices, scene.mesh_bvh)
    ubos = L.frame_ubos(scene.camera_world, scene.fov_y, 1, seed=0x5EED)
    img, st = O.render_frames(osc, ubos, 256, 256, 2)
    np.save(os.path.join(HERE, "cornell_c1_256x256_1spp_2b.npy"), img.astype(np.float16 if False else np.float32))
    json.dump({"config": "cornell_box.gltf 256x256, 1 spp, 2 bounces, Heitz, srand(0x5EED), binned-SAH buffers",
               "seed": [int(ubos["seed"][0, 0]), int(ubos["seed"][0, 1])], "counters": st},
              open(os.path.join(HERE, "cornell_c1_counters.json"), "w"), indent=1)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
