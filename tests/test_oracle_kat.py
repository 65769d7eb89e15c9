"""Pins the CPU oracle: integer-exact RNG vectors, closed-form primitives, transcendental accuracy,
the parity-critical quirks of shaders/path_tracing.comp, and the committed config-1 image."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import binding as O

F32 = np.float32


def test_rng_known_answers():
    for case in json.load(open(os.path.join(GOLDEN, "rng_kat.json"))):
        s = O.Seed(*case["seed"])
        for n, val in zip(case["n"], case["value"]):
            r = O.lib.lpto_rand(C.byref(s))
            assert F32(r) == F32(val), (case["seed"], n)
        assert [s.x, s.y] == case["end_seed"]


def test_rand_can_return_exactly_one():
    # random.glsl:14: float(n) * (1/float(0xffffffff)); n >= 0xFFFFFF80 converts to 2^32
    assert F32(np.uint32(0xFFFFFF80)) * F32(2.0 ** -32) == F32(1.0)
    assert F32(np.uint32(0xFFFFFF7F)) * F32(2.0 ** -32) < F32(1.0)


def test_primitive_known_answers():
    k = json.load(open(os.path.join(GOLDEN, "primitive_kat.json")))
    for c in k["ray_triangle"]:
        t = O.lib.lpto_ray_triangle(O.v3(c["o"]), O.v3(c["d"]), *[O.v3(v) for v in c["tri"]])
        assert F32(t) == F32(c["t"]), c
    for c in k["ray_aabb"]:
        h = O.lib.lpto_ray_aabb_test(O.v3(c["o"]), O.v3(c["d"]), O.v3(c["box"][0]), O.v3(c["box"][1]), c["dist"])
        assert h == c["hit"], c
    for c in k["barycentric"]:
        out = (C.c_float * 3)()
        O.lib.lpto_barycentric(O.v3(c["p"]), *[O.v3(v) for v in c["tri"]], out)
        assert np.allclose(list(out), c["uvw"], atol=1e-6), c
    for c in k["srgb_to_linear"]:
        assert abs(O.lib.lpto_srgb_to_linear(c["x"]) - c["y"]) <= 4e-6 * max(c["y"], 1e-3), c
    for c in k["fresnel"]:
        assert abs(O.lib.lpto_fresnel(c["vdoth"], c["eta"]) - c["F"]) <= 1e-6, c
    for c in k["refract_eta"]:
        out = (C.c_float * 3)()
        O.lib.lpto_refract_eta(O.v3(c["wi"]), O.v3(c["wm"]), c["eta"], out)
        assert np.allclose(list(out), c["out"], atol=1e-6), c


def test_aabb_with_axis_parallel_ray_and_nan_lanes():
    # A zero direction component gives infinite reciprocals.  Strictly between the slabs the lane is
    # (-inf, +inf) and drops out; exactly on a slab plane 0 * inf = NaN, which minNum/maxNum replace by
    # the other operand (+inf here), so a ray lying in a face plane misses the box.
    box = (O.v3([-1, -1, -1]), O.v3([1, 1, 1]))
    assert O.lib.lpto_ray_aabb_test(O.v3([-0.5, 0, 5]), O.v3([0, 0, -1]), *box, 3.4e38) == 1
    assert O.lib.lpto_ray_aabb_test(O.v3([-1.0, 0, 5]), O.v3([0, 0, -1]), *box, 3.4e38) == 0
    assert O.lib.lpto_ray_aabb_test(O.v3([-2.0, 0, 5]), O.v3([0, 0, -1]), *box, 3.4e38) == 0


def test_transcendentals_track_libm():
    rng = np.random.default_rng(1)
    x = rng.uniform(-7.0, 7.0, 4000).astype(F32)
    s = np.array([O.lib.lpto_sin(float(v)) for v in x])
    c = np.array([O.lib.lpto_cos(float(v)) for v in x])
    assert np.max(np.abs(s - np.sin(x.astype(np.float64)))) < 2.5e-7
    assert np.max(np.abs(c - np.cos(x.astype(np.float64)))) < 2.5e-7
    p = np.concatenate([rng.uniform(1e-6, 4.0, 3000), 2.0 ** rng.uniform(-100, 100, 1000)]).astype(F32)
    lg = np.array([O.lib.lpto_log(float(v)) for v in p])
    ref = np.log(p.astype(np.float64))
    assert np.max(np.abs(lg - ref) / np.maximum(np.abs(ref), 1e-3)) < 4e-7
    e = rng.uniform(-20, 20, 3000).astype(F32)
    ex = np.array([O.lib.lpto_exp(float(v)) for v in e])
    assert np.max(np.abs(ex / np.exp(e.astype(np.float64)) - 1)) < 3e-7
    b = rng.uniform(0.01, 1.0, 3000).astype(F32)
    pw = np.array([O.lib.lpto_pow(float(v), 2.4) for v in b])
    assert np.max(np.abs(pw / np.power(b.astype(np.float64), 2.4) - 1)) < 3e-6
    assert O.lib.lpto_log(0.0) == -np.inf and np.isnan(O.lib.lpto_log(-1.0))
    assert O.lib.lpto_exp(-1000.0) == 0.0 and O.lib.lpto_exp(1000.0) == np.inf
    assert abs(O.lib.lpto_tan(0.2) - np.tan(0.2)) < 1e-7


def test_interaction_type_thresholds():
    # heitz/interaction_type.glsl:10-29: metallic=1 always conductor; transmission=1 (metallic 0) always dielectric
    for seed in range(20):
        s = O.Seed(seed, 3 * seed + 1)
        assert O.lib.lpto_interaction_type(1.0, 0.0, C.byref(s)) == O.K_METALLIC
        assert O.lib.lpto_interaction_type(0.0, 1.0, C.byref(s)) == O.K_TRANS
        assert O.lib.lpto_interaction_type(0.0, 0.0, C.byref(s)) == O.K_DIFF
        assert (s.x, s.y) == (seed + 3, 3 * seed + 4)  # one rand() per call


def test_bsdf_random_number_budget_and_returns():
    # dielectric returns F0 whatever the walk did (heitz/BSDF.glsl:208); the diffuse walk multiplies F0 per order
    F0 = O.v3([0.5, 0.25, 0.125])
    view = O.v3([0.3, 0.1, 0.9486833])
    for seed in range(50):
        s = O.Seed(seed * 7919, seed * 104729 + 1)
        ld, w = (C.c_float * 3)(), (C.c_float * 3)()
        O.lib.lpto_bsdf(O.BSDF_HEITZ, O.K_TRANS, F0, view, 0.2, 1.0, 1.5, 1, C.byref(s), ld, w)
        assert list(w) == [0.5, 0.25, 0.125]
        assert abs(np.linalg.norm(list(ld)) - 1) < 1e-4
        s = O.Seed(seed * 7919, seed * 104729 + 1)
        O.lib.lpto_bsdf(O.BSDF_HEITZ, O.K_DIFF, F0, view, 0.5, 0.0, 1.5, 1, C.byref(s), ld, w)
        used = s.x - seed * 7919
        # 1 height sample to leave + 5 per scattering order (1 height + 2 VNDF + 2 disk)
        assert used % 5 == 1 or list(w) == [0.0, 0.0, 0.0]
        order = (used - 1) // 5
        if list(w) != [0.0, 0.0, 0.0]:
            assert np.allclose(list(w), [0.5 ** order, 0.25 ** order, 0.125 ** order], rtol=1e-6)
        assert ld[2] > 0 or list(w) == [0.0, 0.0, 0.0] or order >= 0
    # basic diffuse carries the extra cosine factor of basic/BSDF.glsl:10
    s = O.Seed(5, 9)
    O.lib.lpto_bsdf(O.BSDF_BASIC, O.K_DIFF, F0, view, 0.5, 0.0, 1.5, 1, C.byref(s), ld, w)
    assert np.allclose(list(w), np.array([0.5, 0.25, 0.125]) * ld[2], rtol=1e-6)
    # basic mirror: reflect(-v, z)
    O.lib.lpto_bsdf(O.BSDF_BASIC, O.K_METALLIC, F0, view, 0.5, 0.0, 1.5, 1, C.byref(s), ld, w)
    assert np.allclose(list(ld), [-view[0], -view[1], view[2]])


def test_generate_ray_has_no_half_pixel_offset_and_no_y_flip(cornell):
    import logipathtracer_amd as L
    ubo = L.frame_ubos(cornell.camera_world, cornell.fov_y, 1)
    res = 64
    o, d = (C.c_float * 3)(), (C.c_float * 3)()
    # path_tracing.comp:117: uv = 2*pix/res - 1 + jitter, |jitter| <= 2/res
    for px, py in [(0, 0), (res // 2, res // 2), (res - 1, 5)]:
        s = O.Seed(int(ubo["seed"][0, 0]) * px & 0xFFFFFFFF, int(ubo["seed"][0, 1]) * py & 0xFFFFFFFF)
        O.lib.lpto_generate_ray(ubo.ctypes.data, px, py, res, res, C.byref(s), o, d)
        assert np.allclose(list(o), cornell.camera_world[12:15])
        m = cornell.camera_world.reshape(4, 4)  # rows are columns of the matrix
        dirv = np.array(list(d))
        th = np.tan(cornell.fov_y / 2)
        fwd = -m[2, :3]
        x = dirv @ m[0, :3] / (dirv @ fwd) / th
        y = dirv @ m[1, :3] / (dirv @ fwd) / th
        assert abs(x - (2 * px / res - 1)) <= 2 / res + 1e-5
        assert abs(y - (2 * py / res - 1)) <= 2 / res + 1e-5
        assert abs(np.linalg.norm(dirv) - 1) < 1e-6


def _golden_scene():
    z = np.load(os.path.join(GOLDEN, "cornell_scene_binned_sah.npz"))
    return z, O.OracleScene(z["objects"], z["object_bvh"], z["vertices"], z["mesh_bvh"])


def test_config1_image_and_counters_are_pinned():
    import logipathtracer_amd as L
    z, osc = _golden_scene()
    ubos = L.frame_ubos(z["camera_world"], float(z["fov_y"]), 1, seed=0x5EED)
    meta = json.load(open(os.path.join(GOLDEN, "cornell_c1_counters.json")))
    assert [int(ubos["seed"][0, 0]), int(ubos["seed"][0, 1])] == meta["seed"]
    img, st = O.render_frames(osc, ubos, 256, 256, 2)
    ref = np.load(os.path.join(GOLDEN, "cornell_c1_256x256_1spp_2b.npy"))
    assert np.array_equal(img, ref)
    assert st == meta["counters"]
    # single-threaded and tiled renders give the same pixels
    img1, _ = O.render_frames(osc, ubos, 256, 256, 2, threads=1)
    assert np.array_equal(img1, ref)
    tiled = np.zeros_like(ref)
    for tile in [(0, 0, 128, 256), (128, 0, 128, 100), (128, 100, 128, 156)]:
        O.render_frames(osc, ubos, 256, 256, 2, tile=tile, accum=tiled)
    assert np.array_equal(tiled, ref)


def test_miss_assigns_background_and_drops_gathered_emission():
    # path_tracing.comp:220-223: a miss sets acc = mask*0.2.  Camera far outside looking away: every pixel misses.
    import logipathtracer_amd as L
    z, osc = _golden_scene()
    cam = z["camera_world"].copy()
    cam[12:15] = [60.0, 1.0, 0.0]
    cam[8:11] = -cam[8:11]  # look the other way
    ubos = L.frame_ubos(cam, float(z["fov_y"]), 2, seed=3)
    img, st = O.render_frames(osc, ubos, 32, 32, 8)
    assert st["hits"] == 0 and st["segments"] == st["paths"]
    assert np.all(img[..., :3] == F32(0.2) + F32(0.2)) and np.all(img[..., 3] == 2.0)  # two frames accumulated


def test_reset_stores_instead_of_adding():
    import logipathtracer_amd as L
    z, osc = _golden_scene()
    ubos = L.frame_ubos(z["camera_world"], float(z["fov_y"]), 3, seed=11)
    a, _ = O.render_frames(osc, ubos, 48, 48, 3)
    ubos2 = ubos.copy()
    ubos2["reset"][2] = 1
    b, _ = O.render_frames(osc, ubos2, 48, 48, 3)
    only_last, _ = O.render_frames(osc, ubos2[2:], 48, 48, 3)
    assert np.all(a[..., 3] == 3.0) and np.all(b[..., 3] == 1.0)
    assert np.array_equal(b, only_last)


def test_resolve_formula():
    # tex_to_quad.frag:24-33: 1 - exp(-1.5 * hdr), gamma 1/2.2, display row 0 = accumulation row H-1
    acc = np.zeros((2, 3, 4), dtype=F32)
    acc[0, :, :3] = 4.0
    acc[1, :, :3] = 1.0
    acc[..., 3] = 4
    out = O.resolve(acc, 0.25)
    exp0 = (1 - np.exp(-1.5 * 0.25)) ** (1 / 2.2)
    exp1 = (1 - np.exp(-1.5 * 1.0)) ** (1 / 2.2)
    assert np.allclose(out[0, :, :3], exp0, rtol=2e-6) and np.allclose(out[1, :, :3], exp1, rtol=2e-6)
    assert np.all(out[..., 3] == 1.0)
