"""Procedural scenes: the stand-in for the reference's outside.gltf (binary buffer missing upstream) and
the Cornell-shaped scene.  CPU: generation is deterministic and converts cleanly.  GPU: parity with the
oracle on a deep, open scene with all three mesh builders."""
import hashlib
import os

import numpy as np
import pytest

import logipathtracer_amd as L
from conftest import rel_l2
from logipathtracer_amd import scenes
from oracle import binding as O


@pytest.fixture(scope="module")
def small_outside(tmp_path_factory):
    d = tmp_path_factory.mktemp("scenes")
    path = str(d / "outside_small.gltf")
    n = scenes.write_outside_synth(path, 0.02)
    return path, n


def test_outside_synth_is_deterministic_and_converts(small_outside, tmp_path):
    path, n = small_outside
    again = str(tmp_path / "outside_small.gltf")
    assert scenes.write_outside_synth(again, 0.02) == n
    h = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
    assert h(path.replace(".gltf", ".bin")) == h(again.replace(".gltf", ".bin"))
    s = L.Scene(path, builder=L.BUILDER_BINNED_SAH)
    assert len(s.vertices) == 3 * n and s.n_cameras == 2
    assert 60 <= len(s.objects) <= 80  # the reference scene has 70 primitives
    assert abs(s.fov_y - 0.7696) < 1e-6
    assert (s.objects["transmissionFactor"] > 0.9).sum() == 1 and (s.objects["emissionFactor"].sum(axis=1) > 0).sum() == 2
    assert sorted(set(np.round(s.objects["metallicFactor"].astype(np.float64), 3).tolist())) == [0.0, 0.5, 0.7]
    # the oracle renders it without overflowing its stack; most rays hit (closed sky shell)
    osc = O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh)
    img, st = O.render_frames(osc, L.frame_ubos(s.camera_world, s.fov_y, 1), 64, 36, 4)
    assert st["hits"] / st["segments"] > 0.9 and np.isfinite(img).all()


def test_full_size_outside_synth_matches_the_reference_scene_shape(tmp_path):
    n = scenes.write_outside_synth(str(tmp_path / "outside_synth.gltf"), 1.0)
    assert 380_000 <= n <= 420_000  # resources/outside.gltf describes 399 949 triangles


@pytest.mark.gpu
@pytest.mark.parametrize("builder", [L.BUILDER_SPLIT_BVH, L.BUILDER_BINNED_SAH, L.BUILDER_LBVH, L.BUILDER_PLOC])
def test_outside_synth_parity(device, small_outside, builder):
    path, _ = small_outside
    s = L.Scene(path, builder=builder, device=device)
    osc = O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh)
    w, h, bounces = 240, 136, 8
    ubos = L.frame_ubos(s.camera_world, s.fov_y, 2, seed=77)
    ref, st = O.render_frames(osc, ubos, w, h, bounces)
    device.upload_scene(s)
    for pipeline in (L.PIPELINE_WAVEFRONT, L.PIPELINE_MEGAKERNEL):
        device.resize(w, h)
        device.set_options(bounces, L.BSDF_HEITZ, pipeline, 0)
        device.render_frames(ubos)
        img = device.read_accum()
        assert device.stats().segments == st["segments"]
        assert rel_l2(img, ref) <= 1e-4
        assert np.array_equal(img, ref)


@pytest.mark.gpu
def test_cornell_like_parity_16_bounces(device, tmp_path):
    path = str(tmp_path / "cornell_like.gltf")
    scenes.write_cornell_like(path, 3)
    s = L.Scene(path)
    osc = O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh)
    ubos = L.frame_ubos(s.camera_world, s.fov_y, 2, seed=3)
    ref, st = O.render_frames(osc, ubos, 200, 200, 16)
    device.upload_scene(s)
    device.resize(200, 200)
    device.set_options(16, L.BSDF_HEITZ, L.PIPELINE_WAVEFRONT, 0)
    device.render_frames(ubos)
    assert device.stats().segments == st["segments"]
    assert np.array_equal(device.read_accum(), ref)


def _tiny_scene(path, with_degenerate):
    """One emissive triangle seen by the camera (object tree and mesh tree are single leaves); optionally a
    zero-area triangle and a triangle seen edge-on, whose Moeller-Trumbore determinants are 0 (1/0 = inf,
    0*inf = NaN must fall out through the comparisons as in shaders/common/ray.glsl:63-82)."""
    g = scenes._GltfBuilder()
    m = g.material("M", (0.7, 0.6, 0.5, 1), 0.3, 0.4, emissive=(1.0, 0.5, 0.25))
    tri = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32)
    nrm = np.tile([0, 0, 1], (3, 1)).astype(np.float32)
    g.node("Tri", mesh=g.mesh("Tri", [(tri, nrm, np.array([0, 1, 2], np.uint32), m)]))
    if with_degenerate:
        bad = np.array([[0.2, 0.2, 1], [0.2, 0.2, 1], [0.2, 0.2, 1],        # zero area
                        [0, -0.5, 0.5], [0, 0.5, 0.5], [0, 0, 2.5]], np.float32)  # in the plane x = 0: edge-on
        g.node("Bad", mesh=g.mesh("Bad", [(bad, np.tile([0, 0, 1], (6, 1)).astype(np.float32),
                                           np.array([0, 1, 2, 3, 4, 5], np.uint32), m)]))
    g.node("Camera", camera=g.camera(0.8), translation=(0, 0, 4))
    g.write(path)


@pytest.mark.gpu
@pytest.mark.parametrize("with_degenerate", [False, True])
def test_tiny_and_degenerate_scenes_match_oracle(device, tmp_path, with_degenerate):
    path = str(tmp_path / "tiny.gltf")
    _tiny_scene(path, with_degenerate)
    s = L.Scene(path, builder=L.BUILDER_BINNED_SAH)
    assert len(s.objects) == (2 if with_degenerate else 1)
    osc = O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh)
    device.upload_scene(s)
    for (w, h, bounces, frames) in ((37, 23, 3, 2), (1, 1, 1, 1), (8, 8, 16, 3)):  # ragged sizes, one pixel, deep
        ubos = L.frame_ubos(s.camera_world, s.fov_y, frames, seed=11)
        ref, st = O.render_frames(osc, ubos, w, h, bounces)
        for pipeline in (L.PIPELINE_WAVEFRONT, L.PIPELINE_MEGAKERNEL, L.PIPELINE_PERSISTENT, L.PIPELINE_HYBRID):
            device.resize(w, h)
            device.set_options(bounces, L.BSDF_HEITZ, pipeline, 0)
            device.render_frames(ubos)
            img = device.read_accum()
            assert device.stats().segments == st["segments"]
            assert np.array_equal(np.isnan(img), np.isnan(ref))
            assert np.array_equal(np.nan_to_num(img, nan=-1.0), np.nan_to_num(ref, nan=-1.0)), (w, h, pipeline)


def test_zero_throughput_path_that_draws_zero_at_roulette_turns_nan(tmp_path):
    """shaders/path_tracing.comp:317-323: `if (rand() > q) break; mask *= 1.0f / q;` — a path whose throughput
    is exactly zero (the sky shell of the open scene is black) is dropped unless rand() returns exactly 0.0, and
    then 0 * (1/0) makes its throughput NaN; the emission it meets afterwards makes its sample NaN.  The hash of
    shaders/common/random.glsl:9-15 does return 0.0 now and then.  Frame 19 of the benchmark's seed sequence does
    it at pixel (833, 634) of the 1920x1080 image: the restatement must keep that sample NaN (a renderer that
    clamps or skips it computes a different image from the reference's), and its neighbours finite."""
    path = str(tmp_path / "outside_synth.gltf")
    scenes.write_outside_synth(path, 1.0)
    s = L.Scene(path, builder=L.BUILDER_BINNED_SAH)
    osc = O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh)
    ubos = L.frame_ubos(s.camera_world, s.fov_y, 20, seed=0x5EED)
    ubos["reset"] = 1
    img, st = O.render_frames(osc, ubos[19:20], 1920, 1080, 8, tile=(832, 632, 8, 8), threads=1)
    block = img[632:640, 832:840]
    assert np.isnan(block[2, 1, :3]).all() and block[2, 1, 3] == 1.0
    finite = np.isfinite(block[..., :3]).all(axis=2)
    assert finite.sum() == 63
    assert st["segments"] >= 64 + 7  # the poisoned path runs all 8 bounces: roulette never fires on a NaN throughput
