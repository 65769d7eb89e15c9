"""Larger parity runs than the per-feature tests: tens of millions of ray segments per case, every
kernel organisation bit-identical to the oracle (images and segment counts).  Rare events — equal
hit distances, boxes grazed within rounding, walks at the order cap — only show up at such counts."""
import numpy as np
import pytest

import logipathtracer_amd as L
from logipathtracer_amd import scenes
from oracle import binding as O

pytestmark = pytest.mark.gpu


def _check(scene, w, h, frames, bounces, seed):
    osc = O.OracleScene(scene.objects, scene.object_bvh, scene.vertices, scene.mesh_bvh, getattr(scene, "textures", ()))
    ubos = L.frame_ubos(scene.camera_world, scene.fov_y, frames, seed=seed)
    ref, st = O.render_frames(osc, ubos, w, h, bounces)
    dev = L.Device(0)
    try:
        dev.upload_scene(scene)
        for pipeline in (L.PIPELINE_WAVEFRONT, L.PIPELINE_HYBRID, L.PIPELINE_PERSISTENT):
            dev.resize(w, h)
            dev.set_options(bounces, L.BSDF_HEITZ, pipeline, 0)
            dev.render_frames(ubos)
            img = dev.read_accum()
            assert dev.stats().segments == st["segments"]
            assert np.array_equal(img, ref), f"pipeline {pipeline}"
    finally:
        dev.close()


def test_cornell_20m_segments(cornell):
    _check(cornell, 1024, 768, 8, 8, 123)


def test_cornell_odd_size_ten_bounces(cornell):
    _check(cornell, 1000, 562, 6, 10, 77)


def test_rough_terrain_scene(tmp_path):
    path = str(tmp_path / "outside_synth.gltf")
    scenes.write_outside_synth(path, 1.0)
    _check(L.Scene(path), 640, 360, 4, 8, 5)
