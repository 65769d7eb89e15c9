"""Parity fuzzing: scenes drawn from a seed (logipathtracer_amd.scenes.write_random_scene: degenerate and sliver
triangles, mirrored and nearly flat scales, materials over the whole parameter range, misfitting normals, cameras
inside objects) rendered by every kernel organisation and compared with the oracle bit for bit, NaNs included."""
import numpy as np
import pytest

import logipathtracer_amd as L
from logipathtracer_amd import scenes
from oracle import binding as O

PIPELINES = (L.PIPELINE_WAVEFRONT, L.PIPELINE_MEGAKERNEL, L.PIPELINE_PERSISTENT, L.PIPELINE_HYBRID)


def _same(img, ref):
    return np.array_equal(np.isnan(img), np.isnan(ref)) and \
        np.array_equal(np.nan_to_num(img, nan=-1.0).view(np.uint32), np.nan_to_num(ref, nan=-1.0).view(np.uint32))


def test_random_scenes_convert_and_render_on_the_oracle(tmp_path):
    """CPU: the generator is deterministic, the converter accepts what it writes and the oracle renders it."""
    for seed in (0, 1, 2):
        p = str(tmp_path / f"r{seed}.gltf")
        n = scenes.write_random_scene(p, seed)
        again = str(tmp_path / f"again{seed}.gltf")
        assert scenes.write_random_scene(again, seed) == n
        assert open(p.replace(".gltf", ".bin"), "rb").read() == open(again.replace(".gltf", ".bin"), "rb").read()
        s = L.Scene(p)
        assert len(s.vertices) >= 3 * n  # spatial splits may reference a triangle from several leaves
        osc = O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh)
        _, st = O.render_frames(osc, L.frame_ubos(s.camera_world, s.fov_y, 1, seed=seed), 32, 24, 6)
        assert st["paths"] == 32 * 24 and st["stackOverflow"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(64))
@pytest.mark.parametrize("hard", [False, True], ids=["plain", "hard"])
def test_random_scene_matches_oracle(device, tmp_path, seed, hard):
    path = str(tmp_path / "random.gltf")
    scenes.write_random_scene(path, seed, hard)
    builder = (L.BUILDER_SPLIT_BVH, L.BUILDER_BINNED_SAH, L.BUILDER_LBVH, L.BUILDER_PLOC)[seed % 4]
    s = L.Scene(path, builder=builder, device=device if builder in L.GPU_BUILDERS else None)
    osc = O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh)
    w, h = ((96, 64), (61, 47), (128, 40))[seed % 3]
    bounces = (12, 5, 16)[seed % 3]
    bsdf = L.BSDF_BASIC if seed % 6 == 5 else L.BSDF_HEITZ
    ubos = L.frame_ubos(s.camera_world, s.fov_y, 3, seed=1000 + seed)
    ref, st = O.render_frames(osc, ubos, w, h, bounces, bsdf_mode=bsdf)
    assert st["stackOverflow"] == 0
    device.upload_scene(s)
    for pipeline in PIPELINES:
        device.resize(w, h)
        device.set_options(bounces, bsdf, pipeline, 0)
        device.render_frames(ubos)
        img = device.read_accum()
        assert device.stats().segments == st["segments"], (seed, pipeline)
        assert _same(img, ref), (seed, pipeline, int((img.view(np.uint32) != ref.view(np.uint32)).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_random_call_sequences_keep_the_image_equal_to_the_oracle(tmp_path, cornell, seed):
    """One context driven through a seeded sequence of resize / partition / option / scene changes and renders
    (batched, single-frame, allocation-free, with and without reset); a host-side model of the accumulation image
    built from oracle renders must match read-back, segment counts and the resolve after every render."""
    from logipathtracer_amd.partition import Partition
    rng = np.random.default_rng(4242 + seed)
    scene_files = []
    for k in range(2):
        p = str(tmp_path / f"s{k}.gltf")
        scenes.write_random_scene(p, 100 + 10 * seed + k, hard=bool(k))
        scene_files.append(L.Scene(p, builder=(L.BUILDER_SPLIT_BVH, L.BUILDER_BINNED_SAH)[k]))
    scene_list = [cornell] + scene_files
    oracles = [O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh) for s in scene_list]
    sizes = [(64, 48), (1, 1), (37, 23), (8, 8), (96, 40), (65, 33), (16, 9)]
    dev = L.Device(0)
    try:
        cur = 0
        dev.upload_scene(scene_list[cur])
        w, h = sizes[0]
        dev.resize(w, h)
        part, rank = Partition(w, h, None, None, 1), 0
        bounces, bsdf = 4, L.BSDF_HEITZ
        dev.set_options(bounces, bsdf, L.PIPELINE_AUTO, 0)
        expected = np.zeros((h, w, 4), np.float32)
        spp = 0
        renders = 0
        for step in range(40):
            op = rng.choice(["render", "render", "render", "static", "resize", "partition", "options", "scene"])
            if op == "resize":
                w, h = sizes[int(rng.integers(len(sizes)))]
                dev.resize(w, h)
                part, rank = Partition(w, h, None, None, 1), 0
                expected = np.zeros((h, w, 4), np.float32)
                spp = 0
            elif op == "partition":
                tw, th = int(rng.choice([8, 16, 32, 64])), int(rng.choice([8, 16, 32]))
                n = int(rng.integers(1, 6))
                rank = int(rng.integers(n))
                dev.set_partition(tw, th, rank, n)
                part = Partition(w, h, tw, th, n)
                assert dev.owned_slots() == part.n_slots
                expected = np.zeros((h, w, 4), np.float32)
                spp = 0
            elif op == "options":
                bounces = int(rng.choice([1, 2, 3, 5, 8, 12, 16, 20]))
                bsdf = int(rng.choice([L.BSDF_HEITZ, L.BSDF_HEITZ, L.BSDF_BASIC]))
                pipeline = int(rng.choice([L.PIPELINE_WAVEFRONT, L.PIPELINE_MEGAKERNEL, L.PIPELINE_PERSISTENT,
                                           L.PIPELINE_AUTO, L.PIPELINE_HYBRID]))
                dev.set_options(bounces, bsdf, pipeline, int(rng.choice([0, 0, 1, 2, 7])))
            elif op == "scene":
                cur = int(rng.integers(len(scene_list)))
                dev.upload_scene(scene_list[cur])
            else:
                s = scene_list[cur]
                frames = 1 if op == "static" else int(rng.integers(1, 6))
                reset = bool(rng.random() < 0.3) or spp == 0
                ubos = L.frame_ubos(s.camera_world, s.fov_y, frames, seed=int(rng.integers(1 << 30)), first_reset=reset)
                segments = 0
                for tile in part.owned_tiles(rank):
                    _, st = O.render_frames(oracles[cur], ubos, w, h, bounces, bsdf_mode=bsdf, tile=tile, accum=expected)
                    segments += st["segments"]
                if op == "static":
                    dev.render_frame_static(ubos[0])
                elif frames == 1 and rng.random() < 0.5:
                    dev.render_frames(ubos[0:1])
                else:
                    dev.render_frames(ubos)
                spp = frames if reset else spp + frames
                img = dev.read_accum()
                assert dev.stats().segments == segments, (seed, step, op)
                assert _same(img, expected), (seed, step, op, int((img.view(np.uint32) != expected.view(np.uint32)).sum()))
                renders += 1
                if renders % 4 == 0:
                    assert _same(dev.resolve(1.0 / spp), O.resolve(expected, 1.0 / spp)), (seed, step, "resolve")
        assert renders >= 8
    finally:
        dev.close()
