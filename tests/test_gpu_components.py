"""GPU checks of the pieces around the path: device arithmetic, traversal, builders, batching,
tile sharding, the host renderer class and the display resolve — all through the C-ABI."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import logipathtracer_amd as L
from conftest import CORNELL, GOLDEN, rel_l2
from logipathtracer_amd.partition import Partition
from oracle import binding as O

pytestmark = pytest.mark.gpu
F32 = np.float32


def _oracle_vec(fn, a, b=None):
    if b is None:
        return np.array([fn(float(x)) for x in a], dtype=F32)
    return np.array([fn(float(x), float(y)) for x, y in zip(a, b)], dtype=F32)


def test_device_arithmetic_is_bit_identical_to_the_oracle(device):
    rng = np.random.default_rng(7)
    n = 20000
    ang = np.concatenate([rng.uniform(-7, 7, n - 6), [0.0, np.pi, -np.pi / 2, 1e-30, 8192.0, 9000.0]]).astype(F32)
    pos = np.concatenate([rng.uniform(0, 2, n - 5), [0.0, 1.0, 1e-40, 3e38, 5.9604645e-08]]).astype(F32)
    ex = np.concatenate([rng.uniform(-30, 30, n - 4), [-104.0, 89.0, 0.0, -87.5]]).astype(F32)
    den = rng.uniform(-3, 3, n).astype(F32)
    den[:3] = [0.0, 1e-39, -1e-39]
    with np.errstate(all="ignore"):
        checks = [
            (0, ang, None, _oracle_vec(O.lib.lpto_sin, ang)),
            (1, ang, None, _oracle_vec(O.lib.lpto_cos, ang)),
            (2, ang, None, _oracle_vec(O.lib.lpto_tan, ang)),
            (3, pos, None, _oracle_vec(O.lib.lpto_log, pos)),
            (4, ex, None, _oracle_vec(O.lib.lpto_exp, ex)),
            (5, pos, np.full(n, 2.4, F32), _oracle_vec(O.lib.lpto_pow, pos, np.full(n, 2.4, F32))),
            (6, ang, den, (ang / den).astype(F32)),          # IEEE divide
            (7, pos, None, np.sqrt(pos).astype(F32)),        # IEEE square root
            (8, pos, None, _oracle_vec(O.lib.lpto_srgb_to_linear, pos)),
        ]
    for op, a, b, want in checks:
        got = device.debug_math(op, a, b)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (op, a[~same][:4], got[~same][:4], want[~same][:4])


def _random_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform([-0.95, 0.05, -0.95], [0.95, 1.9, 0.95], size=(n, 3)).astype(F32)
    d = rng.normal(size=(n, 3)).astype(F32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o[: n // 4] = [6, 1, 0]
    d[: n // 4] = (np.array([-1, 0, 0]) + rng.uniform(-0.25, 0.25, size=(n // 4, 3))).astype(F32)
    d[0] = [0, 1, 0]   # axis-parallel directions: infinite reciprocals
    d[1] = [-1, 0, 0]
    return o, d


@pytest.mark.parametrize("builder", [L.BUILDER_SPLIT_BVH, L.BUILDER_BINNED_SAH, L.BUILDER_LBVH, L.BUILDER_PLOC])
def test_traversal_matches_oracle_hit_for_hit(device, builder):
    s = L.Scene(CORNELL, builder=builder, device=device)
    osc = O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh)
    device.upload_scene(s)
    o, d = _random_rays(4000, 5)
    t, obj, prim = device.debug_intersect(o, d)
    brute_miss = 0
    for i in range(len(o)):
        hit, tt, oo, pp = O.scene_intersect(osc, o[i], d[i])
        assert F32(t[i]) == tt and (not hit or (obj[i] == oo and prim[i] == pp)), i
        hb, tb, _, _ = O.scene_intersect(osc, o[i], d[i], brute_force=True)
        brute_miss += int(hb != hit or tb != tt)
    assert brute_miss <= 3  # the tree finds what a linear scan finds (up to slab-test rounding)


@pytest.mark.parametrize("which", [0, 1], ids=["lbvh", "ploc"])
def test_lbvh_tree_is_well_formed(device, which):
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 5, 64, 1000, 20011):
        c = rng.uniform(-1, 1, size=(n, 1, 3)).astype(F32) + rng.uniform(-0.05, 0.05, size=(n, 3, 3)).astype(F32)
        if n == 64:
            c[:] = c[0]  # all triangles identical: every Morton code equal, every merged box the same
        nodes, order = device.build_mesh_bvh(c, which)
        assert sorted(order.tolist()) == list(range(n))
        assert 1 <= len(nodes) <= max(1, 2 * n - 1)
        seen = np.zeros(n, dtype=np.int32)
        stack = [0]
        while stack:
            i = stack.pop()
            nd = nodes[i]
            if nd["isLeaf"]:
                a, b = int(nd["indices"][0]), int(nd["indices"][1])
                assert 0 < b - a <= 4
                tri = c[order[a:b]].reshape(-1, 3)
                assert np.all(tri.min(axis=0) >= nd["minCorner"]) and np.all(tri.max(axis=0) <= nd["maxCorner"])
                seen[a:b] += 1
            else:
                for ch in nd["indices"]:
                    chn = nodes[int(ch)]
                    assert np.all(chn["minCorner"] >= nd["minCorner"]) and np.all(chn["maxCorner"] <= nd["maxCorner"])
                    stack.append(int(ch))
        assert np.all(seen == 1)
    # two builds of the same mesh give the same tree (node numbering does not depend on scheduling)
    again, order2 = device.build_mesh_bvh(c, which)
    assert np.array_equal(again, nodes) and np.array_equal(order2, order)
    nodes, order = device.build_mesh_bvh(np.zeros((0, 3, 3), F32), which)
    assert len(nodes) == 1 and nodes[0]["isLeaf"] == 1 and nodes[0]["indices"][0] == nodes[0]["indices"][1]


def test_config1_matches_committed_golden_image(device):
    z = np.load(os.path.join(GOLDEN, "cornell_scene_binned_sah.npz"))
    s = L.Scene.from_buffers(z["objects"], z["object_bvh"], z["vertices"], z["mesh_bvh"])
    ubos = L.frame_ubos(z["camera_world"], float(z["fov_y"]), 1, seed=0x5EED)
    device.upload_scene(s)
    device.resize(256, 256)
    device.set_options(2, L.BSDF_HEITZ, L.PIPELINE_WAVEFRONT, 0)
    device.render_frames(ubos)
    img = device.read_accum()
    ref = np.load(os.path.join(GOLDEN, "cornell_c1_256x256_1spp_2b.npy"))
    meta = json.load(open(os.path.join(GOLDEN, "cornell_c1_counters.json")))
    assert device.stats().segments == meta["counters"]["segments"]
    assert rel_l2(img, ref) <= 1e-4
    assert np.array_equal(img, ref)


def test_frames_in_flight_do_not_change_pixels(device, cornell, cornell_oracle):
    w, h, n = 160, 96, 7
    ubos = L.frame_ubos(cornell.camera_world, cornell.fov_y, n, seed=21)
    ubos["reset"][4] = 1  # a reset in the middle of a batch
    ref, st = O.render_frames(cornell_oracle, ubos, w, h, 5)
    device.upload_scene(cornell)
    for fif in (1, 3, 7, 0):
        device.resize(w, h)
        device.set_options(5, L.BSDF_HEITZ, L.PIPELINE_WAVEFRONT, fif)
        device.render_frames(ubos)
        img = device.read_accum()
        assert device.stats().segments == st["segments"]
        assert np.array_equal(img, ref), fif
    # frame by frame through lpt_render_frame's path gives the same image
    device.resize(w, h)
    for i in range(n):
        device.render_frames(ubos[i:i + 1])
    assert np.array_equal(device.read_accum(), ref)


def test_tile_sharding_reproduces_the_whole_image(cornell, cornell_oracle):
    torch = pytest.importorskip("torch")
    w, h, tile, ranks, bounces = 200, 120, 32, 3, 4  # edge tiles stick out of the image
    ubos = L.frame_ubos(cornell.camera_world, cornell.fov_y, 2, seed=5)
    ref, st = O.render_frames(cornell_oracle, ubos, w, h, bounces)
    P = Partition(w, h, tile, tile, ranks)
    bufs, devs, seg = [], [], 0
    for r in range(ranks):
        dev = L.Device(0)
        devs.append(dev)
        dev.upload_scene(cornell)
        dev.resize(w, h)
        dev.set_partition(tile, tile, r, ranks)
        dev.set_options(bounces, L.BSDF_HEITZ, (L.PIPELINE_WAVEFRONT, L.PIPELINE_HYBRID, L.PIPELINE_PERSISTENT)[r % 3], 0)
        dev.render_frames(ubos)
        seg += dev.stats().segments
        assert dev.owned_slots() == P.n_slots
        part = dev.read_accum()
        owner, _ = P.pixel_to_slot(*np.meshgrid(np.arange(w), np.arange(h)))
        assert np.array_equal(part[owner == r], ref[owner == r]) and np.all(part[owner != r] == 0)
        n_floats = dev.owned_slots() * 4

        class _Arr:
            __cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (dev.owned_accum_device(), False), "version": 3}

        bufs.append(torch.as_tensor(_Arr(), device="cuda:0").clone())
    assert seg == st["segments"]
    gathered = torch.cat(bufs)  # what an all_gather of the rank buffers delivers
    image = torch.empty(h * w * 4, dtype=torch.float32, device="cuda:0")
    devs[0].assemble_gathered(gathered.data_ptr(), ranks, image.data_ptr())
    devs[0].synchronize()
    assert np.array_equal(image.cpu().numpy().reshape(h, w, 4), ref)
    assert np.array_equal(P.assemble(gathered.cpu().numpy().reshape(-1, 4)), ref)
    for d in devs:
        d.close()


def test_renderer_pt_frame_contract(cornell_oracle, cornell):
    w, h = 128, 72
    r = L.Renderer(w, h)
    r.configure(seed=0x5EED, max_bounces=4)
    r.draw_frame()            # no scene yet: a no-op (src/RendererPT.cpp:608-612)
    assert r.samples == 0
    r.load_scene(CORNELL)
    for _ in range(3):
        r.draw_frame()
    assert r.samples == 3
    ubos = L.frame_ubos(cornell.camera_world, cornell.fov_y, 3, seed=0x5EED)
    ref, _ = O.render_frames(cornell_oracle, ubos, w, h, 4)
    img = r.read_accumulation()
    assert np.array_equal(img, ref)
    assert np.array_equal(r.resolve(), O.resolve(ref, 1.0 / 3.0))
    # batched submission continues the same seed sequence
    r.draw_frames(2)
    ubos5 = L.frame_ubos(cornell.camera_world, cornell.fov_y, 5, seed=0x5EED)
    ref5, _ = O.render_frames(cornell_oracle, ubos5, w, h, 4)
    assert r.samples == 5 and np.array_equal(r.read_accumulation(), ref5)
    # moving the camera restarts accumulation (src/RendererPT.cpp:575-578)
    r.move_camera(2, -0.5)
    r.draw_frame()
    img = r.read_accumulation()
    assert r.samples == 1 and np.all(img[..., 3] == 1.0)
    # so does a resize (src/RendererPT.cpp:225-232)
    r.resize(64, 48)
    r.draw_frames(2)
    assert r.samples == 2 and r.read_accumulation().shape == (48, 64, 4)
    r.close()


def test_error_paths(device, cornell):
    dev = L.Device(0)
    with pytest.raises(L.LptError):
        dev.render_frames(L.frame_ubos(cornell.camera_world, cornell.fov_y, 1))  # nothing uploaded
    bad = L.Scene.from_buffers(cornell.objects.copy(), cornell.object_bvh, cornell.vertices, cornell.mesh_bvh)
    bad.objects["colorTexture"][0] = 0
    with pytest.raises(L.LptError, match="texture that was not uploaded"):
        dev.upload_scene(bad)
    bad = L.Scene.from_buffers(cornell.objects, cornell.object_bvh.copy(), cornell.vertices, cornell.mesh_bvh)
    inner = np.flatnonzero(bad.object_bvh["isLeaf"] == 0)
    bad.object_bvh["indices"][inner[0]] = [10 ** 6, 1]
    with pytest.raises(L.LptError, match="out of range"):
        dev.upload_scene(bad)
    with pytest.raises(L.LptError):
        dev.resize(0, 10)
    dev.upload_scene(cornell)
    dev.resize(64, 64)
    with pytest.raises(L.LptError):
        dev.set_partition(12, 8, 0, 1)  # tile not a multiple of 8
    with pytest.raises(L.LptError):
        dev.set_options(0, 0, 0, 0)
    dev.close()


def test_full_size_properties(device, cornell):
    """1920x1080, 8 bounces (the benchmark's configuration): both kernel organisations agree bit for bit,
    alpha counts samples, a repeat render is identical, and path statistics are consistent."""
    w, h = 1920, 1080
    ubos = L.frame_ubos(cornell.camera_world, cornell.fov_y, 2, seed=0x5EED)
    device.upload_scene(cornell)
    imgs = []
    for pipeline in (L.PIPELINE_WAVEFRONT, L.PIPELINE_MEGAKERNEL, L.PIPELINE_WAVEFRONT):
        device.resize(w, h)
        device.set_options(8, L.BSDF_HEITZ, pipeline, 0)
        device.render_frames(ubos)
        imgs.append((device.read_accum(), device.stats()))
    assert np.array_equal(imgs[0][0], imgs[1][0]) and np.array_equal(imgs[0][0], imgs[2][0])
    assert imgs[0][1].segments == imgs[1][1].segments and imgs[0][1].paths == 2 * w * h
    assert np.all(imgs[0][0][..., 3] == 2.0) and np.isfinite(imgs[0][0]).all()
    assert imgs[0][1].paths <= imgs[0][1].segments <= 8 * imgs[0][1].paths


@pytest.mark.gpu
def test_batch_shrinks_when_device_memory_is_short(cornell):
    """32 frames of 8192x8192 in flight would need 344 GB of ray queues; the renderer halves the batch
    until the pool fits and still renders every frame."""
    dev = L.Device(0)
    try:
        dev.upload_scene(cornell)
        dev.resize(8192, 8192)
        dev.set_options(1, L.BSDF_HEITZ, L.PIPELINE_WAVEFRONT, 32)
        dev.render_frames(L.frame_ubos(cornell.camera_world, cornell.fov_y, 32))
        st = dev.stats()
        assert st.paths == 32 * 8192 * 8192 and st.segments == st.paths
    finally:
        dev.close()


def _leaf(first, last, lo=(-1, -1, -1), hi=(1, 1, 1)):
    n = np.zeros(1, dtype=L._native.NODE_DTYPE)
    n["minCorner"], n["maxCorner"], n["isLeaf"], n["indices"] = lo, hi, 1, (first, last)
    return n


def test_upload_checks_leaf_size_and_shared_mesh_trees(cornell):
    """lpt_upload_scene refuses what the device layer's references cannot express: leaves of more than 63
    primitives, and mesh-tree nodes shared by objects that differ in bvhOffset or verticesOffset.  Instances
    (same bvhOffset, same verticesOffset) are accepted and render like the oracle."""
    dev = L.Device(0)
    try:
        obj = cornell.objects[:1].copy()
        obj["bvhOffset"], obj["verticesOffset"] = 0, 0
        verts = np.zeros(3 * 70, dtype=L._native.VERTEX_DTYPE)
        verts["position"] = np.tile(np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], F32), (70, 1))
        verts["normal"] = (0, 0, 1)
        top = _leaf(0, 1)
        big = L.Scene.from_buffers(obj, top, verts, _leaf(0, 64))
        with pytest.raises(L.LptError, match="more than 63"):
            dev.upload_scene(big)
        dev.upload_scene(L.Scene.from_buffers(obj, top, verts, _leaf(0, 63)))  # the limit itself is fine
        # two objects, one mesh tree, different vertex ranges: refused
        two = np.concatenate([obj, obj])
        two["verticesOffset"] = (0, 3)
        with pytest.raises(L.LptError, match="shared by objects") as e:
            dev.upload_scene(L.Scene.from_buffers(two, _leaf(0, 2), verts, _leaf(0, 4)))
        assert e.value.status == -2
        # an object whose tree starts inside another object's tree: refused as well
        inner = np.zeros(1, dtype=L._native.NODE_DTYPE)
        inner["minCorner"], inner["maxCorner"], inner["indices"] = (-1, -1, -1), (1, 1, 1), (1, 2)
        mesh = np.concatenate([inner, _leaf(0, 2), _leaf(2, 4)])
        two["verticesOffset"] = (0, 0)
        two["bvhOffset"] = (0, 1)
        with pytest.raises(L.LptError, match="shared by objects"):
            dev.upload_scene(L.Scene.from_buffers(two, _leaf(0, 2), verts, mesh))
        # true instances: same tree, same vertices, different transforms
        two["bvhOffset"] = (0, 0)
        two["worldMatrix"][1][12] = 2.5          # second instance moved along x
        two["worldMatrixInverse"][1][12] = -2.5
        cam = np.eye(4, dtype=F32)
        cam[3, :3] = (1.2, 0, 6)
        inst = L.Scene.from_buffers(two, _leaf(0, 2, (-1, -1, -1), (3.5, 1, 1)), verts, mesh, cam.reshape(16), 0.8)
        ubos = L.frame_ubos(inst.camera_world, inst.fov_y, 2, seed=9)
        ref, st = O.render_frames(O.OracleScene(inst.objects, inst.object_bvh, inst.vertices, inst.mesh_bvh), ubos, 96, 64, 3)
        assert st["hits"] > 0
        dev.upload_scene(inst)
        dev.resize(96, 64)
        for pipeline in (L.PIPELINE_WAVEFRONT, L.PIPELINE_HYBRID, L.PIPELINE_MEGAKERNEL):
            dev.set_options(3, L.BSDF_HEITZ, pipeline, 0)
            dev.render_frames(ubos)
            assert dev.stats().segments == st["segments"]
            assert np.array_equal(dev.read_accum(), ref)
    finally:
        dev.close()


def test_persistent_launch_out_of_rounds_is_reported(cornell, cornell_oracle):
    """A persistent launch that runs out of scheduling rounds leaves paths unfinished; the next call that
    waits for the stream must say so (LPT_ERR_TRUNCATED) instead of handing out an incomplete image, and
    the context must work again afterwards."""
    w, h, bounces = 256, 256, 8
    ubos = L.frame_ubos(cornell.camera_world, cornell.fov_y, 4, seed=13)
    dev = L.Device(0)
    try:
        dev.upload_scene(cornell)
        dev.resize(w, h)
        for pipeline in (L.PIPELINE_PERSISTENT, L.PIPELINE_HYBRID, L.PIPELINE_WAVEFRONT):
            dev.set_options(bounces, L.BSDF_HEITZ, pipeline, 0)
            # (the wavefront organisation hands its short queues to a persistent launch once a finished batch
            # has reported how long they are: these small ones from bounce 1 on)
            dev.render_frames(ubos)
            dev.synchronize()
            dev.debug_set_trace_round_limit(16)
            dev.render_frames(ubos)
            with pytest.raises(L.LptError) as e:
                dev.stats()
            assert e.value.status == -6 and "round limit" in str(e.value)
            dev.synchronize()  # the flag is reported once
            dev.render_frames(ubos)
            with pytest.raises(L.LptError):
                dev.read_accum()
            dev.debug_set_trace_round_limit(0)
            dev.resize(w, h)
            dev.render_frames(ubos)
            ref, st = O.render_frames(cornell_oracle, ubos, w, h, bounces)
            assert dev.stats().segments == st["segments"]
            assert np.array_equal(dev.read_accum(), ref)
    finally:
        dev.close()


def test_scene_loads_on_a_second_thread_while_the_draw_loop_spins(tmp_path, cornell, cornell_oracle):
    """src/Main.cpp:45,96: loadScene runs on a thread of its own while the main loop calls drawFrame, which
    must render nothing until the scene is in place (src/RendererPT.cpp:608-612) — and what it renders
    afterwards is what a serial load would have given."""
    from logipathtracer_amd import scenes
    w, h, bounces = 160, 90, 5
    big = str(tmp_path / "outside.gltf")
    scenes.write_outside_synth(big, 0.3)  # takes the builder a moment
    for path, builder in ((CORNELL, L.BUILDER_SPLIT_BVH), (big, L.BUILDER_SPLIT_BVH)):
        r = L.Renderer(w, h)
        r.configure(seed=0x5EED, max_bounces=bounces)
        r.load_scene_async(path, builder)
        spins = 0
        while not r.scene_loaded:
            r.draw_frame()                 # no-op: no scene yet
            assert r.samples == 0 or r.scene_loaded
            spins += 1
        r.wait_for_scene()
        assert spins > 0
        assert r.samples <= 1  # the call that saw the scene arrive may already have drawn the first frame
        while r.samples < 3:
            r.draw_frame()
        threaded = r.read_accumulation()
        r.close()
        serial = L.Renderer(w, h)
        serial.configure(seed=0x5EED, max_bounces=bounces)
        serial.load_scene(path, builder)
        for _ in range(3):
            serial.draw_frame()
        assert np.array_equal(threaded, serial.read_accumulation())
        serial.close()
        if path == CORNELL:
            ubos = L.frame_ubos(cornell.camera_world, cornell.fov_y, 3, seed=0x5EED)
            ref, _ = O.render_frames(cornell_oracle, ubos, w, h, bounces)
            assert np.array_equal(threaded, ref)
    # a failing load on the thread is reported by the wait, and the renderer stays usable
    r = L.Renderer(w, h)
    r.load_scene_async(str(tmp_path / "missing.gltf"))
    with pytest.raises(L.LptError, match="failed to open"):
        r.wait_for_scene()
    r.draw_frame()
    assert r.samples == 0
    r.close()


def test_deferred_submission_renders_the_same_image(cornell):
    """RendererPT::setDeferredFrames: drawFrame() queues frames and renders them in batches; every observation
    (read, resolve) and every change later frames depend on (camera, options, size) sees the same image as the
    reference's frame-by-frame submission."""
    w, h = 144, 80

    def script(r):
        out = []
        r.configure(seed=77, max_bounces=5)
        r.load_scene(CORNELL)
        for _ in range(7):
            r.draw_frame()
        out.append((r.samples, r.read_accumulation()))
        r.move_camera(0, 0.25)            # restarts accumulation in the middle of a queue
        for _ in range(5):
            r.draw_frame()
        r.move_camera(4, 0.1)
        for _ in range(3):
            r.draw_frame()
        out.append((r.samples, r.resolve()))
        r.configure(seed=78, max_bounces=3)   # option change: queued frames keep the old depth
        for _ in range(4):
            r.draw_frame()
        out.append((r.samples, r.read_accumulation()))
        r.resize(96, 64)
        r.draw_frame()
        r.draw_frames(3)
        out.append((r.samples, r.read_accumulation()))
        return out

    a = L.Renderer(w, h)
    want = script(a)
    a.close()
    for n in (4, 64):
        b = L.Renderer(w, h)
        b.set_deferred_frames(n)
        got = script(b)
        b.close()
        assert [s for s, _ in got] == [s for s, _ in want] == [7, 3, 7, 4]
        for (_, x), (_, y) in zip(got, want):
            assert x.shape == y.shape and np.array_equal(x, y), n


def test_caller_owned_stream(cornell, cornell_oracle):
    """lpt_set_stream: the work runs on a stream the caller owns (here a torch stream), ordered with the caller's
    own work on it; NULL gives the context its stream back."""
    torch = pytest.importorskip("torch")
    w, h, bounces = 160, 96, 4
    ubos = L.frame_ubos(cornell.camera_world, cornell.fov_y, 3, seed=17)
    ref, st = O.render_frames(cornell_oracle, ubos, w, h, bounces)
    dev = L.Device(0)
    try:
        dev.upload_scene(cornell)
        dev.resize(w, h)
        dev.set_options(bounces, L.BSDF_HEITZ, L.PIPELINE_AUTO, 0)
        stream = torch.cuda.Stream(device=0)
        dev.set_stream(stream.cuda_stream)
        dev.render_frames(ubos[:2])
        # the compact buffer is read by torch on the same stream: ordered behind the render without a host sync
        n_floats = dev.owned_slots() * 4

        class _Arr:
            __cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (dev.owned_accum_device(), False), "version": 3}

        with torch.cuda.stream(stream):
            alpha_sum = torch.as_tensor(_Arr(), device="cuda:0").view(-1, 4)[:, 3].sum()
        stream.synchronize()
        assert float(alpha_sum) == 2.0 * w * h
        dev.render_frame_static(ubos[2:3])   # the entry point that allocates nothing uses the same stream
        stream.synchronize()
        assert np.array_equal(dev.read_accum(), ref)
        dev.set_stream(None)
        dev.resize(w, h)
        dev.render_frames(ubos)
        assert np.array_equal(dev.read_accum(), ref) and dev.stats().segments == st["segments"]
    finally:
        dev.close()


def test_automatic_organisation_follows_batch_size_and_measured_cost(cornell):
    """LPT_PIPELINE_AUTO: small batches run the hybrid organisation, large ones the wavefront kernels; once a call of the
    same scene, size and options has finished, the choice uses what that call took per path, and it stays the same
    for batches far from the crossing (one and sixteen 1080p frames of the Cornell box; they cross at nine).  The image
    does not depend on the choice."""
    w, h, bounces = 1920, 1080, 4
    ubos = L.frame_ubos(cornell.camera_world, cornell.fov_y, 16, seed=5)
    dev = L.Device(0)
    try:
        dev.upload_scene(cornell)
        dev.resize(w, h)
        dev.set_options(bounces, L.BSDF_HEITZ, L.PIPELINE_AUTO, 0)
        chosen = []
        for k in range(3):  # no finished call yet, then two calls that have one
            dev.render_frames(ubos[k:k + 1])
            chosen.append(dev.stats().pipeline)
        assert chosen == [L.PIPELINE_HYBRID] * 3
        dev.resize(w, h)
        for k in range(3):
            dev.render_frames(ubos)
            chosen.append(dev.stats().pipeline)
        assert chosen[3:] == [L.PIPELINE_WAVEFRONT] * 3
        auto = dev.read_accum()
        dev.resize(w, h)
        dev.set_options(bounces, L.BSDF_HEITZ, L.PIPELINE_HYBRID, 0)
        for k in range(3):
            dev.render_frames(ubos)
        assert np.array_equal(dev.read_accum(), auto)
    finally:
        dev.close()
