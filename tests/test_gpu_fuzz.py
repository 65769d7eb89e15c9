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


def _reachable_tree_is_safe(nodes, root, base, leaf_ok):
    """Walks the binary tree at nodes[root] (child indices offset by `base`) the way a traversal would: False if it
    would read outside the array, never end (a node among its own descendants) or meet a leaf `leaf_ok` refuses."""
    n = len(nodes)
    if root >= n:
        return False
    state = {}  # 1: on the current path, 2: finished
    stack = [(root, False)]
    while stack:
        i, leaving = stack.pop()
        if leaving:
            state[i] = 2
            continue
        if state.get(i) == 1:
            return False
        if state.get(i) == 2:
            continue
        state[i] = 1
        stack.append((i, True))
        a, b = int(nodes["indices"][i][0]), int(nodes["indices"][i][1])
        if nodes["isLeaf"][i]:
            if not leaf_ok(a, b):
                return False
        else:
            if base + a >= n or base + b >= n:
                return False
            stack += [(base + a, False), (base + b, False)]
    return True


@pytest.mark.gpu
def test_upload_refuses_buffers_a_traversal_could_not_walk(device, cornell):
    """Corrupts single fields of the interchange buffers (child indices, leaf ranges, leaf flags, offsets, texture
    ids).  Whatever lpt_upload_scene accepts must be walkable: an independent host-side check of the reachable
    trees agrees that no traversal would leave the arrays or loop.  Nothing is rendered from corrupted buffers."""
    rng = np.random.default_rng(99)
    n_obj, n_vert = len(cornell.objects), len(cornell.vertices)
    values = [0, 1, 2, 3, 62, 63, 64, 65, 1000, n_obj - 1, n_obj, n_obj + 1, n_vert // 3, n_vert // 3 + 1, n_vert,
              len(cornell.mesh_bvh) - 1, len(cornell.mesh_bvh), 2**24, 2**25, 2**31 - 1, 2**31, 2**32 - 1]
    accepted = refused = 0
    try:
        for it in range(1500):
            objects, obvh = cornell.objects.copy(), cornell.object_bvh.copy()
            mbvh = cornell.mesh_bvh.copy()
            for _ in range(int(rng.integers(1, 3))):
                v = int(values[int(rng.integers(len(values)))]) if rng.random() < 0.8 else int(rng.integers(0, 2**32))
                kind = int(rng.integers(0, 7))
                if kind == 0:
                    obvh["indices"][int(rng.integers(len(obvh)))][int(rng.integers(2))] = v
                elif kind == 1:
                    mbvh["indices"][int(rng.integers(len(mbvh)))][int(rng.integers(2))] = v
                elif kind == 2:
                    i = int(rng.integers(len(obvh)))
                    obvh["isLeaf"][i] = 0 if obvh["isLeaf"][i] else int(rng.choice([1, 7, 2**31]))
                elif kind == 3:
                    i = int(rng.integers(len(mbvh)))
                    mbvh["isLeaf"][i] = 0 if mbvh["isLeaf"][i] else 1
                elif kind == 4:
                    objects["bvhOffset"][int(rng.integers(n_obj))] = v
                elif kind == 5:
                    objects["verticesOffset"][int(rng.integers(n_obj))] = v
                else:
                    objects[str(rng.choice(["colorTexture", "emissionTexture", "metallicRoughnessTexture",
                                            "transmissionTexture", "normalTexture"]))][int(rng.integers(n_obj))] = v
            s = L.Scene.from_buffers(objects, obvh, cornell.vertices, mbvh, cornell.camera_world, cornell.fov_y)
            try:
                device.upload_scene(s)
            except L.LptError:
                refused += 1
                continue
            accepted += 1
            ok = _reachable_tree_is_safe(obvh, 0, 0, lambda a, b: a <= b <= n_obj or b <= a)
            entered = set()
            stack = [0]
            while stack and ok:  # objects the object tree's leaves name
                i = stack.pop()
                a, b = int(obvh["indices"][i][0]), int(obvh["indices"][i][1])
                if obvh["isLeaf"][i]:
                    entered.update(range(a, b))
                else:
                    stack += [a, b]
            for o in sorted(entered):
                off, vo = int(objects["bvhOffset"][o]), int(objects["verticesOffset"][o])
                ok = ok and _reachable_tree_is_safe(mbvh, off, off, lambda a, b: b <= a or vo + 3 * b <= n_vert)
                for t in ("colorTexture", "emissionTexture", "metallicRoughnessTexture", "transmissionTexture", "normalTexture"):
                    ok = ok and int(objects[t][o]) == 0xFFFFFFFF  # no textures were uploaded
            assert ok, f"mutation {it} was accepted but is not walkable"
    finally:
        device.upload_scene(cornell)  # the session's device goes on with a sound scene
    assert refused > 300 and accepted > 50, (accepted, refused)


def _chain_scene(n_tris):
    """One object whose mesh tree is a chain: inner node k holds leaf k (pushed) and inner node k + 1 (visited
    first), so a ray through all the boxes stacks one leaf per level: the traversal needs about n_tris + 2 entries."""
    objects = np.zeros(1, dtype=L._native.OBJECT_DTYPE)
    eye = np.eye(4, dtype=np.float32).reshape(16)
    objects["worldMatrix"][0] = eye
    objects["worldMatrixInverse"][0] = eye
    objects["baseColorFactor"][0] = (0.8, 0.7, 0.6, 1.0)
    objects["emissionFactor"][0] = (0.5, 0.25, 0.125)
    objects["roughnessFactor"][0] = 0.5
    objects["ior"][0] = 1.5
    for t in ("colorTexture", "emissionTexture", "metallicRoughnessTexture", "transmissionTexture", "normalTexture"):
        objects[t][0] = 0xFFFFFFFF
    vertices = np.zeros(3 * n_tris, dtype=L._native.VERTEX_DTYPE)
    for k in range(n_tris):  # triangles stacked along z, shrinking towards the camera so that each is partly visible
        s = 1.0 - 0.5 * k / n_tris
        vertices["position"][3 * k:3 * k + 3] = [(-s, -s, 0.05 * k), (s, -s, 0.05 * k), (0.0, s, 0.05 * k)]
    vertices["normal"][:] = (0.0, 0.0, 1.0)
    n_nodes = 2 * n_tris - 1
    nodes = np.zeros(n_nodes, dtype=L._native.NODE_DTYPE)

    def bounds(lo, hi):
        p = vertices["position"][3 * lo:3 * hi]
        return p.min(axis=0), p.max(axis=0)

    for k in range(n_tris - 1):
        inner, leaf = 2 * k, 2 * k + 1
        nodes["minCorner"][inner], nodes["maxCorner"][inner] = bounds(k, n_tris)
        nodes["indices"][inner] = (leaf, 2 * (k + 1))  # x = this level's leaf, y = the rest of the chain
        nodes["minCorner"][leaf], nodes["maxCorner"][leaf] = bounds(k, k + 1)
        nodes["isLeaf"][leaf] = 1
        nodes["indices"][leaf] = (k, k + 1)
    last = 2 * (n_tris - 1)
    nodes["minCorner"][last], nodes["maxCorner"][last] = bounds(n_tris - 1, n_tris)
    nodes["isLeaf"][last] = 1
    nodes["indices"][last] = (n_tris - 1, n_tris)
    obvh = np.zeros(1, dtype=L._native.NODE_DTYPE)
    obvh["minCorner"][0], obvh["maxCorner"][0] = bounds(0, n_tris)
    obvh["isLeaf"][0] = 1
    obvh["indices"][0] = (0, 1)
    cam = np.eye(4, dtype=np.float32)
    cam[3, :3] = (0.0, 0.0, 6.0)  # row-major view of the column-major matrix: translation in elements 12..14
    return L.Scene.from_buffers(objects, obvh, vertices, nodes, cam.reshape(16), 0.6)


@pytest.mark.gpu
@pytest.mark.parametrize("n_tris", [14, 22, 30, 44, 60])
def test_deep_trees_run_on_the_larger_stack_variants(device, n_tris):
    """Chains of 14 to 60 levels need 16 to 62 stack entries: the persistent kernels run the instance that holds them
    (16, 20, 24, 32, 48, 64), the wavefront organisation's search kernel its guarded 16-entry instance from 21 entries
    on, whose rays outgrow the LDS stack here and are searched again by the follow-up launch; all trace like the oracle.
    (test_search_kernel_instances_forced_either_way runs this with the choice forced both ways.)"""
    s = _chain_scene(n_tris)
    osc = O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh)
    ubos = L.frame_ubos(s.camera_world, s.fov_y, 2, seed=n_tris)
    w, h, bounces = 40, 40, 3
    ref, st = O.render_frames(osc, ubos, w, h, bounces)
    assert st["stackOverflow"] == 0 and st["maxStackBlas"] >= n_tris
    assert st["hits"] > 0
    device.upload_scene(s)
    for pipeline in PIPELINES:
        device.resize(w, h)
        device.set_options(bounces, L.BSDF_HEITZ, pipeline, 0)
        device.render_frames(ubos)
        assert device.stats().segments == st["segments"], (n_tris, pipeline)
        assert _same(device.read_accum(), ref), (n_tris, pipeline)


@pytest.mark.gpu
def test_a_tree_deeper_than_the_largest_stack_is_refused(device, cornell):
    try:
        with pytest.raises(L.LptError, match="stack"):
            device.upload_scene(_chain_scene(80))
    finally:
        device.upload_scene(cornell)


@pytest.mark.gpu
def test_object_tree_shapes_the_converter_never_emits(device, cornell):
    """Object trees are the caller's: one leaf naming all objects, leaves of several objects beside an empty leaf,
    and instances (objects that share a mesh tree and its vertices under different transforms) trace like the oracle."""
    n = len(cornell.objects)
    boxes_lo = cornell.object_bvh["minCorner"].min(axis=0)
    boxes_hi = cornell.object_bvh["maxCorner"].max(axis=0)

    def node(leaf, a, b, lo=boxes_lo, hi=boxes_hi):
        r = np.zeros(1, dtype=L._native.NODE_DTYPE)
        r["minCorner"][0], r["maxCorner"][0], r["isLeaf"][0], r["indices"][0] = lo, hi, leaf, (a, b)
        return r

    # an instance: the last object again, moved up by 0.3 (its inverse matrix moves rays down by 0.3)
    objects = np.concatenate([cornell.objects, cornell.objects[-1:]])
    wm = objects["worldMatrix"][-1].reshape(4, 4).copy()   # column-major: row 3 of this view is the translation
    wi = objects["worldMatrixInverse"][-1].reshape(4, 4).copy()
    shift = np.eye(4, dtype=np.float32)
    shift[3, 1] = 0.3
    unshift = np.eye(4, dtype=np.float32)
    unshift[3, 1] = -0.3
    objects["worldMatrix"][-1] = (wm @ shift).reshape(16)        # world = T * M  (column-major product, row view)
    objects["worldMatrixInverse"][-1] = (unshift @ wi).reshape(16)  # inverse = M^-1 * T^-1
    big_lo, big_hi = boxes_lo - 0.5, boxes_hi + 0.5
    shapes = {
        "one leaf": (cornell.objects, node(1, 0, n)),
        "leaves of several objects and an empty one": (cornell.objects, np.concatenate([
            node(0, 1, 2), node(0, 3, 4), node(1, 5, 5), node(1, 0, 6), node(1, 6, n)])),
        "instance": (objects, np.concatenate([node(0, 1, 2, big_lo, big_hi), node(1, 0, n, big_lo, big_hi),
                                              node(1, n, n + 1, big_lo, big_hi)])),
    }
    try:
        for name, (objs, obvh) in shapes.items():
            s = L.Scene.from_buffers(objs, obvh, cornell.vertices, cornell.mesh_bvh, cornell.camera_world, cornell.fov_y)
            osc = O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh)
            ubos = L.frame_ubos(s.camera_world, s.fov_y, 2, seed=17)
            w, h, bounces = 96, 64, 6
            ref, st = O.render_frames(osc, ubos, w, h, bounces)
            device.upload_scene(s)
            for pipeline in PIPELINES:
                device.resize(w, h)
                device.set_options(bounces, L.BSDF_HEITZ, pipeline, 0)
                device.render_frames(ubos)
                assert device.stats().segments == st["segments"], (name, pipeline)
                assert _same(device.read_accum(), ref), (name, pipeline)
    finally:
        device.upload_scene(cornell)


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1], ids=["lbvh", "ploc"])
def test_gpu_builders_keep_their_shape_on_hostile_triangles(device, which):
    """NaN and infinite corners, coordinates 30 orders of magnitude apart, all triangles on one line or in one point:
    the GPU builders still return a tree over every triangle exactly once with leaves of at most four (boxes of
    non-finite input are whatever min/max make of it), or refuse with a message; they never hang or fault."""
    rng = np.random.default_rng(5 + which)
    n = 777
    base = (rng.uniform(-1, 1, size=(n, 1, 3)) + rng.uniform(-0.05, 0.05, size=(n, 3, 3))).astype(np.float32)
    cases = {}
    c = base.copy(); c[rng.random(n) < 0.1] = np.nan; cases["a tenth NaN"] = c
    c = base.copy(); c[:] = np.nan; cases["all NaN"] = c
    c = base.copy(); c[rng.random(n) < 0.05, 0, 0] = np.inf; c[rng.random(n) < 0.05, 1, 2] = -np.inf; cases["infinities"] = c
    c = base.copy(); c[::7] *= 1e30; c[1::7] *= 1e-30; cases["sixty orders of magnitude"] = c
    c = np.zeros_like(base); c[:, :, 0] = np.linspace(0, 1, n)[:, None]; cases["on one line"] = c
    cases["one point"] = np.full_like(base, 0.25)
    c = base.copy(); c[:] = 3.0e38; cases["at the top of the float range"] = c
    for name, corners in cases.items():
        try:
            nodes, order = device.build_mesh_bvh(corners, which)
        except L.LptError as e:
            assert str(e), name
            continue
        assert sorted(order.tolist()) == list(range(n)), name
        assert 1 <= len(nodes) <= 2 * n - 1, name
        seen = np.zeros(n, dtype=np.int32)
        visited = set()
        stack = [0]
        while stack:
            i = stack.pop()
            assert i not in visited and i < len(nodes), name
            visited.add(i)
            nd = nodes[i]
            a, b = int(nd["indices"][0]), int(nd["indices"][1])
            if nd["isLeaf"]:
                assert 0 < b - a <= 4 and b <= n, name
                seen[a:b] += 1
            else:
                stack += [a, b]
        assert np.all(seen == 1), name


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1], ids=["lbvh", "ploc"])
def test_gpu_builders_stay_logarithmic_on_many_equal_boxes(device, which):
    """20 000 identical triangles, 20 000 triangles along one axis line (all merged boxes have area 0), and extents
    whose products underflow to 0 or overflow to infinity: every candidate pair of the clustering builder then has
    the same area.  The build must succeed in a number of rounds that does not grow with the triangle count (seen
    here as a balanced tree: a builder that merged one pair per round would return a chain 20 000 deep, or give up)."""
    import time
    n = 20000
    one = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]], dtype=np.float32)
    cases = {"identical": np.broadcast_to(one, (n, 3, 3)).copy()}
    c = np.zeros((n, 3, 3), dtype=np.float32)
    c[:, :, 0] = (np.arange(n, dtype=np.float32)[:, None] + np.array([0.0, 0.4, 0.8], dtype=np.float32)[None, :])
    cases["on one axis line"] = c
    rng = np.random.default_rng(9)
    cases["extents that underflow"] = (rng.uniform(0, 1, size=(n, 3, 3)) * 1e-25).astype(np.float32)
    cases["extents that overflow"] = (rng.uniform(-1, 1, size=(n, 3, 3)) * 3e25).astype(np.float32)
    for name, corners in cases.items():
        t0 = time.perf_counter()
        nodes, order = device.build_mesh_bvh(corners, which)
        dt = time.perf_counter() - t0
        assert sorted(order.tolist()) == list(range(n)), name
        depth, worst, covered = {0: 1}, 1, 0
        stack = [0]
        while stack:
            i = stack.pop()
            nd = nodes[i]
            a, b = int(nd["indices"][0]), int(nd["indices"][1])
            if nd["isLeaf"]:
                assert 0 < b - a <= 4, name
                covered += b - a
            else:
                for ch in (a, b):
                    depth[ch] = depth[i] + 1
                    worst = max(worst, depth[ch])
                    stack.append(ch)
        assert covered == n, name
        if which == 1:
            assert worst <= 40, (name, worst)  # log2(20000 / 4) is 12.3
        assert dt < 20.0, (name, dt)


@pytest.mark.gpu
def test_walks_carried_across_many_groups_of_hits_match_the_oracle():
    """k_shade sets Heitz walks that go on aside in a pool and advances them whenever enough wait; with the usual grid a
    small test image gives a workgroup one or two groups of hits, so the pool is mostly run to its end at once.  Here
    the grid is capped at one workgroup per CU (the cap is read when the library first launches the kernel, hence a
    process of its own), so that every workgroup carries walks across many groups: random scenes with rough, metallic
    and transmissive materials, 320x240, six frames, every organisation, bit for bit against the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LPT_SHADE_BLOCKS_PER_CU="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz", "parity_sweep.py"), "9400", "9406", "320", "240", "6"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "done 9400 9406 mismatches 0" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("exact", ["0", "1"], ids=["guarded", "exact"])
def test_search_kernel_instances_forced_either_way(exact):
    """LPT_EXTEND_EXACT_STACK (read once per process, hence processes of their own): 1 runs the search kernel instance
    whose LDS stacks hold the scene's bound (20 to 64 entries: not what deep scenes get by default), 0 the guarded
    16-entry instance wherever the bound exceeds 16 (by default only beyond 20).  The chain scenes and a few random
    scenes must match the oracle either way."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LPT_EXTEND_EXACT_STACK=exact)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_fuzz.py"), "-q", "-x", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k",
                        "deep_trees or (random_scene_matches and (hard-1 or hard-2 or hard-3 or hard-4 or hard-5 or hard-6))"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
