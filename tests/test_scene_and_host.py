"""Host logic: glTF loading, PTSceneConverter buffers, BVH builders, frame seeds, pixel partition."""
import ctypes as C
import ctypes.util

import numpy as np
import pytest

import logipathtracer_amd as L
from conftest import CORNELL
from logipathtracer_amd.partition import Partition
from oracle import binding as O


@pytest.fixture(scope="module", params=[L.BUILDER_SPLIT_BVH, L.BUILDER_BINNED_SAH], ids=["split_bvh", "binned_sah"])
def built(request):
    s = L.Scene(CORNELL, builder=request.param)
    return request.param, s, O.OracleScene(s.objects, s.object_bvh, s.vertices, s.mesh_bvh)


def test_cornell_inventory(built):
    builder, s, _ = built
    # resources/cornell_box.gltf: 11 mesh nodes with one primitive each, 12 956 triangles, one camera
    assert len(s.objects) == 11 and s.n_cameras == 1
    if builder == L.BUILDER_BINNED_SAH:
        assert len(s.vertices) == 3 * 12956
    else:
        assert 3 * 12956 <= len(s.vertices) <= int(3 * 12956 * 1.5) + 48  # spatial splits duplicate references
    m = s.camera_world.reshape(4, 4)
    assert np.allclose(m[3, :3], [6, 1, 0]) and np.allclose(m[0, :3], [0, 0, -1], atol=1e-6)
    assert np.allclose(m[1, :3], [0, 1, 0], atol=1e-6) and np.allclose(m[2, :3], [1, 0, 0], atol=1e-6)
    assert abs(s.fov_y - 0.39959652) < 1e-7
    assert np.all(s.objects["colorTexture"] == 0xFFFFFFFF) and np.all(s.objects["normalTexture"] == 0xFFFFFFFF)
    # materials: one emissive object (5,5,5); two transmissive (ior 1.8 and 1.5); one metallic wall
    em = s.objects["emissionFactor"].sum(axis=1)
    assert (em > 0).sum() == 1 and em.max() == 15.0
    assert sorted(s.objects["ior"][s.objects["transmissionFactor"] == 1.0].tolist()) == [1.5, np.float32(1.8)]
    assert (s.objects["metallicFactor"] == 1.0).sum() == 2  # green wall + glTF-default material of the light
    # inverse really is the inverse
    for o in s.objects:
        w = o["worldMatrix"].reshape(4, 4).T.astype(np.float64)
        wi = o["worldMatrixInverse"].reshape(4, 4).T.astype(np.float64)
        assert np.allclose(w @ wi, np.eye(4), atol=1e-5)


def _check_tree(nodes, base, n_prims_expected=None):
    """Every leaf range is visited once, children lie inside their parent, ranges tile [0, n)."""
    ranges = []
    stack = [0]
    depth = {0: 1}
    maxd = 1
    while stack:
        i = stack.pop()
        n = nodes[base + i]
        if n["isLeaf"]:
            ranges.append((int(n["indices"][0]), int(n["indices"][1])))
            continue
        for c in n["indices"]:
            ch = nodes[base + int(c)]
            assert np.all(ch["minCorner"] >= n["minCorner"] - 1e-6) and np.all(ch["maxCorner"] <= n["maxCorner"] + 1e-6)
            depth[int(c)] = depth[i] + 1
            maxd = max(maxd, depth[int(c)])
            stack.append(int(c))
    ranges.sort()
    assert ranges[0][0] == 0
    for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
        assert a1 == b0 and a1 > a0
    if n_prims_expected is not None:
        assert ranges[-1][1] == n_prims_expected
    return ranges, maxd


def test_bvh_invariants(built):
    _, s, _ = built
    _check_tree(s.object_bvh, 0, len(s.objects))
    order = np.argsort(s.objects["verticesOffset"])
    ends = list(s.objects["verticesOffset"][order][1:]) + [len(s.vertices)]
    for oi, end in zip(order, ends):
        o = s.objects[oi]
        ntri = (int(end) - int(o["verticesOffset"])) // 3
        ranges, depth = _check_tree(s.mesh_bvh, int(o["bvhOffset"]), ntri)
        assert depth <= 48
        # every triangle lies inside the box of its leaf
        stack = [0]
        while stack:
            i = stack.pop()
            n = s.mesh_bvh[int(o["bvhOffset"]) + i]
            if n["isLeaf"]:
                v = s.vertices["position"][int(o["verticesOffset"]) + 3 * int(n["indices"][0]):
                                           int(o["verticesOffset"]) + 3 * int(n["indices"][1])]
                # with spatial splits a leaf box only has to cover the part of the triangle inside it
                assert np.all(v.max(axis=0) >= n["minCorner"] - 1e-6) and np.all(v.min(axis=0) <= n["maxCorner"] + 1e-6)
            else:
                stack += [int(c) for c in n["indices"]]


def test_bvh_traversal_equals_brute_force(built):
    _, s, osc = built
    rng = np.random.default_rng(42)
    n = 3000
    o = rng.uniform([-0.95, 0.05, -0.95], [0.95, 1.9, 0.95], size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # camera-like rays too
    o[: n // 3] = [6, 1, 0]
    d[: n // 3] = (np.array([-1, 0, 0]) + rng.uniform(-0.2, 0.2, size=(n // 3, 3))).astype(np.float32)
    mismatches = 0
    for i in range(n):
        h1, t1, obj1, prim1 = O.scene_intersect(osc, o[i], d[i])
        h2, t2, obj2, prim2 = O.scene_intersect(osc, o[i], d[i], brute_force=True)
        if h1 != h2 or t1 != t2:
            mismatches += 1
        elif h1 and obj1 != obj2:
            # equal distance on two objects is a tie; which one wins depends on visit order
            mismatches += 0
    # the slab test is not conservative (rounded reciprocal), so a ray grazing a box corner may skip a
    # triangle the brute-force loop finds; that must stay very rare
    assert mismatches <= 2, mismatches


def test_converter_rejects_dangling_textures_and_missing_files(tmp_path):
    import json
    g = json.load(open(CORNELL))
    g["materials"][0]["pbrMetallicRoughness"]["baseColorTexture"] = {"index": 0}
    p = tmp_path / "tex.gltf"
    p.write_text(json.dumps(g))
    (tmp_path / "cornell_box.bin").write_bytes(open(CORNELL.replace(".gltf", ".bin"), "rb").read())
    with pytest.raises(L.LptError, match="texture index out of range"):  # the file declares no textures
        L.Scene(str(p))
    with pytest.raises(L.LptError, match="failed to open"):
        L.Scene(str(tmp_path / "nope.gltf"))
    (tmp_path / "bad.gltf").write_text("{ not json")
    with pytest.raises(L.LptError, match="JSON"):
        L.Scene(str(tmp_path / "bad.gltf"))


def test_malformed_files_are_refused_not_trusted(tmp_path):
    """Offsets, counts and nesting come from the file: values that would wrap a range check or exhaust the
    stack must end in an error message (found by mutating the Cornell file under AddressSanitizer)."""
    import copy
    import json
    base = json.load(open(CORNELL))
    (tmp_path / "cornell_box.bin").write_bytes(open(CORNELL.replace(".gltf", ".bin"), "rb").read())

    def load(doc=None, raw=None):
        p = tmp_path / "m.gltf"
        p.write_text(raw if raw is not None else json.dumps(doc))
        return L.Scene(str(p))

    pos = base["meshes"][0]["primitives"][0]["attributes"]["POSITION"]
    view = base["accessors"][pos]["bufferView"]
    for field, where, value in (("byteOffset", "view", -1), ("byteOffset", "acc", -8), ("count", "acc", 2**31 - 1),
                                ("count", "acc", 1e30), ("count", "acc", -3), ("byteStride", "view", 2**40),
                                ("byteStride", "view", 1), ("byteOffset", "view", 2**62)):
        g = copy.deepcopy(base)
        (g["bufferViews"][view] if where == "view" else g["accessors"][pos])[field] = value
        with pytest.raises(L.LptError, match="accessor exceeds buffer|byteStride out of range"):
            load(g)
    for raw in ("[" * 100000, '{"a":' * 100000):
        with pytest.raises(L.LptError, match="nesting too deep"):
            load(raw=raw)
    g = copy.deepcopy(base)
    first = len(g["nodes"])
    g["nodes"] += [{"children": [first + i + 1]} for i in range(20000)] + [{}]
    g["scenes"][0]["nodes"].append(first)
    with pytest.raises(L.LptError, match="deeper than 512 levels"):
        load(g)
    g = copy.deepcopy(base)
    g["nodes"][0]["children"] = [0]
    with pytest.raises(L.LptError, match="not a tree"):
        load(g)
    assert len(load(base).objects) == 11  # the unmodified document still loads from the same place


def test_frame_seeds_follow_glibc_rand():
    libc = C.CDLL(ctypes.util.find_library("c"))
    libc.rand.restype = C.c_int
    for seed in (1, 0x5EED, 123456789, 0):
        libc.srand(C.c_uint(seed))
        expect = [libc.rand() for _ in range(40)]
        assert L.glibc_rand(seed, 40).tolist() == expect
    u = L.frame_ubos(np.eye(4, dtype=np.float32).reshape(16), 0.5, 3, seed=0x5EED)
    r = L.glibc_rand(0x5EED, 6)
    assert u["seed"].reshape(-1).tolist() == r.tolist()  # x then y, frame by frame (src/RendererPT.cpp:584-585)
    assert u["reset"].tolist() == [1, 0, 0]


@pytest.mark.parametrize("w,h,tw,th,ranks", [(1920, 1080, 64, 64, 8), (256, 256, 64, 64, 2), (100, 52, 16, 8, 3),
                                               (1920, 1080, None, None, 1)])
def test_partition_maps_are_bijective(w, h, tw, th, ranks):
    P = Partition(w, h, tw, th, ranks)
    ys, xs = np.mgrid[0:h, 0:w]
    owner, slot = P.pixel_to_slot(xs, ys)
    assert owner.min() >= 0 and owner.max() < ranks and slot.max() < P.n_slots
    key = owner.astype(np.int64) * P.n_slots + slot
    assert len(np.unique(key)) == w * h
    # 64 consecutive slots are one 8x8 pixel block
    img = np.stack([xs, ys], axis=-1)
    buf = P.pack(img, 0)
    blk = buf[:64]
    assert blk[:, 0].max() - blk[:, 0].min() == 7 and blk[:, 1].max() - blk[:, 1].min() == 7
    # pack on every rank + assemble is the identity
    gathered = np.concatenate([P.pack(img, r) for r in range(ranks)])
    assert np.array_equal(P.assemble(gathered), img)
    # owned_tiles covers every pixel exactly once
    cover = np.zeros((h, w), dtype=np.int32)
    for r in range(ranks):
        for (x0, y0, tw_, th_) in P.owned_tiles(r):
            cover[y0:y0 + th_, x0:x0 + tw_] += 1
            assert np.all(owner[y0:y0 + th_, x0:x0 + tw_] == r)
    assert np.all(cover == 1)


def test_builder_never_emits_leaves_beyond_the_device_limit(tmp_path, monkeypatch):
    """150 triangles with one centroid and a requested leaf size of 200: no split plane separates them, so
    the builder halves the list — and it must go on halving until every leaf fits the 6-bit count of the
    device layer's traversal references (63), whatever leaf size was asked for."""
    from logipathtracer_amd import scenes
    g = scenes._GltfBuilder()
    m = g.material("M", (0.7, 0.6, 0.5, 1), 0.3, 0.4)
    tri = np.tile(np.array([[-1, -1, 0], [1, -1, 0], [0, 2, 0]], np.float32), (150, 1))
    nrm = np.tile([0, 0, 1], (450, 1)).astype(np.float32)
    g.node("Pile", mesh=g.mesh("Pile", [(tri, nrm, np.arange(450, dtype=np.uint32), m)]))
    g.node("Camera", camera=g.camera(0.8), translation=(0, 0, 4))
    path = str(tmp_path / "pile.gltf")
    g.write(path)
    monkeypatch.setenv("LPT_BVH_CHILD_ORDER", "0")
    monkeypatch.setenv("LPT_BVH_LEAF", "200")
    for builder in (L.BUILDER_SPLIT_BVH, L.BUILDER_BINNED_SAH):
        s = L.Scene(path, builder=builder)
        leaves = s.mesh_bvh[s.mesh_bvh["isLeaf"] != 0]
        counts = leaves["indices"][:, 1].astype(np.int64) - leaves["indices"][:, 0]
        assert counts.sum() == 150 and counts.max() <= 63 and len(leaves) >= 3
        _check_tree(s.mesh_bvh, 0, 150)


def _pack_glb(json_bytes, bin_bytes):
    """Binary glTF container: header, JSON chunk (padded with spaces), BIN chunk (padded with zeros)."""
    import struct
    js = json_bytes + b" " * (-len(json_bytes) % 4)
    chunks = struct.pack("<II", len(js), 0x4E4F534A) + js
    if bin_bytes is not None:
        bn = bin_bytes + b"\0" * (-len(bin_bytes) % 4)
        chunks += struct.pack("<II", len(bn), 0x004E4942) + bn
    return b"glTF" + struct.pack("<II", 2, 12 + len(chunks)) + chunks


def test_binary_gltf_container_loads_like_the_text_form(tmp_path):
    import json
    import os
    doc = json.load(open(CORNELL))
    assert len(doc["buffers"]) == 1
    bin_bytes = open(os.path.join(os.path.dirname(CORNELL), doc["buffers"][0]["uri"]), "rb").read()
    del doc["buffers"][0]["uri"]
    open(str(tmp_path / "cornell.glb"), "wb").write(_pack_glb(json.dumps(doc).encode(), bin_bytes))
    a, b = L.Scene(CORNELL), L.Scene(str(tmp_path / "cornell.glb"))
    for name in ("objects", "object_bvh", "vertices", "mesh_bvh"):
        assert getattr(a, name).tobytes() == getattr(b, name).tobytes(), name
    assert np.array_equal(a.camera_world, b.camera_world) and a.fov_y == b.fov_y
    # damaged containers fail with a message
    good = open(str(tmp_path / "cornell.glb"), "rb").read()
    open(str(tmp_path / "cut.glb"), "wb").write(good[:len(good) // 2])
    with pytest.raises(L.LptError, match="glTF"):
        L.Scene(str(tmp_path / "cut.glb"))
    open(str(tmp_path / "nobin.glb"), "wb").write(_pack_glb(json.dumps(doc).encode(), None))
    with pytest.raises(L.LptError, match="glTF"):
        L.Scene(str(tmp_path / "nobin.glb"))


def test_normalised_integer_texture_coordinates(tmp_path):
    """TEXCOORD_0 as normalised unsigned shorts (core glTF allows them) loads to the same floats."""
    import json
    import os
    from logipathtracer_amd import scenes
    scenes.write_textured_cornell(str(tmp_path / "t.gltf"))
    doc = json.load(open(str(tmp_path / "t.gltf")))
    bin_path = os.path.join(str(tmp_path), doc["buffers"][0]["uri"])
    data = bytearray(open(bin_path, "rb").read())
    changed = 0
    for mesh in doc["meshes"]:
        for prim in mesh["primitives"]:
            a = prim["attributes"].get("TEXCOORD_0")
            if a is None or doc["accessors"][a]["componentType"] != 5126:
                continue
            acc = doc["accessors"][a]
            bv = doc["bufferViews"][acc["bufferView"]]
            off = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
            uv = np.frombuffer(bytes(data[off:off + acc["count"] * 8]), dtype=np.float32).reshape(-1, 2)
            if uv.min() < 0 or uv.max() > 1:
                continue
            q = np.round(uv * 65535).astype(np.uint16)
            assert np.array_equal(q.astype(np.float32) / np.float32(65535), uv)  # 0 and 1 only: exact
            data += b"\0" * (-len(data) % 4)
            doc["bufferViews"].append({"buffer": 0, "byteOffset": len(data), "byteLength": q.nbytes})
            data += q.tobytes()
            acc.update({"bufferView": len(doc["bufferViews"]) - 1, "byteOffset": 0, "componentType": 5123, "normalized": True})
            changed += 1
    assert changed
    reference = L.Scene(str(tmp_path / "t.gltf"))
    doc["buffers"][0]["byteLength"] = len(data)
    open(bin_path, "wb").write(data)
    json.dump(doc, open(str(tmp_path / "q.gltf"), "w"))
    quantised = L.Scene(str(tmp_path / "q.gltf"))
    assert reference.vertices.tobytes() == quantised.vertices.tobytes()
