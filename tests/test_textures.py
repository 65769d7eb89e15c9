"""Textured materials (the five texture slots of shaders/path_tracing.comp:244-261,277-282).

CPU: the PNG and JPEG readers of the glTF loader against PIL / libjpeg pixels, the oracle's texture fetch against hand-derived
values, the sampler hand-over of the scene converter (including the reference's use of wrappingW for the
V axis).  GPU: parity of a scene that uses every slot, filter and wrap mode with the oracle.  Linear
filtering follows the fixed formula of the arithmetic contract; a Vulkan device's own filtering is not
reproduced bit for bit ("parity unpinned" for textured materials)."""
import os
import shutil

import numpy as np
import pytest

import logipathtracer_amd as L
from conftest import rel_l2
from logipathtracer_amd import scenes
from oracle import binding as O

NEAREST, LINEAR = 0, 1
REPEAT, MIRROR, CLAMP, BORDER = 0, 1, 2, 3


@pytest.fixture(scope="module")
def textured(tmp_path_factory):
    d = tmp_path_factory.mktemp("textured")
    path = str(d / "textured.gltf")
    n = scenes.write_textured_cornell(path)
    return path, n, L.Scene(path)


def test_png_reader_matches_pil_for_all_filters_and_colour_types(textured, tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    path, _, scene = textured
    d = os.path.dirname(path)
    names = ["tex_checker.png", "tex_gradient.png", "tex_glow.png", "tex_mr.png", "tex_trans.png", "tex_normal.png"]
    for name in names:  # file i was written with scan-line filter i % 5; scene.textures is in object order
        ref = np.array(PIL.open(os.path.join(d, name)).convert("RGBA"))
        same = [t for t in scene.textures if t["pixels"].shape == ref.shape and np.array_equal(t["pixels"], ref)]
        assert len(same) == 1, name
    # a palette image with transparency and a 16-bit image, written by PIL
    rng = np.random.default_rng(3)
    pal = PIL.fromarray(rng.integers(0, 256, (9, 7, 3), dtype=np.uint8)).convert("P", palette=PIL.ADAPTIVE, colors=17)
    pal.save(str(tmp_path / "pal.png"), transparency=3)
    g16 = PIL.fromarray((rng.integers(0, 65536, (5, 11)).astype(np.uint16)))
    g16.save(str(tmp_path / "g16.png"))
    for name in ("pal.png", "g16.png"):
        g = scenes._GltfBuilder()
        t = g.texture(name)
        m = g.material("M", base_tex=t)
        quad = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
        up = np.tile([0, 1, 0], (4, 1)).astype(np.float32)
        uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
        g.node("Q", mesh=g.mesh("Q", [(quad, up, np.array([0, 2, 1, 0, 3, 2], np.uint32), m, uv)]))
        g.node("Camera", camera=g.camera(0.5), translation=(0, 3, 0))
        g.write(str(tmp_path / (name + ".gltf")))
        s = L.Scene(str(tmp_path / (name + ".gltf")))
        img = PIL.open(str(tmp_path / name))
        if name == "g16.png":
            ref = (np.array(img).astype(np.uint32) >> 8).astype(np.uint8)  # high byte
            ref = np.stack([ref, ref, ref, np.full_like(ref, 255)], -1)
        else:
            ref = np.array(img.convert("RGBA"))
        assert np.array_equal(s.textures[0]["pixels"], ref), name


def _write_interlaced_png(path, px):
    """8-bit Adam7-interlaced PNG of px (h, w, channels 1..4), scan-line filters 0..4 in rotation."""
    import struct
    import zlib
    h, w, c = px.shape
    colour_type = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    x0, y0 = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1]
    dx, dy = [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
    raw = b""
    for p in range(7):
        sub = px[y0[p]::dy[p], x0[p]::dx[p]]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = np.zeros(sub.shape[1] * c, np.int32)
        for r, row in enumerate(sub):
            line = row.reshape(-1).astype(np.int32)
            ft = (p + r) % 5
            a = np.concatenate([np.zeros(c, np.int32), line[:-c]])
            cc = np.concatenate([np.zeros(c, np.int32), prev[:-c]])
            if ft == 0:
                f = line
            elif ft == 1:
                f = line - a
            elif ft == 2:
                f = line - prev
            elif ft == 3:
                f = line - ((a + prev) >> 1)
            else:
                pp = a + prev - cc
                pa, pb, pc = abs(pp - a), abs(pp - prev), abs(pp - cc)
                f = line - np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, cc))
            raw += bytes([ft]) + (f & 255).astype(np.uint8).tobytes()
            prev = line

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, colour_type, 0, 0, 1)) +
                 chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


@pytest.mark.parametrize("shape", [(13, 21, 3), (1, 1, 4), (5, 3, 1), (9, 8, 2), (2, 30, 3)])
def test_interlaced_png(tmp_path, shape):
    """Adam7: seven reduced images, some of them empty for small sizes, each filtered on its own."""
    px = np.random.default_rng(sum(shape)).integers(0, 256, shape, dtype=np.uint8)
    _write_interlaced_png(str(tmp_path / "i.png"), px)
    got = _one_texture_scene(tmp_path, "i.png").textures[0]["pixels"]
    h, w, c = shape
    ref = np.full((h, w, 4), 255, np.uint8)
    if c <= 2:
        ref[..., :3] = px[..., :1]
        if c == 2:
            ref[..., 3] = px[..., 1]
    else:
        ref[..., :c] = px
    assert np.array_equal(got, ref)


def _one_texture_scene(tmp_path, image_name):
    """A quad whose base colour texture is `image_name` (a file in tmp_path); returns the loaded scene."""
    g = scenes._GltfBuilder()
    m = g.material("M", base_tex=g.texture(image_name))
    quad = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
    g.node("Q", mesh=g.mesh("Q", [(quad, quad * 0 + [0, 1, 0], np.array([0, 2, 1, 0, 3, 2], np.uint32), m)]))
    path = str(tmp_path / (image_name + ".gltf"))
    g.write(path)
    return L.Scene(path)


@pytest.mark.parametrize("shape", [(13, 21, 3), (1, 1, 4), (64, 48, 4), (7, 300, 3)])
def test_png_writer_round_trips_through_the_png_reader(tmp_path, shape):
    """The host library's PNG encoder (what lpt_render writes its display image with) against its own reader
    and, where importable, PIL."""
    from logipathtracer_amd import scenes as _scenes
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    px = rng.integers(0, 256, shape).astype(np.uint8)
    L.write_png(tmp_path / "w.png", px)
    g = _scenes._GltfBuilder()
    quad = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
    m = g.material("M", (1, 1, 1, 1), 0.0, 0.5, base_tex=g.texture("w.png"))
    g.node("Q", mesh=g.mesh("Q", [(quad, np.tile([0, 1, 0], (4, 1)).astype(np.float32), np.array([0, 2, 1, 0, 3, 2], np.uint32), m,
                                   np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32))]))
    g.node("Camera", camera=g.camera(0.8), translation=(0, 3, 0))
    g.write(str(tmp_path / "readback.gltf"))
    back = L.Scene(str(tmp_path / "readback.gltf")).textures[0]["pixels"]
    assert back.shape == shape[:2] + (4,)
    assert np.array_equal(back[..., :shape[2]], px)
    if shape[2] == 3:
        assert np.all(back[..., 3] == 255)
    try:
        from PIL import Image
    except ImportError:
        return
    assert np.array_equal(np.asarray(Image.open(tmp_path / "w.png")), px)
    with pytest.raises(ValueError):
        L.write_png(tmp_path / "bad.png", px[..., :1])


def test_missing_and_damaged_images_are_rejected_with_a_message(tmp_path):
    with pytest.raises(L.LptError, match="failed to open"):
        _one_texture_scene(tmp_path, "photo.jpg")
    open(str(tmp_path / "photo.jpg"), "wb").write(b"\xff\xd8\xff\xe0" + b"\0" * 32)
    with pytest.raises(L.LptError, match="JPEG"):
        _one_texture_scene(tmp_path, "photo.jpg")


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["jpeg_444", "jpeg_422", "jpeg_420_restart", "jpeg_grey", "jpeg_optimized",
                                  "jpeg_progressive", "jpeg_progressive_444", "jpeg_progressive_grey"])
def test_jpeg_reader_against_libjpeg_pixels(tmp_path, name):
    """Baseline and progressive JPEG files written by Pillow (tests/golden/make_jpeg_fixtures.py) and the pixels Pillow decodes
    from them.  JPEG allows a decoder a level or two per sample (inverse DCT, chroma interpolation), so the
    comparison is not bit-exact: every sample within 3 levels, mean difference below half a level."""
    import shutil
    shutil.copy(os.path.join(GOLDEN, name + ".jpg"), str(tmp_path / (name + ".jpg")))
    scene = _one_texture_scene(tmp_path, name + ".jpg")
    got = scene.textures[0]["pixels"].astype(np.int32)
    ref = np.load(os.path.join(GOLDEN, "jpeg_expected.npz"))[name].astype(np.int32)
    assert got.shape == ref.shape
    assert np.array_equal(got[..., 3], ref[..., 3])
    diff = np.abs(got - ref)
    assert diff.max() <= 3, diff.max()
    assert diff.mean() < 0.5, diff.mean()


def test_four_component_jpeg_is_refused_by_name(tmp_path):
    import shutil
    shutil.copy(os.path.join(GOLDEN, "jpeg_cmyk.jpg"), str(tmp_path / "c.jpg"))
    with pytest.raises(L.LptError, match="four-component"):
        _one_texture_scene(tmp_path, "c.jpg")


def test_truncated_jpeg_fails_cleanly(tmp_path):
    for name in ("jpeg_420_restart.jpg", "jpeg_progressive_444.jpg"):
        data = open(os.path.join(GOLDEN, name), "rb").read()
        for cut in (len(data) // 3, len(data) - 40, 30):
            open(str(tmp_path / "t.jpg"), "wb").write(data[:cut])
            with pytest.raises(L.LptError, match="JPEG"):
                _one_texture_scene(tmp_path, "t.jpg")


def test_texture_fetch_known_answers():
    px = np.zeros((2, 3, 4), np.uint8)  # H=2, W=3
    px[0, 0] = [255, 0, 0, 255]
    px[0, 1] = [0, 255, 0, 255]
    px[0, 2] = [0, 0, 255, 255]
    px[1, 0] = [51, 102, 153, 204]
    px[1, 1] = [255, 255, 255, 0]
    px[1, 2] = [10, 20, 30, 40]
    f = np.float32
    # nearest: texel centres and the half-open texel interval
    assert np.array_equal(O.sample_texture(px, 0.5 / 3, 0.25, NEAREST), f([1, 0, 0, 1]))
    assert np.array_equal(O.sample_texture(px, 1.0 / 3 + 1e-4, 0.75, NEAREST), f([1, 1, 1, 0]))
    assert np.array_equal(O.sample_texture(px, 0.99, 0.99, NEAREST), f([10, 20, 30, 40]) / f(255))
    # wrap modes, nearest
    assert np.array_equal(O.sample_texture(px, 1.0 + 0.5 / 3, 0.25, NEAREST, REPEAT), f([1, 0, 0, 1]))
    assert np.array_equal(O.sample_texture(px, -0.5 / 3, 0.25, NEAREST, REPEAT), f([0, 0, 1, 1]))
    assert np.array_equal(O.sample_texture(px, -0.5 / 3, 0.25, NEAREST, MIRROR), f([1, 0, 0, 1]))
    assert np.array_equal(O.sample_texture(px, 1.0 + 0.5 / 3, 0.25, NEAREST, MIRROR), f([0, 0, 1, 1]))
    assert np.array_equal(O.sample_texture(px, 7.3, 0.25, NEAREST, CLAMP), f([0, 0, 1, 1]))
    assert np.array_equal(O.sample_texture(px, -7.3, 0.25, NEAREST, CLAMP), f([1, 0, 0, 1]))
    assert np.array_equal(O.sample_texture(px, 1.2, 0.25, NEAREST, BORDER), f([0, 0, 0, 1]))
    assert np.array_equal(O.sample_texture(px, 0.2, -0.1, NEAREST, REPEAT, BORDER), f([0, 0, 0, 1]))
    # linear: at a texel centre the texel itself, halfway between two centres their mean
    assert np.array_equal(O.sample_texture(px, 0.5, 0.25, LINEAR, CLAMP, CLAMP), f([0, 1, 0, 1]))
    mid = O.sample_texture(px, 1.0 / 3, 0.25, LINEAR, CLAMP, CLAMP)
    assert np.array_equal(mid, f([0.5, 0.5, 0, 1]))
    # the fixed formula, evaluated here in float32 step by step
    u, v = f(0.4), f(0.6)
    x, y = u * f(3) - f(0.5), v * f(2) - f(0.5)
    a, b = x - np.floor(x), y - np.floor(y)
    t = lambda i, j: px[j, i].astype(f) / f(255)
    top = t(0, 0) * (f(1) - a) + t(1, 0) * a
    bot = t(0, 1) * (f(1) - a) + t(1, 1) * a
    want = top * (f(1) - b) + bot * b
    assert np.array_equal(O.sample_texture(px, float(u), float(v), LINEAR), want.astype(f))
    # linear + repeat wraps the second tap around the edge
    edge = O.sample_texture(px, 0.0, 0.25, LINEAR, REPEAT, CLAMP)
    assert np.array_equal(edge, (t(2, 0) * f(0.5) + t(0, 0) * f(0.5)).astype(f))
    # degenerate coordinates stay defined
    assert np.isfinite(O.sample_texture(px, float("nan"), 1e30, LINEAR)).all()
    assert np.isfinite(O.sample_texture(px, -1e30, float("inf"), NEAREST, MIRROR, MIRROR)).all()


def test_converter_hands_over_samplers_like_the_reference(textured):
    _, n, scene = textured
    assert len(scene.vertices) >= 3 * n  # spatial splits may reference a triangle from several leaves
    tex = scene.textures
    assert len(tex) == 6  # one GPU texture per material slot in use, in object-visit order
    by_shape = {t["pixels"].shape[:2]: t for t in tex}
    gradient, glow = by_shape[(32, 48)], by_shape[(16, 16)]
    # magFilter and addressModeU come from the glTF sampler ...
    assert (gradient["mag_filter"], gradient["address_u"]) == (LINEAR, MIRROR)
    assert (glow["mag_filter"], glow["address_u"]) == (LINEAR, CLAMP)
    # ... addressModeV is wrappingW() (src/PTSceneConverter.cpp:359), which glTF never sets: always repeat
    assert all(t["address_v"] == REPEAT for t in tex)
    # a texture without sampler: linear, repeat (src/PTSceneConverter.cpp:371-376)
    names = ("colorTexture", "emissionTexture", "metallicRoughnessTexture", "transmissionTexture", "normalTexture")
    used = {k: [int(i) for i in scene.objects[k] if i != 0xFFFFFFFF] for k in names}
    assert [len(used[k]) for k in names] == [2, 1, 1, 1, 1]
    nrm = tex[used["normalTexture"][0]]
    assert (nrm["mag_filter"], nrm["address_u"], nrm["address_v"]) == (LINEAR, REPEAT, REPEAT)
    assert sorted(i for k in names for i in used[k]) == list(range(6))
    # uvs reach the vertex buffer (floor: 0..3)
    assert float(scene.vertices["uv"].max()) == 4.0 and float(scene.vertices["uv"].min()) == -0.5


def test_oracle_sees_the_textures(textured):
    _, _, scene = textured
    ubos = L.frame_ubos(scene.camera_world, scene.fov_y, 2, seed=5)
    with_tex, st = O.render_frames(O.OracleScene(scene.objects, scene.object_bvh, scene.vertices, scene.mesh_bvh,
                                                 scene.textures), ubos, 64, 48, 5)
    plain = scene.objects.copy()
    for k in ("colorTexture", "emissionTexture", "metallicRoughnessTexture", "transmissionTexture", "normalTexture"):
        plain[k] = 0xFFFFFFFF
    without, _ = O.render_frames(O.OracleScene(plain, scene.object_bvh, scene.vertices, scene.mesh_bvh), ubos, 64, 48, 5)
    assert np.isfinite(with_tex).all() and rel_l2(with_tex, without) > 0.05
    # an object that names a texture that does not exist is refused
    bad = scene.objects.copy()
    bad["colorTexture"][0] = 6
    with pytest.raises(RuntimeError):
        O.render_frames(O.OracleScene(bad, scene.object_bvh, scene.vertices, scene.mesh_bvh, scene.textures), ubos, 8, 8, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("bsdf", [L.BSDF_HEITZ, L.BSDF_BASIC])
def test_textured_scene_matches_oracle(device, textured, bsdf):
    _, _, scene = textured
    w, h, bounces = 256, 192, 8
    ubos = L.frame_ubos(scene.camera_world, scene.fov_y, 3, seed=0xBEEF)
    osc = O.OracleScene(scene.objects, scene.object_bvh, scene.vertices, scene.mesh_bvh, scene.textures)
    ref, st = O.render_frames(osc, ubos, w, h, bounces, bsdf_mode=bsdf)
    device.upload_scene(scene)
    for pipeline in (L.PIPELINE_WAVEFRONT, L.PIPELINE_MEGAKERNEL, L.PIPELINE_PERSISTENT, L.PIPELINE_HYBRID):
        device.resize(w, h)
        device.set_options(bounces, bsdf, pipeline, 0)
        device.render_frames(ubos)
        img = device.read_accum()
        gst = device.stats()
        assert gst.segments == st["segments"]
        mism = int((img[..., :3] != ref[..., :3]).any(axis=-1).sum())
        print(f"pipeline={pipeline} rel_l2={rel_l2(img, ref):.3e} mismatching_pixels={mism}")
        assert rel_l2(img, ref) <= 1e-4  # tolerance on pixel radiance; measured: bit-identical


@pytest.mark.gpu
def test_texture_upload_errors(device, textured, cornell):
    _, _, scene = textured
    device.upload_textures(scene.textures[:2])  # replaces the set and drops the scene
    with pytest.raises(L.LptError, match="texture"):
        device._check(L._native.hip.lpt_upload_scene(
            device._ctx, scene.objects.ctypes.data, len(scene.objects), scene.object_bvh.ctypes.data,
            len(scene.object_bvh), scene.vertices.ctypes.data, len(scene.vertices), scene.mesh_bvh.ctypes.data,
            len(scene.mesh_bvh)))
    with pytest.raises(L.LptError):
        device.upload_textures([{"pixels": np.zeros((4, 4, 4), np.uint8), "mag_filter": 7}])
    with pytest.raises(L.LptError):
        device.upload_textures([{"pixels": np.zeros((2, 2, 4), np.uint8)}] * 513)
    device.upload_scene(cornell)  # an untextured scene still loads afterwards


@pytest.mark.gpu
def test_command_line_renderer_writes_the_same_image(textured, tmp_path, device):
    """lpt_render (the headless stand-in for src/Main.cpp) on the textured scene: its PFM is the accumulation
    image divided by the sample count, its PPM the tone-mapped display image (tex_to_quad.frag)."""
    import subprocess
    path, _, scene = textured
    exe = os.path.join(os.path.dirname(L.__file__), "lib", "lpt_render")
    w, h, spp, bounces = 96, 64, 5, 6
    out = str(tmp_path / "img")
    r = subprocess.run([exe, path, "--width", str(w), "--height", str(h), "--spp", str(spp), "--bounces", str(bounces),
                        "--seed", "0x5EED", "--out", out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    # same frames through the Python binding of the C-ABI
    device.upload_scene(scene)
    device.resize(w, h)
    device.set_options(bounces, L.BSDF_HEITZ, L.PIPELINE_WAVEFRONT, 0)
    device.render_frames(L.frame_ubos(scene.camera_world, scene.fov_y, spp, seed=0x5EED))
    acc = device.read_accum()
    with open(out + ".pfm", "rb") as f:
        assert f.readline() == b"PF\n" and f.readline() == f"{w} {h}\n".encode() and f.readline() == b"-1.0\n"
        pfm = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)
    assert np.array_equal(pfm, acc[..., :3] * np.float32(1.0 / spp))
    with open(out + ".ppm", "rb") as f:
        assert f.readline() == b"P6\n" and f.readline() == f"{w} {h}\n".encode() and f.readline() == b"255\n"
        ppm = np.frombuffer(f.read(), dtype=np.uint8).reshape(h, w, 3)
    disp = np.clip(O.resolve(acc, 1.0 / spp)[..., :3], 0.0, 1.0)
    assert np.abs(ppm.astype(np.int32) - (disp * 255.0 + 0.5).astype(np.int32)).max() <= 1
    # the PNG holds the same bytes as the PPM (read back with the package's own PNG reader, via a one-quad scene)
    from logipathtracer_amd import scenes as _scenes
    g = _scenes._GltfBuilder()
    quad = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
    t = g.texture(os.path.basename(out) + ".png")
    m = g.material("M", (1, 1, 1, 1), 0.0, 0.5, base_tex=t)
    g.node("Q", mesh=g.mesh("Q", [(quad, np.tile([0, 1, 0], (4, 1)).astype(np.float32), np.array([0, 2, 1, 0, 3, 2], np.uint32), m,
                                   np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32))]))
    g.node("Camera", camera=g.camera(0.8), translation=(0, 3, 0))
    g.write(str(tmp_path / "readback.gltf"))
    back = L.Scene(str(tmp_path / "readback.gltf")).textures[0]["pixels"]
    assert back.shape == (h, w, 4) and np.array_equal(back[..., :3], ppm) and np.all(back[..., 3] == 255)


def _jpeg_segments(data):
    """(marker, start, end) of every marker segment up to the first scan; `end` of a scan header segment is where its
    entropy-coded data begins."""
    out, pos = [], 2
    while pos + 4 <= len(data):
        assert data[pos] == 0xFF
        marker = data[pos + 1]
        length = (data[pos + 2] << 8) | data[pos + 3]
        out.append((marker, pos, pos + 2 + length))
        if marker == 0xDA:
            break
        pos += 2 + length
    return out


def test_jpeg_resource_use_is_tied_to_the_file_size(tmp_path):
    """A few hundred bytes must not be able to claim gigabytes or hours: frame headers are held to a pixel budget, a scan
    that runs out of entropy-coded data is cut off instead of decoding zero bits for the rest of the image, and the
    number of scans is bounded."""
    base = open(os.path.join(GOLDEN, "jpeg_444.jpg"), "rb").read()
    sof = [s for s in _jpeg_segments(base) if s[0] == 0xC0][0]

    def with_size(width, height):
        d = bytearray(base)
        d[sof[1] + 5:sof[1] + 9] = bytes([height >> 8, height & 255, width >> 8, width & 255])
        return bytes(d)

    # 32768 x 32768 claimed by a small file
    open(str(tmp_path / "huge.jpg"), "wb").write(with_size(32768, 32768))
    with pytest.raises(L.LptError, match="JPEG.*(64 Mi pixels|more pixels than a file of this size)"):
        _one_texture_scene(tmp_path, "huge.jpg")
    # within the 64 Mi budget but far beyond what the file's bytes could hold
    open(str(tmp_path / "claim.jpg"), "wb").write(with_size(8000, 8000))
    with pytest.raises(L.LptError, match="JPEG.*more pixels than a file of this size"):
        _one_texture_scene(tmp_path, "claim.jpg")
    # a frame four times the real one, inside every budget: the scan runs out of data
    w0 = (base[sof[1] + 7] << 8) | base[sof[1] + 8]
    h0 = (base[sof[1] + 5] << 8) | base[sof[1] + 6]
    open(str(tmp_path / "short.jpg"), "wb").write(with_size(2 * w0, 2 * h0))
    with pytest.raises(L.LptError, match="JPEG.*(ends before the scan does|no Huffman code|out of range|beyond the block)"):
        _one_texture_scene(tmp_path, "short.jpg")
    # a progressive file with 300 more scans, each well-formed and complete (a refinement of the DC bit that is
    # already final: one zero bit per block)
    prog = open(os.path.join(GOLDEN, "jpeg_progressive_grey.jpg"), "rb").read()
    assert prog.endswith(b"\xff\xd9")
    sof2 = [s for s in _jpeg_segments(prog) if s[0] == 0xC2][0]
    ph, pw = (prog[sof2[1] + 5] << 8) | prog[sof2[1] + 6], (prog[sof2[1] + 7] << 8) | prog[sof2[1] + 8]
    comp_id = prog[sof2[1] + 10]
    blocks = ((pw + 7) // 8) * ((ph + 7) // 8)
    scan = b"\xff\xda" + bytes([0, 8, 1, comp_id, 0x00, 0, 0, 0x10]) + b"\0" * ((blocks + 7) // 8)
    open(str(tmp_path / "few.jpg"), "wb").write(prog[:-2] + scan * 3 + b"\xff\xd9")
    assert len(_one_texture_scene(tmp_path, "few.jpg").textures) == 1  # a few such scans are fine
    open(str(tmp_path / "scans.jpg"), "wb").write(prog[:-2] + scan * 300 + b"\xff\xd9")
    with pytest.raises(L.LptError, match="JPEG.*more than 256 scans"):
        _one_texture_scene(tmp_path, "scans.jpg")
    # the untouched files still load
    for name in ("jpeg_444.jpg", "jpeg_progressive_grey.jpg"):
        shutil.copy(os.path.join(GOLDEN, name), str(tmp_path / name))
        assert len(_one_texture_scene(tmp_path, name).textures) == 1
