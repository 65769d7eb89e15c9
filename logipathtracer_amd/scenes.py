"""Procedural glTF 2.0 scenes for tests and benchmarks (written as .gltf + .bin).

`write_outside_synth` stands in for the reference's `resources/outside.gltf`, whose 17.8 MB binary buffer is
not part of the repository snapshot: it reproduces the shape the JSON describes — an open scene seen from
a camera near the ground, about 70 primitives and about 400 k triangles dominated by a few large meshes, an
emissive sky shell (0.2, 0.2, 0.18), one water surface (transmission 0.98, ior 1.33, roughness 0), two
partly metallic materials, the rest diffuse with roughness 0.5-1 — from fixed-seed noise, so every run
produces the same bytes.

`write_cornell_like` is a Cornell-box-shaped scene with tunable tessellation for scaling studies.

`write_textured_cornell` is the same box with every one of the five texture slots of the material model in
use (base colour, emission, metallic-roughness, transmission, normal map), both filters and all three glTF
wrap modes; its PNG images are written next to the .gltf by `write_png`.
"""
import json
import os
import struct
import zlib

import numpy as np

_SEED = 0xC0FFEE


def write_png(path, pixels, filter_type=0):
    """Writes an (H,W), (H,W,2), (H,W,3) or (H,W,4) uint8 array as an 8-bit PNG (grey, grey+alpha, RGB, RGBA).
    `filter_type` 0..4 applies that PNG scan-line filter to every row (all are lossless)."""
    px = np.ascontiguousarray(pixels, dtype=np.uint8)
    if px.ndim == 2:
        px = px[:, :, None]
    h, w, ch = px.shape
    color_type = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    rows = px.reshape(h, w * ch).astype(np.int32)
    out = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        cur = rows[y]
        a = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        c = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        b = prev
        if filter_type == 0:
            f = cur
        elif filter_type == 1:
            f = cur - a
        elif filter_type == 2:
            f = cur - b
        elif filter_type == 3:
            f = cur - ((a + b) >> 1)
        else:
            pa, pb, pc = np.abs(b - c), np.abs(a - c), np.abs(a + b - 2 * c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
            f = cur - pred
        out.append(filter_type)
        out += (f & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n")
        fh.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)))
        fh.write(chunk(b"IDAT", zlib.compress(bytes(out), 6)))
        fh.write(chunk(b"IEND", b""))


class _GltfBuilder:
    def __init__(self):
        self.bin = bytearray()
        self.views, self.accessors, self.meshes, self.nodes, self.materials, self.cameras = [], [], [], [], [], []
        self.images, self.samplers, self.textures = [], [], []
        self.scene_nodes = []

    def texture(self, uri, mag=9729, wrap_s=10497, wrap_t=10497, sampler=True):
        """Adds image `uri` (a file next to the .gltf) and a texture using it; returns the texture index.
        glTF constants: filter 9728 nearest / 9729 linear; wrap 10497 repeat, 33648 mirrored, 33071 clamp."""
        self.images.append({"uri": uri})
        t = {"source": len(self.images) - 1}
        if sampler:
            self.samplers.append({"magFilter": mag, "minFilter": mag, "wrapS": wrap_s, "wrapT": wrap_t})
            t["sampler"] = len(self.samplers) - 1
        self.textures.append(t)
        return len(self.textures) - 1

    def _view(self, data, target=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        v = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
        if target:
            v["target"] = target
        self.bin += data
        self.views.append(v)
        return len(self.views) - 1

    def _accessor(self, arr, ctype, gtype, target=None, minmax=False):
        a = {"bufferView": self._view(arr.tobytes(), target), "componentType": ctype, "count": int(len(arr)), "type": gtype}
        if minmax:
            a["min"] = [float(x) for x in arr.min(axis=0)]
            a["max"] = [float(x) for x in arr.max(axis=0)]
        self.accessors.append(a)
        return len(self.accessors) - 1

    def material(self, name, base=(1, 1, 1, 1), metallic=0.0, roughness=1.0, emissive=None, transmission=None, ior=None,
                 base_tex=None, emissive_tex=None, mr_tex=None, transmission_tex=None, normal_tex=None):
        m = {"name": name, "doubleSided": True,
             "pbrMetallicRoughness": {"baseColorFactor": [float(x) for x in base], "metallicFactor": float(metallic),
                                      "roughnessFactor": float(roughness)}}
        if emissive is not None:
            m["emissiveFactor"] = [float(x) for x in emissive]
        if transmission is not None:
            ext = {"transmissionFactor": float(transmission), "ior": float(ior)}
            if transmission_tex is not None:
                ext["transmissionTexture"] = {"index": transmission_tex}
            m["extensions"] = {"ADOBE_materials_thin_transparency": ext}
        if base_tex is not None:
            m["pbrMetallicRoughness"]["baseColorTexture"] = {"index": base_tex}
        if mr_tex is not None:
            m["pbrMetallicRoughness"]["metallicRoughnessTexture"] = {"index": mr_tex}
        if emissive_tex is not None:
            m["emissiveTexture"] = {"index": emissive_tex}
        if normal_tex is not None:
            m["normalTexture"] = {"index": normal_tex}
        self.materials.append(m)
        return len(self.materials) - 1

    def mesh(self, name, primitives):
        """primitives: list of (positions (n,3), normals (n,3), indices (m,), material[, uvs (n,2)])"""
        prims = []
        for prim in primitives:
            pos, nrm, idx, mat = prim[:4]
            pos = np.ascontiguousarray(pos, dtype=np.float32)
            nrm = np.ascontiguousarray(nrm, dtype=np.float32)
            idx = np.ascontiguousarray(idx, dtype=np.uint32)
            attrs = {"POSITION": self._accessor(pos, 5126, "VEC3", 34962, True),
                     "NORMAL": self._accessor(nrm, 5126, "VEC3", 34962)}
            if len(prim) > 4 and prim[4] is not None:
                attrs["TEXCOORD_0"] = self._accessor(np.ascontiguousarray(prim[4], dtype=np.float32), 5126, "VEC2", 34962)
            prims.append({"attributes": attrs, "indices": self._accessor(idx, 5125, "SCALAR", 34963), "material": mat})
        self.meshes.append({"name": name, "primitives": prims})
        return len(self.meshes) - 1

    def node(self, name, mesh=None, camera=None, translation=None, rotation=None, scale=None, root=True):
        n = {"name": name}
        if mesh is not None:
            n["mesh"] = mesh
        if camera is not None:
            n["camera"] = camera
        if translation is not None:
            n["translation"] = [float(x) for x in translation]
        if rotation is not None:
            n["rotation"] = [float(x) for x in rotation]
        if scale is not None:
            n["scale"] = [float(x) for x in scale]
        self.nodes.append(n)
        if root:
            self.scene_nodes.append(len(self.nodes) - 1)
        return len(self.nodes) - 1

    def camera(self, yfov, znear=0.1, zfar=1000.0):
        self.cameras.append({"type": "perspective", "perspective": {"yfov": float(yfov), "znear": znear, "zfar": zfar}})
        return len(self.cameras) - 1

    def write(self, path):
        base = os.path.splitext(os.path.basename(path))[0]
        doc = {"asset": {"version": "2.0", "generator": "logipathtracer_amd.scenes"}, "scene": 0,
               "scenes": [{"name": "Scene", "nodes": self.scene_nodes}], "nodes": self.nodes, "cameras": self.cameras,
               "materials": self.materials, "meshes": self.meshes, "accessors": self.accessors,
               "bufferViews": self.views, "buffers": [{"byteLength": len(self.bin), "uri": base + ".bin"}],
               "extensionsUsed": ["ADOBE_materials_thin_transparency"]}
        if self.textures:
            doc["images"], doc["textures"] = self.images, self.textures
            if self.samplers:
                doc["samplers"] = self.samplers
        with open(path, "w") as f:
            json.dump(doc, f)
        with open(os.path.join(os.path.dirname(path), base + ".bin"), "wb") as f:
            f.write(self.bin)


def _icosphere(level):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        edges = {}
        verts = list(v)

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in edges:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                edges[key] = len(verts) - 1
            return edges[key]

        nf = []
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(verts), np.array(nf, dtype=np.int64)
    return v.astype(np.float32), f.reshape(-1).astype(np.uint32)


def _box():
    p, n, idx = [], [], []
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            u, w = (axis + 1) % 3, (axis + 2) % 3
            base = len(p)
            for du, dw in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
                c = [0.0, 0.0, 0.0]
                c[axis], c[u], c[w] = sgn, du, dw
                p.append(c)
                nn = [0.0, 0.0, 0.0]
                nn[axis] = sgn
                n.append(nn)
            idx += [base, base + 1, base + 2, base, base + 2, base + 3] if sgn > 0 else [base, base + 2, base + 1, base, base + 3, base + 2]
    return np.array(p, np.float32), np.array(n, np.float32), np.array(idx, np.uint32)


def _value_noise(x, z, rng_table, freq):
    xi, zi = np.floor(x * freq).astype(int), np.floor(z * freq).astype(int)
    xf, zf = x * freq - xi, z * freq - zi
    n = rng_table.shape[0]

    def h(a, b):
        return rng_table[a % n, b % n]

    sx, sz = xf * xf * (3 - 2 * xf), zf * zf * (3 - 2 * zf)
    top = h(xi, zi) * (1 - sx) + h(xi + 1, zi) * sx
    bot = h(xi, zi + 1) * (1 - sx) + h(xi + 1, zi + 1) * sx
    return top * (1 - sz) + bot * sz


def _terrain(n, size, rng):
    table = rng.uniform(-1, 1, size=(64, 64))
    xs = np.linspace(-size, size, n + 1)
    x, z = np.meshgrid(xs, xs, indexing="xy")
    y = (1.6 * _value_noise(x, z, table, 0.05) + 0.6 * _value_noise(x, z, table, 0.17) + 0.15 * _value_noise(x, z, table, 0.6))
    pos = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    gy_x = np.gradient(y, xs, axis=1)
    gy_z = np.gradient(y, xs, axis=0)
    nrm = np.stack([-gy_x, np.ones_like(y), -gy_z], axis=-1).reshape(-1, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="xy")
    a = (j * (n + 1) + i).reshape(-1)
    idx = np.stack([a, a + n + 1, a + 1, a + 1, a + n + 1, a + n + 2], axis=1).reshape(-1)
    return pos.astype(np.float32), nrm.astype(np.float32), idx.astype(np.uint32), (xs, y)


def write_outside_synth(path, detail=1.0):
    """Writes `path` (.gltf) and its .bin.  detail=1.0 gives about 400 k triangles; smaller values scale the
    tessellation down for quick tests.  Returns the triangle count."""
    rng = np.random.default_rng(_SEED)
    g = _GltfBuilder()
    ground = g.material("Ground", (0.35, 0.42, 0.22, 1), 0.0, 0.9)
    rock = g.material("Rock", (0.45, 0.43, 0.40, 1), 0.0, 0.7)
    leaf = g.material("Leaf", (0.18, 0.45, 0.12, 1), 0.0, 0.5)
    bark = g.material("Bark", (0.30, 0.20, 0.12, 1), 0.0, 1.0)
    metal_a = g.material("MetalA", (0.80, 0.78, 0.70, 1), 0.7, 0.3)
    metal_b = g.material("MetalB", (0.60, 0.35, 0.25, 1), 0.5, 0.4)
    water = g.material("Water", (0.80, 0.90, 0.95, 1), 0.0, 0.0, transmission=0.98, ior=1.33)
    sky = g.material("Sky", (0, 0, 0, 1), 0.0, 1.0, emissive=(0.2, 0.2, 0.18))
    sun = g.material("Sun", (0, 0, 0, 1), 0.0, 1.0, emissive=(18.0, 16.0, 13.0))
    tris = 0

    n_ground = max(8, int(230 * detail ** 0.5))
    pos, nrm, idx, (xs, height) = _terrain(n_ground, 40.0, rng)
    g.node("Ground", mesh=g.mesh("Ground", [(pos, nrm, idx, ground)]))
    tris += len(idx) // 3

    def ground_y(x, z):
        i = np.clip(np.searchsorted(xs, x), 0, len(xs) - 1)
        j = np.clip(np.searchsorted(xs, z), 0, len(xs) - 1)
        return float(height[j, i])

    wp = np.array([[-40, -0.55, -40], [40, -0.55, -40], [40, -0.55, 40], [-40, -0.55, 40]], np.float32)
    g.node("Water", mesh=g.mesh("Water", [(wp, np.tile([0, 1, 0], (4, 1)).astype(np.float32), np.array([0, 2, 1, 0, 3, 2], np.uint32), water)]))
    tris += 2

    sv, sf = _icosphere(3)
    g.node("Sky", mesh=g.mesh("Sky", [(sv, -sv, sf.reshape(-1, 3)[:, ::-1].reshape(-1), sky)]), scale=(300, 300, 300))
    g.node("Sun", mesh=g.mesh("Sun", [(sv, sv, sf, sun)]), translation=(120, 160, -60), scale=(14, 14, 14))
    tris += 2 * (len(sf) // 3)

    # a few large meshes carry most of the triangles (the reference scene's largest primitives have 47-58 k)
    big_level = 5 if detail >= 0.9 else (4 if detail >= 0.2 else 3)
    bv, bf = _icosphere(big_level)
    mats = [rock, rock, metal_a, rock, metal_b, rock, rock]
    for k in range(7):
        bump = 1.0 + 0.18 * np.sin(7.0 * bv[:, 0] + k) * np.cos(5.0 * bv[:, 1] - k) + 0.08 * rng.normal(size=len(bv))
        p = (bv * bump[:, None]).astype(np.float32)
        # smooth normals from the displaced faces
        f3 = bf.reshape(-1, 3)
        fn = np.cross(p[f3[:, 1]] - p[f3[:, 0]], p[f3[:, 2]] - p[f3[:, 0]])
        vn = np.zeros_like(p)
        for c in range(3):
            np.add.at(vn, f3[:, c], fn)
        vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-12)
        m = g.mesh(f"Boulder{k}", [(p, vn.astype(np.float32), bf, mats[k])])
        x, z = rng.uniform(-22, 22, 2)
        s = rng.uniform(1.5, 3.2)
        g.node(f"Boulder{k}", mesh=m, translation=(x, ground_y(x, z) + 0.6 * s, z), scale=(s, 0.8 * s, s),
               rotation=_quat(rng))
        tris += len(bf) // 3

    # trees: trunk box + crown with two primitives (leaf + bark detail), instanced over the terrain
    cv, cf = _icosphere(4 if detail >= 0.9 else (3 if detail >= 0.2 else 2))
    bp, bn, bi = _box()
    crown = g.mesh("Crown", [(cv, cv, cf, leaf), ((cv * 0.55 + np.array([0, -0.9, 0], np.float32)), cv, cf, bark)])
    trunk = g.mesh("Trunk", [(bp, bn, bi, bark)])
    n_trees = 14
    for k in range(n_trees):
        x, z = rng.uniform(-30, 30, 2)
        h = rng.uniform(2.0, 4.5)
        y = ground_y(x, z)
        g.node(f"Trunk{k}", mesh=trunk, translation=(x, y + h / 2, z), scale=(0.18, h / 2, 0.18))
        g.node(f"Crown{k}", mesh=crown, translation=(x, y + h + 0.8, z), scale=(1.4, 1.7, 1.4), rotation=_quat(rng))
        tris += 2 * (len(cf) // 3) + 12

    # small rocks and crates
    rv, rf = _icosphere(2)
    rockmesh = g.mesh("Pebble", [(rv, rv, rf, rock)])
    crate = g.mesh("Crate", [(bp, bn, bi, metal_b)])
    for k in range(16):
        x, z = rng.uniform(-25, 25, 2)
        s = rng.uniform(0.2, 0.7)
        g.node(f"Pebble{k}", mesh=rockmesh, translation=(x, ground_y(x, z) + 0.5 * s, z), scale=(s, 0.7 * s, s), rotation=_quat(rng))
        tris += len(rf) // 3
    for k in range(4):
        x, z = rng.uniform(-12, 12, 2)
        g.node(f"Crate{k}", mesh=crate, translation=(x, ground_y(x, z) + 0.5, z), scale=(0.5, 0.5, 0.5), rotation=_quat(rng, y_only=True))
        tris += 12

    # cameras: node 0 pose of the reference's outside.gltf (translation and yfov from its JSON)
    cam0 = g.camera(0.7696)
    cam1 = g.camera(0.6)
    yaw = np.deg2rad(-105.0)
    g.node("Camera", camera=cam0, translation=(-15.62, max(4.68, ground_y(-15.62, -4.30) + 2.0), -4.30),
           rotation=(0.0, float(np.sin(yaw / 2)), 0.0, float(np.cos(yaw / 2))))
    g.node("Camera2", camera=cam1, translation=(10.0, 6.0, 18.0), rotation=(0.0, 0.0, 0.0, 1.0))
    g.write(path)
    return tris


def _quat(rng, y_only=False):
    if y_only:
        a = rng.uniform(0, 2 * np.pi)
        return (0.0, float(np.sin(a / 2)), 0.0, float(np.cos(a / 2)))
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return tuple(float(x) for x in q)


def write_cornell_like(path, level=4):
    """Cornell-box-shaped scene: five walls, emissive ceiling panel, a glass sphere, a diffuse blob and a
    metallic cuboid.  `level` is the icosphere subdivision of the two round objects (20*4^level triangles each)."""
    g = _GltfBuilder()
    white = g.material("White", (0.8, 0.8, 0.8, 1), 0.0, 0.11)
    green = g.material("Green", (0.03, 0.79, 0.03, 1), 1.0, 0.1)
    red = g.material("Red", (0.79, 0.03, 0.03, 1), 0.0, 0.1)
    glass = g.material("Glass", (1, 1, 1, 1), 0.0, 0.2, transmission=1.0, ior=1.8)
    light = g.material("Light", (1, 1, 1, 1), 1.0, 1.0, emissive=(5, 5, 5))
    quad = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
    up = np.tile([0, 1, 0], (4, 1)).astype(np.float32)
    qi = np.array([0, 2, 1, 0, 3, 2], np.uint32)
    plane = {m: g.mesh(f"Plane{m}", [(quad, up, qi, m)]) for m in (white, green, red, light)}
    s = float(np.sqrt(0.5))
    g.node("Floor", mesh=plane[white], translation=(0, 0, 0))
    g.node("Ceiling", mesh=plane[white], translation=(0, 2, 0), rotation=(1, 0, 0, 0))
    g.node("Back", mesh=plane[white], translation=(-1, 1, 0), rotation=(0, 0, -s, s))
    g.node("Left", mesh=plane[green], translation=(0, 1, 1), rotation=(-s, 0, 0, s))
    g.node("Right", mesh=plane[red], translation=(0, 1, -1), rotation=(s, 0, 0, s))
    g.node("Light", mesh=plane[light], translation=(0, 1.98, 0), rotation=(1, 0, 0, 0), scale=(0.5, 1, 0.5))
    sv, sf = _icosphere(level)
    g.node("Sphere", mesh=g.mesh("Sphere", [(sv, sv, sf, glass)]), translation=(-0.45, 0.35, 0), scale=(0.35, 0.35, 0.35))
    blob = sv * (1.0 + 0.25 * np.sin(6 * sv[:, :1]) * np.cos(5 * sv[:, 1:2]))
    g.node("Blob", mesh=g.mesh("Blob", [(blob.astype(np.float32), sv, sf, white)]), translation=(0.2, 0.4, -0.6), scale=(0.4, 0.4, 0.4))
    bp, bn, bi = _box()
    g.node("Cuboid", mesh=g.mesh("Cuboid", [(bp, bn, bi, green)]), translation=(0.05, 0.4, 0.55), scale=(0.2, 0.4, 0.2),
           rotation=(0, 0.3826834, 0, 0.9238795))
    cam = g.camera(0.39959652)
    g.node("Camera", camera=cam, translation=(6, 1, 0), rotation=(0, s, 0, s))
    g.write(path)
    return 12 + 2 * (len(sf) // 3) + 12


def _texture_images(rng):
    """The six images of write_textured_cornell as uint8 arrays, keyed by file name."""
    n = 64
    yy, xx = np.mgrid[0:n, 0:n]
    checker = np.where(((xx // 8 + yy // 8) % 2)[..., None] == 0, [230, 220, 200], [60, 90, 160]).astype(np.uint8)
    gradient = np.zeros((32, 48, 4), np.uint8)  # not square, with alpha
    gradient[..., 0] = np.linspace(40, 255, 48)[None, :]
    gradient[..., 1] = np.linspace(255, 30, 32)[:, None]
    gradient[..., 2] = rng.integers(0, 256, (32, 48))
    gradient[..., 3] = 255
    glow = np.zeros((16, 16, 3), np.uint8)
    glow[4:12, 4:12] = [255, 240, 200]
    glow[0:4] = [120, 20, 20]
    mr = np.zeros((n, n, 3), np.uint8)  # g = roughness scale, b = metallic scale
    mr[..., 1] = (40 + 200 * ((xx // 16) % 2)).astype(np.uint8)
    mr[..., 2] = (255 * ((yy // 16) % 2)).astype(np.uint8)
    trans = (255 * (((xx // 4) + (yy // 4)) % 2)).astype(np.uint8)  # greyscale PNG
    ang = 2 * np.pi * xx / 16.0
    nx, ny = 0.45 * np.sin(ang), 0.45 * np.cos(2 * np.pi * yy / 32.0)
    nz = np.sqrt(np.maximum(0.0, 1.0 - nx * nx - ny * ny))
    normal = np.stack([nx, ny, nz], -1)
    normal = np.round((normal * 0.5 + 0.5) * 255).astype(np.uint8)
    return {"tex_checker.png": checker, "tex_gradient.png": gradient, "tex_glow.png": glow, "tex_mr.png": mr,
            "tex_trans.png": trans, "tex_normal.png": normal}


def write_textured_cornell(path):
    """Cornell-like box in which all five texture slots are used: checker base colour on the floor (nearest,
    repeat, uv range 0..3), a linearly filtered RGBA gradient on the back wall (mirrored repeat / clamp,
    uv range -0.5..1.5), an emission texture on the light, a metallic-roughness texture on the cuboid, a
    greyscale transmission texture on a glass pane, and a normal map on the right wall (texture without
    sampler).  Image files are written next to `path`.  Returns the number of triangles."""
    rng = np.random.default_rng(_SEED)
    d = os.path.dirname(path)
    for i, (name, px) in enumerate(_texture_images(rng).items()):
        write_png(os.path.join(d, name), px, filter_type=i % 5)
    g = _GltfBuilder()
    t_checker = g.texture("tex_checker.png", mag=9728, wrap_s=10497, wrap_t=10497)
    t_gradient = g.texture("tex_gradient.png", mag=9729, wrap_s=33648, wrap_t=33071)
    t_glow = g.texture("tex_glow.png", mag=9729, wrap_s=33071, wrap_t=33071)
    t_mr = g.texture("tex_mr.png", mag=9728, wrap_s=10497, wrap_t=33648)
    t_trans = g.texture("tex_trans.png", mag=9729)
    t_normal = g.texture("tex_normal.png", sampler=False)
    floor = g.material("Floor", (1, 1, 1, 1), 0.0, 0.4, base_tex=t_checker)
    back = g.material("Back", (0.9, 0.9, 0.9, 1), 0.0, 0.3, base_tex=t_gradient)
    white = g.material("White", (0.8, 0.8, 0.8, 1), 0.0, 0.11)
    green = g.material("Green", (0.03, 0.79, 0.03, 1), 0.0, 0.3)
    bumpy = g.material("Bumpy", (0.79, 0.2, 0.1, 1), 0.5, 0.3, normal_tex=t_normal)
    metal = g.material("Metal", (0.9, 0.8, 0.5, 1), 1.0, 1.0, mr_tex=t_mr)
    pane = g.material("Pane", (1, 1, 1, 1), 0.0, 0.15, transmission=1.0, ior=1.5, transmission_tex=t_trans)
    light = g.material("Light", (1, 1, 1, 1), 0.0, 1.0, emissive=(6, 6, 6), emissive_tex=t_glow)
    quad = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
    up = np.tile([0, 1, 0], (4, 1)).astype(np.float32)
    qi = np.array([0, 2, 1, 0, 3, 2], np.uint32)
    unit = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)

    def plane(name, mat, uv=None):
        return g.mesh(name, [(quad, up, qi, mat, uv)])

    s = float(np.sqrt(0.5))
    g.node("Floor", mesh=plane("FloorMesh", floor, unit * 3.0), translation=(0, 0, 0))
    g.node("Ceiling", mesh=plane("CeilingMesh", white), translation=(0, 2, 0), rotation=(1, 0, 0, 0))
    g.node("Back", mesh=plane("BackMesh", back, unit * 2.0 - 0.5), translation=(-1, 1, 0), rotation=(0, 0, -s, s))
    g.node("Left", mesh=plane("LeftMesh", green), translation=(0, 1, 1), rotation=(-s, 0, 0, s))
    g.node("Right", mesh=plane("RightMesh", bumpy, unit * 4.0), translation=(0, 1, -1), rotation=(s, 0, 0, s))
    g.node("Light", mesh=plane("LightMesh", light, unit), translation=(0, 1.98, 0), rotation=(1, 0, 0, 0), scale=(0.5, 1, 0.5))
    g.node("Pane", mesh=plane("PaneMesh", pane, unit * 1.5), translation=(0.3, 0.6, -0.3), rotation=(0, 0, -s, s),
           scale=(0.45, 1, 0.45))
    bp, bn, bi = _box()
    buv = np.tile(unit, (len(bp) // 4, 1)).astype(np.float32) if len(bp) % 4 == 0 else np.zeros((len(bp), 2), np.float32)
    g.node("Cuboid", mesh=g.mesh("Cuboid", [(bp, bn, bi, metal, buv)]), translation=(0.05, 0.4, 0.55), scale=(0.25, 0.4, 0.25),
           rotation=(0, 0.3826834, 0, 0.9238795))
    sv, sf = _icosphere(2)
    g.node("Ball", mesh=g.mesh("Ball", [(sv, sv, sf, white)]), translation=(-0.4, 0.3, -0.2), scale=(0.3, 0.3, 0.3))
    cam = g.camera(0.39959652)
    g.node("Camera", camera=cam, translation=(6, 1, 0), rotation=(0, s, 0, s))
    g.write(path)
    return 14 + len(bi) // 3 + len(sf) // 3


def _look_at_quat(eye, target):
    """Rotation (x, y, z, w) of a glTF camera node at `eye` that looks at `target` (cameras look down -z, +y up)."""
    f = np.asarray(target, np.float64) - np.asarray(eye, np.float64)
    f /= max(np.linalg.norm(f), 1e-12)
    up = np.array([0.0, 1.0, 0.0]) if abs(f[1]) < 0.99 else np.array([1.0, 0.0, 0.0])
    r = np.cross(f, up)
    r /= np.linalg.norm(r)
    u = np.cross(r, f)
    m = np.stack([r, u, -f], axis=1)  # columns: camera x, y, z axes in the world
    w = np.sqrt(max(0.0, 1.0 + m[0, 0] + m[1, 1] + m[2, 2])) / 2.0
    if w > 1e-6:
        q = [(m[2, 1] - m[1, 2]) / (4 * w), (m[0, 2] - m[2, 0]) / (4 * w), (m[1, 0] - m[0, 1]) / (4 * w), w]
    else:  # half-turn: take the axis from the largest diagonal entry
        k = int(np.argmax([m[0, 0], m[1, 1], m[2, 2]]))
        a = np.zeros(3)
        a[k] = np.sqrt(max(0.0, (m[k, k] + 1.0) / 2.0))
        for j in range(3):
            if j != k:
                a[j] = m[k, j] / (2.0 * a[k]) if a[k] else 0.0
        q = [a[0], a[1], a[2], 0.0]
    return tuple(float(x) for x in q)


def write_random_scene(path, seed, hard=False):
    """A scene drawn from `seed`, for parity fuzzing: 3-12 objects (boxes, spheres, quads, triangle soups with
    zero-area and sliver triangles) under random rotations and non-uniform, now and then mirrored scales, materials
    over the whole parameter range (roughness 0 and 1, metallic 0 / 1 / in between, transmission with ior below and
    above 1, emitters), normals that need not fit the geometry, sometimes a closed emissive shell around everything,
    and a camera that may sit inside an object.  `hard` adds what produces infinities and NaNs: zero-length normals,
    scales of exactly zero along one axis (singular world matrices) and a camera on a coordinate axis looking along it
    (ray directions with zero components).  Returns the number of triangles."""
    rng = np.random.default_rng(0x51DE0000 + int(seed) + (0x10000 if hard else 0))
    g = _GltfBuilder()

    def rand_material(k):
        base = tuple(float(x) for x in rng.uniform(0.02, 1.0, 3)) + (1.0,)
        metallic = float(rng.choice([0.0, 1.0, rng.uniform(0, 1)]))
        roughness = float(rng.choice([0.0, 0.001, 1.0, rng.uniform(0, 1), rng.uniform(0, 0.2)]))
        emissive = tuple(float(x) for x in rng.uniform(0, 6, 3)) if rng.random() < 0.2 else None
        if rng.random() < 0.3:
            return g.material(f"M{k}", base, metallic, roughness, emissive=emissive,
                              transmission=float(rng.choice([1.0, 0.98, rng.uniform(0, 1)])),
                              ior=float(rng.choice([1.0, 1.33, 1.5, 1.8, 0.8])))
        return g.material(f"M{k}", base, metallic, roughness, emissive=emissive)

    def soup(n):
        c = rng.uniform(-1, 1, (n, 1, 3))
        p = c + rng.normal(scale=rng.choice([0.05, 0.3, 1.0]), size=(n, 3, 3))
        kind = rng.random(n)
        p[kind < 0.05, 1] = p[kind < 0.05, 0]                       # two equal corners
        p[(kind >= 0.05) & (kind < 0.08)] = c[(kind >= 0.05) & (kind < 0.08)]  # a point
        sl = (kind >= 0.08) & (kind < 0.12)                          # slivers: third corner on the first edge
        p[sl, 2] = 0.5 * (p[sl, 0] + p[sl, 1])
        return p.reshape(-1, 3).astype(np.float32)

    tris = 0
    n_obj = int(rng.integers(3, 13))
    for k in range(n_obj):
        mat = rand_material(k)
        shape = rng.choice(["box", "sphere", "quad", "soup", "soup"])
        if shape == "box":
            pos, nrm, idx = _box()
        elif shape == "sphere":
            pos, idx = _icosphere(int(rng.integers(0, 3)))
            nrm = pos.copy()
        elif shape == "quad":
            pos = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
            nrm = np.tile([0, 1, 0], (4, 1)).astype(np.float32)
            idx = np.array([0, 2, 1, 0, 3, 2], np.uint32)
        else:
            pos = soup(int(rng.integers(4, 200)))
            e1, e2 = pos[1::3] - pos[0::3], pos[2::3] - pos[0::3]
            fn = np.cross(e1, e2)
            ln = np.linalg.norm(fn, axis=1, keepdims=True)
            fn = np.where(ln > 1e-20, fn / np.maximum(ln, 1e-20), [[0.0, 1.0, 0.0]])
            nrm = np.repeat(fn, 3, axis=0).astype(np.float32)
            idx = np.arange(len(pos), dtype=np.uint32)
        style = rng.random()
        if style < 0.25:    # smooth-looking normals that do not fit the faces
            nrm = rng.normal(size=nrm.shape)
            nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
        elif style < 0.35:  # not unit length
            nrm = (nrm * rng.uniform(0.1, 5.0, (len(nrm), 1))).astype(np.float32)
        scale = rng.uniform(0.15, 2.0, 3) * rng.choice([1.0, 1.0, 1.0, -1.0], 3)
        if rng.random() < 0.15:
            scale[int(rng.integers(0, 3))] *= 0.01  # nearly flat
        if hard and rng.random() < 0.2:
            nrm = nrm.copy()
            nrm[rng.random(len(nrm)) < 0.3] = 0.0
        if hard and rng.random() < 0.15:
            scale[int(rng.integers(0, 3))] = 0.0
        g.node(f"Obj{k}", mesh=g.mesh(f"Mesh{k}", [(pos, nrm, idx, mat)]), translation=rng.uniform(-2.5, 2.5, 3),
               rotation=_quat(rng), scale=scale)
        tris += len(idx) // 3
    if rng.random() < 0.5:  # closed emissive shell: paths run to their full depth
        sv, sf = _icosphere(1)
        shell = g.material("Shell", (0.5, 0.5, 0.5, 1), 0.0, 0.8, emissive=tuple(float(x) for x in rng.uniform(0.1, 1.0, 3))
                           if rng.random() < 0.5 else None)
        g.node("Shell", mesh=g.mesh("Shell", [(sv, -sv, sf.reshape(-1, 3)[:, ::-1].reshape(-1), shell)]), scale=(9, 9, 9))
        tris += len(sf) // 3
    eye = rng.uniform(-1, 1, 3) * rng.choice([0.5, 4.0, 6.0])
    target = rng.uniform(-1, 1, 3)
    if hard and rng.random() < 0.5:
        axis = int(rng.integers(0, 3))
        eye = np.zeros(3)
        eye[axis] = float(rng.choice([-5.0, 5.0]))
        target = np.zeros(3)
    g.node("Camera", camera=g.camera(float(rng.uniform(0.3, 1.4))), translation=eye, rotation=_look_at_quat(eye, target))
    g.write(path)
    return tris
