"""ctypes binding of the CPU oracle (oracle/lpt_oracle.c).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  It never imports the product package; scene buffers are handed over as
plain numpy arrays with the interchange layouts.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "build", "liblpt_oracle.so")
if not os.path.exists(_LIB):
    raise ImportError(f"{_LIB} is missing: run `make -C oracle`")
lib = C.CDLL(_LIB)

BSDF_HEITZ, BSDF_BASIC = 0, 1
K_DIFF, K_METALLIC, K_TRANS = 1, 2, 4


class Scene(C.Structure):
    _fields_ = [("objects", C.c_void_p), ("nObjects", C.c_uint32),
                ("objectBvh", C.c_void_p), ("nObjectBvh", C.c_uint32),
                ("vertices", C.c_void_p), ("nVertices", C.c_uint32),
                ("meshBvh", C.c_void_p), ("nMeshBvh", C.c_uint32),
                ("textures", C.c_void_p), ("nTextures", C.c_uint32)]


class Texture(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("magFilter", C.c_uint32), ("addressU", C.c_uint32), ("addressV", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32),
                ("w", C.c_int32), ("h", C.c_int32), ("maxBounces", C.c_int32), ("bsdfMode", C.c_int32),
                ("threads", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("hits", C.c_uint64), ("boxTests", C.c_uint64),
                ("rootVisits", C.c_uint64), ("triTests", C.c_uint64), ("objectsEntered", C.c_uint64),
                ("walkSteps", C.c_uint64), ("rands", C.c_uint64), ("maxStackTlas", C.c_uint32),
                ("maxStackBlas", C.c_uint32), ("stackOverflow", C.c_uint32), ("maxSegmentSteps", C.c_uint32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "pad"}


class Seed(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32)]


_f3 = C.c_float * 3
_fp = C.POINTER(C.c_float)


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_sig("lpto_render_frame", C.c_int, C.POINTER(Scene), C.c_void_p, C.POINTER(Params), C.c_void_p, C.POINTER(Stats))
_sig("lpto_resolve", None, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p)
_sig("lpto_rand", C.c_float, C.POINTER(Seed))
_sig("lpto_ray_aabb_test", C.c_int, _f3, _f3, _f3, _f3, C.c_float)
_sig("lpto_ray_triangle", C.c_float, _f3, _f3, _f3, _f3, _f3)
_sig("lpto_barycentric", None, _f3, _f3, _f3, _f3, _f3)
_sig("lpto_srgb_to_linear", C.c_float, C.c_float)
_sig("lpto_fresnel", C.c_float, C.c_float, C.c_float)
_sig("lpto_refract_eta", None, _f3, _f3, C.c_float, _f3)
_sig("lpto_interaction_type", C.c_uint32, C.c_float, C.c_float, C.POINTER(Seed))
_sig("lpto_sample_ggx_vndf", None, _f3, C.c_float, C.POINTER(Seed), _f3)
_sig("lpto_sample_ggx_height", C.c_float, _f3, C.c_float, C.c_float, C.POINTER(Seed))
_sig("lpto_bsdf", None, C.c_int, C.c_uint32, _f3, _f3, C.c_float, C.c_float, C.c_float, C.c_int,
     C.POINTER(Seed), _f3, _f3)
_sig("lpto_generate_ray", None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Seed), _f3, _f3)
_sig("lpto_scene_intersect", C.c_int, C.POINTER(Scene), _f3, _f3, _fp, C.POINTER(C.c_uint32),
     C.POINTER(C.c_uint32), C.POINTER(Stats))
_sig("lpto_brute_force_intersect", C.c_int, C.POINTER(Scene), _f3, _f3, _fp, C.POINTER(C.c_uint32),
     C.POINTER(C.c_uint32))
for _n in ("lpto_sin", "lpto_cos", "lpto_tan", "lpto_log", "lpto_exp"):
    _sig(_n, C.c_float, C.c_float)
_sig("lpto_pow", C.c_float, C.c_float, C.c_float)
_sig("lpto_sample_texture", None, C.POINTER(Texture), C.c_float, C.c_float, C.POINTER(C.c_float * 4))


def v3(a):
    return _f3(*[float(x) for x in a])


class OracleScene:
    """Keeps the numpy buffers alive next to the C struct that points into them."""

    def __init__(self, objects, object_bvh, vertices, mesh_bvh, textures=()):
        """`textures`: sequence of dicts {pixels: (H,W,4) uint8, mag_filter, address_u, address_v}."""
        self.tex_pixels = [np.ascontiguousarray(t["pixels"], dtype=np.uint8) for t in textures]
        self.tex_array = (Texture * max(1, len(textures)))()
        for i, t in enumerate(textures):
            px = self.tex_pixels[i]
            assert px.ndim == 3 and px.shape[2] == 4
            self.tex_array[i] = Texture(px.ctypes.data, px.shape[1], px.shape[0], int(t.get("mag_filter", 1)),
                                        int(t.get("address_u", 0)), int(t.get("address_v", 0)))
        self.objects = np.ascontiguousarray(objects)
        self.object_bvh = np.ascontiguousarray(object_bvh)
        self.vertices = np.ascontiguousarray(vertices)
        self.mesh_bvh = np.ascontiguousarray(mesh_bvh)
        assert self.objects.dtype.itemsize == 208 and self.vertices.dtype.itemsize == 48
        assert self.object_bvh.dtype.itemsize == 48 and self.mesh_bvh.dtype.itemsize == 48
        self.c = Scene(self.objects.ctypes.data, len(self.objects), self.object_bvh.ctypes.data, len(self.object_bvh),
                       self.vertices.ctypes.data, len(self.vertices), self.mesh_bvh.ctypes.data, len(self.mesh_bvh),
                       C.cast(self.tex_array, C.c_void_p).value if textures else None, len(textures))


_libm_variant = None


def libm_variant():
    """The same restatement built with the C library's sinf/cosf/tanf/logf/expf/powf in place of the fixed
    polynomials (oracle/Makefile, -DLPTO_LIBM); pass it as `library=` to render_frames."""
    global _libm_variant
    if _libm_variant is None:
        path = os.path.join(_HERE, "build", "liblpt_oracle_libm.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run `make -C oracle`")
        v = C.CDLL(path)
        v.lpto_render_frame.restype = C.c_int
        v.lpto_render_frame.argtypes = [C.POINTER(Scene), C.c_void_p, C.POINTER(Params), C.c_void_p, C.POINTER(Stats)]
        for n in ("lpto_sin", "lpto_cos", "lpto_tan", "lpto_log", "lpto_exp"):
            getattr(v, n).restype = C.c_float
            getattr(v, n).argtypes = [C.c_float]
        v.lpto_pow.restype = C.c_float
        v.lpto_pow.argtypes = [C.c_float, C.c_float]
        _libm_variant = v
    return _libm_variant


def render_frames(scene, ubos, width, height, max_bounces, bsdf_mode=BSDF_HEITZ, tile=None, threads=0, accum=None,
                  library=None):
    """Renders the frames described by `ubos` (records with the 92-byte uniform layout) in order
    into an RGBA32F accumulation image; returns (image, summed stats dict).  `library`: another build of the
    oracle (libm_variant()) instead of the default one."""
    use = library if library is not None else lib
    if accum is None:
        accum = np.zeros((height, width, 4), dtype=np.float32)
    x0, y0, w, h = tile if tile else (0, 0, width, height)
    p = Params(width, height, x0, y0, w, h, max_bounces, bsdf_mode, threads)
    total = {}
    ubos = np.ascontiguousarray(ubos)
    assert ubos.dtype.itemsize == 92
    for i in range(len(ubos)):
        st = Stats()
        rc = use.lpto_render_frame(C.byref(scene.c), ubos[i:i + 1].ctypes.data, C.byref(p), accum.ctypes.data, C.byref(st))
        if rc != 0:
            raise RuntimeError(f"lpto_render_frame failed with {rc}")
        for k, v in st.as_dict().items():
            total[k] = max(total.get(k, 0), v) if k.startswith("max") else total.get(k, 0) + v
    return accum, total


def sample_texture(pixels, u, v, mag_filter=1, address_u=0, address_v=0):
    px = np.ascontiguousarray(pixels, dtype=np.uint8)
    t = Texture(px.ctypes.data, px.shape[1], px.shape[0], mag_filter, address_u, address_v)
    out = (C.c_float * 4)()
    lib.lpto_sample_texture(C.byref(t), C.c_float(u), C.c_float(v), C.byref(out))
    return np.array(out[:], dtype=np.float32)


def resolve(accum, inv_sample_count):
    h, w = accum.shape[:2]
    out = np.empty_like(accum)
    lib.lpto_resolve(np.ascontiguousarray(accum).ctypes.data, w, h, C.c_float(inv_sample_count), out.ctypes.data)
    return out


def scene_intersect(scene, o, d, brute_force=False):
    t = C.c_float()
    obj = C.c_uint32()
    prim = C.c_uint32()
    if brute_force:
        hit = lib.lpto_brute_force_intersect(C.byref(scene.c), v3(o), v3(d), C.byref(t), C.byref(obj), C.byref(prim))
    else:
        hit = lib.lpto_scene_intersect(C.byref(scene.c), v3(o), v3(d), C.byref(t), C.byref(obj), C.byref(prim), None)
    return bool(hit), np.float32(t.value), int(obj.value), int(prim.value)
