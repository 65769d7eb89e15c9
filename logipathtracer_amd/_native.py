"""ctypes bindings to the native libraries.

`liblpt_hip.so` is the C-ABI device layer (include/lpt.h); `liblpt_host.so` holds the C++
host classes (RendererPT, PTSceneConverter, the glTF loader) behind flat C entry points.
There is no Python fallback: if a library is missing, importing this module fails.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")


def _load(name):
    path = os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build the native libraries first (`make` at the repository root, "
            "or __graft_entry__.build())")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


hip = _load("liblpt_hip.so")
host = _load("liblpt_host.so")

# ---- interchange structures (include/lpt.h) as numpy record types ------------------------

OBJECT_DTYPE = np.dtype([
    ("worldMatrix", "<f4", (16,)), ("worldMatrixInverse", "<f4", (16,)),
    ("baseColorFactor", "<f4", (4,)), ("emissionFactor", "<f4", (3,)),
    ("metallicFactor", "<f4"), ("roughnessFactor", "<f4"), ("transmissionFactor", "<f4"), ("ior", "<f4"),
    ("colorTexture", "<u4"), ("emissionTexture", "<u4"), ("metallicRoughnessTexture", "<u4"),
    ("transmissionTexture", "<u4"), ("normalTexture", "<u4"),
    ("bvhOffset", "<u4"), ("verticesOffset", "<u4"), ("padding", "u1", (8,)),
])
VERTEX_DTYPE = np.dtype([
    ("position", "<f4", (3,)), ("pad0", "<f4"), ("normal", "<f4", (3,)), ("pad1", "<f4"),
    ("uv", "<f4", (2,)), ("pad2", "<f4", (2,)),
])
NODE_DTYPE = np.dtype([
    ("minCorner", "<f4", (3,)), ("pad0", "<f4"), ("maxCorner", "<f4", (3,)), ("isLeaf", "<u4"),
    ("indices", "<u4", (2,)), ("pad1", "<u4", (2,)),
])
UBO_DTYPE = np.dtype([
    ("cameraWorldMatrix", "<f4", (16,)), ("fovY", "<f4"), ("pad", "<f4", (3,)),
    ("seed", "<u4", (2,)), ("reset", "<u4"),
])
assert OBJECT_DTYPE.itemsize == 208 and VERTEX_DTYPE.itemsize == 48
assert NODE_DTYPE.itemsize == 48 and UBO_DTYPE.itemsize == 92


class Texture(C.Structure):
    """lpt_texture of include/lpt.h."""
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("mag_filter", C.c_uint32), ("address_u", C.c_uint32), ("address_v", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("frames", C.c_uint64),
                ("device_ms", C.c_double), ("extend_ms", C.c_double),
                ("extend_launches", C.c_uint32), ("pipeline", C.c_uint32), ("kernel_segments", C.c_uint64),
                ("timed_bounce_lo", C.c_uint32), ("timed_bounce_hi", C.c_uint32),
                ("stack_need", C.c_uint32), ("stack_entries", C.c_uint32)]


# every symbol include/lpt.h declares
LPT_SYMBOLS = [
    "lpt_create", "lpt_destroy", "lpt_last_error", "lpt_set_stream", "lpt_upload_textures", "lpt_upload_scene", "lpt_resize",
    "lpt_set_partition", "lpt_set_options", "lpt_render_frames", "lpt_render_frame", "lpt_synchronize",
    "lpt_get_stats", "lpt_read_accum", "lpt_owned_slots", "lpt_owned_accum_device", "lpt_assemble_gathered",
    "lpt_resolve", "lpt_build_mesh_bvh", "lpt_debug_math", "lpt_debug_intersect", "lpt_debug_set_trace_round_limit",
    "lpt_render_frame_static",
]

_vp, _u32, _i = C.c_void_p, C.c_uint32, C.c_int


def _sig(lib, name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_sig(hip, "lpt_create", _i, _i, C.POINTER(_vp))
_sig(hip, "lpt_destroy", None, _vp)
_sig(hip, "lpt_last_error", C.c_char_p, _vp)
_sig(hip, "lpt_set_stream", _i, _vp, _vp)
_sig(hip, "lpt_upload_textures", _i, _vp, _vp, _u32)
_sig(hip, "lpt_upload_scene", _i, _vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp, _u32)
_sig(hip, "lpt_resize", _i, _vp, _i, _i)
_sig(hip, "lpt_set_partition", _i, _vp, _i, _i, _i, _i)
_sig(hip, "lpt_set_options", _i, _vp, _i, _i, _i, _i)
_sig(hip, "lpt_render_frames", _i, _vp, _vp, _i)
_sig(hip, "lpt_render_frame", _i, _vp, _vp)
_sig(hip, "lpt_render_frame_static", _i, _vp, _vp)
_sig(hip, "lpt_synchronize", _i, _vp)
_sig(hip, "lpt_debug_set_trace_round_limit", _i, _vp, _u32)
_sig(hip, "lpt_get_stats", _i, _vp, C.POINTER(Stats))
_sig(hip, "lpt_read_accum", _i, _vp, _vp)
_sig(hip, "lpt_owned_slots", _i, _vp, C.POINTER(C.c_uint64))
_sig(hip, "lpt_owned_accum_device", _i, _vp, C.POINTER(_vp))
_sig(hip, "lpt_assemble_gathered", _i, _vp, _vp, _i, _vp)
_sig(hip, "lpt_resolve", _i, _vp, C.c_float, _vp)
_sig(hip, "lpt_build_mesh_bvh", _i, _vp, _i, _vp, _u32, _vp, C.POINTER(_u32), _vp)
_sig(hip, "lpt_debug_math", _i, _vp, _i, _vp, _vp, _vp, _u32)
_sig(hip, "lpt_debug_intersect", _i, _vp, _vp, _vp, _u32, _vp, _vp, _vp)

_sig(host, "lpth_last_error", C.c_char_p)
_sig(host, "lpth_write_png", _i, C.c_char_p, _vp, _u32, _u32, _i)
_sig(host, "lpth_scene_load", _vp, C.c_char_p, _i, _vp)
_sig(host, "lpth_scene_free", None, _vp)
_sig(host, "lpth_scene_counts", None, _vp, C.POINTER(_u32 * 5))
for _n in ("lpth_scene_objects", "lpth_scene_object_bvh", "lpth_scene_vertices", "lpth_scene_mesh_bvh"):
    _sig(host, _n, _vp, _vp)
_sig(host, "lpth_scene_camera", _i, _vp, _u32, _vp, C.POINTER(C.c_float))
_sig(host, "lpth_scene_texture_count", _u32, _vp)
_sig(host, "lpth_scene_texture", _vp, _vp, _u32, C.POINTER(_u32 * 5))
_sig(host, "lpth_renderer_create", _vp, _i, _i, C.c_float, _i)
_sig(host, "lpth_renderer_destroy", None, _vp)
_sig(host, "lpth_renderer_load_gltf", _i, _vp, C.c_char_p, _i)
_sig(host, "lpth_renderer_load_gltf_async", _i, _vp, C.c_char_p, _i)
_sig(host, "lpth_renderer_load_wait", _i, _vp)
_sig(host, "lpth_renderer_scene_loaded", _i, _vp)
_sig(host, "lpth_renderer_set_deferred_frames", _i, _vp, _u32)
_sig(host, "lpth_renderer_flush", _i, _vp)
_sig(host, "lpth_renderer_configure", _i, _vp, _u32, _i, _i, _i, _i)
_sig(host, "lpth_renderer_draw_frame", _i, _vp)
_sig(host, "lpth_renderer_draw_frames", _i, _vp, _u32)
_sig(host, "lpth_renderer_resize", _i, _vp, _i, _i)
_sig(host, "lpth_renderer_move_camera", _i, _vp, _i, C.c_float)
_sig(host, "lpth_renderer_samples", _u32, _vp)
_sig(host, "lpth_renderer_device", _vp, _vp)
_sig(host, "lpth_renderer_read", _i, _vp, _vp, C.c_size_t)
_sig(host, "lpth_renderer_resolve", _i, _vp, _vp, C.c_size_t)
_sig(host, "lpth_glibc_rand", None, _u32, _vp, _u32)
