"""Thin object wrappers over the native entry points (see _native.py)."""
import ctypes as C

import numpy as np

from . import _native as N

BSDF_HEITZ, BSDF_BASIC = 0, 1
PIPELINE_WAVEFRONT, PIPELINE_MEGAKERNEL, PIPELINE_PERSISTENT, PIPELINE_AUTO, PIPELINE_HYBRID = 0, 1, 2, 3, 4
BUILDER_SPLIT_BVH, BUILDER_BINNED_SAH, BUILDER_LBVH, BUILDER_PLOC = 0, 1, 2, 3
GPU_BUILDERS = (BUILDER_LBVH, BUILDER_PLOC)


class LptError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"[{status}] {message}")
        self.status = status


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def glibc_rand(seed, n):
    """First n values of rand() after srand(seed) on glibc (31-bit integers)."""
    out = np.zeros(n, dtype=np.uint32)
    N.host.lpth_glibc_rand(np.uint32(seed), _ptr(out), n)
    return out


def frame_ubos(camera_world, fov_y, n_frames, seed=0x5EED, first_reset=True):
    """Uniform blocks for n consecutive frames as RendererPT::preDraw produces them:
    two rand() values per frame after srand(seed); the first frame resets accumulation."""
    r = glibc_rand(seed, 2 * n_frames)
    u = np.zeros(n_frames, dtype=N.UBO_DTYPE)
    u["cameraWorldMatrix"][:] = np.asarray(camera_world, dtype=np.float32).reshape(16)
    u["fovY"] = np.float32(fov_y)
    u["seed"][:, 0] = r[0::2]
    u["seed"][:, 1] = r[1::2]
    u["reset"] = 0
    if first_reset and n_frames:
        u["reset"][0] = 1
    return u


def write_png(path, pixels):
    """Writes an (H, W, 3) or (H, W, 4) uint8 array as a PNG file through the host library's encoder."""
    px = np.ascontiguousarray(pixels, dtype=np.uint8)
    if px.ndim != 3 or px.shape[2] not in (3, 4):
        raise ValueError("pixels must be (H, W, 3) or (H, W, 4)")
    if N.host.lpth_write_png(str(path).encode(), _ptr(px), px.shape[1], px.shape[0], px.shape[2]) != 0:
        raise LptError(-1, N.host.lpth_last_error().decode())


class Scene:
    """glTF file -> flattened scene buffers via the native PTSceneConverter."""

    def __init__(self, gltf_path, builder=BUILDER_SPLIT_BVH, device=None):
        h = N.host.lpth_scene_load(str(gltf_path).encode(), int(builder), device._ctx if device else None)
        if not h:
            raise LptError(-1, N.host.lpth_last_error().decode())
        try:
            counts = (C.c_uint32 * 5)()
            N.host.lpth_scene_counts(h, C.byref(counts))

            def grab(fn, dtype, n):
                if n == 0:
                    return np.zeros(0, dtype=dtype)
                buf = (C.c_char * (n * dtype.itemsize)).from_address(fn(h))
                return np.frombuffer(buf, dtype=dtype, count=n).copy()

            self.objects = grab(N.host.lpth_scene_objects, N.OBJECT_DTYPE, counts[0])
            self.object_bvh = grab(N.host.lpth_scene_object_bvh, N.NODE_DTYPE, counts[1])
            self.vertices = grab(N.host.lpth_scene_vertices, N.VERTEX_DTYPE, counts[2])
            self.mesh_bvh = grab(N.host.lpth_scene_mesh_bvh, N.NODE_DTYPE, counts[3])
            # textures in the order the object records index them: dicts with `pixels` (H,W,4) uint8,
            # `mag_filter`, `address_u`, `address_v` (Vulkan enum values, see include/lpt.h)
            self.textures = []
            for i in range(N.host.lpth_scene_texture_count(h)):
                info = (C.c_uint32 * 5)()
                ptr = N.host.lpth_scene_texture(h, i, C.byref(info))
                w, hgt = int(info[0]), int(info[1])
                buf = (C.c_char * (w * hgt * 4)).from_address(ptr)
                px = np.frombuffer(buf, dtype=np.uint8).reshape(hgt, w, 4).copy()
                self.textures.append({"pixels": px, "mag_filter": int(info[2]), "address_u": int(info[3]),
                                      "address_v": int(info[4])})
            self.n_cameras = int(counts[4])
            self.camera_world = None
            self.fov_y = None
            if self.n_cameras:
                m = np.zeros(16, dtype=np.float32)
                f = C.c_float()
                if N.host.lpth_scene_camera(h, 0, _ptr(m), C.byref(f)) != 0:
                    raise LptError(-1, N.host.lpth_last_error().decode())
                self.camera_world, self.fov_y = m, float(np.float32(f.value))
        finally:
            N.host.lpth_scene_free(h)

    @classmethod
    def from_buffers(cls, objects, object_bvh, vertices, mesh_bvh, camera_world=None, fov_y=None, textures=()):
        s = cls.__new__(cls)
        s.textures = list(textures)
        s.objects = np.ascontiguousarray(objects, dtype=N.OBJECT_DTYPE)
        s.object_bvh = np.ascontiguousarray(object_bvh, dtype=N.NODE_DTYPE)
        s.vertices = np.ascontiguousarray(vertices, dtype=N.VERTEX_DTYPE)
        s.mesh_bvh = np.ascontiguousarray(mesh_bvh, dtype=N.NODE_DTYPE)
        s.camera_world, s.fov_y = camera_world, fov_y
        s.n_cameras = 0 if camera_world is None else 1
        return s


class Device:
    """One lpt_ctx (include/lpt.h): one GPU."""

    def __init__(self, device=0):
        ctx = C.c_void_p()
        rc = N.hip.lpt_create(int(device), C.byref(ctx))
        if rc != 0:
            raise LptError(rc, N.hip.lpt_last_error(None).decode())
        self._ctx = ctx
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_ctx", None):
            N.hip.lpt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise LptError(rc, N.hip.lpt_last_error(self._ctx).decode())

    def set_stream(self, hip_stream):
        self._check(N.hip.lpt_set_stream(self._ctx, C.c_void_p(hip_stream)))

    def upload_textures(self, textures):
        """`textures`: sequence of dicts {pixels (H,W,4) uint8, mag_filter, address_u, address_v}."""
        keep = [np.ascontiguousarray(t["pixels"], dtype=np.uint8) for t in textures]
        arr = (N.Texture * max(1, len(keep)))()
        for i, (t, px) in enumerate(zip(textures, keep)):
            if px.ndim != 3 or px.shape[2] != 4:
                raise ValueError("texture pixels must have shape (H, W, 4)")
            arr[i] = N.Texture(px.ctypes.data, px.shape[1], px.shape[0], int(t.get("mag_filter", 1)),
                               int(t.get("address_u", 0)), int(t.get("address_v", 0)))
        self._check(N.hip.lpt_upload_textures(self._ctx, C.cast(arr, C.c_void_p), len(keep)))

    def upload_scene(self, scene):
        self.upload_textures(getattr(scene, "textures", ()))
        self._check(N.hip.lpt_upload_scene(
            self._ctx, _ptr(scene.objects), len(scene.objects), _ptr(scene.object_bvh), len(scene.object_bvh),
            _ptr(scene.vertices), len(scene.vertices), _ptr(scene.mesh_bvh), len(scene.mesh_bvh)))

    def resize(self, width, height):
        self._check(N.hip.lpt_resize(self._ctx, int(width), int(height)))
        self.width, self.height = int(width), int(height)

    def set_partition(self, tile_w, tile_h, rank, n_ranks):
        self._check(N.hip.lpt_set_partition(self._ctx, tile_w, tile_h, rank, n_ranks))

    def set_options(self, max_bounces=10, bsdf_mode=BSDF_HEITZ, pipeline=PIPELINE_AUTO, frames_in_flight=0):
        self._check(N.hip.lpt_set_options(self._ctx, max_bounces, bsdf_mode, pipeline, frames_in_flight))

    def render_frames(self, ubos):
        ubos = np.ascontiguousarray(ubos, dtype=N.UBO_DTYPE)
        self._check(N.hip.lpt_render_frames(self._ctx, _ptr(ubos), len(ubos)))

    def render_frame_static(self, ubo):
        """One frame through the entry point that allocates nothing (see include/lpt.h)."""
        ubo = np.ascontiguousarray(ubo, dtype=N.UBO_DTYPE).reshape(-1)
        if len(ubo) != 1:
            raise ValueError("exactly one uniform block")
        self._check(N.hip.lpt_render_frame_static(self._ctx, _ptr(ubo)))

    def synchronize(self):
        self._check(N.hip.lpt_synchronize(self._ctx))

    def stats(self):
        s = N.Stats()
        self._check(N.hip.lpt_get_stats(self._ctx, C.byref(s)))
        return s

    def read_accum(self):
        img = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(N.hip.lpt_read_accum(self._ctx, _ptr(img)))
        return img

    def resolve(self, inv_sample_count):
        img = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(N.hip.lpt_resolve(self._ctx, C.c_float(inv_sample_count), _ptr(img)))
        return img

    def owned_slots(self):
        n = C.c_uint64()
        self._check(N.hip.lpt_owned_slots(self._ctx, C.byref(n)))
        return int(n.value)

    def owned_accum_device(self):
        p = C.c_void_p()
        self._check(N.hip.lpt_owned_accum_device(self._ctx, C.byref(p)))
        return p.value

    def assemble_gathered(self, gathered_ptr, n_ranks, image_ptr):
        self._check(N.hip.lpt_assemble_gathered(self._ctx, C.c_void_p(gathered_ptr), n_ranks, C.c_void_p(image_ptr)))

    def build_mesh_bvh(self, corners, builder=0):
        """corners: (n,3,3) float32 triangle corners -> (nodes, triangle order).  builder: 0 LBVH, 1 PLOC."""
        corners = np.ascontiguousarray(corners, dtype=np.float32).reshape(-1, 9)
        n = len(corners)
        nodes = np.zeros(max(1, 2 * n), dtype=N.NODE_DTYPE)
        order = np.zeros(max(1, n), dtype=np.uint32)
        cnt = C.c_uint32()
        self._check(N.hip.lpt_build_mesh_bvh(self._ctx, int(builder), _ptr(corners), n, _ptr(nodes), C.byref(cnt), _ptr(order)))
        return nodes[:cnt.value].copy(), order[:n].copy()

    def debug_set_trace_round_limit(self, rounds):
        """Test hook: caps the scheduling rounds of the persistent kernels (0 = default)."""
        self._check(N.hip.lpt_debug_set_trace_round_limit(self._ctx, int(rounds)))

    def debug_math(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.empty_like(a)
        bp = None
        if b is not None:
            b = np.ascontiguousarray(b, dtype=np.float32)
            bp = _ptr(b)
        self._check(N.hip.lpt_debug_math(self._ctx, op, _ptr(a), bp, _ptr(out), a.size))
        return out

    def debug_intersect(self, origins, directions):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = len(o)
        t = np.empty(n, dtype=np.float32)
        obj = np.empty(n, dtype=np.uint32)
        prim = np.empty(n, dtype=np.uint32)
        self._check(N.hip.lpt_debug_intersect(self._ctx, _ptr(o), _ptr(d), n, _ptr(t), _ptr(obj), _ptr(prim)))
        return t, obj, prim


class Renderer:
    """The native RendererPT (host API of the reference) driven from Python."""

    def __init__(self, width, height, render_scale=1.0, device=0):
        self._h = N.host.lpth_renderer_create(width, height, C.c_float(render_scale), device)
        if not self._h:
            raise LptError(-1, N.host.lpth_last_error().decode())
        self.width, self.height = int(np.ceil(width * render_scale)), int(np.ceil(height * render_scale))

    def close(self):
        if getattr(self, "_h", None):
            N.host.lpth_renderer_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise LptError(rc, N.host.lpth_last_error().decode())

    def load_scene(self, gltf_path, builder=BUILDER_SPLIT_BVH):
        self._check(N.host.lpth_renderer_load_gltf(self._h, str(gltf_path).encode(), builder))

    def load_scene_async(self, gltf_path, builder=BUILDER_SPLIT_BVH):
        """Starts RendererPT::loadScene on a second host thread (as src/Main.cpp:45 does) and returns."""
        self._check(N.host.lpth_renderer_load_gltf_async(self._h, str(gltf_path).encode(), builder))

    def wait_for_scene(self):
        self._check(N.host.lpth_renderer_load_wait(self._h))

    @property
    def scene_loaded(self):
        return bool(N.host.lpth_renderer_scene_loaded(self._h))

    def configure(self, seed=0x5EED, max_bounces=10, bsdf_mode=BSDF_HEITZ, pipeline=PIPELINE_AUTO,
                  frames_in_flight=0):
        self._check(N.host.lpth_renderer_configure(self._h, seed, max_bounces, bsdf_mode, pipeline, frames_in_flight))

    def set_deferred_frames(self, n):
        """drawFrame() queues frames and renders them `n` at a time (1 = render inside every call)."""
        self._check(N.host.lpth_renderer_set_deferred_frames(self._h, int(n)))

    def flush(self):
        self._check(N.host.lpth_renderer_flush(self._h))

    def draw_frame(self):
        self._check(N.host.lpth_renderer_draw_frame(self._h))

    def draw_frames(self, n):
        self._check(N.host.lpth_renderer_draw_frames(self._h, n))

    def resize(self, width, height):
        self._check(N.host.lpth_renderer_resize(self._h, width, height))
        self.width, self.height = width, height

    def move_camera(self, axis, amount):
        self._check(N.host.lpth_renderer_move_camera(self._h, axis, C.c_float(amount)))

    @property
    def samples(self):
        return int(N.host.lpth_renderer_samples(self._h))

    def read_accumulation(self):
        img = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(N.host.lpth_renderer_read(self._h, _ptr(img), img.size))
        return img

    def resolve(self):
        img = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(N.host.lpth_renderer_resolve(self._h, _ptr(img), img.size))
        return img
