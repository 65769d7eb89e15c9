/*
 * lpt.h — C-ABI of the MI355X device layer for the RendererPT compute path.
 *
 * The reference has no FFI layer; its path tracer is a Vulkan compute shader
 * bound through descriptor set 0 (shaders/path_tracing.comp:81-105) and driven
 * by RendererPT (src/RendererPT.cpp).  Each entry point below replaces one of
 * those bindings / host steps and cites it.  All functions return 0 on success
 * and a negative lpt_status otherwise; no C++ exception crosses this boundary.
 * One context serves one GPU; contexts are independent; calls on one context
 * must not overlap.
 */
#ifndef LPT_H
#define LPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- interchange structures (byte-exact with the reference) ---------------- */

/* GPUObjectData, include/PTSceneConverter.hpp:15-42; struct Object,
 * shaders/path_tracing.comp:41-57.  208 bytes.  Matrices are column-major.
 * bvhOffset / verticesOffset are ELEMENT indices (src/PTSceneConverter.cpp:87-88).
 * A texture slot holding 0xFFFFFFFF means "no texture". */
typedef struct lpt_object {
  float worldMatrix[16];
  float worldMatrixInverse[16];
  float baseColorFactor[4];
  float emissionFactor[3];
  float metallicFactor;
  float roughnessFactor;
  float transmissionFactor;
  float ior;
  uint32_t colorTexture;
  uint32_t emissionTexture;
  uint32_t metallicRoughnessTexture;
  uint32_t transmissionTexture;
  uint32_t normalTexture;
  uint32_t bvhOffset;
  uint32_t verticesOffset;
  uint8_t padding[8];
} lpt_object;

/* GPUVertex, include/PTSceneConverter.hpp:44-50; struct Vertex,
 * shaders/path_tracing.comp:59-63.  48 bytes. */
typedef struct lpt_vertex {
  float position[3];
  float pad0;
  float normal[3];
  float pad1;
  float uv[2];
  float pad2[2];
} lpt_vertex;

/* GPUBVHNode, include/PTSceneConverter.hpp:52-59; struct BVHNode,
 * shaders/path_tracing.comp:29-39.  48 bytes.  Inner node: indices = child node
 * indices (relative to the object's bvhOffset for mesh trees, absolute for the
 * object tree).  Leaf: indices = [first, last) primitive range. */
typedef struct lpt_bvh_node {
  float minCorner[3];
  float pad0;
  float maxCorner[3];
  uint32_t isLeaf;
  uint32_t indices[2];
  uint32_t pad1[2];
} lpt_bvh_node;

/* PathTracerUBO, include/RendererPT.h:51-61; UBO block,
 * shaders/path_tracing.comp:83-87.  92 bytes on the host. */
typedef struct lpt_ubo {
  float cameraWorldMatrix[16]; /* camera-to-world, column-major */
  float fovY;                  /* radians */
  float pad[3];
  uint32_t seed[2];            /* ubo_.random, src/RendererPT.cpp:584-585 */
  uint32_t reset;              /* VkBool32 */
} lpt_ubo;

typedef enum lpt_status {
  LPT_OK = 0,
  LPT_ERR_INVALID = -1,     /* bad argument / call order */
  LPT_ERR_UNSUPPORTED = -2, /* e.g. textured materials, BVH leaves too large to encode */
  LPT_ERR_STACK = -3,       /* BVH deeper than the traversal stack variants */
  LPT_ERR_HIP = -4,         /* a HIP runtime call failed; see lpt_last_error */
  LPT_ERR_NOMEM = -5,
  LPT_ERR_TRUNCATED = -6    /* a persistent launch ran out of scheduling rounds with paths still in flight:
                               the image of the last render is incomplete.  Reported by the next call that
                               waits for the stream (lpt_synchronize, lpt_get_stats, lpt_read_accum,
                               lpt_resolve, lpt_set_stream, the next lpt_render_frames), once. */
} lpt_status;

/* #define USE_MICROFACET, shaders/path_tracing.comp:22 */
enum { LPT_BSDF_HEITZ = 0, LPT_BSDF_BASIC = 1 };
/* Kernel organisation; all produce identical pixels.
 * AUTO (default): HYBRID; WAVEFRONT for batches large enough to amortise a launch pair per
 *            bounce — judged by the path count and what the last finished call of the same scene,
 *            image size and options took per path (nine 1080p frames of a Cornell box, four of a
 *            400 k-triangle scene with rough materials; before there is such a call: batches of
 *            18 Mi paths and more); PERSISTENT when the ray queues do not fit in device memory.
 * WAVEFRONT: one kernel per stage and bounce over ray queues in device memory (128 bytes of
 *            queues per path in flight); the first segment runs in the packet kernel, and from the
 *            bounce at which the queues have become short (by the last batch's fills) one persistent
 *            launch finishes the paths.
 * MEGAKERNEL: one thread per pixel runs the whole path (the shader's own organisation).
 * PERSISTENT: one launch per batch, paths stay in registers from the camera ray on (16 bytes
 *            per path in flight).
 * HYBRID:    the first segment of every path in the packet kernel (camera ray, hit, shading), then
 *            one persistent launch finishes the paths that go on (64 bytes per path in flight). */
enum {
  LPT_PIPELINE_WAVEFRONT = 0,
  LPT_PIPELINE_MEGAKERNEL = 1,
  LPT_PIPELINE_PERSISTENT = 2,
  LPT_PIPELINE_AUTO = 3,
  LPT_PIPELINE_HYBRID = 4
};
/* Mesh BVH builders offered by lpt_build_mesh_bvh. */
enum {
  LPT_BVH_LBVH = 0, /* Morton codes, radix sort, radix-tree hierarchy */
  LPT_BVH_PLOC = 1  /* Morton order, then bottom-up clustering by merged surface area: SAH-class trees */
};

typedef struct lpt_ctx lpt_ctx;

/* Work counters of the most recent lpt_render_frames call. */
typedef struct lpt_stats {
  uint64_t paths;         /* camera paths started */
  uint64_t segments;      /* ray segments traced (iterations of traceRay's loop) */
  uint64_t frames;        /* frames rendered */
  double   device_ms;     /* HIP-event time of the whole call on the context's stream */
  double   extend_ms;     /* summed HIP-event time of the launches of the dominant kernel: k_extend
                             (wavefront; the camera rays' packet kernel is not among them) or k_trace */
  uint32_t extend_launches;
  uint32_t pipeline;      /* organisation the call ran with (what LPT_PIPELINE_AUTO resolved to) */
  uint64_t kernel_segments; /* the segments traced by the launches that extend_ms times */
  uint32_t timed_bounce_lo; /* those launches traced the segments of bounces [lo, hi) (of the call's last batch; */
  uint32_t timed_bounce_hi; /*  the wavefront organisation chooses per batch where its persistent launch takes over) */
  uint32_t stack_need;      /* traversal-stack entries a ray could need at once by the shape of the uploaded trees
                               (a bound over all rays, computed at upload) */
  uint32_t stack_entries;   /* entries per lane the search kernel that ran keeps on chip: 16, 20, 24, 32, 48 or 64.
                               Below stack_need only for the wavefront organisation's search kernel, which then leaves
                               the rare rays that want more to a small launch with stacks in device memory */
} lpt_stats;

/* ---- context ---------------------------------------------------------------- */

/* Replaces RendererCore's device selection/creation (src/RendererCore.cpp:72-160)
 * for compute use: binds a context to HIP device `device`. */
int lpt_create(int device, lpt_ctx** out);
void lpt_destroy(lpt_ctx* ctx);
/* Message of the last failing call on ctx (or of lpt_create when ctx is NULL). */
const char* lpt_last_error(const lpt_ctx* ctx);
/* Run on a caller-owned hipStream_t (NULL = the context's own stream).  Replaces
 * the graphics-queue submission of src/RendererCore.cpp:388-398. */
int lpt_set_stream(lpt_ctx* ctx, void* hip_stream);

/* ---- scene: bindings 2..5 of descriptor set 0 ------------------------------- */

/* One GPUTexture (include/GPUTexture.hpp:9-13): the image and sampler created by
 * PTSceneConverter::copyTextureToGPU (src/PTSceneConverter.cpp:243-389) — one mip
 * level, RGBA8 UNORM.  Filter and address modes carry the Vulkan enum values the
 * reference passes on (vk::Filter, vk::SamplerAddressMode).
 * FORMAT: the reference creates the Vulkan image with whatever format and texel size its
 * loader reports (image->getFormat(), image->pixelSize(), src/PTSceneConverter.cpp:248,276,348),
 * so an sRGB or 16-bit image would be sampled as such there.  This structure has no format
 * field: pixels are 8-bit RGBA, sampled as UNORM (texel / 255, no sRGB decode).  Callers
 * holding any other format must convert before the upload; the host library's loader does
 * (8-bit PNG / JPEG expanded to RGBA8), and refuses what it cannot convert. */
typedef struct lpt_texture {
  const void* pixels;   /* width*height*4 bytes, row 0 first */
  uint32_t width, height;
  uint32_t mag_filter;  /* 0 nearest, 1 linear (samplerInfo.magFilter, :356; the compute
                           shader samples level 0, so minFilter/mipmapMode never apply) */
  uint32_t address_u;   /* 0 repeat, 1 mirrored repeat, 2 clamp to edge, 3 clamp to border
                           (opaque black, :363) */
  uint32_t address_v;   /* note the reference passes wrappingW() here (:359) */
} lpt_texture;
#define LPT_MAX_TEXTURES 512u  /* sampler2D textures[512], shaders/path_tracing.comp:105 */
/* Most primitives one BVH leaf may hold (triangles of a mesh-tree leaf, objects of an object-tree leaf):
 * the traversal references of the device layer carry a 6-bit count.  lpt_upload_scene refuses larger
 * leaves with LPT_ERR_UNSUPPORTED; builders must split them. */
#define LPT_MAX_LEAF_PRIMITIVES 63u

/* Replaces the texture half of the scene upload: copyTextureToGPU per material texture
 * (src/PTSceneConverter.cpp:69-84) and binding 6 of descriptor set 0
 * (src/RendererPT.cpp:311-360).  Replaces any textures uploaded before; pixels are
 * copied.  Call it BEFORE lpt_upload_scene: a scene is checked against the textures
 * present at its upload, and uploading textures drops the current scene. */
int lpt_upload_textures(lpt_ctx* ctx, const lpt_texture* textures, uint32_t n_textures);

/* Replaces PTSceneConverter::copyToGPU x4 (src/PTSceneConverter.cpp:146-153) and
 * RendererPT::initializeAndBindSceneBuffer (src/RendererPT.cpp:433-509).  The
 * host arrays are copied; the caller keeps ownership.  Texture indices of the
 * objects (0xFFFFFFFF = none) must address textures uploaded before.
 * Objects may share a mesh tree only as instances, i.e. with equal bvhOffset AND equal
 * verticesOffset; mesh-tree nodes reached from objects that differ in either are refused
 * with LPT_ERR_UNSUPPORTED (the reference's converter never emits such buffers: every
 * sub-mesh gets its own nodes and vertices, src/PTSceneConverter.cpp:87-120). */
int lpt_upload_scene(lpt_ctx* ctx, const lpt_object* objects, uint32_t n_objects,
                     const lpt_bvh_node* object_bvh, uint32_t n_object_bvh,
                     const lpt_vertex* vertices, uint32_t n_vertices,
                     const lpt_bvh_node* mesh_bvh, uint32_t n_mesh_bvh);

/* ---- accumulation image: binding 0 ------------------------------------------ */

/* Replaces RendererPT::initializeAccumulationTexture (src/RendererPT.cpp:234-309):
 * (re)allocates the RGBA32F accumulation image for a width x height render.  The
 * context then owns every pixel (single-GPU case). */
int lpt_resize(lpt_ctx* ctx, int width, int height);
/* Multi-GPU sharding, no reference counterpart: the image is cut into
 * tile_w x tile_h tiles (multiples of 8), tile g belongs to rank g % n_ranks.
 * Pixel coordinates, jitter scale and per-pixel seeds stay those of the full
 * image, so the union of all ranks' pixels equals the single-GPU image bit for bit. */
int lpt_set_partition(lpt_ctx* ctx, int tile_w, int tile_h, int rank, int n_ranks);

/* ---- rendering: the dispatch of src/RendererPT.cpp:527-533 ------------------ */

/* MAX_TRACE_DEPTH (shaders/path_tracing.comp:20), BSDF family (:22), kernel
 * organisation, and how many frames may be in flight at once (0 = automatic: up to
 * 256 Mi paths, fewer if device memory is short). */
int lpt_set_options(lpt_ctx* ctx, int max_bounces, int bsdf_mode, int pipeline, int frames_in_flight);

/* Renders n_frames consecutive frames (one sample per pixel each), frame i with
 * ubos[i]; equivalent to n_frames dispatches of path_tracing.comp::main
 * (shaders/path_tracing.comp:330-351) in order, including `reset` handling.
 * The call queues its work on the context's stream behind whatever the stream still holds and
 * returns without waiting for it, so the host may prepare the next call while the GPU renders this
 * one (RendererCore::drawFrame waits for its single in-flight fence at this point,
 * src/RendererCore.cpp:376; here the only wait is for the pinned staging buffer used two batches
 * ago).  `ubos` is copied before the call returns.  lpt_get_stats reports the latest call. */
int lpt_render_frames(lpt_ctx* ctx, const lpt_ubo* ubos, int n_frames);
/* One frame; the call RendererPT::drawFrame lands on. */
int lpt_render_frame(lpt_ctx* ctx, const lpt_ubo* ubo);
/* One frame without any allocation.  No counterpart in the reference: it is for callers that cannot
 * allocate device memory at this point — code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  Same result as lpt_render_frame (bit for bit), rendered by the persistent
 * organisation whatever lpt_set_options chose; its working memory (ray cursors, the uniform block, one
 * RGBA32F sample per owned pixel slot) is a static device array of 36 MiB that every GPU the library runs
 * on holds until the process ends.  Limits: a context that owns more than 2 359 296 pixel slots (above
 * 1920x1080) gets LPT_ERR_UNSUPPORTED; lpt_upload_scene and lpt_resize (which do allocate) must have been
 * called before.  All contexts on one GPU share the array; the library puts their calls in order (a call
 * from another context or stream first makes its stream wait for the event recorded behind the last call, which
 * allocates nothing), so contexts need not coordinate — but such calls run one after the other, never side by
 * side.  Under stream capture the call records no events: lpt_get_stats then has no timing for it, and the
 * order against other users of the array between graph launches is the caller's to keep.  Queues its work and
 * returns; lpt_synchronize / lpt_get_stats / lpt_read_accum wait for it.  lpt_stats.extend_ms is 0 for it. */
int lpt_render_frame_static(lpt_ctx* ctx, const lpt_ubo* ubo);
/* Blocks until all submitted work of the context has finished. */
int lpt_synchronize(lpt_ctx* ctx);
int lpt_get_stats(lpt_ctx* ctx, lpt_stats* out);

/* Copies the accumulation image to host memory as width*height RGBA32F, row 0
 * first (imageLoad order, no y flip).  Pixels owned by other ranks are zero. */
int lpt_read_accum(lpt_ctx* ctx, float* dst_host);
/* Number of owned pixels slots (tiles x tile_w x tile_h) and a device pointer to the
 * compact owned-pixel buffer (RGBA32F per slot), for device-side gathers. */
int lpt_owned_slots(lpt_ctx* ctx, uint64_t* n_slots);
int lpt_owned_accum_device(lpt_ctx* ctx, void** device_ptr);
/* Scatters `n_ranks` concatenated compact buffers (as produced by an all-gather of
 * every rank's lpt_owned_accum_device buffer, device memory) into a scan-line
 * width*height RGBA32F device image. */
int lpt_assemble_gathered(lpt_ctx* ctx, const void* gathered_device, int n_ranks, void* image_device);

/* Display resolve of shaders/tex_to_quad.frag:24-33 (exposure 1.5, gamma 2.2, y
 * flip) into width*height RGBA32F host memory. */
int lpt_resolve(lpt_ctx* ctx, float inv_sample_count, float* dst_host);

/* ---- BVH construction on the GPU -------------------------------------------- */

/* Replaces lsg::bvh::SplitBVHBuilder<float>::process for one sub-mesh
 * (src/PTSceneConverter.cpp:97-101).  positions: n_triangles*9 floats (three
 * corners per triangle).  Writes at most 2*n_triangles-1 nodes (child indices
 * relative to node 0 of this tree) and the triangle order of the leaves
 * (src/PTSceneConverter.cpp:105).  *n_nodes_out receives the node count. */
int lpt_build_mesh_bvh(lpt_ctx* ctx, int builder, const float* positions, uint32_t n_triangles,
                       lpt_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* prim_order_out);

/* ---- test hooks ---------------------------------------------------------------- */

/* Caps the scheduling rounds a wave of the persistent kernels may run (0 restores the default of
 * 2^24, far above any real batch), so that a test can provoke LPT_ERR_TRUNCATED. */
int lpt_debug_set_trace_round_limit(lpt_ctx* ctx, uint32_t rounds);

/* Evaluates one device math routine element-wise (op: 0 sin, 1 cos, 2 tan, 3 log,
 * 4 exp, 5 pow(a,b), 6 a/b, 7 sqrt, 8 srgb->linear); a, b, out are host arrays. */
int lpt_debug_math(lpt_ctx* ctx, int op, const float* a, const float* b, float* out, uint32_t n);
/* Closest-hit query for n rays (origins/directions: n*3 floats, host memory)
 * against the uploaded scene; t[i] = 3.4e38f on a miss. */
int lpt_debug_intersect(lpt_ctx* ctx, const float* origins, const float* directions, uint32_t n,
                        float* t, uint32_t* object_index, uint32_t* primitive_index);

#ifdef __cplusplus
}
#endif
#endif /* LPT_H */
