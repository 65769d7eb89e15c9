/*
 * CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference path tracer's compute path
 * (shaders/path_tracing.comp and the GLSL files it includes).  It exists so the
 * HIP kernels can be checked against an independent implementation of the same
 * algorithm.  Only tests, the smoke check and the benchmark's cpu_baseline leg
 * may link or load this library; the product library never does.
 *
 * Pinning status: the reference ships no tests, golden images or known-answer
 * vectors, and its shader cannot be compiled or run in this environment
 * (no Vulkan loader, no GLSL compiler, scene-graph submodules empty).  The
 * integer RNG, the closed-form primitives and the struct layouts are pinned by
 * hand-derived vectors (tests/golden/), the microfacet walk by physics and
 * literature checks that do not depend on the shader's text (white furnace,
 * Fresnel/Snell, Smith-GGX single scattering, visible-normal distribution:
 * tests/test_oracle_physics.py); whole-image radiance is "parity unpinned"
 * with respect to the reference binary itself.
 *
 * Arithmetic contract (shared, by specification, with the HIP kernels):
 *   - IEEE-754 binary32, round-to-nearest-even, no fused multiply-add,
 *     denormals kept; +,-,*,/ and sqrt correctly rounded.
 *   - dot(a,b)   = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   - cross(a,b) = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
 *   - normalize(v) = v * (1 / sqrt(dot(v,v)))
 *   - min/max follow IEEE minNum/maxNum (a NaN operand yields the other one)
 *   - mat4 * (p,1) = ((c0*p.x + c1*p.y) + c2*p.z) + c3 ;  mat3 * v likewise
 *   - sin, cos, tan, log, exp, pow are the fixed polynomial algorithms in
 *     lpt_oracle.c (GLSL leaves their precision to the implementation).
 *   - texture(sampler2D, uv) in a compute shader reads mip level 0 with the sampler's
 *     magnification filter; Vulkan leaves the weight precision of linear filtering to
 *     the implementation, so it is fixed here (lpto_sample_texture): RGBA8 texel c ->
 *     c / 255, unnormalised coordinate x = u*W (nearest) or u*W - 0.5 (linear),
 *     i0 = floor(x), a = x - floor(x), result = (t00*(1-a) + t10*a)*(1-b) + (t01*(1-a) + t11*a)*b,
 *     texel indices wrapped per address mode.  Filtering by a real Vulkan device differs
 *     in the last bits: "parity unpinned" for textured materials.
 */
#ifndef LPT_ORACLE_H
#define LPT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/PTSceneConverter.hpp:15-42, shaders/path_tracing.comp:41-57 (208 B) */
typedef struct {
  float worldMatrix[16];        /* column-major */
  float worldMatrixInverse[16]; /* column-major */
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
  uint32_t bvhOffset;      /* element index into the mesh BVH node array */
  uint32_t verticesOffset; /* element index into the vertex array */
  uint8_t padding[8];
} lpto_object;

/* include/PTSceneConverter.hpp:44-50, shaders/path_tracing.comp:59-63 (48 B) */
typedef struct {
  float position[3];
  float pad0;
  float normal[3];
  float pad1;
  float uv[2];
  float pad2[2];
} lpto_vertex;

/* include/PTSceneConverter.hpp:52-59, shaders/path_tracing.comp:29-39 (48 B) */
typedef struct {
  float minCorner[3];
  float pad0;
  float maxCorner[3];
  uint32_t isLeaf;
  uint32_t indices[2];
  uint32_t pad1[2];
} lpto_bvh_node;

/* include/RendererPT.h:51-61, shaders/path_tracing.comp:24-27,83-87 (92 B on the host) */
typedef struct {
  float worldMatrix[16]; /* camera-to-world, column-major */
  float fovY;
  float pad[3];
  uint32_t seed[2];
  uint32_t reset;
} lpto_ubo;

/* One entry of `sampler2D textures[512]` (shaders/path_tracing.comp:105): the image and
 * sampler src/PTSceneConverter.cpp:243-389 creates (one mip level, RGBA8).  Filter and
 * address modes carry the Vulkan enum values (VkFilter, VkSamplerAddressMode). */
typedef struct {
  const uint8_t* pixels; /* width*height*4 bytes, row 0 first */
  uint32_t width, height;
  uint32_t magFilter;    /* 0 nearest, 1 linear */
  uint32_t addressU;     /* 0 repeat, 1 mirrored repeat, 2 clamp to edge, 3 clamp to border (opaque black) */
  uint32_t addressV;
} lpto_texture;

typedef struct {
  const lpto_object* objects;
  uint32_t nObjects;
  const lpto_bvh_node* objectBvh;
  uint32_t nObjectBvh;
  const lpto_vertex* vertices;
  uint32_t nVertices;
  const lpto_bvh_node* meshBvh;
  uint32_t nMeshBvh;
  const lpto_texture* textures; /* may be NULL when nTextures == 0 */
  uint32_t nTextures;
} lpto_scene;

enum { LPTO_BSDF_HEITZ = 0, LPTO_BSDF_BASIC = 1 };

typedef struct {
  int32_t width, height;       /* full image resolution */
  int32_t x0, y0, w, h;        /* pixel rectangle to render (tile) */
  int32_t maxBounces;          /* MAX_TRACE_DEPTH, path_tracing.comp:20 */
  int32_t bsdfMode;            /* LPTO_BSDF_* (USE_MICROFACET, path_tracing.comp:22) */
  int32_t threads;             /* OpenMP threads, <=0: runtime default */
} lpto_params;

/* Per-call work counters (summed over the pixels rendered). */
typedef struct {
  uint64_t paths;          /* pixels traced */
  uint64_t segments;       /* iterations of traceRay's loop that ran sceneIntersect */
  uint64_t hits;           /* segments that hit geometry */
  uint64_t boxTests;       /* rayAABBIntersectTest calls (TLAS + BLAS) */
  uint64_t rootVisits;     /* TLAS root + BLAS roots entered (never box-tested) */
  uint64_t triTests;       /* rayTriangleIntersect calls */
  uint64_t objectsEntered; /* objectIntersect calls */
  uint64_t walkSteps;      /* microsurface random-walk iterations */
  uint64_t rands;          /* rand() calls */
  uint32_t maxStackTlas;
  uint32_t maxStackBlas;
  uint32_t stackOverflow;  /* non-zero if a traversal needed more than LPTO_STACK entries */
  uint32_t maxSegmentSteps; /* largest number of traversal steps (node visits + triangle tests + objects) of one segment */
} lpto_stats;

#define LPTO_STACK 128

/* ---- whole-path entry points -------------------------------------------------- */

/* One frame (one sample per pixel) of shaders/path_tracing.comp:330-351 over the tile.
 * accum is the full-resolution RGBA32F image (width*height*4 floats, row-major,
 * row 0 first).  Returns 0, or a negative error (unsupported texture id, stack). */
int lpto_render_frame(const lpto_scene* scene, const lpto_ubo* ubo, const lpto_params* params,
                      float* accum, lpto_stats* stats);

/* shaders/tex_to_quad.frag:24-33 — resolve accumulated radiance to display values
 * (exposure 1.5, gamma 2.2, y flip).  out is width*height*4 floats. */
void lpto_resolve(const float* accum, int width, int height, float invSampleCount, float* out);

/* ---- primitives, exported for known-answer tests ---------------------------- */

typedef struct { uint32_t x, y; } lpto_seed;

float lpto_rand(lpto_seed* seed);                                    /* common/random.glsl:9-15 */
int lpto_ray_aabb_test(const float o[3], const float d[3], const float mn[3], const float mx[3],
                       float distance);                              /* common/ray.glsl:39-61 */
float lpto_ray_triangle(const float o[3], const float d[3], const float v0[3], const float v1[3],
                        const float v2[3]);                          /* common/ray.glsl:63-82 */
void lpto_barycentric(const float p[3], const float v0[3], const float v1[3], const float v2[3],
                      float out[3]);                                 /* common/util.glsl:23-41 */
float lpto_srgb_to_linear(float srgb);                               /* common/util.glsl:4-8 */
float lpto_fresnel(float vdoth, float eta);                          /* heitz/BSDF.glsl:10-24 */
void lpto_refract_eta(const float wi[3], const float wm[3], float eta, float out[3]); /* :26-32 */
uint32_t lpto_interaction_type(float metallic, float transmission, lpto_seed* seed);
                                                                     /* heitz/interaction_type.glsl:10-29 */
void lpto_sample_ggx_vndf(const float ve[3], float alpha, lpto_seed* seed, float out[3]); /* :41-67 */
float lpto_sample_ggx_height(const float dir[3], float height, float alpha, lpto_seed* seed); /* :72-84 */
/* kind: 1 diffuse, 2 conductor, 4 dielectric (interaction_type.glsl:6-8); mode: LPTO_BSDF_*.
 * Writes lightDir (local frame) and the returned throughput weight. */
void lpto_bsdf(int mode, uint32_t kind, const float F0[3], const float viewDir[3], float roughness,
               float transmission, float ior, int outside, lpto_seed* seed, float lightDir[3],
               float weight[3]);
/* path_tracing.comp:107-127 for pixel (px,py) of a (resx,resy) image. */
void lpto_generate_ray(const lpto_ubo* ubo, int px, int py, int resx, int resy, lpto_seed* seed,
                       float o[3], float d[3]);
/* path_tracing.comp:175-209; returns 1 on hit and fills t/object/first-vertex index. */
int lpto_scene_intersect(const lpto_scene* scene, const float o[3], const float d[3], float* t,
                         uint32_t* objectIndex, uint32_t* primitiveIndex, lpto_stats* stats);
/* Same closest-hit rule with every triangle of every object tested in storage order
 * (no BVH); used to check BVH builders. */
int lpto_brute_force_intersect(const lpto_scene* scene, const float o[3], const float d[3], float* t,
                               uint32_t* objectIndex, uint32_t* primitiveIndex);

/* Analysis aid: when non-NULL, every visit of object-BVH node i / mesh-BVH node j increments
 * tlas[i] / blas[j].  Pass NULLs to switch it off (the default). */
void lpto_set_visit_histograms(uint32_t* tlas, uint32_t* blas);

/* texture(textures[i], uv) as fixed by the arithmetic contract; out = rgba. */
void lpto_sample_texture(const lpto_texture* tex, float u, float v, float out[4]);

/* The fixed transcendental algorithms of the arithmetic contract. */
float lpto_sin(float x);
float lpto_cos(float x);
float lpto_tan(float x);
float lpto_log(float x);
float lpto_exp(float x);
float lpto_pow(float x, float y);

#ifdef __cplusplus
}
#endif
#endif
