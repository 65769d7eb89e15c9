// Device-side scene layout, two-level BVH traversal and surface shading for the
// path-tracing kernels.  Follows shaders/path_tracing.comp of the reference in
// behaviour (visit order, arithmetic, random-number order); the data layout is
// derived at upload time from the interchange buffers for wide, aligned loads.
#pragma once
#include "../../include/lpt.h"
#include "lpt_bsdf.h"

#pragma clang fp contract(off)

namespace lpt {

// shaders/path_tracing.comp:21
constexpr uint32_t kRussianRouletteBounces = 2;

// ---- traversal references -------------------------------------------------------
// A reference is what the traversal stack holds (the reference shader's stack holds
// node indices, path_tracing.comp:141,181).  Inner node: index into the derived node
// array (object tree and all mesh trees share one array).  Leaf: bit 31 set, bits
// 25..30 = primitive count (<=63), bits 0..24 = first primitive: an index into the
// derived triangle array (mesh trees) or an object index (object tree).
constexpr uint32_t kRefLeafBit = 0x80000000u;
constexpr uint32_t kRefSentinel = 0xFFFFFFFFu;  // bottom of stack ("-1" in the shader)
constexpr uint32_t kRefMarker = 0x7FFFFFFFu;    // end of an object's mesh tree
constexpr uint32_t kRefFirstMask = 0x01FFFFFFu;
constexpr uint32_t kMaxLeafCount = LPT_MAX_LEAF_PRIMITIVES;
static_assert(kMaxLeafCount == 63u, "leaf references carry a 6-bit count");
constexpr uint32_t kMaxLeafFirst = 0x01FFFFFEu;

__host__ __device__ inline uint32_t make_leaf_ref(uint32_t first, uint32_t count) {
  return kRefLeafBit | (count << 25) | first;
}
__host__ __device__ inline uint32_t leaf_first(uint32_t ref) { return ref & kRefFirstMask; }
__host__ __device__ inline uint32_t leaf_count(uint32_t ref) { return (ref >> 25) & 0x3Fu; }

// Derived inner node, 64 B: the boxes of BOTH children (the interchange node,
// path_tracing.comp:29-39, stores its own box, so a visit there costs three node
// fetches; here it costs one).
// Components are interleaved child0/child1 so that one 16-byte load yields two aligned
// pairs for packed-f32 arithmetic: [mn.x0 mn.x1 mn.y0 mn.y1] [mn.z0 mn.z1 mx.x0 mx.x1]
// [mx.y0 mx.y1 mx.z0 mx.z1] [ref0 ref1 - -].
struct alignas(16) TNode {
  float mnx[2], mny[2];
  float mnz[2], mxx[2];
  float mxy[2], mxz[2];
  uint32_t ref0, ref1;
  uint32_t pad[2];
};
static_assert(sizeof(TNode) == 64, "TNode must be 64 bytes");

// Triangle for intersection: corner 0 and the two edges of ray.glsl:64-65, plus the
// index of its first vertex in the interchange vertex buffer (what a hit reports).
struct alignas(16) Tri {
  float v0[3];
  uint32_t prim;
  float e1[3], pad1;
  float e2[3], pad2;
};
static_assert(sizeof(Tri) == 48, "Tri must be 48 bytes");

// Per-object record for traversal, 64 B.
struct alignas(16) ObjTrav {
  float inv[12];     // worldMatrixInverse rows 0..2: (m0 m4 m8 m12), (m1 ...), (m2 ...)
  uint32_t rootRef;  // reference of the mesh tree's root (never box-tested, path_tracing.comp:144)
  uint32_t pad[3];
};
static_assert(sizeof(ObjTrav) == 64, "ObjTrav must be 64 bytes");

// Per-object record for shading, 160 B.  The last 32 bytes are only read for objects with
// textures (textured != 0).
constexpr uint32_t kNoTexture16 = 0xFFFFu;
struct alignas(16) ObjShade {
  float inv[12];        // as in ObjTrav
  float nrm[9];         // mat3(worldMatrix) rows
  float baseLinear[3];  // SRGBToLinear(baseColorFactor.rgb), path_tracing.comp:263 (untextured colour)
  float emission[3];
  float roughness;      // max(roughnessFactor, 0.001), path_tracing.comp:238
  float metallic;
  float transmission;
  float ior;
  uint32_t textured;    // non-zero if any of the five texture slots is used
  float baseFactor[4];  // baseColorFactor as uploaded (multiplied with the colour texture before SRGBToLinear)
  uint32_t texColorEmission;         // colour | emission << 16; kNoTexture16 = slot unused
  uint32_t texMetalRoughTransmission;  // metallicRoughness | transmission << 16
  uint32_t texNormal;
  uint32_t pad;
};
static_assert(sizeof(ObjShade) == 160, "ObjShade must be 160 bytes");

// One entry of `sampler2D textures[512]` (path_tracing.comp:105): RGBA8 image, one mip level,
// with the sampler state src/PTSceneConverter.cpp:351-386 sets (Vulkan enum values).
struct alignas(16) DTexture {
  const uint8_t* pixels;  // width*height*4, row 0 first
  uint32_t width, height;
  uint32_t magFilter;     // 0 nearest, 1 linear
  uint32_t addressU, addressV;  // 0 repeat, 1 mirrored repeat, 2 clamp to edge, 3 clamp to border (opaque black)
  uint32_t pad;
};
static_assert(sizeof(DTexture) == 32, "DTexture must be 32 bytes");

// Nodes and triangles are stored hottest first (largest world-space box first, parents
// before children), so that a prefix of each array is what a kernel wants to keep in LDS.
struct DScene {
  const TNode* nodes;
  const Tri* tris;
  const ObjTrav* otrav;
  const ObjShade* oshade;
  const lpt_vertex* verts;
  const DTexture* textures;
  uint32_t rootRef;   // reference of the object tree's root
  uint32_t nNodes, nTris, nObjects;
};

// Queued paths carry their id in one word; bit 31 says that samples[path id] already holds the
// radiance the path has gathered (otherwise it has gathered none yet).
constexpr uint32_t kAccStored = 0x80000000u;

// Address of record `index` as 16-byte words.  Every array here is far below 4 GiB (references
// carry 25-bit indices), so the byte offset is formed in 32 bits: one vector instruction and the
// addressing form "uniform base + 32-bit lane offset" instead of 64-bit multiply-adds.
template <class T>
__device__ __forceinline__ const float4* record_words(const T* base, uint32_t index) {
  return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + index * (uint32_t)sizeof(T));
}

struct Hit {
  float t;
  uint32_t obj;
  uint32_t prim;  // index of the triangle's first vertex (path_tracing.comp:155)
};

__device__ __forceinline__ f3 xform_point(const float* r, f3 p) {
  return f3{((r[0] * p.x + r[1] * p.y) + r[2] * p.z) + r[3], ((r[4] * p.x + r[5] * p.y) + r[6] * p.z) + r[7],
            ((r[8] * p.x + r[9] * p.y) + r[10] * p.z) + r[11]};
}
__device__ __forceinline__ f3 xform_dir(const float* r, f3 v) {
  return f3{(r[0] * v.x + r[1] * v.y) + r[2] * v.z, (r[4] * v.x + r[5] * v.y) + r[6] * v.z,
            (r[8] * v.x + r[9] * v.y) + r[10] * v.z};
}
__device__ __forceinline__ f3 xform_mat3(const float* r, f3 v) {
  return f3{(r[0] * v.x + r[1] * v.y) + r[2] * v.z, (r[3] * v.x + r[4] * v.y) + r[5] * v.z,
            (r[6] * v.x + r[7] * v.y) + r[8] * v.z};
}

// shaders/common/ray.glsl:39-61 with the reciprocal direction hoisted out (it is
// recomputed there from the same direction on every call).
__device__ __forceinline__ bool box_test(f3 o, f3 inv, const float* mn, const float* mx, float dist) {
  float nx = (mn[0] - o.x) * inv.x, ny = (mn[1] - o.y) * inv.y, nz = (mn[2] - o.z) * inv.z;
  float fx = (mx[0] - o.x) * inv.x, fy = (mx[1] - o.y) * inv.y, fz = (mx[2] - o.z) * inv.z;
  float t0 = max_num(max_num(min_num(nx, fx), min_num(ny, fy)), min_num(nz, fz));
  float t1 = min_num(min_num(max_num(nx, fx), max_num(ny, fy)), max_num(nz, fz));
  if (t0 > t1) return false;
  if (t0 > 0.0f) return t0 < dist;
  return t1 > 0.0f;
}

// The same slab test for the two children of a derived node at once.  Subtractions and
// multiplications run as packed-f32 operations (v_pk_add_f32 / v_pk_mul_f32: two IEEE
// single-precision results per lane and instruction, rounded exactly like the scalar forms),
// halving their instruction count; min/max and the compares stay scalar.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void box_test_pair(f3 o, f3 inv, float4 q0, float4 q1, float4 q2, float dist, bool& h0,
                                              bool& h1) {
  const v2f ox{o.x, o.x}, oy{o.y, o.y}, oz{o.z, o.z};
  const v2f ix{inv.x, inv.x}, iy{inv.y, inv.y}, iz{inv.z, inv.z};
  const v2f nx = (v2f{q0.x, q0.y} - ox) * ix, ny = (v2f{q0.z, q0.w} - oy) * iy, nz = (v2f{q1.x, q1.y} - oz) * iz;
  const v2f fx = (v2f{q1.z, q1.w} - ox) * ix, fy = (v2f{q2.x, q2.y} - oy) * iy, fz = (v2f{q2.z, q2.w} - oz) * iz;
  // min/max element-wise on the pairs
  const v2f lo = __builtin_elementwise_max(
      __builtin_elementwise_max(__builtin_elementwise_min(nx, fx), __builtin_elementwise_min(ny, fy)),
      __builtin_elementwise_min(nz, fz));
  const v2f hi = __builtin_elementwise_min(
      __builtin_elementwise_min(__builtin_elementwise_max(nx, fx), __builtin_elementwise_max(ny, fy)),
      __builtin_elementwise_max(nz, fz));
  // The decision  !(t0 > t1) && (t0 > 0 ? t0 < dist : t1 > 0)  in three comparisons, given that the
  // caller's dist is a positive number (a hit distance above kEps, or kInfinity):
  //  - t0 > 0 with t0 <= t1 implies t1 > 0, and t0 <= 0 implies t0 < dist, so each arm of the
  //    conditional equals (t1 > 0) && (t0 < dist);
  //  - t0 is NaN exactly when t1 is (both only when all three axes gave NaN), and then both forms are false.
  h0 = !(lo.x > hi.x) & (hi.x > 0.0f) & (lo.x < dist);
  h1 = !(lo.y > hi.y) & (hi.y > 0.0f) & (lo.y < dist);
}

// shaders/common/ray.glsl:63-82 (Moeller-Trumbore, no culling, no parallel check)
__device__ __forceinline__ float tri_test(f3 o, f3 d, f3 v0, f3 e1, f3 e2) {
  f3 pvec = cross3(d, e2);
  float det = 1.0f / dot3(e1, pvec);
  f3 tvec = o - v0;
  float u = dot3(tvec, pvec) * det;
  if (u < 0.0f || u > 1.0f) return kInfinity;
  f3 qvec = cross3(tvec, e1);
  float v = dot3(d, qvec) * det;
  if (v < 0.0f || u + v > 1.0f) return kInfinity;
  return dot3(e2, qvec) * det;
}

// Closest hit along (o,d): sceneIntersect + objectIntersect of
// shaders/path_tracing.comp:130-209 on one shared stack.  `stk` points at this
// thread's column of an LDS array laid out [entry][thread]; `stride` is the
// number of threads per entry row.  The caller guarantees the scene's stack
// need (computed at upload) fits the array.
// Visit order is the shader's: children are tested in index order (x then y),
// pushed in that order and popped last-in-first-out, so y is visited first;
// a hit replaces the current one only if strictly closer.
__device__ __forceinline__ Hit intersect_scene(const DScene& sc, f3 wo, f3 wd, uint32_t* stk, int stride) {
  Hit hit{kInfinity, 0u, 0u};
  const f3 winv{1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z};
  f3 co = wo, cd = wd, cinv = winv;
  bool inObj = false;
  uint32_t curObj = 0;
  int sp = 0;
  stk[0] = kRefSentinel;
  sp = 1;
  uint32_t cur = sc.rootRef;

  while (cur != kRefSentinel) {
    if (cur == kRefMarker) {  // finished an object's mesh tree: back to world space
      inObj = false;
      co = wo;
      cd = wd;
      cinv = winv;
      cur = stk[(--sp) * stride];
      continue;
    }
    if (!(cur & kRefLeafBit)) {
      const float4* q = reinterpret_cast<const float4*>(sc.nodes + cur);
      float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
      bool h0, h1;
      box_test_pair(co, cinv, q0, q1, q2, hit.t, h0, h1);
      uint32_t r0 = as_uint(q3.x), r1 = as_uint(q3.y);
      if (h1) {
        if (h0) stk[(sp++) * stride] = r0;
        cur = r1;
      } else if (h0) {
        cur = r0;
      } else {
        cur = stk[(--sp) * stride];
      }
      continue;
    }
    uint32_t first = leaf_first(cur);
    uint32_t count = leaf_count(cur);
    if (inObj) {
      for (uint32_t i = first; i < first + count; ++i) {
        const float4* tq = reinterpret_cast<const float4*>(sc.tris + i);
        float4 a = tq[0], b = tq[1], c = tq[2];
        float td = tri_test(co, cd, f3{a.x, a.y, a.z}, f3{b.x, b.y, b.z}, f3{c.x, c.y, c.z});
        if (td > kEps && td < hit.t) {
          hit.t = td;
          hit.obj = curObj;
          hit.prim = as_uint(a.w);
        }
      }
      cur = stk[(--sp) * stride];
    } else {
      if (count == 0u) {
        cur = stk[(--sp) * stride];
        continue;
      }
      if (count > 1u) stk[(sp++) * stride] = make_leaf_ref(first + 1u, count - 1u);
      stk[(sp++) * stride] = kRefMarker;
      const float4* oq = reinterpret_cast<const float4*>(sc.otrav + first);
      float4 r0 = oq[0], r1 = oq[1], r2 = oq[2], r3 = oq[3];
      float rows[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
      co = xform_point(rows, wo);
      cd = xform_dir(rows, wd);
      cinv = f3{1.0f / cd.x, 1.0f / cd.y, 1.0f / cd.z};
      curObj = first;
      inObj = true;
      cur = as_uint(r3.x);
    }
  }
  return hit;
}

// ---- path state ------------------------------------------------------------------

struct Path {
  f3 o, d;     // current ray (world space)
  f3 mask;     // throughput
  f3 acc;      // radiance gathered so far
  Rng rng;
};

// shaders/path_tracing.comp:107-127 for global pixel (px,py) of a resx x resy image
__device__ __forceinline__ void generate_ray(const lpt_ubo& ubo, int px, int py, float resx, float resy, Path& p) {
  p.rng.x = ubo.seed[0] * (uint32_t)px;  // path_tracing.comp:341
  p.rng.y = ubo.seed[1] * (uint32_t)py;
  float r1 = 2.0f * rand01(p.rng);
  float r2 = 2.0f * rand01(p.rng);
  float jx = r1 < 1.0f ? __builtin_sqrtf(r1) - 1.0f : 1.0f - __builtin_sqrtf(2.0f - r1);
  float jy = r2 < 1.0f ? __builtin_sqrtf(r2) - 1.0f : 1.0f - __builtin_sqrtf(2.0f - r2);
  jx = jx / (resx * 0.5f);
  jy = jy / (resy * 0.5f);
  float uvx = ((2.0f * (float)px) / resx - 1.0f) + jx;
  float uvy = ((2.0f * (float)py) / resy - 1.0f) + jy;
  const float* m = ubo.cameraWorldMatrix;
  float aspect = resx / resy;
  float th = tan_fixed(ubo.fovY / 2.0f);
  uvx = uvx * (aspect * th);
  uvy = uvy * th;
  f3 c0{m[0], m[1], m[2]}, c1{m[4], m[5], m[6]}, c2{m[8], m[9], m[10]};
  p.o = f3{m[12], m[13], m[14]};
  p.d = normalize3((c0 * uvx + c1 * uvy) - c2);
  p.mask = f3{1.0f, 1.0f, 1.0f};
  p.acc = f3{0.0f, 0.0f, 0.0f};
}

// shaders/common/util.glsl:23-41
__device__ __forceinline__ f3 barycentric(f3 point, f3 v0, f3 v1, f3 v2) {
  f3 ab = v1 - v0, ac = v2 - v0, ah = point - v0;
  float ab_ab = dot3(ab, ab), ab_ac = dot3(ab, ac), ac_ac = dot3(ac, ac);
  float ab_ah = dot3(ab, ah), ac_ah = dot3(ac, ah);
  float inv_denom = 1.0f / (ab_ab * ac_ac - ab_ac * ab_ac);
  float v = (ac_ac * ab_ah - ab_ac * ac_ah) * inv_denom;
  float w = (ab_ab * ac_ah - ab_ac * ab_ah) * inv_denom;
  float u = (1.0f - v) - w;
  return f3{u, v, w};
}

// One iteration of traceRay's loop after the intersection is known
// (shaders/path_tracing.comp:219-323), in two halves around the BSDF sampling so that
// kernels may run the Heitz random walk as separately scheduled steps.

// ---- texture(sampler2D, uv), path_tracing.comp:244-261,277-282 ------------------------
// A compute shader has no derivatives, so the fetch reads mip level 0 with the sampler's
// magnification filter.  Vulkan leaves the weight precision of linear filtering open; it is
// fixed here: texel c -> c/255, x = u*W - 0.5, a = x - floor(x),
// (t00*(1-a) + t10*a)*(1-b) + (t01*(1-a) + t11*a)*b.

__device__ __forceinline__ int tex_floor_index(float x) {
  float f = __builtin_floorf(x);
  if (!(f == f)) f = 0.0f;
  if (f < -1073741824.0f) f = -1073741824.0f;
  if (f > 1073741824.0f) f = 1073741824.0f;
  return (int)f;
}

// texel index, or -1 for the border colour
__device__ __forceinline__ int tex_wrap(int i, int n, uint32_t mode) {
  if (mode == 0u) {
    int r = i % n;
    return r < 0 ? r + n : r;
  }
  if (mode == 1u) {
    int m = i % (2 * n);
    if (m < 0) m += 2 * n;
    return m < n ? m : 2 * n - 1 - m;
  }
  if (mode == 2u) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
  return (i < 0 || i > n - 1) ? -1 : i;
}

struct f4 {
  float x, y, z, w;
};

__device__ __forceinline__ f4 tex_texel(const DTexture& t, int i, int j) {
  if (i < 0 || j < 0) return f4{0.0f, 0.0f, 0.0f, 1.0f};  // eFloatOpaqueBlack
  // the pointer comes out of a record in memory, so the compiler cannot tell its address space and would
  // emit flat loads: texels always live in global memory
  typedef __attribute__((address_space(1))) const uint32_t global_u32;
  const uint32_t p = ((global_u32*)t.pixels)[(size_t)j * t.width + (size_t)i];
  return f4{(float)(p & 255u) / 255.0f, (float)((p >> 8) & 255u) / 255.0f, (float)((p >> 16) & 255u) / 255.0f,
            (float)(p >> 24) / 255.0f};
}

__device__ __forceinline__ f4 sample_texture(const DTexture* textures, uint32_t index, float u, float v) {
  const DTexture t = textures[index];
  const int W = (int)t.width, H = (int)t.height;
  if (t.magFilter == 0u) {
    const int i = tex_wrap(tex_floor_index(u * (float)W), W, t.addressU);
    const int j = tex_wrap(tex_floor_index(v * (float)H), H, t.addressV);
    return tex_texel(t, i, j);
  }
  const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
  float a = x - __builtin_floorf(x), b = y - __builtin_floorf(y);
  if (!(a == a)) a = 0.0f;
  if (!(b == b)) b = 0.0f;
  const int i0 = tex_floor_index(x), j0 = tex_floor_index(y);
  const int ia = tex_wrap(i0, W, t.addressU), ib = tex_wrap(i0 + 1, W, t.addressU);
  const int ja = tex_wrap(j0, H, t.addressV), jb = tex_wrap(j0 + 1, H, t.addressV);
  const f4 t00 = tex_texel(t, ia, ja), t10 = tex_texel(t, ib, ja), t01 = tex_texel(t, ia, jb),
           t11 = tex_texel(t, ib, jb);
  const float ia1 = 1.0f - a, ib1 = 1.0f - b;
  f4 r;
  r.x = (t00.x * ia1 + t10.x * a) * ib1 + (t01.x * ia1 + t11.x * a) * b;
  r.y = (t00.y * ia1 + t10.y * a) * ib1 + (t01.y * ia1 + t11.y * a) * b;
  r.z = (t00.z * ia1 + t10.z * a) * ib1 + (t01.z * ia1 + t11.z * a) * b;
  r.w = (t00.w * ia1 + t10.w * a) * ib1 + (t01.w * ia1 + t11.w * a) * b;
  return r;
}

// What the BSDF sampling and the rest of the iteration need from the hit surface.
struct Surface {
  f3 u, v, ff;        // shading frame; ff = front-facing normal
  f3 posW;            // hit point, the next ray's origin
  f3 viewDir;         // -ray direction in the shading frame
  f3 baseLinear;
  float roughness, transmission, ior;
  uint32_t interaction;
  bool outside;       // ray arrives on the side the unflipped normal points to (:304)
};

// The tangent frame about the shading normal (path_tracing.comp:274-275, and again :280-281 after a normal map):
// a function of the normal alone, so a kernel that sets a hit aside may keep the normal and rebuild the frame.
__device__ __forceinline__ void shading_frame(f3 ff, f3& u, f3& v) {
  const f3 axis = (__builtin_fabsf(ff.x) > 0.1f) ? f3{0.0f, 1.0f, 0.0f} : f3{1.0f, 0.0f, 0.0f};
  u = normalize3(cross3(axis, ff));
  v = cross3(ff, u);
}

// :219-304.  Returns false on a miss; `sample` then holds the path's radiance.
__device__ __forceinline__ bool shade_begin(const DScene& sc, Path& p, const Hit& hit, Surface& sf, f3& sample) {
  if (hit.t == kInfinity) {  // :220-223 — assignment: radiance gathered so far is dropped
    sample = p.mask * 0.2f;
    return false;
  }
  const float4* sq = reinterpret_cast<const float4*>(sc.oshade + hit.obj);
  float4 s0 = sq[0], s1 = sq[1], s2 = sq[2], s3 = sq[3], s4 = sq[4], s5 = sq[5], s6 = sq[6], s7 = sq[7];
  float inv[12] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
  float nrm[9] = {s3.x, s3.y, s3.z, s3.w, s4.x, s4.y, s4.z, s4.w, s5.x};
  sf.baseLinear = f3{s5.y, s5.z, s5.w};
  f3 emission{s6.x, s6.y, s6.z};
  sf.roughness = s6.w;
  float metallic = s7.x;
  sf.transmission = s7.y;
  sf.ior = s7.z;
  const bool textured = as_uint(s7.w) != 0u;

  const float4* vq = reinterpret_cast<const float4*>(sc.verts + hit.prim);
  float4 p0 = vq[0], n0 = vq[1], p1 = vq[3], n1 = vq[4], p2 = vq[6], n2 = vq[7];

  f3 oo = xform_point(inv, p.o);
  f3 od = xform_dir(inv, p.d);
  sf.posW = p.o + p.d * hit.t;
  f3 bary = barycentric(oo + od * hit.t, f3{p0.x, p0.y, p0.z}, f3{p1.x, p1.y, p1.z}, f3{p2.x, p2.y, p2.z});

  float uvx = 0.0f, uvy = 0.0f;
  uint32_t texNormal = kNoTexture16;
  if (textured) {  // :234, :244-263
    const float4 t0 = vq[2], t1 = vq[5], t2 = vq[8];
    uvx = (bary.x * t0.x + bary.y * t1.x) + bary.z * t2.x;
    uvy = (bary.x * t0.y + bary.y * t1.y) + bary.z * t2.y;
    const float4 s8 = sq[8], s9 = sq[9];
    const uint32_t ce = as_uint(s9.x), mt = as_uint(s9.y);
    texNormal = as_uint(s9.z);
    if ((ce & 0xFFFFu) != kNoTexture16) {
      const f4 c = sample_texture(sc.textures, ce & 0xFFFFu, uvx, uvy);
      sf.baseLinear = f3{srgb_to_linear(s8.x * c.x), srgb_to_linear(s8.y * c.y), srgb_to_linear(s8.z * c.z)};
    }
    if ((ce >> 16) != kNoTexture16) {
      const f4 c = sample_texture(sc.textures, ce >> 16, uvx, uvy);
      emission = emission * f3{c.x, c.y, c.z};
    }
    if ((mt & 0xFFFFu) != kNoTexture16) {
      const f4 c = sample_texture(sc.textures, mt & 0xFFFFu, uvx, uvy);
      metallic = metallic * c.z;
      sf.roughness = sf.roughness * c.y;
    }
    if ((mt >> 16) != kNoTexture16) {
      const f4 c = sample_texture(sc.textures, mt >> 16, uvx, uvy);
      sf.transmission = sf.transmission * c.x;
    }
  }

  sf.interaction = pick_interaction(p.rng, metallic, sf.transmission);  // :266
  p.acc = p.acc + p.mask * emission;                                    // :269

  f3 nsum = (f3{n0.x, n0.y, n0.z} * bary.x + f3{n1.x, n1.y, n1.z} * bary.y) + f3{n2.x, n2.y, n2.z} * bary.z;
  f3 normal = normalize3(xform_mat3(nrm, nsum));  // :272 (mat3(worldMatrix), not inverse-transpose)
  sf.ff = (dot3(normal, p.d) < 0.0f) ? normal : normal * -1.0f;
  shading_frame(sf.ff, sf.u, sf.v);
  if (textured && texNormal != kNoTexture16) {  // :277-282
    const f4 c = sample_texture(sc.textures, texNormal, uvx, uvy);
    const f3 tn = normalize3(f3{c.x * 2.0f - 1.0f, c.y * 2.0f - 1.0f, c.z * 2.0f - 1.0f});
    sf.ff = normalize3((sf.u * tn.x + sf.v * tn.y) + sf.ff * tn.z);
    shading_frame(sf.ff, sf.u, sf.v);
  }
  f3 nd = -p.d;
  sf.viewDir = f3{dot3(nd, sf.u), dot3(nd, sf.v), dot3(nd, sf.ff)};
  sf.outside = dot3(normal, nd) > 0.0f;
  return true;
}

// The basic family's sampling (shaders/basic/BSDF.glsl), selected by the lobe.
__device__ __forceinline__ f3 basic_bsdf(Rng& rng, const Surface& sf, f3& lightDir) {
  if (sf.interaction == kDiff) return basic_diffuse(rng, sf.baseLinear, lightDir);
  if (sf.interaction == kMetallic) return basic_specular(sf.baseLinear, sf.viewDir, lightDir);
  return basic_transmittance(rng, sf.baseLinear, sf.viewDir, sf.transmission, sf.ior, sf.outside, lightDir);
}

// :306-323 once the BSDF weight and direction are known.  Returns true if the path goes on
// to bounce+1; otherwise `sample` holds the path's radiance.
__device__ __forceinline__ bool shade_end(Path& p, const Surface& sf, f3 weight, f3 lightDir, uint32_t bounce,
                                          uint32_t maxBounces, f3& sample) {
  p.mask = p.mask * weight;
  p.d = (sf.u * lightDir.x + sf.v * lightDir.y) + sf.ff * lightDir.z;  // :312
  p.o = sf.posW;                                                        // no origin offset; EPS guards self-hits

  float q = max_num(max_num(p.mask.x, p.mask.y), p.mask.z);  // :317-323
  if (q < 0.5f && bounce > kRussianRouletteBounces) {
    if (rand01(p.rng) > q) {
      sample = p.acc;
      return false;
    }
    p.mask = p.mask * (1.0f / q);
  }
  if (bounce + 1u >= maxBounces) {
    sample = p.acc;
    return false;
  }
  return true;
}

// UNIFIED selects the single-loop form of the three Heitz walks (heitz_bsdf_unified), which
// keeps lanes with different lobes in step; the per-lobe functions give the same values.
template <bool UNIFIED>
__device__ __forceinline__ bool shade_hit(const DScene& sc, Path& p, const Hit& hit, uint32_t bounce,
                                          uint32_t maxBounces, int bsdfMode, f3& sample) {
  Surface sf;
  if (!shade_begin(sc, p, hit, sf, sample)) return false;
  if (bounce + 1u >= maxBounces) {  // the path ends here whatever direction is sampled (shade_end's last test)
    sample = p.acc;
    return false;
  }
  f3 lightDir{0.0f, 0.0f, 0.0f};
  f3 weight;
  if (bsdfMode != LPT_BSDF_HEITZ) {
    weight = basic_bsdf(p.rng, sf, lightDir);
  } else if (UNIFIED) {
    weight = heitz_bsdf_unified(p.rng, sf.interaction, sf.baseLinear, sf.viewDir, sf.roughness, sf.ior, sf.outside,
                                lightDir);
  } else if (sf.interaction == kDiff) {
    weight = diffuse_bsdf(p.rng, sf.baseLinear, sf.viewDir, sf.roughness, lightDir);
  } else if (sf.interaction == kMetallic) {
    weight = conductor_brdf(p.rng, sf.baseLinear, sf.viewDir, sf.roughness, lightDir);
  } else {
    weight = dielectric_bsdf(p.rng, sf.baseLinear, sf.viewDir, sf.roughness, sf.ior, lightDir, sf.outside);
  }
  return shade_end(p, sf, weight, lightDir, bounce, maxBounces, sample);
}

// ---- pixel ownership ---------------------------------------------------------------
// A context owns the tiles g with g % nRanks == rank.  Owned pixels live in a
// compact buffer: slot = localTile * tileW*tileH + (8x8 block, row-major) * 64 +
// (y%8)*8 + (x%8), so 64 consecutive slots are one 8x8 pixel block.
struct Partition {
  int W, H;
  int tileW, tileH;
  int tilesX, tilesY;
  int rank, nRanks;
  int nLocalTiles;   // ceil(tilesX*tilesY / nRanks), equal on every rank
  uint32_t nSlots;   // nLocalTiles * tileW * tileH
  int dense;         // 1 if every slot of this rank is a pixel inside the image
};

__host__ __device__ __forceinline__ bool slot_to_pixel(const Partition& P, uint32_t slot, int& x, int& y) {
  uint32_t perTile = (uint32_t)(P.tileW * P.tileH);
  uint32_t k = slot / perTile, r = slot - k * perTile;
  uint32_t g = k * (uint32_t)P.nRanks + (uint32_t)P.rank;
  if (g >= (uint32_t)(P.tilesX * P.tilesY)) return false;
  int tx = (int)(g % (uint32_t)P.tilesX), ty = (int)(g / (uint32_t)P.tilesX);
  uint32_t blk = r >> 6, in = r & 63u;
  uint32_t bpr = (uint32_t)(P.tileW >> 3);
  int bx = (int)(blk % bpr), by = (int)(blk / bpr);
  x = tx * P.tileW + bx * 8 + (int)(in & 7u);
  y = ty * P.tileH + by * 8 + (int)(in >> 3);
  return x < P.W && y < P.H;
}

// Inverse map: owner rank and slot of pixel (x,y).
__device__ __forceinline__ void pixel_to_slot(const Partition& P, int x, int y, int& owner, uint32_t& slot) {
  int tx = x / P.tileW, ty = y / P.tileH;
  uint32_t g = (uint32_t)(ty * P.tilesX + tx);
  owner = (int)(g % (uint32_t)P.nRanks);
  uint32_t k = g / (uint32_t)P.nRanks;
  int lx = x - tx * P.tileW, ly = y - ty * P.tileH;
  uint32_t blk = (uint32_t)((ly >> 3) * (P.tileW >> 3) + (lx >> 3));
  slot = k * (uint32_t)(P.tileW * P.tileH) + blk * 64u + (uint32_t)((ly & 7) * 8 + (lx & 7));
}

}  // namespace lpt
