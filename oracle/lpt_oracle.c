/*
 * CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See lpt_oracle.h.
 *
 * Every function below restates one function of the reference shaders and cites
 * it.  The order of floating-point operations is part of the contract with the
 * HIP kernels: do not "simplify" expressions here.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp (see oracle/Makefile).
 */
#include "lpt_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* shaders/common/constants.glsl:4-9 */
#define LPT_EPS 0.0001f
#define LPT_PI 3.141592653589f
#define LPT_INFINITY 3.4E+38f
/* shaders/heitz/BSDF.glsl:8 */
#define HEITZ_MAX_ORDER 16
/* shaders/path_tracing.comp:21 */
#define RUSSIAN_ROULETTE_BOUNCES 2
/* shaders/heitz/interaction_type.glsl:6-8 */
#define K_DIFF 1u
#define K_METALLIC 2u
#define K_TRANS 4u

typedef struct { float x, y, z; } v3;

/* ------------------------------------------------------------------------- */
/* Arithmetic contract helpers                                               */
/* ------------------------------------------------------------------------- */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* IEEE minNum / maxNum */
static inline float fmin_(float a, float b) { return (a != a) ? b : ((b < a) ? b : a); }
static inline float fmax_(float a, float b) { return (a != a) ? b : ((a < b) ? b : a); }
static inline float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) {
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length(v3 a) { return sqrtf(dot(a, a)); }
static inline v3 normalize(v3 a) { float s = 1.0f / sqrtf(dot(a, a)); return scale(a, s); }

/* column-major mat4 (m[col*4+row]) times (p,1) */
static inline v3 mat4_point(const float* m, v3 p) {
  v3 r;
  r.x = ((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12];
  r.y = ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13];
  r.z = ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14];
  return r;
}
/* mat3(m) * v */
static inline v3 mat3_vec(const float* m, v3 v) {
  v3 r;
  r.x = (m[0] * v.x + m[4] * v.y) + m[8] * v.z;
  r.y = (m[1] * v.x + m[5] * v.y) + m[9] * v.z;
  r.z = (m[2] * v.x + m[6] * v.y) + m[10] * v.z;
  return r;
}

/* --- fixed transcendental algorithms (single precision, Cephes-style minimax
 *     polynomials; only + - * / floor and bit operations) -------------------
 * GLSL leaves the precision of sin, cos, tan, log, exp and pow to the implementation
 * (the calls: heitz/BSDF.glsl:55-56,81, common/util.glsl:7, path_tracing.comp:119).
 * The contract with the HIP kernels fixes them as the polynomials below.  A second
 * build, -DLPTO_LIBM (build/liblpt_oracle_libm.so), calls the C library instead: a
 * different, equally legitimate implementation of the same shader, used by the
 * tests to show that the converged image does not depend on which one is taken. */

static void sincos_(float x, float* s, float* c) {
#ifdef LPTO_LIBM
  *s = sinf(x);
  *c = cosf(x);
  return;
#endif
  if (!(fabsf(x) <= 8192.0f)) { /* NaN, inf, out of the reduction's range */
    *s = *c = u2f(0x7fc00000u);
    return;
  }
  float t = x * 0.636619772f; /* 2/pi */
  int qi = (int)(t + (t < 0.0f ? -0.5f : 0.5f));
  float q = (float)qi;
  /* three-step Cody-Waite reduction by pi/2 */
  float r = ((x - q * 1.5703125f) - q * 4.837512969970703125e-4f) - q * 7.54978995489188216e-8f;
  float z = r * r;
  float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
  cp = cp - 0.5f * z;
  cp = cp + 1.0f;
  switch (qi & 3) {
    case 0: *s = sp; *c = cp; break;
    case 1: *s = cp; *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
  }
}

float lpto_sin(float x) { float s, c; sincos_(x, &s, &c); return s; }
float lpto_cos(float x) { float s, c; sincos_(x, &s, &c); return c; }
#ifdef LPTO_LIBM
float lpto_tan(float x) { return tanf(x); }
#else
float lpto_tan(float x) { float s, c; sincos_(x, &s, &c); return s / c; }
#endif

float lpto_log(float x) {
#ifdef LPTO_LIBM
  return logf(x);
#endif
  if (x != x) return x;
  if (x < 0.0f) return u2f(0x7fc00000u);
  if (x == 0.0f) return u2f(0xff800000u);
  uint32_t b = f2u(x);
  if (b == 0x7f800000u) return x;
  int e = 0;
  if (b < 0x00800000u) { /* denormal: scale by 2^24 (exact) */
    x = x * 16777216.0f;
    b = f2u(x);
    e = -24;
  }
  e += (int)(b >> 23) - 126;
  float m = u2f((b & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
  if (m < 0.707106781186547524f) {
    e -= 1;
    m = (m + m) - 1.0f;
  } else {
    m = m - 1.0f;
  }
  float z = m * m;
  float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m -
                  1.2420140846e-1f) * m + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m +
               2.0000714765e-1f) * m - 2.4999993993e-1f) * m + 3.3333331174e-1f) * m * z;
  float fe = (float)e;
  y = y + -2.12194440e-4f * fe;
  y = y + -0.5f * z;
  z = m + y;
  z = z + 0.693359375f * fe;
  return z;
}

float lpto_exp(float x) {
#ifdef LPTO_LIBM
  return expf(x);
#endif
  if (x != x) return x;
  if (x > 88.72283905206835f) return u2f(0x7f800000u);
  if (x < -103.278929903431851103f) return 0.0f;
  float z = floorf(1.44269504088896341f * x + 0.5f);
  x = x - z * 0.693359375f;
  x = x - z * -2.12194440e-4f;
  int n = (int)z;
  z = x * x;
  z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x +
        1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
  /* z * 2^n in two exact-or-singly-rounded steps */
  int n1 = n / 2;
  int n2 = n - n1;
  float p1 = u2f((uint32_t)(n1 + 127) << 23);
  float p2 = u2f((uint32_t)(n2 + 127) << 23);
  return (z * p1) * p2;
}

float lpto_pow(float x, float y) {
#ifdef LPTO_LIBM
  return powf(x, y);
#endif
  if (x != x || y != y) return u2f(0x7fc00000u);
  if (x < 0.0f) return u2f(0x7fc00000u);
  if (x == 0.0f) {
    if (y > 0.0f) return 0.0f;
    if (y == 0.0f) return 1.0f;
    return u2f(0x7f800000u);
  }
  return lpto_exp(y * lpto_log(x));
}

/* ------------------------------------------------------------------------- */
/* shaders/common/random.glsl                                                */
/* ------------------------------------------------------------------------- */

typedef struct {
  lpto_seed seed;   /* random.glsl:6 "uvec2 seed" */
  lpto_stats* st;
} path_ctx;

/* random.glsl:9-15 */
static inline float rand_(path_ctx* c) {
  c->seed.x += 1u;
  c->seed.y += 1u;
  uint32_t qx = 1103515245u * ((c->seed.x >> 1) ^ c->seed.y);
  uint32_t qy = 1103515245u * ((c->seed.y >> 1) ^ c->seed.x);
  uint32_t n = 1103515245u * (qx ^ (qy >> 3));
  c->st->rands++;
  /* 1.0 / float(0xffffffffU) == 2^-32 because float(0xffffffff) rounds to 2^32 */
  return (float)n * (1.0f / 4294967296.0f);
}

float lpto_rand(lpto_seed* seed) {
  lpto_stats st;
  memset(&st, 0, sizeof st);
  path_ctx c = {*seed, &st};
  float r = rand_(&c);
  *seed = c.seed;
  return r;
}

/* ------------------------------------------------------------------------- */
/* shaders/common/ray.glsl                                                   */
/* ------------------------------------------------------------------------- */

/* ray.glsl:6-9 plus the reciprocal direction, which ray.glsl:40 recomputes in
 * every box test from the same direction (same value every time). */
typedef struct { v3 origin, direction, invDir; } ray_t;

static inline ray_t make_ray(v3 o, v3 d) {
  ray_t r;
  r.origin = o;
  r.direction = d;
  r.invDir = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  return r;
}

/* ray.glsl:39-61 */
static inline int ray_aabb_test(const ray_t* ray, const float* mn, const float* mx, float distance) {
  v3 nearv = mul(sub(V(mn[0], mn[1], mn[2]), ray->origin), ray->invDir);
  v3 farv = mul(sub(V(mx[0], mx[1], mx[2]), ray->origin), ray->invDir);
  v3 tmin = V(fmin_(nearv.x, farv.x), fmin_(nearv.y, farv.y), fmin_(nearv.z, farv.z));
  v3 tmax = V(fmax_(nearv.x, farv.x), fmax_(nearv.y, farv.y), fmax_(nearv.z, farv.z));
  float t0 = fmax_(fmax_(tmin.x, tmin.y), tmin.z);
  float t1 = fmin_(fmin_(tmax.x, tmax.y), tmax.z);
  if (t0 > t1) return 0;
  if (t0 > 0.0f) return t0 < distance;
  return t1 > 0.0f;
}

/* ray.glsl:63-82 */
static inline float ray_triangle(const ray_t* ray, v3 v0, v3 v1, v3 v2) {
  v3 edge1 = sub(v1, v0);
  v3 edge2 = sub(v2, v0);
  v3 pvec = cross(ray->direction, edge2);
  float det = 1.0f / dot(edge1, pvec);
  v3 tvec = sub(ray->origin, v0);
  float u = dot(tvec, pvec) * det;
  if (u < 0.0f || u > 1.0f) return LPT_INFINITY;
  v3 qvec = cross(tvec, edge1);
  float v = dot(ray->direction, qvec) * det;
  if (v < 0.0f || u + v > 1.0f) return LPT_INFINITY;
  return dot(edge2, qvec) * det;
}

int lpto_ray_aabb_test(const float o[3], const float d[3], const float mn[3], const float mx[3],
                       float distance) {
  ray_t r = make_ray(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]));
  return ray_aabb_test(&r, mn, mx, distance);
}

float lpto_ray_triangle(const float o[3], const float d[3], const float v0[3], const float v1[3],
                        const float v2[3]) {
  ray_t r = make_ray(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]));
  return ray_triangle(&r, V(v0[0], v0[1], v0[2]), V(v1[0], v1[1], v1[2]), V(v2[0], v2[1], v2[2]));
}

/* ------------------------------------------------------------------------- */
/* shaders/common/util.glsl                                                  */
/* ------------------------------------------------------------------------- */

/* util.glsl:4-8 */
float lpto_srgb_to_linear(float srgb) {
  if (srgb <= 0.04045f) return srgb / 12.92f;
  return lpto_pow((srgb + 0.055f) / 1.055f, 2.4f);
}

/* util.glsl:23-41 */
static inline v3 barycentric(v3 point, v3 v0, v3 v1, v3 v2) {
  v3 ab = sub(v1, v0);
  v3 ac = sub(v2, v0);
  v3 ah = sub(point, v0);
  float ab_ab = dot(ab, ab);
  float ab_ac = dot(ab, ac);
  float ac_ac = dot(ac, ac);
  float ab_ah = dot(ab, ah);
  float ac_ah = dot(ac, ah);
  float inv_denom = 1.0f / (ab_ab * ac_ac - ab_ac * ab_ac);
  float v = (ac_ac * ab_ah - ab_ac * ac_ah) * inv_denom;
  float w = (ab_ab * ac_ah - ab_ac * ab_ah) * inv_denom;
  float u = (1.0f - v) - w;
  return V(u, v, w);
}

void lpto_barycentric(const float p[3], const float v0[3], const float v1[3], const float v2[3],
                      float out[3]) {
  v3 b = barycentric(V(p[0], p[1], p[2]), V(v0[0], v0[1], v0[2]), V(v1[0], v1[1], v1[2]),
                     V(v2[0], v2[1], v2[2]));
  out[0] = b.x; out[1] = b.y; out[2] = b.z;
}

/* ------------------------------------------------------------------------- */
/* shaders/heitz/BSDF.glsl                                                   */
/* ------------------------------------------------------------------------- */

/* BSDF.glsl:10-24 */
static inline float fresnel(float vdoth, float eta) {
  float cos_theta_t2 = 1.0f - (1.0f - vdoth * vdoth) / (eta * eta);
  if (cos_theta_t2 <= 0.0f) return 1.0f; /* total internal reflection */
  float cos_theta_t = sqrtf(cos_theta_t2);
  float Rs = (vdoth - eta * cos_theta_t) / (vdoth + eta * cos_theta_t);
  float Rp = (eta * vdoth - cos_theta_t) / (eta * vdoth + cos_theta_t);
  return 0.5f * (Rs * Rs + Rp * Rp);
}
float lpto_fresnel(float vdoth, float eta) { return fresnel(vdoth, eta); }

/* BSDF.glsl:26-32 */
static inline v3 refract_eta(v3 wi, v3 wm, float eta) {
  float cos_theta_i = dot(wi, wm);
  float cos_theta_t2 = 1.0f - (1.0f - cos_theta_i * cos_theta_i) / (eta * eta);
  float cos_theta_t = -sqrtf(fmax_(0.0f, cos_theta_t2));
  float k = dot(wi, wm) / eta + cos_theta_t;
  v3 a = scale(wm, k);
  v3 b = V(wi.x / eta, wi.y / eta, wi.z / eta);
  return sub(a, b);
}
void lpto_refract_eta(const float wi[3], const float wm[3], float eta, float out[3]) {
  v3 r = refract_eta(V(wi[0], wi[1], wi[2]), V(wm[0], wm[1], wm[2]), eta);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* BSDF.glsl:41-67 */
static v3 sample_ggx_vndf(path_ctx* c, v3 Ve, float alpha) {
  float R1 = rand_(c);
  float R2 = rand_(c);
  v3 Vh = normalize(V(alpha * Ve.x, alpha * Ve.y, Ve.z));
  v3 T1 = (Vh.z < 1.0f) ? normalize(cross(V(0.0f, 0.0f, 1.0f), Vh)) : V(1.0f, 0.0f, 0.0f);
  v3 T2 = cross(Vh, T1);
  float r = sqrtf(R1);
  float phi = (2.0f * LPT_PI) * R2;
  float sn, cs;
  sincos_(phi, &sn, &cs);
  float t1 = r * cs;
  float t2 = r * sn;
  float s = 0.5f * (1.0f + Vh.z);
  t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
  float nz = sqrtf(fmax_(0.0f, (1.0f - t1 * t1) - t2 * t2));
  v3 Nh = add(add(scale(T1, t1), scale(T2, t2)), scale(Vh, nz));
  return normalize(V(alpha * Nh.x, alpha * Nh.y, fmax_(0.0f, Nh.z)));
}
void lpto_sample_ggx_vndf(const float ve[3], float alpha, lpto_seed* seed, float out[3]) {
  lpto_stats st; memset(&st, 0, sizeof st);
  path_ctx c = {*seed, &st};
  v3 n = sample_ggx_vndf(&c, V(ve[0], ve[1], ve[2]), alpha);
  *seed = c.seed;
  out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

/* BSDF.glsl:72-84 */
static float sample_ggx_height(path_ctx* c, v3 dir, float height, float alpha) {
  float len = length(mul(dir, V(alpha, alpha, 1.0f)));
  float projectedArea = fmax_(0.5f * (len - dir.z), 1e-7f);
  float R = rand_(c);
  float delta = -lpto_log(1.0f - R) * dir.z / projectedArea;
  return height + delta;
}
float lpto_sample_ggx_height(const float dir[3], float height, float alpha, lpto_seed* seed) {
  lpto_stats st; memset(&st, 0, sizeof st);
  path_ctx c = {*seed, &st};
  float h = sample_ggx_height(&c, V(dir[0], dir[1], dir[2]), height, alpha);
  *seed = c.seed;
  return h;
}

/* BSDF.glsl:88-104 (weight is F0; the Schlick term is commented out at :101) */
static v3 sample_conductor_phase(path_ctx* c, v3 viewDir, float alpha) {
  v3 microNormal = sample_ggx_vndf(c, viewDir, alpha);
  float vdoth = dot(viewDir, microNormal);
  vdoth = clamp_(vdoth, 0.0f, 1.0f);
  return sub(scale(scale(microNormal, 2.0f), vdoth), viewDir);
}

/* BSDF.glsl:108-138 */
static v3 conductor_brdf(path_ctx* c, v3 F0, v3 viewDir, float roughness, v3* lightDir) {
  float alpha = roughness * roughness;
  v3 energy = V(1.0f, 1.0f, 1.0f);
  *lightDir = neg(viewDir);
  float height = 0.0f;
  int order = 0;
  while (order < HEITZ_MAX_ORDER) {
    c->st->walkSteps++;
    height = sample_ggx_height(c, *lightDir, height, alpha);
    if (height > 0.0f) break;
    *lightDir = sample_conductor_phase(c, neg(*lightDir), alpha);
    energy = mul(energy, F0);
    order++;
  }
  return energy;
}

/* BSDF.glsl:140-158 */
static v3 sample_dielectric_phase(path_ctx* c, v3 viewDir, float alpha, float eta, int* outside) {
  v3 microNormal = sample_ggx_vndf(c, viewDir, alpha);
  float vdoth = dot(viewDir, microNormal);
  float F = fresnel(vdoth, eta);
  if (rand_(c) < F) {
    return sub(scale(scale(microNormal, 2.0f), vdoth), viewDir);
  } else {
    *outside = !*outside;
    return normalize(refract_eta(viewDir, microNormal, eta));
  }
}

/* BSDF.glsl:160-209 (the transmittance argument is unused there) */
static v3 dielectric_bsdf(path_ctx* c, v3 F0, v3 viewDir, float roughness, float ior, v3* lightDir,
                          int outside) {
  float alpha = roughness * roughness;
  *lightDir = neg(viewDir);
  float height = 0.0f;
  float iorOut, iorIn;
  if (outside) { iorOut = 1.0f; iorIn = ior; } else { iorOut = ior; iorIn = 1.0f; }
  outside = 1;
  int order = 0;
  while (order < HEITZ_MAX_ORDER) {
    c->st->walkSteps++;
    if (outside) {
      height = sample_ggx_height(c, *lightDir, height, alpha);
      if (height > 0.0f) break;
    } else {
      height = -sample_ggx_height(c, neg(*lightDir), -height, alpha);
      if (height < 0.0f) break;
    }
    *lightDir = sample_dielectric_phase(c, neg(*lightDir), alpha,
                                        outside ? iorIn / iorOut : iorOut / iorIn, &outside);
    order++;
  }
  return F0;
}

/* BSDF.glsl:211-239 */
static v3 sample_diffuse_phase(path_ctx* c, v3 viewDir, float alpha) {
  v3 microNormal = sample_ggx_vndf(c, viewDir, alpha);
  v3 u = (microNormal.z < 1.0f) ? normalize(cross(V(0.0f, 0.0f, 1.0f), microNormal)) : V(1.0f, 0.0f, 0.0f);
  v3 v = cross(microNormal, u);
  float R1 = 2.0f * rand_(c) - 1.0f;
  float R2 = 2.0f * rand_(c) - 1.0f;
  float phi, R;
  if (R1 == 0.0f && R2 == 0.0f) {
    R = phi = 0.0f;
  } else if (R1 * R1 > R2 * R2) {
    R = R1;
    phi = (LPT_PI / 4.0f) * (R2 / R1);
  } else {
    R = R2;
    phi = (LPT_PI / 2.0f) - (R1 / R2) * (LPT_PI / 4.0f);
  }
  float sn, cs;
  sincos_(phi, &sn, &cs);
  float x = R * cs;
  float y = R * sn;
  float z = sqrtf(fmax_(0.0f, (1.0f - x * x) - y * y));
  return add(add(scale(u, x), scale(v, y)), scale(microNormal, z));
}

/* BSDF.glsl:241-275 */
static v3 diffuse_bsdf(path_ctx* c, v3 F0, v3 viewDir, float roughness, v3* lightDir) {
  float alpha = roughness * roughness;
  v3 energy = V(1.0f, 1.0f, 1.0f);
  *lightDir = neg(viewDir);
  float height = 0.0f;
  int order = 0;
  while (order < HEITZ_MAX_ORDER) {
    c->st->walkSteps++;
    height = sample_ggx_height(c, *lightDir, height, alpha);
    if (height > 0.0f) break;
    *lightDir = sample_diffuse_phase(c, neg(*lightDir), alpha);
    energy = mul(energy, F0);
    order++;
  }
  if (order >= HEITZ_MAX_ORDER) {
    *lightDir = V(0.0f, 0.0f, 1.0f);
    return V(0.0f, 0.0f, 0.0f);
  }
  return energy;
}

/* ------------------------------------------------------------------------- */
/* shaders/basic/BSDF.glsl                                                   */
/* ------------------------------------------------------------------------- */

/* GLSL reflect(I,N) = I - 2*dot(N,I)*N ; refract per the GLSL specification. */
static inline v3 glsl_reflect(v3 I, v3 N) { return sub(I, scale(N, 2.0f * dot(N, I))); }
static inline v3 glsl_refract(v3 I, v3 N, float eta) {
  float ndi = dot(N, I);
  float k = 1.0f - (eta * eta) * (1.0f - ndi * ndi);
  if (k < 0.0f) return V(0.0f, 0.0f, 0.0f);
  return sub(scale(I, eta), scale(N, eta * ndi + sqrtf(k)));
}

/* basic/BSDF.glsl:3-11 */
static v3 basic_diffuse(path_ctx* c, v3 F0, v3* lightDir) {
  float r1 = (2.0f * LPT_PI) * rand_(c);
  float r2 = rand_(c);
  float r2s = sqrtf(r2);
  float sn, cs;
  sincos_(r1, &sn, &cs);
  *lightDir = V(cs * r2s, sn * r2s, sqrtf(1.0f - r2));
  return scale(F0, dot(*lightDir, V(0.0f, 0.0f, 1.0f)));
}

/* basic/BSDF.glsl:13-17 */
static v3 basic_specular(v3 F0, v3 viewDir, v3* lightDir) {
  *lightDir = glsl_reflect(neg(viewDir), V(0.0f, 0.0f, 1.0f));
  return F0;
}

/* basic/BSDF.glsl:19-31 */
static float basic_fresnel_reflectance(v3 n, v3 nl, v3 rayDirection, float nc, float nt, v3* tdir) {
  float nnt = dot(rayDirection, n) < 0.0f ? (nc / nt) : (nt / nc);
  *tdir = glsl_refract(rayDirection, nl, nnt);
  float cosThetaInc = dot(nl, rayDirection);
  float cosThetaTra = dot(nl, *tdir);
  float coefPara = (nt * cosThetaInc - nc * cosThetaTra) / (nt * cosThetaInc + nc * cosThetaTra);
  float coefPerp = (nc * cosThetaInc - nt * cosThetaTra) / (nc * cosThetaInc + nt * cosThetaTra);
  return (coefPara * coefPara + coefPerp * coefPerp) * 0.5f;
}

/* basic/BSDF.glsl:33-50 */
static v3 basic_transmittance(path_ctx* c, v3 F0, v3 viewDir, float transmittance, float ior,
                              int outside, v3* lightDir) {
  v3 tdir;
  v3 normal = outside ? V(0.0f, 0.0f, 1.0f) : V(0.0f, 0.0f, -1.0f);
  float Re = basic_fresnel_reflectance(normal, V(0.0f, 0.0f, 1.0f), neg(viewDir), 1.0f, ior, &tdir);
  float r = rand_(c);
  if (r < Re) {
    *lightDir = glsl_reflect(neg(viewDir), normal);
    return V(1.0f, 1.0f, 1.0f);
  } else {
    *lightDir = tdir;
    return scale(F0, transmittance);
  }
}

/* ------------------------------------------------------------------------- */
/* shaders/heitz/interaction_type.glsl:10-29                                 */
/* ------------------------------------------------------------------------- */

static uint32_t interaction_type(path_ctx* c, float metallicFactor, float transmissionFactor) {
  float metallicBRDF = metallicFactor;
  float transmissionBSDF = (1.0f - metallicFactor) * transmissionFactor;
  float dielectricBRDF = (1.0f - transmissionFactor) * (1.0f - metallicFactor);
  float norm = 1.0f / ((metallicBRDF + transmissionBSDF) + dielectricBRDF);
  metallicBRDF *= norm;
  transmissionBSDF *= norm;
  dielectricBRDF *= norm;
  (void)dielectricBRDF;
  float r = rand_(c);
  if (r < metallicBRDF) return K_METALLIC;
  if (r < metallicBRDF + transmissionBSDF) return K_TRANS;
  return K_DIFF;
}

uint32_t lpto_interaction_type(float metallic, float transmission, lpto_seed* seed) {
  lpto_stats st; memset(&st, 0, sizeof st);
  path_ctx c = {*seed, &st};
  uint32_t k = interaction_type(&c, metallic, transmission);
  *seed = c.seed;
  return k;
}

void lpto_bsdf(int mode, uint32_t kind, const float F0[3], const float viewDir[3], float roughness,
               float transmission, float ior, int outside, lpto_seed* seed, float lightDir[3],
               float weight[3]) {
  lpto_stats st; memset(&st, 0, sizeof st);
  path_ctx c = {*seed, &st};
  v3 f0 = V(F0[0], F0[1], F0[2]), vd = V(viewDir[0], viewDir[1], viewDir[2]);
  v3 ld = V(0.0f, 0.0f, 0.0f), w = V(0.0f, 0.0f, 0.0f);
  if (mode == LPTO_BSDF_HEITZ) {
    if (kind == K_DIFF) w = diffuse_bsdf(&c, f0, vd, roughness, &ld);
    else if (kind == K_METALLIC) w = conductor_brdf(&c, f0, vd, roughness, &ld);
    else if (kind == K_TRANS) w = dielectric_bsdf(&c, f0, vd, roughness, ior, &ld, outside);
  } else {
    if (kind == K_DIFF) w = basic_diffuse(&c, f0, &ld);
    else if (kind == K_METALLIC) w = basic_specular(f0, vd, &ld);
    else if (kind == K_TRANS) w = basic_transmittance(&c, f0, vd, transmission, ior, outside, &ld);
  }
  *seed = c.seed;
  lightDir[0] = ld.x; lightDir[1] = ld.y; lightDir[2] = ld.z;
  weight[0] = w.x; weight[1] = w.y; weight[2] = w.z;
}

/* ------------------------------------------------------------------------- */
/* shaders/path_tracing.comp                                                 */
/* ------------------------------------------------------------------------- */

typedef struct { float distance; uint32_t objectIndex; uint32_t primitiveIndex; } isect_t; /* :65-69 */

/* ---- texture(sampler2D, uv), path_tracing.comp:244-261,277-282 ------------------------- */
/* Mip level 0 (no derivatives in a compute shader), magnification filter, address modes of
 * src/PTSceneConverter.cpp:355-360. */

static inline int tex_floor_index(float x) {
  float f = floorf(x);
  if (!(f == f)) f = 0.0f;
  if (f < -1073741824.0f) f = -1073741824.0f;
  if (f > 1073741824.0f) f = 1073741824.0f;
  return (int)f;
}

/* returns the texel index, or -1 for the border colour */
static inline int tex_wrap(int i, int n, uint32_t mode) {
  switch (mode) {
    case 0u: { int r = i % n; return r < 0 ? r + n : r; }
    case 1u: { int m = i % (2 * n); if (m < 0) m += 2 * n; return m < n ? m : 2 * n - 1 - m; }
    case 2u: return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    default: return (i < 0 || i > n - 1) ? -1 : i;
  }
}

static inline void tex_texel(const lpto_texture* t, int i, int j, float out[4]) {
  if (i < 0 || j < 0) { out[0] = out[1] = out[2] = 0.0f; out[3] = 1.0f; return; } /* eFloatOpaqueBlack */
  const uint8_t* p = t->pixels + 4u * ((size_t)j * t->width + (size_t)i);
  for (int k = 0; k < 4; k++) out[k] = (float)p[k] / 255.0f;
}

void lpto_sample_texture(const lpto_texture* t, float u, float v, float out[4]) {
  const int W = (int)t->width, H = (int)t->height;
  if (t->magFilter == 0u) {
    int i = tex_wrap(tex_floor_index(u * (float)W), W, t->addressU);
    int j = tex_wrap(tex_floor_index(v * (float)H), H, t->addressV);
    tex_texel(t, i, j, out);
    return;
  }
  float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
  float a = x - floorf(x), b = y - floorf(y);
  if (!(a == a)) a = 0.0f;
  if (!(b == b)) b = 0.0f;
  int i0 = tex_floor_index(x), j0 = tex_floor_index(y);
  int ia = tex_wrap(i0, W, t->addressU), ib = tex_wrap(i0 + 1, W, t->addressU);
  int ja = tex_wrap(j0, H, t->addressV), jb = tex_wrap(j0 + 1, H, t->addressV);
  float t00[4], t10[4], t01[4], t11[4];
  tex_texel(t, ia, ja, t00);
  tex_texel(t, ib, ja, t10);
  tex_texel(t, ia, jb, t01);
  tex_texel(t, ib, jb, t11);
  for (int k = 0; k < 4; k++) {
    float top = t00[k] * (1.0f - a) + t10[k] * a;
    float bot = t01[k] * (1.0f - a) + t11[k] * a;
    out[k] = top * (1.0f - b) + bot * b;
  }
}

#define LPT_NO_TEXTURE 0xFFFFFFFFu

static inline v3 vpos(const lpto_vertex* v) { return V(v->position[0], v->position[1], v->position[2]); }
static inline v3 vnrm(const lpto_vertex* v) { return V(v->normal[0], v->normal[1], v->normal[2]); }

/* path_tracing.comp:107-127 */
static ray_t generate_ray(path_ctx* c, const lpto_ubo* ubo, int px, int py, float resx, float resy) {
  float r1 = 2.0f * rand_(c);
  float r2 = 2.0f * rand_(c);
  float jx = r1 < 1.0f ? sqrtf(r1) - 1.0f : 1.0f - sqrtf(2.0f - r1);
  float jy = r2 < 1.0f ? sqrtf(r2) - 1.0f : 1.0f - sqrtf(2.0f - r2);
  jx = jx / (resx * 0.5f);
  jy = jy / (resy * 0.5f);
  float uvx = ((2.0f * (float)px) / resx - 1.0f) + jx;
  float uvy = ((2.0f * (float)py) / resy - 1.0f) + jy;
  const float* m = ubo->worldMatrix;
  v3 origin = V(m[12], m[13], m[14]);
  float aspectRatio = resx / resy;
  float th = lpto_tan(ubo->fovY / 2.0f);
  uvx = uvx * (aspectRatio * th);
  uvy = uvy * th;
  v3 c0 = V(m[0], m[1], m[2]), c1 = V(m[4], m[5], m[6]), c2 = V(m[8], m[9], m[10]);
  v3 dir = normalize(sub(add(scale(c0, uvx), scale(c1, uvy)), c2));
  return make_ray(origin, dir);
}

void lpto_generate_ray(const lpto_ubo* ubo, int px, int py, int resx, int resy, lpto_seed* seed,
                       float o[3], float d[3]) {
  lpto_stats st; memset(&st, 0, sizeof st);
  path_ctx c = {*seed, &st};
  ray_t r = generate_ray(&c, ubo, px, py, (float)resx, (float)resy);
  *seed = c.seed;
  o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
  d[0] = r.direction.x; d[1] = r.direction.y; d[2] = r.direction.z;
}

/* Optional visit histograms (analysis aid): one counter per object-BVH / mesh-BVH node,
 * incremented whenever traversal visits that node. */
static uint32_t* g_hist_tlas = 0;
static uint32_t* g_hist_blas = 0;
void lpto_set_visit_histograms(uint32_t* tlas, uint32_t* blas) { g_hist_tlas = tlas; g_hist_blas = blas; }

/* path_tracing.comp:130-173 */
static void object_intersect(const lpto_scene* sc, const ray_t* ray, uint32_t objectIndex, isect_t* isect,
                             lpto_stats* st) {
  const lpto_object* obj = &sc->objects[objectIndex];
  int bvhOffset = (int)obj->bvhOffset;
  uint32_t verticesOffset = obj->verticesOffset;
  st->objectsEntered++;
  st->rootVisits++;

  /* ray to object space; the direction is not renormalised, so the ray
   * parameter means the same distance in both spaces (:136-137) */
  ray_t rayObj = make_ray(mat4_point(obj->worldMatrixInverse, ray->origin),
                          mat3_vec(obj->worldMatrixInverse, ray->direction));

  uint32_t ptr = 0;
  int stack[LPTO_STACK];
  stack[ptr++] = -1;
  int idx = bvhOffset;
  while (idx > -1) {
    const lpto_bvh_node* node = &sc->meshBvh[idx];
    if (g_hist_blas) {
#pragma omp atomic
      g_hist_blas[idx]++;
    }
    if (node->isLeaf) {
      for (uint32_t i = node->indices[0]; i < node->indices[1]; i++) {
        uint32_t firstVertexIdx = verticesOffset + 3u * i;
        st->triTests++;
        float triDistance = ray_triangle(&rayObj, vpos(&sc->vertices[firstVertexIdx]),
                                         vpos(&sc->vertices[firstVertexIdx + 1]),
                                         vpos(&sc->vertices[firstVertexIdx + 2]));
        if (triDistance > LPT_EPS && triDistance < isect->distance) {
          isect->distance = triDistance;
          isect->objectIndex = objectIndex;
          isect->primitiveIndex = firstVertexIdx;
        }
      }
    } else {
      int testIdx = (int)((uint32_t)bvhOffset + node->indices[0]);
      st->boxTests += 2;
      if (ray_aabb_test(&rayObj, sc->meshBvh[testIdx].minCorner, sc->meshBvh[testIdx].maxCorner,
                        isect->distance)) {
        if (ptr < LPTO_STACK) stack[ptr++] = testIdx; else st->stackOverflow = 1;
      }
      testIdx = (int)((uint32_t)bvhOffset + node->indices[1]);
      if (ray_aabb_test(&rayObj, sc->meshBvh[testIdx].minCorner, sc->meshBvh[testIdx].maxCorner,
                        isect->distance)) {
        if (ptr < LPTO_STACK) stack[ptr++] = testIdx; else st->stackOverflow = 1;
      }
    }
    if (ptr > st->maxStackBlas) st->maxStackBlas = ptr;
    idx = stack[--ptr];
  }
}

/* path_tracing.comp:175-209 */
static isect_t scene_intersect(const lpto_scene* sc, const ray_t* ray, lpto_stats* st) {
  isect_t isect;
  isect.distance = LPT_INFINITY;
  isect.objectIndex = 0;
  isect.primitiveIndex = 0;
  st->rootVisits++;

  uint32_t ptr = 0;
  int stack[LPTO_STACK];
  stack[ptr++] = -1;
  int idx = 0;
  while (idx > -1) {
    const lpto_bvh_node* node = &sc->objectBvh[idx];
    if (g_hist_tlas) {
#pragma omp atomic
      g_hist_tlas[idx]++;
    }
    if (node->isLeaf) {
      for (uint32_t i = node->indices[0]; i < node->indices[1]; i++) {
        object_intersect(sc, ray, i, &isect, st);
      }
    } else {
      int testIdx = (int)node->indices[0];
      st->boxTests += 2;
      if (ray_aabb_test(ray, sc->objectBvh[testIdx].minCorner, sc->objectBvh[testIdx].maxCorner,
                        isect.distance)) {
        if (ptr < LPTO_STACK) stack[ptr++] = testIdx; else st->stackOverflow = 1;
      }
      testIdx = (int)node->indices[1];
      if (ray_aabb_test(ray, sc->objectBvh[testIdx].minCorner, sc->objectBvh[testIdx].maxCorner,
                        isect.distance)) {
        if (ptr < LPTO_STACK) stack[ptr++] = testIdx; else st->stackOverflow = 1;
      }
    }
    if (ptr > st->maxStackTlas) st->maxStackTlas = ptr;
    idx = stack[--ptr];
  }
  return isect;
}

int lpto_scene_intersect(const lpto_scene* scene, const float o[3], const float d[3], float* t,
                         uint32_t* objectIndex, uint32_t* primitiveIndex, lpto_stats* stats) {
  lpto_stats local; memset(&local, 0, sizeof local);
  lpto_stats* st = stats ? stats : &local;
  ray_t r = make_ray(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]));
  isect_t is = scene_intersect(scene, &r, st);
  *t = is.distance;
  *objectIndex = is.objectIndex;
  *primitiveIndex = is.primitiveIndex;
  return is.distance != LPT_INFINITY;
}

int lpto_brute_force_intersect(const lpto_scene* sc, const float o[3], const float d[3], float* t,
                               uint32_t* objectIndex, uint32_t* primitiveIndex) {
  ray_t ray = make_ray(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]));
  isect_t is;
  is.distance = LPT_INFINITY;
  is.objectIndex = 0;
  is.primitiveIndex = 0;
  for (uint32_t oi = 0; oi < sc->nObjects; oi++) {
    const lpto_object* obj = &sc->objects[oi];
    ray_t rayObj = make_ray(mat4_point(obj->worldMatrixInverse, ray.origin),
                            mat3_vec(obj->worldMatrixInverse, ray.direction));
    /* the object's vertices run from its verticesOffset to the next larger offset */
    uint32_t end = sc->nVertices;
    for (uint32_t oj = 0; oj < sc->nObjects; oj++) {
      uint32_t vo = sc->objects[oj].verticesOffset;
      if (vo > obj->verticesOffset && vo < end) end = vo;
    }
    for (uint32_t v = obj->verticesOffset; v + 2 < end; v += 3) {
      float td = ray_triangle(&rayObj, vpos(&sc->vertices[v]), vpos(&sc->vertices[v + 1]),
                              vpos(&sc->vertices[v + 2]));
      if (td > LPT_EPS && td < is.distance) {
        is.distance = td;
        is.objectIndex = oi;
        is.primitiveIndex = v;
      }
    }
  }
  *t = is.distance;
  *objectIndex = is.objectIndex;
  *primitiveIndex = is.primitiveIndex;
  return is.distance != LPT_INFINITY;
}

/* path_tracing.comp:211-327 */
static v3 trace_ray(const lpto_scene* sc, path_ctx* c, ray_t ray, int maxBounces, int bsdfMode) {
  v3 accColor = V(0.0f, 0.0f, 0.0f);
  v3 mask = V(1.0f, 1.0f, 1.0f);
  lpto_stats* st = c->st;

  for (uint32_t bounce = 0; bounce < (uint32_t)maxBounces; bounce++) {
    st->segments++;
    const uint64_t stepsBefore = st->boxTests / 2 + st->triTests + st->objectsEntered;
    isect_t isect = scene_intersect(sc, &ray, st);
    const uint64_t steps = st->boxTests / 2 + st->triTests + st->objectsEntered - stepsBefore;
    if (steps > st->maxSegmentSteps) st->maxSegmentSteps = (uint32_t)steps;

    if (isect.distance == LPT_INFINITY) { /* :220-223 — assignment, not accumulation */
      accColor = scale(mask, 0.2f);
      break;
    }
    st->hits++;

    const lpto_object* object = &sc->objects[isect.objectIndex];
    const lpto_vertex* vtx = &sc->vertices[isect.primitiveIndex];

    v3 rayObjOrigin = mat4_point(object->worldMatrixInverse, ray.origin);
    v3 rayObjDirection = mat3_vec(object->worldMatrixInverse, ray.direction);

    v3 isectPositionWorld = add(ray.origin, scale(ray.direction, isect.distance));
    v3 bary = barycentric(add(rayObjOrigin, scale(rayObjDirection, isect.distance)), vpos(&vtx[0]),
                          vpos(&vtx[1]), vpos(&vtx[2]));
    /* :234 */
    float uvx = (bary.x * vtx[0].uv[0] + bary.y * vtx[1].uv[0]) + bary.z * vtx[2].uv[0];
    float uvy = (bary.x * vtx[0].uv[1] + bary.y * vtx[1].uv[1]) + bary.z * vtx[2].uv[1];

    float baseColor[4] = {object->baseColorFactor[0], object->baseColorFactor[1],
                          object->baseColorFactor[2], object->baseColorFactor[3]};
    v3 emissionFactor = V(object->emissionFactor[0], object->emissionFactor[1], object->emissionFactor[2]);
    float roughnessFactor = fmax_(object->roughnessFactor, 0.001f);
    float metallicFactor = object->metallicFactor;
    float transmissionFactor = object->transmissionFactor;
    float ior = object->ior;

    float tx[4];
    if (object->colorTexture != LPT_NO_TEXTURE) { /* :245-247 */
      lpto_sample_texture(&sc->textures[object->colorTexture], uvx, uvy, tx);
      for (int k = 0; k < 4; k++) baseColor[k] = baseColor[k] * tx[k];
    }
    if (object->emissionTexture != LPT_NO_TEXTURE) { /* :249-251 */
      lpto_sample_texture(&sc->textures[object->emissionTexture], uvx, uvy, tx);
      emissionFactor = mul(emissionFactor, V(tx[0], tx[1], tx[2]));
    }
    if (object->metallicRoughnessTexture != LPT_NO_TEXTURE) { /* :253-257 */
      lpto_sample_texture(&sc->textures[object->metallicRoughnessTexture], uvx, uvy, tx);
      metallicFactor = metallicFactor * tx[2];
      roughnessFactor = roughnessFactor * tx[1];
    }
    if (object->transmissionTexture != LPT_NO_TEXTURE) { /* :259-261 */
      lpto_sample_texture(&sc->textures[object->transmissionTexture], uvx, uvy, tx);
      transmissionFactor = transmissionFactor * tx[0];
    }

    /* :263 — SRGBToLinear(vec4); only xyz is used afterwards */
    v3 baseLinear = V(lpto_srgb_to_linear(baseColor[0]), lpto_srgb_to_linear(baseColor[1]),
                      lpto_srgb_to_linear(baseColor[2]));

    uint32_t interaction = interaction_type(c, metallicFactor, transmissionFactor); /* :266 */

    accColor = add(accColor, mul(mask, emissionFactor)); /* :269 */

    /* :272-275 */
    v3 nsum = add(add(scale(vnrm(&vtx[0]), bary.x), scale(vnrm(&vtx[1]), bary.y)), scale(vnrm(&vtx[2]), bary.z));
    v3 normal = normalize(mat3_vec(object->worldMatrix, nsum));
    v3 ffNormal = (dot(normal, ray.direction) < 0.0f) ? normal : scale(normal, -1.0f);
    v3 u = normalize(cross((fabsf(ffNormal.x) > 0.1f ? V(0.0f, 1.0f, 0.0f) : V(1.0f, 0.0f, 0.0f)), ffNormal));
    v3 v = cross(ffNormal, u);

    if (object->normalTexture != LPT_NO_TEXTURE) { /* :277-282 */
      lpto_sample_texture(&sc->textures[object->normalTexture], uvx, uvy, tx);
      v3 tn = normalize(V(tx[0] * 2.0f - 1.0f, tx[1] * 2.0f - 1.0f, tx[2] * 2.0f - 1.0f));
      ffNormal = normalize(add(add(scale(u, tn.x), scale(v, tn.y)), scale(ffNormal, tn.z)));
      u = normalize(cross((fabsf(ffNormal.x) > 0.1f ? V(0.0f, 1.0f, 0.0f) : V(1.0f, 0.0f, 0.0f)), ffNormal));
      v = cross(ffNormal, u);
    }

    v3 nd = neg(ray.direction);
    v3 viewDir = V(dot(nd, u), dot(nd, v), dot(nd, ffNormal)); /* :286-288 */
    v3 lightDir = V(0.0f, 0.0f, 0.0f);

    if (interaction == K_DIFF) {
      mask = mul(mask, bsdfMode == LPTO_BSDF_HEITZ ? diffuse_bsdf(c, baseLinear, viewDir, roughnessFactor, &lightDir)
                                                   : basic_diffuse(c, baseLinear, &lightDir));
    } else if (interaction == K_METALLIC) {
      mask = mul(mask, bsdfMode == LPTO_BSDF_HEITZ ? conductor_brdf(c, baseLinear, viewDir, roughnessFactor, &lightDir)
                                                   : basic_specular(baseLinear, viewDir, &lightDir));
    } else if (interaction == K_TRANS) {
      int outside = dot(normal, nd) > 0.0f; /* :304 — unflipped normal */
      mask = mul(mask, bsdfMode == LPTO_BSDF_HEITZ
                         ? dielectric_bsdf(c, baseLinear, viewDir, roughnessFactor, ior, &lightDir, outside)
                         : basic_transmittance(c, baseLinear, viewDir, transmissionFactor, ior, outside, &lightDir));
    }

    /* :312-315 */
    lightDir = add(add(scale(u, lightDir.x), scale(v, lightDir.y)), scale(ffNormal, lightDir.z));
    ray = make_ray(isectPositionWorld, lightDir);

    /* :317-323 */
    float q = fmax_(fmax_(mask.x, mask.y), mask.z);
    if (q < 0.5f && bounce > RUSSIAN_ROULETTE_BOUNCES) {
      if (rand_(c) > q) break;
      mask = scale(mask, 1.0f / q);
    }
  }
  return accColor;
}

/* path_tracing.comp:330-351 */
int lpto_render_frame(const lpto_scene* sc, const lpto_ubo* ubo, const lpto_params* p, float* accum,
                      lpto_stats* stats) {
  for (uint32_t i = 0; i < sc->nObjects; i++) {
    const lpto_object* o = &sc->objects[i];
    const uint32_t ids[5] = {o->colorTexture, o->emissionTexture, o->metallicRoughnessTexture,
                             o->transmissionTexture, o->normalTexture};
    for (int k = 0; k < 5; k++)
      if (ids[k] != LPT_NO_TEXTURE && ids[k] >= sc->nTextures) return -2; /* texture id out of range */
  }
  for (uint32_t i = 0; i < sc->nTextures; i++)
    if (!sc->textures[i].pixels || sc->textures[i].width == 0 || sc->textures[i].height == 0 ||
        sc->textures[i].width > 32768u || sc->textures[i].height > 32768u)
      return -2;
  if (p->x0 < 0 || p->y0 < 0 || p->x0 + p->w > p->width || p->y0 + p->h > p->height) return -1;

  lpto_stats total;
  memset(&total, 0, sizeof total);
  const float resx = (float)p->width, resy = (float)p->height;
#ifdef _OPENMP
  int nthreads = p->threads > 0 ? p->threads : omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
  {
    lpto_stats st;
    memset(&st, 0, sizeof st);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int y = p->y0; y < p->y0 + p->h; y++) {
      for (int x = p->x0; x < p->x0 + p->w; x++) {
        path_ctx c;
        c.st = &st;
        c.seed.x = ubo->seed[0] * (uint32_t)x; /* :341 */
        c.seed.y = ubo->seed[1] * (uint32_t)y;
        st.paths++;
        ray_t ray = generate_ray(&c, ubo, x, y, resx, resy);
        v3 sampleColor = trace_ray(sc, &c, ray, p->maxBounces, p->bsdfMode);
        float* px = accum + 4 * ((size_t)y * (size_t)p->width + (size_t)x);
        if (ubo->reset) { /* :347-351 */
          px[0] = sampleColor.x; px[1] = sampleColor.y; px[2] = sampleColor.z; px[3] = 1.0f;
        } else {
          px[0] = px[0] + sampleColor.x; px[1] = px[1] + sampleColor.y;
          px[2] = px[2] + sampleColor.z; px[3] = px[3] + 1.0f;
        }
      }
    }
#ifdef _OPENMP
#pragma omp critical
#endif
    {
      total.paths += st.paths; total.segments += st.segments; total.hits += st.hits;
      total.boxTests += st.boxTests; total.rootVisits += st.rootVisits; total.triTests += st.triTests;
      total.objectsEntered += st.objectsEntered; total.walkSteps += st.walkSteps; total.rands += st.rands;
      if (st.maxStackTlas > total.maxStackTlas) total.maxStackTlas = st.maxStackTlas;
      if (st.maxStackBlas > total.maxStackBlas) total.maxStackBlas = st.maxStackBlas;
      total.stackOverflow |= st.stackOverflow;
      if (st.maxSegmentSteps > total.maxSegmentSteps) total.maxSegmentSteps = st.maxSegmentSteps;
    }
  }
  if (stats) *stats = total;
  return total.stackOverflow ? -3 : 0;
}

/* shaders/tex_to_quad.frag:24-33 (display row r shows accumulation row height-1-r) */
void lpto_resolve(const float* accum, int width, int height, float invSampleCount, float* out) {
  const float exposure = 1.5f;
  const float invGamma = 1.0f / 2.2f;
  for (int y = 0; y < height; y++) {
    const float* src = accum + 4 * (size_t)(height - 1 - y) * (size_t)width;
    float* dst = out + 4 * (size_t)y * (size_t)width;
    for (int x = 0; x < width; x++) {
      for (int ch = 0; ch < 3; ch++) {
        float hdr = src[4 * x + ch] * invSampleCount;
        float mapped = 1.0f - lpto_exp(-hdr * exposure);
        dst[4 * x + ch] = lpto_pow(mapped, invGamma);
      }
      dst[4 * x + 3] = 1.0f;
    }
  }
}
