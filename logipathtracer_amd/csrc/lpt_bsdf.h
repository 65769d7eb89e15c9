// BSDF sampling on the device: the multiple-scattering microfacet family
// (shaders/heitz/BSDF.glsl) and the basic family (shaders/basic/BSDF.glsl),
// plus lobe selection (shaders/heitz/interaction_type.glsl).
// All directions are in the local shading frame (z = front-facing normal).
#pragma once
#include "lpt_math.h"

#pragma clang fp contract(off)

namespace lpt {

// shaders/heitz/BSDF.glsl:8
constexpr int kHeitzMaxOrder = 16;
// shaders/heitz/interaction_type.glsl:6-8
constexpr uint32_t kDiff = 1u, kMetallic = 2u, kTrans = 4u;

// heitz/BSDF.glsl:10-24 — exact unpolarised dielectric Fresnel, 1 on total internal reflection
__device__ __forceinline__ float fresnel_dielectric(float vdoth, float eta) {
  float ct2 = 1.0f - (1.0f - vdoth * vdoth) / (eta * eta);
  if (ct2 <= 0.0f) return 1.0f;
  float ct = __builtin_sqrtf(ct2);
  float Rs = (vdoth - eta * ct) / (vdoth + eta * ct);
  float Rp = (eta * vdoth - ct) / (eta * vdoth + ct);
  return 0.5f * (Rs * Rs + Rp * Rp);
}

// heitz/BSDF.glsl:26-32
__device__ __forceinline__ f3 refract_eta(f3 wi, f3 wm, float eta) {
  float ci = dot3(wi, wm);
  float ct2 = 1.0f - (1.0f - ci * ci) / (eta * eta);
  float ct = -__builtin_sqrtf(max_num(0.0f, ct2));
  float k = dot3(wi, wm) / eta + ct;
  return wm * k - f3{wi.x / eta, wi.y / eta, wi.z / eta};
}

// heitz/BSDF.glsl:41-67 — GGX visible-normal sampling, two random numbers
__device__ __forceinline__ f3 sample_ggx_vndf(Rng& rng, f3 Ve, float alpha) {
  float R1 = rand01(rng);
  float R2 = rand01(rng);
  f3 Vh = normalize3(f3{alpha * Ve.x, alpha * Ve.y, Ve.z});
  f3 T1 = (Vh.z < 1.0f) ? normalize3(cross3(f3{0.0f, 0.0f, 1.0f}, Vh)) : f3{1.0f, 0.0f, 0.0f};
  f3 T2 = cross3(Vh, T1);
  float r = __builtin_sqrtf(R1);
  float phi = (2.0f * kPi) * R2;
  float sn, cs;
  sincos_fixed(phi, sn, cs);
  float t1 = r * cs;
  float t2 = r * sn;
  float s = 0.5f * (1.0f + Vh.z);
  t2 = (1.0f - s) * __builtin_sqrtf(1.0f - t1 * t1) + s * t2;
  float nz = __builtin_sqrtf(max_num(0.0f, (1.0f - t1 * t1) - t2 * t2));
  f3 Nh = (T1 * t1 + T2 * t2) + Vh * nz;
  return normalize3(f3{alpha * Nh.x, alpha * Nh.y, max_num(0.0f, Nh.z)});
}

// heitz/BSDF.glsl:72-84 — next height along dir on the Smith microsurface, one random number
__device__ __forceinline__ float sample_ggx_height(Rng& rng, f3 dir, float height, float alpha) {
  float len = length3(dir * f3{alpha, alpha, 1.0f});
  float projectedArea = max_num(0.5f * (len - dir.z), 1e-7f);
  float R = rand01(rng);
  float delta = -log_fixed(1.0f - R) * dir.z / projectedArea;
  return height + delta;
}

// heitz/BSDF.glsl:88-104 — mirror about a visible micro-normal (weight F0, Schlick disabled at :101)
__device__ __forceinline__ f3 conductor_phase(Rng& rng, f3 viewDir, float alpha) {
  f3 m = sample_ggx_vndf(rng, viewDir, alpha);
  float vdoth = clamp_num(dot3(viewDir, m), 0.0f, 1.0f);
  return (m * 2.0f) * vdoth - viewDir;
}

// heitz/BSDF.glsl:108-138
__device__ __forceinline__ f3 conductor_brdf(Rng& rng, f3 F0, f3 viewDir, float roughness, f3& lightDir) {
  float alpha = roughness * roughness;
  f3 energy{1.0f, 1.0f, 1.0f};
  lightDir = -viewDir;
  float height = 0.0f;
  for (int order = 0; order < kHeitzMaxOrder; ++order) {
    height = sample_ggx_height(rng, lightDir, height, alpha);
    if (height > 0.0f) break;
    lightDir = conductor_phase(rng, -lightDir, alpha);
    energy = energy * F0;
  }
  return energy;
}

// heitz/BSDF.glsl:140-158
__device__ __forceinline__ f3 dielectric_phase(Rng& rng, f3 viewDir, float alpha, float eta, bool& outside) {
  f3 m = sample_ggx_vndf(rng, viewDir, alpha);
  float vdoth = dot3(viewDir, m);
  float F = fresnel_dielectric(vdoth, eta);
  if (rand01(rng) < F) {
    return (m * 2.0f) * vdoth - viewDir;
  }
  outside = !outside;
  return normalize3(refract_eta(viewDir, m, eta));
}

// heitz/BSDF.glsl:160-209 — returns F0 whatever the walk did; `transmittance` is unused there
__device__ __forceinline__ f3 dielectric_bsdf(Rng& rng, f3 F0, f3 viewDir, float roughness, float ior, f3& lightDir,
                                              bool outside) {
  float alpha = roughness * roughness;
  lightDir = -viewDir;
  float height = 0.0f;
  float iorOut = outside ? 1.0f : ior;
  float iorIn = outside ? ior : 1.0f;
  outside = true;
  for (int order = 0; order < kHeitzMaxOrder; ++order) {
    if (outside) {
      height = sample_ggx_height(rng, lightDir, height, alpha);
      if (height > 0.0f) break;
    } else {
      height = -sample_ggx_height(rng, -lightDir, -height, alpha);
      if (height < 0.0f) break;
    }
    float eta = outside ? iorIn / iorOut : iorOut / iorIn;
    lightDir = dielectric_phase(rng, -lightDir, alpha, eta, outside);
  }
  return F0;
}

// heitz/BSDF.glsl:211-239 — cosine lobe about a visible micro-normal (concentric disk map)
__device__ __forceinline__ f3 diffuse_phase(Rng& rng, f3 viewDir, float alpha) {
  f3 m = sample_ggx_vndf(rng, viewDir, alpha);
  f3 u = (m.z < 1.0f) ? normalize3(cross3(f3{0.0f, 0.0f, 1.0f}, m)) : f3{1.0f, 0.0f, 0.0f};
  f3 v = cross3(m, u);
  float R1 = 2.0f * rand01(rng) - 1.0f;
  float R2 = 2.0f * rand01(rng) - 1.0f;
  float phi, R;
  if (R1 == 0.0f && R2 == 0.0f) {
    R = 0.0f;
    phi = 0.0f;
  } else if (R1 * R1 > R2 * R2) {
    R = R1;
    phi = (kPi / 4.0f) * (R2 / R1);
  } else {
    R = R2;
    phi = (kPi / 2.0f) - (R1 / R2) * (kPi / 4.0f);
  }
  float sn, cs;
  sincos_fixed(phi, sn, cs);
  float x = R * cs;
  float y = R * sn;
  float z = __builtin_sqrtf(max_num(0.0f, (1.0f - x * x) - y * y));
  return (u * x + v * y) + m * z;
}

// heitz/BSDF.glsl:241-275 — walks that reach the order cap are killed
__device__ __forceinline__ f3 diffuse_bsdf(Rng& rng, f3 F0, f3 viewDir, float roughness, f3& lightDir) {
  float alpha = roughness * roughness;
  f3 energy{1.0f, 1.0f, 1.0f};
  lightDir = -viewDir;
  float height = 0.0f;
  int order = 0;
  for (; order < kHeitzMaxOrder; ++order) {
    height = sample_ggx_height(rng, lightDir, height, alpha);
    if (height > 0.0f) break;
    lightDir = diffuse_phase(rng, -lightDir, alpha);
    energy = energy * F0;
  }
  if (order >= kHeitzMaxOrder) {
    lightDir = f3{0.0f, 0.0f, 1.0f};
    return f3{0.0f, 0.0f, 0.0f};
  }
  return energy;
}

// The three random walks above as ONE step function, for kernels where neighbouring lanes hold
// different lobes: the costly parts of an order (height sample, visible-normal sample) are
// the same code for every lobe and run with all lanes; only the short phase functions are
// lobe-specific.  Per lane, the operations and the random numbers drawn are exactly those of
// diffuse_bsdf / conductor_brdf / dielectric_bsdf (negations are exact, so the mirrored
// height sampling of heitz/BSDF.glsl:194 folds into sign flips).
struct HeitzWalk {
  f3 lightDir;
  float height;
  f3 energy;
  int order;
  bool outside;  // side of the microsurface the walk is on (dielectric only)
};

__device__ __forceinline__ void heitz_walk_begin(HeitzWalk& w, f3 viewDir) {
  w.lightDir = -viewDir;
  w.height = 0.0f;
  w.energy = f3{1.0f, 1.0f, 1.0f};
  w.order = 0;
  w.outside = true;
}

// One scattering order — the body of the `for (order ...)` loops of heitz/BSDF.glsl:116-135, :183-206
// and :249-266 — in its two halves, so that kernels may schedule them separately:
//   heitz_walk_height: the next intersection with the microsurface; true if the walk left it (walk over);
//   heitz_walk_scatter: the scattering event there; true if the order cap is reached (walk over).
// iorOut/iorIn are the indices on the ray's side / the far side.
__device__ __forceinline__ bool heitz_walk_height(Rng& rng, HeitzWalk& w, uint32_t kind, float alpha) {
  const bool flip = (kind == kTrans) && !w.outside;
  const f3 dirH = flip ? -w.lightDir : w.lightDir;
  const float hs = sample_ggx_height(rng, dirH, flip ? -w.height : w.height, alpha);
  w.height = flip ? -hs : hs;
  return flip ? (w.height < 0.0f) : (w.height > 0.0f);
}

__device__ __forceinline__ bool heitz_walk_scatter(Rng& rng, HeitzWalk& w, uint32_t kind, f3 F0, float alpha,
                                                   float iorOut, float iorIn) {
  const f3 vd = -w.lightDir;
  const f3 m = sample_ggx_vndf(rng, vd, alpha);
  if (kind == kMetallic) {
    const float vdoth = clamp_num(dot3(vd, m), 0.0f, 1.0f);
    w.lightDir = (m * 2.0f) * vdoth - vd;
    w.energy = w.energy * F0;
  } else if (kind == kDiff) {
    f3 u = (m.z < 1.0f) ? normalize3(cross3(f3{0.0f, 0.0f, 1.0f}, m)) : f3{1.0f, 0.0f, 0.0f};
    f3 v = cross3(m, u);
    float R1 = 2.0f * rand01(rng) - 1.0f;
    float R2 = 2.0f * rand01(rng) - 1.0f;
    float phi, R;
    if (R1 == 0.0f && R2 == 0.0f) {
      R = 0.0f;
      phi = 0.0f;
    } else if (R1 * R1 > R2 * R2) {
      R = R1;
      phi = (kPi / 4.0f) * (R2 / R1);
    } else {
      R = R2;
      phi = (kPi / 2.0f) - (R1 / R2) * (kPi / 4.0f);
    }
    float sn, cs;
    sincos_fixed(phi, sn, cs);
    float x = R * cs;
    float y = R * sn;
    float z = __builtin_sqrtf(max_num(0.0f, (1.0f - x * x) - y * y));
    w.lightDir = (u * x + v * y) + m * z;
    w.energy = w.energy * F0;
  } else {
    const float vdoth = dot3(vd, m);
    const float eta = w.outside ? iorIn / iorOut : iorOut / iorIn;
    const float F = fresnel_dielectric(vdoth, eta);
    if (rand01(rng) < F) {
      w.lightDir = (m * 2.0f) * vdoth - vd;
    } else {
      w.outside = !w.outside;
      w.lightDir = normalize3(refract_eta(vd, m, eta));
    }
  }
  ++w.order;
  return w.order >= kHeitzMaxOrder;
}

// One whole order.  Returns true when the walk is over (it left the microsurface or reached the order cap).
__device__ __forceinline__ bool heitz_walk_step(Rng& rng, HeitzWalk& w, uint32_t kind, f3 F0, float alpha,
                                                float iorOut, float iorIn) {
  if (heitz_walk_height(rng, w, kind, alpha)) return true;
  return heitz_walk_scatter(rng, w, kind, F0, alpha, iorOut, iorIn);
}

// Weight and sampled direction of a finished walk.
__device__ __forceinline__ f3 heitz_walk_result(const HeitzWalk& w, uint32_t kind, f3 F0, f3& lightDir) {
  lightDir = w.lightDir;
  if (kind == kTrans) return F0;  // heitz/BSDF.glsl:208
  if (kind == kDiff && w.order >= kHeitzMaxOrder) {  // heitz/BSDF.glsl:269-272
    lightDir = f3{0.0f, 0.0f, 1.0f};
    return f3{0.0f, 0.0f, 0.0f};
  }
  return w.energy;
}

__device__ __forceinline__ f3 heitz_bsdf_unified(Rng& rng, uint32_t kind, f3 F0, f3 viewDir, float roughness,
                                                 float ior, bool outsideIn, f3& lightDir) {
  const float alpha = roughness * roughness;
  const float iorOut = outsideIn ? 1.0f : ior;
  const float iorIn = outsideIn ? ior : 1.0f;
  HeitzWalk w;
  heitz_walk_begin(w, viewDir);
  while (!heitz_walk_step(rng, w, kind, F0, alpha, iorOut, iorIn)) {
  }
  return heitz_walk_result(w, kind, F0, lightDir);
}

// ---- basic family -------------------------------------------------------------

__device__ __forceinline__ f3 glsl_reflect(f3 I, f3 N) { return I - N * (2.0f * dot3(N, I)); }
__device__ __forceinline__ f3 glsl_refract(f3 I, f3 N, float eta) {
  float ndi = dot3(N, I);
  float k = 1.0f - (eta * eta) * (1.0f - ndi * ndi);
  if (k < 0.0f) return f3{0.0f, 0.0f, 0.0f};
  return I * eta - N * (eta * ndi + __builtin_sqrtf(k));
}

// basic/BSDF.glsl:3-11
__device__ __forceinline__ f3 basic_diffuse(Rng& rng, f3 F0, f3& lightDir) {
  float r1 = (2.0f * kPi) * rand01(rng);
  float r2 = rand01(rng);
  float r2s = __builtin_sqrtf(r2);
  float sn, cs;
  sincos_fixed(r1, sn, cs);
  lightDir = f3{cs * r2s, sn * r2s, __builtin_sqrtf(1.0f - r2)};
  return F0 * dot3(lightDir, f3{0.0f, 0.0f, 1.0f});
}

// basic/BSDF.glsl:13-17
__device__ __forceinline__ f3 basic_specular(f3 F0, f3 viewDir, f3& lightDir) {
  lightDir = glsl_reflect(-viewDir, f3{0.0f, 0.0f, 1.0f});
  return F0;
}

// basic/BSDF.glsl:19-50
__device__ __forceinline__ f3 basic_transmittance(Rng& rng, f3 F0, f3 viewDir, float transmittance, float ior,
                                                  bool outside, f3& lightDir) {
  const float nc = 1.0f, nt = ior;
  f3 n = outside ? f3{0.0f, 0.0f, 1.0f} : f3{0.0f, 0.0f, -1.0f};
  f3 nl{0.0f, 0.0f, 1.0f};
  f3 rd = -viewDir;
  float nnt = dot3(rd, n) < 0.0f ? (nc / nt) : (nt / nc);
  f3 tdir = glsl_refract(rd, nl, nnt);
  float ci = dot3(nl, rd);
  float ctr = dot3(nl, tdir);
  float para = (nt * ci - nc * ctr) / (nt * ci + nc * ctr);
  float perp = (nc * ci - nt * ctr) / (nc * ci + nt * ctr);
  float Re = (para * para + perp * perp) * 0.5f;
  float r = rand01(rng);
  if (r < Re) {
    lightDir = glsl_reflect(rd, n);
    return f3{1.0f, 1.0f, 1.0f};
  }
  lightDir = tdir;
  return F0 * transmittance;
}

// heitz/interaction_type.glsl:10-29 — one random number picks the lobe
__device__ __forceinline__ uint32_t pick_interaction(Rng& rng, float metallic, float transmission) {
  float wm = metallic;
  float wt = (1.0f - metallic) * transmission;
  float wd = (1.0f - transmission) * (1.0f - metallic);
  float norm = 1.0f / ((wm + wt) + wd);
  wm *= norm;
  wt *= norm;
  float r = rand01(rng);
  if (r < wm) return kMetallic;
  if (r < wm + wt) return kTrans;
  return kDiff;
}

}  // namespace lpt
