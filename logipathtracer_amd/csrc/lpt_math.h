// Device-side arithmetic for the path-tracing kernels (gfx950).
//
// Bit-level contract: IEEE-754 binary32, round-to-nearest-even, NO fused
// multiply-add, denormals kept, correctly rounded divide and square root.
// The translation unit must be compiled with
//   -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
// and without any fast-math option.  Expression order below is significant:
// pixel values are compared bit for bit against an independent CPU
// implementation of the same arithmetic.
//
// GLSL leaves the precision of sin/cos/tan/log/exp/pow to the implementation
// (shaders/common/util.glsl:7, shaders/heitz/BSDF.glsl:55-56,81); here they
// are pinned to fixed single-precision minimax polynomials.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace lpt {

// shaders/common/constants.glsl:4-9
constexpr float kEps = 0.0001f;
constexpr float kPi = 3.141592653589f;
constexpr float kInfinity = 3.4E+38f;

struct f3 {
  float x, y, z;
};

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }

__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
  return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float length3(f3 a) { return __builtin_sqrtf(dot3(a, a)); }
__device__ __forceinline__ f3 normalize3(f3 a) {
  float s = 1.0f / __builtin_sqrtf(dot3(a, a));
  return a * s;
}
// IEEE minNum / maxNum (v_min_f32 / v_max_f32 in IEEE mode)
__device__ __forceinline__ float min_num(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float max_num(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float clamp_num(float x, float lo, float hi) { return min_num(max_num(x, lo), hi); }

__device__ __forceinline__ float as_float(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t as_uint(float f) { return __float_as_uint(f); }

// ---- fixed transcendental algorithms --------------------------------------

__device__ __forceinline__ void sincos_fixed(float x, float& s, float& c) {
  if (!(__builtin_fabsf(x) <= 8192.0f)) {
    s = c = as_float(0x7fc00000u);
    return;
  }
  float t = x * 0.636619772f;  // 2/pi
  int qi = (int)(t + (t < 0.0f ? -0.5f : 0.5f));
  float q = (float)qi;
  float r = ((x - q * 1.5703125f) - q * 4.837512969970703125e-4f) - q * 7.54978995489188216e-8f;
  float z = r * r;
  float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
  cp = cp - 0.5f * z;
  cp = cp + 1.0f;
  int quad = qi & 3;
  float ss = (quad & 1) ? cp : sp;
  float cc = (quad & 1) ? sp : cp;
  s = (quad & 2) ? -ss : ss;
  c = ((quad == 1) || (quad == 2)) ? -cc : cc;
}

__device__ __forceinline__ float tan_fixed(float x) {
  float s, c;
  sincos_fixed(x, s, c);
  return s / c;
}

__device__ __forceinline__ float log_fixed(float x) {
  if (x != x) return x;
  if (x < 0.0f) return as_float(0x7fc00000u);
  if (x == 0.0f) return as_float(0xff800000u);
  uint32_t b = as_uint(x);
  if (b == 0x7f800000u) return x;
  int e = 0;
  if (b < 0x00800000u) {
    x = x * 16777216.0f;
    b = as_uint(x);
    e = -24;
  }
  e += (int)(b >> 23) - 126;
  float m = as_float((b & 0x007fffffu) | 0x3f000000u);
  if (m < 0.707106781186547524f) {
    e -= 1;
    m = (m + m) - 1.0f;
  } else {
    m = m - 1.0f;
  }
  float z = m * m;
  float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m +
                 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m +
             3.3333331174e-1f) * m * z;
  float fe = (float)e;
  y = y + -2.12194440e-4f * fe;
  y = y + -0.5f * z;
  z = m + y;
  z = z + 0.693359375f * fe;
  return z;
}

__device__ __forceinline__ float exp_fixed(float x) {
  if (x != x) return x;
  if (x > 88.72283905206835f) return as_float(0x7f800000u);
  if (x < -103.278929903431851103f) return 0.0f;
  float z = __builtin_floorf(1.44269504088896341f * x + 0.5f);
  x = x - z * 0.693359375f;
  x = x - z * -2.12194440e-4f;
  int n = (int)z;
  z = x * x;
  z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x +
        1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
  int n1 = n / 2;
  int n2 = n - n1;
  float p1 = as_float((uint32_t)(n1 + 127) << 23);
  float p2 = as_float((uint32_t)(n2 + 127) << 23);
  return (z * p1) * p2;
}

__device__ __forceinline__ float pow_fixed(float x, float y) {
  if (x != x || y != y) return as_float(0x7fc00000u);
  if (x < 0.0f) return as_float(0x7fc00000u);
  if (x == 0.0f) {
    if (y > 0.0f) return 0.0f;
    if (y == 0.0f) return 1.0f;
    return as_float(0x7f800000u);
  }
  return exp_fixed(y * log_fixed(x));
}

// shaders/common/util.glsl:4-8
__device__ __forceinline__ float srgb_to_linear(float srgb) {
  if (srgb <= 0.04045f) return srgb / 12.92f;
  return pow_fixed((srgb + 0.055f) / 1.055f, 2.4f);
}

// ---- RNG: shaders/common/random.glsl:6-15 ------------------------------------

struct Rng {
  uint32_t x, y;  // "uvec2 seed"
};

__device__ __forceinline__ float rand01(Rng& s) {
  s.x += 1u;
  s.y += 1u;
  uint32_t qx = 1103515245u * ((s.x >> 1) ^ s.y);
  uint32_t qy = 1103515245u * ((s.y >> 1) ^ s.x);
  uint32_t n = 1103515245u * (qx ^ (qy >> 3));
  // 1.0 / float(0xffffffffU): float(0xffffffff) rounds to 2^32, so the factor is 2^-32
  return (float)n * (1.0f / 4294967296.0f);
}

}  // namespace lpt
