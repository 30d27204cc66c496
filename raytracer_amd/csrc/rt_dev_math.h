// rt_dev_math.h - device-side float32 arithmetic of the rendering path (gfx950).
//
// Implements the "arithmetic contract" of DESIGN.md: IEEE binary32, round-to-nearest-even,
// subnormals kept, no implicit contraction (the TU is built with -ffp-contract=off), fused
// multiply-adds only where __builtin_fmaf is written.  The CPU oracle implements the same
// contract independently (oracle/rt_oracle.c); the two share no code, and the GPU tests
// check that they agree bit-for-bit.
//
// Reference semantics restated here (shaders/raytracer.fs.glsl): dot/cross/normalize/
// reflect/mix built-ins, randomValue :457-461, randomDirection :478-492.  GLSL leaves the
// precision of sin/cos/acos/exp/log to the driver; this build fixes them as the polynomial
// routines below (Cephes single-precision coefficient sets, fma Horner form).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float dot3(v3 a, v3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ v3 cross3(v3 a, v3 b) {
    return V3(fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)));
}
// correctly rounded sqrt and division (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt)
__device__ __forceinline__ float length3(v3 a) { return __builtin_sqrtf(dot3(a, a)); }
__device__ __forceinline__ v3 normalize3(v3 a) { return a * (1.0f / length3(a)); }
__device__ __forceinline__ v3 madd3(v3 o, float t, v3 d) { return V3(fma_(t, d.x, o.x), fma_(t, d.y, o.y), fma_(t, d.z, o.z)); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ v3 reflect3(v3 I, v3 N) {
    float s = 2.0f * dot3(N, I);
    return V3(fma_(-s, N.x, I.x), fma_(-s, N.y, I.y), fma_(-s, N.z, I.z));
}

// ---- fixed elementary functions -------------------------------------------------------
__device__ __forceinline__ void det_sincos(float x, float& s_out, float& c_out) {
    float q = __builtin_rintf(x * 0x1.45f306p-1f);
    float r = fma_(q, -0x1.921fb6p+0f, x);
    r = fma_(q, 0x1.777a5cp-25f, r);
    r = fma_(q, 0x1.ee59dap-50f, r);
    float z = r * r;
    float ps = fma_(fma_(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float s = fma_(r * z, ps, r);
    float pc = fma_(fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float c = fma_(z * z, pc, fma_(-0.5f, z, 1.0f));
    int n = (int)q & 3;
    float ss = (n & 1) ? c : s;
    float cc = (n & 1) ? s : c;
    if (n == 2 || n == 3) ss = -ss;
    if (n == 1 || n == 2) cc = -cc;
    s_out = ss;
    c_out = cc;
}

__device__ __forceinline__ float det_asin_poly(float a, float z) {
    float p = fma_(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = fma_(p, z, 4.5470025998e-2f);
    p = fma_(p, z, 7.4953002686e-2f);
    p = fma_(p, z, 1.6666752422e-1f);
    return fma_(a * z, p, a);
}

__device__ __forceinline__ float det_acos(float x) {
    float ax = __builtin_fabsf(x);
    if (ax > 0.5f) {
        float z = 0.5f * (1.0f - ax);
        float w = __builtin_sqrtf(z);
        float r = 2.0f * det_asin_poly(w, z);
        return x > 0.0f ? r : 0x1.921fb6p+1f - r;
    }
    return 0x1.921fb6p+0f - det_asin_poly(x, x * x);
}

__device__ __noinline__ float det_exp(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return __builtin_inff();
    if (x < -103.97f) return 0.0f;
    float k = __builtin_rintf(x * 0x1.715476p+0f);
    float r = fma_(k, -0x1.62e43p-1f, x);
    r = fma_(k, 0x1.05c61p-29f, r);
    float p = fma_(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    float y = fma_(p, r * r, r) + 1.0f;
    int ki = (int)k;
    int k1 = ki / 2, k2 = ki - k1;
    return (y * __uint_as_float((uint32_t)(k1 + 127) << 23)) * __uint_as_float((uint32_t)(k2 + 127) << 23);
}

__device__ __noinline__ float det_log(float x) {
    if (x != x || x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    int e = 0;
    if (x < 0x1p-126f) { x *= 0x1p23f; e = -23; }
    uint32_t u = __float_as_uint(x);
    e += (int)(u >> 23) - 126;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float p = fma_(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = fma_(p, m, 1.1676998740e-1f);
    p = fma_(p, m, -1.2420140846e-1f);
    p = fma_(p, m, 1.4249322787e-1f);
    p = fma_(p, m, -1.6668057665e-1f);
    p = fma_(p, m, 2.0000714765e-1f);
    p = fma_(p, m, -2.4999993993e-1f);
    p = fma_(p, m, 3.3333331174e-1f);
    float fe = (float)e;
    float y = (m * z) * p;
    y = fma_(-2.12194440e-4f, fe, y);
    y = fma_(-0.5f, z, y);
    return fma_(0.693359375f, fe, m + y);
}

// ---- RNG (raytracer.fs.glsl:457-492) ---------------------------------------------------
__device__ __forceinline__ float random_value(uint32_t& s) {
    uint32_t x = s;
    x = x * ((x + 195439u) * (x + 124395u) * (x + 845921u));
    s = x;
    return (float)x * 0x1p-32f;   // float(state) / 4294967295.0 ; the literal is 2^32 in binary32
}

__device__ __forceinline__ v3 random_direction(uint32_t& s) {
    float u = random_value(s);
    float v = random_value(s);
    float theta = 6.2831855f * u;
    float phi = det_acos(2.0f * v - 1.0f);
    float sp, cp, st, ct;
    det_sincos(phi, sp, cp);
    det_sincos(theta, st, ct);
    return V3(sp * ct, sp * st, cp);
}
