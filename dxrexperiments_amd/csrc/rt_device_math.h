// rt_device_math.h -- gfx950 device-side arithmetic of the progressive path.
//
// Everything the shading kernels compute with is defined here so that results
// are reproducible to the bit: the library is compiled with -ffp-contract=off
// (no FMA is ever formed), divide and sqrt are the correctly rounded forms, and
// sin/cos/exp/log/pow are fixed polynomial kernels built from + - * and integer
// operations only (HLSL leaves their precision to the driver, so the engine
// pins its own: Cephes-style single-precision kernels, see DESIGN.md "Numerics").
//
// Restates: assets/shaders/RaytracingUtils.hlsli:22-130 (RNG, samplers,
// Fresnel) and the HLSL intrinsics they use (normalize, reflect, saturate, pow).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __device__ __forceinline__

namespace rtd {

struct f3 { float x, y, z; };

RT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }

// sums are always left to right: (x + y) + z
RT_DEV float dot(f3 a, f3 b)
{
    float s = a.x * b.x;
    s += a.y * b.y;
    s += a.z * b.z;
    return s;
}
RT_DEV f3 cross(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_DEV float rsqrt_ieee(float x) { return 1.0f / __builtin_sqrtf(x); }
RT_DEV f3 normalize(f3 v) { return v * rsqrt_ieee(dot(v, v)); }
RT_DEV float length(f3 v) { return __builtin_sqrtf(dot(v, v)); }
RT_DEV float fmin2(float a, float b) { return __builtin_fminf(a, b); }   // v_min_f32: NaN-ignoring
RT_DEV float fmax2(float a, float b) { return __builtin_fmaxf(a, b); }
RT_DEV float saturate(float x) { return fmin2(fmax2(x, 0.0f), 1.0f); }
RT_DEV f3 reflect(f3 i, f3 n) { return i - n * (2.0f * dot(i, n)); }

// ---- transcendental kernels -------------------------------------------------

RT_DEV void sincos_det(float x, float &sn, float &cs)
{
    const float k = __builtin_rintf(x * 0.63661977236758134f);
    const int q = (int)k;
    float r = x - k * 1.5703125f;
    r -= k * 4.837512969970703125e-4f;
    r -= k * 7.54978995489188216e-8f;
    const float z = r * r;
    float s = -1.9515295891e-4f * z + 8.3321608736e-3f;
    s = s * z - 1.6666654611e-1f;
    s = s * z * r + r;
    float c = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    c = c * z + 4.166664568298827e-2f;
    c = c * z * z - 0.5f * z + 1.0f;
    const bool swap = (q & 1) != 0;
    const float a = swap ? c : s;
    const float b = swap ? s : c;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}

RT_DEV float exp2i(int n) { return __uint_as_float((uint32_t)(n + 127) << 23); }

RT_DEV float exp_det(float x)
{
    if (x != x) return x;
    if (x > 88.5f) return __uint_as_float(0x7f800000u);
    if (x < -86.0f) return 0.0f;
    const float zf = __builtin_floorf(1.44269504088896341f * x + 0.5f);
    const int n = (int)zf;
    x -= zf * 0.693359375f;
    x -= zf * -2.12194440e-4f;
    const float xx = x * x;
    float p = 1.9875691500e-4f * x + 1.3981999507e-3f;
    p = p * x + 8.3334519073e-3f;
    p = p * x + 4.1665795894e-2f;
    p = p * x + 1.6666665459e-1f;
    p = p * x + 5.0000001201e-1f;
    p = p * xx + x + 1.0f;
    const int h = n / 2;
    p *= exp2i(h);
    p *= exp2i(n - h);
    return p;
}

RT_DEV float log_det(float x)
{
    const uint32_t b = __float_as_uint(x);
    int e = (int)((b >> 23) & 0xffu) - 126;
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
    else m = m - 1.0f;
    const float z = m * m;
    float y = 7.0376836292e-2f * m - 1.1514610310e-1f;
    y = y * m + 1.1676998740e-1f;
    y = y * m - 1.2420140846e-1f;
    y = y * m + 1.4249322787e-1f;
    y = y * m - 1.6668057665e-1f;
    y = y * m + 2.0000714765e-1f;
    y = y * m - 2.4999993993e-1f;
    y = y * m + 3.3333331174e-1f;
    y = y * m * z;
    const float fe = (float)e;
    y += -2.12194440e-4f * fe;
    y += -0.5f * z;
    float r = m + y;
    r += 0.693359375f * fe;
    return r;
}

RT_DEV float pow_det(float x, float y)
{
    if (x == 0.0f) return 0.0f;
    if (!(x > 0.0f)) return __uint_as_float(0x7fc00000u);
    return exp_det(y * log_det(x));
}

// ---- RNG (RaytracingUtils.hlsli:26-45) ---------------------------------------

RT_DEV uint32_t init_rand(uint32_t val0, uint32_t val1)
{
    uint32_t v0 = val0, v1 = val1, s0 = 0;
#pragma unroll
    for (int n = 0; n < 16; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

RT_DEV float next_rand(uint32_t &s)
{
    s = 1664525u * s + 1013904223u;
    return (float)(s & 0x00FFFFFFu) / 16777216.0f;
}

// ---- samplers (RaytracingUtils.hlsli:49-123) ---------------------------------

RT_DEV f3 perpendicular(f3 u)
{
    const float ax = __builtin_fabsf(u.x), ay = __builtin_fabsf(u.y), az = __builtin_fabsf(u.z);
    const uint32_t xm = ((ax - ay) < 0.0f && (ax - az) < 0.0f) ? 1u : 0u;
    const uint32_t ym = ((ay - az) < 0.0f) ? (1u ^ xm) : 0u;
    const uint32_t zm = 1u ^ (xm | ym);
    return cross(u, mk3((float)xm, (float)ym, (float)zm));
}

// x*tangent + y*axis + z*bitangent, left to right
RT_DEV f3 frame_combine(float x, f3 t, float y, f3 n, float z, f3 b)
{
    f3 r = t * x;
    r = r + n * y;
    r = r + b * z;
    return r;
}

RT_DEV f3 cos_hemisphere(uint32_t &seed, f3 n)
{
    const float r0 = next_rand(seed);
    const float r1 = next_rand(seed);
    const f3 bt = perpendicular(n);
    const f3 tg = cross(bt, n);
    const float r = __builtin_sqrtf(r0);
    const float phi = 2.0f * 3.14159265f * r1;
    float s, c;
    sincos_det(phi, s, c);
    return frame_combine(r * c, tg, __builtin_sqrtf(1.0f - r0), n, r * s, bt);
}

RT_DEV f3 uniform_hemisphere(uint32_t &seed, f3 n)
{
    const float r0 = next_rand(seed);
    const float r1 = next_rand(seed);
    const f3 bt = perpendicular(n);
    const f3 tg = cross(bt, n);
    const float ct = r0;
    const float st = __builtin_sqrtf(1.0f - ct * ct);
    const float phi = 2.0f * 3.14159265f * r1;
    float s, c;
    sincos_det(phi, s, c);
    return frame_combine(st * c, tg, ct, n, st * s, bt);
}

RT_DEV f3 phong_lobe(uint32_t &seed, f3 mirror, float exponent, float &pdf, float &brdf)
{
    const float pi = 3.14159265f;
    const float r0 = next_rand(seed);
    const float r1 = next_rand(seed);
    const f3 bt = perpendicular(mirror);
    const f3 tg = cross(bt, mirror);
    const float ct = pow_det(r0, 1.0f / (exponent + 1.0f));
    const float st = __builtin_sqrtf(1.0f - ct * ct);
    const float phi = 2.0f * pi * r1;
    const float pc = pow_det(ct, exponent);
    pdf = (exponent + 1.0f) / (2.0f * pi) * pc;
    brdf = (exponent + 2.0f) / (2.0f * pi) * pc;
    float s, c;
    sincos_det(phi, s, c);
    return frame_combine(st * c, tg, ct, mirror, st * s, bt);
}

RT_DEV f3 fresnel_schlick(f3 I, f3 N, f3 f0)
{
    const float cosi = saturate(dot(-I, N));
    const float p = pow_det(1.0f - cosi, 5.0f);
    return mk3(f0.x + (1.0f - f0.x) * p, f0.y + (1.0f - f0.y) * p, f0.z + (1.0f - f0.z) * p);
}

}  // namespace rtd
