/*
 * oracle_math.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).  Never linked into,
 * imported by or executed from the product path; only tests/, smoke() and
 * bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * Scalar fp32 restatement of the arithmetic the reference's HLSL performs:
 *   RNG            assets/shaders/RaytracingUtils.hlsli:26-45
 *   samplers       assets/shaders/RaytracingUtils.hlsli:49-123
 *   Fresnel        assets/shaders/RaytracingUtils.hlsli:126-130
 *
 * Every expression is written with explicit temporaries, left to right, and
 * the file is compiled with -ffp-contract=off so no FMA is ever formed.  The
 * transcendental functions (sin/cos/exp/log/pow) are NOT libm calls: HLSL
 * leaves their precision to the driver, so this engine defines them as fixed
 * polynomial kernels (published Cephes single-precision algorithms by
 * S. Moshier, restated here) made only of + - * and integer bit operations;
 * the HIP kernels restate the same definitions independently and the tests
 * require bit-for-bit agreement.  sqrt and divide are IEEE correctly rounded
 * on both sides.
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

namespace orc {

struct V3 { float x, y, z; };

static inline V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
static inline V3 vadd(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 vsub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 vmul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 vdiv(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline V3 vscale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 vdivs(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline V3 vneg(V3 a) { return v3(-a.x, -a.y, -a.z); }

static inline float dot3(V3 a, V3 b)
{
    float s = a.x * b.x;
    s = s + a.y * b.y;
    s = s + a.z * b.z;
    return s;
}

static inline V3 cross3(V3 a, V3 b)
{
    return v3(a.y * b.z - a.z * b.y,
              a.z * b.x - a.x * b.z,
              a.x * b.y - a.y * b.x);
}

/* HLSL normalize(v) = v * rsqrt(dot(v,v)); rsqrt defined as 1/sqrt, both IEEE. */
static inline V3 normalize3(V3 v)
{
    float r = 1.0f / sqrtf(dot3(v, v));
    return vscale(v, r);
}
static inline float length3(V3 v) { return sqrtf(dot3(v, v)); }

/* HLSL min/max return the non-NaN operand. */
/* Written out (IEEE-754-2008 minNum/maxNum with -0 < +0, the gfx950
 * v_min_f32 / v_max_f32 behaviour) so the result never depends on libm. */
static inline float fmin_(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return signbit(a) ? a : b;
    return a < b ? a : b;
}
static inline float fmax_(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return signbit(a) ? b : a;
    return a > b ? a : b;
}
static inline float saturate(float x) { return fmin_(fmax_(x, 0.0f), 1.0f); }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* IEEE binary32 -> binary16 -> binary32: what a store to / load from an RGBA16F texture does to a value (the reference's output texture,
 * src/DXRExperimentsApp.cpp:28).  nearest = true: round to nearest even; false: toward zero (the D3D11 functional spec's rule for float ->
 * lower-precision float).  Subnormal halves are produced (no flush), overflow goes to infinity (nearest) or 65504 (toward zero), NaN stays NaN. */
static inline float round_to_half(float x, bool nearest)
{
    const uint32_t u = f2u(x), sign = u & 0x80000000u, a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return x;                                  /* inf, NaN */
    uint32_t h;                                                      /* the half's 15 magnitude bits */
    if (a >= 0x38800000u) {                                          /* normal half range (>= 2^-14) or overflow */
        const uint32_t m = a - 0x38000000u;                          /* rebias 127 -> 15; 13 fraction bits to drop */
        h = m >> 13;
        const uint32_t rest = m & 0x1FFFu;
        if (nearest && (rest > 0x1000u || (rest == 0x1000u && (h & 1u)))) h++;
        if (h >= 0x7C00u) h = nearest ? 0x7C00u : 0x7BFFu;
    } else if (a < 0x33000000u) {                                    /* below half the smallest subnormal half (2^-25) */
        h = 0;
        if (nearest && a > 0x33000000u) h = 1;                       /* (never: kept for symmetry) */
    } else {                                                         /* subnormal half: value = mant * 2^-24 */
        const int e = (int)(a >> 23);                                /* 102 (2^-25) .. 112 (2^-15) */
        const uint32_t mant = (a & 0x7FFFFFu) | 0x800000u;           /* 24-bit significand */
        const int shift = 126 - e;                                   /* bits to drop: 14 .. 24 */
        h = mant >> shift;
        const uint32_t rest = mant & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (nearest && (rest > halfway || (rest == halfway && (h & 1u)))) h++;
    }
    /* back to binary32 (exact) */
    uint32_t out;
    const uint32_t he = h >> 10, hm = h & 0x3FFu;
    if (h == 0) out = 0;
    else if (he == 0x1Fu) out = 0x7F800000u | (hm << 13);
    else if (he != 0) out = ((he + 112u) << 23) | (hm << 13);
    else {                                                           /* subnormal half -> normal float */
        int k = 0;
        uint32_t mm = hm;
        while (!(mm & 0x400u)) { mm <<= 1; k++; }
        out = ((uint32_t)(113 - k) << 23) | ((mm & 0x3FFu) << 13);
    }
    return u2f(sign | out);
}

/* ---- deterministic transcendental kernels ------------------------------ */

/* sin and cos of x (|x| < 2^16 * pi/2).  Quadrant reduction with a 3-term
 * Cody-Waite split of pi/2 (products with the first two terms are exact for
 * the range used), then degree-7 / degree-8 minimax polynomials. */
static inline void sincos_(float x, float *s, float *c)
{
    float kf = rintf(x * 0.63661977236758134f);
    int   q  = (int)kf;
    float r  = x - kf * 1.5703125f;
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.54978995489188216e-8f;
    float z  = r * r;

    float sp = -1.9515295891e-4f * z;
    sp = sp + 8.3321608736e-3f;
    sp = sp * z;
    sp = sp - 1.6666654611e-1f;
    sp = sp * z;
    sp = sp * r;
    sp = sp + r;

    float cp = 2.443315711809948e-5f * z;
    cp = cp - 1.388731625493765e-3f;
    cp = cp * z;
    cp = cp + 4.166664568298827e-2f;
    cp = cp * z;
    cp = cp * z;
    cp = cp - 0.5f * z;
    cp = cp + 1.0f;

    switch (q & 3) {
    case 0:  *s = sp;  *c = cp;  break;
    case 1:  *s = cp;  *c = -sp; break;
    case 2:  *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp;  break;
    }
}

/* 2^n for -126 <= n <= 127 */
static inline float pow2i(int n) { return u2f((uint32_t)(n + 127) << 23); }

/* e^x.  x < -86 returns 0 (no denormal results by definition), x > 88.5 +inf. */
static inline float exp_(float x)
{
    if (x != x) return x;
    if (x > 88.5f) return u2f(0x7f800000u);
    if (x < -86.0f) return 0.0f;
    float zf = floorf(1.44269504088896341f * x + 0.5f);
    int   n  = (int)zf;
    x = x - zf * 0.693359375f;
    x = x - zf * -2.12194440e-4f;
    float zz = x * x;
    float p = 1.9875691500e-4f * x;
    p = p + 1.3981999507e-3f;
    p = p * x;
    p = p + 8.3334519073e-3f;
    p = p * x;
    p = p + 4.1665795894e-2f;
    p = p * x;
    p = p + 1.6666665459e-1f;
    p = p * x;
    p = p + 5.0000001201e-1f;
    p = p * zz;
    p = p + x;
    p = p + 1.0f;
    /* scale by 2^n in two exact steps (n in [-125,128]) */
    int n1 = n / 2;
    int n2 = n - n1;
    p = p * pow2i(n1);
    p = p * pow2i(n2);
    return p;
}

/* natural log of a positive normal float */
static inline float log_(float x)
{
    uint32_t b = f2u(x);
    int   e = (int)((b >> 23) & 0xffu) - 126;
    float m = u2f((b & 0x007fffffu) | 0x3f000000u);   /* [0.5,1) */
    if (m < 0.707106781186547524f) {
        e = e - 1;
        m = m + m;
        m = m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = 7.0376836292e-2f * m;
    y = y - 1.1514610310e-1f;
    y = y * m;
    y = y + 1.1676998740e-1f;
    y = y * m;
    y = y - 1.2420140846e-1f;
    y = y * m;
    y = y + 1.4249322787e-1f;
    y = y * m;
    y = y - 1.6668057665e-1f;
    y = y * m;
    y = y + 2.0000714765e-1f;
    y = y * m;
    y = y - 2.4999993993e-1f;
    y = y * m;
    y = y + 3.3333331174e-1f;
    y = y * m;
    y = y * z;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = m + y;
    r = r + 0.693359375f * fe;
    return r;
}

/* HLSL pow(x,y) = exp2(y*log2 x): defined here as exp_(y*log_(x)); pow(0,y>0)=0,
 * negative base -> NaN, like the D3D lowering. */
static inline float pow_(float x, float y)
{
    if (x == 0.0f) return 0.0f;
    if (!(x > 0.0f)) return u2f(0x7fc00000u);
    float l = log_(x);
    return exp_(y * l);
}

/* ---- RNG: RaytracingUtils.hlsli:26-45 ---------------------------------- */

static inline uint32_t initRand(uint32_t val0, uint32_t val1)
{
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (uint32_t n = 0; n < 16; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

static inline float nextRand(uint32_t *s)
{
    *s = 1664525u * (*s) + 1013904223u;
    return (float)(*s & 0x00FFFFFFu) / (float)0x01000000;
}

/* ---- samplers: RaytracingUtils.hlsli:49-123 ----------------------------- */

static inline V3 getPerpendicularVector(V3 u)
{
    V3 a = v3(fabsf(u.x), fabsf(u.y), fabsf(u.z));
    uint32_t xm = ((a.x - a.y) < 0 && (a.x - a.z) < 0) ? 1u : 0u;
    uint32_t ym = (a.y - a.z) < 0 ? (1u ^ xm) : 0u;
    uint32_t zm = 1u ^ (xm | ym);
    return cross3(u, v3((float)xm, (float)ym, (float)zm));
}

static inline V3 combine3(float x, V3 t, float y, V3 n, float z, V3 b)
{
    /* x * tangent + y * normal + z * bitangent, evaluated left to right */
    V3 r = vscale(t, x);
    r = vadd(r, vscale(n, y));
    r = vadd(r, vscale(b, z));
    return r;
}

static inline V3 getCosHemisphereSample(uint32_t *seed, V3 n)
{
    float r0 = nextRand(seed);
    float r1 = nextRand(seed);
    V3 bitangent = getPerpendicularVector(n);
    V3 tangent = cross3(bitangent, n);
    float r = sqrtf(r0);
    float phi = 2.0f * 3.14159265f * r1;
    float s, c;
    sincos_(phi, &s, &c);
    float x = r * c;
    float z = r * s;
    float y = sqrtf(1.0f - r0);
    return combine3(x, tangent, y, n, z, bitangent);
}

static inline V3 getUniformHemisphereSample(uint32_t *seed, V3 n)
{
    float r0 = nextRand(seed);
    float r1 = nextRand(seed);
    V3 bitangent = getPerpendicularVector(n);
    V3 tangent = cross3(bitangent, n);
    float cosTheta = r0;
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    float phi = 2.0f * 3.14159265f * r1;
    float s, c;
    sincos_(phi, &s, &c);
    float x = sinTheta * c;
    float z = sinTheta * s;
    float y = cosTheta;
    return combine3(x, tangent, y, n, z, bitangent);
}

static inline V3 samplePhongLobe(uint32_t *seed, V3 mirrorDir, float exponent, float *pdf, float *brdf)
{
    const float pi = 3.14159265f;
    float r0 = nextRand(seed);
    float r1 = nextRand(seed);
    V3 bitangent = getPerpendicularVector(mirrorDir);
    V3 tangent = cross3(bitangent, mirrorDir);
    float cosTheta = pow_(r0, 1.0f / (exponent + 1.0f));
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    float phi = 2.0f * pi * r1;
    float poweredCos = pow_(cosTheta, exponent);
    *pdf  = (exponent + 1.0f) / (2.0f * pi) * poweredCos;
    *brdf = (exponent + 2.0f) / (2.0f * pi) * poweredCos;
    float s, c;
    sincos_(phi, &s, &c);
    float x = sinTheta * c;
    float z = sinTheta * s;
    float y = cosTheta;
    return combine3(x, tangent, y, mirrorDir, z, bitangent);
}

/* RaytracingUtils.hlsli:126-130 */
static inline V3 FresnelReflectanceSchlick(V3 I, V3 N, V3 f0)
{
    float cosi = saturate(dot3(vneg(I), N));
    float p = pow_(1.0f - cosi, 5.0f);
    return v3(f0.x + (1.0f - f0.x) * p,
              f0.y + (1.0f - f0.y) * p,
              f0.z + (1.0f - f0.z) * p);
}

/* HLSL reflect(i,n) = i - 2*dot(i,n)*n */
static inline V3 reflect3(V3 i, V3 n)
{
    float k = 2.0f * dot3(i, n);
    return vsub(i, vscale(n, k));
}

}  // namespace orc

#endif
