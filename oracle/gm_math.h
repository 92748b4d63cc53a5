/* gm_math.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped render path.
 *
 * Scalar FP32 definitions of the GLSL ES 3.00 built-in functions the reference's emitted fragment
 * shader calls (game-compiler/src/codegen/glsl.rs). GLSL leaves the precision of sin/cos/atan
 * open and the reference's real executor (a browser's GLSL implementation) is not available, so
 * the transcendental functions here are a fixed, documented algorithm built only from IEEE-754
 * correctly-rounded operations (+ - * / sqrt fma floor rint and integer ops). The same sequence of
 * operations evaluated on any IEEE machine gives the same bits, which is what lets a GPU
 * implementation be compared bit-for-bit. PARITY UNPINNED for these functions: nothing in the
 * reference fixes their last bits.
 *
 * Compile with -ffp-contract=off: every a*b+c below is two roundings unless written as fmaf().
 */
#ifndef GM_MATH_H
#define GM_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t gm_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float gm_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* min/max: a NaN operand yields the other operand; -0 orders below +0. */
static inline float gm_max(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (gm_bits(a) >> 31) ? b : a;
    return a > b ? a : b;
}
static inline float gm_min(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (gm_bits(a) >> 31) ? a : b;
    return a < b ? a : b;
}
static inline float gm_clamp(float x, float lo, float hi) { return gm_min(gm_max(x, lo), hi); }
static inline float gm_abs(float x) { return gm_from_bits(gm_bits(x) & 0x7fffffffu); }
static inline float gm_floor(float x) { return floorf(x); }

/* fract(x) = x - floor(x), kept strictly below 1: for tiny negative x the subtraction rounds to
 * 1.0, which GPU fract instructions (and software GLSL implementations) clamp to the largest
 * float below one. NaN and infinities give NaN. */
static inline float gm_fract(float x)
{
    float f = x - floorf(x);
    if (f != f) return f;
    return f < 1.0f ? f : 0x1.fffffep-1f;
}
/* Quotients by a stage argument or a GLSL constant: one correctly rounded reciprocal, one multiply. */
static inline float gm_rcp(float y) { return 1.0f / y; }
static inline float gm_mod(float x, float y) { return x - y * floorf(x * gm_rcp(y)); }
static inline float gm_step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
/* Contraction points: GLSL leaves a*b+c contraction to the implementation; the render path fixes
 * them, so each place below that says fmaf is one fused rounding on every machine. */
/* mix(a, b, t) = a + t (b - a): one subtraction, one fused multiply-add (GLSL words it x (1 - t) + y t and leaves the evaluation open) */
static inline float gm_mix(float a, float b, float t) { return fmaf(t, b - a, a); }
static inline float gm_smoothstep(float e0, float e1, float x)
{
    float t = gm_clamp((x - e0) * gm_rcp(e1 - e0), 0.0f, 1.0f);
    return t * t * fmaf(-2.0f, t, 3.0f);
}
static inline float gm_length2(float x, float y) { return sqrtf(fmaf(x, x, y * y)); }

/* inversesqrt. For 2^-100 <= x < inf the exact 1 / sqrt(x) rounded ONCE to the nearest float - with one stated
 * exception: for x two floats below a power of four, x = 4^k (1 - 2^-23), the result is 2^-k (the exact value
 * lies 4.5e-8 ulp above the midpoint of 2^-k and its upper neighbour; an error of 0.50000005 ulp instead of at most
 * 0.5. A Newton step from any seed equal to 2^-k lands exactly on that midpoint, which is why the definition takes the
 * even neighbour there). Elsewhere 1 / sqrt(x) as two IEEE operations (0 -> +inf, +inf -> 0, negative -> NaN).
 * The binary64 quotient 1 / sqrt(x) is within two binary64 ulps of the exact value, so the binary32 at or below it (a)
 * and the next one (b) bracket the exact value; which of the two is nearer is decided exactly: with m = (a + b) / 2
 * (25 significant bits, so m * m is exact in binary64) the sign of x * m^2 - 1, one fused operation, says on which
 * side of m the exact value lies. A tie is impossible (m has an odd 25-bit significand, so x * m^2 is never a power
 * of two). */
static inline float gm_rsqrt_cr(float x)
{
    if ((gm_bits(x) & 0x00ffffffu) == 0x007ffffeu)   /* 4^k (1 - 2^-23), biased exponent E = 2 k + 126: the result is 2^-k */
        return gm_from_bits((127u - ((gm_bits(x) >> 23) - 126u) / 2u) << 23);
    const double xd = (double)x;
    const double yd = 1.0 / sqrt(xd);
    float a = (float)yd;
    if ((double)a > yd) a = gm_from_bits(gm_bits(a) - 1u);   /* the binary32 at or below yd (positive, normal) */
    const float b = gm_from_bits(gm_bits(a) + 1u);
    const double m = 0.5 * ((double)a + (double)b);
    return fma(xd, m * m, -1.0) > 0.0 ? a : b;                /* m above the exact value: the exact value is nearer a */
}
static inline float gm_rsqrt(float x)
{
    if (x >= 0x1p-100f && x < INFINITY) return gm_rsqrt_cr(x);
    return 1.0f / sqrtf(x);
}

/* ── sin / cos ────────────────────────────────────────────────────────────────────────────
 * k = rint(x / pi); r = x - k*pi by three-constant Cody-Waite with fma for |x| <= 32768, by
 * two-constant double-precision reduction up to 2^31; beyond that (float spacing >= 256
 * radians) the argument counts as 0. sin(x) = (-1)^k S(r), cos(x) = (-1)^k C(r) with one odd and one even
 * minimax polynomial on [-pi/2, pi/2] (fitted for this file: S within 1.8 ulp, C within 7.4e-8). */
#define GM_ONE_OVER_PI 0.318309886f
#define GM_PI_A 3.140625f
#define GM_PI_B 9.67502593994140625e-4f
#define GM_PI_C 1.509957990978376432e-7f

static inline void gm_sincos(float x, float* s_out, float* c_out)
{
    float ax = gm_abs(x);
    float r;
    uint32_t sign;
    if (!(ax < INFINITY)) { *s_out = *c_out = x - x; return; }   /* inf, NaN -> NaN */
    if (ax <= 32768.0f) {
        float k = rintf(x * GM_ONE_OVER_PI);
        r = fmaf(k, -GM_PI_A, x);
        r = fmaf(k, -GM_PI_B, r);
        r = fmaf(k, -GM_PI_C, r);
        sign = (uint32_t)(int)k << 31;
    } else if (ax >= 2147483648.0f) {
        r = 0.0f;
        sign = 0u;
    } else {
        double xd = (double)x;
        double kd = rint(xd * 0.31830988618379067154);
        double rd = fma(-kd, 3.14159265346825122833, xd);
        rd = fma(-kd, 1.21542010130123844986e-10, rd);
        r = (float)rd;
        sign = (uint32_t)(int)kd << 31;
    }
    float z = r * r;
    float ps = fmaf(fmaf(fmaf(0x1.5c33e8p-19f, z, -0x1.9f5196p-13f), z, 0x1.110e1ap-7f), z, -0x1.555546p-3f);
    float s = fmaf(ps * z, r, r);
    float pc = fmaf(fmaf(fmaf(-0x1.184e1ap-22f, z, 0x1.9f7afap-16f), z, -0x1.6c1410p-10f), z, 0x1.55554ep-5f);
    float c = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    *s_out = gm_from_bits(gm_bits(s) ^ sign);
    *c_out = gm_from_bits(gm_bits(c) ^ sign);
}
static inline float gm_sin(float x) { float s, c; gm_sincos(x, &s, &c); return s; }
static inline float gm_cos(float x) { float s, c; gm_sincos(x, &s, &c); return c; }
static inline float gm_tan(float x) { float s, c; gm_sincos(x, &s, &c); return s / c; }

/* ── atan(y, x) ───────────────────────────────────────────────────────────────────────────
 * t = min(|x|,|y|) / max(|x|,|y|) in [0,1]; Cephes atanf reduction at tan(pi/8) and polynomial;
 * octant fix-ups. Signed zeros follow C atan2f. */
static inline float gm_atan2(float y, float x)
{
    if (x != x || y != y) return x + y;
    float ax = gm_abs(x), ay = gm_abs(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    float t;
    if (mx == 0.0f) t = 0.0f;
    else if (mn == INFINITY) t = 1.0f;
    else t = mn / mx;
    float base = 0.0f;
    if (t > 0.4142135623730950f) { base = 0.78539816339744830962f; t = (t - 1.0f) / (t + 1.0f); }
    float z = t * t;
    float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    float a = base + fmaf(p * z, t, t);
    if (ay > ax) a = 1.57079632679489661923f - a;
    if (gm_bits(x) >> 31) a = 3.14159265358979323846f - a;
    if (gm_bits(y) >> 31) a = -a;
    return a;
}

/* ── exp2 / log2 / pow (gravity force laws only) ──────────────────────────────────────────── */
static inline float gm_exp2(float x)
{
    if (x != x) return x;
    if (x > 128.0f) return INFINITY;
    if (x < -150.0f) return 0.0f;
    float k = rintf(x);
    float f = x - k;                       /* [-0.5, 0.5] */
    float p = 1.530178e-4f;                /* degree-6 Taylor/minimax of 2^f */
    p = fmaf(p, f, 1.339887e-3f);
    p = fmaf(p, f, 9.618437e-3f);
    p = fmaf(p, f, 5.550357e-2f);
    p = fmaf(p, f, 2.402265e-1f);
    p = fmaf(p, f, 6.931472e-1f);
    p = fmaf(p, f, 1.0f);
    int ki = (int)k;
    /* scale by 2^ki in two steps so that subnormal results round once */
    int k1 = ki / 2, k2 = ki - k1;
    float s1 = gm_from_bits((uint32_t)(k1 + 127) << 23);
    float s2 = gm_from_bits((uint32_t)(k2 + 127) << 23);
    return p * s1 * s2;
}
static inline float gm_log2(float x)
{
    if (x != x || x < 0.0f) return x - x + (x - x) / (x - x);   /* NaN */
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    uint32_t u = gm_bits(x);
    if (u < 0x00800000u) { x = x * 8388608.0f; u = gm_bits(x); e = -23; }   /* subnormal */
    /* split x = 2^k * m with m in [sqrt(1/2), sqrt(2)) */
    int k = (int)((u + 0x004afb0du) >> 23) - 127;
    e += k;
    float m = gm_from_bits((uint32_t)((int32_t)u - k * 0x00800000));
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    /* ln(m) = 2s(1 + z/3 + z^2/5 + z^3/7 + z^4/9) */
    float p = fmaf(fmaf(fmaf(fmaf(0.2222222222f, z, 0.2857142857f), z, 0.4f), z, 0.6666666667f), z, 2.0f);
    float ln_m = s * p;
    return fmaf(ln_m, 1.44269504088896340736f, (float)e);
}
static inline float gm_pow(float x, float y) { return gm_exp2(y * gm_log2(x)); }
static inline float gm_exp(float x) { return gm_exp2(x * 1.44269504088896340736f); }
static inline float gm_log(float x) { return gm_log2(x) * 0.69314718055994530942f; }

#endif /* GM_MATH_H */
