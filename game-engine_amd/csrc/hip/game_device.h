// game_device.h — hand-written gfx950 device functions for the fused per-pixel kernel.
//
// One __device__ function per GAME builtin; the kernel generator strings calls to them together
// in pipeline order with the stage arguments baked in as literals, so a whole layer lives in
// VGPRs from `p` to the packed RGBA8 store. Semantics follow the GLSL ES 3.00 text the reference
// compiler emits (game-compiler/src/codegen/glsl.rs; line numbers are cited per function), one
// FP32 rounding per GLSL operator: this file must be compiled with -ffp-contract=off, and fused
// multiply-adds appear only where written as __builtin_fmaf: inside the sin/cos/atan/exp2/log2 kernels
// and at the fixed contraction points (mix, the Hermite factor 3 - 2t, length, the noise remap, the fbm
// accumulate and octave rotation, the glow halo denominator, the RGBA8 pack). GLSL leaves contraction to
// the implementation; fixing the points makes every machine produce the same bits. hash2 is never fused.
//
// The file is compiled by hiprtc (no system headers) and, for the unit-test kernels, by hipcc.
#pragma once

#define GD static __device__ __forceinline__

typedef unsigned int gd_u32;
// One frame's uniform row, as read-only memory of the whole launch (constant address space: scalar, invariant loads).
typedef const __attribute__((address_space(4))) float* gd_uniform_row;
typedef gd_u32 gd_u32x4 __attribute__((ext_vector_type(4)));
typedef float gd_f32x2 __attribute__((ext_vector_type(2)));
typedef float gd_f32x4 __attribute__((ext_vector_type(4)));
typedef float gd_f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));   // 16-byte load at 8-byte alignment

// ── scalar helpers ──────────────────────────────────────────────────────────────────────
GD gd_u32 gd_bits(float f) { return __builtin_bit_cast(gd_u32, f); }
GD float gd_from_bits(gd_u32 u) { return __builtin_bit_cast(float, u); }
GD float gd_abs(float x) { return __builtin_fabsf(x); }
GD float gd_floor(float x) { return __builtin_floorf(x); }           // v_floor_f32
GD float gd_max(float a, float b) { return __builtin_fmaxf(a, b); }  // v_max_f32: NaN -> other operand, -0 < +0
GD float gd_min(float a, float b) { return __builtin_fminf(a, b); }  // v_min_f32
GD float gd_clamp(float x, float lo, float hi) { return gd_min(gd_max(x, lo), hi); }
// Correctly rounded square root. For 2^-100 <= x < inf the five-instruction sequence below (v_rsq_f32,
// g = x r, h = r / 2, one residual correction through two fmas) returns the IEEE result for every
// input; game_hip_check_exhaustive(3) compares gd_sqrt with the IEEE expansion over all 2^32 bit
// patterns. Everything else (zero, subnormal, negative, inf, NaN) takes the compiler's IEEE expansion.
GD float gd_sqrt_core(float x)
{
    float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
// inversesqrt(x) for 2^-100 <= x < inf: the exact 1 / sqrt(x) rounded ONCE to the nearest float, which is what makes it
// the same float on any machine - with one stated exception: for x two floats below a power of four,
// x = 4^k (1 - 2^-23), the exact value lies 4.5e-8 ulp above the midpoint of 2^-k and its upper neighbour, and the
// result is 2^-k (an error of 0.50000005 ulp instead of at most 0.5). From the v_rsq_f32 value y0 (within one ulp) one
// Newton step y0 + (y0 / 2)(1 - x y0^2) whose residual is exact to the last bit that matters: t = x y0 rounded, te its
// rounding error (exact through fma), so 1 - x y0^2 = (1 - t y0) - te y0. In the excepted class y0 = 2^-k, the residual is
// exactly 2^-23 and the step lands on the midpoint itself, which rounds to the even neighbour 2^-k. Seven instructions
// with the v_rsq_f32; checked against that definition, decided in exact arithmetic, over every input of the range by
// game_hip_check_exhaustive(5) and (7).
GD float gd_rsqrt_core(float x)
{
    float y0 = __builtin_amdgcn_rsqf(x);
    float t = x * y0, h = 0.5f * y0;
    float te = __builtin_fmaf(x, y0, -t);
    float e = __builtin_fmaf(-t, y0, 1.0f);
    e = __builtin_fmaf(-te, y0, e);
    return __builtin_fmaf(h, e, y0);
}
// Root (correctly rounded) and reciprocal root (as gd_rsqrt_core) together, from the same v_rsq_f32 value: eleven instructions.
GD void gd_sqrt_rsqrt_core(float x, float& root, float& inv)
{
    float y0 = __builtin_amdgcn_rsqf(x);
    float t = x * y0, h = 0.5f * y0;
    float d = __builtin_fmaf(-t, t, x);
    root = __builtin_fmaf(d, h, t);
    float te = __builtin_fmaf(x, y0, -t);
    float e = __builtin_fmaf(-t, y0, 1.0f);
    e = __builtin_fmaf(-te, y0, e);
    inv = __builtin_fmaf(h, e, y0);
}
// The short sequences and the IEEE expansions agree wherever the short ones are valid, so the choice could
// be made per wave instead of per lane (one scalar branch on "every active lane is in range"). Measured on
// MI355X that form is no faster than the exec-mask if/else, so the per-lane form is the default and
// GD_WAVE_GUARDS selects the other for A/B measurements.
#if defined(GD_ASSUME_RANGES)   // measurement only: every range test counts as passed (wrong results outside the ranges)
GD bool gd_wave_all(bool) { return true; }
#elif defined(GD_WAVE_GUARDS)
GD bool gd_wave_all(bool ok) { return __builtin_expect(__builtin_amdgcn_ballot_w64(!ok) == 0ull, 1); }
#else
GD bool gd_wave_all(bool ok) { return ok; }
#endif
// Range tests without branches. The short sequences above are valid on a range; taking the generic expansion outside
// it costs a branch per test, and the branches keep the compiler from overlapping the arithmetic of a lane's four
// pixels. A kernel may instead SPECULATE: every test counts as passed (the short sequence runs whatever its input:
// it is plain arithmetic and cannot fault) and a failed test is only recorded; once the lane's pixels are done the
// kernel looks at the record, once per wave, and in the rare case that some lane left a range evaluates those pixels
// again with speculate off, where every test is the real one. `speculate` is a literal in the generated kernel, so
// each use folds to one of its two forms.
struct GdGuard { bool speculate; bool left_range; };
GD bool gd_in_range(GdGuard& g, bool ok)
{
    if (g.speculate) { g.left_range = g.left_range || !ok; return true; }
    return gd_wave_all(ok);
}
GD float gd_sqrt(float x)
{
    if (__builtin_constant_p(x)) return __builtin_sqrtf(x);
    if (gd_wave_all(x >= 0x1p-100f && x < __builtin_inff())) return gd_sqrt_core(x);
    return __builtin_sqrtf(x);
}
GD float gd_sqrt(GdGuard& g, float x)
{
    if (__builtin_constant_p(x)) return __builtin_sqrtf(x);
    if (gd_in_range(g, x >= 0x1p-100f && x < __builtin_inff())) return gd_sqrt_core(x);
    return __builtin_sqrtf(x);
}
// v_fract_f32: x - floor(x) clamped to the largest float below 1; NaN for NaN and infinities.
GD float gd_fract(float x) { return __builtin_amdgcn_fractf(x); }
// Quotients whose divisor is a stage argument or a constant of the emitted GLSL are one correctly
// rounded reciprocal and one multiply (GLSL ES allows 2.5 ulp on a/b; this is within 1.5): with a
// literal divisor the reciprocal folds at kernel-compile time and the quotient costs one v_mul_f32.
// Correctly rounded 1 / y: v_rcp_f32 and one Newton step give the IEEE result for every y with
// 2^-126 <= |y| < 2^126 (game_hip_check_exhaustive(4), all 2^32 bit patterns); other inputs divide.
GD float gd_rcp_core(float y)
{
    float r = __builtin_amdgcn_rcpf(y);
    return __builtin_fmaf(__builtin_fmaf(-y, r, 1.0f), r, r);
}
GD float gd_rcp(float y)
{
    if (__builtin_constant_p(y)) return 1.0f / y;
    float a = gd_abs(y);
    if (gd_wave_all(a >= 0x1p-126f && a < 0x1p126f)) return gd_rcp_core(y);
    return 1.0f / y;
}
GD float gd_rcp(GdGuard& g, float y)
{
    if (__builtin_constant_p(y)) return 1.0f / y;
    float a = gd_abs(y);
    if (gd_in_range(g, a >= 0x1p-126f && a < 0x1p126f)) return gd_rcp_core(y);
    return 1.0f / y;
}
// The same for a divisor known to be at least 1 unless it is not a number (the glow halo's 1 + 16 d^2 I^2): the lower
// bound of the short form's range holds by construction, so one comparison decides (NaN and infinity fail it).
GD float gd_rcp_ge1(float y)
{
    if (__builtin_constant_p(y)) return 1.0f / y;
    if (gd_wave_all(y < 0x1p126f)) return gd_rcp_core(y);
    return 1.0f / y;
}
GD float gd_rcp_ge1(GdGuard& g, float y)
{
    if (__builtin_constant_p(y)) return 1.0f / y;
    if (gd_in_range(g, y < 0x1p126f)) return gd_rcp_core(y);
    return 1.0f / y;
}
// t / (t + 1) (the Reinhard tonemap, glsl.rs:473-478) correctly rounded, i.e. the IEEE division: the
// reciprocal of d = t + 1 as above, q = t r, and one correction q + (t - d q) r with the residual taken by an fma. Numerator
// and divisor depend on one value, so the claim has a finite proof: game_hip_check_exhaustive(8) compares this function
// with the compiler's division for all 2^32 bit patterns of t. Seven instructions in place of twelve. One input is kept
// out of the short form: t = -0, whose quotient is -0 while the correction's (+0) + (-0) gives +0.
GD float gd_reinhard_core(float t, float d)
{
    const float r = gd_rcp_core(d);
    const float q = t * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, t), r, q);
}
GD float gd_reinhard(GdGuard& g, float t)
{
    const float d = t + 1.0f;
    const float a = gd_abs(d);
    if (gd_in_range(g, a >= 0x1p-126f && a < 0x1p126f && gd_bits(t) != 0x80000000u)) return gd_reinhard_core(t, d);
    return t / d;
}
GD float gd_reinhard(float t) { GdGuard g; g.speculate = false; g.left_range = false; return gd_reinhard(g, t); }
GD float gd_mod(float x, float y) { return x - y * gd_floor(x * gd_rcp(y)); }
GD float gd_mod(GdGuard& g, float x, float y) { return x - y * gd_floor(x * gd_rcp(g, y)); }
GD float gd_step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
GD float gd_ceil(float x) { return -gd_floor(-x); }
GD float gd_round(float x) { return __builtin_rintf(x); }   // to nearest, ties to even (v_rndne_f32)
GD float gd_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
// mix(a, b, t) as a + t (b - a): one subtraction and one fused multiply-add (GLSL gives x (1 - t) + y t as the formula and
// leaves the evaluation to the implementation; this is the form that needs no 1 - t).
GD float gd_mix(float a, float b, float t) { return __builtin_fmaf(t, b - a, a); }
GD float gd_smoothstep(float e0, float e1, float x)
{
    float t = gd_clamp((x - e0) * gd_rcp(e1 - e0), 0.0f, 1.0f);
    return t * t * __builtin_fmaf(-2.0f, t, 3.0f);
}
GD float gd_smoothstep(GdGuard& g, float e0, float e1, float x)
{
    float t = gd_clamp((x - e0) * gd_rcp(g, e1 - e0), 0.0f, 1.0f);
    return t * t * __builtin_fmaf(-2.0f, t, 3.0f);
}
GD float gd_length2(float x, float y) { return gd_sqrt(__builtin_fmaf(x, x, y * y)); }
GD float gd_length2(GdGuard& g, float x, float y) { return gd_sqrt(g, __builtin_fmaf(x, x, y * y)); }
// `c ? t : f` of a stage-argument expression (expr.rs:132-142 words it select(f, t, c)): c is the 0 / 1 a comparison
// leaves, or any number - non-zero and not NaN counts as true
GD float gd_select(float c, float t, float f) { return (c != 0.0f && c == c) ? t : f; }

// ── sin / cos / atan ────────────────────────────────────────────────────────────────────
// GLSL does not fix the precision of these; the render path uses a fixed algorithm made of
// correctly rounded operations only, so every IEEE machine gets the same bits:
// k = rint(x / pi), r = x - k pi by a three-constant Cody-Waite reduction with fma for |x| <= 32768 (a two-constant
// double precision reduction up to 2^31, argument treated as 0 beyond), so |r| <= pi / 2 and
// sin(x) = (-1)^k S(r), cos(x) = (-1)^k C(r) with ONE odd and ONE even minimax polynomial on [-pi/2, pi/2]
// (S: r + r^3 (S1 + S2 z + S3 z^2 + S4 z^3), 1.8 ulp; C: 1 - z/2 + z^2 (Q0 + ... + Q3 z^3), 7.4e-8 absolute; z = r^2).
// One reduction serves both functions and no quadrant selects are needed: a sine alone costs 14 instructions.
// The reduction for |x| <= 32768 and the two polynomials, shared by both forms below.
GD void gd_sincos_small(float x, float& r, gd_u32& sign)
{
    float k = __builtin_rintf(x * 0.318309886f);
    r = __builtin_fmaf(k, -3.140625f, x);
    r = __builtin_fmaf(k, -9.67502593994140625e-4f, r);
    r = __builtin_fmaf(k, -1.509957990978376432e-7f, r);
    sign = (gd_u32)(int)k << 31;
}
GD void gd_sincos_poly(float r, gd_u32 sign, float& s_out, float& c_out)
{
    float z = r * r;
    float ps = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0x1.5c33e8p-19f, z, -0x1.9f5196p-13f), z, 0x1.110e1ap-7f), z, -0x1.555546p-3f);
    float s = __builtin_fmaf(ps * z, r, r);
    float pc = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-0x1.184e1ap-22f, z, 0x1.9f7afap-16f), z, -0x1.6c1410p-10f), z, 0x1.55554ep-5f);
    float c = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
#ifdef GD_SIN_NUDGE   // measurement only: sine and cosine off by GD_SIN_NUDGE units in the last place (away from zero if positive),
                      // as another correct implementation of the GLSL functions might return them
    s = gd_from_bits(gd_bits(s) + (gd_u32)(GD_SIN_NUDGE));
    c = gd_from_bits(gd_bits(c) + (gd_u32)(GD_SIN_NUDGE));
#endif
    s_out = gd_from_bits(gd_bits(s) ^ sign);
    c_out = gd_from_bits(gd_bits(c) ^ sign);
}
GD void gd_sincos(float x, float& s_out, float& c_out)
{
    float ax = gd_abs(x);
    float r;
    gd_u32 sign;
#if defined(GD_ASSUME_RANGES)
    if (true) {
#else
    if (!(ax < __builtin_inff())) { s_out = c_out = x - x; return; }
    if (ax <= 32768.0f) {
#endif
        gd_sincos_small(x, r, sign);
    } else if (ax >= 2147483648.0f) {
        r = 0.0f;
        sign = 0u;
    } else {
        double xd = (double)x;
        double kd = __builtin_rint(xd * 0.31830988618379067154);
        double rd = __builtin_fma(-kd, 3.14159265346825122833, xd);
        rd = __builtin_fma(-kd, 1.21542010130123844986e-10, rd);
        r = (float)rd;
        sign = (gd_u32)(int)kd << 31;
    }
    gd_sincos_poly(r, sign, s_out, c_out);
}
// Speculating, the short reduction runs for every input and an argument it is not made for (beyond 32768, infinite,
// NaN) is recorded; otherwise the form above.
GD void gd_sincos(GdGuard& g, float x, float& s_out, float& c_out)
{
    if (!g.speculate) { gd_sincos(x, s_out, c_out); return; }
    g.left_range = g.left_range || !(gd_abs(x) <= 32768.0f);
    float r;
    gd_u32 sign;
    gd_sincos_small(x, r, sign);
    gd_sincos_poly(r, sign, s_out, c_out);
}
GD float gd_sin(float x) { float s, c; gd_sincos(x, s, c); return s; }
GD float gd_cos(float x) { float s, c; gd_sincos(x, s, c); return c; }
GD float gd_tan(float x) { float s, c; gd_sincos(x, s, c); return s / c; }
GD float gd_sin(GdGuard& g, float x) { float s, c; gd_sincos(g, x, s, c); return s; }
GD float gd_cos(GdGuard& g, float x) { float s, c; gd_sincos(g, x, s, c); return c; }

GD float gd_atan2(float y, float x)
{
    if (x != x || y != y) return x + y;
    float ax = gd_abs(x), ay = gd_abs(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    float t;
    if (mx == 0.0f) t = 0.0f;
    else if (mn == __builtin_inff()) t = 1.0f;
    else t = mn / mx;
    float base = 0.0f;
    if (t > 0.4142135623730950f) { base = 0.78539816339744830962f; t = (t - 1.0f) / (t + 1.0f); }
    float z = t * t;
    float p = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z,
                             -3.33329491539e-1f);
    float a = base + __builtin_fmaf(p * z, t, t);
    if (ay > ax) a = 1.57079632679489661923f - a;
    if (gd_bits(x) >> 31) a = 3.14159265358979323846f - a;
    if (gd_bits(y) >> 31) a = -a;
    return a;
}

// exp2 / log2 / pow: used by gravity force laws only.
GD float gd_exp2(float x)
{
    if (x != x) return x;
    if (x > 128.0f) return __builtin_inff();
    if (x < -150.0f) return 0.0f;
    float k = __builtin_rintf(x);
    float f = x - k;
    float p = 1.530178e-4f;
    p = __builtin_fmaf(p, f, 1.339887e-3f);
    p = __builtin_fmaf(p, f, 9.618437e-3f);
    p = __builtin_fmaf(p, f, 5.550357e-2f);
    p = __builtin_fmaf(p, f, 2.402265e-1f);
    p = __builtin_fmaf(p, f, 6.931472e-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    int ki = (int)k;
    int k1 = ki / 2, k2 = ki - k1;
    float s1 = gd_from_bits((gd_u32)(k1 + 127) << 23);
    float s2 = gd_from_bits((gd_u32)(k2 + 127) << 23);
    return p * s1 * s2;
}
GD float gd_log2(float x)
{
    if (x != x || x < 0.0f) return x - x + (x - x) / (x - x);
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    int e = 0;
    gd_u32 u = gd_bits(x);
    if (u < 0x00800000u) { x = x * 8388608.0f; u = gd_bits(x); e = -23; }
    int k = (int)((u + 0x004afb0du) >> 23) - 127;
    e += k;
    float m = gd_from_bits((gd_u32)((int)u - k * 0x00800000));
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0.2222222222f, z, 0.2857142857f), z, 0.4f), z, 0.6666666667f), z, 2.0f);
    float ln_m = s * p;
    return __builtin_fmaf(ln_m, 1.44269504088896340736f, (float)e);
}
GD float gd_pow(float x, float y) { return gd_exp2(y * gd_log2(x)); }
GD float gd_exp(float x) { return gd_exp2(x * 1.44269504088896340736f); }
GD float gd_log(float x) { return gd_log2(x) * 0.69314718055994530942f; }

// ── noise (glsl.rs:132-220) ─────────────────────────────────────────────────────────────
// hash2 (glsl.rs:134-138). The three fract/multiply lanes of the GLSL vec3 collapse to two
// because p3.x == p3.z; the sum is evaluated left to right exactly as dot() is written.
GD float gd_hash2(float px, float py)
{
    float x = gd_fract(px * 0.1031f);
    float y = gd_fract(py * 0.1031f);
    float z = x;
    float d = x * (y + 33.33f) + y * (z + 33.33f) + z * (x + 33.33f);
    x = x + d; y = y + d; z = z + d;
    return gd_fract((x + y) * z);
}

// noise2 (glsl.rs:141-150): value noise, bilinear mix of four lattice hashes, remapped to [-1, 1].
GD float gd_noise2(float px, float py)
{
    float ix = gd_floor(px), iy = gd_floor(py);
    float fx = gd_fract(px), fy = gd_fract(py);
    float ux = fx * fx * __builtin_fmaf(-2.0f, fx, 3.0f);
    float uy = fy * fy * __builtin_fmaf(-2.0f, fy, 3.0f);
    float a = gd_hash2(ix, iy);
    float b = gd_hash2(ix + 1.0f, iy);
    float c = gd_hash2(ix, iy + 1.0f);
    float d = gd_hash2(ix + 1.0f, iy + 1.0f);
    return __builtin_fmaf(gd_mix(gd_mix(a, b, ux), gd_mix(c, d, ux), uy), 2.0f, -1.0f);
}

// Lattice-hash table. hash2 is a pure function of the lattice point, and a frame touches only a few
// thousand distinct points per octave while every pixel needs four of them per noise2 call, so the
// runtime tabulates hash2 over the integer points of [-512, 512]^2 once per device. Entry (iy + 512,
// ix + 512) holds what the cell whose lower left corner is (ix, iy) needs (GD_LUT_STRIDE entries of 16 bytes per row):
// {a, c, b - a, d - c} with a = hash2(ix, iy), b = hash2(ix + 1, iy), c = hash2(ix, iy + 1), d = hash2(ix + 1, iy + 1) -
// the two left corners and the two differences that mix(a, b, ux) = a + ux (b - a) and mix(c, d, ux) start from, each
// difference rounded once as the subtraction in gd_mix rounds it. noise2 then costs one range test, a three-instruction
// index and one 16-byte load instead of four hash evaluations and two subtractions. Points outside the table, and a
// null table, take the arithmetic path; both paths return the same bits because the table is filled by gd_hash2 and
// the same subtraction.
#define GD_LUT_STRIDE 2048u
#define GD_LUT_ROWS 1025u
#define GD_LUT_HALF 512
// One noise2 evaluation split in two, so that callers needing several values (the fbm octaves, the two
// components of a warp) can test all their lattice cells first, issue all table loads back to back, and
// only then combine: the loads of one call overlap instead of each waiting out its own latency.
// Byte offset of table entry (iy + 512 + d, ix + 512 + d) for integral ix, iy inside the table and d = 0 or -1, in a table
// of 8-byte entries (gd_lut_offset: the voronoi feature points) or 16-byte entries (gd_lut_offset16: the lattice hashes).
// The entry number k = (iy + 512 + d) * 2048 + (ix + 512 + d) is computed in floating point: iy * 2048 + ix is
// exact (below 2^21), adding 1.5 * 2^23 + (512 + d) * 2049 is exact too and leaves 0x4B400000 + k in the bits;
// shifting left by 3 (or 4) drops the exponent's top bits, and subtracting what is left of them gives 8 k (16 k).
// Three instructions: fma, add, shift-and-add.
GD gd_u32 gd_lut_offset(float ix, float iy, float d)
{
    const float t = __builtin_fmaf(iy, 2048.0f, ix);
    return (gd_bits(t + (12582912.0f + (512.0f + d) * 2049.0f)) << 3) - 0x5A000000u;
}
GD gd_u32 gd_lut_offset16(float ix, float iy, float d)
{
    const float t = __builtin_fmaf(iy, 2048.0f, ix);
    return (gd_bits(t + (12582912.0f + (512.0f + d) * 2049.0f)) << 4) - 0xB4000000u;
}
struct GdCell {
    float ix, iy;    // lattice cell
    float ux, uy;    // Hermite weights of the fractional part
    gd_u32 off;      // byte offset of entry (iy, ix) in the table (valid when ok)
    bool ok;         // all four corners are inside the table
};
GD GdCell gd_cell(float px, float py)
{
    GdCell c;
    c.ix = gd_floor(px); c.iy = gd_floor(py);
    float fx = gd_fract(px), fy = gd_fract(py);
    c.ux = fx * fx * __builtin_fmaf(-2.0f, fx, 3.0f);
    c.uy = fy * fy * __builtin_fmaf(-2.0f, fy, 3.0f);
    c.ok = gd_abs(c.ix) <= 511.0f && gd_abs(c.iy) <= 511.0f;
    c.off = gd_lut_offset16(c.ix, c.iy, 0.0f);
    return c;
}
// The left corners of a cell and the differences to their right neighbours: a = h(x, y), c = h(x, y + 1),
// ab = h(x + 1, y) - a, cd = h(x + 1, y + 1) - c.
struct GdCorners { float a, c, ab, cd; };
GD GdCorners gd_corners_load(const GdCell& c, const float* __restrict__ lut)
{
    const gd_f32x4 t = *(const gd_f32x4*)((const char*)lut + c.off);
    GdCorners k;
    k.a = t[0]; k.c = t[1]; k.ab = t[2]; k.cd = t[3];
    return k;
}
GD GdCorners gd_corners_hash(const GdCell& c)
{
    GdCorners k;
    k.a = gd_hash2(c.ix, c.iy);
    k.c = gd_hash2(c.ix, c.iy + 1.0f);
    k.ab = gd_hash2(c.ix + 1.0f, c.iy) - k.a;
    k.cd = gd_hash2(c.ix + 1.0f, c.iy + 1.0f) - k.c;
    return k;
}
// mix(a, b, ux) and mix(c, d, ux): the x-interpolated values on the lower and the upper edge of the cell
GD float gd_edge_lo(const GdCorners& k, float ux) { return __builtin_fmaf(ux, k.ab, k.a); }
GD float gd_edge_hi(const GdCorners& k, float ux) { return __builtin_fmaf(ux, k.cd, k.c); }
GD float gd_noise2_combine(const GdCell& c, const GdCorners& k)
{
    return __builtin_fmaf(gd_mix(gd_edge_lo(k, c.ux), gd_edge_hi(k, c.ux), c.uy), 2.0f, -1.0f);
}
GD float gd_noise2_lut(float px, float py, const float* __restrict__ lut)
{
    GdCell c = gd_cell(px, py);
    GdCorners k;
    if (lut != nullptr && c.ok) k = gd_corners_load(c, lut);
    else k = gd_corners_hash(c);
    return gd_noise2_combine(c, k);
}
// Two values at once (the pairs of domain_warp, displace and the central differences of curl_noise).
GD void gd_noise2_pair_lut(float ax, float ay, float bx, float by, const float* __restrict__ lut, float& na, float& nb)
{
    GdCell ca = gd_cell(ax, ay), cb = gd_cell(bx, by);
    GdCorners ka, kb;
    if (lut != nullptr && ca.ok && cb.ok) { ka = gd_corners_load(ca, lut); kb = gd_corners_load(cb, lut); }
    else { ka = gd_corners_hash(ca); kb = gd_corners_hash(cb); }
    na = gd_noise2_combine(ca, ka); nb = gd_noise2_combine(cb, kb);
}

// The lattice cells of fbm2's octaves (glsl.rs:161-167): octave i samples at q_i * frequency_i with q_0 = p,
// q_(i+1) = rot q_i and frequency_i = lacunarity^i. `first`: cells below that octave are left to the caller.
// folded: with a lacunarity that is a power of two the sample point Q_i = q_i * frequency_i can be carried instead of
// q_i - Q_(i+1) = (lacunarity rot) Q_i, the rotation's constants scaled exactly - which saves the two multiplications by
// the frequency in every octave. Scaling by a power of two commutes with every rounding of the recurrence as long as no
// intermediate is subnormal or overflows, so the cells are the same bits; gd_fbm_table_ok admits the folded form only
// where that holds.
GD bool gd_fbm_folds(float lacunarity)
{
    return __builtin_constant_p(lacunarity) && (lacunarity == 1.0f || lacunarity == 2.0f || lacunarity == 4.0f || lacunarity == 8.0f);
}
GD void gd_fbm_cells(float px, float py, int octaves, float lacunarity, int first, GdCell* cell, bool folded)
{
    if (folded) {
        const float c8 = 0.8f * lacunarity, c6 = 0.6f * lacunarity;   // exact: lacunarity is a power of two
        float Qx = px, Qy = py;
#pragma unroll
        for (int i = 0; i < octaves; i++) {
            if (i >= first) cell[i] = gd_cell(Qx, Qy);
            float nx = __builtin_fmaf(Qx, c8, -(Qy * c6));
            float ny = __builtin_fmaf(Qx, c6, Qy * c8);
            Qx = nx; Qy = ny;
        }
        return;
    }
    float frequency = 1.0f, qx = px, qy = py;
#pragma unroll
    for (int i = 0; i < octaves; i++) {
        if (i >= first) cell[i] = gd_cell(qx * frequency, qy * frequency);
        frequency = frequency * lacunarity;
        float nx = __builtin_fmaf(qx, 0.8f, -(qy * 0.6f));
        float ny = __builtin_fmaf(qx, 0.6f, qy * 0.8f);
        qx = nx; qy = ny;
    }
}
// Amplitudes and their sum, and the test that lets all octaves read the table. One range test for all octaves: octave i
// samples at (rotation^i p) * frequency_i, whose Euclidean norm is |p| * frequency_i up to a few ulps, so
// |p|^2 * max(frequency)^2 <= 509^2 keeps every lattice cell of every octave within [-510, 510] (two cells of margin
// inside the table; NaN fails the test). Where the folded recurrence will be used both coordinates must also be at
// least 2^-40 in size - tested as |px py| >= 2^-30, one multiplication and one compare, which implies it because the norm
// test has already bounded both by 509 < 2^9: then one of q's coordinates stays above 2^-41 through every rotation (they
// preserve the norm), a product with 0.8 or 0.6 is never subnormal, and a difference that cancels is zero or at least 2^-88 - nothing is
// subnormal in either form, so the folded and the literal cells agree bit for bit. (A coordinate of exactly zero would
// be fine too; it takes the other path with the tiny ones, which costs the centre row and column of an odd-sized frame
// a few hash evaluations.)
GD bool gd_fbm_table_ok(float px, float py, int octaves, float persistence, float lacunarity, float* amp, float& max_val)
{
    float amplitude = 1.0f, frequency = 1.0f, fmax = 1.0f;
    max_val = 0.0f;
#pragma unroll
    for (int i = 0; i < octaves; i++) {
        fmax = !(gd_abs(frequency) <= fmax) ? gd_abs(frequency) : fmax;   // a NaN frequency makes fmax NaN: the test fails
        amp[i] = amplitude;
        max_val = max_val + amplitude;
        amplitude = amplitude * persistence;
        frequency = frequency * lacunarity;
    }
    bool ok = __builtin_fmaf(px, px, py * py) * (fmax * fmax) <= 259081.0f;
    if (gd_fbm_folds(lacunarity)) ok = ok && gd_abs(px * py) >= 0x1p-30f;
    return ok;
}

GD int gd_octaves(float o)   // int(x), clamped to [0, 32]
{
    if (!(o >= 1.0f)) return 0;
    if (o >= 32.0f) return 32;
    return (int)o;
}

// fbm2 (glsl.rs:155-169). With a literal octave count the loop unrolls and the amplitude /
// frequency / max_val recurrences fold to constants.
GD float gd_fbm2(float px, float py, int octaves, float persistence, float lacunarity, const float* __restrict__ lut = nullptr)
{
    if (lut != nullptr && __builtin_constant_p(octaves) && octaves >= 1 && octaves <= 8) {
        // literal octave count: all cells first, then (when every cell is inside the table) all loads, then the sum
        GdCell cell[8];
        float amp[8];
        float max_val = 0.0f;
        GdCorners k[8];
        if (gd_fbm_table_ok(px, py, octaves, persistence, lacunarity, amp, max_val)) {
            gd_fbm_cells(px, py, octaves, lacunarity, 0, cell, gd_fbm_folds(lacunarity));
#pragma unroll
            for (int i = 0; i < octaves; i++) k[i] = gd_corners_load(cell[i], lut);
        } else {
            gd_fbm_cells(px, py, octaves, lacunarity, 0, cell, false);
#pragma unroll
            for (int i = 0; i < octaves; i++) k[i] = gd_corners_hash(cell[i]);
        }
        float value = 0.0f;
#pragma unroll
        for (int i = 0; i < octaves; i++) value = __builtin_fmaf(gd_noise2_combine(cell[i], k[i]), amp[i], value);
        return value * gd_rcp(max_val);
    }
    float value = 0.0f, amplitude = 1.0f, frequency = 1.0f, max_val = 0.0f;
    float qx = px, qy = py;
#pragma unroll
    for (int i = 0; i < octaves; i++) {
        value = __builtin_fmaf(gd_noise2_lut(qx * frequency, qy * frequency, lut), amplitude, value);
        max_val = max_val + amplitude;
        amplitude = amplitude * persistence;
        frequency = frequency * lacunarity;
        float nx = __builtin_fmaf(qx, 0.8f, -(qy * 0.6f));
        float ny = __builtin_fmaf(qx, 0.6f, qy * 0.8f);
        qx = nx; qy = ny;
    }
    return value * gd_rcp(max_val);
}

GD float gd_permute(float x) { return gd_mod((x * 34.0f + 1.0f) * x, 289.0f); }   // glsl.rs:173-175

// What a simplex corner contributes beyond its distance falloff depends only on the permuted lattice value p:
// the gradient (a0, h) and its normalisation factor (glsl.rs:187-194).
struct GdSimplexCorner { float a0, h, nrm; };
GD GdSimplexCorner gd_simplex_corner(float p)
{
    const float Cw = 0.024390243902439f;
    GdSimplexCorner c;
    float xv = 2.0f * gd_fract(p * Cw) - 1.0f;
    c.h = gd_abs(xv) - 0.5f;
    c.a0 = xv - gd_floor(xv + 0.5f);
    c.nrm = 1.79284291400159f - 0.85373472095314f * (c.a0 * c.a0 + c.h * c.h);
    return c;
}
// p = permute(permute(Y) + X) for the lattice point (X, Y) after the mod-289 wrap: a pure function of two
// integers in [0, 289], so the runtime tabulates gd_simplex_corner of it once per device: 290 rows of 290
// entries {a0, h, nrm, 0} (1.35 MB), filled by this same code. A pixel then reads its three corners as three
// 16-byte loads instead of evaluating six permutes and three gradient set-ups.
#define GD_SIMPLEX_DIM 290u
GD float gd_simplex2d(float vx, float vy, const float* __restrict__ slut = nullptr)   // glsl.rs:176-198
{
    const float Cx = 0.211324865405187f, Cy = 0.366025403784439f, Cz = -0.577350269189626f;
    float dv = vx * Cy + vy * Cy;
    float ix = gd_floor(vx + dv), iy = gd_floor(vy + dv);
    float di = ix * Cx + iy * Cx;
    float x0x = vx - ix + di, x0y = vy - iy + di;
    bool gt = x0x > x0y;
    float i1x = gt ? 1.0f : 0.0f, i1y = gt ? 0.0f : 1.0f;
    float x12x = x0x + Cx, x12y = x0y + Cx, x12z = x0x + Cz, x12w = x0y + Cz;
    x12x = x12x - i1x; x12y = x12y - i1y;
    ix = gd_mod(ix, 289.0f); iy = gd_mod(iy, 289.0f);
    GdSimplexCorner c0, c1, c2;
    // the wrap leaves integers in [0, 288] for every finite coordinate below 2^24; anything else (NaN included)
    // fails the test and takes the arithmetic path
    if (slut != nullptr && ix >= 0.0f && ix <= 288.0f && iy >= 0.0f && iy <= 288.0f) {
        // entry number k = iy * 290 + ix as exact float arithmetic; (bits(k + 1.5 * 2^23) << 4) - 0xB4000000 = 16 k
        const gd_u32 off = (gd_bits(__builtin_fmaf(iy, 290.0f, ix) + 12582912.0f) << 4) - 0xB4000000u;
        const char* base = (const char*)slut + off;
        const gd_f32x4_a8 e0 = *(const gd_f32x4_a8*)base;
        const gd_f32x4_a8 e1 = *(const gd_f32x4_a8*)(base + (gt ? 16u : GD_SIMPLEX_DIM * 16u));
        const gd_f32x4_a8 e2 = *(const gd_f32x4_a8*)(base + (GD_SIMPLEX_DIM + 1u) * 16u);
        c0.a0 = e0[0]; c0.h = e0[1]; c0.nrm = e0[2];
        c1.a0 = e1[0]; c1.h = e1[1]; c1.nrm = e1[2];
        c2.a0 = e2[0]; c2.h = e2[1]; c2.nrm = e2[2];
    } else {
        c0 = gd_simplex_corner(gd_permute(gd_permute(iy + 0.0f) + ix + 0.0f));
        c1 = gd_simplex_corner(gd_permute(gd_permute(iy + i1y) + ix + i1x));
        c2 = gd_simplex_corner(gd_permute(gd_permute(iy + 1.0f) + ix + 1.0f));
    }
    float m0 = gd_max(0.5f - (x0x * x0x + x0y * x0y), 0.0f);
    float m1 = gd_max(0.5f - (x12x * x12x + x12y * x12y), 0.0f);
    float m2 = gd_max(0.5f - (x12z * x12z + x12w * x12w), 0.0f);
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    m0 = m0 * c0.nrm;
    m1 = m1 * c1.nrm;
    m2 = m2 * c2.nrm;
    float g0 = c0.a0 * x0x + c0.h * x0y;
    float g1 = c1.a0 * x12x + c1.h * x12y;
    float g2 = c2.a0 * x12z + c2.h * x12w;
    return 130.0f * (m0 * g0 + m1 * g1 + m2 * g2);
}

GD float gd_voronoi2d(float px, float py, float scale)   // glsl.rs:202-219
{
    float psx = px * scale, psy = py * scale;
    float ipx = gd_floor(psx), ipy = gd_floor(psy);
    float fpx = gd_fract(psx), fpy = gd_fract(psy);
    float d = 1.0f;
#pragma unroll
    for (int y = -1; y <= 1; y++) {
#pragma unroll
        for (int x = -1; x <= 1; x++) {
            float nx = (float)x, ny = (float)y;
            float cx = ipx + nx, cy = ipy + ny;
            float hx = gd_fract(gd_sin(cx * 127.1f + cy * 311.7f) * 43758.5453f);
            float hy = gd_fract(gd_sin(cx * 269.5f + cy * 183.3f) * 43758.5453f);
            float ptx = nx + hx - fpx, pty = ny + hy - fpy;
            d = gd_min(d, ptx * ptx + pty * pty);
        }
    }
    return gd_sqrt(d);
}

// The feature point of a voronoi cell is a pure function of the integer cell (two sin hashes), so it is
// tabulated like hash2: entry (cy + 512, cx + 512) of the second table holds {hx, hy}. A pixel reads the
// 3 x 3 block around its cell as three rows of 24 bytes instead of evaluating 18 sines.
GD float gd_voronoi2d_lut(float px, float py, float scale, const float* __restrict__ vlut)
{
    float psx = px * scale, psy = py * scale;
    float ipx = gd_floor(psx), ipy = gd_floor(psy);
    if (!(vlut != nullptr && gd_abs(ipx) <= 510.0f && gd_abs(ipy) <= 510.0f)) return gd_voronoi2d(px, py, scale);
    float fpx = gd_fract(psx), fpy = gd_fract(psy);
    const char* base = (const char*)vlut + gd_lut_offset(ipx, ipy, -1.0f);   // entry (ipy - 1, ipx - 1)
    float d = 1.0f;
#pragma unroll
    for (int y = -1; y <= 1; y++) {
        const char* row = base + (gd_u32)(y + 1) * (GD_LUT_STRIDE * 8u);
        gd_f32x4_a8 e01 = *(const gd_f32x4_a8*)row;
        gd_f32x2 e2 = *(const gd_f32x2*)(row + 16);
        float hx[3] = {e01[0], e01[2], e2[0]}, hy[3] = {e01[1], e01[3], e2[1]};
#pragma unroll
        for (int x = -1; x <= 1; x++) {
            float nx = (float)x, ny = (float)y;
            float ptx = nx + hx[x + 1] - fpx, pty = ny + hy[x + 1] - fpy;
            d = gd_min(d, ptx * ptx + pty * pty);
        }
    }
    return gd_sqrt(d);
}

// max(d, 0)^2 * I * I of the glow halo, evaluated left to right. With a literal intensity that is a power of two not
// below 1 the two multiplications by I are exact scalings, so one multiplication by I^2 gives the same float (both forms
// overflow to infinity together).
GD float gd_glow_spread(float d, float intensity)
{
    const float m = gd_max(d, 0.0f);
    if (__builtin_constant_p(intensity) && intensity >= 1.0f && intensity <= 0x1p60f && (gd_bits(intensity) & 0x007fffffu) == 0u)
        return m * m * (intensity * intensity);
    return m * m * intensity * intensity;
}
GD float gd_apply_glow(float d, float intensity)   // glsl.rs:102-107
{
    float core = gd_smoothstep(0.005f, -0.005f, d);
    // the halo quotient is one correctly rounded reciprocal and one multiply (the denominator is >= 1 or not finite)
    float halo = intensity * gd_rcp_ge1(__builtin_fmaf(gd_glow_spread(d, intensity), 16.0f, 1.0f));
    return core + halo;
}

GD float gd_apply_glow(GdGuard& g, float d, float intensity)
{
    float core = gd_smoothstep(g, 0.005f, -0.005f, d);
    float halo = intensity * gd_rcp_ge1(g, __builtin_fmaf(gd_glow_spread(d, intensity), 16.0f, 1.0f));
    return core + halo;
}

// ── pipeline state ──────────────────────────────────────────────────────────────────────
// Per-pixel registers of one layer: position p, the signed distance, the colour, plus the
// frame-constant values every stage may read.
struct GdPix {
    float px, py;          // `p`
    float sdf;             // `sdf_result`
    float r, g, b, a;      // `color_result`
    float uvx, uvy;        // `uv`
    float uvlen, uvinv;    // length(uv) and 1 / length(uv), set by gd_uv_length for the stages that read them
    float time;            // fract(u_time / 120) * 120
    float bass, mid, treble;
    const float* lut;      // lattice-hash table (uniform; may be null)
    const float* vlut;     // voronoi feature-point table (uniform; may be null)
    const float* slut;     // simplex corner table (uniform; may be null)
    GdGuard guard;         // how range tests are taken (see gd_in_range)
};

// Projection blocks (game-compiler/src/codegen/project.rs): warps of v_uv applied before the layers run.
GD void gd_project_dome(float vx, float vy, float fov_rad, float& ox, float& oy)   // project.rs:62-69
{
    float cx = vx - 0.5f, cy = vy - 0.5f;
    float r = gd_length2(cx, cy) * 2.0f;
    float theta = gd_atan2(cy, cx);
    float phi = r * fov_rad * 0.5f;
    float sp = gd_sin(phi);
    float st, ct;
    gd_sincos(theta, st, ct);
    ox = sp * ct * 0.5f + 0.5f;
    oy = sp * st * 0.5f + 0.5f;
}
GD void gd_project_led(float vx, float count, float& ox, float& oy)                // project.rs:128-137
{
    ox = gd_floor(vx * count) / count;   // t = f32(led_idx) / f32(count)
    oy = 0.5f;
}

// length(uv) and 1 / length(uv), once per pixel: vignette and chromatic (and their WebGPU wordings) all start from
// length(uv), and normalize(uv) is uv * inversesqrt(dot(uv, uv)). For d2 = dot(uv, uv) in [2^-100, inf) they are the correctly
// rounded root and gd_rsqrt_core; outside that range (uv = 0 among them: 0 and +inf) sqrt(d2) and 1 / sqrt(d2).
GD void gd_uv_length(GdPix& s)
{
    const float d2 = __builtin_fmaf(s.uvx, s.uvx, s.uvy * s.uvy);
    // uv is a pixel-centre coordinate: finite, within about [-1, 1], and each component either 0 or at least 2^-25 in size,
    // so d2 lies in [2^-50, 2] unless both are 0. Speculating kernels record exactly that case; the two comparisons depend
    // on the column and on the row alone and leave the per-pixel work.
    const bool ok = s.guard.speculate ? (s.uvx != 0.0f || s.uvy != 0.0f) : (d2 >= 0x1p-100f && d2 < __builtin_inff());
    if (gd_in_range(s.guard, ok)) gd_sqrt_rsqrt_core(d2, s.uvlen, s.uvinv);
    else { s.uvlen = __builtin_sqrtf(d2); s.uvinv = 1.0f / s.uvlen; }
}

// SDF generators: Position -> Sdf
GD void gb_circle(GdPix& s, float radius) { s.sdf = gd_length2(s.guard, s.px, s.py) - radius; }                 // glsl.rs:265-268
GD void gb_ring(GdPix& s, float radius, float width) { s.sdf = gd_abs(gd_length2(s.guard, s.px, s.py) - radius) - width; }  // :269-273
GD void gb_star(GdPix& s, float points, float radius, float inner)                                      // :274-288
{
    float an = 3.14159265359f / points;
    float a = gd_atan2(s.py, s.px);
    float sector = gd_mod(s.guard, a + an, 2.0f * an) - an;
    float len = gd_length2(s.guard, s.px, s.py);
    float sc, ss;
    gd_sincos(s.guard, sector, ss, sc);
    float spx = sc * len, spy = gd_abs(ss) * len;
    float ans, anc;
    gd_sincos(s.guard, an, ans, anc);
    float apx = radius, apy = 0.0f;
    float bpx = inner * anc, bpy = inner * ans;
    float bax = bpx - apx, bay = bpy - apy;
    float t = gd_clamp(((spx - apx) * bax + (spy - apy) * bay) / (bax * bax + bay * bay), 0.0f, 1.0f);
    s.sdf = gd_length2(s.guard, spx - apx - bax * t, spy - apy - bay * t);
}
GD void gb_box(GdPix& s, float w, float h)                                                              // :289-294
{
    float dx = gd_abs(s.px) - w, dy = gd_abs(s.py) - h;
    s.sdf = gd_length2(s.guard, gd_max(dx, 0.0f), gd_max(dy, 0.0f)) + gd_min(gd_max(dx, dy), 0.0f);
}
GD void gb_polygon(GdPix& s, float sides, float radius)                                                 // :295-304
{
    float a = gd_atan2(s.py, s.px);
    float r = gd_length2(s.guard, s.px, s.py);
    float an = 6.28318530718f / sides;
    float sector = gd_cos(s.guard, gd_floor(0.5f + a / an) * an - a) * r;
    s.sdf = sector - radius;
}
GD void gb_simplex(GdPix& s, float scale) { s.sdf = gd_simplex2d(s.px * scale, s.py * scale, s.slut); }         // :305-308
GD void gb_voronoi(GdPix& s, float scale) { s.sdf = gd_voronoi2d_lut(s.px, s.py, scale, s.vlut); }                  // :309-312
GD void gb_concentric_waves(GdPix& s, float amplitude, float width, float frequency)                    // :313-319
{
    float dist = gd_length2(s.guard, s.px, s.py);
    s.sdf = amplitude * gd_sin(s.guard, dist * frequency * 6.28318530718f - s.time * width);
}
GD void gb_fbm(GdPix& s, float scale, int octaves, float persistence, float lacunarity)                 // :320-326
{
    s.sdf = gd_fbm2(s.px * scale, s.py * scale, octaves, persistence, lacunarity, s.lut);
}

// Transforms: Position -> Position
GD void gb_rotate(GdPix& s, float angle)                                                                // :329-334
{
    float rs, rc;
    gd_sincos(s.guard, angle, rs, rc);
    float nx = s.px * rc - s.py * rs, ny = s.px * rs + s.py * rc;
    s.px = nx; s.py = ny;
}
GD void gb_translate(GdPix& s, float x, float y) { s.px = s.px - x; s.py = s.py - y; }                  // :335-339
GD void gb_scale(GdPix& s, float k) { float r = gd_rcp(s.guard, k); s.px = s.px * r; s.py = s.py * r; }                               // :340-343
GD void gb_twist(GdPix& s, float amount)                                                                // :344-349
{
    float ts, tc;
    gd_sincos(s.guard, s.py * amount, ts, tc);
    float nx = s.px * tc - s.py * ts, ny = s.px * ts + s.py * tc;
    s.px = nx; s.py = ny;
}
GD void gb_mirror(GdPix& s, float axis)                                                                 // :350-354
{
    if (axis < 0.5f) s.px = gd_abs(s.px); else s.py = gd_abs(s.py);
}
GD void gb_repeat(GdPix& s, float count)                                                                // :355-361
{
    float ra = 6.28318530718f / count;
    float a = gd_atan2(s.py, s.px);
    a = gd_mod(s.guard, a + ra * 0.5f, ra) - ra * 0.5f;
    float len = gd_length2(s.guard, s.px, s.py);
    float as, ac;
    gd_sincos(s.guard, a, as, ac);
    s.px = ac * len; s.py = as * len;
}
GD void gb_domain_warp(GdPix& s, float amount, float freq)                                              // :362-366
{
    float n0, n1;
    gd_noise2_pair_lut(s.px * freq, s.py * freq, s.px * freq + 5.2f, s.py * freq + 1.3f, s.lut, n0, n1);
    s.px = s.px + n0 * amount; s.py = s.py + n1 * amount;
}
// domain_warp as the first stage of a layer, inside the loop over a lane's rows. The position is then the pixel's own
// p. Its x half, and with it the lattice column and the Hermite weight ux of both noise samples, is the same on every row
// of the lane; its y half is the same for every lane of the workgroup, so one lane per row works it out once
// (gd_warp_rows_y, into shared memory) and the others read it; and the lattice row changes once in hundreds of image
// rows, so the x-interpolated value on the cell's lower edge (mix(a, b, ux)) and its difference to the one on the upper
// edge (mix(c, d, ux) - mix(a, b, ux), the subtraction the outer mix starts with) are kept from row to row and fetched
// again only when the sample enters another lattice row. Every value is computed by the same operations as in
// gb_domain_warp, so the result is the same.
struct GdWarpRows { float iy0, iy1, lo0, d0, lo1, d1; };
struct __attribute__((aligned(16))) GdWarpY { float iy0, uy0, iy1, uy1, ok, pad0, pad1, pad2; };   // y halves of both samples
GD void gd_warp_rows_reset(GdWarpRows& k) { k.iy0 = k.iy1 = __builtin_nanf(""); k.lo0 = k.d0 = k.lo1 = k.d1 = 0.0f; }
GD void gd_warp_rows_y(float py, float freq, float* o)
{
    const float a = py * freq, b = py * freq + 1.3f;
    const float iy0 = gd_floor(a), f0 = gd_fract(a), iy1 = gd_floor(b), f1 = gd_fract(b);
    const float uy0 = f0 * f0 * __builtin_fmaf(-2.0f, f0, 3.0f), uy1 = f1 * f1 * __builtin_fmaf(-2.0f, f1, 3.0f);
    o[0] = iy0; o[1] = uy0;
    o[2] = iy1; o[3] = uy1;
    o[4] = (gd_abs(iy0) <= 511.0f && gd_abs(iy1) <= 511.0f) ? 1.0f : 0.0f;
    o[5] = o[6] = o[7] = 0.0f;
}
GD void gb_domain_warp_rows(GdPix& s, float amount, float freq, GdWarpRows& k, const GdWarpY& y)
{
#ifdef GD_NO_WARP_ROWS   // A/B measurements: the plain form
    { float n0, n1; gd_noise2_pair_lut(s.px * freq, s.py * freq, s.px * freq + 5.2f, s.py * freq + 1.3f, s.lut, n0, n1); s.px = s.px + n0 * amount; s.py = s.py + n1 * amount; return; }
#endif
    const float ax = s.px * freq, bx = s.px * freq + 5.2f;
    const float fa = gd_fract(ax), fb = gd_fract(bx);
    GdCell ca, cb;
    ca.ix = gd_floor(ax); ca.ux = fa * fa * __builtin_fmaf(-2.0f, fa, 3.0f); ca.iy = y.iy0; ca.uy = y.uy0;
    cb.ix = gd_floor(bx); cb.ux = fb * fb * __builtin_fmaf(-2.0f, fb, 3.0f); cb.iy = y.iy1; cb.uy = y.uy1;
    float n0, n1;
    const bool table = s.lut != nullptr && gd_abs(ca.ix) <= 511.0f && gd_abs(cb.ix) <= 511.0f && y.ok != 0.0f;
    if (table && ca.iy != k.iy0) {
        ca.off = gd_lut_offset16(ca.ix, ca.iy, 0.0f);
        GdCorners c = gd_corners_load(ca, s.lut);
        k.lo0 = gd_edge_lo(c, ca.ux); k.d0 = gd_edge_hi(c, ca.ux) - k.lo0; k.iy0 = ca.iy;
    }
    if (table && cb.iy != k.iy1) {
        cb.off = gd_lut_offset16(cb.ix, cb.iy, 0.0f);
        GdCorners c = gd_corners_load(cb, s.lut);
        k.lo1 = gd_edge_lo(c, cb.ux); k.d1 = gd_edge_hi(c, cb.ux) - k.lo1; k.iy1 = cb.iy;
    }
    if (table) {
        n0 = __builtin_fmaf(__builtin_fmaf(y.uy0, k.d0, k.lo0), 2.0f, -1.0f);   // mix(lo, hi, uy) * 2 - 1
        n1 = __builtin_fmaf(__builtin_fmaf(y.uy1, k.d1, k.lo1), 2.0f, -1.0f);
    } else {
        n0 = gd_noise2_combine(ca, gd_corners_hash(ca));
        n1 = gd_noise2_combine(cb, gd_corners_hash(cb));
    }
    s.px = s.px + n0 * amount; s.py = s.py + n1 * amount;
}
// voronoi and fbm as the first stage of a layer get the same treatment (c2b: the three layers of tutorial-05-noise).
// voronoi: the nine feature points of the 3 x 3 block change only with the lattice row, and of a candidate's squared
// distance (nx + hx - fx)^2 + (ny + hy - fy)^2 the first square and ny + hy do not depend on the image row at all.
struct GdVoronoiRows { float iy; float px2[9]; float nyh[9]; };
GD void gd_voronoi_rows_reset(GdVoronoiRows& k)
{
    k.iy = __builtin_nanf("");
#pragma unroll
    for (int i = 0; i < 9; i++) k.px2[i] = k.nyh[i] = 0.0f;
}
GD void gd_voronoi_rows_y(float py, float scale, float* o)   // {floor, fract, inside the table, -}
{
    const float psy = py * scale;
    o[0] = gd_floor(psy); o[1] = gd_fract(psy); o[2] = gd_abs(o[0]) <= 510.0f ? 1.0f : 0.0f; o[3] = 0.0f;
}
GD void gb_voronoi_rows(GdPix& s, float scale, GdVoronoiRows& k, const float* y)
{
    const float psx = s.px * scale;
    const float ipx = gd_floor(psx), fpx = gd_fract(psx);
    const float ipy = y[0], fpy = y[1];
    const bool table = s.vlut != nullptr && gd_abs(ipx) <= 510.0f && y[2] != 0.0f;
    if (table && ipy != k.iy) {
        const char* base = (const char*)s.vlut + gd_lut_offset(ipx, ipy, -1.0f);   // entry (ipy - 1, ipx - 1)
#pragma unroll
        for (int yy = -1; yy <= 1; yy++) {
            const char* row = base + (gd_u32)(yy + 1) * (GD_LUT_STRIDE * 8u);
            gd_f32x4_a8 e01 = *(const gd_f32x4_a8*)row;
            gd_f32x2 e2 = *(const gd_f32x2*)(row + 16);
            float hx[3] = {e01[0], e01[2], e2[0]}, hy[3] = {e01[1], e01[3], e2[1]};
#pragma unroll
            for (int xx = -1; xx <= 1; xx++) {
                const float nx = (float)xx, ny = (float)yy;
                const float ptx = nx + hx[xx + 1] - fpx;
                k.px2[(yy + 1) * 3 + xx + 1] = ptx * ptx;
                k.nyh[(yy + 1) * 3 + xx + 1] = ny + hy[xx + 1];
            }
        }
        k.iy = ipy;
    }
    if (!table) { s.sdf = gd_voronoi2d(s.px, s.py, scale); return; }
    float d = 1.0f;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const float pty = k.nyh[c] - fpy;
        d = gd_min(d, k.px2[c] + pty * pty);
    }
    s.sdf = gd_sqrt(d);
}
// fbm with a literal octave count: octave 0 samples at p * scale itself (the rotation starts with octave 1), so its
// lattice column, its x weight and - from row to row - its x-interpolated edge values stay.
struct GdFbmRows { float iy, lo, d; };   // lattice row, mix(a, b, ux), mix(c, d, ux) - mix(a, b, ux) of octave 0
GD void gd_fbm_rows_reset(GdFbmRows& k) { k.iy = __builtin_nanf(""); k.lo = k.d = 0.0f; }
GD void gd_fbm_rows_y(float py, float scale, float* o)   // {lattice row, uy, row inside the table, -} of octave 0
{
    const float a = py * scale;
    const float f = gd_fract(a);
    o[0] = gd_floor(a); o[1] = f * f * __builtin_fmaf(-2.0f, f, 3.0f); o[2] = gd_abs(o[0]) <= 511.0f ? 1.0f : 0.0f; o[3] = 0.0f;
}
GD void gb_fbm_rows(GdPix& s, float scale, int octaves, float persistence, float lacunarity, GdFbmRows& k, const float* y)
{
    const float px = s.px * scale, py = s.py * scale;
    if (!(__builtin_constant_p(octaves) && octaves >= 1 && octaves <= 8)) {
        s.sdf = gd_fbm2(px, py, octaves, persistence, lacunarity, s.lut);
        return;
    }
    // as gd_fbm2's table path (without a table every pixel takes the hash branch below: one loop body either way, so
    // that k stays in one set of registers); cell 0 (which samples at p * scale itself) takes its y half from the workgroup's row values
    GdCell cell[8];
    float amp[8];
    float max_val = 0.0f;
    {
        const float fx = gd_fract(px);
        cell[0].ix = gd_floor(px); cell[0].iy = y[0];
        cell[0].ux = fx * fx * __builtin_fmaf(-2.0f, fx, 3.0f); cell[0].uy = y[1];
    }
    float value = 0.0f;
    const bool in_range = gd_fbm_table_ok(px, py, octaves, persistence, lacunarity, amp, max_val);
    const bool table = s.lut != nullptr && in_range;
    // The kept edge values follow cell 0 wherever that cell is inside the table, whether or not this pixel's other octaves
    // are (the norm test implies it): a condition of its own keeps the refresh a plain masked update of k's registers.
    const bool refresh = (s.lut != nullptr) & (gd_abs(cell[0].ix) <= 511.0f) & (y[2] != 0.0f) & (cell[0].iy != k.iy);
    if (refresh) {
        cell[0].off = gd_lut_offset16(cell[0].ix, cell[0].iy, 0.0f);
        GdCorners c0 = gd_corners_load(cell[0], s.lut);
        k.lo = gd_edge_lo(c0, cell[0].ux); k.d = gd_edge_hi(c0, cell[0].ux) - k.lo; k.iy = cell[0].iy;
    }
    const float n0 = __builtin_fmaf(__builtin_fmaf(y[1], k.d, k.lo), 2.0f, -1.0f);   // octave 0, taken here so that k is not live across the branch
    if (table) {
        gd_fbm_cells(px, py, octaves, lacunarity, 1, cell, gd_fbm_folds(lacunarity));
        GdCorners c[8];
#pragma unroll
        for (int i = 1; i < octaves; i++) c[i] = gd_corners_load(cell[i], s.lut);
        value = __builtin_fmaf(n0, amp[0], value);
#pragma unroll
        for (int i = 1; i < octaves; i++) value = __builtin_fmaf(gd_noise2_combine(cell[i], c[i]), amp[i], value);
    } else {
        gd_fbm_cells(px, py, octaves, lacunarity, 1, cell, false);
#pragma unroll
        for (int i = 0; i < octaves; i++) value = __builtin_fmaf(gd_noise2_combine(cell[i], gd_corners_hash(cell[i])), amp[i], value);
    }
    s.sdf = value * gd_rcp(max_val);
}
GD void gb_curl_noise(GdPix& s, float frequency, float amplitude)                                       // :367-377
{
    const float eps = 0.001f;
    float bx = s.px * frequency, by = s.py * frequency;
    float n0, n1, n2, n3;
    gd_noise2_pair_lut(bx + 0.0f, by + eps, bx - 0.0f, by - eps, s.lut, n0, n1);
    gd_noise2_pair_lut(bx + eps, by + 0.0f, bx - eps, by - 0.0f, s.lut, n2, n3);
    float cx = (n0 - n1) / (2.0f * eps);
    float cy = -(n2 - n3) / (2.0f * eps);
    float len = gd_length2(cx, cy);
    s.px = s.px + cx / len * amplitude; s.py = s.py + cy / len * amplitude;
}
GD void gb_displace(GdPix& s, float strength)                                                           // :378-381
{
    float t5 = s.time * 0.5f;
    float n0, n1;
    gd_noise2_pair_lut(s.px * 3.0f + t5, s.py * 3.0f + t5, s.px * 3.0f + 5.2f + t5, s.py * 3.0f + 1.3f + t5, s.lut, n0, n1);
    s.px = s.px + n0 * strength; s.py = s.py + n1 * strength;
}

// SDF modifiers: Sdf -> Sdf
GD void gb_mask_arc(GdPix& s, float angle)                                                              // :384-389
{
    float theta = gd_atan2(s.px, s.py) + 3.14159265359f;
    s.sdf = (theta < angle) ? s.sdf : 999.0f;
}
GD void gb_threshold(GdPix& s, float cutoff) { s.sdf = (s.sdf < cutoff) ? -1.0f : 1.0f; }               // :390-393
GD void gb_onion(GdPix& s, float thickness) { s.sdf = gd_abs(s.sdf) - thickness; }                      // :394-397
GD void gb_round(GdPix& s, float radius) { s.sdf = s.sdf - radius; }                                    // :398-401

// Bridges: Sdf -> Color
GD void gb_glow(GdPix& s, float intensity) { float g = gd_apply_glow(s.guard, s.sdf, intensity); s.r = s.g = s.b = g; s.a = 1.0f; }  // :404-408
GD void gb_shade(GdPix& s, float r, float g, float b)                                                   // :409-416
{
    float m = gd_smoothstep(s.guard, 0.005f, -0.005f, s.sdf);
    s.r = r * m; s.g = g * m; s.b = b * m; s.a = 1.0f;
}
GD void gb_emissive(GdPix& s, float intensity) { float g = gd_apply_glow(s.guard, s.sdf, intensity); s.r = s.g = s.b = g; s.a = g; }  // :417-421
GD void gb_palette(GdPix& s, float ar, float ag, float ab, float br, float bg, float bb, float cr, float cg, float cb,
                   float dr, float dg, float db)                                                        // :422-437, :222-226
{
    float t = gd_clamp(s.sdf * 0.5f + 0.5f, 0.0f, 1.0f);
    s.r = ar + br * gd_cos(s.guard, 6.28318f * (cr * t + dr));
    s.g = ag + bg * gd_cos(s.guard, 6.28318f * (cg * t + dg));
    s.b = ab + bb * gd_cos(s.guard, 6.28318f * (cb * t + db));
    s.a = 1.0f;
}

// Colour processors: Color -> Color
GD void gb_tint(GdPix& s, float r, float g, float b) { s.r = s.r * r; s.g = s.g * g; s.b = s.b * b; s.a = 1.0f; }   // :440-445
GD void gb_bloom(GdPix& s, float threshold, float strength)                                             // :446-452
{
    float lum = s.r * 0.299f + s.g * 0.587f + s.b * 0.114f;
    float add = gd_max(lum - threshold, 0.0f) * strength;
    s.r = s.r + add; s.g = s.g + add; s.b = s.b + add; s.a = 1.0f;
}
GD void gb_grain(GdPix& s, float amount)                                                                // :453-457
{
    float n = gd_fract(gd_sin(s.guard, s.px * 12.9898f + s.py * 78.233f + s.time) * 43758.5453f);
    float add = (n - 0.5f) * amount;
    s.r = s.r + add; s.g = s.g + add; s.b = s.b + add;
}
GD void gb_vignette(GdPix& s, float strength, float radius)                                             // :458-464
{
    float dist = s.uvlen;
    float f = 1.0f - gd_smoothstep(s.guard, radius, radius + strength, dist);
    s.r = s.r * f; s.g = s.g * f; s.b = s.b * f;
}
GD void gb_chromatic(GdPix& s, float offset)                                                            // :465-472
{
    float dirx = s.uvx * s.uvinv * offset;   // normalize(uv).x * offset
    s.r = s.r * (1.0f + dirx);
    s.b = s.b * (1.0f - dirx);
}
GD void gb_tonemap(GdPix& s, float exposure)                                                            // :473-478
{
    float r = s.r * exposure, g = s.g * exposure, b = s.b * exposure;
    s.r = gd_reinhard(s.guard, r); s.g = gd_reinhard(s.guard, g); s.b = gd_reinhard(s.guard, b);
}
GD void gb_scanlines(GdPix& s, float frequency, float intensity)                                        // :479-484
{
    float v = gd_sin(s.guard, s.uvy * frequency * 3.14159265359f) * intensity;
    float f = 1.0f - v * 0.5f;
    s.r = s.r * f; s.g = s.g * f; s.b = s.b * f;
}
GD void gb_saturate_color(GdPix& s, float amount)                                                       // :485-489
{
    float lum = s.r * 0.299f + s.g * 0.587f + s.b * 0.114f;
    s.r = gd_mix(lum, s.r, amount); s.g = gd_mix(lum, s.g, amount); s.b = gd_mix(lum, s.b, amount);
}
GD void gb_glitch(GdPix& s, float intensity)                                                            // :490-501
{
    float gt = gd_floor(s.time * 10.0f);
    float n = gd_fract(gd_sin(s.guard, gt * 43758.5453f + s.py * 100.0f) * 43758.5453f);
    float shift = (n - 0.5f) * intensity * 0.1f;
    float block = gd_step(1.0f - intensity * 0.3f, gd_fract(gd_sin(s.guard, gt * 12.9898f + gd_floor(s.py * 20.0f)) * 43758.5453f));
    s.r = s.r + shift * block;
    s.b = s.b - shift * block;
}
GD void gb_blend(GdPix& s, float factor) { s.r = s.r * factor; s.g = s.g * factor; s.b = s.b * factor; }   // :502-505

// Full-screen generators: Position -> Color
GD void gb_gradient(GdPix& s) { float t = s.uvy * 0.5f + 0.5f; s.r = s.g = s.b = t; s.a = 1.0f; }       // :508-515
GD void gb_spectrum(GdPix& s, float bass, float mid, float treble)                                      // :516-525
{
    float sx = s.uvx * 0.5f + 0.5f;
    s.r = gd_smoothstep(s.guard, 0.0f, 0.33f, sx) * (s.bass + bass);
    s.g = gd_smoothstep(s.guard, 0.33f, 0.66f, sx) * (s.mid + mid);
    s.b = gd_smoothstep(s.guard, 0.66f, 1.0f, sx) * (s.treble + treble);
    s.a = 1.0f;
}

// ── WGSL-semantics variants (game-compiler/src/codegen/wgsl.rs:268-540) ─────────────────────────
// The reference's WebGPU emitter words sixteen builtins differently from its WebGL2 emitter (another
// star / polygon fold, gradient simplex, arithmetic-hash voronoi, ACES tonemap, cartesian repeat,
// radial spectrum, ...). A context created with GAME_HIP_SEMANTICS_WGSL calls these instead; every
// other builtin is shared. The same evaluation rules apply (one rounding per operator, the listed
// contraction points, reciprocal-multiply for quotients by a stage argument or constant).
GD float gw_rem(float x, float y) { return x - y * __builtin_truncf(x * gd_rcp(y)); }   // WGSL `%` on f32
GD float gw_step(float edge, float x) { return edge <= x ? 1.0f : 0.0f; }
GD float gw_fold(float angle, float seg) { return gw_rem(gw_rem(angle, seg) + seg, seg); }   // wgsl.rs:291
GD float gw_rem(GdGuard& g, float x, float y) { return x - y * __builtin_truncf(x * gd_rcp(g, y)); }
GD float gw_fold(GdGuard& g, float angle, float seg) { return gw_rem(g, gw_rem(g, angle, seg) + seg, seg); }
GD void gw_star(GdPix& s, float points, float radius, float inner)                                      // wgsl.rs:282-294
{
    float angle = gd_atan2(s.py, s.px);
    float r = gd_length2(s.guard, s.px, s.py);
    float seg = 6.28318f * gd_rcp(s.guard, points);
    float a = gd_abs(gw_fold(s.guard, angle, seg) - seg * 0.5f);
    float sa, ca;
    gd_sincos(s.guard, a, sa, ca);
    s.sdf = r * ca - radius + (r * sa - inner) * 0.5f;
}
GD void gw_polygon(GdPix& s, float sides, float radius)                                                 // wgsl.rs:301-312
{
    float a = gd_atan2(s.py, s.px);
    float r = gd_length2(s.guard, s.px, s.py);
    float seg = 6.28318f * gd_rcp(s.guard, sides);
    float ha = gd_abs(gw_fold(s.guard, a, seg) - seg * 0.5f);
    s.sdf = r * gd_cos(s.guard, ha) - radius;
}
// hash2v (wgsl.rs:206-209) of a lattice point: -1 + 2 * {the voronoi feature point of the WebGL2 path},
// so it reads the same table when the point is inside it.
GD void gw_hash2v(float x, float y, const float* __restrict__ vlut, float& hx, float& hy)
{
    float fx, fy;
    if (vlut != nullptr && gd_abs(x) <= 512.0f && gd_abs(y) <= 512.0f) {
        gd_f32x2 e = *(const gd_f32x2*)((const char*)vlut + gd_lut_offset(x, y, 0.0f));
        fx = e[0]; fy = e[1];
    } else {
        fx = gd_fract(gd_sin(x * 127.1f + y * 311.7f) * 43758.5453f);
        fy = gd_fract(gd_sin(x * 269.5f + y * 183.3f) * 43758.5453f);
    }
    hx = -1.0f + 2.0f * fx; hy = -1.0f + 2.0f * fy;
}
GD float gw_simplex2(float px, float py, const float* __restrict__ vlut)                               // wgsl.rs:211-222
{
    const float K1 = 0.366025404f, K2 = 0.211324865f;
    float sk = (px + py) * K1;
    float ix = gd_floor(px + sk), iy = gd_floor(py + sk);
    float t = (ix + iy) * K2;
    float ax = px - ix + t, ay = py - iy + t;
    float ox = gw_step(ay, ax), oy = gw_step(ax, ay);
    float bx = ax - ox + K2, by = ay - oy + K2;
    float cx = ax - 1.0f + 2.0f * K2, cy = ay - 1.0f + 2.0f * K2;
    float h0 = gd_max(0.5f - (ax * ax + ay * ay), 0.0f);
    float h1 = gd_max(0.5f - (bx * bx + by * by), 0.0f);
    float h2 = gd_max(0.5f - (cx * cx + cy * cy), 0.0f);
    float g0x, g0y, g1x, g1y, g2x, g2y;
    gw_hash2v(ix, iy, vlut, g0x, g0y);
    gw_hash2v(ix + ox, iy + oy, vlut, g1x, g1y);
    gw_hash2v(ix + 1.0f, iy + 1.0f, vlut, g2x, g2y);
    float n0 = h0 * h0 * h0 * h0 * (ax * g0x + ay * g0y);
    float n1 = h1 * h1 * h1 * h1 * (bx * g1x + by * g1y);
    float n2 = h2 * h2 * h2 * h2 * (cx * g2x + cy * g2y);
    return n0 * 70.0f + n1 * 70.0f + n2 * 70.0f;
}
GD void gw_simplex(GdPix& s, float scale) { s.sdf = gw_simplex2(s.px * scale, s.py * scale, s.vlut); }   // wgsl.rs:320-323
GD float gw_voronoi2(float px, float py, float scale, const float* __restrict__ lut)                    // wgsl.rs:187-201
{
    float spx = px * scale, spy = py * scale;
    float ipx = gd_floor(spx), ipy = gd_floor(spy);
    float fpx = gd_fract(spx), fpy = gd_fract(spy);
    const bool table = lut != nullptr && gd_abs(ipx) <= 510.0f && gd_abs(ipy) <= 510.0f;
    float h[3][3];
    if (table) {
        // the entries of rows ipy - 1 and ipy start with {h(y-1), h(y)} and {h(y), h(y+1)}, for columns ipx - 1 .. ipx + 1
        const char* base = (const char*)lut + gd_lut_offset16(ipx, ipy, -1.0f);
#pragma unroll
        for (int x = 0; x < 3; x++) {
            const gd_f32x2 lo = *(const gd_f32x2*)(base + 16 * x);
            const gd_f32x2 hi = *(const gd_f32x2*)(base + GD_LUT_STRIDE * 16u + 16 * x);
            h[0][x] = lo[0]; h[1][x] = lo[1]; h[2][x] = hi[1];
        }
    } else {
#pragma unroll
        for (int y = 0; y < 3; y++)
#pragma unroll
            for (int x = 0; x < 3; x++) h[y][x] = gd_hash2(ipx + (float)(x - 1), ipy + (float)(y - 1));
    }
    float d = 1.0f;
#pragma unroll
    for (int y = -1; y <= 1; y++) {
#pragma unroll
        for (int x = -1; x <= 1; x++) {
            float nx = (float)x, ny = (float)y;
            float pv = h[y + 1][x + 1];
            float ptx = nx + pv - fpx, pty = ny + gd_fract(pv * 17.0f) - fpy;
            d = gd_min(d, ptx * ptx + pty * pty);
        }
    }
    return gd_sqrt(d);
}
GD void gw_voronoi(GdPix& s, float scale) { s.sdf = gw_voronoi2(s.px, s.py, scale, s.lut); }            // wgsl.rs:324-327
GD void gw_concentric_waves(GdPix& s, float amplitude, float width, float frequency)                    // wgsl.rs:328-334
{
    float len = gd_length2(s.guard, s.px, s.py);
    float r = len * frequency;
    s.sdf = gd_sin(s.guard, r - s.time * 2.0f) * amplitude * gd_exp(-len * width);
}
GD void gw_threshold(GdPix& s, float cutoff) { s.sdf = gw_step(cutoff, s.sdf); }                        // wgsl.rs:501-504
GD void gw_vignette(GdPix& s, float strength, float radius)                                             // wgsl.rs:392-398
{
    float d = s.uvlen;
    float vig = gd_smoothstep(s.guard, radius + 0.3f, radius - 0.2f, d);
    float f = gd_mix(1.0f, vig, strength);
    s.r = s.r * f; s.g = s.g * f; s.b = s.b * f;
}
GD void gw_chromatic(GdPix& s, float offset)                                                            // wgsl.rs:399-408
{
    float d = s.uvlen * offset;
    s.r = s.r + d * 0.5f;
    s.b = s.b - d * 0.5f;
}
// The ACES quotient: numerator and divisor are functions of one value again, so the short division of gd_reinhard_core is
// checked the same way (game_hip_check_exhaustive(9), all 2^32 bit patterns of x against the compiler's division).
GD float gw_aces_ratio(GdGuard& g, float x)
{
    const float n = x * (2.51f * x + 0.03f), d = x * (2.43f * x + 0.59f) + 0.14f;
    const float a = gd_abs(d);
    // a numerator below 2^-96 other than +0 divides the long way: the correction's residual would be subnormal (and -0 / d is -0)
    if (gd_in_range(g, a >= 0x1p-126f && a < 0x1p126f && (gd_abs(n) >= 0x1p-96f || gd_bits(n) == 0u))) return gd_reinhard_core(n, d);
    return n / d;
}
GD float gw_aces_ratio(float x) { GdGuard g; g.speculate = false; g.left_range = false; return gw_aces_ratio(g, x); }
GD float gw_aces(GdGuard& g, float x) { return gd_clamp(gw_aces_ratio(g, x), 0.0f, 1.0f); }                 // wgsl.rs:409-419
GD void gw_tonemap(GdPix& s, float exposure)
{
    s.r = gw_aces(s.guard, s.r * exposure); s.g = gw_aces(s.guard, s.g * exposure); s.b = gw_aces(s.guard, s.b * exposure);
}
GD void gw_scanlines(GdPix& s, float frequency, float intensity)                                        // wgsl.rs:420-425
{
    float scan = gd_sin(s.guard, s.uvy * frequency * 3.14159f) * 0.5f + 0.5f;
    float f = 1.0f - intensity * (1.0f - scan);
    s.r = s.r * f; s.g = s.g * f; s.b = s.b * f;
}
GD void gw_glitch(GdPix& s, float intensity)                                                            // wgsl.rs:431-437
{
    float gt = gd_floor(s.time * 8.0f);
    float off = (gd_fract(gd_sin(s.guard, gt * 12.9898f) * 43758.5453f) - 0.5f) * intensity * 0.1f;
    float block = gw_step(0.9f - intensity * 0.3f, gd_fract(gd_sin(s.guard, s.uvy * 20.0f * 12.9898f + gt * 78.233f) * 43758.5453f));
    s.r = gd_mix(s.r, s.r + off, block);
    s.g = gd_mix(s.g, s.g - off * 0.5f, block);
    s.b = gd_mix(s.b, s.b, block);
}
GD void gw_mirror(GdPix& s, float axis)                                                                 // wgsl.rs:464-468
{
    if (axis > 0.5f) s.py = gd_abs(s.py); else s.px = gd_abs(s.px);
}
GD void gw_repeat(GdPix& s, float count)                                                                // wgsl.rs:469-475
{
    float rep = 2.0f * gd_rcp(s.guard, count);
    s.px = gw_fold(s.guard, s.px, rep) - rep * 0.5f;
    s.py = gw_fold(s.guard, s.py, rep) - rep * 0.5f;
}
GD void gw_curl_noise(GdPix& s, float frequency, float amplitude)                                       // wgsl.rs:481-488
{
    const float eps = 0.01f;
    float a, b, c, d;
    gd_noise2_pair_lut((s.px + eps) * frequency, (s.py + 0.0f) * frequency, (s.px - eps) * frequency, (s.py - 0.0f) * frequency, s.lut, a, b);
    gd_noise2_pair_lut((s.px + 0.0f) * frequency, (s.py + eps) * frequency, (s.px - 0.0f) * frequency, (s.py - eps) * frequency, s.lut, c, d);
    float cnx = a - b, cny = c - d;
    const float inv = gd_rcp(2.0f * eps);
    s.px = s.px + cny * amplitude * inv;
    s.py = s.py + -cnx * amplitude * inv;
}
GD void gw_displace(GdPix& s, float strength)                                                           // wgsl.rs:489-492
{
    float n0, n1;
    gd_noise2_pair_lut(s.px * 3.0f, s.py * 3.0f, s.px * 3.0f + 5.0f, s.py * 3.0f + 3.0f, s.lut, n0, n1);
    s.px = s.px + n0 * strength; s.py = s.py + n1 * strength;
}
GD void gw_spectrum(GdPix& s, float bass, float mid, float treble)                                      // wgsl.rs:520-534
{
    float d = gd_length2(s.guard, s.px, s.py);
    float sb = gd_smoothstep(s.guard, 0.35f, 0.15f, d) * bass;
    float sm = gd_smoothstep(s.guard, 0.55f, 0.35f, d) * gd_smoothstep(s.guard, 0.15f, 0.35f, d) * mid;
    float st = gd_smoothstep(s.guard, 0.75f, 0.55f, d) * gd_smoothstep(s.guard, 0.35f, 0.55f, d) * treble;
    s.r = sb * 1.0f + sm * 0.1f + st * 0.2f;
    s.g = sb * 0.2f + sm * 1.0f + st * 0.3f;
    s.b = sb * 0.1f + sm * 0.3f + st * 1.0f;
    s.a = 1.0f;
}

// RGBA8 texel -> float: the byte times the float nearest to 1 / 255 (255 maps to exactly 1).
GD float gd_unorm8_to_float(gd_u32 b) { return (float)b * 0x1.010102p-8f; }
// Memory feedback (game-compiler/src/codegen/memory.rs:38-45): mix with the previous frame's
// RGBA8 texel at the same pixel.
GD void gb_memory_mix(GdPix& s, gd_u32 prev, float decay)
{
    float pr = gd_unorm8_to_float(prev & 0xffu), pg = gd_unorm8_to_float((prev >> 8) & 0xffu);
    float pb = gd_unorm8_to_float((prev >> 16) & 0xffu), pa = gd_unorm8_to_float(prev >> 24);
    s.r = gd_mix(s.r, pr, decay); s.g = gd_mix(s.g, pg, decay);
    s.b = gd_mix(s.b, pb, decay); s.a = gd_mix(s.a, pa, decay);
}

// Framebuffer write: RGBA8 unorm, c * 255 rounded to the nearest integer (ties to even) and saturated to [0, 255],
// NaN -> 0, R in the low byte. v_cvt_pk_u8_f32 does the rounding, the saturation and the byte insert in one instruction
// (checked against this formula over all 2^32 inputs).
GD gd_u32 gd_pack_rgba8(float r, float g, float b, float a)
{
    if (__builtin_constant_p(a) && a == 1.0f) {   // alpha known to be 1 when the kernel is compiled: its byte is a constant
        gd_u32 q = __builtin_amdgcn_cvt_pk_u8_f32(r * 255.0f, 0u, 0xff000000u);
        q = __builtin_amdgcn_cvt_pk_u8_f32(g * 255.0f, 1u, q);
        return __builtin_amdgcn_cvt_pk_u8_f32(b * 255.0f, 2u, q);
    }
    gd_u32 q = __builtin_amdgcn_cvt_pk_u8_f32(r * 255.0f, 0u, 0u);
    q = __builtin_amdgcn_cvt_pk_u8_f32(g * 255.0f, 1u, q);
    q = __builtin_amdgcn_cvt_pk_u8_f32(b * 255.0f, 2u, q);
    return __builtin_amdgcn_cvt_pk_u8_f32(a * 255.0f, 3u, q);
}
