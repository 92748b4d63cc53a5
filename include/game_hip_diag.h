/* game_hip_diag.h - diagnostics of libgame_hip.so. NOT part of the render ABI (include/game_hip.h) and without a
 * counterpart in the reference: entry points that let the tests look inside the device code - element-wise evaluation
 * of the device math and noise functions, and exhaustive checks of the hardware instructions the render path leans on.
 * They live in the same shared library because they run the same ahead-of-time compiled device functions. */
#ifndef GAME_HIP_DIAG_H
#define GAME_HIP_DIAG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Element-wise evaluation of the device math / noise functions on host arrays, so that tests can
 * compare them bit-for-bit with a CPU restatement. op: 0 fract, 1 sin, 2 cos, 3 atan2(a,b),
 * 4 hash2(a,b), 5 noise2(a,b), 6 simplex2d(a,b), 7 voronoi2d(a,b,5), 8 exp2, 9 log2, 10 pow(a,b),
 * 11 max(a,b), 12 min(a,b), 13 fbm2(a,b,6,.5,2), 14 sqrt, 15 a/b, 16 smoothstep(.005,-.005,a),
 * 17 mod(a,b), 18 texel fetch: float of the byte (uint)a, 19 / 20 noise2 / fbm2 through the lattice-hash table,
 * 21 voronoi2d(a,b,5) through the feature-point table, 22 simplex2d(a,b) through the corner table, 23 the byte the
 * framebuffer write stores for the colour value a.
 * b may be NULL for one-argument ops. */
int game_hip_eval(int op, uint32_t n, const float* a, const float* b, float* out);
/* Exhaustive check, over all 2^32 float bit patterns, that the hardware instruction equals its
 * defining formula: which 0 = v_fract_f32 vs min(x - floor(x), 1 - 2^-24); 1 / 2 = v_max_f32 /
 * v_min_f32 vs "NaN yields the other operand, -0 < +0" against probe constants; 3 / 4 = the short
 * v_rsq_f32 / v_rcp_f32 + fma sequences of the render path vs correctly rounded sqrt(x) and 1/x; 5 = the joint root / reciprocal-of-root
 * sequence of the gravity kernel vs sqrtf(x) and 1/sqrtf(x) for 2^-100 <= x < inf; 6 = v_cvt_pk_u8_f32 (the frame store) vs
 * "to the nearest integer, ties to even, saturated to [0, 255], NaN -> 0", in byte 0 of a zero word and byte 2 of an all-ones word;
 * 7 = the reciprocal root alone against the exactly arbitrated 1/sqrt(x); 8 = the tonemap's short quotient x / (x + 1) vs the IEEE division;
 * 9 = the same for the WebGPU wording's ACES quotient (x (2.51 x + 0.03)) / (x (2.43 x + 0.59) + 0.14). */
int game_hip_check_exhaustive(int which, uint64_t* mismatches, uint32_t* first_bad);

#ifdef __cplusplus
}
#endif
#endif /* GAME_HIP_DIAG_H */
