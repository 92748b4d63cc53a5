// Diagnostics compiled ahead of time for gfx950: element-wise evaluation of the device math and
// noise functions on caller-supplied arrays (so tests can compare them bit-for-bit with a CPU
// restatement), and exhaustive self-checks of the hardware instructions the device code leans on
// (v_fract_f32, v_max_f32, v_min_f32) against their defining formulas over all 2^32 inputs.
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include "../front/capi_internal.hpp"
#include "game_device.h"

using namespace game;

namespace {

enum EvalOp { EV_FRACT = 0, EV_SIN, EV_COS, EV_ATAN2, EV_HASH2, EV_NOISE2, EV_SIMPLEX, EV_VORONOI, EV_EXP2, EV_LOG2, EV_POW,
              EV_MAX, EV_MIN, EV_FBM6, EV_SQRT, EV_DIV, EV_SMOOTHSTEP_GLOW, EV_MOD, EV_UNORM8, EV_NOISE2_LUT, EV_FBM6_LUT, EV_VORONOI_LUT, EV_SIMPLEX_LUT, EV_PACK_UNORM8, EV_RSQRT, EV_COUNT };

// Fills the lattice-hash table: entry (row iy + 512, column ix + 512) = the corners of the cell at (ix, iy) as
// gd_corners_hash makes them: {h(ix, iy), h(ix, iy + 1), h(ix + 1, iy) - h(ix, iy), h(ix + 1, iy + 1) - h(ix, iy + 1)}.
__global__ void hash_lut_fill(float* __restrict__ table)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
    if (u >= GD_LUT_ROWS) return;
    GdCell c;
    c.ix = (float)((int)u - GD_LUT_HALF); c.iy = (float)((int)v - GD_LUT_HALF);
    const GdCorners k = gd_corners_hash(c);
    float* e = table + ((size_t)v * GD_LUT_STRIDE + u) * 4u;
    e[0] = k.a; e[1] = k.c; e[2] = k.ab; e[3] = k.cd;
}

// Fills the voronoi table: entry (row cy + 512, column cx + 512) = the cell's feature point (glsl.rs:211-214).
__global__ void voronoi_lut_fill(float* __restrict__ table)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
    if (u >= GD_LUT_ROWS) return;
    const float cx = (float)((int)u - GD_LUT_HALF), cy = (float)((int)v - GD_LUT_HALF);
    float* e = table + ((size_t)v * GD_LUT_STRIDE + u) * 2u;
    e[0] = gd_fract(gd_sin(cx * 127.1f + cy * 311.7f) * 43758.5453f);
    e[1] = gd_fract(gd_sin(cx * 269.5f + cy * 183.3f) * 43758.5453f);
}

// Fills the simplex corner table: entry (row Y, column X), X, Y in [0, 289] = gd_simplex_corner(permute(permute(Y) + X)).
__global__ void simplex_lut_fill(float* __restrict__ table)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
    if (u >= GD_SIMPLEX_DIM) return;
    const GdSimplexCorner c = gd_simplex_corner(gd_permute(gd_permute((float)v) + (float)u));
    float* e = table + ((size_t)v * GD_SIMPLEX_DIM + u) * 4u;
    e[0] = c.a0; e[1] = c.h; e[2] = c.nrm; e[3] = 0.0f;
}

__global__ void eval_kernel(int op, uint32_t n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                            const float* __restrict__ lut)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f;
    switch (op) {
    case EV_FRACT: r = gd_fract(x); break;
    case EV_SIN: r = gd_sin(x); break;
    case EV_COS: r = gd_cos(x); break;
    case EV_ATAN2: r = gd_atan2(x, y); break;
    case EV_HASH2: r = gd_hash2(x, y); break;
    case EV_NOISE2: r = gd_noise2(x, y); break;
    case EV_SIMPLEX: r = gd_simplex2d(x, y); break;
    case EV_VORONOI: r = gd_voronoi2d(x, y, 5.0f); break;
    case EV_EXP2: r = gd_exp2(x); break;
    case EV_LOG2: r = gd_log2(x); break;
    case EV_POW: r = gd_pow(x, y); break;
    case EV_MAX: r = gd_max(x, y); break;
    case EV_MIN: r = gd_min(x, y); break;
    case EV_FBM6: r = gd_fbm2(x, y, 6, 0.5f, 2.0f); break;
    case EV_SQRT: r = gd_sqrt(x); break;
    case EV_DIV: r = x / y; break;
    case EV_SMOOTHSTEP_GLOW: r = gd_smoothstep(0.005f, -0.005f, x); break;
    case EV_MOD: r = gd_mod(x, y); break;
    case EV_UNORM8: r = gd_unorm8_to_float((gd_u32)x); break;
    case EV_NOISE2_LUT: r = gd_noise2_lut(x, y, lut); break;
    case EV_FBM6_LUT: r = gd_fbm2(x, y, 6, 0.5f, 2.0f, lut); break;
    case EV_VORONOI_LUT: r = gd_voronoi2d_lut(x, y, 5.0f, lut); break;
    case EV_SIMPLEX_LUT: r = gd_simplex2d(x, y, lut); break;
    case EV_PACK_UNORM8: r = (float)(gd_pack_rgba8(x, 0.0f, 0.0f, 0.0f) & 0xffu); break;
    case EV_RSQRT: r = (x >= 0x1p-100f && x < __builtin_inff()) ? gd_rsqrt_core(x) : 1.0f / __builtin_sqrtf(x); break;
    default: break;
    }
    out[i] = r;
}

// inversesqrt as the render path defines it for 2^-100 <= x < inf, decided in exact arithmetic: the exact 1 / sqrt(x)
// rounded to nearest, except that the float two below a power of four, 4^k (1 - 2^-23), gives 2^-k (see gd_rsqrt_core).
// The binary64 quotient is within two binary64 ulps of the exact value, so the binary32 at or below it (a) and the next
// one (b) bracket the exact value; with m = (a + b) / 2 (25 significant bits: m * m is exact in binary64) the sign of
// the fused x * m^2 - 1 says on which side of the midpoint the exact value lies. A tie cannot occur.
__device__ float rsqrt_arbitrated(float x)
{
    if ((gd_bits(x) & 0x00ffffffu) == 0x007ffffeu)   // 4^k (1 - 2^-23), biased exponent E = 2 k + 126: the result is 2^-k
        return gd_from_bits((127u - ((gd_bits(x) >> 23) - 126u) / 2u) << 23);
    const double xd = (double)x;
    const double yd = 1.0 / __builtin_sqrt(xd);
    float a = (float)yd;
    if ((double)a > yd) a = gd_from_bits(gd_bits(a) - 1u);
    const float b = gd_from_bits(gd_bits(a) + 1u);
    const double m = 0.5 * ((double)a + (double)b);
    return __builtin_fma(xd, m * m, -1.0) > 0.0 ? a : b;
}

// which: 0 = fract(x) for every bit pattern x; 1 = max(x, c), 2 = min(x, c) for every x and the
// probe constants below; 3 = gd_sqrt(x), 4 = gd_rcp(x) against the compiler's IEEE expansions; 5 = gd_sqrt_rsqrt_core
// against sqrtf(x) and the exactly arbitrated 1 / sqrt(x) for 2^-100 <= x < inf; 6 = v_cvt_pk_u8_f32 against the frame
// store's definition; 7 = gd_rsqrt_core alone against the same arbitrated value; 8 = gd_reinhard(x) against x / (x + 1);
// 9 = gw_aces_ratio(x) against the same quotient divided by the compiler.
__global__ void exhaustive_kernel(int which, unsigned long long* mismatches, uint32_t* first_bad)
{
    const float probes[6] = {0.0f, -0.0f, 1.0f, -1.0f, __builtin_inff(), __builtin_nanf("")};
    unsigned long long total = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (unsigned long long v = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; v < (1ull << 32); v += total) {
        float x = gd_from_bits((gd_u32)v);
        if (which == 0) {
            float hw = gd_fract(x);
            float f = x - gd_floor(x);
            float want = (f != f) ? f : (f < 1.0f ? f : 0x1.fffffep-1f);
            bool same = (hw != hw && want != want) || gd_bits(hw) == gd_bits(want);
            if (!same) { ++bad; atomicMin(first_bad, (gd_u32)v); }
        } else if (which == 3 || which == 4) {
            float hw = which == 3 ? gd_sqrt(x) : gd_rcp(x);
            float want = which == 3 ? __builtin_sqrtf(x) : 1.0f / x;
            bool same = (hw != hw && want != want) || gd_bits(hw) == gd_bits(want);
            if (!same) { ++bad; atomicMin(first_bad, (gd_u32)v); }
        } else if (which == 8 || which == 9) {
            float hw = which == 8 ? gd_reinhard(x) : gw_aces_ratio(x);
            float want = which == 8 ? x / (x + 1.0f) : (x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f);
            bool same = (hw != hw && want != want) || gd_bits(hw) == gd_bits(want);
            if (!same) { ++bad; atomicMin(first_bad, (gd_u32)v); }
        } else if (which == 6) {
            float t = __builtin_rintf(x);
            gd_u32 want = (x != x || !(t > 0.0f)) ? 0u : (t >= 255.0f ? 255u : (gd_u32)t);
            gd_u32 lo = __builtin_amdgcn_cvt_pk_u8_f32(x, 0u, 0u), mid = __builtin_amdgcn_cvt_pk_u8_f32(x, 2u, 0xffffffffu);
            if (lo != want || mid != (0xff00ffffu | (want << 16))) { ++bad; atomicMin(first_bad, (gd_u32)v); }
        } else if (which == 5 || which == 7) {
            if (!(x >= 0x1p-100f && x < __builtin_inff())) continue;
            const float want_inv = rsqrt_arbitrated(x);
            if (which == 5) {
                float root, inv;
                gd_sqrt_rsqrt_core(x, root, inv);
                if (gd_bits(root) != gd_bits(__builtin_sqrtf(x)) || gd_bits(inv) != gd_bits(want_inv)) { ++bad; atomicMin(first_bad, (gd_u32)v); }
            } else if (gd_bits(gd_rsqrt_core(x)) != gd_bits(want_inv)) { ++bad; atomicMin(first_bad, (gd_u32)v); }
        } else {
            for (int p = 0; p < 6; ++p) {
                float c = probes[p];
                float hw = which == 1 ? gd_max(x, c) : gd_min(x, c);
                float want;
                if (x != x) want = c;
                else if (c != c) want = x;
                else if (x == c) {
                    bool xneg = gd_bits(x) >> 31;
                    want = which == 1 ? (xneg ? c : x) : (xneg ? x : c);
                } else want = which == 1 ? (x > c ? x : c) : (x < c ? x : c);
                bool same = (hw != hw && want != want) || gd_bits(hw) == gd_bits(want);
                if (!same) { ++bad; atomicMin(first_bad, (gd_u32)v); }
            }
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

int hip_err(const char* what, hipError_t e) { return set_error(GAME_E_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return hip_err(#expr, _e); } while (0)

// Device allocation released on every return path.
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
};

}  // namespace

namespace game {

static std::mutex g_tables_mu;
static std::map<int, float*> g_tables;   // key = device * 4 + kind

int release_device_tables()
{
    std::lock_guard<std::mutex> lock(g_tables_mu);
    int prev = 0;
    hipError_t e = hipGetDevice(&prev);
    for (auto& kv : g_tables) {
        if (e == hipSuccess) e = hipSetDevice(kv.first / 4);
        if (e == hipSuccess) e = hipFree(kv.second);
    }
    g_tables.clear();
    if (e == hipSuccess) e = hipSetDevice(prev);
    if (e != hipSuccess) return hip_err("game_hip_shutdown", e);
    return GAME_OK;
}

int lut_for_current_device(int kind, const float** table)
{
    std::mutex& mu = g_tables_mu;
    std::map<int, float*>& tables = g_tables;
    if (kind < 0 || kind > 2) return set_error(GAME_E_ARGUMENT, "lut_for_current_device: unknown table");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    dev = dev * 4 + kind;
    std::lock_guard<std::mutex> lock(mu);
    auto it = tables.find(dev);
    if (it == tables.end()) {
        float* t = nullptr;
        // lattice hashes: 16-byte entries; voronoi feature points: 8-byte entries
        const size_t bytes = kind == 2 ? (size_t)GD_SIMPLEX_DIM * GD_SIMPLEX_DIM * 4u * sizeof(float)
                                       : (size_t)GD_LUT_ROWS * GD_LUT_STRIDE * (kind == 0 ? 4u : 2u) * sizeof(float);
        HIP_TRY(hipMalloc((void**)&t, bytes));
        hipError_t e = hipMemset(t, 0, bytes);
        if (e == hipSuccess) {
            if (kind == 2) hipLaunchKernelGGL(simplex_lut_fill, dim3((GD_SIMPLEX_DIM + 255u) / 256u, GD_SIMPLEX_DIM), dim3(256), 0, 0, t);
            else hipLaunchKernelGGL(kind ? voronoi_lut_fill : hash_lut_fill, dim3((GD_LUT_ROWS + 255u) / 256u, GD_LUT_ROWS), dim3(256), 0, 0, t);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { (void)hipFree(t); return hip_err("lut fill", e); }
        it = tables.emplace(dev, t).first;
    }
    *table = it->second;
    return GAME_OK;
}

}  // namespace game

extern "C" {

int game_hip_shutdown(void)
{
    // live contexts hold raw pointers into the tables: freeing them now would leave those contexts reading freed memory
    const int live = live_context_count();
    if (live > 0)
        return set_error(GAME_E_ARGUMENT, "game_hip_shutdown: " + std::to_string(live) + " render context(s) still exist; destroy them first");
    return release_device_tables();
}

int game_hip_eval(int op, uint32_t n, const float* a, const float* b, float* out)
{
    if (op < 0 || op >= EV_COUNT || !a || !out || n == 0) return set_error(GAME_E_ARGUMENT, "game_hip_eval: bad argument");
    DevBuf da, db, dout;
    size_t bytes = (size_t)n * 4;
    HIP_TRY(da.alloc(bytes));
    HIP_TRY(dout.alloc(bytes));
    HIP_TRY(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
    if (b) { HIP_TRY(db.alloc(bytes)); HIP_TRY(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice)); }
    const float* lut = nullptr;
    if (op == EV_NOISE2_LUT || op == EV_FBM6_LUT || op == EV_VORONOI_LUT || op == EV_SIMPLEX_LUT) {
        int rc = lut_for_current_device(op == EV_VORONOI_LUT ? 1 : op == EV_SIMPLEX_LUT ? 2 : 0, &lut);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(eval_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, n, (const float*)da.p, (const float*)db.p, (float*)dout.p, lut);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return GAME_OK;
}

int game_hip_check_exhaustive(int which, uint64_t* mismatches, uint32_t* first_bad)
{
    if (which < 0 || which > 9 || !mismatches || !first_bad) return set_error(GAME_E_ARGUMENT, "game_hip_check_exhaustive: bad argument");
    DevBuf dm, df;
    HIP_TRY(dm.alloc(8));
    HIP_TRY(df.alloc(4));
    HIP_TRY(hipMemset(dm.p, 0, 8));
    HIP_TRY(hipMemset(df.p, 0xff, 4));
    hipLaunchKernelGGL(exhaustive_kernel, dim3(4096), dim3(256), 0, 0, which, (unsigned long long*)dm.p, (uint32_t*)df.p);
    HIP_TRY(hipGetLastError());
    unsigned long long m = 0;
    HIP_TRY(hipMemcpy(&m, dm.p, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(first_bad, df.p, 4, hipMemcpyDeviceToHost));
    *mismatches = m;
    return GAME_OK;
}

}  // extern "C"
