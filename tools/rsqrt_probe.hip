// Exhaustive probe: which short v_rsq_f32 + fma sequences return the CORRECTLY ROUNDED 1 / sqrt(x) for every
// binary32 x of [2^-100, inf)? The reference value is decided exactly: the binary64 quotient 1 / sqrt(x) is within two
// binary64 ulps of the exact value, so the binary32 at or below it (a) and the next one (b) bracket the exact value, and
// the sign of the fused x * m^2 - 1 at the midpoint m = (a + b) / 2 (m * m is exact in binary64) picks the nearer one.
// Prints, per variant, the number of inputs it gets wrong and the first few of them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

#define NV 12
__device__ __forceinline__ float exact_residual(float x, float y0)   // 1 - x y0^2, exact to the last bit that matters
{
    float t = x * y0; float te = __builtin_fmaf(x, y0, -t);
    float e = __builtin_fmaf(-t, y0, 1.0f);
    return __builtin_fmaf(-te, y0, e);
}
__device__ __forceinline__ float variant(int which, float x)
{
    const float y0 = __builtin_amdgcn_rsqf(x);
    switch (which) {
    case 0: { float t = x * y0; float e = __builtin_fmaf(-t, y0, 1.0f); return __builtin_fmaf(0.5f * y0, e, y0); }   // plain Newton
    case 1: return __builtin_fmaf(0.5f * y0, exact_residual(x, y0), y0);                       // exact residual, first order
    case 2: return __builtin_fmaf(0x1.000002p-1f * y0, exact_residual(x, y0), y0);             // half seed biased up by 2^-23
    case 3: return __builtin_fmaf(0x1.000004p-1f * y0, exact_residual(x, y0), y0);             // ... 2^-22
    case 4: return __builtin_fmaf(0x1.000006p-1f * y0, exact_residual(x, y0), y0);             // ... 3 * 2^-23
    case 5: return __builtin_fmaf(0x1.000008p-1f * y0, exact_residual(x, y0), y0);             // ... 2^-21
    case 6: { float e = exact_residual(x, y0); float p = __builtin_fmaf(e, 0.375f, 0.5f); return __builtin_fmaf(y0 * p, e, y0); }   // second order
    case 7: { float e = exact_residual(x, y0); float q = e * __builtin_fmaf(e, 0.375f, 0.5f); return __builtin_fmaf(y0, q, y0); }
    case 8: {   // half folded into the residual
        float t = x * y0; float te = __builtin_fmaf(x, y0, -t); float h = 0.5f * y0;
        float e = __builtin_fmaf(-t, h, 0.5f); e = __builtin_fmaf(-te, h, e);
        return __builtin_fmaf(y0, e, y0);
    }
    case 9: {   // the same with the half biased
        float t = x * y0; float te = __builtin_fmaf(x, y0, -t); float h = 0x1.000002p-1f * y0;
        float e = __builtin_fmaf(-t, h, 0.5f); e = __builtin_fmaf(-te, h, e);
        return __builtin_fmaf(y0, e, y0);
    }
    case 10: {  // residual against a constant one ulp above 1 in the second step (te y0 subtracted from e + 2^-47)
        float t = x * y0; float te = __builtin_fmaf(x, y0, -t);
        float e = __builtin_fmaf(-t, y0, 1.0f); e = __builtin_fmaf(-te, y0, e) + 0x1p-46f;
        return __builtin_fmaf(0.5f * y0, e, y0);
    }
    default: {  // root correctly rounded, reciprocal of it (two roundings; the earlier definition, for reference)
        float g = x * y0, h = 0.5f * y0; float d = __builtin_fmaf(-g, g, x); g = __builtin_fmaf(d, h, g);
        float e = __builtin_fmaf(-g, y0, 1.0f); return __builtin_fmaf(e, y0, y0);
    }
    }
}

__device__ float rsqrt_arbitrated(float x)
{
    const double xd = (double)x;
    const double yd = 1.0 / __builtin_sqrt(xd);
    float a = (float)yd;
    if ((double)a > yd) a = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, a) - 1u);
    const float b = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, a) + 1u);
    const double m = 0.5 * ((double)a + (double)b);
    return __builtin_fma(xd, m * m, -1.0) > 0.0 ? a : b;
}

struct Out { unsigned long long bad[NV]; uint32_t first_bad[NV][16]; };

__global__ void probe(Out* o)
{
    unsigned long long total = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long v = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; v < (1ull << 31); v += total) {
        uint32_t u = (uint32_t)v;
        float x = __builtin_bit_cast(float, u);
        if (!(x >= 0x1p-100f && x < __builtin_inff())) continue;
        uint32_t wb = __builtin_bit_cast(uint32_t, rsqrt_arbitrated(x));
        for (int w = 0; w < NV; ++w) {
            float got = variant(w, x);
            if (__builtin_bit_cast(uint32_t, got) != wb) {
                unsigned long long k = atomicAdd(&o->bad[w], 1ull);
                if (k < 16) o->first_bad[w][k] = u;
            }
        }
    }
}

int main()
{
    Out* d;
    if (hipMalloc(&d, sizeof(Out)) != hipSuccess || hipMemset(d, 0, sizeof(Out)) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(8192), dim3(256), 0, 0, d);
    Out h;
    if (hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int w = 0; w < NV; ++w) {
        printf("variant %d: %llu mismatches", w, h.bad[w]);
        for (unsigned long long k = 0; k < h.bad[w] && k < 16; ++k) printf(" 0x%08x", h.first_bad[w][k]);
        printf("\n");
    }
    return 0;
}
