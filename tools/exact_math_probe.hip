// Exhaustive probes: do short v_rcp_f32 / v_rsq_f32 + fma sequences give the correctly rounded
// 1/x and sqrt(x) for every binary32 input of a range? Prints mismatch counts per exponent band.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float rcp_a(float x) { float y = __builtin_amdgcn_rcpf(x); float e = __builtin_fmaf(-x, y, 1.0f); return __builtin_fmaf(e, y, y); }
__device__ __forceinline__ float rcp_b(float x) { float y = rcp_a(x); float e = __builtin_fmaf(-x, y, 1.0f); return __builtin_fmaf(e, y, y); }
__device__ __forceinline__ float sqrt_a(float x)
{
    float r = __builtin_amdgcn_rsqf(x); float g = x * r, h = 0.5f * r;
    float e = __builtin_fmaf(-h, g, 0.5f); g = __builtin_fmaf(g, e, g); h = __builtin_fmaf(h, e, h);
    float d = __builtin_fmaf(-g, g, x); return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_c(float x)
{
    float r = __builtin_amdgcn_rsqf(x); float g = x * r, h = 0.5f * r;
    float d = __builtin_fmaf(-g, g, x); return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_d(float x)
{
    float g = __builtin_amdgcn_sqrtf(x); float h = 0.5f * __builtin_amdgcn_rcpf(g);
    float d = __builtin_fmaf(-g, g, x); return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_e(float x)   // two residual corrections
{
    float r = __builtin_amdgcn_rsqf(x); float g = x * r, h = 0.5f * r;
    float d = __builtin_fmaf(-g, g, x); g = __builtin_fmaf(d, h, g);
    d = __builtin_fmaf(-g, g, x); return __builtin_fmaf(d, h, g);
}

// reciprocal of the rounded root, refined from the v_rsq_f32 value the root was built from
__device__ __forceinline__ void root_and_inv(float x, int steps, float& g, float& y)
{
    float r = __builtin_amdgcn_rsqf(x); g = x * r; float h = 0.5f * r;
    float d = __builtin_fmaf(-g, g, x); g = __builtin_fmaf(d, h, g);
    y = r;
    for (int i = 0; i < steps; i++) { float e = __builtin_fmaf(-g, y, 1.0f); y = __builtin_fmaf(e, y, y); }
}

__global__ void probe(int which, unsigned long long* bad /* [256] per biased exponent */)
{
    unsigned long long total = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long v = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; v < (1ull << 32); v += total) {
        uint32_t u = (uint32_t)v;
        float x = __builtin_bit_cast(float, u);
        uint32_t ex = (u >> 23) & 0xff;
        if (ex == 0 || ex == 255) continue;
        float got, want;
        if (which >= 6) {
            if (u >> 31) continue;
            if (!(x >= 0x1p-100f)) continue;
            float g; root_and_inv(x, which - 5, g, got); want = 1.0f / g;
        } else if (which >= 2) { if (u >> 31) continue; want = __builtin_sqrtf(x); got = which == 2 ? sqrt_a(x) : which == 3 ? sqrt_c(x) : which == 4 ? sqrt_d(x) : sqrt_e(x); }
        else { want = 1.0f / x; got = which == 0 ? rcp_a(x) : rcp_b(x); }
        if (__builtin_bit_cast(uint32_t, got) != __builtin_bit_cast(uint32_t, want)) atomicAdd(&bad[ex], 1ull);
    }
}

int main()
{
    unsigned long long* d; hipMalloc(&d, 256 * 8);
    const char* names[] = {"rcp: rcp+1 newton", "rcp: rcp+2 newton", "sqrt: rsq, 1 coupled newton, residual", "sqrt: rsq, residual", "sqrt: v_sqrt + rcp, residual", "sqrt: rsq, 2 residuals", "1/root from rsq, 1 newton", "1/root from rsq, 2 newton", "1/root from rsq, 3 newton"};
    for (int w = 0; w < 9; w++) {
        hipMemset(d, 0, 256 * 8);
        hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, w, d);
        unsigned long long h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        unsigned long long tot = 0; int lo = -1, hi = -1;
        for (int e = 0; e < 256; e++) if (h[e]) { tot += h[e]; if (lo < 0) lo = e; hi = e; }
        printf("%-40s mismatches %llu, biased exponents with any: %d..%d;", names[w], tot, lo, hi);
        unsigned long long mid = 0; for (int e = 27; e <= 227; e++) mid += h[e];
        printf(" within 2^-100..2^100: %llu\n", mid);
        if (tot && tot < 100000) { printf("   per exponent:"); for (int e = 0; e < 256; e++) if (h[e]) printf(" %d:%llu", e, h[e]); printf("\n"); }
    }
    return 0;
}
