// Which rounding does v_cvt_pk_u8_f32 apply? Compares the instruction with three models over all 2^32 inputs:
//   0: truncate toward zero, saturate to [0, 255], NaN -> 0
//   1: round to nearest even, saturate, NaN -> 0
//   2: floor(x + 0.5) (as a float addition), saturate, NaN -> 0
// build: hipcc --offload-arch=gfx950 -O3 tools/cvt_pk_probe.hip -o tools/cvt_pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ unsigned model(int m, float x)
{
    if (x != x) return 0u;
    float y = m == 0 ? __builtin_truncf(x) : m == 1 ? __builtin_rintf(x) : __builtin_floorf(x + 0.5f);
    if (!(y > 0.0f)) return 0u;
    if (y >= 255.0f) return 255u;
    return (unsigned)y;
}

__global__ void probe(unsigned long long* bad, unsigned* first)
{
    unsigned long long total = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long b[3] = {0, 0, 0};
    for (unsigned long long v = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; v < (1ull << 32); v += total) {
        float x = __builtin_bit_cast(float, (unsigned)v);
        unsigned hw = __builtin_amdgcn_cvt_pk_u8_f32(x, 0u, 0u);
        unsigned hw2 = __builtin_amdgcn_cvt_pk_u8_f32(x, 2u, 0xffffffffu);   // byte 2 of an all-ones word
        for (int m = 0; m < 3; ++m) {
            unsigned w = model(m, x);
            if (hw != w || hw2 != ((0xffffffffu & ~0x00ff0000u) | (w << 16))) { ++b[m]; atomicMin(&first[m], (unsigned)v); }
        }
    }
    for (int m = 0; m < 3; ++m) if (b[m]) atomicAdd(&bad[m], b[m]);
}

int main()
{
    unsigned long long* bad; unsigned* first;
    hipMalloc(&bad, 24); hipMalloc(&first, 12);
    hipMemset(bad, 0, 24); hipMemset(first, 0xff, 12);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, bad, first);
    unsigned long long hb[3]; unsigned hf[3];
    hipMemcpy(hb, bad, 24, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 12, hipMemcpyDeviceToHost);
    const char* names[3] = {"truncate", "nearest-even", "floor(x+0.5)"};
    for (int m = 0; m < 3; ++m) {
        float f; unsigned u = hf[m]; __builtin_memcpy(&f, &u, 4);
        printf("model %-13s mismatches %llu first 0x%08x (%g)\n", names[m], hb[m], hf[m], hb[m] ? f : 0.0f);
    }
    return 0;
}
