// How fast can a read-modify-write pass over an 8K RGBA8 frame go on this GPU? Variants of a kernel that reads
// one frame and writes another (the memory-feedback traffic pattern: 4 B in + 4 B out per pixel), with a
// configurable amount of dependent FP32 work per pixel.  hipcc --offload-arch=gfx950 -O3 -o copy_microbench copy_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int WORK>
__device__ inline unsigned work(unsigned v, float k)
{
    float f = (float)(v & 0xffu) * (1.0f / 255.0f);
#pragma unroll
    for (int i = 0; i < WORK; ++i) f = __builtin_fmaf(f, k, 0.001f);
    return (v & 0xffffff00u) | ((unsigned)(f * 255.0f) & 0xffu);
}

// one 16-byte vector per lane, one element per thread of the grid
template <int WORK>
__global__ __launch_bounds__(256) void rmw_flat(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n, float k)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32x4 v = in[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = work<WORK>(v[j], k);
    out[i] = v;
}

// VEC vectors per lane, 256 vectors apart, all loads first
template <int WORK, int VEC>
__global__ __launch_bounds__(256) void rmw_multi(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n, float k)
{
    size_t base = (size_t)blockIdx.x * 256 * VEC + threadIdx.x;
    u32x4 v[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) if (base + q * 256 < n) v[q] = in[base + q * 256];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        if (base + q * 256 >= n) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[q][j] = work<WORK>(v[q][j], k);
        out[base + q * 256] = v[q];
    }
}

// persistent grid-stride with a one-deep software pipeline: the next vector is loaded before this one is processed
template <int WORK>
__global__ __launch_bounds__(256) void rmw_stride(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n, float k)
{
    size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32x4 cur = in[i];
    for (; i < n; i += stride) {
        u32x4 nxt = cur;
        if (i + stride < n) nxt = in[i + stride];
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = work<WORK>(cur[j], k);
        out[i] = cur;
        cur = nxt;
    }
}

// write-only: what the stateless cinematics do (4 B per pixel out, nothing in), WORK dependent fma per pixel
template <int WORK>
__global__ __launch_bounds__(256) void fill_flat(u32x4* __restrict__ out, size_t n, float k)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = work<WORK>((unsigned)(i * 4 + j), k);
    out[i] = v;
}

template <typename F>
static double time_us(F launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms * 100.0f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    const size_t px = (size_t)7680 * 4320, n = px / 4;
    u32x4 *a, *b;
    CK(hipMalloc(&a, px * 4)); CK(hipMalloc(&b, px * 4));
    CK(hipMemset(a, 0x3c, px * 4)); CK(hipMemset(b, 0, px * 4));
    const double bytes = 2.0 * px * 4;
    auto report = [&](const char* name, double us) { printf("%-34s %8.2f us  %6.2f TB/s  %8.0f Mpixel/s\n", name, us, bytes / us / 1e6, px / us); fflush(stdout); };
    const unsigned gflat = (unsigned)((n + 255) / 256);
    {
        double us = time_us([&] { CK(hipMemcpyAsync(b, a, px * 4, hipMemcpyDeviceToDevice, 0)); }, 9);
        report("hipMemcpy D2D", us);
    }
#define RUN_FLAT(W) report("flat 16B/lane work=" #W, time_us([&] { rmw_flat<W><<<gflat, 256>>>(a, b, n, 0.999f); std::swap(a, b); }, 9));
    RUN_FLAT(0) RUN_FLAT(16) RUN_FLAT(32) RUN_FLAT(64) RUN_FLAT(96)
#define RUN_MULTI(W, V) report("multi " #V "x16B/lane work=" #W, time_us([&] { rmw_multi<W, V><<<(unsigned)((n + 256 * V - 1) / (256 * V)), 256>>>(a, b, n, 0.999f); std::swap(a, b); }, 9));
    RUN_MULTI(0, 2) RUN_MULTI(0, 4) RUN_MULTI(64, 2) RUN_MULTI(64, 4)
#define RUN_STRIDE(W, G) report("stride grid=" #G " work=" #W, time_us([&] { rmw_stride<W><<<G, 256>>>(a, b, n, 0.999f); std::swap(a, b); }, 9));
    RUN_STRIDE(0, 2048) RUN_STRIDE(0, 4096) RUN_STRIDE(0, 8192) RUN_STRIDE(64, 2048) RUN_STRIDE(64, 4096) RUN_STRIDE(64, 8192)
    // write-only kernels over the same 8K frame (133 MB out): the ceiling of the stateless cinematics
    auto report_w = [&](const char* name, double us) { printf("%-34s %8.2f us  %6.2f TB/s written %8.0f Mpixel/s\n", name, us, px * 4.0 / us / 1e6, px / us); fflush(stdout); };
#define RUN_FILL(W) report_w("write-only 16B/lane work=" #W, time_us([&] { fill_flat<W><<<gflat, 256>>>(b, n, 0.999f); }, 9));
    RUN_FILL(0) RUN_FILL(16) RUN_FILL(32) RUN_FILL(48) RUN_FILL(64) RUN_FILL(96)
    // write-only and read-only ends of the spectrum
    report("hipMemset (write only, x2 bytes)", 2.0 * time_us([&] { CK(hipMemsetAsync(b, 1, px * 4, 0)); }, 9));
    return 0;
}
