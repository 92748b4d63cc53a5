// Separable binomial blur of an RGBA8 frame, one kernel, both passes in LDS.
//
// The reference has no neighbourhood filter (all of its post passes are pointwise); this is the frame filter that
// gives band-sharded rendering a halo. Definition, in integers, so that every machine agrees: weights are the binomial
// row C(2r, k) (sum 4^r); the horizontal pass replaces every channel by (sum of weight * byte + 4^r / 2) >> 2r with
// x clamped to the frame, the vertical pass does the same over y with the horizontal results (themselves bytes).
//
// A workgroup of 256 lanes owns a 64 x 32 pixel tile: it loads the tile plus r pixels all round into LDS (one coalesced
// RGBA8 word per lane per step), runs the horizontal pass over 32 + 2r rows into a second LDS array, then the vertical
// pass, and stores one word per lane per step. Both passes handle two channels per 32-bit operation: R and B sit in
// the two 16-bit halves of (p & 0x00ff00ff), G and A in those of ((p >> 8) & 0x00ff00ff), and since the weights sum to
// at most 256 a 16-bit half never overflows (255 * 256 + 128 < 65536). HBM traffic is one read and one write of the
// frame (halo pixels are shared between neighbouring tiles through L2).
#include <hip/hip_runtime.h>
#include "../front/capi_internal.hpp"
#include "blur.hpp"

namespace {

constexpr uint32_t TX = 64, TY = 32;

template <int R>
struct Binomial;
template <> struct Binomial<1> { static constexpr uint32_t w[3] = {1, 2, 1}; };
template <> struct Binomial<2> { static constexpr uint32_t w[5] = {1, 4, 6, 4, 1}; };
template <> struct Binomial<3> { static constexpr uint32_t w[7] = {1, 6, 15, 20, 15, 6, 1}; };
template <> struct Binomial<4> { static constexpr uint32_t w[9] = {1, 8, 28, 56, 70, 56, 28, 8, 1}; };

template <int R>
__device__ __forceinline__ uint32_t finish(uint32_t rb, uint32_t ga)
{
    constexpr uint32_t half = (1u << (2 * R - 1)) * 0x00010001u;
    rb = ((rb + half) >> (2 * R)) & 0x00ff00ffu;
    ga = ((ga + half) >> (2 * R)) & 0x00ff00ffu;
    return rb | (ga << 8);
}

template <int R>
__global__ __launch_bounds__(256) void blur_kernel(const uint32_t* __restrict__ raw, uint32_t* __restrict__ out, uint32_t w, uint32_t h,
                                                   uint32_t row0, uint32_t rows)
{
    constexpr uint32_t IW = TX + 2 * R, IH = TY + 2 * R;
    __shared__ uint32_t in[IH][IW];
    __shared__ uint32_t mid[IH][TX];
    const uint32_t tid = threadIdx.x;
    const int x0 = (int)(blockIdx.x * TX), y0 = (int)(row0 + blockIdx.y * TY);
    for (uint32_t i = tid; i < IW * IH; i += 256u) {
        const uint32_t ly = i / IW, lx = i - ly * IW;
        int gx = x0 - R + (int)lx, gy = y0 - R + (int)ly;
        gx = gx < 0 ? 0 : (gx >= (int)w ? (int)w - 1 : gx);     // clamp to the frame's edge
        gy = gy < 0 ? 0 : (gy >= (int)h ? (int)h - 1 : gy);
        in[ly][lx] = raw[(size_t)gy * w + (uint32_t)gx];
    }
    __syncthreads();
    for (uint32_t i = tid; i < TX * IH; i += 256u) {
        const uint32_t y = i / TX, x = i % TX;
        uint32_t rb = 0, ga = 0;
#pragma unroll
        for (int k = 0; k <= 2 * R; ++k) {
            const uint32_t p = in[y][x + (uint32_t)k];
            rb += Binomial<R>::w[k] * (p & 0x00ff00ffu);
            ga += Binomial<R>::w[k] * ((p >> 8) & 0x00ff00ffu);
        }
        mid[y][x] = finish<R>(rb, ga);
    }
    __syncthreads();
    const uint32_t x = tid % TX;
    const uint32_t gx = (uint32_t)x0 + x;
#pragma unroll
    for (uint32_t j = 0; j < TY; j += 256u / TX) {
        const uint32_t y = j + tid / TX;
        uint32_t rb = 0, ga = 0;
#pragma unroll
        for (int k = 0; k <= 2 * R; ++k) {
            const uint32_t p = mid[y + (uint32_t)k][x];
            rb += Binomial<R>::w[k] * (p & 0x00ff00ffu);
            ga += Binomial<R>::w[k] * ((p >> 8) & 0x00ff00ffu);
        }
        const uint32_t gy = (uint32_t)y0 + y;
        if (gx < w && gy < row0 + rows) out[(size_t)gy * w + gx] = finish<R>(rb, ga);
    }
}

}  // namespace

namespace game {

int launch_blur(const uint32_t* raw, uint32_t* out, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows, uint32_t radius, hipStream_t stream)
{
    if (!raw || !out || radius < 1 || radius > 4 || rows == 0 || (uint64_t)row0 + rows > h)
        return set_error(GAME_E_ARGUMENT, "blur: radius must be 1..4 and the band inside the frame");
    const dim3 grid((w + TX - 1) / TX, (rows + TY - 1) / TY), block(256);
    switch (radius) {
    case 1: hipLaunchKernelGGL(blur_kernel<1>, grid, block, 0, stream, raw, out, w, h, row0, rows); break;
    case 2: hipLaunchKernelGGL(blur_kernel<2>, grid, block, 0, stream, raw, out, w, h, row0, rows); break;
    case 3: hipLaunchKernelGGL(blur_kernel<3>, grid, block, 0, stream, raw, out, w, h, row0, rows); break;
    default: hipLaunchKernelGGL(blur_kernel<4>, grid, block, 0, stream, raw, out, w, h, row0, rows); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GAME_E_HIP, std::string("blur launch: ") + hipGetErrorString(e));
    return GAME_OK;
}

}  // namespace game
