// Separable binomial blur of an RGBA8 frame, one kernel, both passes in LDS.
//
// The reference has no neighbourhood filter (all of its post passes are pointwise); this is the frame filter that
// gives band-sharded rendering a halo. Definition, in integers, so that every machine agrees: weights are the binomial
// row C(2r, k) (sum 4^r); the horizontal pass replaces every channel by (sum of weight * byte + 4^r / 2) >> 2r with
// x clamped to the frame, the vertical pass does the same over y with the horizontal results (themselves bytes).
//
// A workgroup of 256 lanes owns a 64 x 64 pixel tile: it loads the tile plus four pixels left and right and r rows above
// and below into LDS as 16-byte words (all of a lane's loads in flight together), runs the horizontal pass over
// 64 + 2r rows in place, then the vertical pass, in which every lane makes a 4 x 4 block and stores it as
// four 16-byte words. In both passes four adjacent outputs come from one run of 4 + 2r samples, so every sample is read
// from LDS and split into channels once per four outputs. Frames whose width is not a multiple of 4 take the same
// kernel with 4-byte global accesses. Both passes handle two channels per 32-bit operation: R and B sit in
// the two 16-bit halves of (p & 0x00ff00ff), G and A in those of ((p >> 8) & 0x00ff00ff), worked on with the packed
// 16-bit instructions, and since the weights sum to at most 256 a 16-bit half never overflows (255 * 256 + 128 < 65536).
// HBM traffic is one read and one write of the frame (halo pixels are shared between neighbouring tiles through L2).
#include <hip/hip_runtime.h>
#include "../front/capi_internal.hpp"
#include "blur.hpp"

namespace {

constexpr uint32_t TX = 64, TY = 64;
constexpr int VB = (int)(TY / 16u);   // rows of the block a lane makes in the vertical pass

template <int R>
struct Binomial;
template <> struct Binomial<1> { static constexpr uint32_t w[3] = {1, 2, 1}; };
template <> struct Binomial<2> { static constexpr uint32_t w[5] = {1, 4, 6, 4, 1}; };
template <> struct Binomial<3> { static constexpr uint32_t w[7] = {1, 6, 15, 20, 15, 6, 1}; };
template <> struct Binomial<4> { static constexpr uint32_t w[9] = {1, 8, 28, 56, 70, 56, 28, 8, 1}; };

// Two channels per 32-bit register: R and B of a pixel in the 16-bit halves of (p & 0x00ff00ff), G and A in those of
// ((p >> 8) & 0x00ff00ff). A weighted sum is at most 255 * 256 + 128 < 65536, so nothing carries from one half into the
// other. Every sample is multiplied on its own: it then has at most 24 significant bits and weight * sample + sum is one
// full-rate v_mad_u32_u24. (Adding the two samples at distance d first halves the multiplies but makes 25-bit operands,
// which need the quarter-rate v_mul_lo_u32; the packed 16-bit instructions measured slower than either.)
// weight * sample + sum as the one instruction it should be. Written in C, the compiler factors the equal weights of
// the two samples at distance d back into (a + b) * w, whose 25-bit operand needs the quarter-rate multiply.
__device__ __forceinline__ uint32_t mad24(uint32_t w, uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "s"(w), "v"(x), "v"(acc));
    return r;
}

template <int R>
__device__ __forceinline__ uint32_t finish(uint32_t rb, uint32_t ga)
{
    rb = (rb >> (2 * R)) & 0x00ff00ffu;
    ga = (ga >> (2 * R)) & 0x00ff00ffu;
    return rb | (ga << 8);
}

// One pass over a run of consecutive samples p[OFF .. OFF + N + 2R): N adjacent outputs. Every sample is split into
// its two channel pairs once; the rounding constant 4^R / 2 (in both halves) is where each sum starts.
template <int R, int OFF, int N = 4>
__device__ __forceinline__ void four_outputs(const uint32_t (&p)[12], uint32_t (&o)[N])
{
    uint32_t rb[N + 2 * R], ga[N + 2 * R];
#pragma unroll
    for (int i = 0; i < N + 2 * R; ++i) { rb[i] = p[OFF + i] & 0x00ff00ffu; ga[i] = (p[OFF + i] >> 8) & 0x00ff00ffu; }
    constexpr uint32_t half = (1u << (2 * R - 1)) * 0x00010001u;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        uint32_t arb = half, aga = half;
#pragma unroll
        for (int k = 0; k <= 2 * R; ++k) {
            const uint32_t wk = Binomial<R>::w[k];   // a constant once the loop is unrolled
            if ((wk & (wk - 1u)) == 0u) { arb += wk * rb[j + k]; aga += wk * ga[j + k]; }   // 1, 2, 4, 8: adds and shift-adds
            else { arb = mad24(wk, rb[j + k], arb); aga = mad24(wk, ga[j + k], aga); }
        }
        o[j] = finish<R>(arb, aga);
    }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// VEC: the frame's rows are 16-byte aligned (width a multiple of 4), so the tile moves as 16-byte words.
template <int R, bool VEC>
__global__ __launch_bounds__(256) void blur_kernel(const uint32_t* __restrict__ raw, uint32_t* __restrict__ out, uint32_t w, uint32_t h,
                                                   uint32_t row0, uint32_t rows)
{
    constexpr uint32_t IH = TY + 2 * R;
    constexpr uint32_t CH = TX / 4u + 2u;                 // 16-byte chunks per tile row: the tile and four pixels either side
    constexpr uint32_t IWP = CH * 4u;
    // One array serves both passes: the horizontal pass writes its four outputs over the words it was centred on. That is
    // safe because the 16 lanes that share a row sit in one wave - every lane's reads of the row are issued, in lockstep,
    // before any lane's write - and no other wave touches that row before the barrier. Half the LDS of two arrays: seven
    // workgroups per CU instead of four, so that one workgroup's loads and stores overlap another's arithmetic.
    __shared__ __attribute__((aligned(16))) uint32_t in[IH][IWP];   // column 0 = frame column x0 - 4
    const uint32_t tid = threadIdx.x;
    // Workgroups are handed to the eight XCDs round robin, and each XCD has its own L2. Neighbouring tiles share their
    // halo pixels (and the 128-byte lines their edges sit in), so the launch is one-dimensional and workgroup L takes
    // tile (L mod 8) * chunk + L / 8: every XCD works through one contiguous eighth of the tile grid, in order, and
    // finds its neighbours' lines in its own L2 (measured: HBM reads 2.0x the frame without this, see the profile).
    const uint32_t tiles_x = (w + TX - 1u) / TX, tiles = tiles_x * ((rows + TY - 1u) / TY);
    const uint32_t chunk = (tiles + 7u) / 8u;
    const uint32_t tile = (blockIdx.x % 8u) * chunk + blockIdx.x / 8u;
    if (tile >= tiles) return;                            // the whole workgroup leaves, before any barrier
    const int x0 = (int)((tile % tiles_x) * TX), y0 = (int)(row0 + (tile / tiles_x) * TY);
    // all loads of the lane are issued before the first one is waited for (a fixed, unrolled count)
    constexpr uint32_t NLOAD = (CH * IH + 255u) / 256u;
    u32x4 held[NLOAD];
#pragma unroll
    for (uint32_t k = 0; k < NLOAD; ++k) {
        const uint32_t i = tid + 256u * k;
        const uint32_t ly = i / CH, c = i - ly * CH;
        int gy = y0 - R + (int)ly;
        gy = gy < 0 ? 0 : (gy >= (int)h ? (int)h - 1 : gy);       // clamp to the frame's edge (also keeps idle lanes in range)
        const int gx = x0 - 4 + 4 * (int)c;
        const uint32_t* row = raw + (size_t)(uint32_t)gy * w;
        if (VEC && gx >= 0 && gx + 3 < (int)w) held[k] = *(const u32x4*)(row + gx);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int x = gx + e;
                x = x < 0 ? 0 : (x >= (int)w ? (int)w - 1 : x);
                held[k][e] = row[x];
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < NLOAD; ++k) {
        const uint32_t i = tid + 256u * k;
        const uint32_t ly = i / CH, c = i - ly * CH;
        if (i < CH * IH) *(u32x4*)&in[ly][4u * c] = held[k];
    }
    __syncthreads();
    // horizontal pass: a lane makes four adjacent pixels of one row from twelve consecutive words (three 16-byte reads)
    // Which row a lane works on: a 16-byte LDS read is served in groups of 16 lanes that mix lanes of the two rows a
    // 32-lane half holds, and a row of `in` is 18 slots of 16 bytes, so rows y and y + 1 sit two slots apart in the 16-slot
    // bank row and their reads collide; rows y and y + 8 sit 144 = 9 * 16 slots apart, on the same banks, and the lanes'
    // slots then tile the bank row exactly. Within each block of 16 rows the halves of a wave therefore take rows r and
    // r + 8 (the rows left over at the end keep the plain order).
    for (uint32_t i = tid; i < (TX / 4u) * IH; i += 256u) {
        const uint32_t q = i / (TX / 4u), x = (i % (TX / 4u)) * 4u;
        const uint32_t blk = q & ~15u, r = q & 15u;
        const uint32_t y = blk + 16u <= IH ? blk + (r >> 1) + 8u * (r & 1u) : q;
        uint32_t p[12], o[4];
        const u32x4* src = (const u32x4*)&in[y][x];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            u32x4 q = src[v];
            // all four words, also where a small radius uses one of them: narrowed to a 4-byte read the access is banked
            // modulo 32 words, where the lanes of a row (16 words apart in pairs) and of the two rows collide four ways
            asm volatile("" : "+v"(q));
            p[4 * v] = q[0]; p[4 * v + 1] = q[1]; p[4 * v + 2] = q[2]; p[4 * v + 3] = q[3];
        }
        four_outputs<R, 4 - R>(p, o);                     // output x of the tile is centred on word x + 4 of `in`
        *(u32x4*)&in[y][x + 4u] = u32x4{o[0], o[1], o[2], o[3]};
    }
    __syncthreads();
    // vertical pass: a lane makes a block of 4 x VB pixels from VB + 2R rows of four words. The two block rows a 32-lane
    // half works on are eight rows apart (block rows b and b + 2), for the same reason as in the horizontal pass.
    static_assert(TX / 4u * (TY / (uint32_t)VB) == 256u && VB == 4, "one block per lane, block rows eight image rows apart in pairs");
    const uint32_t g = tid / (TX / 4u);
    const uint32_t bx = (tid % (TX / 4u)) * 4u, by = ((g & ~3u) + ((g & 2u) >> 1) + 2u * (g & 1u)) * (uint32_t)VB;
    uint32_t col[4][12] = {}, o[4][VB];
#pragma unroll
    for (int k = 0; k < VB + 2 * R; ++k) {
        const u32x4 q = *(const u32x4*)&in[by + (uint32_t)k][bx + 4u];
        col[0][k] = q[0]; col[1][k] = q[1]; col[2][k] = q[2]; col[3][k] = q[3];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) four_outputs<R, 0, VB>(col[c], o[c]);
    const uint32_t gx = (uint32_t)x0 + bx;
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)VB; ++j) {
        const uint32_t gy = (uint32_t)y0 + by + j;
        if (gy >= row0 + rows) break;
        uint32_t* dst = out + (size_t)gy * w + gx;
        if (VEC && gx + 3u < w) *(u32x4*)dst = u32x4{o[0][j], o[1][j], o[2][j], o[3][j]};
        else {
#pragma unroll
            for (uint32_t c = 0; c < 4u; ++c) if (gx + c < w) dst[c] = o[c][j];
        }
    }
}

}  // namespace

namespace game {

int launch_blur(const uint32_t* raw, uint32_t* out, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows, uint32_t radius, hipStream_t stream)
{
    if (!raw || !out || radius < 1 || radius > 4 || rows == 0 || (uint64_t)row0 + rows > h)
        return set_error(GAME_E_ARGUMENT, "blur: radius must be 1..4 and the band inside the frame");
    const uint32_t tiles = ((w + TX - 1) / TX) * ((rows + TY - 1) / TY);
    const dim3 grid(((tiles + 7u) / 8u) * 8u), block(256);
    // 16-byte tile moves need every row (and the two frame pointers) on a 16-byte boundary
    const bool vec = w % 4u == 0u && ((uintptr_t)raw & 15u) == 0u && ((uintptr_t)out & 15u) == 0u;
#define GAME_BLUR_CASE(R)                                                                                             \
    case R:                                                                                                            \
        if (vec) hipLaunchKernelGGL((blur_kernel<R, true>), grid, block, 0, stream, raw, out, w, h, row0, rows);      \
        else hipLaunchKernelGGL((blur_kernel<R, false>), grid, block, 0, stream, raw, out, w, h, row0, rows);         \
        break;
    switch (radius) {
        GAME_BLUR_CASE(1)
        GAME_BLUR_CASE(2)
        GAME_BLUR_CASE(3)
        GAME_BLUR_CASE(4)
    default: break;
    }
#undef GAME_BLUR_CASE
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GAME_E_HIP, std::string("blur launch: ") + hipGetErrorString(e));
    return GAME_OK;
}

}  // namespace game
