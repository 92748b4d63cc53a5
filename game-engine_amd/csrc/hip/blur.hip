// Separable binomial blur of an RGBA8 frame, one kernel, both passes in LDS.
//
// The reference has no neighbourhood filter (all of its post passes are pointwise); this is the frame filter that
// gives band-sharded rendering a halo. Definition, in integers, so that every machine agrees: weights are the binomial
// row C(2r, k) (sum 4^r); the horizontal pass replaces every channel by (sum of weight * byte + 4^r / 2) >> 2r with
// x clamped to the frame, the vertical pass does the same over y with the horizontal results (themselves bytes).
//
// A workgroup of 256 lanes owns a tile of 64 x 56 pixels: it loads the tile plus four pixels left and right and r rows
// above and below into LDS as 16-byte words (all of a lane's loads in flight together), runs the horizontal pass over
// its 56 + 2r rows in place, then the vertical pass, and stores. Frames whose width is not a multiple of 4 take the same
// kernel with 4-byte global accesses. HBM traffic is one read and one write of the frame (halo pixels are shared
// between neighbouring tiles through L2).
//
// Arithmetic. Two channels share a 32-bit register: R and B of a pixel sit in the 16-bit halves of (p & 0x00ff00ff), G and
// A in those of ((p >> 8) & 0x00ff00ff); a weighted sum is at most 255 * 256 + 128 < 65536, so nothing carries from one
// half into the other. The binomial row C(2R, k) is [1, 1] convolved with itself 2R times, so a run of N + 2R samples
// becomes N weighted sums after 2R rounds of adding neighbours: plain full-rate 32-bit additions, 2R + R (2R - 1) / N of
// them per output and channel pair - the more outputs a lane makes from one run, the closer to 2R. (Weights as
// multiply-adds, the earlier form of this kernel, cost 2R + 1 half-rate instructions per output and pair - every
// three-operand or multiplying integer instruction issues at half rate on gfx950 - and measured 48.7 / 55.7 / 65.2 us per
// 8K frame at radius 2 / 3 / 4 against 44.5 / 53.3 / 60.2 for this form; radius 1 is the same 41 us either way.)
// A lane makes 16 adjacent pixels of a row in the horizontal pass (24 samples, six 16-byte LDS reads; four lanes per row,
// up to 64 rows: one step for the whole workgroup). In the vertical pass it makes a 2 x 8 block at radius 1 and 2; at radius 3
// and 4 it takes 14 rows of one column and streams its samples through the cascade one at a time (8K frame at radius
// 2 / 3 / 4: blocks 41.6 / 48.8 / 56.4 us, streamed 43.0 / 46.7 / 52.7 - the streamed form has fewer additions per output
// but more, narrower loads and stores, which is what counts while the arithmetic is small).
//
// LDS layout. Rows are 19 slots of 16 bytes (the 18 in use and one of padding): with an odd number of slots per row the
// four lanes of a row, which read slots 4 g + v, and the rows r .. r + 7 of a 32-lane half land on 16 distinct slots in
// each of the hardware's 16-lane groups for 16-byte reads ({0-3, 12-15, 20-27} and {4-11, 16-19, 28-31}: rows
// r + {0, 3, 5, 6} and r + {1, 2, 4, 7}, whose offsets 19 * row are distinct modulo 4). One array serves both passes:
// the horizontal pass writes its outputs over the words they were centred on, which is safe because the four lanes that
// share a row sit in one wave - every lane's reads of the row are issued, in lockstep, before any lane's write - and no
// other wave touches that row before the barrier.
#include <hip/hip_runtime.h>
#include "../front/capi_internal.hpp"
#include "blur.hpp"

namespace {

constexpr uint32_t TX = 64;           // pixels per tile row
constexpr uint32_t CTY = 56;          // output rows of a tile: with the rows above and below, at most 64 in the horizontal pass
constexpr uint32_t CSTRIDE = 76;      // words per LDS row: 64 + 4 either side + 4 of padding

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// v[0 .. N + 2R) -> v[0 .. N): every round replaces v[i] by v[i] + v[i + 1]
template <int R, int N>
__device__ __forceinline__ void cascade(uint32_t* v)
{
#pragma unroll
    for (int s = 1; s <= 2 * R; ++s)
#pragma unroll
        for (int i = 0; i < N + 2 * R - s; ++i) v[i] += v[i + 1];
}

// The two weighted sums of a pixel (R | B and G | A in 16-bit halves) rounded and merged into its four bytes: the G | A
// bytes are wanted in bits 8-15 and 24-31, which is where the sums' bits 2R .. 2R + 7 already are after a shift by 8 - 2R.
template <int R>
__device__ __forceinline__ uint32_t finish_sums(uint32_t rb, uint32_t ga)
{
    constexpr uint32_t half = (1u << (2 * R - 1)) * 0x00010001u;
    rb = ((rb + half) >> (2 * R)) & 0x00ff00ffu;
    ga = ((ga + half) << (8 - 2 * R)) & 0xff00ff00u;
    return rb | ga;
}

// Which 16-byte word of the tile (row ly of the LDS array, word c of its 18) lane tid moves in its k-th step. A 16-byte
// LDS store is served in groups of 8 consecutive lanes and banked modulo 128 bytes (8 words of 16 bytes), so steps 0-3
// give every group eight consecutive words of one row (words 0-7 or 8-15: 64 rows x 2 groups = 4 steps of 32 groups), and
// step 4 moves the two words left over in every row: a group takes words 16 and 17 of rows r, r + 2, r + 4, r + 6, whose
// starts (19 words apart per row) fall on distinct pairs of the eight.
__device__ __forceinline__ void tile_word(uint32_t tid, uint32_t k, uint32_t& ly, uint32_t& c)
{
    if (k < 4u) {
        const uint32_t grp = (tid >> 3) + 32u * k;
        ly = grp >> 1;
        c = 8u * (grp & 1u) + (tid & 7u);
    } else {
        const uint32_t oct = tid >> 3, j = tid & 7u;
        ly = (oct >> 1) * 8u + (oct & 1u) + 2u * (j >> 1);       // tid < 128 only: 16 octets cover the 64 rows
        c = 16u + (j & 1u);
    }
}

// The tile's rows y0 - R .. y0 + CTY + R, columns x0 - 4 .. x0 + 67, as 16-byte words, clamped to the frame; all of a
// lane's loads are issued before the first one is waited for (a fixed, unrolled count).
template <int R, bool VEC>
__device__ __forceinline__ void load_tile(const uint32_t* __restrict__ raw, uint32_t w, uint32_t h, int x0, int y0, uint32_t tid,
                                          u32x4 (&held)[5])
{
#pragma unroll
    for (uint32_t k = 0; k < 5u; ++k) {
        uint32_t ly, c;
        tile_word(tid, k, ly, c);
        int gy = y0 - R + (int)ly;
        gy = gy < 0 ? 0 : (gy >= (int)h ? (int)h - 1 : gy);       // clamp to the frame's edge (also keeps idle lanes in range)
        const int gx = x0 - 4 + 4 * (int)c;
        const uint32_t* row = raw + (size_t)(uint32_t)gy * w;
        if (k == 4u && tid >= 128u) { held[k] = u32x4{0u, 0u, 0u, 0u}; continue; }
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
}

// At most 54 registers and 19 KB of LDS: eight workgroups per CU. Occupancy is what this kernel lives on - a workgroup is a chain
// of load, barrier, pass, barrier, pass, store, and only other workgroups' phases fill its waits. (A form in which a
// workgroup works through several tiles with the next tile's loads in flight in registers needs 96 registers, five
// workgroups per CU, and measured 71 us against 60 at radius 4; with a 4 x 8 block per lane in the vertical pass, 117
// registers, 80 us.)
template <int R, bool VEC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void blur_kernel(const uint32_t* __restrict__ raw, uint32_t* __restrict__ out, uint32_t w, uint32_t h,
                                                           uint32_t row0, uint32_t rows)
{
    constexpr uint32_t IH = CTY + 2 * R;                  // rows of the horizontal pass: at most 64
    __shared__ __attribute__((aligned(16))) uint32_t in[IH][CSTRIDE];   // column 0 = frame column x0 - 4
    const uint32_t tid = threadIdx.x;
    // Workgroups are handed to the eight XCDs round robin, and each XCD has its own L2. Neighbouring tiles share their
    // halo pixels (and the 128-byte lines their edges sit in), so the launch is one-dimensional and workgroup L takes
    // tile (L mod 8) * chunk + L / 8: every XCD works through one contiguous eighth of the tile grid, in order, and
    // finds its neighbours' lines in its own L2 (HBM reads are 2.0x the frame without this, 1.01-1.05x with it).
    const uint32_t tiles_x = (w + TX - 1u) / TX, tiles = tiles_x * ((rows + CTY - 1u) / CTY);
    const uint32_t chunk = (tiles + 7u) / 8u;
    const uint32_t tile = (blockIdx.x % 8u) * chunk + blockIdx.x / 8u;
    if (tile >= tiles) return;                            // the whole workgroup leaves, before any barrier
    const int x0 = (int)((tile % tiles_x) * TX), y0 = (int)(row0 + (tile / tiles_x) * CTY);
    u32x4 held[5];
    load_tile<R, VEC>(raw, w, h, x0, y0, tid, held);
#pragma unroll
    for (uint32_t k = 0; k < 5u; ++k) {
        uint32_t ly, c;
        tile_word(tid, k, ly, c);
        if (ly < IH && (k < 4u || tid < 128u)) *(u32x4*)&in[ly][4u * c] = held[k];
    }
    __syncthreads();
    // horizontal pass, in place: lane g of a row reads words 16 g .. 16 g + 23 and writes its 16 outputs over words
    // 16 g + 4 .. 16 g + 19. The four lanes of a row sit in one 32-lane half of a wave and all six reads of every lane
    // are issued before any write (the empty asm below makes all six values a precondition of everything after it).
    // Which row and which quarter a lane takes: a 32-lane half works on eight rows, and with lane bits b0..b4
    // g = b0 + 2 b2 and row = b1 + 2 (b2 xor b3) + 4 b4 both the 16-byte reads (16-lane groups {0-3, 12-15, 20-27} and
    // {4-11, 16-19, 28-31}, banked modulo 16 words of 16 bytes) and the 16-byte stores (groups of 8 consecutive lanes,
    // modulo 8 words) find every lane of a group on a different bank: rows are 19 words apart, and the word a lane reads
    // or writes is 19 row + 4 g + constant (found by searching the bit-linear lane orders; the plain order - four
    // consecutive lanes per row - reads without conflicts but stores 2-way).
    {
        const uint32_t g = (tid & 1u) | ((tid >> 1) & 2u);
        const uint32_t y = (tid >> 5) * 8u + (((tid >> 1) & 1u) | ((((tid >> 2) ^ (tid >> 3)) & 1u) << 1) | (((tid >> 4) & 1u) << 2));
        if (y < IH) {
            const u32x4* src = (const u32x4*)&in[y][16u * g];
            u32x4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3], q4 = src[4], q5 = src[5];
            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5));
            const u32x4 q[6] = {q0, q1, q2, q3, q4, q5};
            uint32_t rb[24], ga[24];
#pragma unroll
            for (int i = 0; i < 24; ++i) { const uint32_t p = q[i >> 2][i & 3]; rb[i] = p & 0x00ff00ffu; ga[i] = (p >> 8) & 0x00ff00ffu; }
            cascade<R, 16>(rb + (4 - R));             // output j of the lane is centred on word 16 g + 4 + j
            cascade<R, 16>(ga + (4 - R));
            u32x4* dst = (u32x4*)&in[y][16u * g + 4u];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = finish_sums<R>(rb[4 - R + 4 * v + e], ga[4 - R + 4 * v + e]);
                dst[v] = o;
            }
        }
    }
    __syncthreads();
    if constexpr (R <= 2) {
        // vertical pass: a lane makes a block of 2 x 8 pixels from 8 + 2R rows of two words (8-byte LDS reads: the 32
        // lanes of a half read one 256-byte run of a row, which no row stride can make collide), 32 lanes across the
        // tile, seven block rows: 224 of the 256 lanes.
        constexpr int CVB = 8;
        const uint32_t bx = (tid & 31u) * 2u, by = (tid >> 5) * (uint32_t)CVB;
        if (by < CTY) {
            u32x2 colq[CVB + 2 * R];
    #pragma unroll
            for (int k = 0; k < CVB + 2 * R; ++k) colq[k] = *(const u32x2*)&in[by + (uint32_t)k][bx + 4u];
            uint32_t o[2][CVB];
    #pragma unroll
            for (int c = 0; c < 2; ++c) {
                uint32_t rb[CVB + 2 * R], ga[CVB + 2 * R];
    #pragma unroll
                for (int k = 0; k < CVB + 2 * R; ++k) { const uint32_t p = colq[k][c]; rb[k] = p & 0x00ff00ffu; ga[k] = (p >> 8) & 0x00ff00ffu; }
                cascade<R, CVB>(rb);
                cascade<R, CVB>(ga);
    #pragma unroll
                for (int j = 0; j < CVB; ++j) o[c][j] = finish_sums<R>(rb[j], ga[j]);
            }
            const uint32_t gx = (uint32_t)x0 + bx;
            const bool even = w % 2u == 0u && ((uintptr_t)out & 7u) == 0u;   // 8-byte stores need the rows on 8-byte boundaries
    #pragma unroll
            for (uint32_t j = 0; j < (uint32_t)CVB; ++j) {
                const uint32_t gy = (uint32_t)y0 + by + j;
                if (gy >= row0 + rows) break;
                uint32_t* dst = out + (size_t)gy * w + gx;
                if (even && gx + 1u < w) *(u32x2*)dst = u32x2{o[0][j], o[1][j]};
                else {
    #pragma unroll
                    for (uint32_t c = 0; c < 2u; ++c) if (gx + c < w) dst[c] = o[c][j];
                }
            }
        }
    } else {
        // vertical pass: a lane takes one column and 14 of the tile's 56 rows (64 columns x 4 blocks: all 256 lanes) and streams
        // its 14 + 2R samples through the cascade one at a time - stage s keeps the value it saw last, so a sample costs 2R
        // additions per channel pair and the state is 2R registers per pair, however long the run (the longer the run, the
        // fewer additions per output: 2R + R (2R - 1) / 14 here). The rounds the first 2R samples cannot take part in fold
        // away, the state starting from zero. 4-byte LDS reads of 64 consecutive words and 4-byte stores of 256 consecutive
        // bytes per wave: no bank conflicts, whole lines.
        {
            constexpr int RUN = (int)(CTY / 4u);              // 14 output rows per lane
            const uint32_t col = tid & 63u, by = (tid >> 6) * (uint32_t)RUN;
            const uint32_t gx = (uint32_t)x0 + col;
            uint32_t srb[2 * R], sga[2 * R];
    #pragma unroll
            for (int s2 = 0; s2 < 2 * R; ++s2) srb[s2] = sga[s2] = 0u;
    #pragma unroll
            for (int n = 0; n < RUN + 2 * R; ++n) {
                const uint32_t p = in[by + (uint32_t)n][col + 4u];
                uint32_t trb = p & 0x00ff00ffu, tga = (p >> 8) & 0x00ff00ffu;
    #pragma unroll
                for (int s2 = 0; s2 < 2 * R; ++s2) {
                    const uint32_t yrb = trb + srb[s2], yga = tga + sga[s2];
                    srb[s2] = trb; sga[s2] = tga;
                    trb = yrb; tga = yga;
                }
                if (n >= 2 * R) {                             // the sum centred on row by + n - R of the array = output row by + n - 2R
                    const uint32_t gy = (uint32_t)y0 + by + (uint32_t)(n - 2 * R);
                    if (gy < row0 + rows && gx < w) out[(size_t)gy * w + gx] = finish_sums<R>(trb, tga);
                }
            }
        }
    }
}

}  // namespace

namespace game {

int launch_blur(const uint32_t* raw, uint32_t* out, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows, uint32_t radius, hipStream_t stream)
{
    if (!raw || !out || radius < 1 || radius > 4 || rows == 0 || (uint64_t)row0 + rows > h)
        return set_error(GAME_E_ARGUMENT, "blur: radius must be 1..4 and the band inside the frame");
    const uint32_t tiles = ((w + TX - 1) / TX) * ((rows + CTY - 1) / CTY);
    const dim3 grid(((tiles + 7u) / 8u) * 8u), block(256);
    // 16-byte tile moves need every row (and the two frame pointers) on a 16-byte boundary
    const bool vec = w % 4u == 0u && ((uintptr_t)raw & 15u) == 0u && ((uintptr_t)out & 15u) == 0u;
#define GAME_BLUR_LAUNCH(...) hipLaunchKernelGGL((__VA_ARGS__), grid, block, 0, stream, raw, out, w, h, row0, rows)
    switch (radius) {
    case 1: if (vec) GAME_BLUR_LAUNCH(blur_kernel<1, true>); else GAME_BLUR_LAUNCH(blur_kernel<1, false>); break;
    case 2: if (vec) GAME_BLUR_LAUNCH(blur_kernel<2, true>); else GAME_BLUR_LAUNCH(blur_kernel<2, false>); break;
    case 3: if (vec) GAME_BLUR_LAUNCH(blur_kernel<3, true>); else GAME_BLUR_LAUNCH(blur_kernel<3, false>); break;
    case 4: if (vec) GAME_BLUR_LAUNCH(blur_kernel<4, true>); else GAME_BLUR_LAUNCH(blur_kernel<4, false>); break;
    default: break;
    }
#undef GAME_BLUR_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GAME_E_HIP, std::string("blur launch: ") + hipGetErrorString(e));
    return GAME_OK;
}

}  // namespace game
