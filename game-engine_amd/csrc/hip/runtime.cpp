// GPU runtime behind include/game_hip.h: hiprtc assembly of the per-cinematic kernels, the frame
// scheduler (frame-batched launches into a ring of RGBA8 frames resident in HBM), the memory
// feedback store (previous frame = previous ring slot) and the SoA particle integrator.
// Stands in for the reference's browser-side renderer loop (game-compiler/src/runtime/helpers.rs:
// 148-235), its intended ping-pong memory runtime (game-compiler/src/codegen/memory.rs:102-147)
// and GameGravitySim (game-compiler/src/codegen/gravity.rs:123-191).
#include "../front/capi_internal.hpp"
#include "codegen.hpp"
#include "scratch.hpp"
#include "blur.hpp"
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>

using namespace game;

static const char* const kDeviceHeader =
#include "game_device_src.inc"
    ;

struct game_hip_ctx {
    Ir ir;
    int device = 0;
    uint32_t w = 0, h = 0;
    hipStream_t stream = nullptr;
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    std::string source;
    KernelShape shape;
    uint32_t row_floats = 0, ustride = 0;
    unsigned lds_bytes = 0;   // dynamic LDS requested per workgroup: unused by the kernel, caps resident workgroups per CU
    // frame ring
    uint32_t ring = 0, next_slot = 0;
    int last_slot = -1;
    uint32_t* frames = nullptr;
    float* utab_dev = nullptr;
    size_t utab_cap = 0;
    const float* lut = nullptr;   // lattice-hash table of this device (shared, not owned)
    const float* vlut = nullptr;  // voronoi feature-point table of this device (shared, not owned)
    const float* slut = nullptr;  // simplex corner table of this device (shared, not owned)
    std::vector<float> utab_host;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;     // events around the launches of the latest call
    // game_hip_timing_begin/_end: every call in between gets an event pair of its own, so the caller can queue
    // many calls without waiting and read the summed device time afterwards
    struct TimedCall { hipEvent_t e0, e1; uint32_t launches; };
    hipEvent_t ev_own0 = nullptr, ev_own1 = nullptr;   // the pair used outside a timing section
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<TimedCall> ev_pending;
    bool timing = false;
    bool own_recorded = false;   // ev_own0 / ev_own1 hold a completed pair (a call outside a timing section recorded them)
    float last_ms = 0.0f;
    uint32_t last_launches = 0;
    // particles
    hipModule_t gmod = nullptr;
    hipFunction_t gfn = nullptr;      // one particle per lane
    hipFunction_t gfn2 = nullptr;     // two particles per lane (tuning hook)
    bool gravity_fast = false;
    std::string gsource;
    uint32_t n = 0;
    float* px[2] = {nullptr, nullptr};
    float* py[2] = {nullptr, nullptr};
    float* vx = nullptr;
    float* vy = nullptr;
    int cur = 0;
    float sim_time = 0.0f;
    // frame filter (game_hip_set_blur): the fused kernel renders into `raw`, the blur writes the ring slot
    uint32_t blur = 0;
    uint32_t* raw = nullptr;
    bool raw_pending = false;   // game_hip_render_raw_band ran and game_hip_blur_band has not consumed it yet
    float compile_ms = 0.0f, tables_ms = 0.0f;   // host time game_hip_create spent in hiprtc / building lattice tables
    // game_hip_render_into: this device's static scratch array (not owned), a page-locked staging buffer of the same
    // size made at creation, and the event that says the last copy out of the staging buffer has finished
    float* scratch = nullptr;
    hipEvent_t ev_scratch = nullptr;   // this device's "last reader of the scratch array" event (shared, not owned)
    float* stage_host = nullptr;
    hipEvent_t ev_stage = nullptr;
    bool stage_busy = false;
};

namespace game {
static std::atomic<int> g_live_contexts{0};
int live_context_count() { return g_live_contexts.load(); }
}  // namespace game

namespace {

struct FrameParams {
    uint32_t* out;
    const uint32_t* prev;
    const float* utab;
    unsigned long long frame_pixels;
    uint32_t w, h, row0, rows;
    uint32_t ustride, slot0, ring, frame0;
    float inv_w, inv_h;
    const float* lut;
    const float* vlut;
    const float* slut;
};
struct GravityParams {
    const float* xin; const float* yin;
    float* xout; float* yout; float* vx; float* vy;
    uint32_t n, first, count, floor_bits;   // floor_bits: 0x1.0c6f7ap-20f, the floor of the squared pair distance
    float dt, time;
};

const uint32_t kGravityLaunchParticles = 1u << 20;   // particles integrated per launch of the gravity kernel (a multiple of 512)

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int hip_fail(const char* what, hipError_t e)
{
    return set_error(e == hipErrorNoDevice || e == hipErrorInvalidDevice ? GAME_E_NO_DEVICE : GAME_E_HIP,
                     std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return hip_fail(#expr, _e); \
    } while (0)

// Compiles HIP source for gfx950 with the device header attached. No device is needed.
// Code objects already built in this process, keyed by options + source: contexts for the same cinematic (one per
// GPU, or re-created per request) share one hiprtc compile. Bounded: the cache is dropped when it reaches 256 entries.
std::mutex g_code_mu;
std::unordered_map<std::string, std::vector<char>> g_code_cache;

int compile_hip_uncached(const std::string& src, const char* name, const std::vector<std::string>& flags, std::vector<char>& code);

int compile_hip(const std::string& src, const char* name, std::vector<char>& code, const std::vector<std::string>& extra = {})
{
    // One rounding per source operator: contraction off; fma only where the source says fma.
    // No SLP vectorisation: packed FP32 (v_pk_*_f32) issues at half rate on gfx950, so pairing two
    // scalar operations into one packed instruction only adds the moves that build the pairs.
    std::vector<std::string> flags = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"};
    flags.insert(flags.end(), extra.begin(), extra.end());
    // Tuning hook: extra hiprtc options, space separated (later options override earlier ones).
    if (const char* extra = getenv("GAME_HIP_FLAGS")) {
        std::string e(extra);
        size_t i = 0;
        while (i < e.size()) {
            size_t j = e.find(' ', i);
            if (j == std::string::npos) j = e.size();
            if (j > i) flags.push_back(e.substr(i, j - i));
            i = j + 1;
        }
    }
    std::string key;
    for (const std::string& f : flags) { key += f; key += '\n'; }
    key += '\0';
    key += src;
    {
        std::lock_guard<std::mutex> lock(g_code_mu);
        auto it = g_code_cache.find(key);
        if (it != g_code_cache.end()) { code = it->second; return GAME_OK; }
    }
    int rc = compile_hip_uncached(src, name, flags, code);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(g_code_mu);
    if (g_code_cache.size() >= 256) g_code_cache.clear();
    g_code_cache.emplace(std::move(key), code);
    return GAME_OK;
}

int compile_hip_uncached(const std::string& src, const char* name, const std::vector<std::string>& flags, std::vector<char>& code)
{
    hiprtcProgram prog;
    const char* hdr_src[] = {kDeviceHeader};
    const char* hdr_name[] = {"game_device.h"};
    hiprtcResult r = hiprtcCreateProgram(&prog, src.c_str(), name, 1, hdr_src, hdr_name);
    if (r != HIPRTC_SUCCESS) return set_error(GAME_E_HIP, std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r));
    std::vector<const char*> opts;
    for (const std::string& f : flags) opts.push_back(f.c_str());
    r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return set_error(GAME_E_HIP, std::string("hiprtc compile of ") + name + " failed: " + hiprtcGetErrorString(r) + "\n" + log);
    }
    size_t n = 0;
    hiprtcGetCodeSize(prog, &n);
    code.resize(n);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    return GAME_OK;
}

bool uses_builtin(const Ir& ir, uint16_t id)
{
    for (const IrStage& st : ir.stages)
        if (st.builtin == id) return true;
    return false;
}

// True when some stage evaluates noise2 (value noise on the integer lattice).
bool uses_lattice_noise(const Ir& ir)
{
    for (const IrStage& st : ir.stages)
        if (st.builtin == B_FBM || st.builtin == B_DOMAIN_WARP || st.builtin == B_CURL_NOISE || st.builtin == B_DISPLACE) return true;
    return false;
}

// Pixels per lane. Cinematics whose stages are a few dozen instructions per pixel are bound by the
// frame store (and the previous-frame load): four pixels per lane turn those into 16-byte accesses.
// Noise-heavy cinematics are bound by VALU issue and keep one pixel per lane (fewest registers,
// no unrolled copies of a long body). Rows stay 16-byte aligned only when the width is a multiple of 4.
KernelShape shape_for(const Ir& ir, uint32_t width, uint32_t height, uint32_t flags = 0)
{
    KernelShape s;
    s.wgsl = (flags & GAME_HIP_SEMANTICS_WGSL) != 0;
    bool heavy = false;
    for (const IrStage& st : ir.stages)
        switch (st.builtin) {
        case B_FBM: case B_SIMPLEX: case B_VORONOI: case B_DOMAIN_WARP: case B_CURL_NOISE: case B_CONCENTRIC_WAVES: case B_DISPLACE:
            heavy = true; break;
        default: break;
        }
    s.pixels_per_lane = (!heavy && width % 4u == 0u) ? 4u : 1u;
    if (const char* e = getenv("GAME_HIP_PPL")) {   // tuning hook
        int v = atoi(e);
        if (v == 1 || ((v == 4 || v == 8 || v == 16) && width % 4u == 0u)) s.pixels_per_lane = (uint32_t)v;
    }
    // Rows per lane. A lane that walks several rows of its columns computes what depends on the column alone once
    // (the compiler moves it out of the row loop): the x half of the prologue for every cinematic, whole stages for
    // pipelines that start from uv.x (spectrum and what follows it up to the first stage that reads y). Measured:
    // c3 +31 % at 8 rows, c2 +8 % at 8, c1 +9 % at 4 (its 1024-row frames run out of workgroups beyond that).
    // Memory cinematics keep one row per lane: c4 loses 9 % at two rows with the previous-frame load inside the row loop
    // and still 1.5 % (6 % at four rows) with all of a lane's loads requested up front - it runs near its memory
    // floor and prefers many short waves.
    if (!ir.uses_memory) {
        bool column_first = false;
        for (const IrLayer& L : ir.layers)
            if (L.is_pipeline && L.stage_count && ir.stages[L.stage_begin].builtin == B_SPECTRUM) column_first = true;
        s.rows_per_lane = (heavy || column_first) ? 8u : 4u;
        // Noise cinematics gain from longer runs of rows: what a lane keeps across rows (the lattice values of a leading
        // domain_warp, everything computed from the column) is paid for once per run, and the per-row values of the
        // workgroup once per 32 rows. c2 at 4K: +6 % at 32 rows over 8, +7 % at 64. Frames need the height for it: a
        // 2160-row frame still has 68 workgroup rows at 32.
        if (heavy) s.rows_per_lane = height >= 1024u ? 32u : height >= 512u ? 16u : 8u;
    }
    if (const char* e = getenv("GAME_HIP_RPL")) {   // tuning hook
        int v = atoi(e);
        if (v >= 1 && v <= 64) s.rows_per_lane = (uint32_t)v;
    }
    if (const char* e = getenv("GAME_HIP_ROWLDS")) s.row_shared = atoi(e) != 0;   // tuning hook
    // Feedback frames too large for the two of them (the one read, the one written) to stay in the 256 MB memory-side
    // cache stream the previous frame in with a non-temporal hint: 8K 44.9 -> 42.2 us per frame; at 4K and below the
    // plain load is as fast or faster (the previous frame is still cached when it is read).
    s.stream_prev = ir.uses_memory && (unsigned long long)width * height * 8ull > 200000000ull;
    if (const char* e = getenv("GAME_HIP_NTLOAD")) s.stream_prev = atoi(e) != 0;   // tuning hook
    // Four-pixel lanes run a little faster in one-wave workgroups (a row of 3840 pixels is 15 of them instead of 3.75
    // four-wave ones: c3 +1.7 %, c4 +3 %, c1 unchanged in alternating bench runs on one box).
    if (s.pixels_per_lane == 4u) s.wg = 64u;
    if (const char* e = getenv("GAME_HIP_WG")) {    // tuning hook
        int v = atoi(e);
        if (v == 64 || v == 128 || v == 256 || v == 512 || v == 1024) s.wg = (uint32_t)v;
    }
    // Four-pixel lanes (no noise stage) speculate on their range tests: one check per wave and row instead of a branch per
    // test, which lets the four pixels' arithmetic overlap (c3 -10 %, c1 -4 % with the scheduler chosen for such kernels).
    s.speculate = s.pixels_per_lane == 4u;
    if (const char* e = getenv("GAME_HIP_SPECULATE")) s.speculate = atoi(e) != 0 && s.pixels_per_lane == 4u;   // tuning hook
    uint32_t row = BU_COUNT + (uint32_t)ir.uniforms.size();
    s.time_slot = row;
    s.aspect_slot = row + 1;
    s.animate = (flags & GAME_HIP_ANIMATE_ARGS) != 0 && (!ir.derived.empty() || !ir.pixel_exprs.empty());
    s.derived_slot = row + 2;
    s.lens_post = (flags & GAME_HIP_LENS_POST) != 0;
    s.project = (flags & GAME_HIP_PROJECT) != 0;
    return s;
}

// Compiler options that go with a kernel shape. A speculating kernel is one long basic block per row (four independent
// pixel chains); the default scheduler (tuned for occupancy) groups the slow instructions of the four pixels together and
// the block then runs a third slower than the branchy form, while the ILP scheduler interleaves the chains: c3 0.37 ms
// per 25 frames with the default, 0.244 ms with max-ilp, 0.27 ms for the branchy kernel.
std::vector<std::string> shape_flags(const KernelShape& s)
{
    if (s.speculate) return {"-mllvm", "-amdgpu-sched-strategy=max-ilp"};
    return {};
}

// fract as the kernels define it: x - floor(x), kept below 1.
float host_fract(float x)
{
    float f = x - floorf(x);
    if (f != f) return f;
    return f < 1.0f ? f : 0x1.fffffep-1f;
}

int ensure_ring(game_hip_ctx* c, uint32_t ring)
{
    if (c->frames && c->ring == ring) return GAME_OK;
    // reject a ring whose byte count does not fit the device before anything is released
    const unsigned long long frame_bytes = (unsigned long long)c->w * c->h * 4ull;
    size_t dev_free = 0, dev_total = 0;
    HIP_TRY(hipMemGetInfo(&dev_free, &dev_total));
    if ((unsigned long long)ring > (unsigned long long)dev_total / frame_bytes)
        return set_error(GAME_E_ARGUMENT, "game_hip_set_ring: " + std::to_string(ring) + " frames of " + std::to_string(frame_bytes) +
                                              " bytes exceed the device's memory (" + std::to_string(dev_total) + " bytes)");
    if (c->frames) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->frames)); c->frames = nullptr; c->ring = 0; }
    size_t bytes = (size_t)ring * frame_bytes;
    HIP_TRY(hipMalloc((void**)&c->frames, bytes));
    HIP_TRY(hipMemsetAsync(c->frames, 0, bytes, c->stream));
    c->ring = ring;
    c->next_slot = 0;
    c->last_slot = -1;
    return GAME_OK;
}

// The kernel's frame rows: the caller's floats, then the two values the shader derives once per frame, then (animate
// shapes) the values of the expression-valued stage arguments.
void build_rows(const game_hip_ctx* c, const float* uniforms, uint32_t nframes, float* out)
{
    for (uint32_t f = 0; f < nframes; ++f) {
        const float* src = uniforms + (size_t)f * c->row_floats;
        float* dst = out + (size_t)f * c->ustride;
        memcpy(dst, src, (size_t)c->row_floats * 4);
        dst[c->shape.time_slot] = host_fract(src[BU_TIME] / 120.0f) * 120.0f;   // glsl.rs:56
        dst[c->shape.aspect_slot] = src[BU_RES_X] / src[BU_RES_Y];              // glsl.rs:55
        if (c->shape.animate) eval_derived(c->ir, src, dst + c->shape.derived_slot);
    }
}

int upload_rows(game_hip_ctx* c, const float* uniforms, uint32_t nframes)
{
    size_t need = (size_t)nframes * c->ustride;
    c->utab_host.resize(need);
    build_rows(c, uniforms, nframes, c->utab_host.data());
    if (need > c->utab_cap) {
        if (c->utab_dev) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->utab_dev)); c->utab_dev = nullptr; c->utab_cap = 0; }
        size_t cap = need < 4096 ? 4096 : need;
        HIP_TRY(hipMalloc((void**)&c->utab_dev, cap * 4));
        c->utab_cap = cap;
    }
    HIP_TRY(hipMemcpyAsync(c->utab_dev, c->utab_host.data(), need * 4, hipMemcpyHostToDevice, c->stream));
    return GAME_OK;
}

// Frames [frame0, frame0 + count) of the uploaded rows `utab` into slots slot0.. of the `ring`-slot frame array `out`.
int launch_frames_into(game_hip_ctx* c, uint32_t* out, uint32_t ring, const float* utab, uint32_t frame0, uint32_t count, uint32_t slot0,
                       const uint32_t* prev, uint32_t row0, uint32_t rows)
{
    // the kernel wraps slot0 + frame index with one conditional subtract
    if (count == 0 || count > ring || slot0 >= ring) return set_error(GAME_E_ARGUMENT, "launch_frames: frame batch does not fit the ring");
    FrameParams P;
    P.out = out;
    P.prev = prev;
    P.utab = utab;
    P.frame_pixels = (unsigned long long)c->w * c->h;
    P.w = c->w; P.h = c->h; P.row0 = row0; P.rows = rows;
    P.ustride = c->ustride; P.slot0 = slot0; P.ring = ring; P.frame0 = frame0;
    P.inv_w = 1.0f / (float)c->w;
    P.inv_h = 1.0f / (float)c->h;
    P.lut = c->lut;
    P.vlut = c->vlut;
    P.slut = c->slut;
    void* args[] = {&P};
    uint32_t gx = (c->w + c->shape.wg * c->shape.pixels_per_lane - 1u) / (c->shape.wg * c->shape.pixels_per_lane);
    uint32_t gy = (rows + c->shape.rows_per_lane - 1u) / c->shape.rows_per_lane;   // rows <= height <= 65535 is checked at creation
    HIP_TRY(hipModuleLaunchKernel(c->fn, gx, gy, count, c->shape.wg, 1, 1, c->lds_bytes, c->stream, args, nullptr));
    return GAME_OK;
}

int launch_frames(game_hip_ctx* c, uint32_t frame0, uint32_t count, uint32_t slot0, const uint32_t* prev, uint32_t row0,
                  uint32_t rows)
{
    return launch_frames_into(c, c->frames, c->ring, c->utab_dev, frame0, count, slot0, prev, row0, rows);
}

// Picks the event pair for the call that is starting.
int begin_call_events(game_hip_ctx* c)
{
    if (!c->timing) { c->ev0 = c->ev_own0; c->ev1 = c->ev_own1; return GAME_OK; }
    if (c->ev_pool.empty()) {
        hipEvent_t a = nullptr, b = nullptr;
        HIP_TRY(hipEventCreate(&a));
        hipError_t e = hipEventCreate(&b);
        if (e != hipSuccess) { (void)hipEventDestroy(a); return hip_fail("hipEventCreate", e); }
        c->ev_pool.emplace_back(a, b);
    }
    c->ev0 = c->ev_pool.back().first; c->ev1 = c->ev_pool.back().second;
    c->ev_pool.pop_back();
    return GAME_OK;
}
void end_call_events(game_hip_ctx* c)
{
    if (c->timing) c->ev_pending.push_back({c->ev0, c->ev1, c->last_launches});
    else c->own_recorded = true;
}
// Between begin_call_events and end_call_events: a call that fails on the way hands its event pair back to the pool
// (inside a timing section the pair would otherwise belong to neither the pool nor the pending list).
struct CallEvents {
    game_hip_ctx* c;
    bool done = false;
    explicit CallEvents(game_hip_ctx* ctx) : c(ctx) {}
    void finish() { end_call_events(c); done = true; }
    ~CallEvents()
    {
        if (done) return;
        if (c->timing) { c->ev_pool.emplace_back(c->ev0, c->ev1); c->ev0 = c->ev_own0; c->ev1 = c->ev_own1; }
        else c->own_recorded = false;   // the context's own pair may hold a start without an end
    }
};

int render_impl(game_hip_ctx* c, const float* uniforms, uint32_t nframes, uint32_t row0, uint32_t rows, uint8_t* host)
{
    if (!c || !uniforms) return set_error(GAME_E_ARGUMENT, "game_hip_render: null argument");
    if (rows == 0 || (uint64_t)row0 + rows > c->h) return set_error(GAME_E_ARGUMENT, "game_hip_render: band outside the image");
    if (nframes == 0) return GAME_OK;
    if (!c->frames || c->ring == 0) return set_error(GAME_E_HIP, "game_hip_render: the frame ring is not allocated (an earlier game_hip_set_ring failed)");
    HIP_TRY(hipSetDevice(c->device));
    int rc = upload_rows(c, uniforms, nframes);
    if (rc) return rc;
    const size_t frame_px = (size_t)c->w * c->h;
    const size_t band_bytes = (size_t)rows * c->w * 4;
    c->last_launches = 0;
    rc = begin_call_events(c);
    if (rc) return rc;
    CallEvents events(c);
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    if (c->blur) {
        // frame by frame: fused kernel -> raw frame, blur -> ring slot; a memory cinematic reads the blurred slot of the
        // frame before. Rows of `raw` outside the band are read as they are (the caller's halo rows).
        for (uint32_t f = 0; f < nframes; ++f) {
            uint32_t slot = c->next_slot;
            const uint32_t* prev = c->ir.uses_memory && c->last_slot >= 0 ? c->frames + (size_t)c->last_slot * frame_px : nullptr;
            rc = launch_frames_into(c, c->raw, 1, c->utab_dev, f, 1, 0, prev, row0, rows);
            if (rc) return rc;
            rc = launch_blur(c->raw, c->frames + (size_t)slot * frame_px, c->w, c->h, row0, rows, c->blur, c->stream);
            if (rc) return rc;
            c->last_launches += 2;
            if (host)
                HIP_TRY(hipMemcpyAsync(host + (size_t)f * band_bytes, c->frames + (size_t)slot * frame_px + (size_t)row0 * c->w,
                                       band_bytes, hipMemcpyDeviceToHost, c->stream));
            c->last_slot = (int)slot;
            c->next_slot = (slot + 1) % c->ring;
        }
    } else if (c->ir.uses_memory) {
        // frame-serial: frame k reads the slot frame k-1 was written to
        for (uint32_t f = 0; f < nframes; ++f) {
            uint32_t slot = c->next_slot;
            const uint32_t* prev = c->last_slot >= 0 ? c->frames + (size_t)c->last_slot * frame_px : nullptr;
            rc = launch_frames(c, f, 1, slot, prev, row0, rows);
            if (rc) return rc;
            ++c->last_launches;
            if (host)
                HIP_TRY(hipMemcpyAsync(host + (size_t)f * band_bytes, c->frames + (size_t)slot * frame_px + (size_t)row0 * c->w,
                                       band_bytes, hipMemcpyDeviceToHost, c->stream));
            c->last_slot = (int)slot;
            c->next_slot = (slot + 1) % c->ring;
        }
    } else {
        uint32_t done = 0;
        while (done < nframes) {
            uint32_t chunk = nframes - done < c->ring ? nframes - done : c->ring;
            if (chunk > 65535u) chunk = 65535u;   // grid.z limit
            uint32_t slot0 = c->next_slot;
            rc = launch_frames(c, done, chunk, slot0, nullptr, row0, rows);
            if (rc) return rc;
            ++c->last_launches;
            if (host)
                for (uint32_t k = 0; k < chunk; ++k) {
                    uint32_t slot = (slot0 + k) % c->ring;
                    HIP_TRY(hipMemcpyAsync(host + (size_t)(done + k) * band_bytes,
                                           c->frames + (size_t)slot * frame_px + (size_t)row0 * c->w, band_bytes,
                                           hipMemcpyDeviceToHost, c->stream));
                }
            c->next_slot = (slot0 + chunk) % c->ring;
            c->last_slot = (int)((slot0 + chunk - 1) % c->ring);
            done += chunk;
        }
    }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    events.finish();
    if (host) HIP_TRY(hipStreamSynchronize(c->stream));
    return GAME_OK;
}

}  // namespace

extern "C" {

int game_hip_device_count(int* count)
{
    if (!count) return set_error(GAME_E_ARGUMENT, "game_hip_device_count: null argument");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) return hip_fail("hipGetDeviceCount", e);
    return GAME_OK;
}

int game_hip_compile_only(const game_ir* ir, size_t* code_bytes) { return game_hip_compile_only_ex(ir, 0, code_bytes); }

int game_hip_compile_only_ex(const game_ir* ir, uint32_t flags, size_t* code_bytes)
{
    if (!ir) return set_error(GAME_E_ARGUMENT, "game_hip_compile_only: null ir");
    std::vector<char> code;
    const KernelShape shape = shape_for(*ir->ir, 3840u, 2160u, flags);
    std::string src = generate_frame_kernel(*ir->ir, shape);
    int rc = compile_hip(src, "game_frame.hip", code, shape_flags(shape));
    if (rc) return rc;
    size_t total = code.size();
    if (ir->ir->gravity.present) {
        rc = compile_hip(generate_gravity_kernel(*ir->ir), "game_gravity.hip", code);
        if (rc) return rc;
        total += code.size();
    }
    if (code_bytes) *code_bytes = total;
    return GAME_OK;
}

int game_hip_dump_code(const game_ir* ir, const char* path_prefix)
{
    if (!ir || !path_prefix) return set_error(GAME_E_ARGUMENT, "game_hip_dump_code: null argument");
    auto write = [&](const std::string& path, const void* data, size_t n) {
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) return false;
        bool ok = fwrite(data, 1, n, f) == n;
        fclose(f);
        return ok;
    };
    std::vector<char> code;
    const KernelShape shape = shape_for(*ir->ir, 3840u, 2160u);
    std::string src = generate_frame_kernel(*ir->ir, shape);
    int rc = compile_hip(src, "game_frame.hip", code, shape_flags(shape));
    if (rc) return rc;
    std::string pre(path_prefix);
    if (!write(pre + "_frame.hip", src.data(), src.size()) || !write(pre + "_frame.co", code.data(), code.size()))
        return set_error(GAME_E_ARGUMENT, "game_hip_dump_code: cannot write " + pre + "_frame.*");
    if (ir->ir->gravity.present) {
        src = generate_gravity_kernel(*ir->ir);
        rc = compile_hip(src, "game_gravity.hip", code);
        if (rc) return rc;
        if (!write(pre + "_gravity.hip", src.data(), src.size()) || !write(pre + "_gravity.co", code.data(), code.size()))
            return set_error(GAME_E_ARGUMENT, "game_hip_dump_code: cannot write " + pre + "_gravity.*");
    }
    return GAME_OK;
}

int game_hip_create(const game_ir* ir, uint32_t width, uint32_t height, int device, game_hip_ctx** out)
{
    return game_hip_create_ex(ir, width, height, device, 0, out);
}

int game_hip_create_ex(const game_ir* ir, uint32_t width, uint32_t height, int device, uint32_t flags, game_hip_ctx** out)
{
    if (!ir || !out) return set_error(GAME_E_ARGUMENT, "game_hip_create: null argument");
    if (flags & ~(uint32_t)(GAME_HIP_SEMANTICS_WGSL | GAME_HIP_ANIMATE_ARGS | GAME_HIP_LENS_POST | GAME_HIP_GRAVITY_FAST | GAME_HIP_PROJECT))
        return set_error(GAME_E_ARGUMENT, "game_hip_create_ex: unknown flag");
    *out = nullptr;
    if (width == 0 || height == 0 || width > 65536 || height > 65535)
        return set_error(GAME_E_ARGUMENT, "game_hip_create: width must be in 1..65536 and height in 1..65535");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return set_error(GAME_E_NO_DEVICE, std::string("game_hip_create: no HIP device available (") +
                                               (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") + ")");
    if (device < 0 || device >= ndev) return set_error(GAME_E_ARGUMENT, "game_hip_create: device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    auto* c = new game_hip_ctx;
    ++g_live_contexts;   // game_hip_destroy, which every failure path below goes through, takes it back
    c->ir = *ir->ir;
    c->device = device;
    c->w = width; c->h = height;
    c->shape = shape_for(c->ir, c->w, c->h, flags);
    if (const char* e = getenv("GAME_HIP_LDS_BYTES")) c->lds_bytes = (unsigned)atoi(e);   // tuning hook, read per context
    c->row_floats = BU_COUNT + (uint32_t)c->ir.uniforms.size();
    c->ustride = c->row_floats + 2 + (c->shape.animate ? (uint32_t)c->ir.derived.size() : 0u);
    c->source = generate_frame_kernel(c->ir, c->shape);
    std::vector<char> code;
    double t0 = now_ms();
    int rc = compile_hip(c->source, "game_frame.hip", code, shape_flags(c->shape));
    c->compile_ms = (float)(now_ms() - t0);
    if (rc) { game_hip_destroy(c); return rc; }
    auto fail = [&](const char* what, hipError_t err) { int r = hip_fail(what, err); game_hip_destroy(c); return r; };
    if ((e = hipModuleLoadData(&c->mod, code.data())) != hipSuccess) return fail("hipModuleLoadData", e);
    if ((e = hipModuleGetFunction(&c->fn, c->mod, "game_frame")) != hipSuccess) return fail("hipModuleGetFunction", e);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipEventCreate(&c->ev_own0)) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev_own1)) != hipSuccess) return fail("hipEventCreate", e);
    c->ev0 = c->ev_own0; c->ev1 = c->ev_own1;
    if ((e = hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipHostMalloc((void**)&c->stage_host, GAME_HIP_SCRATCH_BYTES, hipHostMallocDefault)) != hipSuccess) return fail("hipHostMalloc", e);
    {
        void* sp = nullptr; void* sev = nullptr;
        rc = scratch_for_current_device(&sp, &sev);
        if (rc) { game_hip_destroy(c); return rc; }
        c->scratch = (float*)sp; c->ev_scratch = (hipEvent_t)sev;
    }
    rc = ensure_ring(c, c->ir.uses_memory ? 2u : 4u);
    if (rc) { game_hip_destroy(c); return rc; }
    t0 = now_ms();
    if (uses_lattice_noise(c->ir) && !getenv("GAME_HIP_NO_LUT")) {   // the variable is a tuning hook
        rc = lut_for_current_device(0, &c->lut);
        if (rc) { game_hip_destroy(c); return rc; }
    }
    // the feature-point table serves voronoi in the WebGL2 wording and the simplex gradients in the WebGPU wording;
    // the WebGPU voronoi hashes with hash2 and reads the lattice table
    const bool wgsl = c->shape.wgsl;
    if (wgsl && uses_builtin(c->ir, B_VORONOI) && !c->lut && !getenv("GAME_HIP_NO_LUT")) {
        rc = lut_for_current_device(0, &c->lut);
        if (rc) { game_hip_destroy(c); return rc; }
    }
    if (uses_builtin(c->ir, wgsl ? B_SIMPLEX : B_VORONOI) && !getenv("GAME_HIP_NO_LUT")) {
        rc = lut_for_current_device(1, &c->vlut);
        if (rc) { game_hip_destroy(c); return rc; }
    }
    if (!wgsl && uses_builtin(c->ir, B_SIMPLEX) && !getenv("GAME_HIP_NO_LUT")) {
        rc = lut_for_current_device(2, &c->slut);
        if (rc) { game_hip_destroy(c); return rc; }
    }
    c->tables_ms = (float)(now_ms() - t0);
    if (c->ir.gravity.present) {
        c->gravity_fast = (flags & GAME_HIP_GRAVITY_FAST) != 0;
        c->gsource = generate_gravity_kernel(c->ir, c->gravity_fast);
        t0 = now_ms();
        rc = compile_hip(c->gsource, "game_gravity.hip", code);
        c->compile_ms += (float)(now_ms() - t0);
        if (rc) { game_hip_destroy(c); return rc; }
        if ((e = hipModuleLoadData(&c->gmod, code.data())) != hipSuccess) return fail("hipModuleLoadData(gravity)", e);
        if ((e = hipModuleGetFunction(&c->gfn, c->gmod, "game_gravity")) != hipSuccess) return fail("hipModuleGetFunction(gravity)", e);
        if ((e = hipModuleGetFunction(&c->gfn2, c->gmod, "game_gravity_p2")) != hipSuccess) return fail("hipModuleGetFunction(gravity)", e);
    }
    *out = c;
    return GAME_OK;
}

void game_hip_destroy(game_hip_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->frames) (void)hipFree(c->frames);
    if (c->raw) (void)hipFree(c->raw);
    if (c->utab_dev) (void)hipFree(c->utab_dev);
    for (int i = 0; i < 2; ++i) {
        if (c->px[i]) (void)hipFree(c->px[i]);
        if (c->py[i]) (void)hipFree(c->py[i]);
    }
    if (c->vx) (void)hipFree(c->vx);
    if (c->vy) (void)hipFree(c->vy);
    if (c->stage_host) (void)hipHostFree(c->stage_host);
    if (c->ev_stage) (void)hipEventDestroy(c->ev_stage);
    if (c->ev_own0) (void)hipEventDestroy(c->ev_own0);
    if (c->ev_own1) (void)hipEventDestroy(c->ev_own1);
    for (auto& p : c->ev_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto& t : c->ev_pending) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
    if (c->mod) (void)hipModuleUnload(c->mod);
    if (c->gmod) (void)hipModuleUnload(c->gmod);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    --g_live_contexts;
}

int game_hip_set_ring(game_hip_ctx* c, uint32_t ring_frames)
{
    if (!c) return set_error(GAME_E_ARGUMENT, "game_hip_set_ring: null context");
    if (ring_frames == 0 || (c->ir.uses_memory && ring_frames < 2))
        return set_error(GAME_E_ARGUMENT, "game_hip_set_ring: ring needs >= 1 slot (>= 2 for memory cinematics)");
    HIP_TRY(hipSetDevice(c->device));
    return ensure_ring(c, ring_frames);
}

int game_hip_render(game_hip_ctx* c, const float* uniforms, uint32_t nframes, uint8_t* rgba8_host)
{
    if (!c) return set_error(GAME_E_ARGUMENT, "game_hip_render: null context");
    return render_impl(c, uniforms, nframes, 0, c->h, rgba8_host);
}

int game_hip_render_band(game_hip_ctx* c, const float* uniforms, uint32_t nframes, uint32_t row0, uint32_t rows,
                         uint8_t* rgba8_host)
{
    if (!c) return set_error(GAME_E_ARGUMENT, "game_hip_render_band: null context");
    return render_impl(c, uniforms, nframes, row0, rows, rgba8_host);
}

int game_hip_render_into(game_hip_ctx* c, const float* uniforms, uint32_t nframes, void* rgba8_dev)
{
    if (!c || !uniforms || !rgba8_dev) return set_error(GAME_E_ARGUMENT, "game_hip_render_into: null argument");
    if (nframes == 0) return GAME_OK;
    const size_t row_bytes = (size_t)c->ustride * 4;
    if (nframes > 65535u || (size_t)nframes * row_bytes > GAME_HIP_SCRATCH_BYTES)
        return set_error(GAME_E_ARGUMENT, "game_hip_render_into: " + std::to_string(nframes) + " frames of " + std::to_string(row_bytes) +
                                              " bytes of uniforms exceed the static scratch (" + std::to_string(GAME_HIP_SCRATCH_BYTES) +
                                              " bytes, at most " + std::to_string(std::min<size_t>(65535u, GAME_HIP_SCRATCH_BYTES / row_bytes)) +
                                              " frames per call for this cinematic)");
    if (((uintptr_t)rgba8_dev & 15u) != 0) return set_error(GAME_E_ARGUMENT, "game_hip_render_into: the output must be 16-byte aligned");
    // the frame filter needs the raw frame and a second kernel per frame; this entry point renders unfiltered frames only
    if (c->blur || c->raw_pending)
        return set_error(GAME_E_UNSUPPORTED, "game_hip_render_into: the context has a frame filter set (game_hip_set_blur); use game_hip_render");
    HIP_TRY(hipSetDevice(c->device));
    // the staging buffer is free again once the previous call's copy out of it has run
    if (c->stage_busy) { HIP_TRY(hipEventSynchronize(c->ev_stage)); c->stage_busy = false; }
    build_rows(c, uniforms, nframes, c->stage_host);
    // every context of this GPU stages its rows in the same array: the refill waits (on the device) for the kernels of
    // whichever context read it last
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_scratch, 0));
    HIP_TRY(hipMemcpyAsync(c->scratch, c->stage_host, (size_t)nframes * row_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(c->ev_stage, c->stream));
    c->stage_busy = true;
    uint32_t* out = (uint32_t*)rgba8_dev;
    const size_t frame_px = (size_t)c->w * c->h;
    const bool timed = !c->timing;   // the context's own event pair; a timing section's per-call pairs may need creating
    c->last_launches = 0;
    if (timed) { c->own_recorded = false; HIP_TRY(hipEventRecord(c->ev_own0, c->stream)); }
    if (c->ir.uses_memory) {
        // frame-serial: frame 0 continues from the context's feedback state, frame k reads frame k - 1 of the output
        const uint32_t* prev = c->last_slot >= 0 && c->frames ? c->frames + (size_t)c->last_slot * frame_px : nullptr;
        for (uint32_t f = 0; f < nframes; ++f) {
            int rc = launch_frames_into(c, out, nframes, c->scratch, f, 1, f, prev, 0, c->h);
            if (rc) return rc;
            ++c->last_launches;
            prev = out + (size_t)f * frame_px;
        }
        if (c->frames && c->ring) {   // the last frame becomes the state the next call (of either kind) continues from
            const uint32_t slot = c->next_slot;
            HIP_TRY(hipMemcpyAsync(c->frames + (size_t)slot * frame_px, out + (size_t)(nframes - 1) * frame_px, frame_px * 4,
                                   hipMemcpyDeviceToDevice, c->stream));
            c->last_slot = (int)slot;
            c->next_slot = (slot + 1u) % c->ring;
        }
    } else {
        int rc = launch_frames_into(c, out, nframes, c->scratch, 0, nframes, 0, nullptr, 0, c->h);
        if (rc) return rc;
        ++c->last_launches;
    }
    HIP_TRY(hipEventRecord(c->ev_scratch, c->stream));
    if (timed) { HIP_TRY(hipEventRecord(c->ev_own1, c->stream)); c->own_recorded = true; }
    return GAME_OK;
}

int game_hip_set_blur(game_hip_ctx* c, uint32_t radius)
{
    if (!c) return set_error(GAME_E_ARGUMENT, "game_hip_set_blur: null context");
    if (radius > 4) return set_error(GAME_E_ARGUMENT, "game_hip_set_blur: radius must be 0 (off) to 4");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (radius && !c->raw) {
        const size_t bytes = (size_t)c->w * c->h * 4;
        HIP_TRY(hipMalloc((void**)&c->raw, bytes));
        HIP_TRY(hipMemsetAsync(c->raw, 0, bytes, c->stream));
    }
    if (!radius && c->raw) { HIP_TRY(hipFree(c->raw)); c->raw = nullptr; }
    c->blur = radius;
    c->raw_pending = false;
    return GAME_OK;
}

int game_hip_render_raw_band(game_hip_ctx* c, const float* uniforms, uint32_t row0, uint32_t rows)
{
    if (!c || !uniforms) return set_error(GAME_E_ARGUMENT, "game_hip_render_raw_band: null argument");
    if (!c->blur || !c->raw) return set_error(GAME_E_UNSUPPORTED, "game_hip_render_raw_band: no frame filter is set (game_hip_set_blur)");
    if (rows == 0 || (uint64_t)row0 + rows > c->h) return set_error(GAME_E_ARGUMENT, "game_hip_render_raw_band: band outside the image");
    HIP_TRY(hipSetDevice(c->device));
    int rc = upload_rows(c, uniforms, 1);
    if (rc) return rc;
    const size_t frame_px = (size_t)c->w * c->h;
    const uint32_t* prev = c->ir.uses_memory && c->last_slot >= 0 ? c->frames + (size_t)c->last_slot * frame_px : nullptr;
    rc = launch_frames_into(c, c->raw, 1, c->utab_dev, 0, 1, 0, prev, row0, rows);
    if (rc) return rc;
    c->raw_pending = true;
    return GAME_OK;
}

void* game_hip_raw_rows(game_hip_ctx* c, uint32_t row)
{
    if (!c || !c->raw || row >= c->h) return nullptr;
    return c->raw + (size_t)row * c->w;
}

int game_hip_blur_band(game_hip_ctx* c, uint32_t row0, uint32_t rows, uint8_t* rgba8_host)
{
    if (!c) return set_error(GAME_E_ARGUMENT, "game_hip_blur_band: null context");
    if (!c->blur || !c->raw_pending) return set_error(GAME_E_UNSUPPORTED, "game_hip_blur_band: no raw band is waiting (game_hip_render_raw_band)");
    if (rows == 0 || (uint64_t)row0 + rows > c->h) return set_error(GAME_E_ARGUMENT, "game_hip_blur_band: band outside the image");
    if (!c->frames || c->ring == 0) return set_error(GAME_E_HIP, "game_hip_blur_band: the frame ring is not allocated");
    HIP_TRY(hipSetDevice(c->device));
    const size_t frame_px = (size_t)c->w * c->h;
    const uint32_t slot = c->next_slot;
    int rc = launch_blur(c->raw, c->frames + (size_t)slot * frame_px, c->w, c->h, row0, rows, c->blur, c->stream);
    if (rc) return rc;
    c->raw_pending = false;
    c->last_slot = (int)slot;
    c->next_slot = (slot + 1) % c->ring;
    if (rgba8_host) {
        HIP_TRY(hipMemcpyAsync(rgba8_host, c->frames + (size_t)slot * frame_px + (size_t)row0 * c->w, (size_t)rows * c->w * 4,
                               hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return GAME_OK;
}

int game_hip_reset_memory(game_hip_ctx* c)
{
    if (!c) return set_error(GAME_E_ARGUMENT, "game_hip_reset_memory: null context");
    HIP_TRY(hipSetDevice(c->device));
    c->last_slot = -1;
    c->next_slot = 0;
    return GAME_OK;
}

int game_hip_set_memory(game_hip_ctx* c, const uint8_t* rgba8_frame, uint32_t row0, uint32_t rows)
{
    if (!c || !rgba8_frame) return set_error(GAME_E_ARGUMENT, "game_hip_set_memory: null argument");
    if (rows == 0 || (uint64_t)row0 + rows > c->h) return set_error(GAME_E_ARGUMENT, "game_hip_set_memory: band outside the image");
    if (!c->frames || c->ring == 0) return set_error(GAME_E_HIP, "game_hip_set_memory: the frame ring is not allocated");
    HIP_TRY(hipSetDevice(c->device));
    // the frame goes into the slot before the next one to be written, which the next frame reads as `previous`
    const uint32_t slot = c->last_slot >= 0 ? (uint32_t)c->last_slot : (c->next_slot + c->ring - 1u) % c->ring;
    if (c->last_slot < 0 && rows != c->h)   // rows outside the band must read as the zero frame a fresh context has
        HIP_TRY(hipMemsetAsync(c->frames + (size_t)slot * c->w * c->h, 0, (size_t)c->w * c->h * 4, c->stream));
    HIP_TRY(hipMemcpyAsync(c->frames + (size_t)slot * c->w * c->h + (size_t)row0 * c->w, rgba8_frame, (size_t)rows * c->w * 4,
                           hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // the caller's buffer may go away
    c->last_slot = (int)slot;
    c->next_slot = (slot + 1u) % c->ring;
    return GAME_OK;
}

int game_hip_particles_set_time(game_hip_ctx* c, float sim_time)
{
    if (!c) return set_error(GAME_E_ARGUMENT, "game_hip_particles_set_time: null context");
    c->sim_time = sim_time;
    return GAME_OK;
}

int game_hip_sync(game_hip_ctx* c)
{
    if (!c) return set_error(GAME_E_ARGUMENT, "game_hip_sync: null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GAME_OK;
}

void* game_hip_frame_ptr(game_hip_ctx* c, uint32_t slot)
{
    if (!c || slot >= c->ring) return nullptr;
    return c->frames + (size_t)slot * c->w * c->h;
}
uint32_t game_hip_ring_frames(const game_hip_ctx* c) { return c ? c->ring : 0; }

int game_hip_last_kernel_ms(game_hip_ctx* c, float* ms, uint32_t* launches)
{
    if (!c || !ms) return set_error(GAME_E_ARGUMENT, "game_hip_last_kernel_ms: null argument");
    HIP_TRY(hipSetDevice(c->device));
    // nothing timed yet (no call so far, the last one failed, or the last timed call's pair was handed back by
    // game_hip_timing_end): zero time over zero launches rather than an error from an unrecorded event
    const bool own = c->ev0 == c->ev_own0;
    if (own && !c->own_recorded) { *ms = 0.0f; if (launches) *launches = 0; return GAME_OK; }
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    if (launches) *launches = c->last_launches;
    return GAME_OK;
}

int game_hip_timing_begin(game_hip_ctx* c)
{
    if (!c) return set_error(GAME_E_ARGUMENT, "game_hip_timing_begin: null context");
    if (c->timing) return set_error(GAME_E_ARGUMENT, "game_hip_timing_begin: a timing section is already open");
    c->timing = true;
    return GAME_OK;
}

int game_hip_timing_end(game_hip_ctx* c, float* total_ms, uint32_t* launches)
{
    if (!c || !total_ms) return set_error(GAME_E_ARGUMENT, "game_hip_timing_end: null argument");
    if (!c->timing) return set_error(GAME_E_ARGUMENT, "game_hip_timing_end: no timing section is open");
    HIP_TRY(hipSetDevice(c->device));
    c->timing = false;
    c->ev0 = c->ev_own0; c->ev1 = c->ev_own1;
    double sum = 0.0;
    uint32_t n = 0;
    hipError_t err = hipSuccess;
    for (auto& t : c->ev_pending) {
        float ms = 0.0f;
        if (err == hipSuccess) err = hipEventSynchronize(t.e1);
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, t.e0, t.e1);
        sum += ms;
        n += t.launches;
        c->ev_pool.emplace_back(t.e0, t.e1);
    }
    c->ev_pending.clear();
    if (err != hipSuccess) return hip_fail("game_hip_timing_end", err);
    *total_ms = (float)sum;
    if (launches) *launches = n;
    return GAME_OK;
}

void* game_hip_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (bytes == 0) { set_error(GAME_E_ARGUMENT, "game_hip_host_alloc: zero bytes"); return nullptr; }
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { hip_fail("hipHostMalloc", e); return nullptr; }
    return p;
}

void game_hip_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

const char* game_hip_kernel_source(const game_hip_ctx* c) { return c ? c->source.c_str() : ""; }
const char* game_hip_gravity_source(const game_hip_ctx* c) { return c ? c->gsource.c_str() : ""; }
const char* game_hip_device_header_source(void) { return kDeviceHeader; }
void* game_hip_stream(game_hip_ctx* c) { return c ? (void*)c->stream : nullptr; }

int game_hip_setup_ms(const game_hip_ctx* c, float* compile_ms, float* tables_ms)
{
    if (!c) return set_error(GAME_E_ARGUMENT, "game_hip_setup_ms: null context");
    if (compile_ms) *compile_ms = c->compile_ms;
    if (tables_ms) *tables_ms = c->tables_ms;
    return GAME_OK;
}

// ── particles ────────────────────────────────────────────────────────────────────────────
static int alloc_particles(game_hip_ctx* c, uint32_t n)
{
    if (c->n == n && c->px[0]) return GAME_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n = 0;   // stays 0 if an allocation below fails, so a later call starts over
    for (int i = 0; i < 2; ++i) {
        if (c->px[i]) HIP_TRY(hipFree(c->px[i]));
        if (c->py[i]) HIP_TRY(hipFree(c->py[i]));
        c->px[i] = c->py[i] = nullptr;
    }
    if (c->vx) HIP_TRY(hipFree(c->vx));
    if (c->vy) HIP_TRY(hipFree(c->vy));
    c->vx = c->vy = nullptr;
    size_t bytes = (size_t)n * 4;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipMalloc((void**)&c->px[i], bytes));
        HIP_TRY(hipMalloc((void**)&c->py[i], bytes));
    }
    HIP_TRY(hipMalloc((void**)&c->vx, bytes));
    HIP_TRY(hipMalloc((void**)&c->vy, bytes));
    c->n = n;
    return GAME_OK;
}

int game_hip_particles_init(game_hip_ctx* c, uint32_t n, const float* pos_xy, const float* vel_xy)
{
    if (!c || !pos_xy || n == 0) return set_error(GAME_E_ARGUMENT, "game_hip_particles_init: bad argument");
    if (!c->ir.gravity.present) return set_error(GAME_E_UNSUPPORTED, "cinematic has no gravity block");
    HIP_TRY(hipSetDevice(c->device));
    int rc = alloc_particles(c, n);
    if (rc) return rc;
    std::vector<float> x(n), y(n), vx(n, 0.0f), vy(n, 0.0f);
    for (uint32_t i = 0; i < n; ++i) {
        x[i] = pos_xy[2 * i]; y[i] = pos_xy[2 * i + 1];
        if (vel_xy) { vx[i] = vel_xy[2 * i]; vy[i] = vel_xy[2 * i + 1]; }
    }
    size_t bytes = (size_t)n * 4;
    c->cur = 0;
    c->sim_time = 0.0f;
    // The uploads go through the context's own stream: it does not synchronise with the null stream, and steps queued
    // by an earlier game_hip_particles_step may still be reading and writing these arrays.
    HIP_TRY(hipMemcpyAsync(c->px[0], x.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->py[0], y.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->px[1], x.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->py[1], y.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->vx, vx.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->vy, vy.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // the host vectors go away
    return GAME_OK;
}

int game_hip_particles_seed(game_hip_ctx* c, uint32_t n, uint64_t seed)
{
    if (!c || n == 0) return set_error(GAME_E_ARGUMENT, "game_hip_particles_seed: bad argument");
    std::vector<float> pos((size_t)n * 2);
    uint64_t s = seed;
    for (size_t i = 0; i < (size_t)n * 2; ++i) {   // SplitMix64, top 24 bits -> [-1, 1)
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        pos[i] = (float)(z >> 40) * 0x1p-23f - 1.0f;
    }
    return game_hip_particles_init(c, n, pos.data(), nullptr);
}

int game_hip_particles_step(game_hip_ctx* c, float dt, uint32_t steps, uint32_t first, uint32_t count)
{
    if (!c) return set_error(GAME_E_ARGUMENT, "game_hip_particles_step: null context");
    if (!c->gfn || !c->px[0]) return set_error(GAME_E_UNSUPPORTED, "particles not initialised (or no gravity block)");
    if (count == 0 || (uint64_t)first + count > c->n) return set_error(GAME_E_ARGUMENT, "game_hip_particles_step: range outside particle set");
    if (count < c->n && steps > 1)
        return set_error(GAME_E_ARGUMENT, "game_hip_particles_step: a partial range advances one step per call (exchange positions between steps)");
    HIP_TRY(hipSetDevice(c->device));
    // Particles per lane: one. The two-particle build halves the LDS reads per pair, but the loop is bound by VALU
    // issue and it measured no faster; it stays selectable for measurements.
    uint32_t ppl = 1u;
    if (const char* e = getenv("GAME_HIP_GRAV_PPL")) {   // tuning hook
        int v = atoi(e);
        if (v == 1 || v == 2) ppl = (uint32_t)v;
    }
    hipFunction_t gfn = ppl == 2 ? c->gfn2 : c->gfn;
    const uint32_t span = 256u * ppl;
    c->last_launches = 0;
    { int erc = begin_call_events(c); if (erc) return erc; }
    CallEvents events(c);
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    for (uint32_t s = 0; s < steps; ++s) {
        c->sim_time = c->sim_time + dt;   // `_time += dt` before the dispatch (gravity.rs:158)
        GravityParams P;
        P.xin = c->px[c->cur]; P.yin = c->py[c->cur];
        P.xout = c->px[1 - c->cur]; P.yout = c->py[1 - c->cur];
        P.vx = c->vx; P.vy = c->vy;
        P.n = c->n; P.first = first; P.count = count; P.floor_bits = 0x358637bdu;
        P.dt = dt; P.time = c->sim_time;
        // One launch per 2^20 particles of the range: at 2^24 particles a whole step is about 80 s of GPU time, and a
        // launch should stay in the seconds. The launches of a step read the same positions and write disjoint ranges.
        for (uint32_t off = 0; off < count; off += kGravityLaunchParticles) {
            P.first = first + off;
            P.count = count - off < kGravityLaunchParticles ? count - off : kGravityLaunchParticles;
            void* args[] = {&P};
            HIP_TRY(hipModuleLaunchKernel(gfn, (P.count + span - 1u) / span, 1, 1, 256, 1, 1, 0, c->stream, args, nullptr));
            ++c->last_launches;
        }
        c->cur = 1 - c->cur;
    }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    events.finish();
    return GAME_OK;
}

int game_hip_particles_read(game_hip_ctx* c, float* pos_xy, float* vel_xy)
{
    if (!c || !c->px[0]) return set_error(GAME_E_ARGUMENT, "game_hip_particles_read: particles not initialised");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<float> a(c->n), b(c->n);
    size_t bytes = (size_t)c->n * 4;
    if (pos_xy) {
        HIP_TRY(hipMemcpy(a.data(), c->px[c->cur], bytes, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(b.data(), c->py[c->cur], bytes, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < c->n; ++i) { pos_xy[2 * i] = a[i]; pos_xy[2 * i + 1] = b[i]; }
    }
    if (vel_xy) {
        HIP_TRY(hipMemcpy(a.data(), c->vx, bytes, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(b.data(), c->vy, bytes, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < c->n; ++i) { vel_xy[2 * i] = a[i]; vel_xy[2 * i + 1] = b[i]; }
    }
    return GAME_OK;
}

int game_hip_particles_positions(game_hip_ctx* c, float** x_dev, float** y_dev)
{
    if (!c || !c->px[0] || !x_dev || !y_dev) return set_error(GAME_E_ARGUMENT, "game_hip_particles_positions: bad argument");
    *x_dev = c->px[c->cur];
    *y_dev = c->py[c->cur];
    return GAME_OK;
}

int game_hip_particles_time(game_hip_ctx* c, float* sim_time)
{
    if (!c || !sim_time) return set_error(GAME_E_ARGUMENT, "game_hip_particles_time: bad argument");
    *sim_time = c->sim_time;
    return GAME_OK;
}

}  // extern "C"
