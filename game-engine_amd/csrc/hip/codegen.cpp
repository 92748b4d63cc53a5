// Kernel generator: flat IR -> HIP source of one fused per-pixel kernel (and, for cinematics
// with a gravity block, one all-pairs particle kernel). This is the back end that takes the place
// of the reference's GLSL/WGSL emitters (game-compiler/src/codegen/glsl.rs:20-80, :228-259 for
// the fragment program; game-compiler/src/codegen/gravity.rs:43-120 for the compute program).
// The generated text only sequences calls into game_device.h with the stage arguments baked in.
#include "codegen.hpp"
#include <cstdio>
#include <cmath>
#include <cstring>

namespace game {

std::string float_literal(float v)
{
    if (v != v) return "__builtin_nanf(\"\")";
    if (v == INFINITY) return "__builtin_inff()";
    if (v == -INFINITY) return "(-__builtin_inff())";
    char buf[64];
    snprintf(buf, sizeof buf, "%af", (double)v);   // hex float: exact
    return buf;
}

namespace {

// A cinematic's name inside a `//` comment of the generated source: a line break would end the comment and a trailing
// backslash would splice the next line into it, so anything but plain printable characters becomes '?'.
std::string comment_safe(const std::string& name)
{
    std::string o = name;
    for (char& c : o) if ((unsigned char)c < 0x20 || (unsigned char)c > 0x7e || c == '\\') c = '?';
    return o;
}

// Reverse Polish program -> one HIP expression over the device functions (f32, one rounding per operator).
// gravity = true: the force law of a gravity block (variables `dist` and the simulation time, gravity.rs:10-41);
// otherwise a stage-argument expression that reads the pixel (`p`, `uv`) and the frame row.
std::string rpn_expr(const std::vector<IrLawOp>& prog, bool gravity)
{
    static const char* fn1[] = {"gd_sin", "gd_cos", "gd_tan", "gd_abs", "gd_sqrt", "gd_exp", "gd_log", "gd_floor", "gd_ceil",
                                "gd_fract", "gd_sign", "", "", "", "", "", "", "", "gd_exp2", "gd_log2", "gd_round"};
    std::vector<std::string> st;
    if (prog.empty()) return "1.0f";
    for (const IrLawOp& o : prog) {
        auto pop = [&]() { std::string t = st.back(); st.pop_back(); return t; };
        switch (o.op) {
        case LAW_NUM: st.push_back(float_literal(o.val)); break;
        case LAW_DIST: st.push_back("dist"); break;
        case LAW_TIME: st.push_back("P.time"); break;
        case LAW_UNIFORM: st.push_back(!gravity && o.fn == BU_TIME ? std::string("s.time") : "U[" + std::to_string(o.fn) + "]"); break;
        case LAW_PX: st.push_back("s.px"); break;
        case LAW_PY: st.push_back("s.py"); break;
        case LAW_UVX: st.push_back("s.uvx"); break;
        case LAW_UVY: st.push_back("s.uvy"); break;
        case LAW_ADD: { auto r = pop(), l = pop(); st.push_back("(" + l + " + " + r + ")"); break; }
        case LAW_SUB: { auto r = pop(), l = pop(); st.push_back("(" + l + " - " + r + ")"); break; }
        case LAW_MUL: { auto r = pop(), l = pop(); st.push_back("(" + l + " * " + r + ")"); break; }
        case LAW_DIV: { auto r = pop(), l = pop(); st.push_back("(" + l + " / " + r + ")"); break; }
        case LAW_POW: { auto r = pop(), l = pop(); st.push_back("gd_pow(" + l + ", " + r + ")"); break; }
        case LAW_GT: { auto r = pop(), l = pop(); st.push_back("((" + l + " > " + r + ") ? 1.0f : 0.0f)"); break; }
        case LAW_LT: { auto r = pop(), l = pop(); st.push_back("((" + l + " < " + r + ") ? 1.0f : 0.0f)"); break; }
        case LAW_NEG: { auto v = pop(); st.push_back("(-" + v + ")"); break; }
        case LAW_CALL1: { auto v = pop(); st.push_back(std::string(fn1[o.fn]) + "(" + v + ")"); break; }
        case LAW_CALL2: {
            auto b = pop(), a = pop();
            const char* f = o.fn == LF_MIN ? "gd_min" : o.fn == LF_MAX ? "gd_max" : o.fn == LF_POW ? "gd_pow" : o.fn == LF_MOD ? "gw_rem" :
                            o.fn == LF_ATAN2 ? "gd_atan2" : o.fn == LF_LENGTH2 ? "gd_length2" : "gd_step";
            st.push_back(std::string(f) + "(" + a + ", " + b + ")");
            break;
        }
        case LAW_CALL3: {
            auto c = pop(), b = pop(), a = pop();
            const char* f = o.fn == LF_CLAMP ? "gd_clamp" : o.fn == LF_MIX ? "gd_mix" : o.fn == LF_SELECT ? "gd_select" : "gd_smoothstep";
            st.push_back(std::string(f) + "(" + a + ", " + b + ", " + c + ")");
            break;
        }
        default: break;
        }
    }
    return st.empty() ? "1.0f" : st.back();
}

// Frame-row layout used by the kernels: the caller's 10 + n floats, then two derived values the
// host computes once per frame (glsl.rs:55-56), then (animate shapes only) one float per derived value.
std::string arg_expr(const IrArg& a, const KernelShape& shape, const Ir& ir)
{
    if (shape.animate && a.pixel >= 0) return rpn_expr(ir.pixel_exprs[(size_t)a.pixel].prog, false);
    if (shape.animate && a.derived >= 0) return "U[" + std::to_string(shape.derived_slot + (uint32_t)a.derived) + "]";
    switch (a.kind) {
    case ARG_LIT: return float_literal(a.lit);
    case ARG_USER: return "U[" + std::to_string(BU_COUNT + a.slot) + "]";
    default: return a.slot == BU_TIME ? std::string("s.time") : "U[" + std::to_string(a.slot) + "]";
    }
}

int octaves_of(float o)
{
    if (!(o >= 1.0f)) return 0;
    if (o >= 32.0f) return 32;
    return (int)o;
}

const float PALETTES[8][12] = {   // glsl.rs:424-434
    {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 1.0f, 1.0f, 1.0f, 0.00f, 0.10f, 0.20f},
    {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 1.0f, 1.0f, 1.0f, 0.30f, 0.20f, 0.20f},
    {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 1.0f, 1.0f, 1.0f, 0.00f, 0.33f, 0.67f},
    {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 1.0f, 0.7f, 0.4f, 0.00f, 0.15f, 0.20f},
    {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 1.0f, 1.0f, 0.5f, 0.80f, 0.90f, 0.30f},
    {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 2.0f, 1.0f, 0.0f, 0.50f, 0.20f, 0.25f},
    {0.8f, 0.5f, 0.4f, 0.2f, 0.4f, 0.2f, 2.0f, 1.0f, 1.0f, 0.00f, 0.25f, 0.25f},
    {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 1.0f, 1.0f, 1.0f, 0.00f, 0.10f, 0.20f},
};

// Builtins the WebGPU emitter words differently (wgsl.rs:268-540); they have a gw_* device function.
bool has_wgsl_variant(uint16_t b)
{
    switch (b) {
    case B_STAR: case B_POLYGON: case B_SIMPLEX: case B_VORONOI: case B_CONCENTRIC_WAVES: case B_THRESHOLD: case B_VIGNETTE:
    case B_CHROMATIC: case B_TONEMAP: case B_SCANLINES: case B_GLITCH: case B_MIRROR: case B_REPEAT: case B_CURL_NOISE:
    case B_DISPLACE: case B_SPECTRUM:
        return true;
    default:
        return false;
    }
}

std::string stage_call(const IrStage& st, const KernelShape& shape, const Ir& ir)
{
    const bool wgsl = shape.wgsl;
    auto A = [&](int i) { return arg_expr(st.args[i], shape, ir); };
    auto call = [&](const char* fn, int n) {
        std::string name(fn);
        if (wgsl && has_wgsl_variant(st.builtin)) name[1] = 'w';   // gb_xxx -> gw_xxx
        std::string c = name + "(s";
        for (int i = 0; i < n; ++i) c += ", " + A(i);
        return c + ");";
    };
    switch (st.builtin) {
    case B_CIRCLE: return call("gb_circle", 1);
    case B_RING: return call("gb_ring", 2);
    case B_STAR: return call("gb_star", 3);
    case B_BOX: return call("gb_box", 2);
    case B_POLYGON: return call("gb_polygon", 2);
    case B_SIMPLEX: return call("gb_simplex", 1);
    case B_VORONOI: return call("gb_voronoi", 1);
    case B_CONCENTRIC_WAVES: return call("gb_concentric_waves", 3);
    case B_FBM: {
        std::string oct = st.args[1].kind == ARG_LIT ? std::to_string(octaves_of(st.args[1].lit)) : "gd_octaves(" + A(1) + ")";
        return "gb_fbm(s, " + A(0) + ", " + oct + ", " + A(2) + ", " + A(3) + ");";
    }
    case B_ROTATE: return call("gb_rotate", 1);
    case B_TRANSLATE: return call("gb_translate", 2);
    case B_SCALE: return call("gb_scale", 1);
    case B_TWIST: return call("gb_twist", 1);
    case B_MIRROR: return call("gb_mirror", 1);
    case B_REPEAT: return call("gb_repeat", 1);
    case B_DOMAIN_WARP: return call("gb_domain_warp", 2);
    case B_CURL_NOISE: return call("gb_curl_noise", 2);
    case B_DISPLACE: return call("gb_displace", 1);
    case B_MASK_ARC: return call("gb_mask_arc", 1);
    case B_THRESHOLD: return call("gb_threshold", 1);
    case B_ONION: return call("gb_onion", 1);
    case B_ROUND: return call("gb_round", 1);
    case B_GLOW: return call("gb_glow", 1);
    case B_SHADE: return call("gb_shade", 3);
    case B_EMISSIVE: return call("gb_emissive", 1);
    case B_PALETTE: {
        const float* P = PALETTES[st.aux < 8 ? st.aux : 2];
        std::string c = "gb_palette(s";
        for (int i = 0; i < 12; ++i) c += ", " + float_literal(P[i]);
        return c + ");";
    }
    case B_TINT: return call("gb_tint", 3);
    case B_BLOOM: return call("gb_bloom", 2);
    case B_GRAIN: return call("gb_grain", 1);
    case B_VIGNETTE: return call("gb_vignette", 2);
    case B_CHROMATIC: return call("gb_chromatic", 1);
    case B_TONEMAP: return call("gb_tonemap", 1);
    case B_SCANLINES: return call("gb_scanlines", 2);
    case B_SATURATE_COLOR: return call("gb_saturate_color", 1);
    case B_GLITCH: return call("gb_glitch", 1);
    case B_BLEND: return call("gb_blend", 1);
    case B_GRADIENT: return "gb_gradient(s);";
    case B_SPECTRUM: return call("gb_spectrum", 3);
    default: return "/* unknown builtin */";
    }
}

}  // namespace

std::string generate_frame_kernel(const Ir& ir, const KernelShape& shape)
{
    std::string k;
    k.reserve(8192);
    k += "// generated for cinematic \"" + comment_safe(ir.name) + "\"\n";
    k += "#include \"game_device.h\"\n\n";
    k += "struct GameFrameParams {\n"
         "    gd_u32* out;\n"
         "    const gd_u32* prev;\n"
         "    const float* utab;\n"
         "    unsigned long long frame_pixels;\n"
         "    gd_u32 w, h, row0, rows;\n"
         "    gd_u32 ustride, slot0, ring, frame0;\n"
         "    float inv_w, inv_h;\n"
         "    const float* lut;\n"
         "    const float* vlut;\n"
         "    const float* slut;\n"
         "};\n\n";
    const std::string wg = std::to_string(shape.wg);
    k += "extern \"C\" __global__ __launch_bounds__(" + wg + ") void game_frame(GameFrameParams P)\n{\n";
    // A lane owns pixels_per_lane consecutive pixels of one row, a wave 64 * pixels_per_lane of them,
    // a 256-thread workgroup four such strips side by side: every store instruction of a wave writes
    // one contiguous run of the row (256 B, or 1 KiB as 16-byte stores when pixels_per_lane is 4).
    // Shapes with 8 or 16 pixels per lane give a lane two or four such 16-byte vectors one workgroup strip (4 x workgroup size pixels) apart (every
    // load and store instruction of a wave still covers one contiguous KiB) and issue all previous-frame loads
    // before any arithmetic: more bytes in flight per wave for the cinematics the memory system bounds.
    const uint32_t nvec = shape.pixels_per_lane >= 4 ? shape.pixels_per_lane / 4u : 0u;
    // how the previous frame is read: every byte once, so for frames that cannot stay in cache from one launch to the next
    // the load says so (non-temporal: it does not displace what the cache holds)
    const std::string ld4 = shape.stream_prev ? "__builtin_nontemporal_load((const gd_u32x4*)" : "*((const gd_u32x4*)";
    const std::string ld1 = shape.stream_prev ? "__builtin_nontemporal_load((const gd_u32*)" : "*((const gd_u32*)";
    const std::string ppl = std::to_string(shape.pixels_per_lane);
    if (nvec > 1) k += "    const gd_u32 x0 = blockIdx.x * " + std::to_string(shape.wg * shape.pixels_per_lane) + "u + threadIdx.x * 4u;\n";
    else k += "    const gd_u32 x0 = (blockIdx.x * " + wg + "u + threadIdx.x) * " + ppl + "u;\n";
    // Several rows per lane: what a row needs from y alone (uv.y, and the y halves of a leading domain_warp's lattice
    // samples) is the same for every lane of the workgroup, so lane r of the workgroup works it out for row r once, into
    // shared memory, and the row loop reads it from there. The early return of lanes beyond the right edge then comes
    // after the barrier.
    const bool project_warp = shape.project && ir.project.present && (ir.project.mode == ProjectMode::Dome || ir.project.mode == ProjectMode::Led);
    const bool row_lds = shape.row_shared && shape.rows_per_lane > 1 && !project_warp && shape.wg >= shape.rows_per_lane;
    if (!row_lds) k += "    if (x0 >= P.w) return;\n";
    // The uniform row is read through the constant address space: nothing writes it while the kernel runs, so its
    // loads are scalar and invariant. Every slot the pipeline reads is loaded here, ahead of the row loop, so that
    // whatever is computed from uniforms and the column alone leaves the loop with them.
    k += "    const gd_uniform_row U = (gd_uniform_row)(P.utab + (unsigned long long)(P.frame0 + blockIdx.z) * P.ustride);\n";
    const size_t uniform_loads_at = k.size();
    // A lane walks rows_per_lane rows of its columns, one after the other: whatever a pipeline computes from the column
    // alone (v_uv.x, uv.x, p.x, a spectrum, a bloom on it ...) does not change from row to row and leaves the loop.
    const std::string rpl = std::to_string(shape.rows_per_lane);
    if (shape.rows_per_lane > 1) {
        k += "#pragma nounroll\n    for (gd_u32 ry = 0; ry < " + rpl + "u; ++ry) {\n";
        k += "    const gd_u32 ty = P.row0 + blockIdx.y * " + rpl + "u + ry;\n";
        k += "    if (ty >= P.row0 + P.rows) break;\n";
    } else {
        k += "    {\n    const gd_u32 ty = P.row0 + blockIdx.y;\n";
    }
    k += "    const gd_u32 fidx = blockIdx.z;\n";
    // slot0 < ring and fidx < ring (checked by the host), so one conditional subtract wraps the ring
    k += "    const gd_u32 slot_raw = P.slot0 + fidx;\n";
    k += "    const gd_u32 slot = slot_raw >= P.ring ? slot_raw - P.ring : slot_raw;\n";
    k += "    gd_u32* __restrict__ dst = P.out + (unsigned long long)slot * P.frame_pixels + (unsigned long long)ty * P.w;\n";
    k += "    GdPix s;\n";
    k += "    s.lut = P.lut; s.vlut = P.vlut; s.slut = P.slut;\n";
    // Speculating (see GdGuard): the four pixels of a lane run with every range test recorded instead of branched on, then
    // one test per wave; only when some lane left a range do the four pixels run again with the tests taken.
    const bool speculate = shape.speculate && nvec == 1;
    k += std::string("    s.guard.speculate = ") + (speculate ? "true" : "false") + "; s.guard.left_range = false;\n";
    k += "    s.time = U[" + std::to_string(shape.time_slot) + "];\n";
    k += "    const float aspect = U[" + std::to_string(shape.aspect_slot) + "];\n";
    k += "    s.bass = U[1]; s.mid = U[2]; s.treble = U[3];\n";
    if (row_lds) {
        k += "    s.uvy = gd_rowv[ry * GD_ROWV];\n";
    } else {
        k += "    const float vy = ((float)(P.h - 1u - ty) + 0.5f) * P.inv_h;\n";   // GL row 0 is the bottom row
        k += "    s.uvy = vy * 2.0f - 1.0f;\n";
    }
    size_t pixel_loop_begin = 0;
    if (nvec > 1) {
        const std::string nv = std::to_string(nvec);
        if (ir.uses_memory) {
            k += "    gd_u32x4 pv[" + nv + "];\n";
            k += "#pragma unroll\n    for (gd_u32 v = 0; v < " + nv + "u; ++v) {\n"
                 "        pv[v] = gd_u32x4{0u, 0u, 0u, 0u};\n"
                 "        if (P.prev && x0 + v * " + std::to_string(shape.wg * 4u) + "u < P.w) pv[v] = " + ld4 + "(P.prev + (unsigned long long)ty * P.w + x0 + v * " + std::to_string(shape.wg * 4u) + "u));\n"
                 "    }\n";
        }
        k += "#pragma unroll\n    for (gd_u32 v = 0; v < " + nv + "u; ++v) {\n";
        k += "    const gd_u32 xv = x0 + v * " + std::to_string(shape.wg * 4u) + "u;\n";
        k += "    if (xv >= P.w) break;\n";
        k += "    gd_u32x4 q;\n";
        k += "#pragma unroll\n    for (gd_u32 kx = 0; kx < 4u; ++kx) {\n    const gd_u32 x = xv + kx;\n";
        k += "    {\n";
        if (ir.uses_memory) k += "    const gd_u32 prev = pv[v][kx];\n";
    } else if (nvec == 1) {
        // the host picks this shape only when the width is a multiple of 4: rows stay 16-byte aligned
        if (ir.uses_memory)
            k += "    gd_u32x4 pv = {0u, 0u, 0u, 0u};\n"
                 "    if (P.prev) pv = " + ld4 + "(P.prev + (unsigned long long)ty * P.w + x0));\n";
        k += "    gd_u32x4 q;\n";
        pixel_loop_begin = k.size();
        k += "#pragma unroll\n    for (gd_u32 kx = 0; kx < 4u; ++kx) {\n    const gd_u32 x = x0 + kx;\n";
        k += "    {\n";
        if (ir.uses_memory) k += "    const gd_u32 prev = pv[kx];\n";
    } else {
        k += "    {\n    const gd_u32 x = x0;\n    {\n";
        if (ir.uses_memory)
            k += "    const gd_u32 prev = P.prev ? " + ld1 + "(P.prev + (unsigned long long)ty * P.w + x)) : 0u;\n";
    }
    k += "    const float vx = ((float)x + 0.5f) * P.inv_w;\n";
    const bool warp = shape.project && ir.project.present && (ir.project.mode == ProjectMode::Dome || ir.project.mode == ProjectMode::Led);
    if (warp) {
        // projection block: the fragment starts from a warped v_uv (project.rs:62-69 dome, :128-137 led)
        k += "    float wvx, wvy;\n";
        if (ir.project.mode == ProjectMode::Dome)
            k += "    gd_project_dome(vx, vy, " + float_literal(ir.project.fov_deg * 3.14159265359f / 180.0f) + ", wvx, wvy);\n";
        else
            k += "    gd_project_led(vx, " + float_literal((float)ir.project.count) + ", wvx, wvy);\n";
        k += "    s.uvx = wvx * 2.0f - 1.0f;\n    s.uvy = wvy * 2.0f - 1.0f;\n";
    } else
    k += "    s.uvx = vx * 2.0f - 1.0f;\n";
    bool reads_uv_length = false;
    for (const IrLayer& L : ir.layers) {
        if (!L.is_pipeline || L.end_state != State::Color || (L.post && !shape.lens_post)) continue;
        for (uint32_t si = 0; si < L.stage_count; ++si) {
            const uint16_t b = ir.stages[L.stage_begin + si].builtin;
            reads_uv_length = reads_uv_length || b == B_VIGNETTE || b == B_CHROMATIC;
        }
    }
    if (reads_uv_length) k += "    gd_uv_length(s);\n";   // once per pixel, shared by every stage that reads length(uv)
    k += "    float fr = 0.0f, fg = 0.0f, fb = 0.0f, fa = 1.0f;\n";
    std::string row_state;   // declarations that live across the rows of a lane, placed ahead of the row loop
    std::string row_prologue;   // what lane r of the workgroup stores for row r besides uv.y
    uint32_t row_floats = 4;     // uv.y and padding first
    uint32_t row_regs = 0;       // registers a lane holds across its rows for the cached stages
    for (size_t li = 0; li < ir.layers.size(); ++li) {
        const IrLayer& L = ir.layers[li];
        if (!L.is_pipeline || L.end_state != State::Color) continue;
        if (L.post) {
            // the lens post chain (ignored by the reference's emitters): Color -> Color stages on the composite
            if (!shape.lens_post) continue;
            k += "    {   // lens post\n";
            k += "        s.px = s.uvx * aspect; s.py = s.uvy; s.sdf = 0.0f; s.r = fr; s.g = fg; s.b = fb; s.a = fa;\n";
            for (uint32_t si = 0; si < L.stage_count; ++si)
                k += "        " + stage_call(ir.stages[L.stage_begin + si], shape, ir) + "\n";
            k += "        fr = s.r; fg = s.g; fb = s.b; fa = s.a;\n";
            k += "    }\n";
            continue;
        }
        k += "    {   // layer " + std::to_string(li) + "\n";
        k += "        s.px = s.uvx * aspect; s.py = s.uvy; s.sdf = 0.0f; s.r = s.g = s.b = s.a = 0.0f;\n";
        for (uint32_t si = 0; si < L.stage_count; ++si) {
            const IrStage& st = ir.stages[L.stage_begin + si];
            std::string text = stage_call(st, shape, ir);
            // A domain_warp that opens the layer warps the pixel's own p, whose x half does not change along the lane's rows:
            // it keeps its x-interpolated lattice values from row to row (gb_domain_warp_rows). Not when an argument reads
            // the pixel, and not under a projection, which moves uv.x from row to row.
            bool reads_pixel = false;
            for (uint32_t ai = 0; ai < 4; ++ai) reads_pixel = reads_pixel || (shape.animate && st.args[ai].pixel >= 0);
            // kind of cache, floats it takes in the workgroup's row values, registers it holds, the argument its y half is
            // scaled by, and the call with the cache and the row values appended
            const char* kind = nullptr; const char* type = nullptr; uint32_t floats = 0, regs = 0; int yarg = 0;
            auto A = [&](int i) { return arg_expr(st.args[i], shape, ir); };
            std::string head;
            if (si == 0 && row_lds && shape.pixels_per_lane == 1 && !reads_pixel) {
                if (st.builtin == B_DOMAIN_WARP) {
                    kind = "warp"; type = "GdWarpRows"; floats = 8; regs = 6; yarg = 1;
                    head = "gb_domain_warp_rows(s, " + A(0) + ", " + A(1);
                } else if (st.builtin == B_VORONOI && !shape.wgsl) {
                    kind = "voronoi"; type = "GdVoronoiRows"; floats = 4; regs = 19; yarg = 0;
                    head = "gb_voronoi_rows(s, " + A(0);
                } else if (st.builtin == B_FBM && st.args[1].kind == ARG_LIT && octaves_of(st.args[1].lit) >= 1 && octaves_of(st.args[1].lit) <= 8) {
                    kind = "fbm"; type = "GdFbmRows"; floats = 4; regs = 3; yarg = 0;
                    head = "gb_fbm_rows(s, " + A(0) + ", " + std::to_string(octaves_of(st.args[1].lit)) + ", " + A(2) + ", " + A(3);
                }
            }
            // a cinematic of many layers keeps this state for the first few only: 48 registers and 44 row values at most
            if (kind && (row_regs + regs > 48u || row_floats + floats > 48u)) kind = nullptr;
            if (kind) {
                row_regs += regs;
                const std::string cache = std::string(kind) + "_rows_" + std::to_string(li);
                std::string scale = A(yarg);
                if (scale == "s.time") scale = "U[" + std::to_string(shape.time_slot) + "]";          // ahead of the loop there is no `s` yet
                const std::string at = std::to_string(row_floats);
                const std::string rowp = "&gd_rowv[ry * GD_ROWV + " + at + "]";
                text = head + ", " + cache + ", " + (st.builtin == B_DOMAIN_WARP ? "*(const GdWarpY*)" + rowp : rowp) + ");";
                row_state += "    " + std::string(type) + " " + cache + "; gd_" + kind + "_rows_reset(" + cache + ");\n";
                row_prologue += "        gd_" + std::string(kind) + "_rows_y(uvy, " + scale + ", &gd_rowv[threadIdx.x * GD_ROWV + " + at + "]);\n";
                row_floats += floats;
            }
            k += "        " + text + "\n";
        }
        if (L.has_memory) k += "        gb_memory_mix(s, prev, " + float_literal(L.decay) + ");\n";
        if (ir.multi_layer) {
            k += "        fr = fr + s.r - fr * s.r; fg = fg + s.g - fg * s.g; fb = fb + s.b - fb * s.b; fa = 1.0f;\n";
        } else {
            k += "        fr = s.r; fg = s.g; fb = s.b; fa = s.a;\n";
        }
        k += "    }\n";
    }
    if (nvec > 1) {
        k += "    q[kx] = gd_pack_rgba8(fr, fg, fb, fa);\n";
        k += "    }\n    }\n";
        k += "    *(gd_u32x4*)(dst + xv) = q;\n";
        k += "    }\n";
    } else if (nvec == 1) {
        k += "    q[kx] = gd_pack_rgba8(fr, fg, fb, fa);\n";
        k += "    }\n    }\n";
        if (speculate) {
            const std::string loop = k.substr(pixel_loop_begin);
            k += "    if (__builtin_expect(__builtin_amdgcn_ballot_w64(s.guard.left_range) != 0ull, 0)) {   // rare: some lane left a fast range\n";
            k += "    s.guard.speculate = false;\n";
            k += loop;
            k += "    s.guard.speculate = true; s.guard.left_range = false;\n";
            k += "    }\n";
        }
        k += "    *(gd_u32x4*)(dst + x0) = q;\n";
    } else {
        k += "    dst[x] = gd_pack_rgba8(fr, fg, fb, fa);\n";
        k += "    }\n    }\n";
    }
    k += "    }\n";   // rows of this lane
    k += "}\n";
    // U[n] in the body becomes the scalar u_n loaded once at the top
    std::string loads;
    std::vector<bool> seen;
    for (size_t at = k.find("U[", uniform_loads_at); at != std::string::npos; at = k.find("U[", at)) {
        size_t end = at + 2;
        unsigned n = 0;
        while (end < k.size() && k[end] >= '0' && k[end] <= '9') n = n * 10u + (unsigned)(k[end++] - '0');
        if (end == at + 2 || end >= k.size() || k[end] != ']' || (at > 0 && (isalnum((unsigned char)k[at - 1]) || k[at - 1] == '_'))) { at = end; continue; }
        const std::string name = "u_" + std::to_string(n);
        k.replace(at, end + 1 - at, name);
        if (seen.size() <= n) seen.resize(n + 1, false);
        if (!seen[n]) { seen[n] = true; loads += "    const float " + name + " = U[" + std::to_string(n) + "];\n"; }
        at += name.size();
    }
    std::string shared;
    if (row_lds) {
        const std::string rowv = std::to_string(row_floats);
        shared += "#define GD_ROWV " + rowv + "u\n";
        shared += "    __shared__ __attribute__((aligned(16))) float gd_rowv[" + rpl + " * GD_ROWV];\n";
        shared += "    if (threadIdx.x < " + rpl + "u) {\n";
        shared += "        const gd_u32 ty = P.row0 + blockIdx.y * " + rpl + "u + threadIdx.x;\n";
        shared += "        const float vy = ((float)(P.h - 1u - ty) + 0.5f) * P.inv_h;\n";   // GL row 0 is the bottom row
        shared += "        const float uvy = vy * 2.0f - 1.0f;\n";
        shared += "        gd_rowv[threadIdx.x * GD_ROWV] = uvy;\n";
        shared += row_prologue;
        shared += "    }\n    __syncthreads();\n    if (x0 >= P.w) return;\n";
    }
    k.insert(uniform_loads_at, loads + shared + row_state);
    return k;
}

// ── gravity ──────────────────────────────────────────────────────────────────────────────
namespace {

}  // namespace

// One gravity kernel: `ppl` particles per lane (register tiling: every LDS read of a position feeds ppl pair
// evaluations), exact or fast arithmetic. Every particle still sums its pairs in index order j = 0..n-1, so the result
// does not depend on ppl.
static std::string gravity_kernel_body(const Ir& ir, const char* name, unsigned ppl, bool fast)
{
    std::string k;
    const std::string P_ = std::to_string(ppl);
    const std::string span = std::to_string(256u * ppl) + "u";
    k += "extern \"C\" __global__ __launch_bounds__(256) void " + std::string(name) + "(GameGravityParams P)\n{\n";
    k += "    __shared__ float tx[256];\n    __shared__ float ty[256];\n";
    k += "    const gd_u32 wg_local = blockIdx.x * " + span + ";\n";
    k += "    gd_u32 idx[" + P_ + "]; bool live[" + P_ + "]; float sx[" + P_ + "], sy[" + P_ + "], fx[" + P_ + "], fy[" + P_ + "];\n";
    k += "    bool own_wild = false;\n";
    k += "#pragma unroll\n    for (gd_u32 q = 0; q < " + P_ + "u; ++q) {\n";
    k += "        const gd_u32 local = wg_local + q * 256u + threadIdx.x;\n";
    k += "        live[q] = local < P.count;\n";
    k += "        idx[q] = P.first + (live[q] ? local : 0u);\n";
    k += "        sx[q] = P.xin[idx[q]]; sy[q] = P.yin[idx[q]];\n";
    k += "        fx[q] = 0.0f; fy[q] = 0.0f;\n";
    k += "        own_wild = own_wild || !(gd_abs(sx[q]) <= 0x1p60f && gd_abs(sy[q]) <= 0x1p60f);\n";
    k += "    }\n";
    // Inner loop over one LDS tile, in three builds. dist = max(length(diff), 0.001) is computed as
    // sqrt(max(d2, c)) with c = 0x1.0c6f7ap-20, the smallest float whose correctly rounded root is
    // 0.001f (sqrt is monotonic, so both forms agree for every d2, NaN included), and 1 / dist as
    // inversesqrt(max(d2, c)), rounded once from the exact value (gd_rsqrt_core: the same float on a CPU); the short
    // exact sequences apply because c <= max(d2, c) < inf whenever d2 is not +inf. A force law that does not read
    // `dist` (the default rule 1.0) never computes the root.
    //   check_inf : some position of the tile or of this workgroup is NaN or beyond 2^60, so d2 may be
    //               +inf: keep the IEEE results for that case (dist = inf, 1 / dist = 0)
    //   check_self: the tile overlaps this workgroup's own particles (the j == idx term is skipped)
    // The fast build takes root and reciprocal root from one bare v_rsq_f32 (about 1 ulp each instead of correctly rounded).
    auto tile_loop = [&](bool check_self, bool check_inf) {
        std::string b;
        b += "#pragma unroll GD_GRAV_UNROLL\n";
        b += "            for (gd_u32 t = 0; t < lim; ++t) {\n";
        b += "                const float jx = tx[t], jy = ty[t];\n";
        b += "#pragma unroll\n                for (gd_u32 q = 0; q < " + P_ + "u; ++q) {\n";
        b += "                const float dx = jx - sx[q], dy = jy - sy[q];\n";
        b += "                const float d2 = __builtin_fmaf(dx, dx, dy * dy);\n";
        b += "                float root, rinv;\n";
        if (fast) {
            b += "                const float d2c = gd_max(d2, d2_floor);\n";
            b += "                rinv = __builtin_amdgcn_rsqf(d2c); root = d2c * rinv;\n";
        } else {
            b += "                gd_sqrt_rsqrt_core(gd_max(d2, d2_floor), root, rinv);\n";
        }
        if (check_inf) {
            b += "                const bool finite = d2 < __builtin_inff() || d2 != d2;\n";
            b += "                const float dist = finite ? root : d2;\n";
            b += "                const float inv = finite ? rinv : 0.0f;\n";
        } else {
            b += "                const float dist = root;\n";
            b += "                const float inv = rinv;\n";
        }
        // force += dir * law with dir = diff / dist, evaluated as diff * (law / dist): one product law * (1 / dist), then
        // one fused multiply-add per component (with the default law 1.0 the product is the reciprocal root itself)
        b += "                const float f = " + rpn_expr(ir.gravity.law, true) + ";\n";
        b += "                const float fs = inv * f;\n";
        if (check_self) {
            b += "                const bool self = (base + t) == idx[q];\n";
            b += "                fx[q] = self ? fx[q] : __builtin_fmaf(dx, fs, fx[q]);\n";
            b += "                fy[q] = self ? fy[q] : __builtin_fmaf(dy, fs, fy[q]);\n";
        } else {
            b += "                fx[q] = __builtin_fmaf(dx, fs, fx[q]);\n";
            b += "                fy[q] = __builtin_fmaf(dy, fs, fy[q]);\n";
        }
        b += "                (void)dist;\n";
        b += "                }\n";
        b += "            }\n";
        return b;
    };
    // the floor of the squared distance: a literal, or (GD_GRAV_FLOOR_SGPR, a tuning hook) the same value read from the
    // launch parameters, which keeps it in a scalar register instead of a 32-bit literal in every v_max_f32
    k += "#ifdef GD_GRAV_FLOOR_SGPR\n    const float d2_floor = __builtin_bit_cast(float, P.floor_bits);\n#else\n    const float d2_floor = 0x1.0c6f7ap-20f;\n#endif\n";
    k += "    const gd_u32 wg_first = P.first + wg_local;\n";
    k += "    for (gd_u32 base = 0; base < P.n; base += 256u) {\n";
    k += "        const gd_u32 j0 = base + threadIdx.x;\n";
    k += "        const float lx = j0 < P.n ? P.xin[j0] : 0.0f, ly = j0 < P.n ? P.yin[j0] : 0.0f;\n";
    k += "        tx[threadIdx.x] = lx;\n";
    k += "        ty[threadIdx.x] = ly;\n";
    k += "        const int wild = __syncthreads_or((int)(own_wild || !(gd_abs(lx) <= 0x1p60f && gd_abs(ly) <= 0x1p60f)));\n";
    k += "        const gd_u32 lim = (P.n - base) < 256u ? (P.n - base) : 256u;\n";
    k += "        const bool overlaps = base < wg_first + " + span + " && wg_first < base + 256u;\n";
    k += "        if (wild) {\n" + tile_loop(true, true) + "        } else if (overlaps) {\n" + tile_loop(true, false) +
         "        } else {\n" + tile_loop(false, false) + "        }\n";
    k += "        __syncthreads();\n";
    k += "    }\n";
    k += "#pragma unroll\n    for (gd_u32 q = 0; q < " + P_ + "u; ++q) {\n";
    k += "    if (!live[q]) continue;\n";
    k += "    float vx = P.vx[idx[q]] * " + float_literal(ir.gravity.damping) + " + fx[q] * P.dt;\n";
    k += "    float vy = P.vy[idx[q]] * " + float_literal(ir.gravity.damping) + " + fy[q] * P.dt;\n";
    k += "    float px = sx[q] + vx * P.dt, py = sy[q] + vy * P.dt;\n";
    if (ir.gravity.bounds == Bounds::Reflect) {
        k += "    if (px < -1.0f) { px = -1.0f; vx = -vx; }\n    if (px > 1.0f) { px = 1.0f; vx = -vx; }\n";
        k += "    if (py < -1.0f) { py = -1.0f; vy = -vy; }\n    if (py > 1.0f) { py = 1.0f; vy = -vy; }\n";
    } else if (ir.gravity.bounds == Bounds::Wrap) {
        k += "    px = gd_fract((px + 1.0f) * 0.5f) * 2.0f - 1.0f;\n    py = gd_fract((py + 1.0f) * 0.5f) * 2.0f - 1.0f;\n";
    }
    k += "    P.xout[idx[q]] = px; P.yout[idx[q]] = py; P.vx[idx[q]] = vx; P.vy[idx[q]] = vy;\n";
    k += "    }\n";
    k += "}\n";
    return k;
}

std::string generate_gravity_kernel(const Ir& ir, bool fast)
{
    std::string k;
    k += "// generated gravity step for cinematic \"" + comment_safe(ir.name) + "\"" + (fast ? " (fast arithmetic)" : "") + "\n";
    k += "#include \"game_device.h\"\n\n";
    k += "#ifndef GD_GRAV_UNROLL\n#define GD_GRAV_UNROLL 8\n#endif\n";
    k += "\n";
    k += "struct GameGravityParams {\n"
         "    const float* xin; const float* yin;\n"
         "    float* xout; float* yout; float* vx; float* vy;\n"
         "    gd_u32 n, first, count, floor_bits;\n"
         "    float dt, time;\n"
         "};\n\n";
    // Positions of all n particles stream through LDS in tiles of 256 and every lane walks j = 0..n-1 in index order
    // (the summation order of the sequential loop). Two builds: one particle per lane (the default) and two (fewer
    // LDS reads per pair, fewer waves in flight). Measured on MI355X the loop is bound by VALU issue, not by LDS: two
    // particles per lane run 7 % slower with the exact arithmetic at 1 M particles and 2 % faster with the fast one,
    // four are slower in both, so the second build is only a tuning hook.
    k += gravity_kernel_body(ir, "game_gravity", 1, fast);
    k += "\n";
    k += gravity_kernel_body(ir, "game_gravity_p2", 2, fast);
    return k;
}

}  // namespace game
