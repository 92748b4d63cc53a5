// game — command-line host of the MI355X render path.
//
//   game check  a.game [b.game ...]      the reference's `game check` (game-compiler/src/main.rs:213-257):
//                                        parse + validate, errors rendered with source context, warnings,
//                                        "[game] <file>: ok", exit status 1 on any error
//   game render a.game --hip [--size WxH] [--frames N] [--fps F] [--batch B] [--device D]
//               [--cinematic I] [--out PREFIX] [--particles N --steps S]
//                                        the render command the reference lacks (its frames are drawn by a
//                                        browser): renders N frames with u_time = k / F on the GPU; with
//                                        --out writes PREFIX_00000.ppm ... (binary P6, RGB of the RGBA8 frame),
//                                        or PREFIX_00000.png (8-bit RGBA, stored deflate blocks) with --png;
//                                        --out - writes the frames as raw RGBA8 to stdout, top row first (for a pipe
//                                        into an encoder, e.g. ffmpeg -f rawvideo -pix_fmt rgba -s WxH -r FPS -i -);
//                                        the cinematic's arc timeline (runtime/arc.rs) drives its parameters unless
//                                        --no-arcs; --animate evaluates expression-valued stage arguments per frame;
//                                        --project applies the projection block that names the cinematic
//                                        (the reference's emitters use the parameter defaults) and runs `~` modulation
//                                        and the temporal operators of the parameters on the host; --semantics wgsl renders the builtins as the reference's WebGPU emitter
//                                        words them (wgsl.rs:268-540) instead of its WebGL2 emitter
//   game compare A B [--threshold T]     two frames (.ppm or .png written by this tool): share of pixels whose
//                                        channels all differ by <= T (the measure of snapshot.rs:240-258,
//                                        default T = 2), max |difference| and PSNR; exit 1 when not identical
//   game ir     a.game                   resolved IR of every cinematic as JSON
//   game kernel a.game [--size WxH]      generated HIP source of the fused per-pixel kernel
//   game builtins                        the builtin table (name, input -> output state, parameters)
//
// Only the C ABI of include/game_hip.h is used.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/game_hip.h"

namespace {

bool read_file(const std::string& path, std::string& out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

int usage()
{
    fprintf(stderr,
            "usage: game check FILE...\n"
            "       game render FILE --hip [--size WxH] [--frames N] [--fps F] [--batch B] [--device D]\n"
            "                   [--cinematic I] [--out PREFIX|-] [--particles N --steps S]\n"
            "                   [--png] [--semantics glsl|wgsl] [--no-arcs] [--animate] [--lens-post] [--project] [--imports] [--lib DIR]... [--json] [--gpus N]\n"
            "       game compare A B [--threshold T]\n"
            "       game ir FILE [--cinematic I --out X.gir] | game kernel FILE [--size WxH] | game builtins\n"
            "       FILE is .game text or a resolved-IR blob (X.gir)\n");
    return 2;
}

struct Args {
    std::vector<std::string> files;
    bool hip = false, png = false, wgsl = false, arcs = true, animate = false, lens_post = false, project = false, json = false;
    int threshold = 2;
    uint32_t w = 1920, h = 1080, frames = 60, batch = 16, cinematic = 0, particles = 0, steps = 1, gpus = 1;
    double fps = 60.0;
    int device = 0;
    std::string out;
    bool imports = false;                 // resolve file imports (resolver.rs) before compiling
    std::vector<std::string> lib_dirs;
};

bool parse_args(int argc, char** argv, int first, Args& a)
{
    for (int i = first; i < argc; ++i) {
        std::string s = argv[i];
        auto value = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "game: %s needs a value\n", name); return nullptr; }
            return argv[++i];
        };
        if (s == "--hip") a.hip = true;
        else if (s == "--png") a.png = true;
        else if (s == "--no-arcs") a.arcs = false;
        else if (s == "--animate") a.animate = true;
        else if (s == "--lens-post") a.lens_post = true;
        else if (s == "--project") a.project = true;
        else if (s == "--json") a.json = true;
        else if (s == "--semantics") {
            const char* v = value("--semantics");
            if (!v) return false;
            if (!strcmp(v, "wgsl")) a.wgsl = true;
            else if (strcmp(v, "glsl")) { fprintf(stderr, "game: --semantics wants glsl or wgsl\n"); return false; }
        }
        else if (s == "--threshold") { const char* v = value("--threshold"); if (!v) return false; a.threshold = atoi(v); }
        else if (s == "--size") {
            const char* v = value("--size");
            if (!v || sscanf(v, "%ux%u", &a.w, &a.h) != 2) { fprintf(stderr, "game: --size wants WxH\n"); return false; }
        } else if (s == "--frames") { const char* v = value("--frames"); if (!v) return false; a.frames = (uint32_t)strtoul(v, nullptr, 10); }
        else if (s == "--fps") { const char* v = value("--fps"); if (!v) return false; a.fps = atof(v); }
        else if (s == "--batch") { const char* v = value("--batch"); if (!v) return false; a.batch = (uint32_t)strtoul(v, nullptr, 10); }
        else if (s == "--imports") a.imports = true;
        else if (s == "--lib") { const char* v = value("--lib"); if (!v) return false; a.lib_dirs.push_back(v); a.imports = true; }
        else if (s == "--gpus") { const char* v = value("--gpus"); if (!v) return false; a.gpus = (uint32_t)strtoul(v, nullptr, 10); if (a.gpus < 1 || a.gpus > 64) { fprintf(stderr, "game: --gpus wants 1..64\n"); return false; } }
        else if (s == "--device") { const char* v = value("--device"); if (!v) return false; a.device = atoi(v); }
        else if (s == "--cinematic") { const char* v = value("--cinematic"); if (!v) return false; a.cinematic = (uint32_t)strtoul(v, nullptr, 10); }
        else if (s == "--particles") { const char* v = value("--particles"); if (!v) return false; a.particles = (uint32_t)strtoul(v, nullptr, 10); }
        else if (s == "--steps") { const char* v = value("--steps"); if (!v) return false; a.steps = (uint32_t)strtoul(v, nullptr, 10); }
        else if (s == "--out") { const char* v = value("--out"); if (!v) return false; a.out = v; }
        else if (!s.empty() && s[0] == '-') { fprintf(stderr, "game: unknown option %s\n", s.c_str()); return false; }
        else a.files.push_back(s);
    }
    return true;
}

int cmd_check(const Args& a)
{
    if (a.files.empty()) return usage();
    bool had_errors = false;
    for (const std::string& path : a.files) {
        std::string src;
        if (!read_file(path, src)) { fprintf(stderr, "Error: read: %s\n", path.c_str()); return 1; }
        game_program* prog = nullptr;
        if (game_front_compile(src.data(), src.size(), &prog) != GAME_OK) {
            fprintf(stderr, "%s\n", game_last_error_rendered());
            had_errors = true;
            continue;
        }
        size_t nw = game_program_warning_count(prog);
        for (size_t i = 0; i < nw; ++i) fprintf(stderr, "warning: %s: %s\n", path.c_str(), game_program_warning(prog, i));
        if (!had_errors && nw == 0) fprintf(stderr, "[game] %s: ok\n", path.c_str());
        game_program_free(prog);
    }
    return had_errors ? 1 : 0;
}

game_program* load_program(const std::string& path, const Args& a)
{
    std::string src;
    if (!read_file(path, src)) { fprintf(stderr, "Error: read: %s\n", path.c_str()); return nullptr; }
    game_program* prog = nullptr;
    int rc;
    if (src.size() >= 4 && memcmp(src.data(), "GIR1", 4) == 0) {   // a resolved-IR blob (game ir FILE --out X.gir, or a foreign host's)
        if (game_ir_from_blob(src.data(), src.size(), &prog) != GAME_OK) {
            fprintf(stderr, "Error: %s: %s\n", path.c_str(), game_last_error());
            return nullptr;
        }
        return prog;
    }
    if (a.imports) {   // imports are looked up next to the file, then in each --lib directory
        size_t slash = path.find_last_of('/');
        std::string base = slash == std::string::npos ? "." : path.substr(0, slash ? slash : 1);
        std::vector<const char*> libs;
        for (const std::string& d : a.lib_dirs) libs.push_back(d.c_str());
        rc = game_front_compile_with_imports(src.data(), src.size(), base.c_str(), libs.data(), libs.size(), &prog);
    } else {
        rc = game_front_compile(src.data(), src.size(), &prog);
    }
    if (rc != GAME_OK) {
        fprintf(stderr, "%s\n", game_last_error_rendered());
        return nullptr;
    }
    return prog;
}

bool write_ppm(const std::string& path, const uint8_t* rgba, uint32_t w, uint32_t h)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "P6\n%u %u\n255\n", w, h);
    std::vector<uint8_t> row((size_t)w * 3);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* s = rgba + (size_t)y * w * 4;
        for (uint32_t x = 0; x < w; ++x) { row[3 * x] = s[4 * x]; row[3 * x + 1] = s[4 * x + 1]; row[3 * x + 2] = s[4 * x + 2]; }
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    return true;
}

// ── PNG (8-bit RGBA, no compression: zlib stream of stored blocks) ───────────────────────────────────────
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return crc;
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }

void png_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data)
{
    put_be32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32_update(0xffffffffu, out.data() + start, out.size() - start) ^ 0xffffffffu);
}

bool write_png(const std::string& path, const uint8_t* rgba, uint32_t w, uint32_t h)
{
    std::vector<uint8_t> raw;                       // filter byte 0 + row, per row
    raw.reserve((size_t)h * ((size_t)w * 4 + 1));
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgba + (size_t)y * w * 4, rgba + (size_t)(y + 1) * w * 4);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;                          // Adler-32 of the raw data
    for (size_t i = 0; i < raw.size();) {
        size_t n = raw.size() - i < 5552 ? raw.size() - i : 5552;
        for (size_t k = 0; k < n; ++k) { a += raw[i + k]; b += a; }
        a %= 65521u; b %= 65521u;
        i += n;
    }
    for (size_t i = 0; i < raw.size() || i == 0;) {
        size_t n = raw.size() - i < 65535 ? raw.size() - i : 65535;
        z.push_back(i + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + i, raw.begin() + i + n);
        i += n;
        if (n == 0) break;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w); put_be32(ihdr, h);
    ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0});       // 8 bits, RGBA, deflate, no filter method, no interlace
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", {});
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}

// Reads a frame written by this tool: binary PPM (RGB) or the stored-block PNG above (RGBA). Alpha of a PPM is 255.
bool read_frame(const std::string& path, std::vector<uint8_t>& rgba, uint32_t& w, uint32_t& h)
{
    std::string data;
    if (!read_file(path, data)) return false;
    const uint8_t* p = (const uint8_t*)data.data();
    if (data.size() > 11 && data[0] == 'P' && data[1] == '6') {
        int maxv = 0, used = 0;
        if (sscanf(data.c_str(), "P6 %u %u %d%n", &w, &h, &maxv, &used) != 3 || maxv != 255) return false;
        size_t off = (size_t)used + 1;
        if (data.size() < off + (size_t)w * h * 3) return false;
        rgba.resize((size_t)w * h * 4);
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            rgba[4 * i] = p[off + 3 * i]; rgba[4 * i + 1] = p[off + 3 * i + 1]; rgba[4 * i + 2] = p[off + 3 * i + 2]; rgba[4 * i + 3] = 255;
        }
        return true;
    }
    if (data.size() < 8 + 25 || memcmp(p, "\x89PNG\r\n\x1a\n", 8) != 0) return false;
    auto be32 = [&](size_t o) { return ((uint32_t)p[o] << 24) | ((uint32_t)p[o + 1] << 16) | ((uint32_t)p[o + 2] << 8) | p[o + 3]; };
    std::vector<uint8_t> z;
    bool have_ihdr = false;
    for (size_t o = 8; o + 12 <= data.size();) {
        uint32_t len = be32(o);
        if (o + 12 + len > data.size()) return false;
        if (!memcmp(p + o + 4, "IHDR", 4)) {
            w = be32(o + 8); h = be32(o + 12);
            if (p[o + 16] != 8 || p[o + 17] != 6 || p[o + 20] != 0) return false;
            have_ihdr = true;
        } else if (!memcmp(p + o + 4, "IDAT", 4)) z.insert(z.end(), p + o + 8, p + o + 8 + len);
        o += 12 + len;
    }
    if (!have_ihdr || z.size() < 6) return false;
    std::vector<uint8_t> raw;
    for (size_t i = 2; i + 5 <= z.size();) {          // stored blocks only (what write_png emits)
        if ((z[i] & 6) != 0) return false;
        bool last = z[i] & 1;
        size_t n = z[i + 1] | ((size_t)z[i + 2] << 8);
        if (i + 5 + n > z.size()) return false;
        raw.insert(raw.end(), z.begin() + i + 5, z.begin() + i + 5 + n);
        i += 5 + n;
        if (last) break;
    }
    if (raw.size() != (size_t)h * ((size_t)w * 4 + 1)) return false;
    rgba.resize((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; ++y) {
        if (raw[(size_t)y * (w * 4 + 1)] != 0) return false;
        memcpy(rgba.data() + (size_t)y * w * 4, raw.data() + (size_t)y * (w * 4 + 1) + 1, (size_t)w * 4);
    }
    return true;
}

int cmd_compare(const Args& a)
{
    if (a.files.size() != 2) return usage();
    std::vector<uint8_t> x, y;
    uint32_t w0 = 0, h0 = 0, w1 = 0, h1 = 0;
    if (!read_frame(a.files[0], x, w0, h0)) { fprintf(stderr, "game compare: cannot read %s\n", a.files[0].c_str()); return 2; }
    if (!read_frame(a.files[1], y, w1, h1)) { fprintf(stderr, "game compare: cannot read %s\n", a.files[1].c_str()); return 2; }
    if (w0 != w1 || h0 != h1) { fprintf(stderr, "game compare: sizes differ (%ux%u vs %ux%u)\n", w0, h0, w1, h1); return 1; }
    size_t px = (size_t)w0 * h0, matching = 0;
    int max_abs = 0;
    double sq = 0.0;
    for (size_t i = 0; i < px; ++i) {
        bool ok = true;
        for (int c = 0; c < 3; ++c) {                // RGB, as compare_pixels in snapshot.rs:240-258
            int d = abs((int)x[4 * i + c] - (int)y[4 * i + c]);
            if (d > a.threshold) ok = false;
            if (d > max_abs) max_abs = d;
            sq += (double)d * d;
        }
        matching += ok;
    }
    double mse = sq / (3.0 * (double)px);
    if (mse == 0.0) printf("similarity %.4f %% (threshold %d), max abs diff 0, PSNR inf dB\n", 100.0 * matching / (double)px, a.threshold);
    else printf("similarity %.4f %% (threshold %d), max abs diff %d, PSNR %.2f dB\n", 100.0 * matching / (double)px, a.threshold, max_abs,
                10.0 * log10(255.0 * 255.0 / mse));
    return max_abs == 0 ? 0 : 1;
}

// Renders frames first, first + stride, ... of the cinematic on one GPU context (one thread per context when
// --gpus N splits a stateless cinematic frame k -> GPU k mod N). The animator is stepped through every frame
// index so that its temporal operators see the whole sequence whichever frames this worker renders.
bool render_frames(const game_ir* ir, const Args& a, game_hip_ctx* ctx, uint32_t first, uint32_t stride, double& device_ms, std::string& error)
{
    auto fail = [&](const std::string& m) { error = m; return false; };
    const uint32_t batch = a.batch ? a.batch : 1;
    if (game_hip_set_ring(ctx, game_ir_uses_memory(ir) ? 2u : batch) != GAME_OK) return fail(game_last_error());
    const uint32_t nf = game_ir_frame_floats(ir);
    std::vector<float> rows((size_t)batch * nf);
    std::vector<uint32_t> index(batch);
    // frames come back into page-locked memory (DMA at PCIe rate)
    struct Pinned {
        uint8_t* p = nullptr;
        ~Pinned() { game_hip_host_free(p); }
        bool empty() const { return p == nullptr; }
        uint8_t* data() const { return p; }
    } host;
    if (!a.out.empty()) {
        host.p = (uint8_t*)game_hip_host_alloc((size_t)batch * a.w * a.h * 4);
        if (!host.p) return fail(game_last_error());
    }
    // --animate: modulation, temporal operators and resonance of the parameters run on the host, frame by frame
    struct Anim {
        game_animator* p = nullptr;
        ~Anim() { game_animator_destroy(p); }
    } anim;
    if (a.animate && game_animator_create(ir, a.fps, &anim.p) != GAME_OK) return fail(game_last_error());
    std::vector<float> scratch(nf);
    uint32_t n = 0;
    auto flush = [&]() -> bool {
        if (n == 0) return true;
        if (game_hip_render(ctx, rows.data(), n, host.empty() ? nullptr : host.data()) != GAME_OK) return fail(game_last_error());
        float ms = 0; uint32_t launches = 0;
        game_hip_last_kernel_ms(ctx, &ms, &launches);
        device_ms += ms;
        for (uint32_t k = 0; k < n && !host.empty(); ++k) {
            const uint8_t* frame = host.data() + (size_t)k * a.w * a.h * 4;
            if (a.out == "-") {   // raw RGBA8 frames, top row first, back to back: pipe into an encoder
                if (fwrite(frame, 1, (size_t)a.w * a.h * 4, stdout) != (size_t)a.w * a.h * 4) return fail("short write to stdout");
                continue;
            }
            char name[32];
            snprintf(name, sizeof name, a.png ? "_%05u.png" : "_%05u.ppm", index[k]);
            if (!(a.png ? write_png(a.out + name, frame, a.w, a.h) : write_ppm(a.out + name, frame, a.w, a.h))) return fail("cannot write " + a.out + name);
        }
        n = 0;
        return true;
    };
    for (uint32_t f = 0; f < a.frames; ++f) {
        const bool mine = f >= first && (f - first) % stride == 0;
        if (!mine && !anim.p) continue;
        float* row = mine ? rows.data() + (size_t)n * nf : scratch.data();
        game_ir_default_uniforms(ir, (float)(f / a.fps), a.w, a.h, row);
        if (anim.p) game_animator_step(anim.p, f / a.fps, 1.0 / a.fps, row, (a.arcs ? 1u : 0u) | 6u, nullptr, nullptr, 0);
        else if (a.arcs) game_ir_arc_update(ir, f / a.fps, row);   // the cinematic's own timeline animates its parameters
        if (!mine) continue;
        index[n++] = f;
        if (n == batch && !flush()) return false;
    }
    if (!flush()) return false;
    if (game_hip_sync(ctx) != GAME_OK) return fail(game_last_error());
    return true;
}

int cmd_render(const Args& a)
{
    if (a.files.size() != 1) return usage();
    if (!a.hip) { fprintf(stderr, "game render: only the --hip back end exists; pass --hip\n"); return 2; }
    game_program* prog = load_program(a.files[0], a);
    if (!prog) return 1;
    int rc = 1;
    game_hip_ctx* ctx = nullptr;
    do {
        if (a.cinematic >= game_program_cinematic_count(prog)) { fprintf(stderr, "game render: no cinematic %u\n", a.cinematic); break; }
        const game_ir* ir = game_program_ir(prog, a.cinematic);
        if (game_hip_create_ex(ir, a.w, a.h, a.device, (a.wgsl ? (uint32_t)GAME_HIP_SEMANTICS_WGSL : 0u) | (a.animate ? (uint32_t)GAME_HIP_ANIMATE_ARGS : 0u) | (a.lens_post ? (uint32_t)GAME_HIP_LENS_POST : 0u) | (a.project ? (uint32_t)GAME_HIP_PROJECT : 0u), &ctx) != GAME_OK) { fprintf(stderr, "game render: %s\n", game_last_error()); break; }
        if (a.particles) {
            if (game_hip_particles_seed(ctx, a.particles, 0x9E3779B97F4A7C15ull) != GAME_OK ||
                game_hip_particles_step(ctx, (float)(1.0 / a.fps), a.steps, 0, a.particles) != GAME_OK || game_hip_sync(ctx) != GAME_OK) {
                fprintf(stderr, "game render: %s\n", game_last_error());
                break;
            }
            float ms = 0; uint32_t launches = 0;
            game_hip_last_kernel_ms(ctx, &ms, &launches);
            fprintf(stderr, "[game] %s: %u particles, %u steps, %.3f ms on device (%.1f Mpair/s)\n", game_ir_name(ir), a.particles, a.steps, ms,
                    (double)a.particles * (a.particles - 1.0) * a.steps / (ms * 1e3));
            rc = 0;
            break;
        }
        const uint32_t flags = (a.wgsl ? (uint32_t)GAME_HIP_SEMANTICS_WGSL : 0u) | (a.animate ? (uint32_t)GAME_HIP_ANIMATE_ARGS : 0u) | (a.lens_post ? (uint32_t)GAME_HIP_LENS_POST : 0u) | (a.project ? (uint32_t)GAME_HIP_PROJECT : 0u);
        auto t0 = std::chrono::steady_clock::now();
        double device_ms = 0;
        if (a.gpus <= 1) {
            std::string error;
            if (!render_frames(ir, a, ctx, 0, 1, device_ms, error)) { fprintf(stderr, "game render: %s\n", error.c_str()); break; }
        } else {
            // one context and one host thread per GPU, frame k -> GPU k mod N, no exchange between them
            if (game_ir_uses_memory(ir)) { fprintf(stderr, "game render: --gpus splits whole frames; a memory cinematic is frame-serial (shard it by bands, one process per GPU)\n"); break; }
            if (a.out == "-") { fprintf(stderr, "game render: --out - keeps frame order and needs --gpus 1\n"); break; }
            int ndev = 0;
            if (game_hip_device_count(&ndev) != GAME_OK || ndev < 1) { fprintf(stderr, "game render: %s\n", game_last_error()); break; }
            const bool share = getenv("GAME_CLI_SHARE_DEVICE") != nullptr;   // test hook: every worker on --device
            if (!share && a.device + (int)a.gpus > ndev) { fprintf(stderr, "game render: --gpus %u from device %d needs %d devices, %d visible\n", a.gpus, a.device, a.device + (int)a.gpus, ndev); break; }
            std::vector<game_hip_ctx*> ctxs(a.gpus, nullptr);
            std::vector<std::string> errors(a.gpus);
            std::vector<double> ms(a.gpus, 0.0);
            std::vector<char> good(a.gpus, 0);
            ctxs[0] = ctx;
            bool created = true;
            for (uint32_t g = 1; g < a.gpus && created; ++g)
                if (game_hip_create_ex(ir, a.w, a.h, share ? a.device : a.device + (int)g, flags, &ctxs[g]) != GAME_OK) { fprintf(stderr, "game render: %s\n", game_last_error()); created = false; }
            if (created) {
                std::vector<std::thread> workers;
                for (uint32_t g = 0; g < a.gpus; ++g)
                    workers.emplace_back([&, g]() { good[g] = render_frames(ir, a, ctxs[g], g, a.gpus, ms[g], errors[g]) ? 1 : 0; });
                for (std::thread& t : workers) t.join();
            }
            bool all = created;
            for (uint32_t g = 0; g < a.gpus; ++g) {
                if (created && !good[g]) { fprintf(stderr, "game render: GPU worker %u: %s\n", g, errors[g].c_str()); all = false; }
                if (ms[g] > device_ms) device_ms = ms[g];   // the workers run side by side: the slowest one is the device time
                if (g > 0 && ctxs[g]) game_hip_destroy(ctxs[g]);
            }
            if (!all) break;
        }
        if (a.out == "-") fflush(stdout);
        double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        double px = (double)a.frames * a.w * a.h;
        if (a.json)   // one machine-readable line instead of the sentence
            fprintf(stderr, "{\"cinematic\": \"%s\", \"frames\": %u, \"width\": %u, \"height\": %u, \"device_ms\": %.3f, \"device_mpixel_s\": %.1f, "
                            "\"wall_s\": %.4f, \"wall_mpixel_s\": %.1f}\n", game_ir_name(ir), a.frames, a.w, a.h, device_ms, px / (device_ms * 1e3), wall, px / wall / 1e6);
        else
            fprintf(stderr, "[game] %s: %u frames at %ux%u, %.2f ms on device (%.0f Mpixel/s), %.3f s wall (%.0f Mpixel/s)\n", game_ir_name(ir),
                    a.frames, a.w, a.h, device_ms, px / (device_ms * 1e3), wall, px / wall / 1e6);
        rc = 0;
    } while (false);
    if (ctx) game_hip_destroy(ctx);
    game_program_free(prog);
    return rc;
}

int cmd_ir(const Args& a)
{
    if (a.files.size() != 1) return usage();
    game_program* prog = load_program(a.files[0], a);
    if (!prog) return 1;
    int rc = 0;
    if (!a.out.empty()) {   // the flat blob of one cinematic: what `game render X.gir` and game_ir_from_blob take
        const void* data = nullptr;
        const size_t n = a.cinematic < game_program_cinematic_count(prog) ? game_ir_blob(game_program_ir(prog, a.cinematic), &data) : 0;
        FILE* f = n ? fopen(a.out.c_str(), "wb") : nullptr;
        if (!f || fwrite(data, 1, n, f) != n) { fprintf(stderr, "Error: write: %s\n", a.out.c_str()); rc = 1; }
        if (f) fclose(f);
    } else {
        printf("[");
        for (size_t i = 0; i < game_program_cinematic_count(prog); ++i) printf("%s%s", i ? ",\n" : "", game_ir_json(game_program_ir(prog, i)));
        printf("]\n");
    }
    game_program_free(prog);
    return rc;
}

int cmd_kernel(const Args& a)
{
    if (a.files.size() != 1) return usage();
    game_program* prog = load_program(a.files[0], a);
    if (!prog) return 1;
    int rc = 1;
    // game_hip_dump_code writes the generated source next to the gfx950 code object
    std::string prefix = a.out.empty() ? std::string("game_kernel") : a.out;
    if (a.cinematic < game_program_cinematic_count(prog) && game_hip_dump_code(game_program_ir(prog, a.cinematic), prefix.c_str()) == GAME_OK) {
        std::string src;
        if (read_file(prefix + "_frame.hip", src)) { fputs(src.c_str(), stdout); rc = 0; }
    } else {
        fprintf(stderr, "game kernel: %s\n", game_last_error());
    }
    game_program_free(prog);
    return rc;
}

int cmd_builtins()
{
    for (size_t i = 0; i < game_builtin_count(); ++i) printf("%s\n", game_builtin_describe(i));
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 2) return usage();
    std::string cmd = argv[1];
    Args a;
    if (!parse_args(argc, argv, 2, a)) return 2;
    if (cmd == "check") return cmd_check(a);
    if (cmd == "render") return cmd_render(a);
    if (cmd == "compare") return cmd_compare(a);
    if (cmd == "ir") return cmd_ir(a);
    if (cmd == "kernel") return cmd_kernel(a);
    if (cmd == "builtins") return cmd_builtins();
    return usage();
}
