// game — command-line host of the MI355X render path.
//
//   game check  a.game [b.game ...]      the reference's `game check` (game-compiler/src/main.rs:213-257):
//                                        parse + validate, errors rendered with source context, warnings,
//                                        "[game] <file>: ok", exit status 1 on any error
//   game render a.game --hip [--size WxH] [--frames N] [--fps F] [--batch B] [--device D]
//               [--cinematic I] [--out PREFIX] [--particles N --steps S]
//                                        the render command the reference lacks (its frames are drawn by a
//                                        browser): renders N frames with u_time = k / F on the GPU; with
//                                        --out writes PREFIX_00000.ppm ... (binary P6, RGB of the RGBA8 frame)
//   game ir     a.game                   resolved IR of every cinematic as JSON
//   game kernel a.game [--size WxH]      generated HIP source of the fused per-pixel kernel
//   game builtins                        the builtin table (name, input -> output state, parameters)
//
// Only the C ABI of include/game_hip.h is used.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
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
            "                   [--cinematic I] [--out PREFIX] [--particles N --steps S]\n"
            "       game ir FILE | game kernel FILE [--size WxH] | game builtins\n");
    return 2;
}

struct Args {
    std::vector<std::string> files;
    bool hip = false;
    uint32_t w = 1920, h = 1080, frames = 60, batch = 16, cinematic = 0, particles = 0, steps = 1;
    double fps = 60.0;
    int device = 0;
    std::string out;
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
        else if (s == "--size") {
            const char* v = value("--size");
            if (!v || sscanf(v, "%ux%u", &a.w, &a.h) != 2) { fprintf(stderr, "game: --size wants WxH\n"); return false; }
        } else if (s == "--frames") { const char* v = value("--frames"); if (!v) return false; a.frames = (uint32_t)strtoul(v, nullptr, 10); }
        else if (s == "--fps") { const char* v = value("--fps"); if (!v) return false; a.fps = atof(v); }
        else if (s == "--batch") { const char* v = value("--batch"); if (!v) return false; a.batch = (uint32_t)strtoul(v, nullptr, 10); }
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

game_program* load_program(const std::string& path)
{
    std::string src;
    if (!read_file(path, src)) { fprintf(stderr, "Error: read: %s\n", path.c_str()); return nullptr; }
    game_program* prog = nullptr;
    if (game_front_compile(src.data(), src.size(), &prog) != GAME_OK) {
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

int cmd_render(const Args& a)
{
    if (a.files.size() != 1) return usage();
    if (!a.hip) { fprintf(stderr, "game render: only the --hip back end exists; pass --hip\n"); return 2; }
    game_program* prog = load_program(a.files[0]);
    if (!prog) return 1;
    int rc = 1;
    game_hip_ctx* ctx = nullptr;
    do {
        if (a.cinematic >= game_program_cinematic_count(prog)) { fprintf(stderr, "game render: no cinematic %u\n", a.cinematic); break; }
        const game_ir* ir = game_program_ir(prog, a.cinematic);
        if (game_hip_create(ir, a.w, a.h, a.device, &ctx) != GAME_OK) { fprintf(stderr, "game render: %s\n", game_last_error()); break; }
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
        const uint32_t batch = a.batch ? a.batch : 1;
        if (game_hip_set_ring(ctx, game_ir_uses_memory(ir) ? 2u : batch) != GAME_OK) { fprintf(stderr, "game render: %s\n", game_last_error()); break; }
        const uint32_t nf = game_ir_frame_floats(ir);
        std::vector<float> rows((size_t)batch * nf);
        std::vector<uint8_t> host;
        if (!a.out.empty()) host.resize((size_t)batch * a.w * a.h * 4);
        auto t0 = std::chrono::steady_clock::now();
        double device_ms = 0;
        bool ok = true;
        for (uint32_t done = 0; done < a.frames && ok; done += batch) {
            uint32_t n = a.frames - done < batch ? a.frames - done : batch;
            for (uint32_t k = 0; k < n; ++k) game_ir_default_uniforms(ir, (float)((done + k) / a.fps), a.w, a.h, rows.data() + (size_t)k * nf);
            if (game_hip_render(ctx, rows.data(), n, host.empty() ? nullptr : host.data()) != GAME_OK) { ok = false; break; }
            float ms = 0; uint32_t launches = 0;
            game_hip_last_kernel_ms(ctx, &ms, &launches);
            device_ms += ms;
            for (uint32_t k = 0; k < n && !host.empty(); ++k) {
                char name[32];
                snprintf(name, sizeof name, "_%05u.ppm", done + k);
                if (!write_ppm(a.out + name, host.data() + (size_t)k * a.w * a.h * 4, a.w, a.h)) {
                    fprintf(stderr, "game render: cannot write %s%s\n", a.out.c_str(), name);
                    ok = false;
                    break;
                }
            }
        }
        if (!ok || game_hip_sync(ctx) != GAME_OK) { fprintf(stderr, "game render: %s\n", game_last_error()); break; }
        double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        double px = (double)a.frames * a.w * a.h;
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
    game_program* prog = load_program(a.files[0]);
    if (!prog) return 1;
    printf("[");
    for (size_t i = 0; i < game_program_cinematic_count(prog); ++i) printf("%s%s", i ? ",\n" : "", game_ir_json(game_program_ir(prog, i)));
    printf("]\n");
    game_program_free(prog);
    return 0;
}

int cmd_kernel(const Args& a)
{
    if (a.files.size() != 1) return usage();
    game_program* prog = load_program(a.files[0]);
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
    if (cmd == "ir") return cmd_ir(a);
    if (cmd == "kernel") return cmd_kernel(a);
    if (cmd == "builtins") return cmd_builtins();
    return usage();
}
