// Host-only driver for the sanitizer build (make -C game-engine_amd asan): runs every .game file named on the command
// line through the whole front end and the kernel generator - lexer, token display, parser, AST dumps, import
// resolution, define expansion, optimiser, validation, IR lowering, JSON and blob serialisation, the host-side
// animation (derived values, arcs, animator steps) and the HIP source generators in every flag combination - without
// touching HIP. Compile errors are expected for some inputs and only counted; the point is that nothing reads or
// writes out of bounds on the way.
//
// With --mutate N every IR blob produced on the way is then damaged N times at random (single bytes, whole fields set
// to extreme values, truncations with the length field fixed up) and handed to game_ir_from_blob; what the loader
// accepts goes through the JSON dump, the derived-value evaluator and the kernel generators again.
//
//     game_asan [--lib DIR]... [--blobs DIR] [--mutate N [--seed S]] file.game...
#include "../front/capi_internal.hpp"
#include "../hip/codegen.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

using namespace game;

static void exercise_ir(const game_ir* ir, size_t& kernels)
{
    (void)game_ir_json(ir);
    std::vector<float> row(game_ir_frame_floats(ir)), derived(game_ir_derived_count(ir) + 1);
    game_ir_default_uniforms(ir, 0.25f, 64, 36, row.data());
    game_ir_eval_derived(ir, row.data(), derived.data());
    for (uint32_t flags = 0; flags < 16; ++flags) {   // wgsl, animate, lens post, project in every combination
        KernelShape s;
        s.wgsl = flags & 1; s.lens_post = (flags & 4) != 0; s.project = (flags & 8) != 0;
        const uint32_t rowf = BU_COUNT + (uint32_t)ir->ir->uniforms.size();
        s.time_slot = rowf; s.aspect_slot = rowf + 1; s.derived_slot = rowf + 2;
        s.animate = (flags & 2) != 0 && (!ir->ir->derived.empty() || !ir->ir->pixel_exprs.empty());
        for (uint32_t ppl : {1u, 4u, 8u}) {
            s.pixels_per_lane = ppl;
            kernels += !generate_frame_kernel(*ir->ir, s).empty();
        }
    }
    if (game_ir_has_gravity(ir)) kernels += !generate_gravity_kernel(*ir->ir, false).empty() + !generate_gravity_kernel(*ir->ir, true).empty();
}

// the ahead-of-time HIP part is not linked into this build
namespace game { int lut_for_current_device(int, const float**) { return GAME_E_NO_DEVICE; } int release_device_tables() { return GAME_OK; } }

int main(int argc, char** argv)
{
    std::vector<std::string> libs, files;
    std::string blob_dir;
    unsigned long mutate = 0;
    unsigned long long rng = 0x9E3779B97F4A7C15ull;
    std::vector<std::vector<uint8_t>> kept;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--lib" && i + 1 < argc) libs.push_back(argv[++i]);
        else if (a == "--mutate" && i + 1 < argc) mutate = strtoul(argv[++i], nullptr, 10);
        else if (a == "--seed" && i + 1 < argc) rng ^= strtoull(argv[++i], nullptr, 10) * 0x2545F4914F6CDD1Dull;
        else if (a == "--blobs" && i + 1 < argc) blob_dir = argv[++i];
        else files.push_back(a);
    }
    std::vector<const char*> lib_ptrs;
    for (const std::string& l : libs) lib_ptrs.push_back(l.c_str());
    size_t compiled = 0, failed = 0, kernels = 0, blobs = 0;
    for (const std::string& path : files) {
        std::ifstream f(path, std::ios::binary);
        if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); return 2; }
        std::stringstream ss;
        ss << f.rdbuf();
        const std::string src = ss.str();
        size_t ntok = 0;
        (void)game_front_lex(src.data(), src.size(), &ntok, nullptr, 0);
        (void)game_front_tokens(src.data(), src.size());
        (void)game_front_ast_json(src.data(), src.size(), 0);
        (void)game_front_ast_json(src.data(), src.size(), 1);
        const size_t slash = path.find_last_of('/');
        const std::string dir = slash == std::string::npos ? "." : path.substr(0, slash);
        game_program* prog = nullptr;
        int rc = game_front_compile_with_imports(src.data(), src.size(), dir.c_str(), lib_ptrs.data(), lib_ptrs.size(), &prog);
        if (rc != GAME_OK) {
            (void)game_last_error_rendered();
            game_program* plain = nullptr;   // the reference's compile() leaves imports alone: that path too
            if (game_front_compile(src.data(), src.size(), &plain) == GAME_OK) game_program_free(plain);
            ++failed;
            continue;
        }
        ++compiled;
        for (size_t w = 0; w < game_program_warning_count(prog); ++w) (void)game_program_warning(prog, w);
        for (size_t ci = 0; ci < game_program_cinematic_count(prog); ++ci) {
            const game_ir* ir = game_program_ir(prog, ci);
            (void)game_ir_json(ir);
            const void* data = nullptr;
            const size_t n = game_ir_blob(ir, &data);
            if (!blob_dir.empty()) {
                std::ofstream o(blob_dir + "/" + std::to_string(blobs++) + ".gir", std::ios::binary);
                o.write((const char*)data, (std::streamsize)n);
            }
            std::vector<float> row(game_ir_frame_floats(ir)), derived(game_ir_derived_count(ir) + 1);
            game_animator* an = nullptr;
            if (game_animator_create(ir, 60.0, &an) == GAME_OK) {
                const char* names[] = {"bass", "nothing"};
                const double values[] = {0.5, 1.0};
                for (int k = 0; k < 4; ++k) {
                    game_ir_default_uniforms(ir, (float)k / 60.0f, 64, 36, row.data());
                    game_ir_arc_update(ir, k * 0.7, row.data());
                    (void)game_animator_step(an, k / 60.0, 1 / 60.0, row.data(), 0, names, values, 2);
                    game_ir_eval_derived(ir, row.data(), derived.data());
                }
                game_animator_reset(an);
                game_animator_destroy(an);
            }
            exercise_ir(ir, kernels);
            if (mutate) kept.emplace_back((const uint8_t*)data, (const uint8_t*)data + n);
        }
        game_program_free(prog);
    }
    size_t loaded = 0, refused = 0;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    for (const std::vector<uint8_t>& original : kept) {
        {   // the undamaged blob loads and serialises to the same bytes
            game_program* p = nullptr;
            if (game_ir_from_blob(original.data(), original.size(), &p) != GAME_OK) { fprintf(stderr, "own blob refused: %s\n", game_last_error()); return 3; }
            const void* d = nullptr;
            const size_t n = game_ir_blob(game_program_ir(p, 0), &d);
            if (n != original.size() || memcmp(d, original.data(), n) != 0) { fprintf(stderr, "blob does not round-trip\n"); return 3; }
            game_program_free(p);
        }
        for (unsigned long m = 0; m < mutate; ++m) {
            std::vector<uint8_t> b = original;
            const unsigned kind = (unsigned)(next() % 4);
            if (kind == 0) {
                for (unsigned k = 0, n = 1 + (unsigned)(next() % 4); k < n; ++k) b[next() % b.size()] = (uint8_t)next();
            } else if (kind == 1) {
                static const uint32_t extreme[] = {0u, 1u, 0xffffffffu, 0x7fffffffu, 0x80000000u, 0x7fc00000u, 0x7f800000u, 65536u, 4097u};
                const uint32_t v = extreme[next() % 9];
                memcpy(&b[(next() % (b.size() / 4)) * 4], &v, 4);
            } else if (kind == 2) {
                b.resize(next() % b.size());
                if (b.size() >= 8) { const uint32_t n = (uint32_t)b.size(); memcpy(&b[4], &n, 4); }
            } else {
                const size_t extra = 4 * (1 + next() % 8);
                b.resize(b.size() + extra, (uint8_t)next());
                const uint32_t n = (uint32_t)b.size(); memcpy(&b[4], &n, 4);
            }
            game_program* p = nullptr;
            if (game_ir_from_blob(b.data(), b.size(), &p) == GAME_OK) {
                ++loaded;
                exercise_ir(game_program_ir(p, 0), kernels);
                game_program_free(p);
            } else {
                ++refused;
            }
        }
    }
    printf("%zu files compiled, %zu refused, %zu kernels generated, %zu blobs written, %zu damaged blobs loaded, %zu refused, no fault\n",
           compiled, failed, kernels, blobs, loaded, refused);
    return 0;
}
