// Host-only driver for the sanitizer build (make -C game-engine_amd asan): runs every .game file named on the command
// line through the whole front end and the kernel generator - lexer, token display, parser, AST dumps, import
// resolution, define expansion, optimiser, validation, IR lowering, JSON and blob serialisation, the host-side
// animation (derived values, arcs, animator steps) and the HIP source generators in every flag combination - without
// touching HIP. Compile errors are expected for some inputs and only counted; the point is that nothing reads or
// writes out of bounds on the way.
//
//     game_asan [--lib DIR]... [--blobs DIR] file.game...
#include "../front/capi_internal.hpp"
#include "../hip/codegen.hpp"
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

using namespace game;

// the ahead-of-time HIP part is not linked into this build
namespace game { int lut_for_current_device(int, const float**) { return GAME_E_NO_DEVICE; } int release_device_tables() { return GAME_OK; } }

int main(int argc, char** argv)
{
    std::vector<std::string> libs, files;
    std::string blob_dir;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--lib" && i + 1 < argc) libs.push_back(argv[++i]);
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
        game_program_free(prog);
    }
    printf("%zu files compiled, %zu refused, %zu kernels generated, %zu blobs written, no fault\n", compiled, failed, kernels, blobs);
    return 0;
}
