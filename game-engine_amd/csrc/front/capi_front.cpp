// C ABI for the front end (include/game_hip.h, "front end" and "IR accessors" sections).
#include "front.hpp"
#include "capi_internal.hpp"
#include <vector>
#include <cstring>

using namespace game;

namespace game {
thread_local std::string g_last_error;
thread_local std::string g_last_rendered;

int set_error(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

int error_code_of(const CompileError& e)
{
    if (e.code >= 1 && e.code <= 10) return e.code;
    switch (e.kind) {
    case ErrKind::Lex: return GAME_E_LEX;
    case ErrKind::Parse: return GAME_E003_PARSE;
    case ErrKind::Validation: return GAME_E_VALIDATION;
    case ErrKind::Codegen: return GAME_E_CODEGEN;
    default: return GAME_OK;
    }
}
}  // namespace game

extern "C" {

const char* game_last_error(void) { return g_last_error.c_str(); }
const char* game_last_error_rendered(void) { return g_last_rendered.c_str(); }

namespace {
int finish_compile(game_program* p, const CompileError& e, const std::string& source, game_program** out)
{
    if (!e.ok()) {
        delete p;
        g_last_rendered = e.render_with_source(source);
        return set_error(error_code_of(e), e.display());
    }
    p->irs.resize(p->prog.irs.size());
    for (size_t i = 0; i < p->prog.irs.size(); ++i) {
        p->irs[i].ir = &p->prog.irs[i];
        p->irs[i].blob = &p->prog.blobs[i];
        p->irs[i].json = &p->prog.jsons[i];
    }
    g_last_error.clear();
    g_last_rendered.clear();
    *out = p;
    return GAME_OK;
}
}  // namespace

int game_front_compile(const char* src, size_t len, game_program** out)
{
    if (!src || !out) return set_error(GAME_E_ARGUMENT, "game_front_compile: null argument");
    *out = nullptr;
    std::string source(src, len);
    auto* p = new game_program;
    CompileError e = compile_source(source, p->prog);
    return finish_compile(p, e, source, out);
}

int game_front_compile_with_imports(const char* src, size_t len, const char* base_dir, const char* const* lib_dirs, size_t n_lib_dirs,
                                    game_program** out)
{
    if (!src || !out || !base_dir || (n_lib_dirs && !lib_dirs)) return set_error(GAME_E_ARGUMENT, "game_front_compile_with_imports: null argument");
    *out = nullptr;
    std::string source(src, len);
    std::vector<std::string> libs;
    for (size_t i = 0; i < n_lib_dirs; ++i) if (lib_dirs[i]) libs.emplace_back(lib_dirs[i]);
    auto* p = new game_program;
    CompileError e = compile_source_with_imports(source, base_dir, libs, p->prog);
    return finish_compile(p, e, source, out);
}

int game_ir_from_blob(const void* data, size_t len, game_program** out)
{
    if (!data || !out) return set_error(GAME_E_ARGUMENT, "game_ir_from_blob: null argument");
    *out = nullptr;
    auto* p = new game_program;
    p->prog.irs.resize(1);
    std::string why;
    if (!ir_deserialize((const uint8_t*)data, len, p->prog.irs[0], why)) {
        delete p;
        g_last_rendered.clear();
        return set_error(GAME_E_ARGUMENT, "game_ir_from_blob: " + why);
    }
    p->prog.blobs.push_back(ir_serialize(p->prog.irs[0]));
    p->prog.jsons.push_back(ir_to_json(p->prog.irs[0]));
    return finish_compile(p, CompileError{}, std::string(), out);
}

void game_program_free(game_program* p) { delete p; }
size_t game_program_cinematic_count(const game_program* p) { return p ? p->irs.size() : 0; }
const game_ir* game_program_ir(const game_program* p, size_t i) { return (p && i < p->irs.size()) ? &p->irs[i] : nullptr; }
size_t game_program_warning_count(const game_program* p) { return p ? p->prog.warnings.size() : 0; }
const char* game_program_warning(const game_program* p, size_t i)
{
    return (p && i < p->prog.warnings.size()) ? p->prog.warnings[i].c_str() : "";
}

int game_front_lex(const char* src, size_t len, size_t* ntokens, size_t* spans, size_t cap)
{
    if (!src || !ntokens) return set_error(GAME_E_ARGUMENT, "game_front_lex: null argument");
    std::vector<Token> toks;
    std::string source(src, len);
    CompileError e = lex(source, toks);
    if (!e.ok()) {
        g_last_rendered = e.render_with_source(source);
        return set_error(error_code_of(e), e.display());
    }
    *ntokens = toks.size();
    if (spans)
        for (size_t i = 0; i < toks.size() && i < cap; ++i) { spans[2 * i] = toks[i].start; spans[2 * i + 1] = toks[i].end; }
    return GAME_OK;
}

size_t game_builtin_count(void) { return builtin_count(); }
const char* game_builtin_describe(size_t index)
{
    static thread_local std::string s;
    const Builtin* b = builtin_by_id((uint16_t)index);
    if (!b) return "";
    s = std::string(b->name) + "|" + state_name(b->input) + "|" + state_name(b->output) + "|";
    for (int i = 0; i < b->nparams; ++i) { if (i) s += ","; s += b->params[i].name; }
    return s.c_str();
}

const char* game_ir_name(const game_ir* ir) { return ir ? ir->ir->name.c_str() : ""; }
const char* game_ir_json(const game_ir* ir) { return ir ? ir->json->c_str() : ""; }
size_t game_ir_blob(const game_ir* ir, const void** data)
{
    if (!ir) return 0;
    if (data) *data = ir->blob->data();
    return ir->blob->size();
}
uint32_t game_ir_uniform_count(const game_ir* ir) { return ir ? (uint32_t)ir->ir->uniforms.size() : 0; }
const char* game_ir_uniform_name(const game_ir* ir, uint32_t i)
{
    return (ir && i < ir->ir->uniforms.size()) ? ir->ir->uniforms[i].name.c_str() : "";
}
double game_ir_uniform_default(const game_ir* ir, uint32_t i)
{
    return (ir && i < ir->ir->uniforms.size()) ? ir->ir->uniforms[i].def : 0.0;
}
uint32_t game_ir_flags(const game_ir* ir) { return ir ? ir->ir->flags : 0; }
int game_ir_uses_memory(const game_ir* ir) { return ir && ir->ir->uses_memory; }
int game_ir_has_gravity(const game_ir* ir) { return ir && ir->ir->gravity.present; }
uint32_t game_ir_frame_floats(const game_ir* ir) { return ir ? BU_COUNT + (uint32_t)ir->ir->uniforms.size() : 0; }

void game_ir_default_uniforms(const game_ir* ir, float time_s, uint32_t w, uint32_t h, float* row)
{
    if (!ir || !row) return;
    for (uint32_t i = 0; i < BU_COUNT; ++i) row[i] = 0.0f;
    row[BU_TIME] = time_s;
    row[BU_RES_X] = (float)w;
    row[BU_RES_Y] = (float)h;
    for (size_t i = 0; i < ir->ir->uniforms.size(); ++i) row[BU_COUNT + i] = (float)ir->ir->uniforms[i].def;
}

uint32_t game_ir_derived_count(const game_ir* ir) { return ir ? (uint32_t)ir->ir->derived.size() : 0; }
int game_ir_project_mode(const game_ir* ir) { return ir && ir->ir->project.present ? (int)ir->ir->project.mode : -1; }
uint32_t game_ir_pixel_expr_count(const game_ir* ir) { return ir ? (uint32_t)ir->ir->pixel_exprs.size() : 0; }

void game_ir_eval_derived(const game_ir* ir, const float* row, float* out)
{
    if (ir && row && out) eval_derived(*ir->ir, row, out);
}

struct game_animator { Ir ir; Animator a; };   // owns a copy: the program may be freed first

int game_animator_create(const game_ir* ir, double fps, game_animator** out)
{
    if (!ir || !out) return set_error(GAME_E_ARGUMENT, "game_animator_create: null argument");
    if (!(fps >= 0.0) || fps > 1.0e6) return set_error(GAME_E_ARGUMENT, "game_animator_create: fps must be 0 (= 60) or a positive rate");
    auto* an = new game_animator;
    an->ir = *ir->ir;
    an->a.init(an->ir, fps);
    *out = an;
    return GAME_OK;
}

void game_animator_destroy(game_animator* an) { delete an; }
void game_animator_reset(game_animator* an) { if (an) an->a.reset(); }

int game_animator_step(game_animator* an, double time_s, double dt, float* row, uint32_t what,
                       const char* const* signal_names, const double* signal_values, uint32_t nsignals)
{
    if (!an || !row) return set_error(GAME_E_ARGUMENT, "game_animator_step: null argument");
    if (what > 7u) return set_error(GAME_E_ARGUMENT, "game_animator_step: unknown bit in `what`");
    if (nsignals && (!signal_names || !signal_values)) return set_error(GAME_E_ARGUMENT, "game_animator_step: signals without names or values");
    an->a.step(time_s, dt, row, what, signal_names, signal_values, nsignals);
    return GAME_OK;
}

uint32_t game_ir_arc_count(const game_ir* ir) { return ir ? (uint32_t)ir->ir->arcs.size() : 0; }

void game_ir_arc_update(const game_ir* ir, double time_s, float* row)
{
    if (!ir || !row || ir->ir->arcs.empty()) return;
    const size_t n = ir->ir->uniforms.size();
    std::vector<double> params(n);
    for (size_t i = 0; i < n; ++i) params[i] = (double)row[BU_COUNT + i];
    arc_update(*ir->ir, time_s, params.data());
    for (size_t i = 0; i < n; ++i) row[BU_COUNT + i] = (float)params[i];   // uploaded as Float32
}

}  // extern "C"
