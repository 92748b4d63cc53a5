// GAME front end (host side of the render path): tokens, AST, errors, passes, flat IR.
//
// Mirrors the reference compiler's stages up to the point where it would emit shader text:
//   lexer   -> game-compiler/src/lexer.rs:162      (lex)
//   parser  -> game-compiler/src/parser.rs:229     (Parser::parse)
//   defines -> game-compiler/src/codegen/analysis.rs:24 (expand_defines)
//   fold    -> game-compiler/src/optimize.rs:25    (optimize_cinematic)
//   check   -> game-compiler/src/codegen/mod.rs:110 (validate), :54 (extract_uniforms)
// The output is a flat IR (struct Ir) that the HIP back end and the frame scheduler consume.
#pragma once
#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>
#include <memory>

namespace game {

// ── tokens (token.rs:6-80) ───────────────────────────────────────────────
enum class Tok : uint8_t {
    Cinematic, Layer, Import, As, Arc, Resonate, Memory, Cast, Over, Listen, Voice, Score, Breed,
    From, Inherit, Mutate, Gravity, Project, Signals, Route, Hear, Feel, Lens, React, Define,
    Expose, Ease, All,
    Pipe, Tilde, LBrace, RBrace, LParen, RParen, LBracket, RBracket, Colon, Comma, Dot, Plus,
    Minus, Star, Slash, Caret, Eq, Arrow, ShiftRight, Diamond, BangBang, DotDot, Greater, Less,
    Question,
    Float, Integer, StringLit, Ident,
    Seconds, Millis, Bars, Degrees,
};

struct Token {
    Tok kind;
    double num = 0.0;     // Float / Seconds / Millis / Degrees
    int64_t inum = 0;     // Integer / Bars
    std::string text;     // StringLit (unescaped) / Ident
    size_t start = 0, end = 0;
};

std::string token_display(const Token& t);   // token.rs:82-149
std::string fmt_f64_display(double v);       // Rust `{}` for f64 (shortest round trip, no exponent)

// ── errors (error.rs:173-292) ────────────────────────────────────────────
enum class ErrKind : uint8_t { None, Lex, Parse, Validation, Codegen };

struct CompileError {
    ErrKind kind = ErrKind::None;
    int code = 0;                 // 0 = none, 1..10 = E001..E010
    std::string message;
    std::string help;             // non-empty == wrapped in WithHelp
    size_t a = 0, b = 0;          // Lex: span start/end. Parse: the reference's "line"/"col" (token byte span)
    bool ok() const { return kind == ErrKind::None; }
    std::string display() const;                            // thiserror Display strings
    std::string render_with_source(const std::string& src) const;  // error.rs:296
};

size_t levenshtein(const std::string& a, const std::string& b);                       // error.rs:133
const char* suggest_similar(const std::string& typed, const std::vector<const char*>& c);  // error.rs:155

// ── AST (ast.rs) ─────────────────────────────────────────────────────────
enum class BinOp : uint8_t { Add, Sub, Mul, Div, Pow, Gt, Lt };
enum class DurKind : uint8_t { Seconds, Millis, Bars };
struct Duration { DurKind kind = DurKind::Seconds; double v = 0.0; int64_t bars = 0; };

struct Arg;
struct Expr {
    enum Kind : uint8_t { Number, String, Ident, Dotted, Array, Paren, Neg, Bin, Call, Dur, Ternary } kind = Number;
    double num = 0.0;
    std::string s;          // String / Ident / Dotted.object / Call.name
    std::string field;      // Dotted.field
    BinOp op = BinOp::Add;
    Duration dur;
    std::vector<Expr> kids; // Paren/Neg: 1, Bin: 2, Ternary: 3, Array: n
    std::vector<Arg> args;  // Call
    static Expr number(double v) { Expr e; e.kind = Number; e.num = v; return e; }
};
struct Arg { bool named = false; std::string name; Expr value; };
struct Stage { std::string name; std::vector<Arg> args; };

struct TemporalOp { enum Kind : uint8_t { Delay, Smooth, Trigger, Range } kind; Duration dur; Expr lo, hi; };
struct Param { std::string name; Expr value; bool has_mod = false; Expr modulation; std::vector<TemporalOp> temporal; };

struct Layer {
    std::string name;
    std::vector<Param> opts;
    bool has_memory = false; double memory = 0.0;
    bool has_cast = false; std::string cast;
    bool is_pipeline = false;             // LayerBody::Pipeline vs LayerBody::Params
    std::vector<Param> params;
    std::vector<Stage> stages;
};

struct ArcEntry { std::string target; Expr from, to; Duration dur; bool has_easing = false; std::string easing; };
struct ArcBlock { std::vector<ArcEntry> entries; };
struct ResonateEntry { std::string source, target, field; Expr weight; };
struct ResonateBlock { std::vector<ResonateEntry> entries; };
struct NamedNode { std::string name, kind; std::vector<Param> params; };   // listen signal / voice node
struct Motif { std::string name; std::vector<ArcEntry> entries; };
struct NameList { std::string name; std::vector<std::string> refs; };
struct ScoreBlock { double tempo_bpm = 120.0; std::vector<Motif> motifs; std::vector<NameList> phrases, sections; std::vector<std::string> arrange; };
enum class Bounds : uint8_t { Reflect, Wrap, None };
struct GravityBlock { Expr force_law = Expr::number(1.0); double damping = 0.99; Bounds bounds = Bounds::Reflect; };
struct Lens { bool has_name = false; std::string name; std::vector<Param> properties; std::vector<Stage> post; };
struct Reaction { Expr signal, action; };
struct DefineBlock { std::string name; std::vector<std::string> params; std::vector<Stage> body; };

struct Cinematic {
    std::string name;
    std::vector<Layer> layers;
    std::vector<ArcBlock> arcs;
    std::vector<ResonateBlock> resonates;
    bool has_listen = false; std::vector<NamedNode> listen;
    bool has_voice = false; std::vector<NamedNode> voice;
    bool has_score = false; ScoreBlock score;
    bool has_gravity = false; GravityBlock gravity;
    std::vector<Lens> lenses;
    bool has_react = false; std::vector<Reaction> react;
    std::vector<DefineBlock> defines;
};

struct Import { std::string path, alias; std::vector<std::string> exposed; };
struct InheritRule { std::string target, strategy; double weight = 0.0; };
struct Mutation { std::string target; double range = 0.0; };
struct BreedBlock { std::string name; std::vector<std::string> parents; std::vector<InheritRule> inherit; std::vector<Mutation> mutations; };
enum class ProjectMode : uint8_t { Flat, Dome, Cube, Led };
struct ProjectBlock { ProjectMode mode = ProjectMode::Flat; std::string source; std::vector<Param> params; };

struct Program {
    std::vector<Import> imports;
    std::vector<Cinematic> cinematics;
    std::vector<BreedBlock> breeds;
    std::vector<ProjectBlock> projects;
};

// ── builtins (builtins.rs:221-293) ───────────────────────────────────────
enum class State : uint8_t { Position = 0, Sdf = 1, Color = 2 };
const char* state_name(State s);

struct BuiltinParam { const char* name; bool has_default; double def; };
struct Builtin {
    const char* name;
    uint16_t id;
    uint8_t nparams;
    BuiltinParam params[4];
    State input, output;
};
enum BuiltinId : uint16_t {
    B_CIRCLE, B_RING, B_STAR, B_FBM, B_SIMPLEX, B_GLOW, B_SHADE, B_EMISSIVE, B_TINT, B_BLOOM, B_GRAIN,
    B_TRANSLATE, B_ROTATE, B_SCALE, B_MASK_ARC, B_THRESHOLD, B_TWIST, B_MIRROR, B_REPEAT, B_GRADIENT,
    B_BLEND, B_VIGNETTE, B_VORONOI, B_ONION, B_DOMAIN_WARP, B_CURL_NOISE, B_BOX, B_POLYGON,
    B_CONCENTRIC_WAVES, B_ROUND, B_TONEMAP, B_SCANLINES, B_CHROMATIC, B_SATURATE_COLOR, B_GLITCH,
    B_DISPLACE, B_SPECTRUM, B_PALETTE, B_COUNT
};
const Builtin* builtin_lookup(const std::string& name);   // builtins.rs:296
const Builtin* builtin_by_id(uint16_t id);
size_t builtin_count();
const char* builtin_suggest(const std::string& name);     // builtins.rs:306

// ── front-end stages ─────────────────────────────────────────────────────
CompileError lex(const std::string& src, std::vector<Token>& out);         // lexer.rs:162
CompileError parse(std::vector<Token> toks, Program& out);                 // parser.rs:229
CompileError expand_defines(Cinematic& c);                                  // analysis.rs:24

struct OptimizeStats {                                                      // optimize.rs:16
    size_t constants_folded = 0, noop_stages_removed = 0, dead_defines_removed = 0;
    std::vector<std::string> dead_uniforms, semantic_warnings;
};
size_t constant_fold(Expr& e);                                              // optimize.rs:104
size_t eliminate_noop_stages(std::vector<Stage>& stages);                   // optimize.rs:242
OptimizeStats optimize_cinematic(Cinematic& c);                             // optimize.rs:25
std::vector<std::string> check_define_semantics(const Cinematic& c);        // optimize.rs:433
std::vector<std::string> check_arity(const Cinematic& c);                   // optimize.rs:499
std::vector<std::string> check_program(const Program& p);                   // lib.rs:83

CompileError validate_pipeline(const std::vector<Stage>& stages, State& end_state);  // stages.rs:45
CompileError validate_cinematic(const Cinematic& c);                        // codegen/mod.rs:110 (+ cast.rs:60)

struct UniformInfo { std::string name; double def = 0.0; };                 // codegen/mod.rs:29
std::vector<UniformInfo> extract_uniforms(const Cinematic& c);              // codegen/mod.rs:54

// Text the reference would paste into the shader for one stage argument (stages.rs:8-42).
struct ArgText {
    enum Kind : uint8_t { Literal, Ident, Dotted } kind = Literal;
    std::string text;
    const Expr* dropped = nullptr;   // the expression stages.rs:8-22 replaces by the default (not Number / Ident / Dotted)
};
ArgText get_arg(const std::vector<Arg>& args, const char* name, size_t pos, const char* stage_name);
std::string fmt_fixed6(double v);                                           // Rust `{v:.6}`

// Force-law expression printed as the reference prints it into WGSL (gravity.rs:10-41).
std::string force_expr_text(const Expr& e);

// ── flat IR ──────────────────────────────────────────────────────────────
enum ArgKind : uint32_t {
    ARG_LIT = 0,          // literal f32 (parsed from the "%.6f" text, as a GLSL compiler would)
    ARG_USER = 1,         // user uniform, slot = index into Ir::uniforms
    ARG_BUILTIN = 2,      // built-in per-frame uniform, slot = BuiltinUniform
};
enum BuiltinUniform : uint32_t {   // order of the per-frame uniform vector (runtime/helpers.rs:222-232)
    BU_TIME = 0, BU_AUDIO_BASS, BU_AUDIO_MID, BU_AUDIO_TREBLE, BU_AUDIO_ENERGY, BU_AUDIO_BEAT,
    BU_RES_X, BU_RES_Y, BU_MOUSE_X, BU_MOUSE_Y, BU_COUNT
};
// Reasons the reference's emitted GLSL would not compile or leaves a value undefined; the IR then
// carries the evident intent and these bits say which liberty was taken.
enum IrFlag : uint32_t {
    IRF_DOTTED_ARG       = 1u << 0,  // `audio.bass` -> undeclared `audio_bass`; mapped to the built-in uniform
    IRF_TIME_IDENT       = 1u << 1,  // bare `time` argument redeclares `float time`; mapped to shader time
    IRF_BUILTIN_IDENT    = 1u << 2,  // identifier that names a builtin is pasted undeclared; treated as 0
    IRF_BLOCK_SCOPED_SDF = 1u << 3,  // star/polygon declare sdf_result inside a block
    IRF_REDECLARED       = 1u << 4,  // a second generator/bridge/vignette/... redeclares a variable
    IRF_NO_COLOR         = 1u << 5,  // a pipeline layer does not end in Color state (no color_result)
    IRF_UNKNOWN_DOTTED   = 1u << 6,  // dotted identifier with no built-in meaning; treated as 0
    IRF_NO_OUTPUT        = 1u << 7,  // no pipeline layer at all: fragColor never written
    IRF_GRAVITY_IDENT    = 1u << 8,  // force law references an unknown identifier / function
    IRF_COLOR_BEFORE_DECL= 1u << 9,  // a Sdf/Color-input stage is reached with that value undeclared
};

// derived >= 0: the source wrote an expression here. The reference's emitters drop it for the parameter's
// default (kind/slot/lit above hold that); a context created with GAME_HIP_ANIMATE_ARGS instead evaluates
// Ir::derived[derived] on the host once per frame and passes the value as an extra uniform.
// pixel >= 0: the expression also reads the pixel (`p.x`, `p.y`, `uv.x`, `uv.y`); such a context evaluates
// Ir::pixel_exprs[pixel] inside the kernel, in f32 with the device functions, where the stage runs.
struct IrArg { uint32_t kind = ARG_LIT; uint32_t slot = 0; float lit = 0.0f; std::string text; int derived = -1; int pixel = -1; };
struct IrStage { uint16_t builtin = 0; uint32_t nargs = 0; IrArg args[4]; uint32_t aux = 0; std::string name; };
struct IrLayer {
    std::string name;
    bool is_pipeline = false;
    uint32_t stage_begin = 0, stage_count = 0;
    bool has_memory = false; float decay = 0.0f; std::string decay_text;
    State end_state = State::Position;
    bool post = false;   // the lens post chain: runs on the composited colour, only when the renderer is asked to
};
enum LawOp : uint32_t { LAW_NUM = 0, LAW_DIST, LAW_TIME, LAW_ADD, LAW_SUB, LAW_MUL, LAW_DIV, LAW_POW, LAW_GT, LAW_LT,
                        LAW_NEG, LAW_CALL1, LAW_CALL2, LAW_CALL3, LAW_UNIFORM, LAW_PX, LAW_PY, LAW_UVX, LAW_UVY };
enum LawFn : uint32_t { LF_SIN = 0, LF_COS, LF_TAN, LF_ABS, LF_SQRT, LF_EXP, LF_LOG, LF_FLOOR, LF_CEIL, LF_FRACT, LF_SIGN,
                        LF_MIN, LF_MAX, LF_POW, LF_STEP, LF_CLAMP, LF_MIX, LF_SMOOTHSTEP, LF_EXP2, LF_LOG2,
                        LF_ROUND, LF_MOD, LF_ATAN2, LF_LENGTH2, LF_SELECT, LF_COUNT };
struct IrLawOp { uint32_t op = LAW_NUM; float val = 0.0f; uint32_t fn = 0; };
struct IrGravity {
    bool present = false;
    float damping = 0.99f; std::string damping_text;
    Bounds bounds = Bounds::Reflect;
    std::vector<IrLawOp> law;     // postfix program over {dist, time}
    std::string law_text;         // as printed into WGSL by the reference
};
// `project <mode> (params) { source: name }` (codegen/project.rs): the reference replaces the WebGPU vertex stage, which
// changes the v_uv every fragment starts from. Here it is a per-pixel warp of v_uv before the layers run.
//   flat, cube: v_uv unchanged (cube only adds a face index the fragment stage never reads, project.rs:79-110)
//   dome      : fisheye, project.rs:62-69 (fov_deg from the parameter fov / fov_deg / fisheye, default 180)
//   led       : every pixel of LED k = floor(v_uv.x * count) samples (k / count, 0.5), project.rs:128-137
//               (count from the parameter count / leds, default 60)
struct IrProject {
    bool present = false;
    ProjectMode mode = ProjectMode::Flat;
    float fov_deg = 180.0f;
    uint32_t count = 60;
};
// One entry of the arc timeline (runtime/arc.rs:52-125): uniform `idx` moves from `from` to `to` over
// [start, start + dur) seconds through `easing`; entries run back to back in source order. idx < 0: the
// target names no uniform, or an endpoint is not a constant number (the emitted JS would skip or fail there).
struct IrArc {
    int idx = -1;
    std::string target;
    double from = 0.0, to = 0.0, start = 0.0, dur = 0.0;
    std::string easing;
};

// A frame-level scalar expression in reverse Polish form (the operators and functions of IrLawOp; LAW_UNIFORM
// pushes slot `fn` of the frame row, where slot 0 is the wrapped time the shader calls `time`).
struct IrDerived { std::vector<IrLawOp> prog; std::string text; };

// Host-side animation of one user uniform: `name: base ~ modulation <> 50ms >> 200ms .. [0, 1]`
// (temporal.rs:90-176). The modulation is a frame-level program (prog empty: none, or it reads something that
// is not a frame constant); ops run left to right on the modulated value.
struct IrTemporal { uint8_t kind = 0; double secs = 0.0, lo = 0.0, hi = 0.0; };   // kind: TemporalOp::Kind
struct IrParamAnim { int idx = -1; std::string name; std::vector<IrLawOp> mod; std::string mod_text; std::vector<IrTemporal> ops; };
// `source -> target.field * weight` (resonance.rs:32-61): params[idx] += signal(source) * weight * dt
struct IrResonance { int idx = -1; std::string source, target; double weight = 0.0; };

struct Ir {
    std::string name;
    bool multi_layer = false;     // cinematic.layers.len() > 1, params layers included (glsl.rs:66)
    uint32_t flags = 0;
    std::vector<IrLayer> layers;
    std::vector<IrStage> stages;
    std::vector<UniformInfo> uniforms;
    IrGravity gravity;
    bool uses_memory = false;
    std::vector<IrArc> arcs;
    std::vector<IrDerived> derived;
    std::vector<IrDerived> pixel_exprs;   // programs that also use LAW_PX / LAW_PY / LAW_UVX / LAW_UVY
    std::vector<IrParamAnim> anims;
    std::vector<IrResonance> resonances;
    IrProject project;
};

// Value of a frame-level program over one frame row, binary64. wrap_time: `time` is the shader's wrapped f32
// time (stage arguments); otherwise the raw u_time of the row (host-side modulation, as JS sees it).
double eval_frame_prog(const std::vector<IrLawOp>& prog, const float* row, bool wrap_time);

// Per-context state of the temporal operators (the JS classes of temporal.rs:31-82) plus the frame step that
// applies arcs, modulation + temporal operators and resonance to the user uniforms of a frame row.
struct Animator {
    struct OpState { std::vector<float> ring; size_t head = 0; bool has_value = false; double value = 0.0, prev = 0.0, envelope = 0.0; };
    const Ir* ir = nullptr;
    double fps = 60.0;
    std::vector<std::vector<OpState>> state;   // [anim][op]
    void init(const Ir& ir_, double fps_);
    void reset();
    void step(double time_s, double dt, float* row, uint32_t what, const char* const* names, const double* values, uint32_t nsignals);
};

// Values of Ir::derived for one frame row (10 built-ins + user uniforms), f32 arithmetic.
void eval_derived(const Ir& ir, const float* row, float* out);

// arcUpdate(time, params) of runtime/arc.rs:110-122 in JS number (double) arithmetic: applies the timeline to
// `params` (one value per uniform, in uniform order).
void arc_update(const Ir& ir, double time, double* params);
// JS-number value of a constant expression as expr.rs:147 compile_js would word it; false when it is not one.
bool eval_js_const(const Expr& e, double& out);

Ir lower_cinematic(const Cinematic& c);                  // after validate; never fails
std::vector<uint8_t> ir_serialize(const Ir& ir);         // flat little-endian blob (layout: include/game_hip.h)
std::string ir_to_json(const Ir& ir);
// The inverse of ir_serialize, for hosts that bring their own resolved IR: every index, count, string and program of the
// blob is checked (a damaged or hostile blob is refused with `why`, never read out of bounds). What a blob does not carry
// (layer names, source texts, arcs, modulation) is empty in the result.
bool ir_deserialize(const uint8_t* data, size_t len, Ir& ir, std::string& why);

// Whole-source entry point (lib.rs:117 up to and including codegen::generate's validation).
struct CompiledProgram {
    Program ast;
    std::vector<Ir> irs;                 // one per cinematic
    std::vector<std::vector<uint8_t>> blobs;
    std::vector<std::string> jsons;
    std::vector<std::string> warnings;   // check() output
};
CompileError compile_source(const std::string& src, CompiledProgram& out);
// resolver.rs:18: drains the file imports of `program`, merging the imported defines into its first cinematic.
CompileError resolve_imports(Program& program, const std::string& base_dir, const std::vector<std::string>& lib_dirs);
// compile_source with the file imports resolved between parsing and define expansion.
CompileError compile_source_with_imports(const std::string& src, const std::string& base_dir,
                                         const std::vector<std::string>& lib_dirs, CompiledProgram& out);

}  // namespace game
