// Lowering of a validated cinematic to the flat IR, its serialised forms, and the whole-source
// entry point. The IR states, per stage, exactly what the reference's GLSL emitter would have
// pasted for each argument (game-compiler/src/codegen/glsl.rs:261-531 reads its arguments through
// stages.rs:25 `get_arg` with the parameter lists of builtins.rs:43-217).
#include "front.hpp"
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>

namespace game {
namespace {

// palette names -> table index (glsl.rs:424-434); anything else is the default (rainbow, index 2)
uint32_t palette_index(const std::string& name)
{
    static const char* names[] = {"fire", "ice", "rainbow", "ocean", "forest", "neon", "sunset", "plasma"};
    for (uint32_t i = 0; i < 8; ++i) if (name == names[i]) return i;
    return 2;
}

// Variables each stage declares at layer scope in the emitted GLSL (glsl.rs:261-531); stages that
// wrap their text in `{ }` (star, polygon, rotate, ...) declare nothing visible afterwards.
const char* const* declared_names(uint16_t id)
{
    static const char* const none[] = {nullptr};
    static const char* const sdf[] = {"sdf_result", nullptr};
    static const char* const box[] = {"box_d", "sdf_result", nullptr};
    static const char* const cw[] = {"cw_dist", "sdf_result", nullptr};
    static const char* const arc[] = {"arc_theta", nullptr};
    static const char* const glow[] = {"glow_result", "color_result", nullptr};
    static const char* const shade[] = {"aa", "color_result", nullptr};
    static const char* const pal[] = {"pal_t", "color_result", nullptr};
    static const char* const bloom[] = {"pp_lum", nullptr};
    static const char* const grain[] = {"grain_noise", nullptr};
    static const char* const vig[] = {"vig_dist", "vig_factor", nullptr};
    static const char* const scan[] = {"scan_val", nullptr};
    static const char* const grad[] = {"grad_t", "color_result", nullptr};
    static const char* const spec[] = {"spec_x", "spec_r", "spec_g", "spec_b", "color_result", nullptr};
    switch (id) {
    case B_CIRCLE: case B_RING: case B_SIMPLEX: case B_VORONOI: case B_FBM: return sdf;
    case B_BOX: return box;
    case B_CONCENTRIC_WAVES: return cw;
    case B_MASK_ARC: return arc;
    case B_GLOW: case B_EMISSIVE: return glow;
    case B_SHADE: return shade;
    case B_PALETTE: return pal;
    case B_BLOOM: return bloom;
    case B_GRAIN: return grain;
    case B_VIGNETTE: return vig;
    case B_SCANLINES: return scan;
    case B_GRADIENT: return grad;
    case B_SPECTRUM: return spec;
    default: return none;
    }
}

float text_to_f32(const std::string& text) { return strtof(text.c_str(), nullptr); }

LawFn law_fn(const std::string& n, size_t nargs, bool& ok)
{
    struct E { const char* n; LawFn f; size_t a; };
    static const E tab[] = {
        {"sin", LF_SIN, 1}, {"cos", LF_COS, 1}, {"tan", LF_TAN, 1}, {"abs", LF_ABS, 1}, {"sqrt", LF_SQRT, 1},
        {"exp", LF_EXP, 1}, {"log", LF_LOG, 1}, {"floor", LF_FLOOR, 1}, {"ceil", LF_CEIL, 1}, {"fract", LF_FRACT, 1},
        {"sign", LF_SIGN, 1}, {"exp2", LF_EXP2, 1}, {"log2", LF_LOG2, 1}, {"round", LF_ROUND, 1},
        {"min", LF_MIN, 2}, {"max", LF_MAX, 2}, {"pow", LF_POW, 2}, {"step", LF_STEP, 2}, {"mod", LF_MOD, 2}, {"atan2", LF_ATAN2, 2},
        {"clamp", LF_CLAMP, 3}, {"mix", LF_MIX, 3}, {"smoothstep", LF_SMOOTHSTEP, 3},
    };
    for (const E& e : tab) if (n == e.n && nargs == e.a) { ok = true; return e.f; }
    ok = false;
    return LF_SIN;
}

void lower_law(const Expr& e, Ir& ir)
{
    auto push = [&](uint32_t op, float v = 0.0f, uint32_t fn = 0) {
        IrLawOp o;
        o.op = op; o.val = v; o.fn = fn;
        ir.gravity.law.push_back(o);
    };
    switch (e.kind) {
    case Expr::Number: push(LAW_NUM, text_to_f32(fmt_fixed6(e.num))); return;
    case Expr::Ident:
        if (e.s == "dist") push(LAW_DIST);
        else if (e.s == "time") push(LAW_TIME);
        else { ir.flags |= IRF_GRAVITY_IDENT; push(LAW_NUM, 0.0f); }
        return;
    case Expr::Bin: {
        lower_law(e.kids[0], ir);
        lower_law(e.kids[1], ir);
        static const uint32_t ops[] = {LAW_ADD, LAW_SUB, LAW_MUL, LAW_DIV, LAW_POW, LAW_GT, LAW_LT};
        push(ops[(int)e.op]);
        return;
    }
    case Expr::Neg: lower_law(e.kids[0], ir); push(LAW_NEG); return;
    case Expr::Paren: lower_law(e.kids[0], ir); return;
    case Expr::Call: {
        bool ok;
        LawFn f = law_fn(e.s, e.args.size(), ok);
        if (!ok) { ir.flags |= IRF_GRAVITY_IDENT; push(LAW_NUM, 0.0f); return; }
        for (const Arg& a : e.args) lower_law(a.value, ir);
        push(e.args.size() == 1 ? LAW_CALL1 : e.args.size() == 2 ? LAW_CALL2 : LAW_CALL3, 0.0f, f);
        return;
    }
    default: push(LAW_NUM, 1.0f); return;   // gravity.rs:39 fallback
    }
}

// The expression as the source would spell it (for the IR's JSON view and diagnostics).
std::string expr_source_text(const Expr& e)
{
    switch (e.kind) {
    case Expr::Number: return fmt_fixed6(e.num);
    case Expr::Ident: return e.s;
    case Expr::Dotted: return e.s + "." + e.field;
    case Expr::Paren: return "(" + expr_source_text(e.kids[0]) + ")";
    case Expr::Array: {
        std::string t = "[";
        for (size_t i = 0; i < e.kids.size(); ++i) t += (i ? ", " : "") + expr_source_text(e.kids[i]);
        return t + "]";
    }
    case Expr::Neg: return "-" + expr_source_text(e.kids[0]);
    case Expr::Bin: {
        static const char* const ops[] = {" + ", " - ", " * ", " / ", " ^ ", " > ", " < "};
        return "(" + expr_source_text(e.kids[0]) + ops[(int)e.op] + expr_source_text(e.kids[1]) + ")";
    }
    case Expr::Call: {
        std::string t = e.s + "(";
        for (size_t i = 0; i < e.args.size(); ++i) t += (i ? ", " : "") + expr_source_text(e.args[i].value);
        return t + ")";
    }
    case Expr::Ternary:
        return "(" + expr_source_text(e.kids[0]) + " ? " + expr_source_text(e.kids[1]) + " : " + expr_source_text(e.kids[2]) + ")";
    default: return "?";
    }
}

// A lowered (sub)expression: one postfix program per component - one for a scalar, two for a vec2, three for a
// colour. Vector values exist only while lowering: `p`, `uv`, the nineteen named colours (expr.rs:13-36), two- and
// three-element arrays and the operators / functions over them (expr.rs:50-144 words them as WGSL vec2f / vec3f
// arithmetic) are expanded component by component, so what reaches the IR is always a scalar program. length(v)
// of a vec2 becomes the two-argument function length2(x, y); dot, distance and normalize expand to their defining
// sums, differences and quotients.
struct LoweredValue {
    int dim = 1;
    std::vector<IrLawOp> c[3];
    bool vec() const { return dim > 1; }
    const std::vector<IrLawOp>& comp(int k) const { return dim > 1 ? c[k] : c[0]; }   // a scalar is broadcast
};

void append(std::vector<IrLawOp>& dst, const std::vector<IrLawOp>& src) { dst.insert(dst.end(), src.begin(), src.end()); }
IrLawOp law_op(uint32_t op, float v = 0.0f, uint32_t fn = 0) { IrLawOp o; o.op = op; o.val = v; o.fn = fn; return o; }

// The named colours of expr.rs:13-36 (`vec3f(r, g, b)`).
const float* named_color(const std::string& n)
{
    static const struct { const char* n; float v[3]; } tab[] = {
        {"black", {0.0f, 0.0f, 0.0f}}, {"white", {1.0f, 1.0f, 1.0f}}, {"red", {1.0f, 0.0f, 0.0f}}, {"green", {0.0f, 1.0f, 0.0f}},
        {"blue", {0.0f, 0.0f, 1.0f}}, {"gold", {0.831f, 0.686f, 0.216f}}, {"midnight", {0.039f, 0.039f, 0.118f}},
        {"obsidian", {0.071f, 0.059f, 0.082f}}, {"ember", {0.898f, 0.318f, 0.129f}}, {"cyan", {0.0f, 1.0f, 1.0f}},
        {"ivory", {1.0f, 1.0f, 0.941f}}, {"frost", {0.686f, 0.878f, 0.953f}}, {"orange", {1.0f, 0.647f, 0.0f}},
        {"deep_blue", {0.0f, 0.098f, 0.392f}}, {"ash", {0.467f, 0.467f, 0.467f}}, {"charcoal", {0.212f, 0.212f, 0.212f}},
        {"plasma", {0.580f, 0.0f, 0.827f}}, {"violet", {0.541f, 0.169f, 0.886f}}, {"magenta", {1.0f, 0.0f, 1.0f}}};
    for (auto& k : tab) if (n == k.n) return k.v;
    return nullptr;
}

// Two operand shapes combine when they are equal or one of them is a scalar; the result has the larger one.
bool combine_dims(int a, int b, int& out)
{
    if (a != b && a != 1 && b != 1) return false;
    out = a > b ? a : b;
    return true;
}

bool lower_value(const Expr& e, const std::map<std::string, uint32_t>& uslot, bool allow_pixel, LoweredValue& out)
{
    out = LoweredValue{};
    auto rec = [&](const Expr& k, LoweredValue& v) { return lower_value(k, uslot, allow_pixel, v); };
    switch (e.kind) {
    case Expr::Number: out.c[0].push_back(law_op(LAW_NUM, (float)e.num)); return true;
    case Expr::Dur:
        out.c[0].push_back(law_op(LAW_NUM, (float)(e.dur.kind == DurKind::Seconds ? e.dur.v : e.dur.kind == DurKind::Millis ? e.dur.v / 1000.0 : (double)e.dur.bars * 2.0)));
        return true;
    case Expr::Ident: {
        if (e.s == "time") { out.c[0].push_back(law_op(LAW_UNIFORM, 0.0f, BU_TIME)); return true; }
        if (const float* col = named_color(e.s)) {     // expr.rs:60-67: a colour name first, then a constant, then the name itself
            out.dim = 3;
            for (int k = 0; k < 3; ++k) out.c[k].push_back(law_op(LAW_NUM, col[k]));
            return true;
        }
        static const struct { const char* n; float v; } consts[] = {   // expr.rs:39-47
            {"pi", 3.14159265358979f}, {"tau", 6.28318530717959f}, {"e", 2.71828182845905f}, {"phi", 1.61803398874989f}};
        for (auto& k : consts) if (e.s == k.n) { out.c[0].push_back(law_op(LAW_NUM, k.v)); return true; }
        auto it = uslot.find(e.s);
        if (it != uslot.end()) { out.c[0].push_back(law_op(LAW_UNIFORM, 0.0f, BU_COUNT + it->second)); return true; }
        if (allow_pixel && (e.s == "p" || e.s == "uv")) {   // the position / the normalised pixel coordinate as vec2
            out.dim = 2;
            out.c[0].push_back(law_op(e.s == "p" ? LAW_PX : LAW_UVX));
            out.c[1].push_back(law_op(e.s == "p" ? LAW_PY : LAW_UVY));
            return true;
        }
        return false;
    }
    case Expr::Dotted: {
        static const struct { const char* o; const char* f; uint32_t slot; } dotted[] = {
            {"audio", "bass", BU_AUDIO_BASS}, {"audio", "mid", BU_AUDIO_MID}, {"audio", "treble", BU_AUDIO_TREBLE},
            {"audio", "energy", BU_AUDIO_ENERGY}, {"audio", "beat", BU_AUDIO_BEAT}, {"mouse", "x", BU_MOUSE_X}, {"mouse", "y", BU_MOUSE_Y}};
        for (auto& d : dotted) if (e.s == d.o && e.field == d.f) { out.c[0].push_back(law_op(LAW_UNIFORM, 0.0f, d.slot)); return true; }
        if (allow_pixel && (e.s == "p" || e.s == "uv") && (e.field == "x" || e.field == "y")) {
            out.c[0].push_back(law_op(e.s == "p" ? (e.field == "x" ? LAW_PX : LAW_PY) : (e.field == "x" ? LAW_UVX : LAW_UVY)));
            return true;
        }
        return false;
    }
    case Expr::Array: {   // [a, b] -> vec2f(a, b), [a, b, c] -> vec3f(a, b, c) (expr.rs:70-81); longer arrays have no use here
        if (e.kids.size() != 2 && e.kids.size() != 3) return false;
        out.dim = (int)e.kids.size();
        for (int k = 0; k < out.dim; ++k) {
            LoweredValue a;
            if (!rec(e.kids[(size_t)k], a) || a.vec()) return false;
            out.c[k] = a.c[0];
        }
        return true;
    }
    case Expr::Paren: return rec(e.kids[0], out);
    case Expr::Neg: {
        if (!rec(e.kids[0], out)) return false;
        for (int k = 0; k < out.dim; ++k) out.c[k].push_back(law_op(LAW_NEG));
        return true;
    }
    case Expr::Bin: {
        LoweredValue l, r;
        if (!rec(e.kids[0], l) || !rec(e.kids[1], r)) return false;
        static const uint32_t ops[] = {LAW_ADD, LAW_SUB, LAW_MUL, LAW_DIV, LAW_POW, LAW_GT, LAW_LT};
        const uint32_t op = ops[(int)e.op];
        if (!combine_dims(l.dim, r.dim, out.dim)) return false;
        if (out.vec() && (op == LAW_GT || op == LAW_LT)) return false;   // a vector of bools has no scalar meaning as an argument
        for (int k = 0; k < out.dim; ++k) {                              // a scalar operand is broadcast
            append(out.c[k], l.comp(k)); append(out.c[k], r.comp(k)); out.c[k].push_back(law_op(op));
        }
        return true;
    }
    case Expr::Ternary: {
        // expr.rs:132-142 words `c ? t : f` as select(f, t, c). The condition is the 0 / 1 a comparison leaves (or any
        // scalar: non-zero and not NaN counts as true, the JavaScript wording's truthiness); both branches are
        // evaluated, as select does.
        LoweredValue cnd, t, f;
        if (!rec(e.kids[0], cnd) || cnd.vec() || !rec(e.kids[1], t) || !rec(e.kids[2], f)) return false;
        if (!combine_dims(t.dim, f.dim, out.dim)) return false;
        for (int k = 0; k < out.dim; ++k) {
            append(out.c[k], cnd.c[0]); append(out.c[k], t.comp(k)); append(out.c[k], f.comp(k));
            out.c[k].push_back(law_op(LAW_CALL3, 0.0f, LF_SELECT));
        }
        return true;
    }
    case Expr::Call: {
        std::vector<LoweredValue> args(e.args.size());
        int dim = 1;
        for (size_t i = 0; i < e.args.size(); ++i) {
            if (!rec(e.args[i].value, args[i])) return false;
            if (!combine_dims(dim, args[i].dim, dim)) return false;
        }
        const bool any_vec = dim > 1;
        auto sub = [&](const LoweredValue& a, const LoweredValue& b, int k, std::vector<IrLawOp>& dst) {
            append(dst, a.comp(k)); append(dst, b.comp(k)); dst.push_back(law_op(LAW_SUB));
        };
        if (e.s == "length" && args.size() == 1) {
            if (dim == 3) return false;
            if (!any_vec) { out.c[0] = args[0].c[0]; out.c[0].push_back(law_op(LAW_CALL1, 0.0f, LF_ABS)); return true; }
            append(out.c[0], args[0].c[0]); append(out.c[0], args[0].c[1]); out.c[0].push_back(law_op(LAW_CALL2, 0.0f, LF_LENGTH2));
            return true;
        }
        if (e.s == "distance" && args.size() == 2) {
            if (dim == 3) return false;
            if (!any_vec) { sub(args[0], args[1], 0, out.c[0]); out.c[0].push_back(law_op(LAW_CALL1, 0.0f, LF_ABS)); return true; }
            sub(args[0], args[1], 0, out.c[0]); sub(args[0], args[1], 1, out.c[0]); out.c[0].push_back(law_op(LAW_CALL2, 0.0f, LF_LENGTH2));
            return true;
        }
        if (e.s == "dot" && args.size() == 2) {   // the products summed left to right
            for (int k = 0; k < dim; ++k) {
                append(out.c[0], args[0].comp(k)); append(out.c[0], args[1].comp(k)); out.c[0].push_back(law_op(LAW_MUL));
                if (k) out.c[0].push_back(law_op(LAW_ADD));
            }
            return true;
        }
        if (e.s == "normalize" && args.size() == 1 && dim == 2) {   // v * (1 / length(v))
            out.dim = 2;
            for (int k = 0; k < 2; ++k) {
                std::vector<IrLawOp>& dst = out.c[k];
                append(dst, args[0].comp(k));
                dst.push_back(law_op(LAW_NUM, 1.0f));
                append(dst, args[0].c[0]); append(dst, args[0].c[1]); dst.push_back(law_op(LAW_CALL2, 0.0f, LF_LENGTH2));
                dst.push_back(law_op(LAW_DIV));
                dst.push_back(law_op(LAW_MUL));
            }
            return true;
        }
        bool ok;
        LawFn f = law_fn(e.s, e.args.size(), ok);
        if (!ok) return false;
        out.dim = dim;                                                  // component-wise, scalar arguments broadcast
        for (int k = 0; k < out.dim; ++k) {
            for (const LoweredValue& a : args) append(out.c[k], a.comp(k));
            out.c[k].push_back(law_op(args.size() == 1 ? LAW_CALL1 : args.size() == 2 ? LAW_CALL2 : LAW_CALL3, 0.0f, f));
        }
        return true;
    }
    default: return false;
    }
}

// Frame-level scalar expression -> RPN; false when it reads something that is not a frame constant
// (`p`, `uv` unless allow_pixel, an unknown name, a string), calls an unknown function, or is vector-valued.
bool lower_frame_expr(const Expr& e, const std::map<std::string, uint32_t>& uslot, std::vector<IrLawOp>& prog, bool allow_pixel = false)
{
    LoweredValue v;
    if (!lower_value(e, uslot, allow_pixel, v) || v.vec()) return false;
    append(prog, v.c[0]);
    return true;
}

}  // namespace

// Arithmetic is binary64 with the C library's functions. Two callers, two wordings of the same expression language:
// wrap_time = true evaluates a stage argument as the shader would (expr.rs:50-144, the WGSL wording: `time` is the
// wrapped f32 the shader sees, round() goes to the nearest integer with ties to even); wrap_time = false evaluates a
// modulation as the page's JavaScript would (expr.rs:147-260: raw u_time, Math.round = floor(x + 0.5)). `%` is the
// exact truncated remainder in both.
double eval_frame_prog(const std::vector<IrLawOp>& prog, const float* row, bool wrap_time)
{
    auto call = [wrap_time](uint32_t fn, double a, double b, double c) -> double {
        switch (fn) {
        case LF_SIN: return sin(a); case LF_COS: return cos(a); case LF_TAN: return tan(a); case LF_ABS: return fabs(a);
        case LF_SQRT: return sqrt(a); case LF_EXP: return exp(a); case LF_LOG: return log(a); case LF_FLOOR: return floor(a);
        case LF_CEIL: return ceil(a); case LF_FRACT: return a - floor(a); case LF_SIGN: return a > 0.0 ? 1.0 : a < 0.0 ? -1.0 : 0.0;
        case LF_EXP2: return exp2(a); case LF_LOG2: return log2(a); case LF_ROUND: return wrap_time ? nearbyint(a) : floor(a + 0.5);
        case LF_MOD: return fmod(a, b); case LF_ATAN2: return atan2(a, b); case LF_LENGTH2: return sqrt(a * a + b * b);
        case LF_MIN: return a < b ? a : b; case LF_MAX: return a > b ? a : b; case LF_POW: return pow(a, b);
        case LF_STEP: return b < a ? 0.0 : 1.0;
        case LF_CLAMP: { double lo = a > b ? a : b; return lo < c ? lo : c; }
        case LF_MIX: return a * (1.0 - c) + b * c;
        case LF_SMOOTHSTEP: { double t = (c - a) / (b - a); t = t < 0.0 ? 0.0 : t > 1.0 ? 1.0 : t; return t * t * (3.0 - 2.0 * t); }
        case LF_SELECT: return (a != 0.0 && a == a) ? b : c;
        default: return 0.0;
        }
    };
    double st[64];
    int sp = 0;
    for (const IrLawOp& o : prog) {
        switch (o.op) {
        case LAW_NUM: st[sp++] = (double)o.val; break;
        case LAW_UNIFORM: {
            float v = row[o.fn];
            if (o.fn == BU_TIME && wrap_time) {   // glsl.rs:56
                float q = v / 120.0f, f = q - floorf(q);
                if (f == f && !(f < 1.0f)) f = 0x1.fffffep-1f;
                v = f * 120.0f;
            }
            st[sp++] = (double)v;
            break;
        }
        case LAW_ADD: sp--; st[sp - 1] = st[sp - 1] + st[sp]; break;
        case LAW_SUB: sp--; st[sp - 1] = st[sp - 1] - st[sp]; break;
        case LAW_MUL: sp--; st[sp - 1] = st[sp - 1] * st[sp]; break;
        case LAW_DIV: sp--; st[sp - 1] = st[sp - 1] / st[sp]; break;
        case LAW_POW: sp--; st[sp - 1] = pow(st[sp - 1], st[sp]); break;
        case LAW_GT: sp--; st[sp - 1] = st[sp - 1] > st[sp] ? 1.0 : 0.0; break;
        case LAW_LT: sp--; st[sp - 1] = st[sp - 1] < st[sp] ? 1.0 : 0.0; break;
        case LAW_NEG: st[sp - 1] = -st[sp - 1]; break;
        case LAW_CALL1: st[sp - 1] = call(o.fn, st[sp - 1], 0.0, 0.0); break;
        case LAW_CALL2: sp--; st[sp - 1] = call(o.fn, st[sp - 1], st[sp], 0.0); break;
        case LAW_CALL3: sp -= 2; st[sp - 1] = call(o.fn, st[sp - 1], st[sp], st[sp + 1]); break;
        default: break;
        }
    }
    return sp >= 1 ? st[sp - 1] : 0.0;
}

// Stage-argument values: `time` is the f32 value the shader itself sees (u_time wrapped to [0,120));
// each result is rounded to f32 once.
void eval_derived(const Ir& ir, const float* row, float* out)
{
    for (size_t d = 0; d < ir.derived.size(); ++d) out[d] = (float)eval_frame_prog(ir.derived[d].prog, row, true);
}

// ── temporal operators, modulation, resonance ────────────────────────────
void Animator::init(const Ir& ir_, double fps_)
{
    ir = &ir_;
    fps = fps_ > 0.0 ? fps_ : 60.0;   // the JS classes default to 60 (temporal.rs:33,52,69)
    reset();
}

void Animator::reset()
{
    state.assign(ir->anims.size(), {});
    for (size_t i = 0; i < ir->anims.size(); ++i) {
        state[i].resize(ir->anims[i].ops.size());
        for (size_t k = 0; k < ir->anims[i].ops.size(); ++k) {
            const IrTemporal& op = ir->anims[i].ops[k];
            if (op.kind == TemporalOp::Delay) {   // GameRingBuffer: max(1, ceil(duration * fps)) zeroed f32 slots
                double n = std::ceil(op.secs * fps);
                if (!(n >= 1.0)) n = 1.0;
                if (n > 16777216.0) n = 16777216.0;
                state[i][k].ring.assign((size_t)n, 0.0f);
            }
        }
    }
}

void Animator::step(double time_s, double dt, float* row, uint32_t what, const char* const* names, const double* values, uint32_t nsignals)
{
    if (what == 0) what = 7u;
    float* user = row + BU_COUNT;
    if ((what & 1u) && !ir->arcs.empty()) {
        std::vector<double> params(ir->uniforms.size());
        for (size_t i = 0; i < params.size(); ++i) params[i] = user[i];
        arc_update(*ir, time_s, params.data());
        for (size_t i = 0; i < params.size(); ++i) user[i] = (float)params[i];
    }
    if (what & 2u) {
        // every modulation reads the row as it stood before this pass, so the order of the parameters does not matter
        std::vector<double> mods(ir->anims.size(), 0.0);
        for (size_t i = 0; i < ir->anims.size(); ++i)
            if (!ir->anims[i].mod.empty()) mods[i] = eval_frame_prog(ir->anims[i].mod, row, false);
        for (size_t i = 0; i < ir->anims.size(); ++i) {
            const IrParamAnim& pa = ir->anims[i];
            if (pa.idx < 0) continue;
            double v = (double)user[pa.idx] + mods[i];
            for (size_t k = 0; k < pa.ops.size(); ++k) {
                const IrTemporal& op = pa.ops[k];
                OpState& s = state[i][k];
                switch (op.kind) {
                case TemporalOp::Delay:        // push, then read the slot the head now points at (temporal.rs:38-44)
                    s.ring[s.head] = (float)v;
                    s.head = (s.head + 1) % s.ring.size();
                    v = (double)s.ring[s.head];
                    break;
                case TemporalOp::Smooth: {     // temporal.rs:52-61
                    double samples = op.secs * fps;
                    if (!(samples >= 1.0)) samples = 1.0;
                    const double alpha = 2.0 / (samples + 1.0);
                    if (!s.has_value) { s.has_value = true; s.value = v; }
                    else s.value += alpha * (v - s.value);
                    v = s.value;
                    break;
                }
                case TemporalOp::Trigger: {    // temporal.rs:69-80
                    double n = op.secs * fps;
                    if (!(n >= 1.0)) n = 1.0;
                    const double delta = v - s.prev;
                    s.prev = v;
                    if (delta > 0.01) s.envelope = 1.0;
                    else { s.envelope -= 1.0 / n; if (!(s.envelope > 0.0)) s.envelope = 0.0; }
                    v = s.envelope;
                    break;
                }
                default:                       // Range: Math.min(Math.max(v, lo), hi)
                    if (v == v) {              // Math.max / Math.min keep a NaN
                        v = v > op.lo ? v : op.lo;
                        v = v < op.hi ? v : op.hi;
                    }
                    break;
                }
            }
            user[pa.idx] = (float)v;
        }
    }
    if (what & 4u) {
        for (const IrResonance& r : ir->resonances) {
            if (r.idx < 0) continue;
            double sig = 0.0;                  // signals[source] || 0
            for (uint32_t i = 0; i < nsignals; ++i)
                if (names && values && names[i] && r.source == names[i]) { sig = values[i]; break; }
            if (sig != sig) sig = 0.0;
            user[r.idx] = (float)((double)user[r.idx] + sig * r.weight * dt);
        }
    }
}

Ir lower_cinematic(const Cinematic& c)
{
    Ir ir;
    ir.name = c.name;
    ir.multi_layer = c.layers.size() > 1;
    ir.uniforms = extract_uniforms(c);
    std::map<std::string, uint32_t> uslot;
    for (uint32_t i = 0; i < ir.uniforms.size(); ++i) uslot[ir.uniforms[i].name] = i;

    std::set<std::string> declared_params;   // names the cinematic declares as layer parameters
    for (const Layer& l : c.layers) {
        if (!l.is_pipeline) for (const Param& p : l.params) declared_params.insert(p.name);
        for (const Param& p : l.opts) declared_params.insert(p.name);
    }

    // one stage call -> builtin id + resolved arguments (stages.rs:8-42)
    auto lower_stage = [&](const Stage& s, const Builtin* b) {
        IrStage is;
        is.builtin = b->id;
        is.name = b->name;
        is.nargs = b->nparams;
        for (uint32_t p = 0; p < b->nparams; ++p) {
            ArgText t = get_arg(s.args, b->params[p].name, p, b->name);
            IrArg& a = is.args[p];
            a.text = t.text;
            if (t.kind == ArgText::Literal) {
                a.kind = ARG_LIT;
                a.lit = text_to_f32(t.text);
            } else if (t.kind == ArgText::Ident) {
                if (b->id == B_PALETTE) {          // the name is matched as text, never evaluated
                    a.kind = ARG_LIT; a.lit = 0.0f;
                } else if (t.text == "time") {
                    a.kind = ARG_BUILTIN; a.slot = BU_TIME;
                    ir.flags |= IRF_TIME_IDENT;
                } else if (uslot.count(t.text)) {
                    a.kind = ARG_USER; a.slot = uslot[t.text];
                } else {
                    a.kind = ARG_LIT; a.lit = 0.0f;
                    ir.flags |= IRF_BUILTIN_IDENT;
                }
            } else {
                static const struct { const char* n; uint32_t slot; } dotted[] = {
                    {"audio_bass", BU_AUDIO_BASS}, {"audio_mid", BU_AUDIO_MID}, {"audio_treble", BU_AUDIO_TREBLE},
                    {"audio_energy", BU_AUDIO_ENERGY}, {"audio_beat", BU_AUDIO_BEAT},
                    {"mouse_x", BU_MOUSE_X}, {"mouse_y", BU_MOUSE_Y},
                };
                bool found = false;
                for (auto& d : dotted) if (t.text == d.n) { a.kind = ARG_BUILTIN; a.slot = d.slot; found = true; }
                if (b->id == B_PALETTE) { a.kind = ARG_LIT; a.lit = 0.0f; found = true; }
                else ir.flags |= IRF_DOTTED_ARG;
                if (!found) { a.kind = ARG_LIT; a.lit = 0.0f; ir.flags |= IRF_UNKNOWN_DOTTED; }
            }
            // an expression the emitters drop: keep it as a frame-level program when it is one
            // (not for palette names or the fbm octave count, which must stay compile-time values)
            if (t.dropped && b->id != B_PALETTE && !(b->id == B_FBM && p == 1)) {
                IrDerived d;
                if (lower_frame_expr(*t.dropped, uslot, d.prog) && d.prog.size() < 48 && ir.derived.size() < 0xfffe) {
                    d.text = expr_source_text(*t.dropped);
                    a.derived = (int)ir.derived.size();
                    ir.derived.push_back(std::move(d));
                } else {
                    d.prog.clear();   // not a frame constant: maybe a function of the pixel as well
                    if (lower_frame_expr(*t.dropped, uslot, d.prog, true) && d.prog.size() < 120 && ir.pixel_exprs.size() < 0xfffe) {
                        d.text = expr_source_text(*t.dropped);
                        a.pixel = (int)ir.pixel_exprs.size();
                        ir.pixel_exprs.push_back(std::move(d));
                    }
                }
            }
        }
        // A colour where three scalars are expected (`tint(gold)`, `shade(mix(ember, frost, audio.bass))`,
        // `tint(uv.x > 0.0 ? gold : [0.2, 0.3, 0.9])`): the stage's only argument lands on `r` of an (r, g, b)
        // parameter list and is vec3-valued by the expression compiler's wording (expr.rs:60-81: colour names and
        // three-element arrays are vec3f). The reference's emitters see one scalar there - a name becomes a uniform
        // that defaults to 0, anything else the default of `r` - and that stays the resolved argument; the three
        // components are kept beside it as expression values, which a renderer that animates arguments reads instead.
        // A name the cinematic declares as a parameter is that parameter, not a colour.
        if ((b->id == B_TINT || b->id == B_SHADE) && s.args.size() == 1 && (!s.args[0].named || s.args[0].name == "r")) {
            const Expr& ex = s.args[0].value;
            const bool bare_colour = ex.kind == Expr::Ident && named_color(ex.s) && !declared_params.count(ex.s);
            if (bare_colour || (ex.kind != Expr::Number && ex.kind != Expr::Ident && ex.kind != Expr::Dotted)) {
                LoweredValue v;
                if (lower_value(ex, uslot, false, v) && v.dim == 3 && v.c[0].size() < 48 && v.c[1].size() < 48 && v.c[2].size() < 48 &&
                    ir.derived.size() + 3 < 0xfffe) {
                    for (int k = 0; k < 3; ++k) {
                        IrDerived d;
                        d.prog = v.c[k];
                        d.text = expr_source_text(ex) + "." + "rgb"[k];
                        is.args[k].derived = (int)ir.derived.size();
                        ir.derived.push_back(std::move(d));
                    }
                } else if (lower_value(ex, uslot, true, v) && v.dim == 3 && v.c[0].size() < 120 && v.c[1].size() < 120 && v.c[2].size() < 120 &&
                           ir.pixel_exprs.size() + 3 < 0xfffe) {
                    for (int k = 0; k < 3; ++k) {
                        IrDerived d;
                        d.prog = v.c[k];
                        d.text = expr_source_text(ex) + "." + "rgb"[k];
                        is.args[k].pixel = (int)ir.pixel_exprs.size();
                        ir.pixel_exprs.push_back(std::move(d));
                    }
                }
            }
        }
        if (b->id == B_PALETTE) is.aux = palette_index(is.args[0].text);
        return is;
    };

    bool any_pipeline = false;
    for (const Layer& l : c.layers) {
        IrLayer il;
        il.name = l.name;
        il.is_pipeline = l.is_pipeline;
        il.has_memory = l.has_memory;
        if (l.has_memory) {
            ir.uses_memory = true;
            il.decay_text = fmt_fixed6(l.memory);
            il.decay = text_to_f32(il.decay_text);
        }
        il.stage_begin = (uint32_t)ir.stages.size();
        if (l.is_pipeline) {
            any_pipeline = true;
            State st = State::Position;
            std::set<std::string> declared;   // names visible at layer scope in the emitted GLSL
            for (const Stage& s : l.stages) {
                const Builtin* b = builtin_lookup(s.name);
                if (!b) continue;   // unreachable after validation
                IrStage is = lower_stage(s, b);
                // GLSL scoping facts of the emitted text (glsl.rs:261-531)
                if (b->input == State::Sdf && !declared.count("sdf_result")) ir.flags |= IRF_COLOR_BEFORE_DECL;
                if (b->input == State::Color && !declared.count("color_result")) ir.flags |= IRF_COLOR_BEFORE_DECL;
                if (b->id == B_STAR || b->id == B_POLYGON) ir.flags |= IRF_BLOCK_SCOPED_SDF;
                for (const char* const* d = declared_names(b->id); *d; ++d)
                    if (!declared.insert(*d).second) ir.flags |= IRF_REDECLARED;
                st = b->output;
                ir.stages.push_back(std::move(is));
            }
            il.end_state = st;
            if (st != State::Color) ir.flags |= IRF_NO_COLOR;
        }
        il.stage_count = (uint32_t)ir.stages.size() - il.stage_begin;
        ir.layers.push_back(std::move(il));
    }
    if (!any_pipeline) ir.flags |= IRF_NO_OUTPUT;

    // `lens { post: bloom(0.4, 1.5) | grain(0.015) }` (parser.rs:1197-1234): parsed, and ignored by both of the
    // reference's emitters. The chain is kept as one trailing layer marked `post` that renderers skip unless
    // asked to apply it to the composited colour; only Color -> Color builtins can run there, others are dropped.
    // The flags describe the text the reference emits, so this layer must not touch them.
    {
        const uint32_t saved_flags = ir.flags;
        IrLayer pl;
        pl.name = "lens.post";
        pl.is_pipeline = true;
        pl.post = true;
        pl.end_state = State::Color;
        pl.stage_begin = (uint32_t)ir.stages.size();
        for (const Lens& lens : c.lenses)
            for (const Stage& s : lens.post) {
                const Builtin* b = builtin_lookup(s.name);
                if (!b || b->input != State::Color || b->output != State::Color) continue;
                ir.stages.push_back(lower_stage(s, b));
            }
        pl.stage_count = (uint32_t)ir.stages.size() - pl.stage_begin;
        if (pl.stage_count) ir.layers.push_back(std::move(pl));
        ir.flags = saved_flags;
    }

    // arc timeline (runtime/arc.rs:62-93): blocks and entries concatenate, the cursor advances by each duration
    double cursor = 0.0;
    for (const ArcBlock& blk : c.arcs) {
        for (const ArcEntry& e : blk.entries) {
            IrArc a;
            a.target = e.target;
            a.dur = e.dur.kind == DurKind::Seconds ? e.dur.v : e.dur.kind == DurKind::Millis ? e.dur.v / 1000.0 : (double)e.dur.bars * 2.0;
            a.start = cursor;
            a.easing = e.has_easing ? e.easing : "linear";
            std::string dotted = e.target;
            for (char& ch : dotted) if (ch == '.') ch = '_';
            size_t dot = e.target.rfind('.');
            std::string bare = dot == std::string::npos ? e.target : e.target.substr(dot + 1);
            for (size_t i = 0; i < ir.uniforms.size() && a.idx < 0; ++i) if (ir.uniforms[i].name == dotted) a.idx = (int)i;
            for (size_t i = 0; i < ir.uniforms.size() && a.idx < 0; ++i) if (ir.uniforms[i].name == bare) a.idx = (int)i;
            if (!eval_js_const(e.from, a.from) || !eval_js_const(e.to, a.to)) a.idx = -1;
            ir.arcs.push_back(std::move(a));
            cursor += ir.arcs.back().dur;
        }
    }

    // modulation and temporal operators of layer parameters (Params-style layers as codegen/mod.rs:122-135 collects
    // them, and the inline parameters of pipeline layers, which are uniforms by codegen/mod.rs:75-87)
    auto dur_secs = [](const Duration& d) { return d.kind == DurKind::Seconds ? d.v : d.kind == DurKind::Millis ? d.v / 1000.0 : (double)d.bars * 2.0; };
    auto add_anim = [&](const Param& p) {
        if (!p.has_mod && p.temporal.empty()) return;
        IrParamAnim pa;
        pa.name = p.name;
        auto it = uslot.find(p.name);
        if (it != uslot.end()) pa.idx = (int)it->second;
        if (p.has_mod) {
            if (!lower_frame_expr(p.modulation, uslot, pa.mod) || pa.mod.size() >= 48) pa.mod.clear();
            pa.mod_text = expr_source_text(p.modulation);
        }
        for (const TemporalOp& t : p.temporal) {
            IrTemporal o;
            o.kind = (uint8_t)t.kind;
            if (t.kind == TemporalOp::Range) {   // temporal.rs:21-28: numbers pass, anything else is not a number here
                if (!eval_js_const(t.lo, o.lo)) o.lo = 0.0;
                if (!eval_js_const(t.hi, o.hi)) o.hi = 0.0;
            } else {
                o.secs = dur_secs(t.dur);
            }
            pa.ops.push_back(o);
        }
        ir.anims.push_back(std::move(pa));
    };
    for (const Layer& l : c.layers) {
        if (!l.is_pipeline) for (const Param& p : l.params) add_anim(p);
        for (const Param& p : l.opts) add_anim(p);
    }
    // resonance (resonance.rs:32-61): the uniform is "target_field", else "field"
    for (const ResonateBlock& blk : c.resonates) {
        for (const ResonateEntry& e : blk.entries) {
            IrResonance r;
            r.source = e.source;
            r.target = e.target + "." + e.field;
            for (size_t i = 0; i < ir.uniforms.size() && r.idx < 0; ++i) if (ir.uniforms[i].name == e.target + "_" + e.field) r.idx = (int)i;
            for (size_t i = 0; i < ir.uniforms.size() && r.idx < 0; ++i) if (ir.uniforms[i].name == e.field) r.idx = (int)i;
            if (!eval_js_const(e.weight, r.weight)) r.idx = -1;
            ir.resonances.push_back(std::move(r));
        }
    }

    if (c.has_gravity) {
        ir.gravity.present = true;
        ir.gravity.damping_text = fmt_f64_display(c.gravity.damping);   // `{damping}` (gravity.rs:92-95)
        ir.gravity.damping = (float)c.gravity.damping;
        ir.gravity.bounds = c.gravity.bounds;
        ir.gravity.law_text = force_expr_text(c.gravity.force_law);
        lower_law(c.gravity.force_law, ir);
    }
    return ir;
}

// ── arcs ─────────────────────────────────────────────────────────────────
bool eval_js_const(const Expr& e, double& out)
{
    auto truthy = [](double v) { return v != 0.0 && v == v; };
    switch (e.kind) {
    case Expr::Number: out = e.num; return true;
    case Expr::Paren: return eval_js_const(e.kids[0], out);
    case Expr::Neg: if (!eval_js_const(e.kids[0], out)) return false; out = -out; return true;
    case Expr::Dur: out = e.dur.kind == DurKind::Seconds ? e.dur.v : e.dur.kind == DurKind::Millis ? e.dur.v / 1000.0 : (double)e.dur.bars * 2.0; return true;
    case Expr::Ident:   // expr.rs:39-47 constants; `time`, colours (arrays) and free names are not numbers
        if (e.s == "pi") { out = 3.14159265358979; return true; }
        if (e.s == "tau") { out = 6.28318530717959; return true; }
        if (e.s == "e") { out = 2.71828182845905; return true; }
        if (e.s == "phi") { out = 1.61803398874989; return true; }
        return false;
    case Expr::Bin: {
        double l, r;
        if (!eval_js_const(e.kids[0], l) || !eval_js_const(e.kids[1], r)) return false;
        switch (e.op) {
        case BinOp::Add: out = l + r; break;
        case BinOp::Sub: out = l - r; break;
        case BinOp::Mul: out = l * r; break;
        case BinOp::Div: out = l / r; break;
        case BinOp::Pow: out = std::pow(l, r); break;
        case BinOp::Gt: out = l > r ? 1.0 : 0.0; break;
        case BinOp::Lt: out = l < r ? 1.0 : 0.0; break;
        }
        return true;
    }
    case Expr::Ternary: {
        double c;
        if (!eval_js_const(e.kids[0], c)) return false;
        return eval_js_const(e.kids[truthy(c) ? 1 : 2], out);
    }
    case Expr::Call: {
        std::vector<double> a(e.args.size());
        for (size_t i = 0; i < e.args.size(); ++i) if (!eval_js_const(e.args[i].value, a[i])) return false;
        const std::string& n = e.s;
        const size_t k = a.size();
        if (n == "mod" && k == 2) { out = std::fmod(a[0], a[1]); return true; }            // JS %
        if (n == "mod" && k == 1) { out = std::fmod(a[0], 1.0); return true; }
        if (n == "fract" && k == 1) { out = std::fmod(a[0], 1.0); return true; }
        if (n == "abs" && k == 1) { out = std::fabs(a[0]); return true; }
        if (n == "sin" && k == 1) { out = std::sin(a[0]); return true; }
        if (n == "cos" && k == 1) { out = std::cos(a[0]); return true; }
        if (n == "tan" && k == 1) { out = std::tan(a[0]); return true; }
        if (n == "sqrt" && k == 1) { out = std::sqrt(a[0]); return true; }
        if (n == "floor" && k == 1) { out = std::floor(a[0]); return true; }
        if (n == "ceil" && k == 1) { out = std::ceil(a[0]); return true; }
        if (n == "round" && k == 1) { out = std::floor(a[0] + 0.5); return true; }          // Math.round
        if (n == "exp" && k == 1) { out = std::exp(a[0]); return true; }
        if (n == "log" && k == 1) { out = std::log(a[0]); return true; }
        if (n == "sign" && k == 1) { out = a[0] > 0 ? 1.0 : a[0] < 0 ? -1.0 : a[0]; return true; }
        if (n == "min" && k >= 1) { out = a[0]; for (double v : a) out = (v != v || out != out) ? NAN : std::min(out, v); return true; }
        if (n == "max" && k >= 1) { out = a[0]; for (double v : a) out = (v != v || out != out) ? NAN : std::max(out, v); return true; }
        if (n == "pow" && k == 2) { out = std::pow(a[0], a[1]); return true; }
        if (n == "atan2" && k == 2) { out = std::atan2(a[0], a[1]); return true; }
        if (n == "mix" && k == 3) { out = a[0] + (a[1] - a[0]) * a[2]; return true; }
        if (n == "clamp" && k == 3) { out = std::min(std::max(a[0], a[1]), a[2]); return true; }
        if (n == "smoothstep" && k == 3) {
            double t = std::min(std::max((a[2] - a[0]) / (a[1] - a[0]), 0.0), 1.0);
            out = t * t * (3 - 2 * t);
            return true;
        }
        if (n == "length" && k >= 1) { double s2 = 0; for (double v : a) s2 += v * v; out = std::sqrt(s2); return true; }
        return false;
    }
    default: return false;   // strings, arrays, dotted names
    }
}

namespace {
// The seven easing curves of runtime/arc.rs:22-30; unknown names fall back to linear (`_ease[a.ease] || _ease.linear`).
double ease(const std::string& name, double t)
{
    const double PI = 3.141592653589793;
    if (name == "expo_in") return t == 0 ? 0 : std::pow(2, 10 * (t - 1));
    if (name == "expo_out") return t == 1 ? 1 : 1 - std::pow(2, -10 * t);
    if (name == "cubic_in_out") return t < 0.5 ? 4 * t * t * t : 1 - std::pow(-2 * t + 2, 3) / 2;
    if (name == "smooth") return t * t * (3 - 2 * t);
    if (name == "elastic") return t == 0 ? 0 : t == 1 ? 1 : -std::pow(2, 10 * (t - 1)) * std::sin((t - 1.1) * 5 * PI);
    if (name == "bounce") {
        const double n = 7.5625, d = 2.75;
        if (t < 1 / d) return n * t * t;
        if (t < 2 / d) { t -= 1.5 / d; return n * t * t + 0.75; }
        if (t < 2.5 / d) { t -= 2.25 / d; return n * t * t + 0.9375; }
        t -= 2.625 / d;
        return n * t * t + 0.984375;
    }
    return t;
}
}  // namespace

void arc_update(const Ir& ir, double time, double* params)
{
    for (const IrArc& a : ir.arcs) {
        if (a.idx < 0) continue;
        if (time < a.start) continue;
        if (time >= a.start + a.dur) { params[a.idx] = a.to; continue; }
        const double t = (time - a.start) / a.dur;
        params[a.idx] = a.from + (a.to - a.from) * ease(a.easing, t);
    }
}

// ── serialisation ────────────────────────────────────────────────────────
namespace {
void put_u32(std::vector<uint8_t>& b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
void put_f32(std::vector<uint8_t>& b, float f) { uint32_t v; memcpy(&v, &f, 4); put_u32(b, v); }
void put_str(std::vector<uint8_t>& b, const std::string& s, size_t n)
{
    for (size_t i = 0; i < n; ++i) b.push_back(i + 1 < n && i < s.size() ? (uint8_t)s[i] : 0);
}
std::string json_escape(const std::string& s)
{
    std::string o;
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back((char)c); }
        else if (c == '\n') o += "\\n";
        else if (c == '\t') o += "\\t";
        else if (c == '\r') o += "\\r";
        else if (c < 0x20) { char buf[8]; snprintf(buf, sizeof buf, "\\u%04x", c); o += buf; }
        else o.push_back((char)c);
    }
    return o;
}
std::string f32_json(float f)
{
    if (f != f || f - f != 0.0f) return "null";
    char buf[64];
    snprintf(buf, sizeof buf, "%.9g", (double)f);
    return buf;
}
}  // namespace

std::vector<uint8_t> ir_serialize(const Ir& ir)
{
    std::vector<uint8_t> b;
    put_u32(b, 0x31524947u);   // "GIR1"
    put_u32(b, 0);             // total bytes, patched below
    put_u32(b, (uint32_t)ir.layers.size());
    put_u32(b, (uint32_t)ir.stages.size());
    put_u32(b, (uint32_t)ir.uniforms.size());
    put_u32(b, (uint32_t)ir.gravity.law.size());
    put_u32(b, ir.flags);
    put_u32(b, ir.multi_layer ? 1 : 0);
    put_u32(b, ir.uses_memory ? 1 : 0);
    put_u32(b, ir.gravity.present ? 1 : 0);
    put_f32(b, ir.gravity.damping);
    put_u32(b, (uint32_t)ir.gravity.bounds);
    put_str(b, ir.name, 64);
    for (const IrLayer& l : ir.layers) {
        put_u32(b, l.is_pipeline ? 1 : 0);
        put_u32(b, l.stage_begin);
        put_u32(b, l.stage_count);
        put_u32(b, l.has_memory ? 1 : 0);
        put_f32(b, l.decay);
        put_u32(b, (uint32_t)l.end_state);
        put_u32(b, l.post ? 1 : 0);
        put_u32(b, 0);
    }
    for (const IrStage& s : ir.stages) {
        put_u32(b, s.builtin);
        put_u32(b, s.nargs);
        put_u32(b, s.aux);
        put_u32(b, 0);
        for (int i = 0; i < 4; ++i) {
            put_u32(b, s.args[i].kind);
            put_u32(b, s.args[i].slot);
            put_f32(b, s.args[i].lit);
            // low half: index + 1 into the derived values, high half: index + 1 into the pixel expressions (0 = none)
            put_u32(b, (uint32_t)(s.args[i].derived + 1) | ((uint32_t)(s.args[i].pixel + 1) << 16));
        }
    }
    for (const UniformInfo& u : ir.uniforms) {
        put_f32(b, (float)u.def);
        put_str(b, u.name, 60);
    }
    for (const IrLawOp& o : ir.gravity.law) {
        put_u32(b, o.op);
        put_f32(b, o.val);
        put_u32(b, o.fn);
        put_u32(b, 0);
    }
    // trailing section: the pixel-expression programs
    put_u32(b, (uint32_t)ir.pixel_exprs.size());
    for (const IrDerived& d : ir.pixel_exprs) {
        put_u32(b, (uint32_t)d.prog.size());
        for (const IrLawOp& o : d.prog) {
            put_u32(b, o.op);
            put_f32(b, o.val);
            put_u32(b, o.fn);
            put_u32(b, 0);
        }
    }
    // trailing section: the projection block that names this cinematic (mode + 1, 0 = none)
    put_u32(b, ir.project.present ? (uint32_t)ir.project.mode + 1u : 0u);
    put_f32(b, ir.project.fov_deg);
    put_u32(b, ir.project.count);
    put_u32(b, 0);
    // trailing section: the frame-level programs of expression-valued stage arguments (what game_ir_eval_derived runs)
    put_u32(b, (uint32_t)ir.derived.size());
    for (const IrDerived& d : ir.derived) {
        put_u32(b, (uint32_t)d.prog.size());
        for (const IrLawOp& o : d.prog) {
            put_u32(b, o.op);
            put_f32(b, o.val);
            put_u32(b, o.fn);
            put_u32(b, 0);
        }
    }
    uint32_t total = (uint32_t)b.size();
    for (int i = 0; i < 4; ++i) b[4 + i] = (uint8_t)(total >> (8 * i));
    return b;
}

// ── blob -> IR ───────────────────────────────────────────────────────────────────────────
namespace {
struct BlobReader {
    const uint8_t* p; const uint8_t* end; bool ok = true;
    uint32_t u32() { if (end - p < 4) { ok = false; return 0; } uint32_t v = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; p += 4; return v; }
    float f32() { uint32_t v = u32(); float f; memcpy(&f, &v, 4); return f; }
    // fixed-width field holding a NUL-terminated string
    std::string str(size_t n)
    {
        if ((size_t)(end - p) < n) { ok = false; return ""; }
        const void* z = memchr(p, 0, n);
        if (!z) { ok = false; p += n; return ""; }
        std::string s((const char*)p, (const char*)z);
        p += n;
        return s;
    }
};
bool finite32(float f) { return f - f == 0.0f; }

// A reverse Polish program is usable when every operation is known, finds its operands on the stack, the stack stays
// within the evaluators' depth and the result is the one value left on it at the end. frame_floats: slots of a frame row it may
// read (0 = none, a gravity force law); pixel: it may read p / uv.
bool prog_well_formed(const std::vector<IrLawOp>& prog, uint32_t frame_floats, bool pixel, bool gravity)
{
    int sp = 0;
    for (const IrLawOp& o : prog) {
        int need = 0;
        switch (o.op) {
        case LAW_NUM: break;
        case LAW_DIST: case LAW_TIME: if (!gravity) return false; break;   // the pair distance and the simulation time
        case LAW_UNIFORM: if (o.fn >= frame_floats) return false; break;
        case LAW_PX: case LAW_PY: case LAW_UVX: case LAW_UVY: if (!pixel) return false; break;
        case LAW_ADD: case LAW_SUB: case LAW_MUL: case LAW_DIV: case LAW_POW: case LAW_GT: case LAW_LT: need = 2; break;
        case LAW_NEG: need = 1; break;
        // a call names a function of that many arguments
        case LAW_CALL1: need = 1; if (!(o.fn <= LF_SIGN || (o.fn >= LF_EXP2 && o.fn <= LF_ROUND))) return false; break;
        case LAW_CALL2: need = 2; if (!((o.fn >= LF_MIN && o.fn <= LF_STEP) || (o.fn >= LF_MOD && o.fn <= LF_LENGTH2))) return false; break;
        case LAW_CALL3: need = 3; if (!((o.fn >= LF_CLAMP && o.fn <= LF_SMOOTHSTEP) || o.fn == LF_SELECT)) return false; break;
        default: return false;
        }
        if (sp < need) return false;
        sp = sp - need + 1;
        if (sp > 60) return false;
    }
    return prog.empty() || sp == 1;
}

bool read_prog(BlobReader& r, uint32_t n, std::vector<IrLawOp>& prog)
{
    prog.resize(n);
    for (IrLawOp& o : prog) { o.op = r.u32(); o.val = r.f32(); o.fn = r.u32(); r.u32(); }
    return r.ok;
}
}  // namespace

bool ir_deserialize(const uint8_t* data, size_t len, Ir& ir, std::string& why)
{
    auto fail = [&](const char* m) { why = m; return false; };
    ir = Ir{};
    if (!data || len < 112) return fail("shorter than the header");
    BlobReader r{data, data + len};
    if (r.u32() != 0x31524947u) return fail("no 'GIR1' magic");
    if (r.u32() != len) return fail("total_bytes does not match the length");
    const uint32_t n_layers = r.u32(), n_stages = r.u32(), n_uniforms = r.u32(), n_law = r.u32();
    if (n_layers > 4096 || n_stages > 65536 || n_uniforms > 4096 || n_law > 4096) return fail("a section count is beyond the format's limits");
    ir.flags = r.u32();
    ir.multi_layer = r.u32() != 0;
    ir.uses_memory = r.u32() != 0;
    ir.gravity.present = r.u32() != 0;
    ir.gravity.damping = r.f32();
    const uint32_t bounds = r.u32();
    if (bounds > 2) return fail("unknown gravity bounds mode");
    ir.gravity.bounds = (Bounds)bounds;
    if (!finite32(ir.gravity.damping)) return fail("gravity damping is not finite");
    ir.gravity.damping_text = fmt_f64_display((double)ir.gravity.damping);
    ir.name = r.str(64);
    if (!r.ok) return fail("cinematic name is not NUL-terminated");
    bool any_memory = false;
    ir.layers.resize(n_layers);
    for (IrLayer& l : ir.layers) {
        l.is_pipeline = r.u32() != 0;
        l.stage_begin = r.u32();
        l.stage_count = r.u32();
        l.has_memory = r.u32() != 0;
        l.decay = r.f32();
        const uint32_t end_state = r.u32();
        l.post = r.u32() != 0;
        r.u32();
        if (!r.ok) return fail("truncated in the layer table");
        if ((uint64_t)l.stage_begin + l.stage_count > n_stages) return fail("a layer's stages lie outside the stage table");
        if (end_state > 2) return fail("unknown layer end state");
        if (!finite32(l.decay)) return fail("a memory decay is not finite");
        l.end_state = (State)end_state;
        l.decay_text = fmt_f64_display((double)l.decay);
        any_memory = any_memory || l.has_memory;     // as the lowering sets it: any layer with a memory clause
    }
    if (any_memory != ir.uses_memory) return fail("uses_memory disagrees with the layers");
    struct Packed { uint32_t v[4]; };
    std::vector<Packed> packed(n_stages);
    ir.stages.resize(n_stages);
    for (uint32_t si = 0; si < n_stages; ++si) {
        IrStage& s = ir.stages[si];
        const uint32_t id = r.u32();
        s.nargs = r.u32();
        s.aux = r.u32();
        r.u32();
        for (int i = 0; i < 4; ++i) {
            s.args[i].kind = r.u32();
            s.args[i].slot = r.u32();
            s.args[i].lit = r.f32();
            packed[si].v[i] = r.u32();
        }
        if (!r.ok) return fail("truncated in the stage table");
        const Builtin* b = id < builtin_count() ? builtin_by_id((uint16_t)id) : nullptr;
        if (!b) return fail("unknown builtin id");
        if (s.nargs > 4) return fail("a stage has more than four arguments");
        s.builtin = (uint16_t)id;
        s.name = b->name;
        for (int i = 0; i < 4; ++i) {
            const IrArg& a = s.args[i];
            if (a.kind > ARG_BUILTIN) return fail("unknown stage argument kind");
            if (a.kind == ARG_USER && a.slot >= n_uniforms) return fail("a stage argument names a user uniform that does not exist");
            if (a.kind == ARG_BUILTIN && a.slot >= BU_COUNT) return fail("a stage argument names a built-in uniform that does not exist");
            if (!finite32(a.lit)) return fail("a literal stage argument is not finite");
        }
    }
    ir.uniforms.resize(n_uniforms);
    for (UniformInfo& u : ir.uniforms) {
        u.def = (double)r.f32();
        u.name = r.str(60);
        if (!r.ok) return fail("truncated or unterminated name in the uniform table");
    }
    if (!read_prog(r, n_law, ir.gravity.law)) return fail("truncated in the force law");
    if (!prog_well_formed(ir.gravity.law, 0, false, true)) return fail("the force law is not a well-formed program");
    const uint32_t frame_floats = BU_COUNT + n_uniforms;
    const uint32_t n_pix = r.u32();
    if (!r.ok || n_pix > 256) return fail("too many pixel expressions");
    ir.pixel_exprs.resize(n_pix);
    for (IrDerived& d : ir.pixel_exprs) {
        const uint32_t n = r.u32();
        if (!r.ok || n > 128 || !read_prog(r, n, d.prog)) return fail("truncated or oversized pixel expression");
        if (d.prog.empty() || !prog_well_formed(d.prog, frame_floats, true, false)) return fail("a pixel expression is not a well-formed program");
    }
    const uint32_t pmode = r.u32();
    ir.project.fov_deg = r.f32();
    ir.project.count = r.u32();
    r.u32();
    if (!r.ok) return fail("truncated in the projection block");
    if (pmode > 4) return fail("unknown projection mode");
    ir.project.present = pmode != 0;
    if (pmode) ir.project.mode = (ProjectMode)(pmode - 1);
    if (!finite32(ir.project.fov_deg) || ir.project.count < 1 || ir.project.count > 1048576) return fail("projection parameters out of range");
    const uint32_t n_derived = r.u32();
    if (!r.ok || n_derived > 65535) return fail("too many derived values");
    ir.derived.resize(n_derived);
    for (IrDerived& d : ir.derived) {
        const uint32_t n = r.u32();
        if (!r.ok || n > 4096 || !read_prog(r, n, d.prog)) return fail("truncated or oversized derived-value program");
        if (d.prog.empty() || !prog_well_formed(d.prog, frame_floats, false, false)) return fail("a derived value is not a well-formed program");
    }
    if (r.p != r.end) return fail("bytes left over after the last section");
    for (uint32_t si = 0; si < n_stages; ++si)
        for (int i = 0; i < 4; ++i) {
            const uint32_t lo = packed[si].v[i] & 0xffffu, hi = packed[si].v[i] >> 16;
            if (lo > n_derived || hi > n_pix) return fail("a stage argument names an expression that does not exist");
            ir.stages[si].args[i].derived = (int)lo - 1;
            ir.stages[si].args[i].pixel = (int)hi - 1;
        }
    return true;
}

std::string ir_to_json(const Ir& ir)
{
    std::string j = "{";
    j += "\"name\":\"" + json_escape(ir.name) + "\"";
    j += ",\"multi_layer\":" + std::string(ir.multi_layer ? "true" : "false");
    j += ",\"uses_memory\":" + std::string(ir.uses_memory ? "true" : "false");
    j += ",\"flags\":" + std::to_string(ir.flags);
    j += ",\"uniforms\":[";
    for (size_t i = 0; i < ir.uniforms.size(); ++i) {
        if (i) j += ",";
        char buf[64];
        snprintf(buf, sizeof buf, "%.17g", ir.uniforms[i].def);
        j += "{\"name\":\"" + json_escape(ir.uniforms[i].name) + "\",\"default\":" + buf + "}";
    }
    j += "],\"layers\":[";
    for (size_t li = 0; li < ir.layers.size(); ++li) {
        const IrLayer& l = ir.layers[li];
        if (li) j += ",";
        j += "{\"name\":\"" + json_escape(l.name) + "\",\"pipeline\":" + (l.is_pipeline ? "true" : "false") +
             ",\"post\":" + (l.post ? "true" : "false");
        j += ",\"memory\":" + (l.has_memory ? "\"" + l.decay_text + "\"" : std::string("null"));
        j += ",\"end_state\":\"" + std::string(state_name(l.end_state)) + "\",\"stages\":[";
        for (uint32_t si = 0; si < l.stage_count; ++si) {
            const IrStage& s = ir.stages[l.stage_begin + si];
            if (si) j += ",";
            j += "{\"name\":\"" + s.name + "\",\"id\":" + std::to_string(s.builtin) + ",\"aux\":" + std::to_string(s.aux) + ",\"args\":[";
            for (uint32_t a = 0; a < s.nargs; ++a) {
                if (a) j += ",";
                const IrArg& ia = s.args[a];
                const char* kind = ia.kind == ARG_LIT ? "lit" : ia.kind == ARG_USER ? "user" : "builtin";
                j += "{\"kind\":\"" + std::string(kind) + "\",\"slot\":" + std::to_string(ia.slot) + ",\"lit\":" + f32_json(ia.lit) +
                     ",\"text\":\"" + json_escape(ia.text) + "\",\"derived\":" + std::to_string(ia.derived) + ",\"pixel\":" + std::to_string(ia.pixel) + "}";
            }
            j += "]}";
        }
        j += "]}";
    }
    j += "],\"gravity\":";
    if (ir.gravity.present) {
        j += "{\"damping\":\"" + ir.gravity.damping_text + "\",\"bounds\":" + std::to_string((int)ir.gravity.bounds);
        j += ",\"law\":\"" + json_escape(ir.gravity.law_text) + "\",\"ops\":" + std::to_string(ir.gravity.law.size()) + "}";
    } else {
        j += "null";
    }
    j += ",\"derived\":[";
    for (size_t i = 0; i < ir.derived.size(); ++i) j += (i ? ",\"" : "\"") + json_escape(ir.derived[i].text) + "\"";
    j += "]";
    j += ",\"pixel_exprs\":[";
    for (size_t i = 0; i < ir.pixel_exprs.size(); ++i) j += (i ? ",\"" : "\"") + json_escape(ir.pixel_exprs[i].text) + "\"";
    j += "]";
    j += ",\"anims\":[";
    for (size_t i = 0; i < ir.anims.size(); ++i) {
        const IrParamAnim& pa = ir.anims[i];
        static const char* const kinds[] = {"delay", "smooth", "trigger", "range"};
        j += std::string(i ? "," : "") + "{\"name\":\"" + json_escape(pa.name) + "\",\"idx\":" + std::to_string(pa.idx) +
             ",\"modulation\":" + (pa.mod_text.empty() ? std::string("null") : "\"" + json_escape(pa.mod_text) + "\"") +
             ",\"evaluated\":" + (pa.mod.empty() ? "false" : "true") + ",\"ops\":[";
        for (size_t k = 0; k < pa.ops.size(); ++k)
            j += std::string(k ? "," : "") + "{\"kind\":\"" + kinds[pa.ops[k].kind & 3] + "\",\"secs\":" + fmt_f64_display(pa.ops[k].secs) +
                 ",\"lo\":" + fmt_f64_display(pa.ops[k].lo) + ",\"hi\":" + fmt_f64_display(pa.ops[k].hi) + "}";
        j += "]}";
    }
    j += "],\"resonances\":[";
    for (size_t i = 0; i < ir.resonances.size(); ++i)
        j += std::string(i ? "," : "") + "{\"source\":\"" + json_escape(ir.resonances[i].source) + "\",\"target\":\"" +
             json_escape(ir.resonances[i].target) + "\",\"idx\":" + std::to_string(ir.resonances[i].idx) + ",\"weight\":" +
             fmt_f64_display(ir.resonances[i].weight) + "}";
    j += "]";
    j += ",\"arcs\":[";
    for (size_t i = 0; i < ir.arcs.size(); ++i) {
        const IrArc& a = ir.arcs[i];
        char buf[256];
        snprintf(buf, sizeof buf, "{\"idx\":%d,\"from\":%.17g,\"to\":%.17g,\"start\":%.17g,\"dur\":%.17g,", a.idx,
                 a.idx < 0 ? 0.0 : a.from, a.idx < 0 ? 0.0 : a.to, a.start, a.dur);
        j += (i ? "," : "") + std::string(buf) + "\"target\":\"" + json_escape(a.target) + "\",\"ease\":\"" + json_escape(a.easing) + "\"}";
    }
    j += "]";
    if (ir.project.present) {
        static const char* modes[] = {"flat", "dome", "cube", "led"};
        char buf[128];
        snprintf(buf, sizeof buf, ",\"project\":{\"mode\":\"%s\",\"fov_deg\":%.9g,\"count\":%u}", modes[(int)ir.project.mode],
                 (double)ir.project.fov_deg, ir.project.count);
        j += buf;
    }
    j += "}";
    return j;
}

// ── whole-source entry point (lib.rs:117-155) ────────────────────────────
namespace {
CompileError compile_impl(const std::string& src, const std::string* base_dir, const std::vector<std::string>* lib_dirs, CompiledProgram& out)
{
    out = CompiledProgram{};
    std::vector<Token> toks;
    CompileError e = lex(src, toks);
    if (!e.ok()) return e;
    e = parse(std::move(toks), out.ast);
    if (!e.ok()) return e;
    if (base_dir) {
        e = resolve_imports(out.ast, *base_dir, *lib_dirs);
        if (!e.ok()) return e;
    }
    out.warnings = check_program(out.ast);
    for (Cinematic& c : out.ast.cinematics) (void)expand_defines(c);   // result ignored, like `let _ =` (lib.rs:122)
    for (Cinematic& c : out.ast.cinematics) optimize_cinematic(c);
    for (const Cinematic& c : out.ast.cinematics) {
        e = validate_cinematic(c);
        if (!e.ok()) return e;
        out.irs.push_back(lower_cinematic(c));
        // the first project block whose source names this cinematic (lib.rs:187-192 overrides that output's vertex stage)
        for (const ProjectBlock& pb : out.ast.projects) {
            if (pb.source != c.name) continue;
            IrProject& pr = out.irs.back().project;
            pr.present = true;
            pr.mode = pb.mode;
            for (const Param& prm : pb.params) {
                double v = 0.0;
                if (!eval_js_const(prm.value, v)) continue;
                if (prm.name == "fov" || prm.name == "fov_deg" || prm.name == "fisheye") pr.fov_deg = (float)v;
                else if ((prm.name == "count" || prm.name == "leds") && v >= 1.0 && v <= 1048576.0) pr.count = (uint32_t)v;
            }
            break;
        }
        out.blobs.push_back(ir_serialize(out.irs.back()));
        out.jsons.push_back(ir_to_json(out.irs.back()));
    }
    return CompileError{};
}
}  // namespace

CompileError compile_source(const std::string& src, CompiledProgram& out) { return compile_impl(src, nullptr, nullptr, out); }

CompileError compile_source_with_imports(const std::string& src, const std::string& base_dir,
                                         const std::vector<std::string>& lib_dirs, CompiledProgram& out)
{
    return compile_impl(src, &base_dir, &lib_dirs, out);
}

}  // namespace game
