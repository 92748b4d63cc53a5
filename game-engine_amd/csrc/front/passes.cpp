// AST passes between parsing and lowering: builtin registry, define expansion, constant folding,
// no-op stage elimination, semantic checks, pipeline validation, uniform extraction, argument
// resolution. Each function cites the reference routine whose observable behaviour it keeps.
#include "front.hpp"
#include <cmath>
#include <cfloat>
#include <cstring>
#include <map>
#include <set>
#include <algorithm>

namespace game {

// ── builtin registry (builtins.rs:43-293; same order as BUILTINS) ────────
#define P(n, d) BuiltinParam{n, true, d}
#define PN(n) BuiltinParam{n, false, 0.0}
#define NOP BuiltinParam{nullptr, false, 0.0}
static const Builtin REGISTRY[] = {
    {"circle", B_CIRCLE, 1, {P("radius", 0.2), NOP, NOP, NOP}, State::Position, State::Sdf},
    {"ring", B_RING, 2, {P("radius", 0.3), P("width", 0.02), NOP, NOP}, State::Position, State::Sdf},
    {"star", B_STAR, 3, {P("points", 5.0), P("radius", 0.3), P("inner", 0.15), NOP}, State::Position, State::Sdf},
    {"fbm", B_FBM, 4, {P("scale", 1.0), P("octaves", 4.0), P("persistence", 0.5), P("lacunarity", 2.0)}, State::Position, State::Sdf},
    {"simplex", B_SIMPLEX, 1, {P("scale", 1.0), NOP, NOP, NOP}, State::Position, State::Sdf},
    {"glow", B_GLOW, 1, {P("intensity", 1.5), NOP, NOP, NOP}, State::Sdf, State::Color},
    {"shade", B_SHADE, 3, {P("r", 1.0), P("g", 1.0), P("b", 1.0), NOP}, State::Sdf, State::Color},
    {"emissive", B_EMISSIVE, 1, {P("intensity", 1.0), NOP, NOP, NOP}, State::Sdf, State::Color},
    {"tint", B_TINT, 3, {P("r", 1.0), P("g", 1.0), P("b", 1.0), NOP}, State::Color, State::Color},
    {"bloom", B_BLOOM, 2, {P("threshold", 0.3), P("strength", 2.0), NOP, NOP}, State::Color, State::Color},
    {"grain", B_GRAIN, 1, {P("amount", 0.1), NOP, NOP, NOP}, State::Color, State::Color},
    {"translate", B_TRANSLATE, 2, {P("x", 0.0), P("y", 0.0), NOP, NOP}, State::Position, State::Position},
    {"rotate", B_ROTATE, 1, {P("angle", 0.0), NOP, NOP, NOP}, State::Position, State::Position},
    {"scale", B_SCALE, 1, {P("s", 1.0), NOP, NOP, NOP}, State::Position, State::Position},
    {"mask_arc", B_MASK_ARC, 1, {PN("angle"), NOP, NOP, NOP}, State::Sdf, State::Sdf},
    {"threshold", B_THRESHOLD, 1, {P("cutoff", 0.5), NOP, NOP, NOP}, State::Sdf, State::Sdf},
    {"twist", B_TWIST, 1, {P("amount", 0.0), NOP, NOP, NOP}, State::Position, State::Position},
    {"mirror", B_MIRROR, 1, {P("axis", 0.0), NOP, NOP, NOP}, State::Position, State::Position},
    {"repeat", B_REPEAT, 1, {P("count", 4.0), NOP, NOP, NOP}, State::Position, State::Position},
    {"gradient", B_GRADIENT, 3, {PN("color_a"), PN("color_b"), PN("mode"), NOP}, State::Position, State::Color},
    {"blend", B_BLEND, 1, {P("factor", 0.5), NOP, NOP, NOP}, State::Color, State::Color},
    {"vignette", B_VIGNETTE, 2, {P("strength", 0.5), P("radius", 0.8), NOP, NOP}, State::Color, State::Color},
    {"voronoi", B_VORONOI, 1, {P("scale", 5.0), NOP, NOP, NOP}, State::Position, State::Sdf},
    {"onion", B_ONION, 1, {P("thickness", 0.02), NOP, NOP, NOP}, State::Sdf, State::Sdf},
    {"domain_warp", B_DOMAIN_WARP, 2, {P("amount", 0.1), P("freq", 3.0), NOP, NOP}, State::Position, State::Position},
    {"curl_noise", B_CURL_NOISE, 2, {P("frequency", 1.0), P("amplitude", 0.1), NOP, NOP}, State::Position, State::Position},
    {"box", B_BOX, 2, {P("w", 0.2), P("h", 0.2), NOP, NOP}, State::Position, State::Sdf},
    {"polygon", B_POLYGON, 2, {P("sides", 6.0), P("radius", 0.3), NOP, NOP}, State::Position, State::Sdf},
    {"concentric_waves", B_CONCENTRIC_WAVES, 3, {P("amplitude", 1.0), P("width", 0.5), P("frequency", 3.0), NOP}, State::Position, State::Sdf},
    {"round", B_ROUND, 1, {P("radius", 0.02), NOP, NOP, NOP}, State::Sdf, State::Sdf},
    {"tonemap", B_TONEMAP, 1, {P("exposure", 1.0), NOP, NOP, NOP}, State::Color, State::Color},
    {"scanlines", B_SCANLINES, 2, {P("frequency", 200.0), P("intensity", 0.3), NOP, NOP}, State::Color, State::Color},
    {"chromatic", B_CHROMATIC, 1, {P("offset", 0.005), NOP, NOP, NOP}, State::Color, State::Color},
    {"saturate_color", B_SATURATE_COLOR, 1, {P("amount", 1.0), NOP, NOP, NOP}, State::Color, State::Color},
    {"glitch", B_GLITCH, 1, {P("intensity", 0.5), NOP, NOP, NOP}, State::Color, State::Color},
    {"displace", B_DISPLACE, 1, {P("strength", 0.1), NOP, NOP, NOP}, State::Position, State::Position},
    {"spectrum", B_SPECTRUM, 3, {P("bass", 0.0), P("mid", 0.0), P("treble", 0.0), NOP}, State::Position, State::Color},
    {"palette", B_PALETTE, 1, {PN("name"), NOP, NOP, NOP}, State::Sdf, State::Color},
};
#undef P
#undef PN
#undef NOP
static_assert(sizeof(REGISTRY) / sizeof(REGISTRY[0]) == B_COUNT, "registry must cover every builtin id");

const char* state_name(State s)
{
    switch (s) { case State::Position: return "Position"; case State::Sdf: return "Sdf"; default: return "Color"; }
}
size_t builtin_count() { return B_COUNT; }
const Builtin* builtin_by_id(uint16_t id) { return id < B_COUNT ? &REGISTRY[id] : nullptr; }
const Builtin* builtin_lookup(const std::string& name)
{
    for (const Builtin& b : REGISTRY) if (name == b.name) return &b;
    return nullptr;
}
const char* builtin_suggest(const std::string& name)
{
    std::vector<const char*> names;
    for (const Builtin& b : REGISTRY) names.push_back(b.name);
    return suggest_similar(name, names);
}

// ── define expansion (analysis.rs:24-153) ────────────────────────────────
namespace {

Expr substitute_expr(const Expr& e, const std::map<std::string, const Expr*>& m)
{
    switch (e.kind) {
    case Expr::Ident: {
        auto it = m.find(e.s);
        return it != m.end() ? *it->second : e;
    }
    case Expr::Paren: case Expr::Neg: case Expr::Bin: case Expr::Array: case Expr::Ternary: {
        Expr out = e;
        for (size_t i = 0; i < e.kids.size(); ++i) out.kids[i] = substitute_expr(e.kids[i], m);
        return out;
    }
    case Expr::Call: {
        Expr out = e;
        for (size_t i = 0; i < e.args.size(); ++i) out.args[i].value = substitute_expr(e.args[i].value, m);
        return out;
    }
    default: return e;
    }
}

std::vector<Stage> substitute_define(const DefineBlock& def, const std::vector<Arg>& actual)
{
    std::map<std::string, const Expr*> m;
    for (size_t i = 0; i < def.params.size(); ++i) {
        const Arg* found = nullptr;
        for (const Arg& a : actual) if (a.named && a.name == def.params[i]) { found = &a; break; }
        if (!found && i < actual.size()) found = &actual[i];
        if (found) m[def.params[i]] = &found->value;   // later duplicates of a formal overwrite, like HashMap::insert
    }
    std::vector<Stage> out;
    for (const Stage& s : def.body) {
        Stage ns;
        ns.name = s.name;
        for (const Arg& a : s.args) {
            Arg na;
            na.named = a.named; na.name = a.name;
            na.value = substitute_expr(a.value, m);
            ns.args.push_back(std::move(na));
        }
        out.push_back(std::move(ns));
    }
    return out;
}

}  // namespace

CompileError expand_defines(Cinematic& c)
{
    const size_t MAX_DEPTH = 16;
    // name -> block; a later define of the same name replaces an earlier one (HashMap collect)
    std::map<std::string, const DefineBlock*> defs;
    for (const DefineBlock& d : c.defines) defs[d.name] = &d;
    if (defs.empty()) return CompileError{};
    for (Layer& layer : c.layers) {
        if (!layer.is_pipeline) continue;
        for (size_t depth = 0; depth < MAX_DEPTH; ++depth) {
            bool expanded = false;
            std::vector<Stage> ns;
            for (const Stage& s : layer.stages) {
                auto it = defs.find(s.name);
                if (it != defs.end()) {
                    for (Stage& x : substitute_define(*it->second, s.args)) ns.push_back(std::move(x));
                    expanded = true;
                } else {
                    ns.push_back(s);
                }
            }
            layer.stages = std::move(ns);
            if (!expanded) break;
            if (depth == MAX_DEPTH - 1) {
                CompileError e;
                e.kind = ErrKind::Codegen;
                e.message = "define expansion exceeded max depth (16) in layer '" + layer.name + "' — possible recursive define";
                return e;
            }
        }
    }
    return CompileError{};
}

// ── constant folding (optimize.rs:84-236) ────────────────────────────────
namespace {

inline bool is_zero(double v) { return std::fabs(v) < DBL_EPSILON; }
inline bool is_one(double v) { return std::fabs(v - 1.0) < DBL_EPSILON; }

bool as_number(const Expr& e, double& v)
{
    if (e.kind == Expr::Number) { v = e.num; return true; }
    if (e.kind == Expr::Paren) return as_number(e.kids[0], v);
    return false;
}

// f64::min / f64::max: if one side is NaN the other is returned
inline double rust_min(double a, double b) { return std::fmin(a, b); }
inline double rust_max(double a, double b) { return std::fmax(a, b); }

bool fold_known_call(const std::string& name, const std::vector<Arg>& args, Expr& out)
{
    std::vector<double> pos;
    for (const Arg& a : args) {
        if (a.named) continue;
        double v;
        if (as_number(a.value, v)) pos.push_back(v);
    }
    if (pos.size() == 1) {
        double v = pos[0];
        if (name == "sin") { out = Expr::number(is_zero(v) ? 0.0 : std::sin(v)); return true; }
        if (name == "cos") { out = Expr::number(is_zero(v) ? 1.0 : std::cos(v)); return true; }
        if (name == "abs") { out = Expr::number(std::fabs(v)); return true; }
        if (name == "floor") { out = Expr::number(std::floor(v)); return true; }
        if (name == "ceil") { out = Expr::number(std::ceil(v)); return true; }
        if (name == "sqrt" && v >= 0.0) { out = Expr::number(std::sqrt(v)); return true; }
        return false;
    }
    if (pos.size() == 2) {
        if (name == "min") { out = Expr::number(rust_min(pos[0], pos[1])); return true; }
        if (name == "max") { out = Expr::number(rust_max(pos[0], pos[1])); return true; }
        if (name == "pow") { out = Expr::number(std::pow(pos[0], pos[1])); return true; }
    }
    return false;
}

}  // namespace

size_t constant_fold(Expr& e)
{
    size_t count = 0;
    switch (e.kind) {
    case Expr::Bin: case Expr::Neg: case Expr::Paren: case Expr::Array: case Expr::Ternary:
        for (Expr& k : e.kids) count += constant_fold(k);
        break;
    case Expr::Call:
        for (Arg& a : e.args) count += constant_fold(a.value);
        break;
    default: break;
    }
    bool replaced = false;
    Expr r;
    if (e.kind == Expr::Neg) {
        double v;
        if (as_number(e.kids[0], v)) { r = Expr::number(-v); replaced = true; }
    } else if (e.kind == Expr::Bin) {
        double a = 0, b = 0;
        bool la = as_number(e.kids[0], a), rb = as_number(e.kids[1], b);
        if (la && rb) {
            switch (e.op) {
            case BinOp::Add: r = Expr::number(a + b); replaced = true; break;
            case BinOp::Sub: r = Expr::number(a - b); replaced = true; break;
            case BinOp::Mul: r = Expr::number(a * b); replaced = true; break;
            case BinOp::Div: if (!is_zero(b)) { r = Expr::number(a / b); replaced = true; } break;
            case BinOp::Pow: r = Expr::number(std::pow(a, b)); replaced = true; break;
            default: break;
            }
        } else if (!la && rb) {
            switch (e.op) {
            case BinOp::Add: case BinOp::Sub:
                if (is_zero(b)) { r = std::move(e.kids[0]); replaced = true; }
                break;
            case BinOp::Mul:
                if (is_one(b)) { r = std::move(e.kids[0]); replaced = true; }
                else if (is_zero(b)) { r = Expr::number(0.0); replaced = true; }
                break;
            case BinOp::Div:
                if (is_one(b)) { r = std::move(e.kids[0]); replaced = true; }
                else if (!is_zero(b)) {   // x / C -> x * (1/C)
                    Expr m;
                    m.kind = Expr::Bin; m.op = BinOp::Mul;
                    m.kids.push_back(std::move(e.kids[0]));
                    m.kids.push_back(Expr::number(1.0 / b));
                    r = std::move(m);
                    replaced = true;
                }
                break;
            default: break;
            }
        } else if (la && !rb) {
            switch (e.op) {
            case BinOp::Add: if (is_zero(a)) { r = std::move(e.kids[1]); replaced = true; } break;
            case BinOp::Mul:
                if (is_one(a)) { r = std::move(e.kids[1]); replaced = true; }
                else if (is_zero(a)) { r = Expr::number(0.0); replaced = true; }
                break;
            default: break;
            }
        }
    } else if (e.kind == Expr::Call) {
        replaced = fold_known_call(e.s, e.args, r);
    }
    if (replaced) {
        Expr tmp = std::move(r);
        e = std::move(tmp);
        ++count;
    }
    return count;
}

// ── no-op stage elimination (optimize.rs:242-272) ────────────────────────
static bool is_noop_stage(const Stage& s)
{
    std::vector<std::pair<bool, double>> pos;   // (is number, value) per positional arg
    for (const Arg& a : s.args) {
        if (a.named) continue;
        double v = 0;
        bool ok = as_number(a.value, v);
        pos.push_back({ok, v});
    }
    if (s.name == "translate") {
        if (pos.empty()) return false;
        for (auto& p : pos) if (!(p.first && is_zero(p.second))) return false;
        return true;
    }
    if (s.name == "scale") return pos.size() == 1 && pos[0].first && is_one(pos[0].second);
    if (s.name == "rotate" || s.name == "twist") return pos.size() == 1 && pos[0].first && is_zero(pos[0].second);
    return false;
}

size_t eliminate_noop_stages(std::vector<Stage>& stages)
{
    size_t before = stages.size();
    stages.erase(std::remove_if(stages.begin(), stages.end(), is_noop_stage), stages.end());
    return before - stages.size();
}

// ── dead uniforms / dead defines / semantic checks (optimize.rs:278-541) ──
namespace {

void collect_idents(const Expr& e, std::set<std::string>& out)
{
    switch (e.kind) {
    case Expr::Ident: out.insert(e.s); break;
    case Expr::Dotted: out.insert(e.s); out.insert(e.s + "." + e.field); break;
    case Expr::Bin: case Expr::Neg: case Expr::Paren: case Expr::Array: case Expr::Ternary:
        for (const Expr& k : e.kids) collect_idents(k, out);
        break;
    case Expr::Call:
        for (const Arg& a : e.args) collect_idents(a.value, out);
        break;
    default: break;
    }
}

void collect_param(const Param& p, std::set<std::string>& out)
{
    collect_idents(p.value, out);
    if (p.has_mod) collect_idents(p.modulation, out);
}

std::vector<std::string> find_dead_uniforms(const Cinematic& c)
{
    std::vector<std::string> declared;
    for (const Layer& l : c.layers)
        if (!l.is_pipeline) for (const Param& p : l.params) declared.push_back(p.name);
    if (declared.empty()) return {};
    std::set<std::string> ref;
    for (const Layer& l : c.layers) {
        for (const Param& p : l.opts) collect_param(p, ref);
        if (l.is_pipeline) { for (const Stage& s : l.stages) for (const Arg& a : s.args) collect_idents(a.value, ref); }
        else for (const Param& p : l.params) collect_param(p, ref);
    }
    for (const Lens& lens : c.lenses) {
        for (const Param& p : lens.properties) collect_param(p, ref);
        for (const Stage& s : lens.post) for (const Arg& a : s.args) collect_idents(a.value, ref);
    }
    for (const ArcBlock& arc : c.arcs) for (const ArcEntry& en : arc.entries) { collect_idents(en.from, ref); collect_idents(en.to, ref); }
    for (const ResonateBlock& rb : c.resonates) for (const ResonateEntry& en : rb.entries) {
        collect_idents(en.weight, ref);
        ref.insert(en.source);
        ref.insert(en.target);
    }
    for (const DefineBlock& d : c.defines) for (const Stage& s : d.body) for (const Arg& a : s.args) collect_idents(a.value, ref);
    if (c.has_react) for (const Reaction& r : c.react) { collect_idents(r.signal, ref); collect_idents(r.action, ref); }
    std::vector<std::string> dead;
    for (const std::string& n : declared) if (!ref.count(n)) dead.push_back(n);
    return dead;
}

size_t eliminate_dead_defines(Cinematic& c)
{
    std::set<std::string> used;
    for (const Layer& l : c.layers) if (l.is_pipeline) for (const Stage& s : l.stages) used.insert(s.name);
    for (const Lens& lens : c.lenses) for (const Stage& s : lens.post) used.insert(s.name);
    size_t before = c.defines.size();
    c.defines.erase(std::remove_if(c.defines.begin(), c.defines.end(),
                                   [&](const DefineBlock& d) { return !used.count(d.name); }),
                    c.defines.end());
    return before - c.defines.size();
}

bool arg_references_name(const Expr& e, const std::string& name)
{
    switch (e.kind) {
    case Expr::Ident: return e.s == name;
    case Expr::Dotted: return e.s == name;
    case Expr::Bin: case Expr::Neg: case Expr::Paren: case Expr::Array: case Expr::Ternary:
        for (const Expr& k : e.kids) if (arg_references_name(k, name)) return true;
        return false;
    case Expr::Call:
        for (const Arg& a : e.args) if (arg_references_name(a.value, name)) return true;
        return false;
    default: return false;
    }
}

}  // namespace

std::vector<std::string> check_define_semantics(const Cinematic& c)
{
    std::vector<std::string> w;
    std::set<std::string> names;
    for (const DefineBlock& d : c.defines) names.insert(d.name);
    for (const DefineBlock& d : c.defines) {
        for (const Stage& s : d.body)
            if (!builtin_lookup(s.name) && !names.count(s.name))
                w.push_back("define '" + d.name + "': unknown function '" + s.name + "'");
        for (const std::string& p : d.params) {
            bool used = false;
            for (const Stage& s : d.body) for (const Arg& a : s.args) if (arg_references_name(a.value, p)) used = true;
            if (!used) w.push_back("define '" + d.name + "': parameter '" + p + "' is never used");
        }
    }
    return w;
}

static void arity_of(const std::vector<Stage>& stages, const std::string& where, std::vector<std::string>& w)
{
    for (const Stage& s : stages) {
        const Builtin* b = builtin_lookup(s.name);
        if (!b) continue;
        size_t n = 0;
        for (const Arg& a : s.args) if (!a.named) ++n;
        if (n > b->nparams)
            w.push_back(where + ": '" + s.name + "' accepts " + std::to_string(b->nparams) + " parameters, but " +
                        std::to_string(n) + " were given");
    }
}

std::vector<std::string> check_arity(const Cinematic& c)
{
    std::vector<std::string> w;
    for (const Layer& l : c.layers) if (l.is_pipeline) arity_of(l.stages, "layer '" + l.name + "'", w);
    for (const Lens& lens : c.lenses) arity_of(lens.post, "lens '" + (lens.has_name ? lens.name : std::string("<unnamed>")) + "'", w);
    return w;
}

std::vector<std::string> check_program(const Program& p)
{
    std::vector<std::string> w;
    for (const Cinematic& c : p.cinematics) {
        for (auto& s : check_define_semantics(c)) w.push_back(s);
        for (auto& s : check_arity(c)) w.push_back(s);
    }
    return w;
}

OptimizeStats optimize_cinematic(Cinematic& c)
{
    OptimizeStats st;
    for (Layer& l : c.layers) {
        for (Param& p : l.opts) {
            st.constants_folded += constant_fold(p.value);
            if (p.has_mod) st.constants_folded += constant_fold(p.modulation);
        }
        if (l.is_pipeline) {
            for (Stage& s : l.stages) for (Arg& a : s.args) st.constants_folded += constant_fold(a.value);
        } else {
            for (Param& p : l.params) {
                st.constants_folded += constant_fold(p.value);
                if (p.has_mod) st.constants_folded += constant_fold(p.modulation);
            }
        }
    }
    for (Layer& l : c.layers) if (l.is_pipeline) st.noop_stages_removed += eliminate_noop_stages(l.stages);
    st.dead_uniforms = find_dead_uniforms(c);
    st.dead_defines_removed = eliminate_dead_defines(c);
    for (auto& s : check_define_semantics(c)) st.semantic_warnings.push_back(s);
    for (auto& s : check_arity(c)) st.semantic_warnings.push_back(s);
    return st;
}

// ── validation (stages.rs:45-76, cast.rs:12-65, codegen/mod.rs:110-119) ──
static CompileError verr(std::string msg, int code = 0)
{
    CompileError e;
    e.kind = ErrKind::Validation; e.code = code; e.message = std::move(msg);
    return e;
}

CompileError validate_pipeline(const std::vector<Stage>& stages, State& end_state)
{
    State state = State::Position;
    const char* prev = nullptr;
    for (const Stage& s : stages) {
        const Builtin* b = builtin_lookup(s.name);
        if (!b) {
            CompileError e = verr("unknown stage function: '" + s.name + "'", 1);
            if (const char* sug = builtin_suggest(s.name)) e.help = std::string("did you mean '") + sug + "'?";
            return e;
        }
        if (b->input != state && b->input != State::Position) {
            std::string p = prev ? prev : "(start)";
            return verr("type mismatch: '" + s.name + "' expects " + state_name(b->input) + " input, but pipeline is in " +
                        state_name(state) + " state\n  help: the pipeline flows Position -> Sdf -> Color. '" + p +
                        "' produces " + state_name(state) + " output.", 2);
        }
        prev = s.name.c_str();
        state = b->output;
    }
    end_state = state;
    return CompileError{};
}

CompileError validate_cinematic(const Cinematic& c)
{
    for (const Layer& l : c.layers) {
        if (!l.is_pipeline) continue;
        State st;
        CompileError e = validate_pipeline(l.stages, st);
        if (!e.ok()) return e;
    }
    for (const Layer& l : c.layers) {
        if (!l.has_cast) continue;
        State expected;
        if (l.cast == "sdf" || l.cast == "distance") expected = State::Sdf;
        else if (l.cast == "color" || l.cast == "rgba") expected = State::Color;
        else if (l.cast == "position" || l.cast == "uv") expected = State::Position;
        else return verr("unknown cast type '" + l.cast + "'. Expected: sdf, color, position");
        if (!l.is_pipeline) continue;
        if (l.stages.empty()) return verr("layer '" + l.name + "' is cast as '" + l.cast + "' but has an empty pipeline");
        State actual;
        CompileError e = validate_pipeline(l.stages, actual);
        if (!e.ok()) return e;
        if (actual != expected)
            return verr("layer '" + l.name + "' is cast as '" + l.cast + "' (" + state_name(expected) +
                        ") but pipeline produces " + state_name(actual));
    }
    return CompileError{};
}

// ── uniforms (codegen/mod.rs:54-107) ─────────────────────────────────────
std::vector<UniformInfo> extract_uniforms(const Cinematic& c)
{
    std::vector<UniformInfo> out;
    std::set<std::string> seen;
    auto add_param = [&](const Param& p) {
        if (seen.insert(p.name).second) out.push_back({p.name, p.value.kind == Expr::Number ? p.value.num : 0.0});
    };
    for (const Layer& l : c.layers) {
        if (!l.is_pipeline) for (const Param& p : l.params) add_param(p);
        for (const Param& p : l.opts) add_param(p);
        if (l.is_pipeline)
            for (const Stage& s : l.stages)
                for (const Arg& a : s.args)
                    if (a.value.kind == Expr::Ident && !builtin_lookup(a.value.s) && seen.insert(a.value.s).second)
                        out.push_back({a.value.s, 0.0});
    }
    return out;
}

// ── argument resolution (stages.rs:8-42) ─────────────────────────────────
static ArgText default_text(const char* stage_name, size_t pos)
{
    ArgText t;
    const Builtin* b = builtin_lookup(stage_name);
    if (b && pos < b->nparams && b->params[pos].has_default) t.text = fmt_fixed6(b->params[pos].def);
    else t.text = "0.0";
    return t;
}

static ArgText resolve_arg(const Arg& a, size_t pos, const char* stage_name)
{
    ArgText t;
    switch (a.value.kind) {
    case Expr::Number: t.text = fmt_fixed6(a.value.num); return t;
    case Expr::Ident: t.kind = ArgText::Ident; t.text = a.value.s; return t;
    case Expr::Dotted: t.kind = ArgText::Dotted; t.text = a.value.s + "_" + a.value.field; return t;
    default: t = default_text(stage_name, pos); t.dropped = &a.value; return t;
    }
}

ArgText get_arg(const std::vector<Arg>& args, const char* name, size_t pos, const char* stage_name)
{
    for (const Arg& a : args) if (a.named && a.name == name) return resolve_arg(a, pos, stage_name);
    if (pos < args.size()) return resolve_arg(args[pos], pos, stage_name);   // positional index counts named args too
    return default_text(stage_name, pos);
}

// ── force law text (gravity.rs:10-41) ────────────────────────────────────
std::string force_expr_text(const Expr& e)
{
    switch (e.kind) {
    case Expr::Number: return fmt_fixed6(e.num);
    case Expr::Ident:
        if (e.s == "dist") return "dist";
        if (e.s == "time") return "params.time";
        return e.s;
    case Expr::Bin: {
        std::string l = force_expr_text(e.kids[0]), r = force_expr_text(e.kids[1]);
        switch (e.op) {
        case BinOp::Add: return "(" + l + " + " + r + ")";
        case BinOp::Sub: return "(" + l + " - " + r + ")";
        case BinOp::Mul: return "(" + l + " * " + r + ")";
        case BinOp::Div: return "(" + l + " / " + r + ")";
        case BinOp::Pow: return "pow(" + l + ", " + r + ")";
        case BinOp::Gt: return "select(0.0, 1.0, " + l + " > " + r + ")";
        case BinOp::Lt: return "select(0.0, 1.0, " + l + " < " + r + ")";
        }
        return "1.0";
    }
    case Expr::Neg: return "(-" + force_expr_text(e.kids[0]) + ")";
    case Expr::Call: {
        std::string s = e.s + "(";
        for (size_t i = 0; i < e.args.size(); ++i) {
            if (i) s += ", ";
            s += force_expr_text(e.args[i].value);
        }
        return s + ")";
    }
    case Expr::Paren: return "(" + force_expr_text(e.kids[0]) + ")";
    default: return "1.0";
    }
}

}  // namespace game
