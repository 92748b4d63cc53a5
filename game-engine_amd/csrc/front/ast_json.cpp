// JSON dump of the AST and of the token stream, for tooling and for tests that restate the
// reference's parser/optimizer unit tests (game-compiler/src/parser_tests.rs, optimize.rs:543+).
#include "front.hpp"
#include "capi_internal.hpp"
#include <cstdio>

namespace game {
namespace {

std::string esc(const std::string& s)
{
    std::string o;
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back((char)c); }
        else if (c == '\n') o += "\\n";
        else if (c == '\t') o += "\\t";
        else if (c == '\r') o += "\\r";
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o.push_back((char)c);
    }
    return o;
}
std::string q(const std::string& s) { return "\"" + esc(s) + "\""; }
std::string num(double v)
{
    if (v != v || v - v != 0.0) return "null";
    char b[64];
    snprintf(b, sizeof b, "%.17g", v);
    return b;
}
std::string dur(const Duration& d)
{
    switch (d.kind) {
    case DurKind::Seconds: return "{\"unit\":\"s\",\"v\":" + num(d.v) + "}";
    case DurKind::Millis: return "{\"unit\":\"ms\",\"v\":" + num(d.v) + "}";
    default: return "{\"unit\":\"bars\",\"v\":" + std::to_string((long long)d.bars) + "}";
    }
}
std::string expr(const Expr& e);
std::string args(const std::vector<Arg>& as)
{
    std::string s = "[";
    for (size_t i = 0; i < as.size(); ++i) {
        if (i) s += ",";
        s += "{\"name\":" + (as[i].named ? q(as[i].name) : std::string("null")) + ",\"value\":" + expr(as[i].value) + "}";
    }
    return s + "]";
}
std::string kids(const Expr& e)
{
    std::string s = "[";
    for (size_t i = 0; i < e.kids.size(); ++i) { if (i) s += ","; s += expr(e.kids[i]); }
    return s + "]";
}
std::string expr(const Expr& e)
{
    static const char* ops[] = {"Add", "Sub", "Mul", "Div", "Pow", "Gt", "Lt"};
    switch (e.kind) {
    case Expr::Number: return "{\"t\":\"Number\",\"v\":" + num(e.num) + "}";
    case Expr::String: return "{\"t\":\"String\",\"v\":" + q(e.s) + "}";
    case Expr::Ident: return "{\"t\":\"Ident\",\"v\":" + q(e.s) + "}";
    case Expr::Dotted: return "{\"t\":\"DottedIdent\",\"object\":" + q(e.s) + ",\"field\":" + q(e.field) + "}";
    case Expr::Array: return "{\"t\":\"Array\",\"elems\":" + kids(e) + "}";
    case Expr::Paren: return "{\"t\":\"Paren\",\"inner\":" + expr(e.kids[0]) + "}";
    case Expr::Neg: return "{\"t\":\"Neg\",\"inner\":" + expr(e.kids[0]) + "}";
    case Expr::Bin: return std::string("{\"t\":\"BinOp\",\"op\":\"") + ops[(int)e.op] + "\",\"left\":" + expr(e.kids[0]) + ",\"right\":" + expr(e.kids[1]) + "}";
    case Expr::Call: return "{\"t\":\"Call\",\"name\":" + q(e.s) + ",\"args\":" + args(e.args) + "}";
    case Expr::Dur: return "{\"t\":\"Duration\",\"v\":" + dur(e.dur) + "}";
    case Expr::Ternary: return "{\"t\":\"Ternary\",\"condition\":" + expr(e.kids[0]) + ",\"if_true\":" + expr(e.kids[1]) + ",\"if_false\":" + expr(e.kids[2]) + "}";
    }
    return "null";
}
std::string stages(const std::vector<Stage>& st)
{
    std::string s = "[";
    for (size_t i = 0; i < st.size(); ++i) {
        if (i) s += ",";
        s += "{\"name\":" + q(st[i].name) + ",\"args\":" + args(st[i].args) + "}";
    }
    return s + "]";
}
std::string params(const std::vector<Param>& ps)
{
    std::string s = "[";
    for (size_t i = 0; i < ps.size(); ++i) {
        const Param& p = ps[i];
        if (i) s += ",";
        s += "{\"name\":" + q(p.name) + ",\"value\":" + expr(p.value) + ",\"modulation\":" + (p.has_mod ? expr(p.modulation) : std::string("null")) + ",\"temporal_ops\":[";
        for (size_t k = 0; k < p.temporal.size(); ++k) {
            const TemporalOp& t = p.temporal[k];
            if (k) s += ",";
            static const char* names[] = {"Delay", "Smooth", "Trigger", "Range"};
            s += std::string("{\"t\":\"") + names[t.kind] + "\"";
            if (t.kind == TemporalOp::Range) s += ",\"min\":" + expr(t.lo) + ",\"max\":" + expr(t.hi);
            else s += ",\"duration\":" + dur(t.dur);
            s += "}";
        }
        s += "]}";
    }
    return s + "]";
}
std::string strs(const std::vector<std::string>& v)
{
    std::string s = "[";
    for (size_t i = 0; i < v.size(); ++i) { if (i) s += ","; s += q(v[i]); }
    return s + "]";
}
std::string arc_entries(const std::vector<ArcEntry>& es)
{
    std::string s = "[";
    for (size_t i = 0; i < es.size(); ++i) {
        const ArcEntry& e = es[i];
        if (i) s += ",";
        s += "{\"target\":" + q(e.target) + ",\"from\":" + expr(e.from) + ",\"to\":" + expr(e.to) + ",\"duration\":" + dur(e.dur) +
             ",\"easing\":" + (e.has_easing ? q(e.easing) : std::string("null")) + "}";
    }
    return s + "]";
}
std::string nodes(const std::vector<NamedNode>& ns, const char* kind_key)
{
    std::string s = "[";
    for (size_t i = 0; i < ns.size(); ++i) {
        if (i) s += ",";
        s += "{\"name\":" + q(ns[i].name) + ",\"" + kind_key + "\":" + q(ns[i].kind) + ",\"params\":" + params(ns[i].params) + "}";
    }
    return s + "]";
}
std::string cinematic(const Cinematic& c)
{
    std::string s = "{\"name\":" + q(c.name) + ",\"layers\":[";
    for (size_t i = 0; i < c.layers.size(); ++i) {
        const Layer& l = c.layers[i];
        if (i) s += ",";
        s += "{\"name\":" + q(l.name) + ",\"opts\":" + params(l.opts) + ",\"memory\":" + (l.has_memory ? num(l.memory) : std::string("null")) +
             ",\"cast\":" + (l.has_cast ? q(l.cast) : std::string("null")) + ",\"body\":";
        if (l.is_pipeline) s += "{\"Pipeline\":" + stages(l.stages) + "}";
        else s += "{\"Params\":" + params(l.params) + "}";
        s += "}";
    }
    s += "],\"arcs\":[";
    for (size_t i = 0; i < c.arcs.size(); ++i) { if (i) s += ","; s += "{\"entries\":" + arc_entries(c.arcs[i].entries) + "}"; }
    s += "],\"resonates\":[";
    for (size_t i = 0; i < c.resonates.size(); ++i) {
        if (i) s += ",";
        s += "{\"entries\":[";
        for (size_t k = 0; k < c.resonates[i].entries.size(); ++k) {
            const ResonateEntry& e = c.resonates[i].entries[k];
            if (k) s += ",";
            s += "{\"source\":" + q(e.source) + ",\"target\":" + q(e.target) + ",\"field\":" + q(e.field) + ",\"weight\":" + expr(e.weight) + "}";
        }
        s += "]}";
    }
    s += "],\"listen\":" + (c.has_listen ? "{\"signals\":" + nodes(c.listen, "algorithm") + "}" : std::string("null"));
    s += ",\"voice\":" + (c.has_voice ? "{\"nodes\":" + nodes(c.voice, "kind") + "}" : std::string("null"));
    if (c.has_score) {
        s += ",\"score\":{\"tempo_bpm\":" + num(c.score.tempo_bpm) + ",\"motifs\":[";
        for (size_t i = 0; i < c.score.motifs.size(); ++i) {
            if (i) s += ",";
            s += "{\"name\":" + q(c.score.motifs[i].name) + ",\"entries\":" + arc_entries(c.score.motifs[i].entries) + "}";
        }
        s += "],\"phrases\":[";
        for (size_t i = 0; i < c.score.phrases.size(); ++i) { if (i) s += ","; s += "{\"name\":" + q(c.score.phrases[i].name) + ",\"motifs\":" + strs(c.score.phrases[i].refs) + "}"; }
        s += "],\"sections\":[";
        for (size_t i = 0; i < c.score.sections.size(); ++i) { if (i) s += ","; s += "{\"name\":" + q(c.score.sections[i].name) + ",\"phrases\":" + strs(c.score.sections[i].refs) + "}"; }
        s += "],\"arrange\":" + strs(c.score.arrange) + "}";
    } else s += ",\"score\":null";
    if (c.has_gravity) {
        static const char* b[] = {"Reflect", "Wrap", "None"};
        s += ",\"gravity\":{\"force_law\":" + expr(c.gravity.force_law) + ",\"damping\":" + num(c.gravity.damping) + ",\"bounds\":\"" + b[(int)c.gravity.bounds] + "\"}";
    } else s += ",\"gravity\":null";
    s += ",\"lenses\":[";
    for (size_t i = 0; i < c.lenses.size(); ++i) {
        const Lens& l = c.lenses[i];
        if (i) s += ",";
        s += "{\"name\":" + (l.has_name ? q(l.name) : std::string("null")) + ",\"properties\":" + params(l.properties) + ",\"post\":" + stages(l.post) + "}";
    }
    s += "],\"react\":";
    if (c.has_react) {
        s += "{\"reactions\":[";
        for (size_t i = 0; i < c.react.size(); ++i) { if (i) s += ","; s += "{\"signal\":" + expr(c.react[i].signal) + ",\"action\":" + expr(c.react[i].action) + "}"; }
        s += "]}";
    } else s += "null";
    s += ",\"defines\":[";
    for (size_t i = 0; i < c.defines.size(); ++i) {
        if (i) s += ",";
        s += "{\"name\":" + q(c.defines[i].name) + ",\"params\":" + strs(c.defines[i].params) + ",\"body\":" + stages(c.defines[i].body) + "}";
    }
    return s + "]}";
}

std::string program_json(const Program& p)
{
    std::string s = "{\"imports\":[";
    for (size_t i = 0; i < p.imports.size(); ++i) {
        if (i) s += ",";
        s += "{\"path\":" + q(p.imports[i].path) + ",\"alias\":" + q(p.imports[i].alias) + ",\"exposed\":" + strs(p.imports[i].exposed) + "}";
    }
    s += "],\"cinematics\":[";
    for (size_t i = 0; i < p.cinematics.size(); ++i) { if (i) s += ","; s += cinematic(p.cinematics[i]); }
    s += "],\"breeds\":[";
    for (size_t i = 0; i < p.breeds.size(); ++i) {
        const BreedBlock& b = p.breeds[i];
        if (i) s += ",";
        s += "{\"name\":" + q(b.name) + ",\"parents\":" + strs(b.parents) + ",\"inherit_rules\":[";
        for (size_t k = 0; k < b.inherit.size(); ++k) {
            if (k) s += ",";
            s += "{\"target\":" + q(b.inherit[k].target) + ",\"strategy\":" + q(b.inherit[k].strategy) + ",\"weight\":" + num(b.inherit[k].weight) + "}";
        }
        s += "],\"mutations\":[";
        for (size_t k = 0; k < b.mutations.size(); ++k) {
            if (k) s += ",";
            s += "{\"target\":" + q(b.mutations[k].target) + ",\"range\":" + num(b.mutations[k].range) + "}";
        }
        s += "]}";
    }
    s += "],\"projects\":[";
    static const char* modes[] = {"Flat", "Dome", "Cube", "Led"};
    for (size_t i = 0; i < p.projects.size(); ++i) {
        if (i) s += ",";
        s += std::string("{\"mode\":\"") + modes[(int)p.projects[i].mode] + "\",\"source\":" + q(p.projects[i].source) + ",\"params\":" + params(p.projects[i].params) + "}";
    }
    return s + "]}";
}

thread_local std::string g_json;

}  // namespace
}  // namespace game

using namespace game;

extern "C" {

// stage 0: as parsed (compile_to_ast, lib.rs:66). stage 1: after expand_defines + optimize_cinematic
// (lib.rs:121-129). Returns NULL on a lex/parse error (message in game_last_error()).
const char* game_front_ast_json(const char* src, size_t len, int stage)
{
    if (!src) { set_error(GAME_E_ARGUMENT, "game_front_ast_json: null source"); return nullptr; }
    std::string source(src, len);
    std::vector<Token> toks;
    CompileError e = lex(source, toks);
    Program prog;
    if (e.ok()) e = parse(std::move(toks), prog);
    if (!e.ok()) { set_error(error_code_of(e), e.display()); return nullptr; }
    std::string stats = "[";
    if (stage >= 1) {
        for (Cinematic& c : prog.cinematics) (void)expand_defines(c);
        for (size_t i = 0; i < prog.cinematics.size(); ++i) {
            OptimizeStats st = optimize_cinematic(prog.cinematics[i]);
            if (i) stats += ",";
            stats += "{\"constants_folded\":" + std::to_string(st.constants_folded) + ",\"noop_stages_removed\":" +
                     std::to_string(st.noop_stages_removed) + ",\"dead_defines_removed\":" + std::to_string(st.dead_defines_removed) +
                     ",\"dead_uniforms\":" + strs(st.dead_uniforms) + ",\"semantic_warnings\":" + strs(st.semantic_warnings) + "}";
        }
    }
    stats += "]";
    g_json = program_json(prog);
    g_json.insert(g_json.size() - 1, ",\"optimize\":" + stats);
    return g_json.c_str();
}

// Tokens as "<start>\x1f<end>\x1f<display text>" records separated by \x1e (token.rs Display).
// NULL on a lex error.
const char* game_front_tokens(const char* src, size_t len)
{
    if (!src) { set_error(GAME_E_ARGUMENT, "game_front_tokens: null source"); return nullptr; }
    std::vector<Token> toks;
    CompileError e = lex(std::string(src, len), toks);
    if (!e.ok()) { set_error(error_code_of(e), e.display()); return nullptr; }
    g_json.clear();
    for (const Token& t : toks) g_json += std::to_string(t.start) + "\x1f" + std::to_string(t.end) + "\x1f" + token_display(t) + "\x1e";
    return g_json.c_str();
}

}  // extern "C"
