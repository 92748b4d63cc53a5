// Recursive-descent parser for .game source -> Program.
// Grammar, precedence and error-recovery behaviour follow game-compiler/src/parser.rs:
//   program  := (import | cinematic | breed | project)*                      (:229-291)
//   layer    := `layer` [ident] [(opts)] [memory: num] [cast ident] { body } (:449-489)
//   body     := `fn:` pipeline params* | params | pipeline                   (:509-570)
//   expr     := ternary < comparison < additive < term < factor(^, right) < atom  (:1290-1443)
// Only the first recorded error is reported, but recovery is reproduced because it decides which
// error is first (a failed recovery can surface a different error, e.g. at the closing brace).
#include "front.hpp"

namespace game {
namespace {

struct ParseFail { CompileError err; };

class Parser {
public:
    explicit Parser(std::vector<Token> toks) : toks_(std::move(toks)) {}

    CompileError parse_program(Program& out)
    {
        std::vector<CompileError> errors;
        while (!at_end()) {
            Tok k = peek()->kind;
            try {
                if (k == Tok::Import) out.imports.push_back(parse_import());
                else if (k == Tok::Cinematic) out.cinematics.push_back(parse_cinematic());
                else if (k == Tok::Breed) out.breeds.push_back(parse_breed());
                else if (k == Tok::Project) out.projects.push_back(parse_project());
                else {
                    auto [a, b] = current_pos();
                    Token t = advance();
                    throw ParseFail{perr(a, b, "expected `import`, `cinematic`, `breed`, or `project` at top level, found `" +
                                               token_display(t) + "`")};
                }
            } catch (ParseFail& f) {
                errors.push_back(std::move(f.err));
                skip_to_recovery();
            }
        }
        if (errors.empty()) return CompileError{};
        return errors.front();
    }

private:
    std::vector<Token> toks_;
    size_t pos_ = 0;

    // -- navigation --------------------------------------------------------
    const Token* peek() const { return pos_ < toks_.size() ? &toks_[pos_] : nullptr; }
    const Token* peek_at(size_t off) const { return pos_ + off < toks_.size() ? &toks_[pos_ + off] : nullptr; }
    bool at_end() const { return pos_ >= toks_.size(); }
    bool check(Tok k) const { return peek() && peek()->kind == k; }
    Token advance() { return toks_[pos_++]; }
    bool try_advance(Token& t) { if (at_end()) return false; t = toks_[pos_++]; return true; }
    std::pair<size_t, size_t> current_pos() const
    {
        if (pos_ < toks_.size()) return {toks_[pos_].start, toks_[pos_].end};
        if (!toks_.empty()) return {toks_.back().start, toks_.back().end};
        return {0, 0};
    }
    static CompileError perr(size_t a, size_t b, std::string msg)
    {
        CompileError e;
        e.kind = ErrKind::Parse; e.code = 3; e.a = a; e.b = b; e.message = std::move(msg);
        return e;
    }
    static std::string tok_text(Tok k) { Token t; t.kind = k; return token_display(t); }

    // -- expect helpers ----------------------------------------------------
    Token expect(Tok k)
    {
        auto [a, b] = current_pos();
        Token t;
        if (!try_advance(t)) throw ParseFail{perr(a, b, "expected `" + tok_text(k) + "`, found end of input")};
        if (t.kind != k) throw ParseFail{perr(a, b, "expected `" + tok_text(k) + "`, found `" + token_display(t) + "`")};
        return t;
    }
    std::string expect_ident(bool allow_all = false)
    {
        auto [a, b] = current_pos();
        Token t;
        if (!try_advance(t)) throw ParseFail{perr(a, b, "expected identifier, found end of input")};
        if (t.kind == Tok::Ident) return t.text;
        if (allow_all && t.kind == Tok::All) return "ALL";
        throw ParseFail{perr(a, b, "expected identifier, found `" + token_display(t) + "`")};
    }
    std::string expect_string()
    {
        auto [a, b] = current_pos();
        Token t;
        if (!try_advance(t)) throw ParseFail{perr(a, b, "expected string literal, found end of input")};
        if (t.kind == Tok::StringLit) return t.text;
        throw ParseFail{perr(a, b, "expected string literal, found `" + token_display(t) + "`")};
    }
    double expect_number()
    {
        auto [a, b] = current_pos();
        Token t;
        if (!try_advance(t)) throw ParseFail{perr(a, b, "expected number, found end of input")};
        if (t.kind == Tok::Float) return t.num;
        if (t.kind == Tok::Integer) return (double)t.inum;
        throw ParseFail{perr(a, b, "expected number, found `" + token_display(t) + "`")};
    }

    // -- recovery (parser.rs:173-223) ---------------------------------------
    void skip_to_recovery()
    {
        int depth = 0;
        while (const Token* t = peek()) {
            if (t->kind == Tok::LBrace) { ++depth; ++pos_; }
            else if (t->kind == Tok::RBrace && depth > 0) { --depth; ++pos_; }
            else if (t->kind == Tok::RBrace) { ++pos_; return; }
            else ++pos_;
        }
    }
    void skip_brace_block()   // keyword `{ ... }`
    {
        ++pos_;               // the keyword (callers only use this when a token is present)
        expect(Tok::LBrace);
        unsigned depth = 1;
        while (depth > 0) {
            Token t;
            if (!try_advance(t)) {
                auto [a, b] = current_pos();
                throw ParseFail{perr(a, b, "unexpected end of input while skipping block")};
            }
            if (t.kind == Tok::LBrace) ++depth;
            else if (t.kind == Tok::RBrace) --depth;
        }
    }
    void skip_brace_block_safe()
    {
        int depth = 1;
        while (depth > 0) {
            const Token* t = peek();
            if (!t) break;
            if (t->kind == Tok::LBrace) ++depth;
            else if (t->kind == Tok::RBrace) --depth;
            ++pos_;
        }
    }

    // -- import --------------------------------------------------------------
    Import parse_import()
    {
        expect(Tok::Import);
        Import imp;
        imp.path = expect_string();
        if (check(Tok::Expose)) {
            ++pos_;
            imp.exposed.push_back(expect_ident(true));
            while (check(Tok::Comma)) { ++pos_; imp.exposed.push_back(expect_ident(true)); }
        } else {
            expect(Tok::As);
            imp.alias = expect_ident();
        }
        return imp;
    }

    // -- cinematic -----------------------------------------------------------
    Cinematic parse_cinematic()
    {
        expect(Tok::Cinematic);
        Cinematic c;
        c.name = expect_string();
        expect(Tok::LBrace);
        std::vector<CompileError> errors;
        while (!at_end() && !check(Tok::RBrace)) {
            Tok k = peek()->kind;
            try {
                switch (k) {
                case Tok::Layer: c.layers.push_back(parse_layer()); break;
                case Tok::Arc: c.arcs.push_back(parse_arc()); break;
                case Tok::Resonate: c.resonates.push_back(parse_resonate()); break;
                case Tok::Listen: c.listen = parse_named_nodes(Tok::Listen); c.has_listen = true; break;
                case Tok::Voice: c.voice = parse_named_nodes(Tok::Voice); c.has_voice = true; break;
                case Tok::Score: c.score = parse_score(); c.has_score = true; break;
                case Tok::Gravity: c.gravity = parse_gravity(); c.has_gravity = true; break;
                case Tok::Lens: c.lenses.push_back(parse_lens()); break;
                case Tok::React: c.react = parse_react(); c.has_react = true; break;
                case Tok::Define: c.defines.push_back(parse_define()); break;
                case Tok::Signals: case Tok::Route: case Tok::Hear: case Tok::Feel:
                    try { skip_brace_block(); } catch (ParseFail& f) { errors.push_back(std::move(f.err)); }
                    break;
                default: {
                    auto [a, b] = current_pos();
                    errors.push_back(perr(a, b, "unexpected `" + token_display(*peek()) + "` inside cinematic block"));
                    ++pos_;
                    break;
                }
                }
            } catch (ParseFail& f) {
                errors.push_back(std::move(f.err));
                skip_brace_block_safe();
            }
        }
        expect(Tok::RBrace);
        if (!errors.empty()) throw ParseFail{errors.front()};
        return c;
    }

    // -- layer ---------------------------------------------------------------
    Layer parse_layer()
    {
        expect(Tok::Layer);
        Layer l;
        if (check(Tok::Ident)) l.name = expect_ident();
        else l.name = "_layer_" + std::to_string(pos_);
        if (check(Tok::LParen)) l.opts = parse_paren_params();
        if (check(Tok::Memory)) {
            ++pos_;
            expect(Tok::Colon);
            l.memory = expect_number();
            l.has_memory = true;
        }
        if (check(Tok::Cast)) {
            ++pos_;
            l.cast = expect_ident();
            l.has_cast = true;
        }
        expect(Tok::LBrace);
        parse_layer_body(l);
        expect(Tok::RBrace);
        return l;
    }

    std::vector<Param> parse_paren_params()   // `(key: val, ...)` (parser.rs:491, :894)
    {
        expect(Tok::LParen);
        std::vector<Param> ps;
        while (!at_end() && !check(Tok::RParen)) {
            Param p;
            p.name = expect_ident();
            expect(Tok::Colon);
            p.value = parse_expr();
            ps.push_back(std::move(p));
            if (!check(Tok::RParen)) expect(Tok::Comma);
        }
        expect(Tok::RParen);
        return ps;
    }

    void parse_layer_body(Layer& l)
    {
        if (at_end() || check(Tok::RBrace)) { l.is_pipeline = false; return; }
        const Token* t0 = peek();
        const Token* t1 = peek_at(1);
        if (t0->kind == Tok::Ident && t0->text == "fn" && t1 && t1->kind == Tok::Colon) {
            pos_ += 2;
            l.is_pipeline = true;
            l.stages = parse_stage_pipeline();
            std::vector<Param> inl = parse_param_entries();
            for (auto& p : inl) l.opts.push_back(std::move(p));
            return;
        }
        if (t0->kind == Tok::Ident && t1 && t1->kind == Tok::LParen) {
            l.is_pipeline = true;
            l.stages = parse_stage_pipeline();
            return;
        }
        l.is_pipeline = false;
        l.params = parse_param_entries();
    }

    std::vector<Param> parse_param_entries()   // name: expr [~ expr] temporal*  until `}`
    {
        std::vector<Param> ps;
        while (!at_end() && !check(Tok::RBrace)) {
            Param p;
            p.name = expect_ident();
            expect(Tok::Colon);
            p.value = parse_expr();
            if (check(Tok::Tilde)) { ++pos_; p.modulation = parse_expr(); p.has_mod = true; }
            p.temporal = parse_temporal_ops();
            ps.push_back(std::move(p));
        }
        return ps;
    }

    std::vector<TemporalOp> parse_temporal_ops()
    {
        std::vector<TemporalOp> ops;
        for (;;) {
            TemporalOp op;
            if (check(Tok::ShiftRight)) { ++pos_; op.kind = TemporalOp::Delay; op.dur = parse_duration(); }
            else if (check(Tok::Diamond)) { ++pos_; op.kind = TemporalOp::Smooth; op.dur = parse_duration(); }
            else if (check(Tok::BangBang)) { ++pos_; op.kind = TemporalOp::Trigger; op.dur = parse_duration(); }
            else if (check(Tok::DotDot)) {
                ++pos_;
                op.kind = TemporalOp::Range;
                expect(Tok::LBracket);
                op.lo = parse_expr();
                expect(Tok::Comma);
                op.hi = parse_expr();
                expect(Tok::RBracket);
            } else break;
            ops.push_back(std::move(op));
        }
        return ops;
    }

    std::vector<Stage> parse_stage_pipeline()
    {
        std::vector<Stage> st;
        st.push_back(parse_stage());
        while (check(Tok::Pipe)) { ++pos_; st.push_back(parse_stage()); }
        return st;
    }
    Stage parse_stage()
    {
        Stage s;
        s.name = expect_ident();
        expect(Tok::LParen);
        s.args = parse_arg_list();
        expect(Tok::RParen);
        return s;
    }
    std::vector<Arg> parse_arg_list()
    {
        std::vector<Arg> args;
        if (check(Tok::RParen)) return args;
        args.push_back(parse_arg());
        while (check(Tok::Comma)) { ++pos_; args.push_back(parse_arg()); }
        return args;
    }
    Arg parse_arg()
    {
        Arg a;
        const Token* t0 = peek();
        const Token* t1 = peek_at(1);
        if (t0 && t0->kind == Tok::Ident && t1 && t1->kind == Tok::Colon) {
            a.named = true;
            a.name = expect_ident();
            expect(Tok::Colon);
        }
        a.value = parse_expr();
        return a;
    }

    // -- arc -----------------------------------------------------------------
    static bool is_num(const Token* t) { return t && (t->kind == Tok::Integer || t->kind == Tok::Float); }
    ArcBlock parse_arc()
    {
        expect(Tok::Arc);
        expect(Tok::LBrace);
        ArcBlock blk;
        while (!at_end() && !check(Tok::RBrace)) {
            const Token* t1 = peek_at(1);
            if (is_num(peek()) && t1 && t1->kind == Tok::Colon && is_num(peek_at(2))) {
                for (auto& e : parse_arc_timestamp_entry()) blk.entries.push_back(std::move(e));
            } else {
                blk.entries.push_back(parse_arc_entry());
            }
        }
        expect(Tok::RBrace);
        return blk;
    }
    std::vector<ArcEntry> parse_arc_timestamp_entry()
    {
        expect_number();
        expect(Tok::Colon);
        expect_number();
        if (check(Tok::StringLit)) expect_string();
        expect(Tok::LBrace);
        std::vector<ArcEntry> es;
        while (!at_end() && !check(Tok::RBrace)) {
            ArcEntry e;
            e.target = expect_ident();
            if (check(Tok::Colon)) {
                ++pos_;
                e.from = parse_expr();
                e.to = e.from;
                e.dur = Duration{DurKind::Seconds, 0.0, 0};
                es.push_back(std::move(e));
            } else if (check(Tok::Arrow)) {
                ++pos_;
                e.to = parse_expr();
                if (check(Tok::Ease)) {
                    ++pos_;
                    expect(Tok::LParen);
                    e.easing = expect_ident();
                    e.has_easing = true;
                    expect(Tok::RParen);
                }
                if (check(Tok::Over)) { ++pos_; e.dur = parse_duration(); }
                else e.dur = Duration{DurKind::Seconds, 0.0, 0};
                e.from = Expr::number(0.0);
                es.push_back(std::move(e));
            }
        }
        expect(Tok::RBrace);
        return es;
    }
    ArcEntry parse_arc_entry()
    {
        ArcEntry e;
        e.target = parse_dotted_ident();
        expect(Tok::Colon);
        e.from = parse_expr();
        expect(Tok::Arrow);
        e.to = parse_expr();
        expect(Tok::Over);
        e.dur = parse_duration();
        if (check(Tok::Ident)) { e.easing = expect_ident(); e.has_easing = true; }
        return e;
    }
    std::string parse_dotted_ident()
    {
        std::string s = expect_ident();
        while (check(Tok::Dot)) { ++pos_; s += "." + expect_ident(); }
        return s;
    }
    Duration parse_duration()
    {
        auto [a, b] = current_pos();
        Token t;
        if (!try_advance(t)) throw ParseFail{perr(a, b, "expected duration, found end of input")};
        switch (t.kind) {
        case Tok::Seconds: return Duration{DurKind::Seconds, t.num, 0};
        case Tok::Millis: return Duration{DurKind::Millis, t.num, 0};
        case Tok::Bars: return Duration{DurKind::Bars, 0.0, t.inum};
        case Tok::Float:
            throw ParseFail{perr(a, b, "expected duration (e.g. 2s, 500ms, 4bars), found bare number " + fmt_f64_display(t.num))};
        case Tok::Integer:
            throw ParseFail{perr(a, b, "expected duration (e.g. 2s, 500ms, 4bars), found bare number " + std::to_string((long long)t.inum))};
        default:
            throw ParseFail{perr(a, b, "expected duration, found `" + token_display(t) + "`")};
        }
    }

    // -- resonate ------------------------------------------------------------
    ResonateBlock parse_resonate()
    {
        expect(Tok::Resonate);
        expect(Tok::LBrace);
        ResonateBlock blk;
        while (!at_end() && !check(Tok::RBrace)) {
            ResonateEntry e;
            e.source = expect_ident();
            expect(Tok::Arrow);
            e.target = expect_ident();
            expect(Tok::Dot);
            e.field = expect_ident();
            expect(Tok::Star);
            e.weight = parse_expr();
            blk.entries.push_back(std::move(e));
        }
        expect(Tok::RBrace);
        return blk;
    }

    // -- listen / voice --------------------------------------------------------
    std::vector<NamedNode> parse_named_nodes(Tok kw)
    {
        expect(kw);
        expect(Tok::LBrace);
        std::vector<NamedNode> nodes;
        while (!at_end() && !check(Tok::RBrace)) {
            NamedNode n;
            n.name = expect_ident();
            expect(Tok::Colon);
            n.kind = expect_ident();
            if (check(Tok::LParen)) n.params = parse_paren_params();
            nodes.push_back(std::move(n));
        }
        expect(Tok::RBrace);
        return nodes;
    }

    // -- score ---------------------------------------------------------------
    bool is_score_ref_ident() const
    {
        const Token* t = peek();
        if (!t || t->kind != Tok::Ident) return false;
        return t->text != "motif" && t->text != "phrase" && t->text != "section" && t->text != "arrange";
    }
    ScoreBlock parse_score()
    {
        expect(Tok::Score);
        ScoreBlock sb;
        if (check(Tok::Ident) && peek()->text == "tempo") {
            ++pos_;
            expect(Tok::LParen);
            sb.tempo_bpm = expect_number();
            expect(Tok::RParen);
        }
        expect(Tok::LBrace);
        while (!at_end() && !check(Tok::RBrace)) {
            const Token* t = peek();
            if (t->kind == Tok::Ident && t->text == "motif") {
                ++pos_;
                Motif m;
                m.name = expect_ident();
                expect(Tok::LBrace);
                while (!at_end() && !check(Tok::RBrace)) m.entries.push_back(parse_arc_entry());
                expect(Tok::RBrace);
                sb.motifs.push_back(std::move(m));
            } else if (t->kind == Tok::Ident && t->text == "phrase") {
                ++pos_;
                NameList p;
                p.name = expect_ident();
                expect(Tok::Eq);
                while (is_score_ref_ident()) {
                    p.refs.push_back(expect_ident());
                    if (check(Tok::Pipe)) ++pos_;
                }
                sb.phrases.push_back(std::move(p));
            } else if (t->kind == Tok::Ident && t->text == "section") {
                ++pos_;
                NameList s;
                s.name = expect_ident();
                expect(Tok::Eq);
                while (is_score_ref_ident()) s.refs.push_back(expect_ident());
                sb.sections.push_back(std::move(s));
            } else if (t->kind == Tok::Ident && t->text == "arrange") {
                ++pos_;
                expect(Tok::Colon);
                while (is_score_ref_ident()) sb.arrange.push_back(expect_ident());
            } else {
                auto [a, b] = current_pos();
                throw ParseFail{perr(a, b, "expected `motif`, `phrase`, `section`, or `arrange` in score, found `" +
                                           token_display(*t) + "`")};
            }
        }
        expect(Tok::RBrace);
        return sb;
    }

    // -- breed ---------------------------------------------------------------
    BreedBlock parse_breed()
    {
        expect(Tok::Breed);
        BreedBlock b;
        b.name = expect_string();
        expect(Tok::From);
        b.parents.push_back(expect_string());
        while (check(Tok::Plus)) { ++pos_; b.parents.push_back(expect_string()); }
        expect(Tok::LBrace);
        while (!at_end() && !check(Tok::RBrace)) {
            if (check(Tok::Inherit)) {
                ++pos_;
                InheritRule r;
                r.target = expect_ident();
                expect(Tok::Colon);
                r.strategy = expect_ident();
                expect(Tok::LParen);
                r.weight = expect_number();
                expect(Tok::RParen);
                b.inherit.push_back(std::move(r));
            } else if (check(Tok::Mutate)) {
                ++pos_;
                Mutation m;
                m.target = expect_ident();
                expect(Tok::Colon);
                double v = expect_number();
                m.range = v < 0 ? -v : v;
                b.mutations.push_back(std::move(m));
            } else {
                auto [a, bb] = current_pos();
                throw ParseFail{perr(a, bb, "expected `inherit` or `mutate` in breed, found `" + token_display(*peek()) + "`")};
            }
        }
        expect(Tok::RBrace);
        return b;
    }

    // -- gravity -------------------------------------------------------------
    GravityBlock parse_gravity()
    {
        expect(Tok::Gravity);
        expect(Tok::LBrace);
        GravityBlock g;
        while (!at_end() && !check(Tok::RBrace)) {
            std::string key = expect_ident();
            expect(Tok::Colon);
            if (key == "rule") g.force_law = parse_expr();
            else if (key == "damping") g.damping = expect_number();
            else if (key == "bounds") {
                std::string mode = expect_ident();
                if (mode == "reflect") g.bounds = Bounds::Reflect;
                else if (mode == "wrap") g.bounds = Bounds::Wrap;
                else if (mode == "none") g.bounds = Bounds::None;
                else {
                    auto [a, b] = current_pos();
                    throw ParseFail{perr(a, b, "unknown bounds mode `" + mode + "`, expected reflect/wrap/none")};
                }
            } else {
                auto [a, b] = current_pos();
                throw ParseFail{perr(a, b, "unknown gravity property `" + key + "`, expected rule/damping/bounds")};
            }
            if (check(Tok::Comma)) ++pos_;
        }
        expect(Tok::RBrace);
        return g;
    }

    // -- project -------------------------------------------------------------
    ProjectBlock parse_project()
    {
        expect(Tok::Project);
        ProjectBlock p;
        std::string mode = expect_ident();
        if (mode == "flat") p.mode = ProjectMode::Flat;
        else if (mode == "dome") p.mode = ProjectMode::Dome;
        else if (mode == "cube") p.mode = ProjectMode::Cube;
        else if (mode == "led") p.mode = ProjectMode::Led;
        else {
            auto [a, b] = current_pos();
            throw ParseFail{perr(a, b, "unknown projection mode `" + mode + "`, expected flat/dome/cube/led")};
        }
        if (check(Tok::LParen)) p.params = parse_paren_params();
        expect(Tok::LBrace);
        while (!at_end() && !check(Tok::RBrace)) {
            std::string key = expect_ident();
            expect(Tok::Colon);
            if (key == "source") p.source = expect_ident();
            else expect_ident();
            if (check(Tok::Comma)) ++pos_;
        }
        expect(Tok::RBrace);
        return p;
    }

    // -- lens ----------------------------------------------------------------
    Lens parse_lens()
    {
        expect(Tok::Lens);
        Lens l;
        if (check(Tok::Ident)) { l.name = expect_ident(); l.has_name = true; }
        expect(Tok::LBrace);
        while (!at_end() && !check(Tok::RBrace)) {
            const Token* t0 = peek();
            const Token* t1 = peek_at(1);
            if (t0->kind == Tok::Ident && t0->text == "post" && t1 && t1->kind == Tok::Colon) {
                pos_ += 2;
                l.post.push_back(parse_stage());
                while (check(Tok::Pipe)) { ++pos_; l.post.push_back(parse_stage()); }
                continue;
            }
            Param p;
            p.name = expect_ident();
            expect(Tok::Colon);
            p.value = parse_expr();
            l.properties.push_back(std::move(p));
        }
        expect(Tok::RBrace);
        return l;
    }

    // -- react ---------------------------------------------------------------
    std::vector<Reaction> parse_react()
    {
        expect(Tok::React);
        expect(Tok::LBrace);
        std::vector<Reaction> rs;
        while (!at_end() && !check(Tok::RBrace)) {
            Reaction r;
            r.signal = parse_expr();
            expect(Tok::Arrow);
            r.action = parse_expr();
            rs.push_back(std::move(r));
        }
        expect(Tok::RBrace);
        return rs;
    }

    // -- define --------------------------------------------------------------
    DefineBlock parse_define()
    {
        expect(Tok::Define);
        DefineBlock d;
        d.name = expect_ident();
        if (check(Tok::LParen)) {
            ++pos_;
            while (!at_end() && !check(Tok::RParen)) {
                d.params.push_back(expect_ident());
                if (!check(Tok::RParen)) expect(Tok::Comma);
            }
            expect(Tok::RParen);
        }
        expect(Tok::LBrace);
        d.body.push_back(parse_stage());
        while (check(Tok::Pipe)) { ++pos_; d.body.push_back(parse_stage()); }
        expect(Tok::RBrace);
        return d;
    }

    // -- expressions ---------------------------------------------------------
    static Expr make_bin(BinOp op, Expr l, Expr r)
    {
        Expr e;
        e.kind = Expr::Bin; e.op = op;
        e.kids.push_back(std::move(l));
        e.kids.push_back(std::move(r));
        return e;
    }
    Expr parse_expr()
    {
        Expr e = parse_comparison();
        if (check(Tok::Question)) {
            ++pos_;
            Expr t = parse_expr();
            expect(Tok::Colon);
            Expr f = parse_expr();
            Expr out;
            out.kind = Expr::Ternary;
            out.kids.push_back(std::move(e));
            out.kids.push_back(std::move(t));
            out.kids.push_back(std::move(f));
            return out;
        }
        return e;
    }
    Expr parse_comparison()
    {
        Expr left = parse_additive();
        while (check(Tok::Greater) || check(Tok::Less)) {
            BinOp op = advance().kind == Tok::Greater ? BinOp::Gt : BinOp::Lt;
            Expr right = parse_additive();
            left = make_bin(op, std::move(left), std::move(right));
        }
        return left;
    }
    Expr parse_additive()
    {
        Expr left = parse_term();
        while (check(Tok::Plus) || check(Tok::Minus)) {
            BinOp op = advance().kind == Tok::Plus ? BinOp::Add : BinOp::Sub;
            Expr right = parse_term();
            left = make_bin(op, std::move(left), std::move(right));
        }
        return left;
    }
    Expr parse_term()
    {
        Expr left = parse_factor();
        while (check(Tok::Star) || check(Tok::Slash)) {
            BinOp op = advance().kind == Tok::Star ? BinOp::Mul : BinOp::Div;
            Expr right = parse_factor();
            left = make_bin(op, std::move(left), std::move(right));
        }
        return left;
    }
    Expr parse_factor()
    {
        Expr base = parse_atom();
        if (check(Tok::Caret)) {
            ++pos_;
            Expr ex = parse_factor();
            return make_bin(BinOp::Pow, std::move(base), std::move(ex));
        }
        return base;
    }
    Expr parse_atom()
    {
        auto [a, b] = current_pos();
        const Token* t = peek();
        if (!t) throw ParseFail{perr(a, b, "unexpected end of input in expression")};
        Expr e;
        switch (t->kind) {
        case Tok::Float: e = Expr::number(t->num); ++pos_; return e;
        case Tok::Integer: e = Expr::number((double)t->inum); ++pos_; return e;
        case Tok::Degrees: e = Expr::number(t->num); ++pos_; return e;
        case Tok::Seconds: e.kind = Expr::Dur; e.dur = Duration{DurKind::Seconds, t->num, 0}; ++pos_; return e;
        case Tok::Millis: e.kind = Expr::Dur; e.dur = Duration{DurKind::Millis, t->num, 0}; ++pos_; return e;
        case Tok::Bars: e.kind = Expr::Dur; e.dur = Duration{DurKind::Bars, 0.0, t->inum}; ++pos_; return e;
        case Tok::StringLit: e.kind = Expr::String; e.s = t->text; ++pos_; return e;
        case Tok::Ident: {
            std::string name = t->text;
            ++pos_;
            if (check(Tok::LParen)) {
                ++pos_;
                e.kind = Expr::Call;
                e.s = name;
                e.args = parse_arg_list();
                expect(Tok::RParen);
                return e;
            }
            if (check(Tok::Dot)) {
                ++pos_;
                e.kind = Expr::Dotted;
                e.s = name;
                e.field = expect_ident();
                return e;
            }
            e.kind = Expr::Ident;
            e.s = name;
            return e;
        }
        case Tok::LParen: {
            ++pos_;
            Expr inner = parse_expr();
            expect(Tok::RParen);
            e.kind = Expr::Paren;
            e.kids.push_back(std::move(inner));
            return e;
        }
        case Tok::LBracket: {
            ++pos_;
            e.kind = Expr::Array;
            if (!check(Tok::RBracket)) {
                e.kids.push_back(parse_expr());
                while (check(Tok::Comma)) { ++pos_; e.kids.push_back(parse_expr()); }
            }
            expect(Tok::RBracket);
            return e;
        }
        case Tok::Minus: {
            ++pos_;
            Expr inner = parse_factor();
            e.kind = Expr::Neg;
            e.kids.push_back(std::move(inner));
            return e;
        }
        default:
            throw ParseFail{perr(a, b, "unexpected token `" + token_display(*t) + "` in expression")};
        }
    }
};

}  // namespace

CompileError parse(std::vector<Token> toks, Program& out)
{
    out = Program{};
    Parser p(std::move(toks));
    return p.parse_program(out);
}

}  // namespace game
