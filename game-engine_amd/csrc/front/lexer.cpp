// Tokenizer for .game source, plus error formatting helpers.
// Behaviour follows the reference's logos lexer (game-compiler/src/lexer.rs:11-159): whitespace,
// `//` and `#` comments are skipped; the longest match wins; keywords beat identifiers of the
// same length; numbers may carry a unit suffix (s, ms, bars, deg).
#include "front.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cerrno>
#include <algorithm>

namespace game {

// ── number formatting ────────────────────────────────────────────────────
std::string fmt_fixed6(double v)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%.6f", v);
    return buf;
}

// Rust's Display for f64: shortest digits that round-trip, never exponent notation.
std::string fmt_f64_display(double v)
{
    if (v != v) return "NaN";
    if (v == 1.0 / 0.0) return "inf";
    if (v == -1.0 / 0.0) return "-inf";
    char buf[64];
    int prec = 1;
    for (; prec <= 17; ++prec) {
        snprintf(buf, sizeof buf, "%.*e", prec - 1, v);
        if (strtod(buf, nullptr) == v) break;
    }
    // buf = d.ddddde[+-]xx ; expand to plain decimal
    std::string s(buf);
    bool neg = s[0] == '-';
    if (neg) s.erase(0, 1);
    size_t epos = s.find('e');
    int exp10 = atoi(s.c_str() + epos + 1);
    std::string digits;
    for (size_t i = 0; i < epos; ++i) if (s[i] != '.') digits.push_back(s[i]);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    std::string out;
    if (exp10 >= 0) {
        if ((int)digits.size() <= exp10 + 1) {
            out = digits + std::string(exp10 + 1 - digits.size(), '0');
        } else {
            out = digits.substr(0, exp10 + 1) + "." + digits.substr(exp10 + 1);
        }
    } else {
        out = "0." + std::string(-exp10 - 1, '0') + digits;
    }
    if (out == "0" && neg) return "-0";
    return neg ? "-" + out : out;
}

// ── token display (token.rs:82-149) ──────────────────────────────────────
std::string token_display(const Token& t)
{
    switch (t.kind) {
    case Tok::Cinematic: return "cinematic"; case Tok::Layer: return "layer"; case Tok::Import: return "import";
    case Tok::As: return "as"; case Tok::Arc: return "arc"; case Tok::Resonate: return "resonate";
    case Tok::Memory: return "memory"; case Tok::Cast: return "cast"; case Tok::Over: return "over";
    case Tok::Listen: return "listen"; case Tok::Voice: return "voice"; case Tok::Score: return "score";
    case Tok::Breed: return "breed"; case Tok::From: return "from"; case Tok::Inherit: return "inherit";
    case Tok::Mutate: return "mutate"; case Tok::Gravity: return "gravity"; case Tok::Project: return "project";
    case Tok::Signals: return "signals"; case Tok::Route: return "route"; case Tok::Hear: return "hear";
    case Tok::Feel: return "feel"; case Tok::Lens: return "lens"; case Tok::React: return "react";
    case Tok::Define: return "define"; case Tok::Expose: return "expose"; case Tok::Ease: return "ease";
    case Tok::All: return "ALL";
    case Tok::Pipe: return "|"; case Tok::Tilde: return "~"; case Tok::LBrace: return "{"; case Tok::RBrace: return "}";
    case Tok::LParen: return "("; case Tok::RParen: return ")"; case Tok::LBracket: return "["; case Tok::RBracket: return "]";
    case Tok::Colon: return ":"; case Tok::Comma: return ","; case Tok::Dot: return "."; case Tok::Plus: return "+";
    case Tok::Minus: return "-"; case Tok::Star: return "*"; case Tok::Slash: return "/"; case Tok::Caret: return "^";
    case Tok::Eq: return "="; case Tok::Arrow: return "->"; case Tok::ShiftRight: return ">>"; case Tok::Diamond: return "<>";
    case Tok::BangBang: return "!!"; case Tok::DotDot: return ".."; case Tok::Greater: return ">"; case Tok::Less: return "<";
    case Tok::Question: return "?";
    case Tok::Float: return fmt_f64_display(t.num);
    case Tok::Integer: return std::to_string((long long)t.inum);
    case Tok::StringLit: return "\"" + t.text + "\"";
    case Tok::Ident: return t.text;
    case Tok::Seconds: return fmt_f64_display(t.num) + "s";
    case Tok::Millis: return fmt_f64_display(t.num) + "ms";
    case Tok::Bars: return std::to_string((long long)t.inum) + "bars";
    case Tok::Degrees: return fmt_f64_display(t.num) + "deg";
    }
    return "?";
}

// ── errors ───────────────────────────────────────────────────────────────
static const char* code_name(int c)
{
    static const char* n[] = {"", "E001", "E002", "E003", "E004", "E005", "E006", "E007", "E008", "E009", "E010"};
    return (c >= 1 && c <= 10) ? n[c] : "";
}

std::string CompileError::display() const
{
    std::string inner;
    switch (kind) {
    case ErrKind::Lex:
        inner = "lex error at Span { start: " + std::to_string(a) + ", end: " + std::to_string(b) + " }: " + message; break;
    case ErrKind::Parse:
        inner = "parse error at " + std::to_string(a) + ":" + std::to_string(b) + ": " + message; break;
    case ErrKind::Validation: inner = "validation error: " + message; break;
    case ErrKind::Codegen: inner = "codegen error: " + message; break;
    case ErrKind::None: return "";
    }
    if (!help.empty()) inner += "\n  help: " + help;
    return inner;
}

std::string CompileError::render_with_source(const std::string& src) const
{
    std::string out = code ? std::string("error[") + code_name(code) + "]: " + display() : "error: " + display();
    if (kind == ErrKind::Lex && a <= src.size()) {
        size_t line_num = std::count(src.begin(), src.begin() + a, '\n') + 1;
        size_t nl = src.rfind('\n', a == 0 ? 0 : a - 1);
        size_t line_start = (a == 0 || nl == std::string::npos) ? 0 : nl + 1;
        if (a > 0 && src[a - 1] == '\n') line_start = a;
        size_t le = src.find('\n', a);
        size_t line_end = le == std::string::npos ? src.size() : le;
        std::string line = src.substr(line_start, line_end - line_start);
        size_t col = a - line_start;
        size_t room = line.size() > col ? line.size() - col : 0;
        size_t ul = std::min(std::max<size_t>(b - a, 1), room);
        std::string ln = std::to_string(line_num);
        out += "\n --> line " + ln + ":" + std::to_string(col);
        out += "\n  " + ln + " | " + line;
        out += "\n  " + std::string(ln.size(), ' ') + " | " + std::string(col, ' ') + std::string(ul, '^');
    }
    if (!help.empty()) out += "\n  help: " + help;
    return out;
}

size_t levenshtein(const std::string& a, const std::string& b)
{
    std::vector<size_t> prev(b.size() + 1), cur(b.size() + 1);
    for (size_t j = 0; j <= b.size(); ++j) prev[j] = j;
    for (size_t i = 1; i <= a.size(); ++i) {
        cur[0] = i;
        for (size_t j = 1; j <= b.size(); ++j) {
            size_t cost = a[i - 1] == b[j - 1] ? 0 : 1;
            cur[j] = std::min(std::min(prev[j] + 1, cur[j - 1] + 1), prev[j - 1] + cost);
        }
        std::swap(prev, cur);
    }
    return prev[b.size()];
}

const char* suggest_similar(const std::string& typed, const std::vector<const char*>& candidates)
{
    const char* best = nullptr;
    size_t best_d = 0;
    for (const char* c : candidates) {
        size_t d = levenshtein(typed, c);
        if (d <= 2 && d < typed.size() && (!best || d < best_d)) { best = c; best_d = d; }
    }
    return best;
}

// ── lexer ────────────────────────────────────────────────────────────────
namespace {

struct Kw { const char* text; Tok kind; };
const Kw KEYWORDS[] = {
    {"cinematic", Tok::Cinematic}, {"layer", Tok::Layer}, {"import", Tok::Import}, {"as", Tok::As},
    {"arc", Tok::Arc}, {"resonate", Tok::Resonate}, {"over", Tok::Over}, {"memory", Tok::Memory},
    {"cast", Tok::Cast}, {"listen", Tok::Listen}, {"voice", Tok::Voice}, {"score", Tok::Score},
    {"breed", Tok::Breed}, {"from", Tok::From}, {"inherit", Tok::Inherit}, {"mutate", Tok::Mutate},
    {"gravity", Tok::Gravity}, {"project", Tok::Project}, {"signals", Tok::Signals}, {"route", Tok::Route},
    {"hear", Tok::Hear}, {"feel", Tok::Feel}, {"lens", Tok::Lens}, {"react", Tok::React},
    {"define", Tok::Define}, {"expose", Tok::Expose}, {"ease", Tok::Ease}, {"ALL", Tok::All},
};

inline bool is_digit(unsigned char c) { return c >= '0' && c <= '9'; }
inline bool is_ident_start(unsigned char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_' || c >= 0x80; }
inline bool is_ident_char(unsigned char c) { return is_ident_start(c) || is_digit(c); }

std::string unescape(const std::string& inner)   // lexer.rs:188-211
{
    std::string r;
    r.reserve(inner.size());
    for (size_t i = 0; i < inner.size(); ++i) {
        char c = inner[i];
        if (c == '\\') {
            if (i + 1 >= inner.size()) { r.push_back('\\'); break; }
            char n = inner[++i];
            switch (n) {
            case 'n': r.push_back('\n'); break;
            case 't': r.push_back('\t'); break;
            case 'r': r.push_back('\r'); break;
            case '\\': r.push_back('\\'); break;
            case '"': r.push_back('"'); break;
            default: r.push_back('\\'); r.push_back(n); break;
            }
        } else {
            r.push_back(c);
        }
    }
    return r;
}

double parse_f64(const std::string& s) { return strtod(s.c_str(), nullptr); }
int64_t parse_i64(const std::string& s)   // str::parse::<i64>().unwrap_or(0): overflow -> 0
{
    errno = 0;
    long long v = strtoll(s.c_str(), nullptr, 10);
    if (errno == ERANGE) return 0;
    return v;
}

}  // namespace

CompileError lex(const std::string& src, std::vector<Token>& out)
{
    out.clear();
    const size_t n = src.size();
    size_t i = 0;
    auto lex_err = [&](size_t s, size_t e) {
        CompileError err;
        err.kind = ErrKind::Lex; err.a = s; err.b = e;
        err.message = "unexpected character: '" + src.substr(s, e - s) + "'";
        return err;
    };
    while (i < n) {
        unsigned char c = (unsigned char)src[i];
        if (c == ' ' || c == '\t' || c == '\r' || c == '\n') { ++i; continue; }
        if (c == '#' || (c == '/' && i + 1 < n && src[i + 1] == '/')) {
            while (i < n && src[i] != '\n') ++i;
            continue;
        }
        Token t;
        t.start = i;
        if (is_digit(c)) {
            size_t j = i;
            while (j < n && is_digit((unsigned char)src[j])) ++j;
            bool is_float = false;
            if (j + 1 < n && src[j] == '.' && is_digit((unsigned char)src[j + 1])) {
                is_float = true;
                ++j;
                while (j < n && is_digit((unsigned char)src[j])) ++j;
            }
            std::string numtext = src.substr(i, j - i);
            // unit suffixes; the longest matching alternative wins
            auto starts = [&](const char* suf) { return src.compare(j, strlen(suf), suf) == 0; };
            if (!is_float && starts("bars")) { t.kind = Tok::Bars; t.inum = parse_i64(numtext); j += 4; }
            else if (starts("deg")) { t.kind = Tok::Degrees; t.num = parse_f64(numtext); j += 3; }
            else if (starts("ms")) { t.kind = Tok::Millis; t.num = parse_f64(numtext); j += 2; }
            else if (starts("s")) { t.kind = Tok::Seconds; t.num = parse_f64(numtext); j += 1; }
            else if (is_float) { t.kind = Tok::Float; t.num = parse_f64(numtext); }
            else { t.kind = Tok::Integer; t.inum = parse_i64(numtext); }
            t.end = j;
            out.push_back(std::move(t));
            i = j;
            continue;
        }
        if (is_ident_start(c)) {
            size_t j = i + 1;
            while (j < n && is_ident_char((unsigned char)src[j])) ++j;
            std::string word = src.substr(i, j - i);
            t.kind = Tok::Ident;
            for (const Kw& k : KEYWORDS) if (word == k.text) { t.kind = k.kind; break; }
            if (t.kind == Tok::Ident) t.text = word;   // `true`/`false` stay identifiers (lexer.rs:244-245)
            t.end = j;
            out.push_back(std::move(t));
            i = j;
            continue;
        }
        if (c == '"') {
            size_t j = i + 1;
            bool closed = false;
            while (j < n) {
                if (src[j] == '\\') {
                    // `\\.`: the dot does not match a newline
                    if (j + 1 < n && src[j + 1] != '\n') { j += 2; continue; }
                    break;
                }
                if (src[j] == '"') { closed = true; break; }
                ++j;
            }
            if (!closed) return lex_err(i, i + 1);
            t.kind = Tok::StringLit;
            t.text = unescape(src.substr(i + 1, j - i - 1));
            t.end = j + 1;
            out.push_back(std::move(t));
            i = j + 1;
            continue;
        }
        auto two = [&](char a, char b) { return c == (unsigned char)a && i + 1 < n && src[i + 1] == b; };
        size_t len = 1;
        if (two('>', '>')) { t.kind = Tok::ShiftRight; len = 2; }
        else if (two('<', '>')) { t.kind = Tok::Diamond; len = 2; }
        else if (two('!', '!')) { t.kind = Tok::BangBang; len = 2; }
        else if (two('.', '.')) { t.kind = Tok::DotDot; len = 2; }
        else if (two('-', '>')) { t.kind = Tok::Arrow; len = 2; }
        else switch (c) {
            case '>': t.kind = Tok::Greater; break; case '<': t.kind = Tok::Less; break;
            case '?': t.kind = Tok::Question; break; case '|': t.kind = Tok::Pipe; break;
            case '~': t.kind = Tok::Tilde; break; case '+': t.kind = Tok::Plus; break;
            case '-': t.kind = Tok::Minus; break; case '*': t.kind = Tok::Star; break;
            case '/': t.kind = Tok::Slash; break; case '^': t.kind = Tok::Caret; break;
            case '=': t.kind = Tok::Eq; break; case '{': t.kind = Tok::LBrace; break;
            case '}': t.kind = Tok::RBrace; break; case '(': t.kind = Tok::LParen; break;
            case ')': t.kind = Tok::RParen; break; case '[': t.kind = Tok::LBracket; break;
            case ']': t.kind = Tok::RBracket; break; case ':': t.kind = Tok::Colon; break;
            case ',': t.kind = Tok::Comma; break; case '.': t.kind = Tok::Dot; break;
            default: return lex_err(i, i + 1);
        }
        t.end = i + len;
        out.push_back(std::move(t));
        i += len;
    }
    return CompileError{};
}

}  // namespace game
