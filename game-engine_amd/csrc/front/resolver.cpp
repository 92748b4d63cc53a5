// File imports: `import "path" as alias` / `import "path" expose a, b` / `expose ALL`
// (game-compiler/src/resolver.rs:18-147). The imported file is lexed and parsed, its own imports are
// resolved first, and the `define` blocks it holds are merged into the first cinematic of the importer.
// URI-schemed imports (shadertoy://, midi://, osc://, camera://; adapters/mod.rs) are host adapters in the
// reference, not files; they are left in Program::imports untouched.
#include "front.hpp"
#include <algorithm>
#include <filesystem>
#include <fstream>
#include <set>
#include <sstream>

namespace game {
namespace fs = std::filesystem;

namespace {
CompileError verr(std::string msg)
{
    CompileError e;
    e.kind = ErrKind::Validation; e.message = std::move(msg);
    return e;
}

bool is_uri(const std::string& p) { return p.find("://") != std::string::npos; }

// resolver.rs:67-89: relative to the importing file's directory, then each library directory, as written and
// with ".game" appended.
bool find_file(const std::string& path, const fs::path& base, const std::vector<std::string>& lib_dirs, fs::path& out)
{
    std::vector<fs::path> dirs{base};
    for (const std::string& d : lib_dirs) dirs.emplace_back(d);
    std::error_code ec;
    for (const fs::path& d : dirs) {
        fs::path direct = d / path;
        if (fs::is_regular_file(direct, ec)) { out = direct; return true; }
        fs::path with_ext = d / (path + ".game");
        if (fs::is_regular_file(with_ext, ec)) { out = with_ext; return true; }
    }
    return false;
}

CompileError resolve_recursive(Program& program, const fs::path& base, const std::vector<std::string>& lib_dirs, std::set<std::string>& visited)
{
    std::vector<Import> imports;
    std::vector<Import> kept;
    imports.swap(program.imports);
    for (const Import& imp : imports) {
        if (is_uri(imp.path)) { kept.push_back(imp); continue; }
        fs::path file;
        if (!find_file(imp.path, base, lib_dirs, file)) return verr("import not found: '" + imp.path + "'");
        std::error_code ec;
        fs::path canonical = fs::canonical(file, ec);
        if (ec) return verr("cannot canonicalize '" + file.string() + "': " + ec.message());
        if (!visited.insert(canonical.string()).second) return verr("circular import detected: '" + imp.path + "'");

        std::ifstream in(canonical, std::ios::binary);
        if (!in) return verr("io error: cannot read '" + canonical.string() + "'");
        std::stringstream ss;
        ss << in.rdbuf();
        std::vector<Token> toks;
        CompileError e = lex(ss.str(), toks);
        if (!e.ok()) return e;
        Program imported;
        e = parse(std::move(toks), imported);
        if (!e.ok()) return e;
        e = resolve_recursive(imported, canonical.parent_path(), lib_dirs, visited);
        if (!e.ok()) return e;

        // resolver.rs:92-120: every define of every cinematic of the imported file, filtered by the import style
        std::vector<DefineBlock> defines;
        const bool all = std::find(imp.exposed.begin(), imp.exposed.end(), "ALL") != imp.exposed.end();
        for (const Cinematic& c : imported.cinematics) {
            for (const DefineBlock& d : c.defines) {
                if (imp.exposed.empty()) {            // `as` style: the alias becomes a name prefix
                    DefineBlock n = d;
                    n.name = imp.alias + "." + d.name;
                    defines.push_back(std::move(n));
                } else if (all || std::find(imp.exposed.begin(), imp.exposed.end(), d.name) != imp.exposed.end()) {
                    defines.push_back(d);
                }
            }
        }
        // resolver.rs:123-147
        if (defines.empty() && !imp.exposed.empty()) {
            std::string names = "[";
            for (size_t i = 0; i < imp.exposed.size(); ++i) names += (i ? ", \"" : "\"") + imp.exposed[i] + "\"";
            return verr("import '" + imp.path + "': no matching defines found for exposed names " + names + "]");
        }
        if (defines.empty()) continue;
        if (program.cinematics.empty()) return verr("cannot merge imported defines: program has no cinematics");
        for (DefineBlock& d : defines) program.cinematics[0].defines.push_back(std::move(d));
    }
    program.imports = std::move(kept);
    return CompileError{};
}
}  // namespace

CompileError resolve_imports(Program& program, const std::string& base_dir, const std::vector<std::string>& lib_dirs)
{
    std::set<std::string> visited;
    std::error_code ec;
    fs::path root = fs::canonical(fs::path(base_dir) / "__root__", ec);   // resolver.rs:24-26 (only if such a file exists)
    if (!ec) visited.insert(root.string());
    return resolve_recursive(program, fs::path(base_dir), lib_dirs, visited);
}

}  // namespace game
