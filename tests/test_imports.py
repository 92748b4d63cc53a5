"""File imports (game-compiler/src/resolver.rs:18-147; its tests :162-278 are restated here). The reference's
compile() leaves file imports alone (lib.rs:150); compile(..., base_dir=...) resolves them first."""
import os

import numpy as np
import pytest

from conftest import REFERENCE, reference_available

LIB = '''cinematic "lib" {
  define soft_orb(size, power) { circle(size) | glow(power) }
  define frame(size) { polygon(6.0, size) | onion(0.02) | glow(2.0) }
}'''


def defines_used(game, src, base, libs=()):
    prog = game.compile(src, base_dir=str(base), lib_dirs=[str(d) for d in libs])
    return prog.cinematics[0]


def test_import_found_relative_with_and_without_extension(game, tmp_path):
    (tmp_path / "utils.game").write_text(LIB)
    for path in ("utils.game", "utils"):
        cin = defines_used(game, f'import "{path}" expose soft_orb\ncinematic "m" {{ layer a {{ fn: soft_orb(0.25, 3.0) | tint(1.0, 0.5, 0.2) }} }}', tmp_path)
        assert [s["name"] for s in cin.ir["layers"][0]["stages"]] == ["circle", "glow", "tint"]


def test_import_found_in_library_directory(game, tmp_path):
    base, lib = tmp_path / "base", tmp_path / "std"
    base.mkdir()
    lib.mkdir()
    (lib / "shapes.game").write_text(LIB)
    src = 'import "shapes" expose ALL\ncinematic "m" { layer a { fn: frame(0.3) | tint(0.2, 0.8, 1.0) } }'
    cin = defines_used(game, src, base, [lib])
    assert [s["name"] for s in cin.ir["layers"][0]["stages"]] == ["polygon", "onion", "glow", "tint"]
    with pytest.raises(game.GameError, match="import not found: 'shapes'"):
        defines_used(game, src, base)


def test_imported_define_renders_like_the_inline_pipeline(game, oracle, tmp_path):
    (tmp_path / "lib.game").write_text(LIB)
    a = defines_used(game, 'import "lib" expose soft_orb\ncinematic "m" { layer a { fn: translate(0.2, 0.1) | soft_orb(0.25, 3.0) | tint(1.0, 0.5, 0.2) } }', tmp_path)
    b = game.compile('cinematic "m" { layer a { fn: translate(0.2, 0.1) | circle(0.25) | glow(3.0) | tint(1.0, 0.5, 0.2) } }').cinematics[0]
    assert a.blob == b.blob
    row = a.uniform_rows([0.5], 64, 36)[0]
    assert np.array_equal(oracle.render(a.blob, 64, 36, row), oracle.render(b.blob, 64, 36, row))


def test_without_base_dir_imports_are_left_alone(game):
    # the reference's compile(): "standard file import - future work"; an unknown stage is then an error
    with pytest.raises(game.GameError, match="unknown stage function: 'soft_orb'"):
        game.compile('import "lib" expose soft_orb\ncinematic "m" { layer a { fn: soft_orb(0.2, 2.0) } }')
    prog = game.compile('import "nowhere" as n\ncinematic "m" { layer a { fn: circle(0.2) | glow(2.0) } }')
    assert len(prog.cinematics) == 1


def test_circular_import_detected(game, tmp_path):
    (tmp_path / "a.game").write_text('import "b" as b  cinematic "a" {}')
    (tmp_path / "b.game").write_text('import "a" as a  cinematic "b" {}')
    with pytest.raises(game.GameError, match="circular import detected"):
        game.compile_file(str(tmp_path / "a.game"))


def test_transitive_imports_and_expose_filter(game, tmp_path):
    (tmp_path / "inner.game").write_text('cinematic "i" { define dot(r) { circle(r) | glow(1.5) } define unused(r) { ring(r, 0.02) | glow(1.0) } }')
    (tmp_path / "outer.game").write_text('import "inner" expose dot\ncinematic "o" { define twin(r) { mirror(0.0) | dot(r) } }')
    src = 'import "outer" expose ALL\ncinematic "m" { layer a { fn: twin(0.1) | tint(1.0, 1.0, 0.5) } }'
    cin = defines_used(game, src, tmp_path)
    assert [s["name"] for s in cin.ir["layers"][0]["stages"]] == ["mirror", "circle", "glow", "tint"]
    with pytest.raises(game.GameError, match="no matching defines found for exposed names"):
        defines_used(game, 'import "inner" expose nothing_like_it\ncinematic "m" { layer a { fn: circle(0.1) | glow(1.0) } }', tmp_path)


def test_alias_import_prefixes_define_names(game, tmp_path):
    # resolver.rs:99-108: "fx.shimmer"; a stage name cannot carry a dot, so the plain name stays unknown
    (tmp_path / "fx.game").write_text('cinematic "fx" { define shimmer(rate) { circle(rate) | glow(2.0) } }')
    with pytest.raises(game.GameError, match="unknown stage function: 'shimmer'"):
        defines_used(game, 'import "fx" as fx\ncinematic "m" { layer a { fn: shimmer(0.2) } }', tmp_path)
    cin = defines_used(game, 'import "fx" as fx\ncinematic "m" { layer a { fn: circle(0.2) | glow(2.0) } }', tmp_path)
    assert cin.name == "m"


def test_import_needs_a_cinematic_to_merge_into(game, tmp_path):
    (tmp_path / "lib.game").write_text(LIB)
    with pytest.raises(game.GameError, match="cannot merge imported defines: program has no cinematics"):
        game.compile('import "lib" expose ALL', base_dir=str(tmp_path))


def test_uri_imports_are_not_files(game, tmp_path):
    prog = game.compile('import "midi://1" as m\ncinematic "m" { layer a { fn: circle(0.2) | glow(2.0) } }', base_dir=str(tmp_path))
    assert len(prog.cinematics) == 1


def test_errors_inside_an_imported_file_surface(game, tmp_path):
    (tmp_path / "bad.game").write_text('cinematic "b" { define x( { } }')
    with pytest.raises(game.GameError):
        game.compile('import "bad" expose ALL\ncinematic "m" { layer a { fn: circle(0.2) | glow(2.0) } }', base_dir=str(tmp_path))


@pytest.mark.skipif(not reference_available(), reason="needs the reference's stdlib/*.game files")
def test_reference_stdlib_files_import(game, tmp_path):
    """Every library of the reference resolves through `expose ALL`; the animate mode keeps the audio-driven
    expressions the defines pass to their stages."""
    std = os.path.join(REFERENCE, "stdlib")
    names = sorted(f[:-5] for f in os.listdir(std) if f.endswith(".game"))
    assert len(names) >= 10
    for n in names:
        prog = game.compile(f'import "{n}" expose ALL\ncinematic "m" {{ layer a {{ fn: circle(0.2) | glow(2.0) }} }}',
                            base_dir=str(tmp_path), lib_dirs=[std])
        assert len(prog.cinematics) == 1
    cin = game.compile('import "audio" expose bass_pulse\ncinematic "m" { layer a { fn: bass_pulse(0.2) | tint(1.0, 0.4, 0.1) } }',
                       base_dir=str(tmp_path), lib_dirs=[std]).cinematics[0]
    assert cin.derived_count == 2 and cin.ir["derived"][0] == "(0.200000 + (audio.bass * 0.100000))"
