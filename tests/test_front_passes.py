"""Front-end parity: the reference's optimizer, define-expansion, validation, uniform-extraction,
argument-resolution, memory and gravity unit tests, restated as facts about the AST / flat IR.

The reference asserts on emitted shader text; the IR keeps, per stage argument, the exact text the
reference would have pasted ("0.200000", an identifier, ...), so the same facts are checked here.
Reference tests are cited as file:line of game-compiler/src.
"""
import os

import pytest

from conftest import C1_HELLO, C2_FBM, C2B_NOISE, C3_SHOWCASE, C4_TRAILS, C5_GRAVITY, REFERENCE, reference_available


def layer_src(pipeline, extra=""):
    return 'cinematic "t" { %s layer main { %s } }' % (extra, pipeline)


def opt(game, pipeline, extra=""):
    p = game.parse(layer_src(pipeline, extra), optimized=True)
    return p["cinematics"][0], p["optimize"][0]


def arg0(game, expr):
    c, st = opt(game, "circle(%s) | glow(1.0)" % expr)
    return c["layers"][0]["body"]["Pipeline"][0]["args"][0]["value"], st


def ir_of(game, src):
    return game.compile(src).cinematics[0].ir


def stage_args(ir, layer=0, stage=0):
    return [a["text"] for a in ir["layers"][layer]["stages"][stage]["args"]]


# ── optimize.rs:565-700 constant folding ────────────────────────────────────────────────
@pytest.mark.parametrize("expr,value,folds", [
    ("2.0 + 3.0", 5.0, 1),              # fold_literal_add
    ("3.0 * 4.0", 12.0, 1),             # fold_literal_mul
    ("10.0 / 4.0", 2.5, 1),             # fold_literal_div
    ("(2.0 + 3.0) * 4.0", 20.0, 2),     # fold_nested
    ("x * 0.0", 0.0, 1),                # fold_multiply_by_zero
    ("-7.0", -7.0, 1),                  # fold_neg_literal
    ("sin(0.0)", 0.0, 1),               # fold_sin_zero
    ("cos(0.0)", 1.0, 1),               # fold_cos_zero
    ("min(3.0, 7.0)", 3.0, 1),          # fold_min_max
    ("max(3.0, 7.0)", 7.0, 1),
    ("2.0 ^ 10.0", 1024.0, 1),
    ("sqrt(16.0)", 4.0, 1),
    ("floor(2.7) + ceil(2.2) + abs(-1.0)", 6.0, 6),
])
def test_constant_fold_values(game, expr, value, folds):
    v, st = arg0(game, expr)
    assert v == {"t": "Number", "v": value}
    assert st["constants_folded"] == folds


@pytest.mark.parametrize("expr,name", [("x * 1.0", "x"), ("y + 0.0", "y"), ("z / 1.0", "z"), ("0.0 + w", "w"), ("1.0 * q", "q"), ("r - 0.0", "r")])
def test_constant_fold_identities(game, expr, name):  # fold_multiply_by_one, fold_add_zero, fold_identity_div
    v, _ = arg0(game, expr)
    assert v == {"t": "Ident", "v": name}


def test_strength_reduction(game):  # optimize.rs:771-806
    v, st = arg0(game, "x / 2.0")
    assert st["constants_folded"] == 1
    assert v["t"] == "BinOp" and v["op"] == "Mul" and v["right"] == {"t": "Number", "v": 0.5}
    v, _ = arg0(game, "x / 4.0")
    assert v["op"] == "Mul" and v["right"]["v"] == 0.25
    v, _ = arg0(game, "x / 0.0")            # division by zero is left alone
    assert v["op"] == "Div"


def test_fold_quirk_non_numeric_args_are_dropped(game):
    # optimize.rs:216-221: non-numeric positional args are filtered out before matching, so
    # abs(x, 2.0) folds to 2.0.
    v, _ = arg0(game, "abs(x, 2.0)")
    assert v == {"t": "Number", "v": 2.0}


# ── optimize.rs:667-712 no-op stage elimination ─────────────────────────────────────────
def test_noop_stage_elimination(game):
    c, st = opt(game, "translate(0.0, 0.0) | circle(0.2) | glow(1.0)")     # noop_translate_zero
    assert st["noop_stages_removed"] == 1
    assert [s["name"] for s in c["layers"][0]["body"]["Pipeline"]] == ["circle", "glow"]
    assert opt(game, "scale(1.0) | circle(0.2) | glow(1.0)")[1]["noop_stages_removed"] == 1   # noop_scale_one
    assert opt(game, "rotate(0.0) | circle(0.2) | glow(1.0)")[1]["noop_stages_removed"] == 1  # noop_rotate_zero
    assert opt(game, "twist(0.0) | circle(0.2) | glow(1.0)")[1]["noop_stages_removed"] == 1
    assert opt(game, "translate(1.0, 0.0) | scale(2.0) | circle(0.2) | glow(1.0)")[1]["noop_stages_removed"] == 0  # not_noop_nonzero
    assert opt(game, "rotate(angle) | circle(0.2) | glow(1.0)")[1]["noop_stages_removed"] == 0   # not_noop_dynamic
    assert opt(game, "translate(1.0 - 1.0, 0.0) | circle(0.2) | glow(1.0)")[1]["noop_stages_removed"] == 1  # folded first
    assert opt(game, "translate() | circle(0.2) | glow(1.0)")[1]["noop_stages_removed"] == 0     # needs >= 1 positional


def test_dead_uniform_detection(game):  # optimize.rs:714
    src = 'cinematic "t" { layer uniforms { color: 1.0 unused_param: 0.5 } layer main { circle(color) | glow(1.0) } }'
    st = game.parse(src, optimized=True)["optimize"][0]
    assert st["dead_uniforms"] == ["unused_param"]


def test_dead_define_elimination(game):  # optimize.rs:830-912, :1164
    c, st = opt(game, "circle(0.3) | glow(1.5)", "define a(x) { circle(x) } define b(y) { ring(y) }")
    assert st["dead_defines_removed"] == 2 and c["defines"] == []
    c, st = opt(game, "circle(0.3) | glow(1.5)")
    assert st["dead_defines_removed"] == 0


def test_semantic_warnings(game):  # optimize.rs:916-1104, batch_compile.rs:262-374
    assert game.compile('cinematic "test" { layer { fn: circle(0.3) | glow(1.5) } }').warnings == []
    w = game.compile('cinematic "t" { define bad_shape() { nonexistent_fn(0.5) } layer { fn: circle(0.3) } }').warnings
    assert any("unknown function 'nonexistent_fn'" in x and "define 'bad_shape'" in x for x in w)
    w = game.compile('cinematic "t" { define wasteful(r, unused) { circle(r) } layer { fn: circle(0.3) } }').warnings
    assert w == ["define 'wasteful': parameter 'unused' is never used"]
    w = game.compile('cinematic "t" { define a(r) { circle(r) } define b(r) { a(r) | glow(1.0) } layer { fn: circle(0.3) } }').warnings
    assert w == []                                                       # cross-define references allowed
    w = game.compile('cinematic "t" { layer main { circle(0.1, 0.2, 0.3) } }').warnings
    assert w == ["layer 'main': 'circle' accepts 1 parameters, but 3 were given"]    # arity_detects_too_many_args
    assert game.compile('cinematic "t" { layer main { circle(radius: 0.1) } }').warnings == []   # named args not counted
    w = game.compile('cinematic "t" { lens { post: bloom(1.0, 2.0, 3.0) } layer main { circle(0.1) } }').warnings
    assert w == ["lens '<unnamed>': 'bloom' accepts 2 parameters, but 3 were given"]


# ── analysis.rs:342-483 define expansion ────────────────────────────────────────────────
def test_expand_defines(game):
    c, _ = opt(game, "my_shape(0.5) | glow(1.0)", "define my_shape(r) { circle(radius: r) }")
    st = c["layers"][0]["body"]["Pipeline"]
    assert st[0]["name"] == "circle" and st[0]["args"][0] == {"name": "radius", "value": {"t": "Number", "v": 0.5}}
    c, _ = opt(game, "styled(1.0, 0.0, 0.0)", "define styled(color_r, color_g, color_b) { circle() | glow() | tint(r: color_r, g: color_g, b: color_b) }")
    st = c["layers"][0]["body"]["Pipeline"]
    assert [s["name"] for s in st] == ["circle", "glow", "tint"] and st[2]["args"][0]["value"]["v"] == 1.0
    c, _ = opt(game, "my_circle(size: 0.3) | glow(1.0)", "define my_circle(size) { circle(radius: size) }")
    assert c["layers"][0]["body"]["Pipeline"][0]["args"][0]["value"]["v"] == 0.3    # expand_defines_named_args
    c, _ = opt(game, "outer(0.25)", "define inner(r) { circle(r) | glow(2.0) } define outer(q) { translate(0.1, 0.0) | inner(q) }")
    assert [s["name"] for s in c["layers"][0]["body"]["Pipeline"]] == ["translate", "circle", "glow"]   # nested


def test_recursive_define_surfaces_as_unknown_stage(game):
    # analysis.rs:59-65 returns an error that lib.rs:122 discards; validation then trips on the
    # still-unexpanded call.
    with pytest.raises(game.GameError) as ei:
        game.compile('cinematic "t" { define loop(r) { loop(r) } layer main { loop(0.1) } }')
    assert ei.value.code == 1 and "unknown stage function: 'loop'" in str(ei.value)


# ── builtins.rs:315-370 ─────────────────────────────────────────────────────────────────
ALL_BUILTINS = ["circle", "ring", "star", "glow", "tint", "bloom", "grain", "translate", "rotate", "scale", "shade",
                "emissive", "fbm", "simplex", "mask_arc", "gradient", "threshold", "twist", "mirror", "repeat", "blend",
                "vignette", "voronoi", "onion", "domain_warp", "box", "round", "polygon", "curl_noise", "tonemap",
                "scanlines", "chromatic", "saturate_color", "glitch", "concentric_waves", "displace", "spectrum", "palette"]


def test_builtin_registry(game):  # all_builtins_reachable, state_transitions_valid, palette_is_sdf_to_color_bridge
    b = {x["name"]: x for x in game.list_builtins()}
    assert sorted(b) == sorted(ALL_BUILTINS) and len(b) == 38
    assert (b["circle"]["input"], b["circle"]["output"]) == ("Position", "Sdf") and "radius" in b["circle"]["params"]
    assert (b["glow"]["input"], b["glow"]["output"]) == ("Sdf", "Color")
    assert (b["tint"]["input"], b["tint"]["output"]) == ("Color", "Color")
    assert (b["palette"]["input"], b["palette"]["output"], b["palette"]["params"]) == ("Sdf", "Color", ["name"])
    assert any(x["input"] == "Position" and x["output"] == "Sdf" for x in b.values())   # batch_compile.rs:285


@pytest.mark.parametrize("typo,want", [("cicle", "circle"), ("circl", "circle"), ("tnt", "tint"), ("glo", "glow"), ("blom", "bloom")])
def test_suggestions(game, typo, want):  # builtins.rs:358 suggest_finds_close_builtin, stages.rs:138
    with pytest.raises(game.GameError) as ei:
        game.compile(layer_src("%s(0.3)" % typo))
    assert ei.value.code == 1
    assert str(ei.value) == "validation error: unknown stage function: '%s'\n  help: did you mean '%s'?" % (typo, want)
    assert ei.value.rendered.startswith("error[E001]:")           # error.rs:506


def test_no_suggestion_for_gibberish(game):  # stages.rs:147, batch_compile.rs:233
    for name in ("xyzxyzxyz", "aaaaaaa", "zzzzzzz"):
        with pytest.raises(game.GameError) as ei:
            game.compile(layer_src("%s(0.3)" % name))
        assert name in str(ei.value) and "help" not in str(ei.value)


# ── stages.rs:87-197 pipeline validation and argument text ──────────────────────────────
def test_valid_pipelines(game):
    assert ir_of(game, layer_src("circle() | glow() | tint()"))["layers"][0]["end_state"] == "Color"
    assert ir_of(game, layer_src("translate() | rotate() | circle()"))["layers"][0]["end_state"] == "Sdf"   # transforms_stay_in_position
    assert ir_of(game, layer_src("fbm() | palette()"))["layers"][0]["end_state"] == "Color"
    assert ir_of(game, layer_src("voronoi() | palette() | tint()"))["layers"][0]["end_state"] == "Color"
    assert ir_of(game, layer_src("fbm(2.0) | displace(0.1) | circle(0.2) | glow(1.0)"))["layers"][0]["end_state"] == "Color"  # implicit reset


def test_type_mismatch_messages(game):  # stages.rs:155, :189; batch_compile.rs:221
    with pytest.raises(game.GameError) as ei:
        game.compile(layer_src("glow()"))
    msg = str(ei.value)
    assert ei.value.code == 2 and "type mismatch" in msg and "Position -> Sdf -> Color" in msg and "'(start)'" in msg
    with pytest.raises(game.GameError) as ei:
        game.compile(layer_src("circle() | glow() | tint() | glow()"))
    assert "'tint' produces Color output" in str(ei.value)
    with pytest.raises(game.GameError):
        game.compile(layer_src("palette()"))                       # palette_rejects_position_input
    with pytest.raises(game.GameError) as ei:
        game.compile('cinematic "t" { layer { fn: circle(0.3) | tint(1.0, 0.0, 0.0) } }')
    assert "type mismatch" in str(ei.value)


def test_get_arg_text(game):  # stages.rs:110-135
    assert stage_args(ir_of(game, layer_src("circle() | glow()"))) == ["0.200000"]                 # get_arg_with_default
    assert stage_args(ir_of(game, layer_src("circle(0.5) | glow()"))) == ["0.500000"]              # get_arg_positional
    assert stage_args(ir_of(game, layer_src("circle(radius: 0.75) | glow()"))) == ["0.750000"]     # get_arg_named
    assert stage_args(ir_of(game, layer_src("ring() | mask_arc() | glow()")), stage=1) == ["0.0"]  # no default -> "0.0"
    assert stage_args(ir_of(game, layer_src("ring() | mask_arc(4.0) | glow()")), stage=1) == ["4.000000"]   # glsl.rs:637


def test_get_arg_quirks(game):
    # stages.rs:8-22: anything that is not a number, identifier or dotted identifier falls back to
    # the builtin default; -0.3 is a number only because the optimizer folded the negation first.
    ir = ir_of(game, layer_src("translate(sin(time) * 0.3, -0.3) | circle(0.1 + r) | glow(2.0 + audio.energy) | tint(\"red\", [1.0], 2s)"))
    assert stage_args(ir, stage=0) == ["0.000000", "-0.300000"]
    assert stage_args(ir, stage=1) == ["0.200000"]
    assert stage_args(ir, stage=2) == ["1.500000"]
    assert stage_args(ir, stage=3) == ["1.000000", "1.000000", "1.000000"]
    # stages.rs:32-35: the positional index counts named arguments too
    ir = ir_of(game, layer_src("fbm(octaves: 6) | glow()"))
    assert stage_args(ir) == ["6.000000", "6.000000", "0.500000", "2.000000"]
    ir = ir_of(game, C2_FBM)
    assert stage_args(ir, stage=1) == ["3.000000", "6.000000", "0.500000", "2.000000"]
    # numbers are printed with six decimals before the shader compiler sees them
    ir = ir_of(game, layer_src("circle(0.12345678) | glow(1.0)"))
    a = ir["layers"][0]["stages"][0]["args"][0]
    assert a["text"] == "0.123457" and a["lit"] == pytest.approx(0.123457, abs=1e-9)
    assert stage_args(ir_of(game, layer_src("rotate(45deg) | circle() | glow()"))) == ["45.000000"]


# ── codegen/mod.rs:241-427 uniforms, validation through generate ────────────────────────
def test_extract_uniforms(game):
    c = game.compile(layer_src("circle(my_radius) | glow()")).cinematics[0]                 # extract_ident_uniforms
    assert c.uniforms == [("my_radius", 0.0)]
    a = c.ir["layers"][0]["stages"][0]["args"][0]
    assert (a["kind"], a["slot"], a["text"]) == ("user", 0, "my_radius")
    c = game.compile('cinematic "t" { layer config { intensity: 0.5 } }').cinematics[0]     # extract_param_uniforms
    assert c.uniforms == [("intensity", 0.5)]
    # order: Params-layer params, then opts (incl. fn: inline params), then stage identifiers; first wins
    src = 'cinematic "t" { layer a (k: 3.0) { fn: circle(r) | glow(g) | tint(k, z, r) r: 0.25 ~ audio.bass } layer cfg { z: 7.0 q: -1.5 } }'
    c = game.compile(src).cinematics[0]
    assert c.uniforms == [("k", 3.0), ("r", 0.25), ("g", 0.0), ("z", 0.0), ("q", -1.5)]
    assert c.frame_floats == 10 + 5
    c = game.compile(C3_SHOWCASE).cinematics[0]
    assert c.uniforms == [("bass", 0.0), ("mid", 0.0), ("treble", 0.0)]
    c = game.compile(C2B_NOISE).cinematics[0]       # tint(frost) -> user uniform frost = 0
    assert c.uniforms == [("frost", 0.0), ("ember", 0.0), ("violet", 0.0)]
    assert stage_args(c.ir, layer=0, stage=2) == ["frost", "1.000000", "1.000000"]
    c = game.compile(layer_src("fbm(2.0) | palette(fire)")).cinematics[0]
    assert c.uniforms == [("fire", 0.0)] and c.ir["layers"][0]["stages"][1]["aux"] == 0     # glsl.rs:704
    assert game.compile(layer_src("fbm(2.0) | palette(sunset)")).cinematics[0].ir["layers"][0]["stages"][1]["aux"] == 6
    assert game.compile(layer_src("fbm(2.0) | palette(nope)")).cinematics[0].ir["layers"][0]["stages"][1]["aux"] == 2


def test_validate_rejects_bad_pipeline_and_casts(game):  # codegen/mod.rs:272, :309-350; cast.rs tests
    with pytest.raises(game.GameError):
        game.compile(layer_src("glow()"))
    game.compile('cinematic "t" { layer main cast sdf { circle() } }')
    game.compile('cinematic "t" { layer main cast distance { circle() } }')
    game.compile('cinematic "t" { layer main cast rgba { circle() | glow() } }')
    game.compile('cinematic "t" { layer prep cast position { translate(0.1, 0.0) | rotate(1.0) } }')
    with pytest.raises(game.GameError) as ei:
        game.compile('cinematic "t" { layer main cast sdf { circle() | glow() } }')
    assert ei.value.code == 65 and "cast as 'sdf'" in str(ei.value) and "Color" in str(ei.value)
    with pytest.raises(game.GameError) as ei:
        game.compile('cinematic "t" { layer main cast color { circle() } }')
    assert "cast as 'color'" in str(ei.value) and "Sdf" in str(ei.value)
    with pytest.raises(game.GameError) as ei:
        game.compile('cinematic "t" { layer main cast banana { circle() } }')
    assert "unknown cast type 'banana'" in str(ei.value)
    with pytest.raises(game.GameError) as ei:
        game.compile('cinematic "t" { layer a cast sdf { circle() } layer b cast color { circle() } }')
    assert "layer 'b'" in str(ei.value)


def test_multi_layer_flag(game):  # glsl.rs:66 counts params layers too
    assert ir_of(game, C1_HELLO)["multi_layer"] is False
    assert ir_of(game, C2B_NOISE)["multi_layer"] is True
    assert ir_of(game, 'cinematic "t" { layer cfg { a: 1.0 } layer m { circle() | glow() } }')["multi_layer"] is True


def test_compile_public_api(game):  # batch_compile.rs:305-393
    p = game.compile('cinematic "t" { layer { fn: circle(1.0 + 1.0) | glow(1.5) } }')          # constant_folding_works
    assert stage_args(p.cinematics[0].ir) == ["2.000000"]
    game.compile('cinematic "t" { define unused_thing(x) { circle(x) | glow(1.5) } layer { fn: circle(0.3) | glow(1.5) } }')
    assert game.compile('cinematic "MyComponent" { layer { fn: star(5.0, 0.3, 0.15) | shade(1.0, 0.8, 0.2) } }').cinematics[0].name == "MyComponent"
    two = game.compile('cinematic "a" { layer { fn: circle(0.3) | glow(1.0) } } cinematic "b" { layer { fn: ring() | glow(1.0) } }')
    assert [c.name for c in two.cinematics] == ["a", "b"]


# ── memory.rs:154-210 ───────────────────────────────────────────────────────────────────
def test_memory_layers(game):
    ir = ir_of(game, 'cinematic "t" { layer a { circle() | glow() } layer b memory: 0.95 { ring() | glow() } }')
    assert ir["uses_memory"] is True                                                  # detects_memory_layers
    assert [l["memory"] for l in ir["layers"]] == [None, "0.950000"]                    # glsl_memory_mix_emits_correct_code
    assert ir_of(game, C1_HELLO)["uses_memory"] is False                              # no_memory_when_absent
    c = game.compile(C4_TRAILS).cinematics[0]
    assert c.ir["layers"][0]["memory"] == "0.970000" and c.uses_memory               # wgsl_memory_mix_emits_correct_code
    a = c.ir["layers"][0]["stages"][0]["args"][0]
    assert (a["kind"], a["slot"], a["text"]) == ("builtin", 1, "audio_bass")
    assert c.flags & 1                                                                # the reference pastes an undeclared name


# ── gravity.rs:214-367 ──────────────────────────────────────────────────────────────────
def gravity_ir(game, body):
    return ir_of(game, 'cinematic "t" { gravity { %s } layer s { circle() | glow() } }' % body)["gravity"]


def test_gravity_force_law_text(game):
    assert gravity_ir(game, "rule: 1.0 / dist, damping: 0.99")["law"] == "(1.000000 / dist)"                    # :260
    assert gravity_ir(game, "rule: 1.0 / (dist * dist), damping: 0.995")["law"] == "(1.000000 / ((dist * dist)))"
    assert gravity_ir(game, "rule: sin(dist * 3.14159)")["law"] == "sin((dist * 3.141590))"                      # :289
    assert gravity_ir(game, "rule: -(1.0 / (dist * dist))")["law"] == "(-((1.000000 / ((dist * dist)))))"
    assert "params.time" in gravity_ir(game, "rule: time * (1.0 / dist)")["law"]                                 # :337
    assert gravity_ir(game, "rule: dist ^ 2.0")["law"] == "pow(dist, 2.000000)"
    assert gravity_ir(game, "rule: dist > 0.5")["law"] == "select(0.0, 1.0, dist > 0.500000)"
    assert gravity_ir(game, "rule: dist > 0.5 ? 1.0 : 2.0")["law"] == "1.0"                                      # unsupported -> 1.0


def test_gravity_defaults_and_damping(game):
    g = ir_of(game, C5_GRAVITY)["gravity"]
    assert g["damping"] == "0.995" and g["bounds"] == 0 and g["law"] == "1.000000"    # damping_applied, reflect_bounds_emitted
    assert gravity_ir(game, "bounds: wrap")["bounds"] == 1 and gravity_ir(game, "bounds: none")["bounds"] == 2
    assert gravity_ir(game, "")["damping"] == "0.99"
    assert ir_of(game, C1_HELLO)["gravity"] is None                                   # generate_default has no compute
    with pytest.raises(game.GameError) as ei:
        gravity_ir(game, "bounds: bounce")
    assert "unknown bounds mode `bounce`" in str(ei.value)


# ── corpus (tests/batch_compile.rs:69-205) ──────────────────────────────────────────────
needs_reference = pytest.mark.skipif(not reference_available(), reason="reference tree not present on this machine")


def compile_dir(game, rel):
    d = os.path.join(REFERENCE, rel)
    ok, bad = [], []
    for f in sorted(os.listdir(d)):
        if f.endswith(".game"):
            try:
                game.compile(open(os.path.join(d, f), "rb").read())
                ok.append(f)
            except game.GameError as e:
                bad.append((f, str(e)))
    return ok, bad


@needs_reference
def test_corpus_compiles(game):
    ok, bad = compile_dir(game, "game-compiler/examples")
    assert len(ok) >= 8, bad                    # all_examples_compile
    ok, bad = compile_dir(game, "examples")
    assert len(ok) >= 14, bad                   # runyourempire_examples_compile
    tutorials_bad = [b for b in bad if b[0].startswith("tutorial-")]
    assert tutorials_bad == []                  # all_tutorial_files_compile
    ok, bad = compile_dir(game, "stdlib")
    assert len(ok) >= 6, bad                    # all_stdlib_compile
    ok, bad = compile_dir(game, "presets")
    assert len(ok) >= 10, bad                   # all_presets_compile


@needs_reference
def test_benchmark_fixtures_match_reference_files(game):
    """The inline benchmark cinematics are the reference's example files (ignoring comments)."""
    def ir_file(rel):
        return game.compile(open(os.path.join(REFERENCE, rel), "rb").read()).cinematics[0].ir
    assert ir_file("examples/001-hello.game") == ir_of(game, C1_HELLO)
    assert ir_file("examples/tutorial-05-noise.game") == ir_of(game, C2B_NOISE)
    assert ir_file("examples/007-showcase.game") == ir_of(game, C3_SHOWCASE)
    assert ir_file("examples/003-memory-trails.game") == ir_of(game, C4_TRAILS)
    assert ir_file("examples/010-gravity-particles.game") == ir_of(game, C5_GRAVITY)
