"""Front-end parity: the reference's lexer and parser unit tests, restated through the C ABI.

Each test names the reference test it restates (game-compiler/src/lexer.rs:328-427 and
game-compiler/src/parser_tests.rs). The reference builds token vectors by hand for the parser
tests; here the same constructs are written as source text, which also exercises the lexer.
"""
import pytest


def texts(game, src):
    return [t for _, _, t in game.tokens(src)]


def cin(game, body, name="t"):
    return game.parse('cinematic "%s" { %s }' % (name, body))["cinematics"][0]


# ── lexer.rs ────────────────────────────────────────────────────────────────────────────
def test_lex_keywords(game):  # lexer.rs:329
    assert texts(game, "cinematic") == ["cinematic"]
    assert texts(game, "import as arc resonate over memory cast") == ["import", "as", "arc", "resonate", "over", "memory", "cast"]
    # a keyword is only a keyword as a whole word
    assert texts(game, "layered castle") == ["layered", "castle"]


def test_lex_numbers(game):  # lexer.rs:339
    assert texts(game, "42") == ["42"]
    assert texts(game, "3.14") == ["3.14"]


def test_lex_units(game):  # lexer.rs:345
    assert texts(game, "2s") == ["2s"]
    assert texts(game, "0.5s") == ["0.5s"]
    assert texts(game, "500ms") == ["500ms"]
    assert texts(game, "4bars") == ["4bars"]
    assert texts(game, "180deg") == ["180deg"]
    # longest match: the unit is split off the front of a longer word
    assert texts(game, "2sec") == ["2s", "ec"]
    assert texts(game, "1.5bars") == ["1.5", "bars"]


def test_lex_string_and_escapes(game):  # lexer.rs:354, :406-427
    assert texts(game, '"hello"') == ['"hello"']
    assert texts(game, r'"hello\nworld"') == ['"hello\nworld"']
    assert texts(game, r'"say \"hello\""') == ['"say "hello""']
    assert texts(game, r'"path\\to\\file"') == ['"path\\to\\file"']
    assert texts(game, r'"col1\tcol2"') == ['"col1\tcol2"']


def test_lex_identifiers_and_operators(game):  # lexer.rs:359, :367
    assert texts(game, "foo bar_baz") == ["foo", "bar_baz"]
    assert texts(game, "| ~ >> <> !! .. ->") == ["|", "~", ">>", "<>", "!!", "..", "->"]
    assert texts(game, "true false") == ["true", "false"]  # identifiers, lexer.rs:244-245


def test_lex_full_layer_and_comments(game):  # lexer.rs:375, :384
    toks = texts(game, 'layer ring { circle(0.2) | glow(1.5) | tint(0.831, 0.686, 0.216) }')
    assert toks[:3] == ["layer", "ring", "{"] and "|" in toks
    assert texts(game, "foo // comment\nbar") == ["foo", "bar"]
    assert texts(game, "foo # comment\nbar") == ["foo", "bar"]
    assert "~" in texts(game, "radius: 0.3 ~ audio.bass * 0.1")  # lex_modulation


def test_lex_spans_start_at_zero(game):  # tests/batch_compile.rs:247 lex_produces_tokens
    toks = game.tokens('cinematic "test" { layer { fn: circle(0.3) } }')
    assert toks and toks[0][0] == 0


def test_lex_error_rendering(game):  # error.rs:389 render_lex_error_with_source
    with pytest.raises(game.GameError) as ei:
        game.tokens("layer @bad {}")
    assert ei.value.code == 64
    assert "unexpected character: '@'" in str(ei.value)
    assert "line 1:6" in ei.value.rendered and "^" in ei.value.rendered


# ── parser_tests.rs ─────────────────────────────────────────────────────────────────────
def test_parse_empty_program(game):  # :14
    p = game.parse("")
    assert p["imports"] == [] and p["cinematics"] == []


def test_parse_import(game):  # :26, :803, :1044
    p = game.parse('import "lib/base.game" as base')
    assert p["imports"] == [{"path": "lib/base.game", "alias": "base", "exposed": []}]
    p = game.parse('import "stdlib/noise.game" expose fbm, perlin')
    assert p["imports"][0]["alias"] == "" and p["imports"][0]["exposed"] == ["fbm", "perlin"]
    assert game.parse('import "x.game" expose ALL')["imports"][0]["exposed"] == ["ALL"]


def test_parse_basic_cinematic_with_layer(game):  # :45
    p = game.parse('cinematic "intro" { layer bg { } }')
    assert len(p["cinematics"]) == 1
    c = p["cinematics"][0]
    assert c["name"] == "intro" and len(c["layers"]) == 1 and c["layers"][0]["name"] == "bg"


def test_parse_layer_with_pipe_stages(game):  # :72
    c = cin(game, "layer fx { circle(0.5) | glow(2.0) }")
    st = c["layers"][0]["body"]["Pipeline"]
    assert [s["name"] for s in st] == ["circle", "glow"]


def test_parse_modulation(game):  # :110
    c = cin(game, "layer bg { opacity: 1.0 ~ audio.bass }")
    ps = c["layers"][0]["body"]["Params"]
    assert len(ps) == 1 and ps[0]["name"] == "opacity" and ps[0]["modulation"] is not None


def test_parse_arc_block(game):  # :147
    c = cin(game, "arc { bg.opacity: 0.0 -> 1.0 over 2s ease_in }")
    assert len(c["arcs"]) == 1
    e = c["arcs"][0]["entries"][0]
    assert e["target"] == "bg.opacity" and e["easing"] == "ease_in" and e["duration"] == {"unit": "s", "v": 2.0}


def test_parse_resonate_block(game):  # :180
    c = cin(game, "resonate { kick -> bg.scale * 0.3 }")
    e = c["resonates"][0]["entries"][0]
    assert (e["source"], e["target"], e["field"]) == ("kick", "bg", "scale")


def _expr(game, text):
    return cin(game, "layer l { v: %s }" % text)["layers"][0]["body"]["Params"][0]["value"]


def test_parse_expr_precedence(game):  # :211  1 + 2 * 3
    e = _expr(game, "1.0 + 2.0 * 3.0")
    assert e["t"] == "BinOp" and e["op"] == "Add" and e["left"] == {"t": "Number", "v": 1.0}
    assert e["right"]["t"] == "BinOp" and e["right"]["op"] == "Mul"


def test_parse_expr_power_right_assoc(game):  # :232  2 ^ 3 ^ 4
    e = _expr(game, "2.0 ^ 3.0 ^ 4.0")
    assert e["op"] == "Pow" and e["right"]["t"] == "BinOp" and e["right"]["op"] == "Pow"


def test_parse_layer_with_memory_and_cast(game):  # :256, :283
    assert cin(game, "layer trail memory: 0.95 { }")["layers"][0]["memory"] == 0.95
    assert cin(game, "layer d cast sdf { }")["layers"][0]["cast"] == "sdf"


def test_parse_multiple_layers(game):  # :309
    c = cin(game, "layer a { } layer b { }")
    assert [l["name"] for l in c["layers"]] == ["a", "b"]


def test_error_unexpected_token_at_top_level(game):  # :342
    with pytest.raises(game.GameError) as ei:
        game.parse("{")
    assert "expected" in str(ei.value)
    with pytest.raises(game.GameError) as ei:
        game.compile("{")
    assert ei.value.code == 3  # E003
    assert str(ei.value) == "parse error at 0:1: expected `import`, `cinematic`, `breed`, or `project` at top level, found `{`"


def test_parse_atoms(game):  # :360 array, :383 negative, :395 call, :418 dotted
    assert len(_expr(game, "[1.0, 2.0, 3.0]")["elems"]) == 3
    assert _expr(game, "-1.0")["t"] == "Neg"
    e = _expr(game, "sin(time)")
    assert e["t"] == "Call" and e["name"] == "sin" and len(e["args"]) == 1
    e = _expr(game, "layer2.opacity")
    assert e == {"t": "DottedIdent", "object": "layer2", "field": "opacity"}
    assert _expr(game, "45deg") == {"t": "Number", "v": 45.0}  # degrees carry no conversion (parser.rs:1385)


def test_parse_temporal_ops(game):  # :444 - :552
    def ops(text):
        return cin(game, "layer l { v: 1.0 %s }" % text)["layers"][0]["body"]["Params"][0]["temporal_ops"]
    assert ops(">> 200ms") == [{"t": "Delay", "duration": {"unit": "ms", "v": 200.0}}]
    assert ops("<> 50ms") == [{"t": "Smooth", "duration": {"unit": "ms", "v": 50.0}}]
    assert ops("!! 300ms")[0]["t"] == "Trigger"
    assert ops(".. [0.0, 1.0]")[0]["t"] == "Range"
    assert [o["t"] for o in ops("<> 50ms >> 200ms .. [0.0, 1.0]")] == ["Smooth", "Delay", "Range"]


def test_parse_listen_voice_score(game):  # :563, :589, :621
    c = cin(game, "listen { onset: attack(threshold: 0.7) }")
    s = c["listen"]["signals"]
    assert len(s) == 1 and s[0]["name"] == "onset" and s[0]["algorithm"] == "attack" and len(s[0]["params"]) == 1
    c = cin(game, "voice { tone: sine(freq: 440.0) filt: lowpass(cutoff: 2000.0) }")
    n = c["voice"]["nodes"]
    assert [(x["name"], x["kind"]) for x in n] == [("tone", "sine"), ("filt", "lowpass")]
    c = cin(game, "score tempo(120) { motif rise { scale: 0.5 -> 2.0 over 4bars } phrase build = rise arrange: build }")
    sc = c["score"]
    assert sc["tempo_bpm"] == 120.0 and sc["motifs"][0]["name"] == "rise" and len(sc["phrases"]) == 1 and sc["arrange"] == ["build"]


def test_parse_named_arg_in_stage(game):  # :657
    st = cin(game, "layer l { noise(rate: 3.0) }")["layers"][0]["body"]["Pipeline"][0]
    assert len(st["args"]) == 1 and st["args"][0]["name"] == "rate"


def test_parse_breed_block(game):  # :677
    p = game.parse('breed "child" from "fire" + "ice" { inherit layers: mix(0.6) mutate scale: 0.3 }')
    b = p["breeds"][0]
    assert b["name"] == "child" and b["parents"] == ["fire", "ice"]
    assert b["inherit_rules"] == [{"target": "layers", "strategy": "mix", "weight": 0.6}]
    assert b["mutations"] == [{"target": "scale", "range": 0.3}]


def test_parse_gravity_block(game):  # :721
    c = cin(game, "gravity { rule: 1.0 / (dist * dist), damping: 0.995, bounds: wrap }")
    g = c["gravity"]
    assert g["damping"] == 0.995 and g["bounds"] == "Wrap" and g["force_law"]["op"] == "Div"
    d = cin(game, "gravity { }")["gravity"]  # defaults, parser.rs:1086-1088
    assert d == {"force_law": {"t": "Number", "v": 1.0}, "damping": 0.99, "bounds": "Reflect"}


def test_parse_project_block(game):  # :752
    p = game.parse("project dome(segments: 8) { source: main }")
    pr = p["projects"][0]
    assert pr["mode"] == "Dome" and pr["source"] == "main" and [x["name"] for x in pr["params"]] == ["segments"]


def test_parse_hash_comments(game):  # :782
    p = game.parse('# top\ncinematic "x" {\n  # inside\n  layer bg { }\n}')
    assert p["cinematics"][0]["layers"][0]["name"] == "bg"


def test_parse_lens_react_define(game):  # :825, :854, :876
    c = cin(game, "lens main { fov: 60.0 post: bloom(0.5) | vignette(0.3) }")
    l = c["lenses"][0]
    assert l["name"] == "main" and [p["name"] for p in l["properties"]] == ["fov"] and [s["name"] for s in l["post"]] == ["bloom", "vignette"]
    assert len(cin(game, "react { mouse.click -> burst }")["react"]["reactions"]) == 1
    d = cin(game, "define glow_ring(r, g) { circle(r) | glow(g) }")["defines"][0]
    assert d["name"] == "glow_ring" and d["params"] == ["r", "g"] and [s["name"] for s in d["body"]] == ["circle", "glow"]


def test_parse_ternary_and_comparison(game):  # :907, :935
    e = _expr(game, "x > 0.5 ? 1.0 : 0.0")
    assert e["t"] == "Ternary" and e["condition"]["op"] == "Gt"
    assert e["if_true"] == {"t": "Number", "v": 1.0} and e["if_false"] == {"t": "Number", "v": 0.0}
    assert _expr(game, "a > b")["op"] == "Gt" and _expr(game, "a < b")["op"] == "Lt"


def test_parse_unnamed_layer(game):  # :962
    l = cin(game, "layer { circle(0.3) | glow(2.0) }")["layers"][0]
    assert l["name"].startswith("_layer_") and len(l["body"]["Pipeline"]) == 2
    assert l["name"] == "_layer_4"  # token index after `layer` (parser.rs:456)


def test_parse_fn_mixed_body(game):  # :996
    l = cin(game, "layer fill { fn: ring(0.3, 0.04) | glow(2.0) fill_angle: 0.0 ~ data.progress * 6.283 }")["layers"][0]
    assert l["name"] == "fill" and len(l["body"]["Pipeline"]) == 2
    assert [o["name"] for o in l["opts"]] == ["fill_angle"] and l["opts"][0]["modulation"] is not None


def test_parse_alias_and_api(game):  # batch_compile.rs:255 parse_alias_works
    p = game.parse('cinematic "test" { layer { fn: circle(0.3) } }')
    assert len(p["cinematics"]) == 1 and p["cinematics"][0]["name"] == "test"


def test_recovery_decides_the_reported_error(game):
    # parser.rs:73-90, :340-346, :433: expect(`)`) consumes the layer's `}` before failing, so
    # skip_brace_block_safe swallows the cinematic's own closing brace and the final expect(`}`)
    # error is what the caller sees instead of the first recorded one.
    with pytest.raises(game.GameError) as ei:
        game.parse('cinematic "x" { layer a { circle(0.3 } }')
    assert "expected `}`, found end of input" in str(ei.value)
    # here the failing token is only peeked, recovery closes the layer, and the first error stands
    with pytest.raises(game.GameError) as ei:
        game.parse('cinematic "x" { layer a { circle( } }')
    assert "unexpected token `}` in expression" in str(ei.value)
    # when recovery lands cleanly, the first recorded error is reported
    with pytest.raises(game.GameError) as ei:
        game.parse('cinematic "x" { 42 layer a { } }')
    assert "unexpected `42` inside cinematic block" in str(ei.value)
    with pytest.raises(game.GameError) as ei:
        game.parse('cinematic "x" { arc { a: 0.0 -> 1.0 over 2.0 } }')
    assert "expected duration (e.g. 2s, 500ms, 4bars), found bare number 2" in str(ei.value)
