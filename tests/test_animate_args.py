"""Expression-valued stage arguments. The reference's stage emitters replace any argument that is not a number,
an identifier or a dotted name by the parameter's default (game-compiler/src/codegen/stages.rs:8-22), although
its expression compiler (codegen/expr.rs:50-144) handles them; that stays the default behaviour here. With the
animate flag the host evaluates those expressions once per frame and the kernel reads the values as uniforms.
The checker for the values is a Python statement of each expression (float64 = JS numbers, one rounding to f32)."""
import math

import numpy as np
import pytest

SRC = '''cinematic "spin" {
  layer a {
    fn: rotate(time * 1.5) | star(5.0, 0.3 + sin(time * 2.0) * 0.05, 0.12) | glow(1.0 + audio.bass * 3.0) | tint(0.9, 0.4 + 0.3 * cos(time), mix(0.1, 0.6, audio.beat))
  }
  layer b {
    fn: translate(mouse.x - 0.5, -(mouse.y - 0.5)) | circle(size * 0.5 + 0.05) | glow(2.0 ^ 1.5) | tint(0.2, 0.5, 1.0)
    size: 0.3
  }
}'''

F = np.float32


def f32(x):
    return float(F(x))


def wrapped_time(t):
    q = F(t) / F(120.0)
    return float((q - np.floor(q)) * F(120.0))


def expected(row):
    """The eight expressions of SRC in source order (`2.0 ^ 1.5` is folded to a number by the front end); literals pass through f32 like every literal of the language."""
    t = wrapped_time(row[0])
    bass, beat, mx, my, size = (float(row[1]), float(row[5]), float(row[8]), float(row[9]), float(row[10]))
    return np.array([
        t * f32(1.5),
        f32(0.3) + math.sin(t * f32(2.0)) * f32(0.05),
        f32(1.0) + bass * f32(3.0),
        f32(0.4) + f32(0.3) * math.cos(t),
        f32(0.1) * (1.0 - beat) + f32(0.6) * beat,
        mx - f32(0.5),
        -(my - f32(0.5)),
        size * f32(0.5) + f32(0.05),
    ], dtype=np.float64).astype(np.float32)


def test_derived_programs_are_kept_in_source_order(game):
    cin = game.compile(SRC).cinematics[0]
    assert cin.derived_count == 8
    ir = cin.ir
    assert ir["derived"][0] == "(time * 1.500000)" and ir["derived"][2] == "(1.000000 + (audio.bass * 3.000000))"
    stages = [st for layer in ir["layers"] for st in layer["stages"]]
    idx = [a["derived"] for st in stages for a in st["args"]]
    assert [i for i in idx if i >= 0] == list(range(8))
    # the default every such argument keeps (what the reference's emitters substitute) is still in the IR
    rot = stages[0]["args"][0]
    assert rot["kind"] == "lit" and rot["text"] == "0.000000"


def test_derived_values_follow_the_expressions(game):
    cin = game.compile(SRC).cinematics[0]
    times = [0.0, 1.0 / 60.0, 0.5, 3.25, 119.99, 130.0, 1000.5]
    rows = cin.uniform_rows(times, 64, 36, audio={"bass": 0.25, "beat": 0.75}, user={"size": 0.4})
    rows[:, 8] = np.linspace(0.0, 1.0, len(times), dtype=np.float32)    # mouse.x
    rows[:, 9] = np.linspace(1.0, 0.0, len(times), dtype=np.float32)    # mouse.y
    got = cin.eval_derived(rows)
    want = np.stack([expected(r) for r in rows])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("src,n", [
    ('cinematic "a" { layer x { fn: circle(nope * 2.0) | glow(1.0) } }', 0),          # unknown name: not a frame constant
    ('cinematic "a" { layer x { fn: circle(0.2) | palette(time + 1.0) } }', 0),   # palette names are text
    ('cinematic "a" { layer x { fn: fbm(2.0, 3.0 + time, 0.5, 2.0) | glow(1.0) } }', 0),      # octave count stays compile-time
    ('cinematic "a" { layer x { fn: fbm(2.0 * time, 4.0, 0.5, 2.0) | glow(1.0) } }', 1),
    ('cinematic "a" { layer x { fn: circle(radius: 0.1 + mouse.x) | glow(1.0) } }', 1),         # named argument
    ('cinematic "a" { layer x { fn: circle(0.2) | glow(time) } }', 0),                         # plain identifiers are uniforms already
    ('cinematic "a" { layer x { fn: circle(nofunc(time)) | glow(1.0) } }', 0),
])
def test_which_arguments_become_programs(game, src, n):
    assert game.compile(src).cinematics[0].derived_count == n


def test_default_behaviour_is_the_references(game, oracle):
    """Without the flag an expression renders as the parameter default: same frame as the source with defaults."""
    a = game.compile('cinematic "a" { layer x { fn: rotate(time * 1.5) | star(5.0, 0.3 + audio.bass * 0.1, 0.12) | glow(2.0) } }').cinematics[0]
    b = game.compile('cinematic "a" { layer x { fn: rotate(0.0) | star(5.0, 0.3, 0.12) | glow(2.0) } }').cinematics[0]
    row = a.uniform_rows([1.25], 48, 27, audio={"bass": 1.0})[0]
    assert np.array_equal(oracle.render(a.blob, 48, 27, row), oracle.render(b.blob, 48, 27, row))
    # and with the values passed in, the oracle renders what the same values do as plain uniforms
    d = a.eval_derived(row[None])[0]
    assert np.array_equal(d, np.array([1.875, float(np.float32(0.3)) + float(np.float32(0.1))], dtype=np.float32))
    c = game.compile('cinematic "a" { layer x { fn: rotate(ang) | star(5.0, rad, 0.12) | glow(2.0) } }').cinematics[0]
    crow = c.uniform_rows([1.25], 48, 27, audio={"bass": 1.0}, user={"ang": float(d[0]), "rad": float(d[1])})[0]
    animated = oracle.render(a.blob, 48, 27, row, derived=d)
    assert np.array_equal(animated, oracle.render(c.blob, 48, 27, crow))
    assert not np.array_equal(animated, oracle.render(a.blob, 48, 27, row))


def test_animate_kernels_compile_for_gfx950(game):
    cin = game.compile(SRC).cinematics[0]
    assert cin.compile_only(animate=True) > 0
    assert cin.compile_only(semantics="wgsl", animate=True) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("semantics", ["glsl", "wgsl"])
def test_gpu_animated_arguments_match_oracle(game, oracle, semantics):
    cin = game.compile(SRC).cinematics[0]
    w, h = 160, 90
    times = [0.0, 0.4, 2.5, 130.0]
    rows = cin.uniform_rows(times, w, h, audio={"bass": 0.25, "beat": 0.75}, user={"size": 0.4})
    rows[:, 8] = 0.7
    rows[:, 9] = 0.35
    derived = np.stack([expected(r) for r in rows])
    r = cin.renderer(w, h, semantics=semantics, animate=True)
    got = r.render(rows)
    r.close()
    want = oracle.render_sequence(cin.blob, w, h, rows, False, semantics=semantics, derived=derived)
    assert np.array_equal(got, want)
    # the flag changes the picture: the default-argument frames differ
    still = oracle.render_sequence(cin.blob, w, h, rows, False, semantics=semantics)
    assert not np.array_equal(want, still)
    # and a context without the flag renders exactly those default-argument frames
    r = cin.renderer(w, h, semantics=semantics)
    got = r.render(rows)
    r.close()
    assert np.array_equal(got, still)


@pytest.mark.gpu
def test_gpu_animated_memory_sequence(game, oracle):
    src = '''cinematic "trail" {
      layer orb memory: 0.9 { fn: translate(sin(time * 3.0) * 0.5, cos(time * 2.0) * 0.3) | circle(0.08 + audio.bass * 0.1) | glow(2.5) | tint(1.0, 0.6, 0.2) }
    }'''
    cin = game.compile(src).cinematics[0]
    assert cin.uses_memory and cin.derived_count == 3
    w, h = 128, 72
    rows = cin.uniform_rows([k / 30 for k in range(6)], w, h, audio={"bass": 0.5})
    derived = cin.eval_derived(rows)
    r = cin.renderer(w, h, animate=True)
    got = r.render(rows)
    r.close()
    want = oracle.render_sequence(cin.blob, w, h, rows, True, derived=derived)
    assert np.array_equal(got, want)


PIX = '''cinematic "pix" {
  layer a { fn: rotate(p.x * 0.5) | circle(0.2 + 0.05 * sin(p.x * 20.0 + time)) | glow(1.0 + abs(uv.y)) | tint(0.5 + 0.5 * uv.x, 0.6, max(p.y, 0.2)) }
  layer b memory: 0.8 { fn: translate(0.3, -0.2) | ring(0.25, 0.02 + 0.01 * cos(p.y * 30.0) * audio.bass) | glow(knob * (1.0 - p.x * p.x)) | tint(1.0, 0.4, 0.1)
    knob: 1.2 }
}'''


def test_pixel_expressions_are_kept_apart_from_frame_expressions(game):
    cin = game.compile(PIX).cinematics[0]
    ir = cin.ir
    assert cin.derived_count == 0 and cin.pixel_expr_count == 7
    assert ir["pixel_exprs"][0] == "(p.x * 0.500000)" and ir["pixel_exprs"][-1] == "(knob * ((1.000000 - (p.x * p.x))))"
    args = [a for l in ir["layers"] for st in l["stages"] for a in st["args"]]
    assert [a["pixel"] for a in args if a["pixel"] >= 0] == list(range(7))
    # things that are not scalars of the pixel stay defaults: a vector-valued argument, unknown fields
    for src in ('circle(p * 2.0)', 'circle(p.z)', 'circle(normalize(p))', 'circle([0.1, 0.2])', 'circle(p > uv)'):
        c = game.compile('cinematic "a" { layer x { fn: %s | glow(1.0) } }' % src).cinematics[0]
        assert c.pixel_expr_count == 0 and c.derived_count == 0


def test_oracle_pixel_expressions_change_the_frame_only_when_live(game, oracle):
    cin = game.compile(PIX).cinematics[0]
    plain = game.compile(PIX.replace("rotate(p.x * 0.5)", "rotate(0.0)")).cinematics[0]
    row = cin.uniform_rows([0.4], 64, 36, audio={"bass": 0.7}, user={"knob": 1.5})[0]
    still = oracle.render(cin.blob, 64, 36, row)
    live = oracle.render(cin.blob, 64, 36, row, animate=True)
    assert not np.array_equal(still, live)
    # without the flag every expression renders as its default, whatever it reads
    prow = plain.uniform_rows([0.4], 64, 36, audio={"bass": 0.7}, user={"knob": 1.5})[0]
    assert np.array_equal(still, oracle.render(plain.blob, 64, 36, prow))
    # a pixel expression that happens to be constant equals the literal
    a = game.compile('cinematic "a" { layer x { fn: circle(0.25 + p.x * 0.0) | glow(2.0) } }').cinematics[0]
    b = game.compile('cinematic "a" { layer x { fn: circle(0.25) | glow(2.0) } }').cinematics[0]
    r = a.uniform_rows([0.0], 48, 27)[0]
    assert np.array_equal(oracle.render(a.blob, 48, 27, r, animate=True), oracle.render(b.blob, 48, 27, r))


def test_pixel_expression_kernels_compile_for_gfx950(game):
    cin = game.compile(PIX).cinematics[0]
    assert cin.compile_only(animate=True) > 0 and cin.compile_only(semantics="wgsl", animate=True, lens_post=True) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("semantics", ["glsl", "wgsl"])
def test_gpu_pixel_expressions_match_oracle(game, oracle, semantics):
    cin = game.compile(PIX).cinematics[0]
    w, h = 160, 90
    rows = cin.uniform_rows([k / 20 for k in range(4)], w, h, audio={"bass": 0.7}, user={"knob": 1.5})
    r = cin.renderer(w, h, semantics=semantics, animate=True)
    got = r.render(rows)
    r.close()
    want = oracle.render_sequence(cin.blob, w, h, rows, True, semantics=semantics, animate=True)
    assert np.array_equal(got, want)
    r = cin.renderer(w, h, semantics=semantics)
    got = r.render(rows)
    r.close()
    assert np.array_equal(got, oracle.render_sequence(cin.blob, w, h, rows, True, semantics=semantics))


# ── vector-valued subexpressions (expr.rs:50-144 words them as WGSL vec2f arithmetic) ──────────────────────────────
VEC = '''cinematic "vec" {
  layer a {
    fn: rotate(atan2(p.y, p.x) * 0.1) | circle(0.05 + 0.1 * length(p * 2.0 - [0.5, 0.25])) | glow(1.0 + dot(normalize(p), [0.6, 0.8]) * 0.5)
        | tint(length(uv) * 0.7, distance(uv, [0.25, -0.5]) * 0.5, 0.5 + 0.5 * sin(dot(uv, uv * 3.0) + time))
  }
  layer b {
    fn: translate(mod(p.x * 3.0, 0.4) - 0.2, round(p.y * 4.0) * 0.1) | ring(0.2 + 0.02 * length(-p), 0.03) | glow(1.5)
        | tint(min(abs(p), [0.9, 0.8]) * 1.2 - p * 0.0 + [0.1, 0.2] - [0.1, 0.2] > 0.5, 0.4, 0.9)
  }
}'''


def test_vector_subexpressions_lower_to_scalar_programs(game):
    cin = game.compile(VEC).cinematics[0]
    ir = cin.ir
    # the vec2 > scalar comparison in the last tint has no scalar meaning: that one argument keeps its default
    assert cin.pixel_expr_count == 9 and cin.derived_count == 0
    assert ir["pixel_exprs"][1] == "(0.050000 + (0.100000 * length(((p * 2.000000) - [0.500000, 0.250000]))))"
    # dot(p, [a, b]) is p.x * a + p.y * b, operator for operator
    a = game.compile('cinematic "a" { layer x { fn: circle(dot(p, [0.5, 0.25])) | glow(1.0) } }').cinematics[0]
    b = game.compile('cinematic "a" { layer x { fn: circle(p.x * 0.5 + p.y * 0.25) | glow(1.0) } }').cinematics[0]
    assert a.blob[a.blob.index(b"GIR1") + 8:] == b.blob[b.blob.index(b"GIR1") + 8:]


def test_oracle_vector_expressions_follow_their_definitions(game, oracle):
    """length(v) = sqrt(fma(x, x, y * y)), distance(a, b) = length(a - b), dot = x1 * x2 + y1 * y2, one rounding each."""
    from np_restatement import F, fma, pack, pixel_grid
    src = 'cinematic "a" { layer x { fn: gradient() | tint(length(uv) * 0.5, distance(uv, [0.25, -0.5]) * 0.5, dot(uv, uv * 0.5)) } }'
    cin = game.compile(src).cinematics[0]
    w, h = 53, 31
    got = oracle.render(cin.blob, w, h, cin.uniform_rows([0.0], w, h)[0], animate=True)
    uvx, uvy = pixel_grid(w, h)
    t = uvy * F(0.5) + F(0.5)
    r = np.sqrt(fma(uvx, uvx, uvy * uvy)) * F(0.5)
    dx, dy = uvx - F(0.25), uvy - F(-0.5)
    g = np.sqrt(fma(dx, dx, dy * dy)) * F(0.5)
    b = uvx * (uvx * F(0.5)) + uvy * (uvy * F(0.5))
    assert np.array_equal(got, pack(t * r, t * g, t * b, 1.0))


def test_vector_expression_kernels_compile_for_gfx950(game):
    cin = game.compile(VEC).cinematics[0]
    assert cin.compile_only(animate=True) > 0 and cin.compile_only(semantics="wgsl", animate=True) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("semantics", ["glsl", "wgsl"])
def test_gpu_vector_expressions_match_oracle(game, oracle, semantics):
    cin = game.compile(VEC).cinematics[0]
    w, h = 160, 90
    rows = cin.uniform_rows([k / 7 for k in range(3)], w, h)
    r = cin.renderer(w, h, semantics=semantics, animate=True)
    got = r.render(rows)
    r.close()
    want = oracle.render_sequence(cin.blob, w, h, rows, False, semantics=semantics, animate=True)
    assert np.array_equal(got, want)


# ── ternaries and named colours (expr.rs:132-142 `select(f, t, c)`, expr.rs:13-36 `vec3f(r, g, b)`) ─────────────────
SEL = '''cinematic "sel" {
  layer a {
    fn: rotate(time > 1.0 ? 0.5 : -0.5) | circle(audio.bass > 0.5 ? 0.3 : 0.1 + knob * 0.1) | glow(p.x > 0.0 ? 2.0 : (p.y < 0.0 ? 1.0 : 0.5))
        | tint(uv.x > 0.0 ? gold : [0.2, 0.3, 0.9])
    knob: 0.5
  }
  layer b { fn: circle(0.2) | shade(mix(ember, frost, audio.mid)) }
  layer c { fn: ring(0.3, 0.02) | glow(1.5) | tint(gold) }
  layer d { fn: ring(0.4, 0.02) | glow(1.5) | tint(violet * 0.5 + [0.1, 0.0, 0.2]) }
}'''

COLOURS = {"black": (0.0, 0.0, 0.0), "white": (1.0, 1.0, 1.0), "red": (1.0, 0.0, 0.0), "green": (0.0, 1.0, 0.0), "blue": (0.0, 0.0, 1.0),
           "gold": (0.831, 0.686, 0.216), "midnight": (0.039, 0.039, 0.118), "obsidian": (0.071, 0.059, 0.082),
           "ember": (0.898, 0.318, 0.129), "cyan": (0.0, 1.0, 1.0), "ivory": (1.0, 1.0, 0.941), "frost": (0.686, 0.878, 0.953),
           "orange": (1.0, 0.647, 0.0), "deep_blue": (0.0, 0.098, 0.392), "ash": (0.467, 0.467, 0.467),
           "charcoal": (0.212, 0.212, 0.212), "plasma": (0.580, 0.0, 0.827), "violet": (0.541, 0.169, 0.886), "magenta": (1.0, 0.0, 1.0)}


def test_ternary_and_colour_arguments_lower_to_programs(game):
    cin = game.compile(SEL).cinematics[0]
    ir = cin.ir
    # frame-level: rotate, circle, shade r/g/b, tint(gold) r/g/b, tint(violet ...) r/g/b; pixel-level: glow, tint(uv.x > 0 ? ...) r/g/b
    assert cin.derived_count == 2 + 3 + 3 + 3 and cin.pixel_expr_count == 1 + 3
    assert ir["derived"][0] == "((time > 1.000000) ? 0.500000 : -0.500000)"
    assert ir["derived"][5:8] == ["gold.r", "gold.g", "gold.b"]
    assert ir["pixel_exprs"][1] == "((uv.x > 0.000000) ? gold : [0.200000, 0.300000, 0.900000]).r"
    # what the reference's emitters see stays what the IR resolves: `gold` is a uniform that defaults to 0
    # (codegen/mod.rs:89-101), the expressions keep the default of `r`, and g / b their own defaults
    stages = {(li, st["name"]): st for li, layer in enumerate(ir["layers"]) for st in layer["stages"]}
    gold = stages[(2, "tint")]["args"]
    assert gold[0]["kind"] == "user" and [a["text"] for a in gold] == ["gold", "1.000000", "1.000000"]
    assert [a["derived"] for a in gold] == [5, 6, 7]
    assert [u["name"] for u in ir["uniforms"]] == ["knob", "gold"] and cin.ir["uniforms"][1]["default"] == 0.0
    # a name the cinematic declares as a parameter is that parameter, not a colour; a colour is no scalar
    c = game.compile('cinematic "a" { layer x { fn: circle(0.2) | glow(1.0) | tint(gold)\n gold: 0.5 } }').cinematics[0]
    assert c.derived_count == 0
    for src in ('circle(gold)', 'glow(ember * 2.0)', 'tint(gold, 0.5)', 'tint([0.1, 0.2])', 'tint(uv ? gold : red)', 'tint(p.x > 0.0 ? gold : uv)'):
        c = game.compile('cinematic "a" { layer x { fn: circle(0.2) | glow(1.0) | %s } }' % src if not src.startswith(("circle", "glow"))
                         else 'cinematic "a" { layer x { fn: %s | glow(1.0) } }' % src if src.startswith("circle")
                         else 'cinematic "a" { layer x { fn: circle(0.2) | %s } }' % src).cinematics[0]
        assert c.derived_count == 0 and c.pixel_expr_count == 0, src


def test_every_named_colour_expands_to_its_components(game):
    for name, rgb in COLOURS.items():
        cin = game.compile('cinematic "a" { layer x { fn: circle(0.2) | glow(1.0) | tint(%s) } }' % name).cinematics[0]
        assert cin.derived_count == 3, name
        got = cin.eval_derived(cin.uniform_rows([0.0], 8, 8))[0]
        assert np.array_equal(got, np.array(rgb, dtype=np.float32)), name


def test_ternary_values_follow_the_expressions(game):
    cin = game.compile(SEL).cinematics[0]
    for t, bass, mid in ((0.5, 0.2, 0.0), (1.5, 0.7, 0.25), (1.0, 0.5, 1.0), (130.0, float("nan"), 0.5)):
        rows = cin.uniform_rows([t], 16, 9, audio={"bass": bass, "mid": mid}, user={"knob": 0.4})
        got = cin.eval_derived(rows)[0]
        tw = wrapped_time(rows[0][0])
        want = [f32(0.5) if tw > 1.0 else -f32(0.5),
                f32(0.3) if bass > 0.5 else f32(0.1) + f32(0.4) * f32(0.1)]
        m = float(rows[0][2])
        want += [f32(a) * (1.0 - m) + f32(b) * m for a, b in zip(COLOURS["ember"], COLOURS["frost"])]
        want += [f32(v) for v in COLOURS["gold"]]
        want += [f32(v) * f32(0.5) + f32(o) for v, o in zip(COLOURS["violet"], (0.1, 0.0, 0.2))]
        assert np.array_equal(got.view(np.uint32), np.array(want, dtype=np.float64).astype(np.float32).view(np.uint32)), (t, bass, mid)


def test_oracle_select_takes_the_branch_per_pixel(game, oracle):
    from np_restatement import F, pack, pixel_grid
    src = 'cinematic "a" { layer x { fn: gradient() | tint(uv.x > 0.0 ? gold : [0.2, 0.3, 0.9]) } }'
    cin = game.compile(src).cinematics[0]
    w, h = 37, 21
    got = oracle.render(cin.blob, w, h, cin.uniform_rows([0.0], w, h)[0], animate=True)
    uvx, uvy = pixel_grid(w, h)
    t = uvy * F(0.5) + F(0.5)
    right = uvx > 0
    r, g, b = (np.where(right, F(a), F(c)) for a, c in zip(COLOURS["gold"], (0.2, 0.3, 0.9)))
    assert np.array_equal(got, pack(t * r, t * g, t * b, 1.0))
    # without the flag `gold` is the reference's uniform (0), green and blue keep their defaults
    still = oracle.render(cin.blob, w, h, cin.uniform_rows([0.0], w, h)[0])
    assert np.array_equal(still, pack(t * F(1.0), t, t, 1.0))       # an expression argument keeps the default of r (1.0)


def test_select_and_colour_kernels_compile_for_gfx950(game):
    cin = game.compile(SEL).cinematics[0]
    assert cin.compile_only(animate=True) > 0 and cin.compile_only(semantics="wgsl", animate=True) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("semantics", ["glsl", "wgsl"])
def test_gpu_ternary_and_colour_arguments_match_oracle(game, oracle, semantics):
    cin = game.compile(SEL).cinematics[0]
    w, h = 160, 90
    rows = cin.uniform_rows([0.25, 1.5, 3.0], w, h, audio={"bass": [0.2, 0.7, 0.5], "mid": [0.0, 0.5, 1.0]}, user={"knob": 0.4})
    r = cin.renderer(w, h, semantics=semantics, animate=True)
    got = r.render(rows)
    assert "gd_select(" in r.kernel_source
    r.close()
    want = oracle.render_sequence(cin.blob, w, h, rows, False, semantics=semantics, animate=True, derived=cin.eval_derived(rows))
    assert np.array_equal(got, want)
    # without the flag: the reference's substitution (defaults; `gold` = the uniform at 0)
    r = cin.renderer(w, h, semantics=semantics)
    got = r.render(rows)
    r.close()
    still = oracle.render_sequence(cin.blob, w, h, rows, False, semantics=semantics)
    assert np.array_equal(got, still) and not np.array_equal(still, want)
