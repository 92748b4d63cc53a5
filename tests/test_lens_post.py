"""`lens { post: a(..) | b(..) }`: parsed by the reference (game-compiler/src/parser.rs:1197-1234) and ignored by both of
its emitters. The IR keeps the chain as a trailing layer marked `post`; renderers skip it unless asked."""
import numpy as np
import pytest

SRC = '''cinematic "lensed" {
  layer a { fn: circle(0.3) | glow(2.0) | tint(1.0, 0.5, 0.2) }
  layer b { fn: translate(0.4, 0.0) | ring(0.2, 0.02) | glow(1.5) | tint(0.2, 0.6, 1.0) }
  lens {
    mode: flat
    post: bloom(0.4, 1.5) | vignette(0.5, 0.8) | circle(0.2) | nosuch(1.0) | grain(0.05)
  }
}'''
PLAIN = SRC[:SRC.index("  lens {")] + "}"
SINGLE = '''cinematic "one" {
  layer a memory: 0.8 { fn: translate(sin(time), 0.0) | circle(0.2) | glow(2.0) | tint(1.0, 0.9, 0.4) }
  lens { post: scanlines(200.0, 0.3) | tonemap(1.2) }
}'''


def test_ir_keeps_the_color_stages_of_the_chain(game):
    cin = game.compile(SRC).cinematics[0]
    layers = cin.ir["layers"]
    assert [l["post"] for l in layers] == [False, False, True]
    assert [s["name"] for s in layers[2]["stages"]] == ["bloom", "vignette", "grain"]      # not Color -> Color: dropped
    assert cin.ir["multi_layer"] is True
    plain = game.compile(PLAIN).cinematics[0]
    assert plain.ir["flags"] == cin.ir["flags"] and [l["post"] for l in plain.ir["layers"]] == [False, False]


def test_default_render_ignores_the_chain_like_the_reference(game, oracle):
    a = game.compile(SRC).cinematics[0]
    b = game.compile(PLAIN).cinematics[0]
    row = a.uniform_rows([0.3], 64, 36)[0]
    assert np.array_equal(oracle.render(a.blob, 64, 36, row), oracle.render(b.blob, 64, 36, row))
    with_post = oracle.render(a.blob, 64, 36, row, lens_post=True)
    assert not np.array_equal(with_post, oracle.render(a.blob, 64, 36, row))
    # a cinematic without a chain renders the same either way
    assert np.array_equal(oracle.render(b.blob, 64, 36, row, lens_post=True), oracle.render(b.blob, 64, 36, row))


def test_chain_equals_the_same_stages_appended_to_a_single_layer(game, oracle):
    """For one layer, post-processing the composite is the same arithmetic as appending the stages to the layer."""
    a = game.compile('cinematic "x" { layer a { fn: circle(0.3) | glow(2.0) | tint(1.0, 0.5, 0.2) }\n lens { post: vignette(0.5, 0.8) | grain(0.05) } }').cinematics[0]
    b = game.compile('cinematic "x" { layer a { fn: circle(0.3) | glow(2.0) | tint(1.0, 0.5, 0.2) | vignette(0.5, 0.8) | grain(0.05) } }').cinematics[0]
    row = a.uniform_rows([0.7], 80, 45)[0]
    for sem in ("glsl", "wgsl"):
        assert np.array_equal(oracle.render(a.blob, 80, 45, row, lens_post=True, semantics=sem),
                              oracle.render(b.blob, 80, 45, row, semantics=sem))


def test_kernels_with_the_chain_compile_for_gfx950(game):
    for src in (SRC, SINGLE):
        cin = game.compile(src).cinematics[0]
        assert cin.compile_only(lens_post=True) > 0
        assert cin.compile_only(semantics="wgsl", lens_post=True, animate=True) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("semantics", ["glsl", "wgsl"])
def test_gpu_lens_post_matches_oracle(game, oracle, semantics):
    cin = game.compile(SRC).cinematics[0]
    w, h = 160, 90
    rows = cin.uniform_rows([0.0, 0.25, 0.5], w, h)
    r = cin.renderer(w, h, semantics=semantics, lens_post=True)
    got = r.render(rows)
    r.close()
    assert np.array_equal(got, oracle.render_sequence(cin.blob, w, h, rows, False, semantics=semantics, lens_post=True))
    r = cin.renderer(w, h, semantics=semantics)
    got = r.render(rows)
    r.close()
    assert np.array_equal(got, oracle.render_sequence(cin.blob, w, h, rows, False, semantics=semantics))


@pytest.mark.gpu
def test_gpu_lens_post_feeds_the_memory_ring(game, oracle):
    cin = game.compile(SINGLE).cinematics[0]
    w, h = 128, 72
    times = [k / 30 for k in range(5)]
    rows = cin.uniform_rows(times, w, h)
    derived = cin.eval_derived(rows)
    r = cin.renderer(w, h, lens_post=True, animate=True)
    got = r.render(rows)
    r.close()
    want = oracle.render_sequence(cin.blob, w, h, rows, True, derived=derived, lens_post=True)
    assert np.array_equal(got, want)
