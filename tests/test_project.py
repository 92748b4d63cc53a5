"""Projection blocks: `project flat|dome|cube|led (params) { source: name }` (game-compiler/src/codegen/project.rs,
applied at lib.rs:187-192). The reference swaps the WebGPU vertex stage of the named cinematic; here the block is a
per-pixel warp of v_uv, opt-in per context. Front-end facts restate the reference's parser (parser.rs:1140-1191) and
its four vertex variants; pixels are checked against the oracle and, for the mappings themselves, against numpy."""
import numpy as np
import pytest

BODY = 'layer a { fn: ring(0.5, 0.05) | glow(2.0) | tint(0.2, 0.6, 1.0) } layer b { fn: fbm(4.0, 3, 0.5, 2.0) | glow(1.0) | tint(0.9, 0.5, 0.1) }'


def src(mode, params="", body=BODY, memory=False):
    layers = body.replace("layer a {", "layer a memory: 0.9 {") if memory else body
    return 'cinematic "sky" { %s }\nproject %s %s { source: sky }' % (layers, mode, params)


def test_project_blocks_reach_the_ir(game):
    assert game.compile('cinematic "sky" { %s }' % BODY).cinematics[0].project_mode is None
    for mode in ("flat", "dome", "cube", "led"):
        cin = game.compile(src(mode)).cinematics[0]
        assert cin.project_mode == mode and cin.ir["project"]["mode"] == mode
    cin = game.compile(src("dome", "(segments: 8, fisheye: 160)")).cinematics[0]
    assert cin.ir["project"] == {"mode": "dome", "fov_deg": 160, "count": 60}
    assert game.compile(src("led", "(count: 144)")).cinematics[0].ir["project"]["count"] == 144
    # a block that names another cinematic leaves this one alone; an unknown mode is the reference's parse error
    other = game.compile('cinematic "sky" { %s }\nproject dome { source: ground }' % BODY).cinematics[0]
    assert other.project_mode is None
    with pytest.raises(game.GameError, match="unknown projection mode `sphere`, expected flat/dome/cube/led") as e:
        game.compile(src("sphere"))
    assert e.value.code == 3


def test_oracle_projection_mappings(game, oracle):
    """dome: project.rs:62-69 per pixel; led: every pixel of LED k samples (k / count, 0.5); flat / cube: unchanged."""
    from np_restatement import F, fma, pack
    w, h = 40, 24
    plain = game.compile('cinematic "sky" { %s }' % BODY).cinematics[0]
    row = plain.uniform_rows([0.3], w, h)[0]
    base = oracle.render(plain.blob, w, h, row)
    for mode in ("flat", "cube"):
        cin = game.compile(src(mode)).cinematics[0]
        assert np.array_equal(oracle.render(cin.blob, w, h, row, project=True), base)
    # without the flag a block changes nothing
    dome = game.compile(src("dome", "(fov: 120)")).cinematics[0]
    assert np.array_equal(oracle.render(dome.blob, w, h, row), base)
    assert not np.array_equal(oracle.render(dome.blob, w, h, row, project=True), base)
    # led: the frame is `count` vertical strips, each one colour: the colour of the plain frame's formula at (k / count, 0.5)
    g = 'layer a { fn: gradient() | tint(0.5 + 0.5 * uv.x, 0.5 + 0.5 * uv.y, 0.25) }'
    led = game.compile(src("led", "(count: 8)", body=g)).cinematics[0]
    img = oracle.render(led.blob, w, h, row, project=True, animate=True)
    assert (img == img[0:1]).all()                       # constant down every column
    vx = (np.arange(w, dtype=np.float32) + F(0.5)) * (F(1.0) / F(w))
    t = np.floor(vx * F(8.0)) / F(8.0)
    uvx = t * F(2.0) - F(1.0)
    uvy = np.full_like(uvx, F(0.5) * F(2.0) - F(1.0))
    lum = uvy * F(0.5) + F(0.5)
    want = pack(lum * (F(0.5) + F(0.5) * uvx), lum * (F(0.5) + F(0.5) * uvy), lum * F(0.25), 1.0)
    assert np.array_equal(img[0], want)


def test_project_kernels_compile_for_gfx950(game):
    for mode, params in (("dome", "(fisheye: 170)"), ("led", "(count: 32)"), ("cube", "")):
        cin = game.compile(src(mode, params)).cinematics[0]
        assert cin.compile_only(project=True) > 0 and cin.compile_only(semantics="wgsl", project=True) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("semantics", ["glsl", "wgsl"])
@pytest.mark.parametrize("mode,params", [("flat", ""), ("dome", ""), ("dome", "(fisheye: 100)"), ("cube", ""), ("led", "(count: 24)")])
def test_gpu_projection_matches_oracle(game, oracle, semantics, mode, params):
    cin = game.compile(src(mode, params)).cinematics[0]
    w, h = 160, 90
    rows = cin.uniform_rows([0.0, 0.7], w, h)
    r = cin.renderer(w, h, semantics=semantics, project=True)
    got = r.render(rows)
    r.close()
    want = oracle.render_sequence(cin.blob, w, h, rows, False, semantics=semantics, project=True)
    assert np.array_equal(got, want)
    r = cin.renderer(w, h, semantics=semantics)          # the block is inert without the flag
    assert np.array_equal(r.render(rows), oracle.render_sequence(cin.blob, w, h, rows, False, semantics=semantics))
    r.close()


@pytest.mark.gpu
def test_gpu_projection_with_memory_and_bands(game, oracle):
    cin = game.compile(src("dome", "(fisheye: 140)", memory=True)).cinematics[0]
    w, h = 96, 64
    rows = cin.uniform_rows([k / 30 for k in range(5)], w, h)
    r = cin.renderer(w, h, project=True)
    got = r.render(rows)
    r.reset_memory()
    band = r.render_band(rows, 16, 24)
    r.close()
    want = oracle.render_sequence(cin.blob, w, h, rows, True, project=True)
    assert np.array_equal(got, want)
    assert np.array_equal(band, want[:, 16:40])
