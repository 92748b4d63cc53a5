"""The only pixels in the reference tree that its own emitted shaders produced: three dev-server screenshots
(screenshots/01-load.png: presets/dashboard-gauge.game; screenshots/arc-01-load.png: "layered-scene" in the first,
un-animated moment of its arc timeline, glow_i = 0.5 and orb_glow = 1.0 as the panel shows; screenshots/arc-09-params.png:
the same scene 17 ms into the second moment - see further down; the other five files repeat these renders or show text). A browser rendered them
from the shader text of an OLDER generation of the compiler, kept in dist/dashboard-gauge.js and dist/layered-scene.js:
same vertex stage, uv, aspect and `p` as game-compiler/src/codegen/glsl.rs:53-61,:240 and the same ring / circle
distance functions (glsl.rs:96-100,:269-273), but glow = exp(-max(d, 0) * intensity * 8), additive layer compositing
and resolved colour names - so they are not vectors for today's `glow` or screen blend.

What they can pin is everything around the builtins: pixel centres, the uv / aspect / p convention at three aspect
ratios, length / abs / max / exp / mix of the expression evaluator, and the RGBA8 store - on a GPU that is not ours.
The two cinematics below restate that older shader text with this build's expression-valued arguments (an opt-in
extension, GAME_HIP_ANIMATE_ARGS): `circle(10.0) | shade(r, g, b)` stores exactly (r, g, b). The crops are committed
as data (tests/golden/make_reference_screenshot_crops.py); the dev server's two text overlays on the preview pane are
masked. Bar: every channel within one level of the browser's pixel, at least 95 % of them equal (exp and the final
rounding are the browser's own)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

RING = "exp(0.0 - max(abs(length(p) - 0.35) - {t}, 0.0) * {i} * 8.0)"
DOT = "exp(0.0 - max(length(p) - {r}, 0.0) * {i} * 8.0)"

# dist/dashboard-gauge.js: track ring(0.35, 0.025) glow 0.8 tint (0.85, 0.92, 1.0); fill arc masked away at fill = 0
# (mask_arc -> 999 -> exp underflows to 0); centre circle(0.04) glow 1.5 tint white; layers added.
_G_RING, _G_DOT = RING.format(t="0.025", i="0.8"), DOT.format(r="0.04", i="1.5")
GAUGE = f'''cinematic "gauge-as-rendered" {{
  layer {{
    fn: circle(10.0) | shade(0.85 * {_G_RING} + {_G_DOT}, 0.92 * {_G_RING} + {_G_DOT}, {_G_RING} + {_G_DOT})
  }}
}}'''

# dist/layered-scene.js: gradient mix((0, 0.02, 0.15), (0, 0, 0.1), length(uv)); ring(0.35, 0.03) glow glow_i tint
# (0, 1, 1); circle(0.08) glow orb_glow tint (0.831, 0.686, 0.216); layers added.
_L_RING, _L_DOT = RING.format(t="0.03", i="0.5"), DOT.format(r="0.08", i="1.0")
LAYERED = f'''cinematic "layered-as-rendered" {{
  layer {{
    fn: circle(10.0) | shade(0.831 * {_L_DOT},
                             mix(0.02, 0.0, length(uv)) + {_L_RING} + 0.686 * {_L_DOT},
                             mix(0.15, 0.1, length(uv)) + {_L_RING} + 0.216 * {_L_DOT})
  }}
}}'''

CASES = [("gauge_component", GAUGE), ("gauge_pane", GAUGE), ("layered_component", LAYERED), ("layered_pane", LAYERED)]


def _crops():
    return np.load(os.path.join(HERE, "golden", "reference_screenshots.npz"))


def _mask(h, w):
    """True where the canvas shows through: the preview pane carries a label at its top left and an fps counter at
    its bottom right; the component canvas carries nothing."""
    m = np.ones((h, w), dtype=bool)
    if w > 200:
        m[:24, :100] = False
        m[h - 30:, 630:] = False
    return m


def _compare(frame, crop):
    h, w, _ = crop.shape
    d = np.abs(frame[:, :, :3].astype(np.int32) - crop.astype(np.int32))[_mask(h, w)]
    assert d.max() <= 1, f"a channel is {d.max()} levels from the reference's pixel"
    assert (d == 0).mean() >= 0.95
    assert np.all(frame[:, :, 3] == 255)
    return (d == 0).mean()


@pytest.mark.parametrize("name,src", CASES)
def test_oracle_reproduces_the_reference_screenshot(game, oracle, name, src):
    crop = _crops()[name]
    h, w, _ = crop.shape
    cin = game.compile(src).cinematics[0]
    assert cin.pixel_expr_count == 3 and cin.derived_count == 0
    row = cin.uniform_rows([0.0], w, h)[0]
    for semantics in ("glsl", "wgsl"):
        _compare(oracle.render(cin.blob, w, h, row, semantics=semantics, animate=True), crop)
    # without the opt-in the expressions fall back to the parameter defaults, as the reference resolves them
    plain = oracle.render(cin.blob, w, h, row)
    assert np.abs(plain[:, :, :3].astype(np.int32) - crop.astype(np.int32)).max() > 100


def test_the_overlay_mask_is_what_keeps_the_pane_from_matching(game, oracle):
    crop = _crops()["gauge_pane"]
    h, w, _ = crop.shape
    cin = game.compile(GAUGE).cinematics[0]
    frame = oracle.render(cin.blob, w, h, cin.uniform_rows([0.0], w, h)[0], animate=True)
    d = np.abs(frame[:, :, :3].astype(np.int32) - crop.astype(np.int32)).max(axis=2)
    assert d[_mask(h, w)].max() <= 1 and d[~_mask(h, w)].max() > 30      # the label and the counter, nothing else


# A third render: the same cinematic just after the second moment of its arc timeline has begun (arc-09-params.png; its
# panel shows glow_i 0.61 and orb_glow 1.15, one 240 Hz frame behind the canvas). The runtime of that generation
# (dist/layered-scene.js, the `ignite` moment at t = 2 s) moves both parameters with one clock:
#   glow_i = 0.5 + 2.5 (1 - 2^(-10 e / 2)),  orb_glow = 1 + 5 (1 - 2^(-10 e / 3)),  e = seconds since the moment began.
# Fitted separately the two intensities come out at 0.6458 and 1.1964, which the formulas map back to e = 17.34 ms both
# times - so the test has ONE free number: some e between 16 and 19 ms must put every channel within one level of the
# browser's pixel. (The preview pane of that screenshot runs on its own clock, 1.1 s further on, where both glows are so
# tight that the fit of the second is loose: within one level at (2.95, 5.5), not used.) Today's arc runtime
# (runtime/arc.rs, which this build's animator follows) lays the same block out differently - transitions follow one
# another from a cursor instead of starting at their moment's time - so it is the shader conventions this render pins, at
# a third pair of intensities, not the timeline.
def _layered_at(e):
    glow_i = 0.5 + 2.5 * (1.0 - 2.0 ** (-10.0 * e / 2.0))
    orb_glow = 1.0 + 5.0 * (1.0 - 2.0 ** (-10.0 * e / 3.0))
    ring, dot = RING.format(t="0.03", i="%.6f" % glow_i), DOT.format(r="0.08", i="%.6f" % orb_glow)
    return f'''cinematic "layered-ignite" {{
  layer {{
    fn: circle(10.0) | shade(0.831 * {dot},
                             mix(0.02, 0.0, length(uv)) + {ring} + 0.686 * {dot},
                             mix(0.15, 0.1, length(uv)) + {ring} + 0.216 * {dot})
  }}
}}'''


def _fit_ignite(game, oracle, crop):
    best = None
    for e in np.arange(0.0160, 0.01901, 0.0002):
        cin = game.compile(_layered_at(float(e))).cinematics[0]
        frame = oracle.render(cin.blob, 200, 200, cin.uniform_rows([0.0], 200, 200)[0], animate=True)
        d = np.abs(frame[:, :, :3].astype(np.int32) - crop.astype(np.int32))
        key = (int(d.max()), -float((d == 0).mean()))
        if best is None or key < best[0]:
            best = (key, float(e))
    return best


def test_oracle_reproduces_the_render_inside_the_arc(game, oracle):
    crop = _crops()["layered_component_ignite"]
    (worst, neg_equal), e = _fit_ignite(game, oracle, crop)
    assert worst <= 1 and -neg_equal >= 0.95 and 0.0168 <= e <= 0.0180
    # the un-animated moment's intensities, and the ones 17 ms further on, do not fit these pixels
    for other in (0.0, 2.0 * e):
        cin = game.compile(_layered_at(other)).cinematics[0]
        frame = oracle.render(cin.blob, 200, 200, cin.uniform_rows([0.0], 200, 200)[0], animate=True)
        assert np.abs(frame[:, :, :3].astype(np.int32) - crop.astype(np.int32)).max() > 4


@pytest.mark.gpu
def test_gpu_reproduces_the_render_inside_the_arc(game, oracle):
    crop = _crops()["layered_component_ignite"]
    _, e = _fit_ignite(game, oracle, crop)
    cin = game.compile(_layered_at(e)).cinematics[0]
    rows = cin.uniform_rows([0.0], 200, 200)
    r = cin.renderer(200, 200, animate=True)
    got = r.render(rows)[0]
    r.close()
    _compare(got, crop)
    assert np.array_equal(got, oracle.render(cin.blob, 200, 200, rows[0], animate=True))


# Today's builtins against the same pixels, where the generations agree: the ring's distance function. In the screenshot
# the track ring's term is exactly 1 wherever ring(0.35, 0.025) <= 0 and falls off as exp(-6.4 d) outside, so the red channel
# equals round(255 (0.85 + centre-dot term)) inside the ring and is a level lower from d = 4e-4 on; today's
# `ring(0.35, 0.025) | threshold(0.0) | shade(1, 1, 1)` is white exactly where the distance is negative. The two masks must
# agree on every pixel of the annulus around the ring that is not within 1.5e-3 (an eighth of a pixel of the component
# canvas, half a pixel of the pane) of one of its two edges: that puts `p`, the aspect factor, `length` and `ring` of the
# v0.3.0 path on the reference's own pixels.
EDGE = 'cinematic "edge" { layer { fn: ring(0.35, 0.025) | threshold(0.0) | shade(1.0, 1.0, 1.0) } }'


def _ring_masks_agree(frame, crop):
    h, w, _ = crop.shape
    x = ((np.arange(w) + 0.5) / w * 2 - 1) * (w / h)
    y = ((np.arange(h) + 0.5) / h * 2 - 1)[::-1]
    ln = np.sqrt(x[None, :] ** 2 + y[:, None] ** 2)                    # float64 geometry, only to name the pixels near an edge
    d = np.abs(ln - 0.35) - 0.025
    dot = np.exp(-np.maximum(ln - 0.04, 0.0) * 1.5 * 8.0)
    in_reference = crop[:, :, 0].astype(np.int32) >= np.rint(255 * (0.85 + dot)) - 1       # the browser's exp may sit one level lower
    # white where the distance is negative; the WebGPU wording's threshold is step(cutoff, d) = 0 there, which shade turns
    # into smoothstep's midpoint: grey 128 in place of white
    in_ours = frame[:, :, 0] > 0
    assert set(np.unique(frame[:, :, 0]).tolist()) in ({0, 255}, {0, 128})
    annulus = (ln > 0.2) & (ln < 0.5) & _mask(h, w)
    differ = (in_ours != in_reference) & annulus
    assert in_ours[annulus].sum() > 1000 and (np.abs(d[differ]) < 1.5e-3).all()
    assert differ.sum() <= (np.abs(d[annulus]) < 1.5e-3).sum()


@pytest.mark.parametrize("name", ["gauge_component", "gauge_pane"])
def test_oracle_ring_edges_fall_on_the_reference_pixels(game, oracle, name):
    crop = _crops()[name]
    h, w, _ = crop.shape
    cin = game.compile(EDGE).cinematics[0]
    assert cin.pixel_expr_count == 0 and cin.derived_count == 0          # plain v0.3.0 stages, no extension involved
    for semantics in ("glsl", "wgsl"):
        _ring_masks_agree(oracle.render(cin.blob, w, h, cin.uniform_rows([0.0], w, h)[0], semantics=semantics), crop)
    # a ring half a percent larger no longer fits
    other = game.compile(EDGE.replace("0.35", "0.352")).cinematics[0]
    with pytest.raises(AssertionError):
        _ring_masks_agree(oracle.render(other.blob, w, h, other.uniform_rows([0.0], w, h)[0]), crop)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gauge_component", "gauge_pane"])
def test_gpu_ring_edges_fall_on_the_reference_pixels(game, name):
    crop = _crops()[name]
    h, w, _ = crop.shape
    cin = game.compile(EDGE).cinematics[0]
    r = cin.renderer(w, h)
    got = r.render(cin.uniform_rows([0.0], w, h))[0]
    r.close()
    _ring_masks_agree(got, crop)


@pytest.mark.gpu
@pytest.mark.parametrize("name,src", CASES)
def test_gpu_reproduces_the_reference_screenshot(game, oracle, name, src):
    crop = _crops()[name]
    h, w, _ = crop.shape
    cin = game.compile(src).cinematics[0]
    rows = cin.uniform_rows([0.0], w, h)
    for semantics in ("glsl", "wgsl"):
        r = cin.renderer(w, h, semantics=semantics, animate=True)
        got = r.render(rows)[0]
        r.close()
        _compare(got, crop)
        assert np.array_equal(got, oracle.render(cin.blob, w, h, rows[0], semantics=semantics, animate=True))
