"""CPU tests of the oracle itself (oracle/game_oracle.c): its math layer against libm/numpy, its
pixel pipeline against an independent numpy float32 restatement, and regression goldens.

The reference holds no pixel or particle golden vectors (its tests assert on emitted text), so the
oracle's numeric output is pinned only by (a) the text-level facts checked in test_front_*.py,
(b) the independent numpy restatement here, and (c) self-generated regression fixtures under
tests/golden/ (made by tests/golden/make_golden.py). Pixel parity with a browser is unpinned.
"""
import os

import numpy as np
import pytest

import np_restatement as NP
from conftest import C1_HELLO, C2_FBM, C2B_NOISE, C3_SHOWCASE, C4_TRAILS, C5_GRAVITY

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def vec(fn, *arrays):
    return np.array([fn(*[float(v) for v in row]) for row in zip(*arrays)], dtype=np.float32)


# ── math layer ──────────────────────────────────────────────────────────────────────────
def test_sin_cos_accuracy(oracle):
    rng = np.random.default_rng(1)
    for scale, tol in ((10.0, 2.5e-7), (1000.0, 2.5e-7), (30000.0, 3e-7), (5.0e7, 3e-7)):
        x = (rng.random(4000, dtype=np.float32) * 2 - 1) * np.float32(scale)
        s = vec(oracle.L.game_oracle_sin, x)
        c = vec(oracle.L.game_oracle_cos, x)
        assert np.max(np.abs(s - np.sin(x.astype(np.float64)))) < tol
        assert np.max(np.abs(c - np.cos(x.astype(np.float64)))) < tol
    assert oracle.L.game_oracle_sin(0.0) == 0.0 and oracle.L.game_oracle_cos(0.0) == 1.0
    assert np.isnan(oracle.L.game_oracle_sin(float("inf"))) and np.isnan(oracle.L.game_oracle_cos(float("nan")))
    assert oracle.L.game_oracle_sin(3.0e9) == 0.0 and oracle.L.game_oracle_cos(-3.0e9) == 1.0   # beyond 2^31: argument counts as 0


def test_atan2_accuracy_and_quadrants(oracle):
    rng = np.random.default_rng(2)
    y = (rng.random(5000, dtype=np.float32) * 2 - 1)
    x = (rng.random(5000, dtype=np.float32) * 2 - 1)
    a = vec(oracle.L.game_oracle_atan2, y, x)
    assert np.max(np.abs(a - np.arctan2(y.astype(np.float64), x.astype(np.float64)))) < 5e-7
    at = oracle.L.game_oracle_atan2
    assert at(0.0, 1.0) == 0.0 and at(0.0, 0.0) == 0.0
    assert at(0.0, -1.0) == pytest.approx(np.pi, abs=1e-6) and at(-0.0, -1.0) == pytest.approx(-np.pi, abs=1e-6)
    assert at(1.0, 0.0) == pytest.approx(np.pi / 2, abs=1e-6) and at(-1.0, 0.0) == pytest.approx(-np.pi / 2, abs=1e-6)
    assert at(float("inf"), float("inf")) == pytest.approx(np.pi / 4, abs=1e-6)
    assert np.isnan(at(float("nan"), 1.0))


def test_fract_edges(oracle):
    fr = oracle.L.game_oracle_fract
    assert fr(1.25) == 0.25 and fr(-0.25) == 0.75 and fr(3.0) == 0.0 and fr(-0.0) == 0.0
    assert fr(-1e-9) == float(np.nextafter(np.float32(1.0), np.float32(0.0)))     # never 1.0
    assert np.isnan(fr(float("inf"))) and np.isnan(fr(float("nan")))


def test_pow_exp_log(oracle):
    rng = np.random.default_rng(3)
    x = rng.random(2000, dtype=np.float32) * 20 - 10
    e = vec(oracle.L.game_oracle_exp, x)
    assert np.max(np.abs(e / np.exp(x.astype(np.float64)) - 1)) < 2e-6
    p = rng.random(2000, dtype=np.float32) * 100 + np.float32(1e-3)
    l = vec(oracle.L.game_oracle_log, p)
    assert np.max(np.abs(l - np.log(p.astype(np.float64)))) < 2e-6
    assert oracle.L.game_oracle_pow(2.0, 10.0) == pytest.approx(1024.0, rel=1e-6)
    assert oracle.L.game_oracle_pow(0.5, 2.0) == pytest.approx(0.25, rel=1e-6)


def test_noise_matches_numpy_restatement(oracle):
    rng = np.random.default_rng(4)
    px = ((rng.random(3000, dtype=np.float32) * 2 - 1) * np.float32(40)).astype(np.float32)
    py = ((rng.random(3000, dtype=np.float32) * 2 - 1) * np.float32(40)).astype(np.float32)
    ix, iy = np.floor(px), np.floor(py)
    assert np.array_equal(vec(oracle.L.game_oracle_hash2, ix, iy), NP.hash2(ix, iy))
    assert np.array_equal(vec(oracle.L.game_oracle_noise2, px, py), NP.noise2(px, py))
    got = np.array([oracle.L.game_oracle_fbm2(float(a), float(b), 6, 0.5, 2.0) for a, b in zip(px[:500], py[:500])], dtype=np.float32)
    assert np.array_equal(got, NP.fbm2(px[:500], py[:500], 6, 0.5, 2.0))
    h = vec(oracle.L.game_oracle_hash2, ix, iy)
    assert h.min() >= 0.0 and h.max() < 1.0 and 0.45 < h.mean() < 0.55
    n = vec(oracle.L.game_oracle_noise2, px, py)
    assert n.min() >= -1.0 and n.max() <= 1.0


def test_simplex_voronoi_ranges(oracle):
    rng = np.random.default_rng(5)
    px = ((rng.random(2000, dtype=np.float32) * 2 - 1) * np.float32(20)).astype(np.float32)
    py = ((rng.random(2000, dtype=np.float32) * 2 - 1) * np.float32(20)).astype(np.float32)
    s = vec(oracle.L.game_oracle_simplex2d, px, py)
    assert np.all(np.abs(s) <= 1.05) and s.std() > 0.2
    v = np.array([oracle.L.game_oracle_voronoi2d(float(a), float(b), 5.0) for a, b in zip(px, py)], dtype=np.float32)
    assert v.min() >= 0.0 and v.max() <= 1.0          # d starts at 1.0 (glsl.rs:206)
    # simplex noise is continuous: tiny steps move it little
    s2 = vec(oracle.L.game_oracle_simplex2d, px + np.float32(1e-4), py)
    assert np.max(np.abs(s2 - s)) < 5e-3


# ── whole frames against the numpy restatement (bit-exact) ──────────────────────────────
def frame(game, oracle, src, w, h, t=0.0, audio=None, threads=4):
    cin = game.compile(src).cinematics[0]
    rows = cin.uniform_rows([t], w, h, audio=audio)
    return oracle.render(cin.blob, w, h, rows[0], threads=threads), cin


@pytest.mark.parametrize("w,h", [(64, 64), (96, 54), (33, 17), (1, 1), (7, 1), (1, 9)])
def test_c1_frame_matches_numpy(game, oracle, w, h):
    got, _ = frame(game, oracle, C1_HELLO, w, h)
    assert np.array_equal(got, NP.render_c1(w, h))


@pytest.mark.parametrize("w,h", [(96, 54), (40, 40)])
def test_c2_frame_matches_numpy(game, oracle, w, h):
    got, _ = frame(game, oracle, C2_FBM, w, h)
    assert np.array_equal(got, NP.render_c2(w, h))


def test_c3_frame_matches_numpy_without_grain(game, oracle):
    src = C3_SHOWCASE.replace("| grain(0.025)", "")
    audio = {"bass": 0.8, "mid": 0.6, "treble": 0.4, "energy": 0.6}
    got, _ = frame(game, oracle, src, 96, 54, audio=audio)
    assert np.array_equal(got, NP.render_c3_no_grain(96, 54, 0.8, 0.6, 0.4))
    # odd width puts a pixel centre exactly on uv = (0, 0): normalize(0) is NaN, red and blue clamp to 0
    got, _ = frame(game, oracle, src, 33, 17, audio=audio)
    assert np.array_equal(got, NP.render_c3_no_grain(33, 17, 0.8, 0.6, 0.4))
    assert got[8, 16, 0] == 0 and got[8, 16, 2] == 0


def test_c3_frame_matches_numpy_with_grain(game, oracle):
    """The whole of examples/007-showcase.game, grain's sine hash included, at two times."""
    audio = {"bass": 0.8, "mid": 0.6, "treble": 0.4, "energy": 0.6}
    for t in (0.0, 2.75):
        got, _ = frame(game, oracle, C3_SHOWCASE, 96, 54, t=t, audio=audio)
        assert np.array_equal(got, NP.render_c3(96, 54, 0.8, 0.6, 0.4, t))          # t < 120: the wrapped time is t itself
    assert not np.array_equal(NP.render_c3(96, 54, 0.8, 0.6, 0.4, 0.0), NP.render_c3_no_grain(96, 54, 0.8, 0.6, 0.4))


def test_c3b_frame_matches_numpy(game, oracle):
    """The showcase chain with the tonemap stage the file lacks appended (workload c3b)."""
    from conftest import _W
    audio = {"bass": 0.8, "mid": 0.6, "treble": 0.4, "energy": 0.6}
    got, cin = frame(game, oracle, _W.C3B_SHOWCASE_TONEMAP, 96, 54, t=1.5, audio=audio)
    assert [st["name"] for st in cin.ir["layers"][0]["stages"]][-2:] == ["grain", "tonemap"]
    assert np.array_equal(got, NP.render_c3(96, 54, 0.8, 0.6, 0.4, 1.5, tonemap=1.0))
    assert not np.array_equal(got, NP.render_c3(96, 54, 0.8, 0.6, 0.4, 1.5))


@pytest.mark.parametrize("w,h", [(96, 54), (31, 40)])
def test_c2b_frame_matches_numpy(game, oracle, w, h):
    """examples/tutorial-05-noise.game: fbm, the permute-289 simplex, the sine-hash voronoi and the screen blend of
    three layers, against the numpy statement of glsl.rs:155-220, :251-258."""
    got, cin = frame(game, oracle, C2B_NOISE, w, h)
    assert [n for n, _ in cin.uniforms] == ["frost", "ember", "violet"]
    assert np.array_equal(got, NP.render_c2b(w, h))


@pytest.mark.parametrize("w,h", [(96, 54), (31, 40)])
def test_c2b_frame_in_the_webgpu_wording_matches_numpy(game, oracle, w, h):
    """The same cinematic as the reference's WebGPU emitter words it: arithmetic-hash voronoi (wgsl.rs:187-201) and
    sine-gradient simplex (wgsl.rs:206-222); the two wordings give different pictures."""
    cin = game.compile(C2B_NOISE).cinematics[0]
    got = oracle.render(cin.blob, w, h, cin.uniform_rows([0.0], w, h)[0], threads=4, semantics="wgsl")
    assert np.array_equal(got, NP.render_c2b_wgsl(w, h))
    assert not np.array_equal(got, NP.render_c2b(w, h))


def test_sine_of_the_render_path_matches_numpy(oracle):
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-32768, 32768, 20000), rng.uniform(-8, 8, 20000), [0.0, -0.0, 32768.0, -32768.0]]).astype(np.float32)
    got = np.array([oracle.L.game_oracle_sin(float(v)) for v in x], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), NP.sin32(x).view(np.uint32))


def test_wgsl_semantics_post_chain_matches_numpy(game, oracle):
    """The WebGPU wording of spectrum / chromatic / vignette / tonemap (radial bands, additive fringe, mixed
    vignette, ACES) against an independent numpy statement of wgsl.rs."""
    src = ('cinematic "w" { layer { fn: spectrum(0.9, 0.7, 0.5) | bloom(0.4, 2.0) | chromatic(0.6) | '
           'vignette(0.35, 0.8) | tonemap(1.2) } }')
    cin = game.compile(src).cinematics[0]
    for w, h in ((96, 54), (33, 17)):
        rows = cin.uniform_rows([0.0], w, h)
        got = oracle.render(cin.blob, w, h, rows[0], semantics="wgsl")
        assert np.array_equal(got, NP.render_wgsl_post_chain(w, h))
        assert not np.array_equal(got, oracle.render(cin.blob, w, h, rows[0]))     # the two wordings differ


def test_golden_builtin_gallery_wgsl(game, oracle):
    from gallery import GALLERY
    data = np.load(os.path.join(GOLDEN, "gallery_wgsl.npz"))
    assert sorted(data.files) == sorted(GALLERY)
    for name, src in GALLERY.items():
        cin = game.compile(src).cinematics[0]
        rows = cin.uniform_rows([2.5], 48, 27, audio={"bass": 0.7, "mid": 0.2, "treble": 0.9})
        assert np.array_equal(oracle.render(cin.blob, 48, 27, rows[0], threads=4, semantics="wgsl"), data[name]), name


def test_memory_feedback_recurrence(game, oracle):
    """Frame k = mix(current, RGBA8(frame k-1), 0.97) per channel, re-quantised every frame."""
    w, h, n = 48, 27, 8
    cin = game.compile(C4_TRAILS).cinematics[0]
    bass = [0.5 + 0.5 * np.sin(2 * np.pi * 0.5 * k / 60) for k in range(n)]
    rows = cin.uniform_rows([k / 60 for k in range(n)], w, h, audio={"bass": bass})
    seq = oracle.render_sequence(cin.blob, w, h, rows, True, threads=2)
    uvx, uvy = NP.pixel_grid(w, h)
    prev = np.zeros((h, w, 4), dtype=np.uint8)
    for k in range(n):
        px = uvx * (NP.F(w) / NP.F(h)) - NP.F(bass[k])
        py = uvy - NP.F(0.0)
        sdf = np.sqrt(px * px + py * py) - NP.F(0.1)
        g = NP.apply_glow(sdf, 2.5)
        cur = np.stack([g * NP.F(1.0), g * NP.F(0.5), g * NP.F(0.2), np.ones_like(g)], axis=-1)
        mixed = NP.memory_mix(cur, prev, 0.97)
        want = NP.pack(mixed[..., 0], mixed[..., 1], mixed[..., 2], mixed[..., 3])
        assert np.array_equal(seq[k], want), f"frame {k}"
        prev = want
    assert seq[0].max() > 0 and not np.array_equal(seq[0], seq[-1])


def test_band_rendering_equals_full_frame(game, oracle):
    cin = game.compile(C2B_NOISE).cinematics[0]
    w, h = 64, 36
    row = cin.uniform_rows([1.25], w, h)[0]
    full = oracle.render(cin.blob, w, h, row, threads=1)
    parts = [oracle.render(cin.blob, w, h, row, row0=r0, rows=n, threads=3) for r0, n in ((0, 5), (5, 20), (25, 11))]
    assert np.array_equal(np.concatenate(parts), full)


def test_layer_compositing_rules(game, oracle):
    w, h = 32, 18
    one, _ = frame(game, oracle, 'cinematic "t" { layer a { circle(0.3) | glow(2.0) | tint(0.5, 0.25, 0.125) } }', w, h)
    # a params layer makes the cinematic "multi": screen blend over black gives the same rgb, alpha 1
    two, _ = frame(game, oracle, 'cinematic "t" { layer cfg { k: 1.0 } layer a { circle(0.3) | glow(2.0) | tint(0.5, 0.25, 0.125) } }', w, h)
    assert np.array_equal(one, two)
    # emissive writes alpha = glow; a single layer keeps it, multi-layer forces alpha to 1
    e1, _ = frame(game, oracle, 'cinematic "t" { layer a { circle(0.3) | emissive(0.2) } }', w, h)
    assert e1[..., 3].min() < 255
    e2, _ = frame(game, oracle, 'cinematic "t" { layer cfg { k: 1.0 } layer a { circle(0.3) | emissive(0.2) } }', w, h)
    assert e2[..., 3].min() == 255
    # no pipeline layer at all: nothing is written, opaque black
    none, cin = frame(game, oracle, 'cinematic "t" { layer cfg { k: 1.0 } }', w, h)
    assert np.array_equal(none[..., :3], np.zeros((h, w, 3), np.uint8)) and (cin.flags & 128)


def test_time_wrap_and_uniforms(game, oracle):
    src = 'cinematic "t" { layer a { concentric_waves(1.0, 0.5, 3.0) | glow(1.0) } }'
    a, _ = frame(game, oracle, src, 32, 18, t=5.0)
    b, _ = frame(game, oracle, src, 32, 18, t=125.0)       # time = fract(t / 120) * 120 (glsl.rs:56)
    c, _ = frame(game, oracle, src, 32, 18, t=6.0)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    cin = game.compile('cinematic "t" { layer a { circle(r) | glow(2.0) } layer cfg { r: 0.25 } }').cinematics[0]
    rows = cin.uniform_rows([0.0, 0.0], 32, 18, user={"r": [0.25, 0.4]})
    f0 = oracle.render(cin.blob, 32, 18, rows[0])
    f1 = oracle.render(cin.blob, 32, 18, rows[1])
    assert not np.array_equal(f0, f1)
    lit, _ = frame(game, oracle, 'cinematic "t" { layer cfg { k: 1.0 } layer a { circle(0.25) | glow(2.0) } }', 32, 18)
    assert np.array_equal(f0, lit)


# ── regression goldens ──────────────────────────────────────────────────────────────────
GOLDEN_CASES = {
    "c1_hello": (C1_HELLO, 64, 64, 0.0, None),
    "c2_fbm": (C2_FBM, 96, 54, 0.0, None),
    "c2b_noise": (C2B_NOISE, 96, 54, 0.5, None),
    "c3_showcase": (C3_SHOWCASE, 96, 54, 1.0, {"bass": 0.8, "mid": 0.6, "treble": 0.4, "energy": 0.6}),
}


@pytest.mark.parametrize("name", sorted(GOLDEN_CASES))
def test_golden_frames(game, oracle, name):
    src, w, h, t, audio = GOLDEN_CASES[name]
    got, _ = frame(game, oracle, src, w, h, t=t, audio=audio)
    want = np.load(os.path.join(GOLDEN, name + ".npy"))
    assert np.array_equal(got, want)


def test_golden_builtin_gallery(game, oracle):
    from gallery import GALLERY
    data = np.load(os.path.join(GOLDEN, "gallery.npz"))
    for name, src in GALLERY.items():
        got, _ = frame(game, oracle, src, 48, 27, t=2.5, audio={"bass": 0.7, "mid": 0.2, "treble": 0.9})
        assert np.array_equal(got, data[name]), name


# ── gravity ─────────────────────────────────────────────────────────────────────────────
def test_gravity_two_body_by_hand(game, oracle):
    cin = game.compile(C5_GRAVITY).cinematics[0]
    p = np.array([[-0.5, 0.0, 0.0, 0.0], [0.5, 0.0, 0.0, 0.0]], dtype=np.float32)
    out, t = oracle.gravity(cin.blob, p, 0.1, 1)
    # constant attraction 1.0 along +x for particle 0: v = 0*0.995 + 1*0.1, x = -0.5 + v*0.1
    v = np.float32(0.0) * np.float32(0.995) + np.float32(1.0) * np.float32(0.1)
    assert out[0, 2] == v and out[0, 0] == np.float32(-0.5) + v * np.float32(0.1)
    assert out[1, 2] == -v and t == pytest.approx(0.1)
    assert np.all(out[:, 1] == 0) and np.all(out[:, 3] == 0)


def test_gravity_bounds_modes(game, oracle):
    base = 'cinematic "t" { gravity { rule: 0.0, damping: 1.0, bounds: %s } layer s { circle() | glow() } }'
    p = np.array([[0.95, -0.95, 1.0, -1.0]], dtype=np.float32)
    refl, _ = oracle.gravity(game.compile(base % "reflect").cinematics[0].blob, p, 0.1, 1)
    assert refl[0, 0] == 1.0 and refl[0, 2] == -1.0 and refl[0, 1] == -1.0 and refl[0, 3] == 1.0
    wrap, _ = oracle.gravity(game.compile(base % "wrap").cinematics[0].blob, p, 0.1, 1)
    assert wrap[0, 0] == pytest.approx(-0.95, abs=1e-6) and wrap[0, 1] == pytest.approx(0.95, abs=1e-6)
    none, _ = oracle.gravity(game.compile(base % "none").cinematics[0].blob, p, 0.1, 1)
    assert none[0, 0] == pytest.approx(1.05, abs=1e-6) and none[0, 1] == pytest.approx(-1.05, abs=1e-6)


def test_gravity_force_laws_and_threads(game, oracle):
    src = 'cinematic "t" { gravity { rule: 0.01 / (dist * dist + 0.01) * sin(time * 3.0 + 1.0), damping: 0.98, bounds: wrap } layer s { circle() | glow() } }'
    cin = game.compile(src).cinematics[0]
    p = oracle.seed_particles(257)
    a, ta = oracle.gravity(cin.blob, p, 1 / 60, 5, threads=1)
    b, tb = oracle.gravity(cin.blob, p, 1 / 60, 5, threads=5)
    assert np.array_equal(a, b) and ta == tb
    assert np.all(np.abs(a[:, :2]) <= 1.0) and not np.array_equal(a, p)
    g = np.load(os.path.join(GOLDEN, "gravity_257.npy"))
    assert np.array_equal(a, g)


def test_inversesqrt_is_correctly_rounded(oracle):
    """gm_rsqrt (the reciprocal root of the gravity step and of normalize(uv)) against integer arithmetic over every
    float of [1, 4) - all significands at both exponent parities, which stand for every normal input because
    1 / sqrt(4^k x) = 2^-k / sqrt(x) exactly - and its special values."""
    import ctypes
    first = ctypes.c_uint32(0)
    assert oracle.L.game_oracle_rsqrt_selfcheck(ctypes.byref(first)) == 0, hex(first.value)
    L = oracle.L
    assert L.game_oracle_rsqrt(4.0) == 0.5 and L.game_oracle_rsqrt(0.25) == 2.0 and L.game_oracle_rsqrt(2.0 ** -100) == 2.0 ** 50
    assert L.game_oracle_rsqrt(0.0) == np.inf and L.game_oracle_rsqrt(np.inf) == 0.0 and np.isnan(L.game_oracle_rsqrt(-1.0))
    # the one excepted class, 4^k (1 - 2^-23) -> 2^-k, at both ends of the range and in the middle
    for k in (-49, 0, 1, 64):
        x = np.float32(4.0 ** k if k < 64 else np.inf)
        x = np.nextafter(np.nextafter(x, np.float32(0)), np.float32(0))
        assert L.game_oracle_rsqrt(float(x)) == 2.0 ** -k, k
        from np_restatement import rsqrt as np_rsqrt
        assert float(np_rsqrt(np.array([x], dtype=np.float32))[0]) == 2.0 ** -k
    # scaling by powers of four is exact, and the numpy statement of the same definition agrees
    from np_restatement import rsqrt
    rng = np.random.default_rng(3)
    x = (rng.random(20000, dtype=np.float32) * 3 + 1).astype(np.float32)
    for k in (-40, -7, 0, 9, 50):
        xs = (x * np.float32(4.0 ** k)).astype(np.float32)
        got = np.empty_like(xs)
        L.game_oracle_rsqrt_array(xs.size, xs.ctypes.data, got.ctypes.data)
        assert np.array_equal(got, rsqrt(xs))
        base = np.empty_like(x)
        L.game_oracle_rsqrt_array(x.size, x.ctypes.data, base.ctypes.data)
        assert np.array_equal(got, (base * np.float32(2.0 ** -k)).astype(np.float32))


def test_gravity_range_equals_whole_step(game, oracle):
    """game_oracle_gravity_range (a few particles against all) returns what the whole-set step gives those particles."""
    cin = game.compile(C5_GRAVITY).cinematics[0]
    p = oracle.seed_particles(1500, seed=5)
    whole, t = oracle.gravity(cin.blob, p, 1 / 60, 1)
    for first, count in ((0, 1500), (0, 7), (700, 300), (1499, 1)):
        part = oracle.gravity_range(cin.blob, p, 1 / 60, t, first, count, threads=3)
        assert np.array_equal(part, whole[first:first + count]), (first, count)
    law = 'cinematic "t" { gravity { rule: 0.1 / (dist + 0.2) * cos(time), damping: 0.9, bounds: wrap } layer s { circle() | glow() } }'
    cin = game.compile(law).cinematics[0]
    whole, t = oracle.gravity(cin.blob, p, 1 / 60, 1, sim_time=0.5)
    assert np.array_equal(oracle.gravity_range(cin.blob, p, 1 / 60, t, 300, 500), whole[300:800])


def test_seed_particles_known_answer(oracle):
    p = oracle.seed_particles(4)
    # SplitMix64 from seed 0x9E3779B97F4A7C15, top 24 bits of each draw mapped to [-1, 1)
    s, vals = 0x9E3779B97F4A7C15, []
    for _ in range(8):
        s = (s + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        z ^= z >> 31
        vals.append(np.float32(z >> 40) * np.float32(2.0 ** -23) - np.float32(1.0))
    assert np.array_equal(p[:, :2].reshape(-1), np.array(vals, dtype=np.float32)) and np.all(p[:, 2:] == 0)


def test_unorm8_texel_fetch_and_framebuffer_write_round_trip():
    """A stored byte is fetched as b * RN(1 / 255) (255 -> exactly 1, within one ulp of b / 255 everywhere) and a colour
    is stored as c * 255 rounded to nearest (ties to even), saturated: fetch then store returns every byte."""
    from np_restatement import pack
    b = np.arange(256, dtype=np.float32)
    r = np.float32(1.0) / np.float32(255.0)
    assert float(r).hex() == "0x1.0101020000000p-8"
    f = b * r
    assert f[255] == np.float32(1.0) and f[0] == 0
    exact = b / np.float32(255.0)
    assert np.all(np.abs(f.view(np.int32) - exact.view(np.int32)) <= 1)
    assert np.array_equal(pack(f, f, f, 1.0)[:, 0], np.arange(256, dtype=np.uint8))
    edge = np.array([-1.0, -0.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 0.5, 1.0, 7.0, np.inf, -np.inf, np.nan], dtype=np.float32)
    assert pack(edge, edge, edge, 1.0)[:, 0].tolist() == [0, 0, 0, 2, 2, 128, 255, 255, 255, 0, 0]


def test_merged_distance_floor_constant():
    """The gravity kernel computes max(sqrt(d2), 0.001f) as sqrt(max(d2, c)) with c the smallest float
    whose correctly rounded root is 0.001f; both forms must agree for every d2."""
    c = np.array([0x358637BD], dtype=np.uint32).view(np.float32)[0]
    assert float(c).hex() == "0x1.0c6f7a0000000p-20"
    floor = np.float32(0.001)
    assert np.sqrt(c) == floor and np.sqrt(np.nextafter(c, np.float32(0))) < floor
    near = (np.arange(-2_000_000, 2_000_001, dtype=np.int64) + int(c.view(np.uint32))).astype(np.uint32).view(np.float32)
    rng = np.random.default_rng(5)
    wide = rng.integers(0, 0x7F800000, size=4_000_000, dtype=np.uint32).view(np.float32)   # every non-negative finite range
    special = np.array([0.0, np.inf, np.nan, 1e-45, 1e-38, 3.4e38], dtype=np.float32)
    for d2 in (near, wide, special):
        want = np.maximum(np.sqrt(d2), floor)
        want = np.where(np.isnan(d2), floor, want)          # max(NaN, 0.001) = 0.001 in the render path
        got = np.sqrt(np.where(np.isnan(d2), c, np.maximum(d2, c)))
        assert np.array_equal(got.view(np.uint32), want.astype(np.float32).view(np.uint32))


@pytest.mark.parametrize("bounds", ["reflect", "wrap", "none"])
def test_gravity_matches_numpy_restatement(game, oracle, bounds):
    """Independent numpy statement of the particle step (no transcendental in the law) against the C oracle."""
    src = ('cinematic "t" { gravity { rule: 0.02 / (dist * dist + 0.05) - 0.01, damping: 0.97, bounds: %s } '
           'layer s { circle() | glow() } }') % bounds
    cin = game.compile(src).cinematics[0]
    p = oracle.seed_particles(96)
    p[:, 2:] = (np.random.default_rng(3).random((96, 2), dtype=np.float32) - np.float32(0.5)) * np.float32(4.0)
    p[5, :2] = p[6, :2]                                    # a coincident pair exercises the 0.001 floor
    want = p.copy()
    t = np.float32(0.0)
    law = lambda dist, time: np.float32(0.02) / (dist * dist + np.float32(0.05)) - np.float32(0.01)
    for _ in range(3):
        t = np.float32(t + np.float32(1 / 60))
        want = NP.gravity_step(want, np.float32(1 / 60), 0.97, law, bounds, t)
    got, _ = oracle.gravity(cin.blob, p, 1 / 60, 3)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_blocked_gravity_loop_equals_scalar_loop(game, oracle):
    """Constant force laws take an eight-particles-at-a-time loop (AVX2); it must give every particle the bits the
    one-particle loop gives it, wild positions included."""
    for bounds in ("reflect", "wrap", "none"):
        cin = game.compile('cinematic "t" { gravity { rule: 0.7, damping: 0.98, bounds: %s } layer s { circle() | glow() } }' % bounds).cinematics[0]
        p = oracle.seed_particles(1501, seed=3)
        p[5, 0] = np.nan
        p[7, :2] = p[8, :2]
        p[9, 0] = 1e30
        p[11, 1] = np.inf
        a, ta = oracle.gravity(cin.blob, p, 1 / 60, 3, threads=3)
        b, tb = oracle.gravity(cin.blob, p, 1 / 60, 3, threads=3, blocked=False)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and ta == tb


def _psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def test_deviations_from_the_literal_wording_stay_inside_the_image_tolerance(game, oracle):
    """Where GLSL leaves the evaluation open the render path picks the cheap form (fused multiply-adds at fixed points,
    quotients as reciprocal-multiply, the frame store as round-to-nearest-even of c * 255, the texel fetch as
    byte * RN(1 / 255)). Against the reference's wording taken literally - every operator one IEEE operation, true
    divisions, round half up - the frames must stay inside the stated bar: at most 1/255 per channel, PSNR >= 50 dB."""
    w, h = 96, 54
    cases = []
    for src, fn, args in ((C1_HELLO, NP.render_c1, (w, h)), (C2_FBM, NP.render_c2, (w, h)),
                          (C3_SHOWCASE.replace("\n      | grain(0.025)", ""), NP.render_c3_no_grain, (w, h, 0.8, 0.6, 0.4))):
        cin = game.compile(src).cinematics[0]
        row = cin.uniform_rows([0.0], w, h, audio={"bass": 0.8, "mid": 0.6, "treble": 0.4})[0]
        got = oracle.render(cin.blob, w, h, row)
        with NP.literal_form():
            lit = fn(*args)
        cases.append((cin.name, got, lit))
    # the feedback recurrence over 24 frames: differences must not build up through the re-quantised previous frame
    cin = game.compile(C4_TRAILS).cinematics[0]
    n = 24
    bass = [0.5 + 0.5 * np.sin(2 * np.pi * 0.5 * k / 60) for k in range(n)]
    rows = cin.uniform_rows([k / 60 for k in range(n)], w, h, audio={"bass": bass})
    seq = oracle.render_sequence(cin.blob, w, h, rows, True, threads=2)
    uvx, uvy = NP.pixel_grid(w, h)
    prev = np.zeros((h, w, 4), dtype=np.uint8)
    with NP.literal_form():
        for k in range(n):
            px, py = uvx * (NP.F(w) / NP.F(h)) - NP.F(bass[k]), uvy - NP.F(0.0)
            g = NP.apply_glow(np.sqrt(px * px + py * py) - NP.F(0.1), 2.5)
            cur = np.stack([g * NP.F(1.0), g * NP.F(0.5), g * NP.F(0.2), np.ones_like(g)], axis=-1)
            mixed = NP.memory_mix(cur, prev, 0.97)
            prev = NP.pack(mixed[..., 0], mixed[..., 1], mixed[..., 2], mixed[..., 3])
    cases.append(("trails frame 24", seq[-1], prev))
    for name, got, lit in cases:
        ok = ~(np.isnan(lit.astype(np.float32)))            # (uint8: always true; kept for symmetry with the uv = 0 pixel)
        diff = np.abs(got.astype(np.int16) - lit.astype(np.int16))[ok]
        assert diff.max() <= 1, (name, int(diff.max()))
        assert _psnr(got, lit) >= 50.0, (name, _psnr(got, lit))


def test_gravity_deviation_from_the_literal_wording(game, oracle):
    """gravity.rs:82-87 words the pair term as `dir = diff / dist; force = force + dir * f`; the render path multiplies
    by one correctly rounded reciprocal and accumulates with a fused multiply-add. One step of both forms agrees to
    1e-5 relative (the bar for 100 steps is 1e-4; over many steps of a chaotic system only bit-identical forms agree)."""
    cin = game.compile('cinematic "t" { gravity { rule: 0.05 / (dist + 0.2), damping: 0.97, bounds: none } layer s { circle() | glow() } }').cinematics[0]
    p0 = oracle.seed_particles(300, seed=21)
    got, _ = oracle.gravity(cin.blob, p0, 1 / 60, 1)
    law = lambda d, t: NP.F(0.05) / (d + NP.F(0.2))
    with NP.literal_form():
        lit = NP.gravity_step(p0, np.float32(1 / 60), 0.97, law, "none", np.float32(1 / 60))
    rel = np.abs(got - lit) / np.maximum(np.abs(lit), 1e-3)
    assert rel.max() <= 1e-5, rel.max()
    assert not np.array_equal(got, lit)
