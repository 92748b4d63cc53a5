"""GPU parity: the HIP render path (through the C ABI) against the CPU oracle on the same inputs.

Bar: every RGBA8 byte identical (the device code performs the same IEEE operations in the same
order as the oracle, so the tolerance the task allows — 1/255 per channel, PSNR >= 50 dB — is met
with zero difference); particle positions bit-identical in the default (IEEE division / sqrt)
integrator. At the benchmark's full sizes the oracle is too slow for whole frames, so parity there
is checked on row bands plus size-independent properties (band decomposition, determinism,
time-invariance, the feedback recurrence).
"""
import numpy as np
import pytest

from conftest import C1_HELLO, C2_FBM, C2B_NOISE, C3_SHOWCASE, C4_TRAILS, C5_GRAVITY
from gallery import GALLERY

pytestmark = pytest.mark.gpu

AUDIO = {"bass": 0.8, "mid": 0.6, "treble": 0.4, "energy": 0.6}


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def check(game, oracle, src, w, h, times, audio=None, user=None, semantics="glsl"):
    cin = game.compile(src).cinematics[0]
    rows = cin.uniform_rows(times, w, h, audio=audio, user=user)
    r = cin.renderer(w, h, semantics=semantics)
    try:
        got = r.render(rows)
    finally:
        r.close()
    want = oracle.render_sequence(cin.blob, w, h, rows, cin.uses_memory, semantics=semantics)
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and psnr(got, want) >= 50.0       # the stated tolerance
    assert diff.max() == 0, f"{np.count_nonzero(diff.max(axis=-1))} pixels differ"   # and in fact exact
    return got


def test_device_present(game):
    assert game.device_count() >= 1


# ── hardware instructions vs their defining formulas, all 2^32 inputs ───────────────────
@pytest.mark.parametrize("which", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_exhaustive_instruction_semantics(game, which):
    mismatches, first = game.check_exhaustive(which)
    assert mismatches == 0, hex(first)


# ── device math vs the oracle's math, bit for bit ───────────────────────────────────────
def _bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def test_device_math_matches_oracle(game, oracle):
    rng = np.random.default_rng(7)
    n = 20000
    x = np.concatenate([(rng.random(n, dtype=np.float32) * 2 - 1) * np.float32(s) for s in (4.0, 300.0, 30000.0, 5.0e7, 4.0e9)])
    x = np.concatenate([x, np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 32768.0, -32768.0, 32768.004, 2147483648.0], dtype=np.float32)])
    y = rng.permutation(x).astype(np.float32)
    L = oracle.L
    for op, fn, args in (("sin", L.game_oracle_sin, (x,)), ("cos", L.game_oracle_cos, (x,)), ("fract", L.game_oracle_fract, (x,)),
                         ("atan2", L.game_oracle_atan2, (x, y))):
        want = np.array([fn(*[float(v) for v in row]) for row in zip(*args)], dtype=np.float32)
        got = game.device_eval(op, *args)
        same = (_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (op, x[~same][:5], got[~same][:5], want[~same][:5])
    px = ((rng.random(n, dtype=np.float32) * 2 - 1) * np.float32(50)).astype(np.float32)
    py = ((rng.random(n, dtype=np.float32) * 2 - 1) * np.float32(50)).astype(np.float32)
    for op, fn in (("hash2", L.game_oracle_hash2), ("noise2", L.game_oracle_noise2), ("simplex2d", L.game_oracle_simplex2d)):
        want = np.array([fn(float(a), float(b)) for a, b in zip(px, py)], dtype=np.float32)
        assert np.array_equal(_bits(game.device_eval(op, px, py)), _bits(want)), op
    want = np.array([L.game_oracle_voronoi2d(float(a), float(b), 5.0) for a, b in zip(px[:4000], py[:4000])], dtype=np.float32)
    assert np.array_equal(_bits(game.device_eval("voronoi2d", px[:4000], py[:4000])), _bits(want))
    want = np.array([L.game_oracle_fbm2(float(a), float(b), 6, 0.5, 2.0) for a, b in zip(px[:4000], py[:4000])], dtype=np.float32)
    assert np.array_equal(_bits(game.device_eval("fbm6", px[:4000], py[:4000])), _bits(want))
    pos = np.abs(x[np.isfinite(x)][:5000]) + np.float32(1e-30)
    want = np.array([L.game_oracle_pow(float(a), float(b)) for a, b in zip(pos, y[:pos.size] * np.float32(1e-3))], dtype=np.float32)
    got = game.device_eval("pow", pos, y[:pos.size] * np.float32(1e-3))
    assert ((_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))).all()


def test_device_division_and_sqrt_are_correctly_rounded(game):
    rng = np.random.default_rng(8)
    a = ((rng.random(50000, dtype=np.float32) * 2 - 1) * np.float32(1e3)).astype(np.float32)
    b = ((rng.random(50000, dtype=np.float32) * 2 - 1) * np.float32(7)).astype(np.float32)
    b[b == 0] = 1
    assert np.array_equal(_bits(game.device_eval("div", a, b)), _bits(a / b))
    assert np.array_equal(_bits(game.device_eval("sqrt", np.abs(a))), _bits(np.sqrt(np.abs(a))))
    tiny = (rng.random(1000, dtype=np.float32) * np.float32(1e-38)).astype(np.float32)     # subnormal range
    assert np.array_equal(_bits(game.device_eval("sqrt", tiny)), _bits(np.sqrt(tiny)))
    assert np.array_equal(_bits(game.device_eval("div", tiny, b[:1000])), _bits(tiny / b[:1000]))


def test_device_inversesqrt_matches_oracle(game, oracle):
    """The seven-instruction reciprocal root (one v_rsq_f32, one Newton step with an exact residual) against the
    oracle's correctly rounded inversesqrt - which test_oracle checks against integer arithmetic - element by element;
    game_hip_check_exhaustive(5) and (7) cover every input against the same arbitration evaluated on the device."""
    rng = np.random.default_rng(31)
    x = np.concatenate([np.exp(rng.uniform(-69.0, 88.0, 400000)).astype(np.float32),
                        (rng.random(200000, dtype=np.float32) * 8).astype(np.float32),
                        np.array([2.0 ** -100, 2.0 ** -101, 1e-38, 1e-45, 0.0, -0.0, 1.0, 4.0, 3.4e38, np.inf, -1.0, np.nan,
                                  float.fromhex("0x1.0c6f7ap-20")], dtype=np.float32)])
    want = np.empty_like(x)
    oracle.L.game_oracle_rsqrt_array(x.size, x.ctypes.data, want.ctypes.data)
    got = game.device_eval("rsqrt", x)
    same = (_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), (x[~same][:5], got[~same][:5], want[~same][:5])


def test_lattice_hash_table_path_equals_arithmetic_path(game):
    """noise2 / fbm2 through the per-device hash table against the arithmetic path, on coordinates inside
    the table, on its edges (|floor| = 511, 512), far outside, and on non-finite inputs."""
    rng = np.random.default_rng(17)
    n = 200000
    parts = [(rng.random((n, 2), dtype=np.float32) * 2 - 1) * np.float32(s) for s in (3.0, 200.0, 515.0, 5000.0, 3.0e7)]
    edge = np.array([[511.0, 0.5], [511.99, -511.99], [512.0, 3.0], [-512.0, 3.0], [-511.5, 511.5], [-512.01, 0.0], [0.0, 512.5],
                     [np.nan, 1.0], [1.0, np.nan], [np.inf, 0.0], [0.0, -np.inf], [-0.0, -0.0], [1e-40, -1e-40]], dtype=np.float32)
    # tiny coordinates: the table path of fbm carries the sample point through the octaves (no frequency multiplies) only
    # where no intermediate of the literal recurrence can be subnormal - both coordinates at least 2^-40 in size; around
    # that bound, far below it, subnormal, and with one coordinate tiny or zero beside an ordinary one
    m = 20000
    tiny = [(rng.random((m, 2), dtype=np.float32) * 2 - 1) * np.float32(s) for s in (2.0 ** -37, 2.0 ** -40, 2.0 ** -43, 1e-30, 3e-38, 1e-41)]
    mixed = rng.random((m, 2), dtype=np.float32) * 2 - 1
    mixed[: m // 2, 0] *= np.float32(2.0 ** -41)
    mixed[m // 2:, 1] *= np.float32(1e-39)
    mixed[::7, 1] = 0.0
    mixed[3::11, 0] = -0.0
    xy = np.concatenate(parts + tiny + [mixed, edge]).astype(np.float32)
    x, y = np.ascontiguousarray(xy[:, 0]), np.ascontiguousarray(xy[:, 1])
    for lut_op, op in (("noise2_lut", "noise2"), ("fbm6_lut", "fbm6"), ("voronoi2d_lut", "voronoi2d")):
        if op == "voronoi2d":        # scale 5: put the table edges (|floor| = 510, 511) inside the sample
            x, y = (x * np.float32(0.2)).astype(np.float32), (y * np.float32(0.2)).astype(np.float32)
        got, want = game.device_eval(lut_op, x, y), game.device_eval(op, x, y)
        same = (_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (lut_op, xy[~same][:5], got[~same][:5], want[~same][:5])


@pytest.mark.gpu
def test_simplex_corner_table_path_equals_arithmetic_path(game):
    """simplex2d through the per-device corner table against the arithmetic path: ordinary coordinates, the
    mod-289 wrap (cells 288 | 0 and negatives), magnitudes where the wrap stops yielding integers, non-finite."""
    rng = np.random.default_rng(23)
    n = 200000
    parts = [(rng.random((n, 2), dtype=np.float32) * 2 - 1) * np.float32(s) for s in (3.0, 300.0, 1.0e5, 3.0e7, 1.0e12)]
    edge = np.array([[288.2, 0.3], [288.9, 288.9], [289.0, 289.0], [-0.2, -0.1], [-289.5, 577.9], [0.0, 0.0], [-0.0, 0.0],
                     [np.nan, 1.0], [1.0, np.nan], [np.inf, 0.0], [0.0, -np.inf], [1e-40, -1e-40], [16777216.0, 3.0], [3.4e38, -3.4e38]],
                    dtype=np.float32)
    xy = np.concatenate(parts + [edge]).astype(np.float32)
    x, y = np.ascontiguousarray(xy[:, 0]), np.ascontiguousarray(xy[:, 1])
    got, want = game.device_eval("simplex2d_lut", x, y), game.device_eval("simplex2d", x, y)
    same = (_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), (xy[~same][:5], got[~same][:5], want[~same][:5])


def test_unorm8_texel_fetch_for_all_bytes(game):
    b = np.arange(256, dtype=np.float32)
    assert np.array_equal(_bits(game.device_eval("unorm8", b)), _bits(b * (np.float32(1.0) / np.float32(255.0))))


def test_rgba8_pack_rounds_to_nearest_even_and_saturates(game):
    """The framebuffer write (one v_cvt_pk_u8_f32 per channel) against its formula, over ties, the edges of the range
    and non-finite values."""
    from np_restatement import pack
    rng = np.random.default_rng(5)
    ties = (np.arange(0, 256, dtype=np.float32) + np.float32(0.5)) / np.float32(255.0)
    c = np.concatenate([ties, np.nextafter(ties, np.float32(2)), np.nextafter(ties, np.float32(-2)),
                        rng.uniform(-0.5, 1.5, 4096).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, 2.0, -3.0, np.inf, -np.inf, np.nan, 1e-30, 3e38], dtype=np.float32)])
    got = game.device_eval("pack_unorm8", c).astype(np.uint32)
    assert np.array_equal(got.astype(np.uint8), pack(c, c, c, 1.0)[:, 0])


def test_reciprocal_multiply_quotients(game):
    # mod and smoothstep are defined with one correctly rounded reciprocal and one multiply
    rng = np.random.default_rng(9)
    x = ((rng.random(50000, dtype=np.float32) * 2 - 1) * np.float32(1e4)).astype(np.float32)
    y = np.full_like(x, 289.0)
    want = x - y * np.floor(x * (np.float32(1.0) / y))
    assert np.array_equal(_bits(game.device_eval("mod", x, y)), _bits(want))
    d = ((rng.random(50000, dtype=np.float32) * 2 - 1) * np.float32(0.02)).astype(np.float32)
    from np_restatement import smoothstep
    assert np.array_equal(_bits(game.device_eval("smoothstep_glow", d)), _bits(smoothstep(0.005, -0.005, d)))


# ── the five benchmark cinematics at oracle-sized frames ────────────────────────────────
def test_c1_hello(game, oracle):
    got = check(game, oracle, C1_HELLO, 256, 256, [0.0, 1.0 / 60, 2.5])
    assert np.array_equal(got[0], got[2])          # no stage reads time


@pytest.mark.parametrize("w,h", [(320, 180), (257, 129), (31, 9), (1, 1), (8, 8), (33, 1)])
def test_c2_fbm_showcase(game, oracle, w, h):
    check(game, oracle, C2_FBM, w, h, [0.0])


@pytest.mark.parametrize("warp", ["domain_warp(0.3, 40.0)", "domain_warp(0.2, 300.0)", "domain_warp(0.1, knob)", "domain_warp(0.1 + p.y, 3.0)",
                                  "domain_warp(-0.0, 0.0)", "domain_warp(1000000.0, 0.000001)", "domain_warp(0.2, time)",
                                  "domain_warp(0.2, 2.0 + sin(time) * 30.0)"],
                         ids=["fine-lattice", "leaves-table", "uniform-frequency", "pixel-argument", "zero", "extreme", "time-frequency",
                              "expression-frequency"])
def test_domain_warp_across_lattice_rows(game, oracle, warp):
    """A domain_warp that opens a layer keeps its x-interpolated lattice values from one row of a lane to the next and
    fetches again when the sample enters another lattice row: frames tall enough that every lane crosses many lattice
    rows, frequencies whose cells leave the hash table on part of the frame (the arithmetic path in between table
    rows), a frequency from a uniform, from the shader's time and from a per-frame expression, an argument that reads the pixel (which must take the plain
    form), bands that start inside a lane's run of rows, both wordings."""
    src = 'cinematic "w" { layer a { fn: %s | fbm(2.0, octaves: 2) | glow(1.5) } layer b { fn: %s | circle(0.4) | glow(2.0) | tint(0.2, 0.5, 1.0) } }' % (warp, warp)
    cin = game.compile(src).cinematics[0]
    w, h = 48, 531
    rows = cin.uniform_rows([0.0, 7.4], w, h, user={"knob": [7.0, 23.5]} if "knob" in warp else None)
    for sem in ("glsl", "wgsl"):
        animate = "p.y" in warp or "sin(" in warp
        r = cin.renderer(w, h, semantics=sem, animate=animate)
        got = r.render(rows)
        want = oracle.render_sequence(cin.blob, w, h, rows, False, semantics=sem, derived=cin.eval_derived(rows) if animate else None, animate=animate)
        assert np.array_equal(got, want), (warp, sem)
        band = r.render_band(rows[:1], 13, 211)[0]
        r.close()
        assert np.array_equal(band, want[0, 13:224])


@pytest.mark.parametrize("first", ["voronoi(40.0)", "voronoi(400.0)", "voronoi(knob)", "voronoi(time)", "voronoi(0.0)", "voronoi(3.0 + p.x)",
                                   "fbm(40.0, octaves: 3)", "fbm(300.0, octaves: 2)", "fbm(knob, octaves: 8, persistence: 0.7, lacunarity: 1.7)",
                                   "fbm(time, 1.0)", "fbm(2.0, knob)", "fbm(2.0 + sin(time), 4.0, 0.5, 2.0)", "fbm(0.000001, 6.0)"],
                         ids=["voronoi-fine", "voronoi-leaves-table", "voronoi-uniform", "voronoi-time", "voronoi-zero", "voronoi-pixel-argument",
                              "fbm-fine", "fbm-leaves-table", "fbm-uniform-8", "fbm-time", "fbm-uniform-octaves", "fbm-expression", "fbm-tiny"])
def test_noise_generators_across_lattice_rows(game, oracle, first):
    """voronoi and fbm that open a layer keep per-lane lattice state from row to row (the nine feature points' x terms;
    octave 0's x-interpolated values) and take their y halves from the workgroup's row values: tall frames that cross
    many lattice rows, scales that leave the tables, scales from uniforms / time / a per-frame expression, an octave count
    from a uniform and a pixel-reading argument (plain forms), bands, both wordings."""
    src = 'cinematic "n" { layer a { fn: %s | glow(1.5) | tint(1.0, 0.6, 0.3) } layer b { fn: %s | glow(0.7) } }' % (first, first)
    cin = game.compile(src).cinematics[0]
    w, h = 40, 1100
    rows = cin.uniform_rows([0.0, 9.3], w, h, user={"knob": [5.0, 31.5]} if "knob" in first else None)
    for sem in ("glsl", "wgsl"):
        animate = "p.x" in first or "sin(" in first
        r = cin.renderer(w, h, semantics=sem, animate=animate)
        got = r.render(rows)
        want = oracle.render_sequence(cin.blob, w, h, rows, False, semantics=sem, derived=cin.eval_derived(rows) if animate else None, animate=animate)
        assert np.array_equal(got, want), (first, sem)
        band = r.render_band(rows[1:], 7, 333)[0]
        r.close()
        assert np.array_equal(band, want[1, 7:340])


def test_more_layers_than_row_state(game, oracle):
    """Twelve layers that each open with voronoi: only the first two keep row state (a lane's registers are bounded), the
    rest take the plain form, and the frame is the same."""
    layers = "\n".join("  layer l%d { fn: voronoi(%d.0) | glow(1.0) | tint(0.1, 0.%d, 0.3) }" % (i, i + 2, i % 10) for i in range(12))
    check(game, oracle, 'cinematic "many" {\n' + layers + "\n}", 48, 600, [0.3])


def test_c2b_tutorial_noise(game, oracle):
    check(game, oracle, C2B_NOISE, 320, 180, [0.0, 3.7], user={"frost": [0.0, 0.6], "ember": [0.0, 0.9], "violet": [0.0, 0.3]})


def test_c3_showcase(game, oracle):
    check(game, oracle, C3_SHOWCASE, 320, 180, [0.0, 1.0 / 60, 7.3, 119.99, 120.5], audio=AUDIO)
    check(game, oracle, C3_SHOWCASE + " ", 33, 17, [0.25], audio=AUDIO)          # pixel centre on uv = 0: NaN path
    check(game, oracle, C3_SHOWCASE.replace("| grain(0.025)", "| grain(0.025) | tonemap(1.0)"), 160, 90, [0.5], audio=AUDIO)


def test_c4_memory_trails(game, oracle):
    n = 24
    bass = [0.5 + 0.5 * np.sin(2 * np.pi * 0.5 * k / 60) for k in range(n)]
    got = check(game, oracle, C4_TRAILS, 256, 144, [k / 60 for k in range(n)], audio={"bass": bass})
    assert not np.array_equal(got[0], got[-1])


def test_memory_state_carries_across_calls_and_resets(game, oracle):
    cin = game.compile(C4_TRAILS).cinematics[0]
    w, h, n = 128, 72, 10
    rows = cin.uniform_rows([k / 60 for k in range(n)], w, h, audio={"bass": [0.1 * k for k in range(n)]})
    want = oracle.render_sequence(cin.blob, w, h, rows, True)
    r = cin.renderer(w, h)
    a = r.render(rows[:4])
    b = r.render(rows[4:])                     # continues from frame 3
    assert np.array_equal(np.concatenate([a, b]), want)
    r.reset_memory()
    assert np.array_equal(r.render(rows[:2]), want[:2])
    r.set_ring(5)                              # a longer ring changes nothing
    assert np.array_equal(r.render(rows), want)
    r.close()


# ── every builtin, argument kind and compositing rule ───────────────────────────────────
@pytest.mark.parametrize("name", sorted(GALLERY))
def test_gallery(game, oracle, name):
    check(game, oracle, GALLERY[name], 96, 54, [2.5, 60.0, 119.5], audio={"bass": 0.7, "mid": 0.2, "treble": 0.9, "beat": 3.0})


# The same gallery in the reference's WebGPU wording of the builtins (wgsl.rs:268-540)
@pytest.mark.parametrize("name", sorted(GALLERY))
def test_gallery_wgsl_semantics(game, oracle, name):
    check(game, oracle, GALLERY[name], 96, 54, [2.5, 60.0, 119.5], audio={"bass": 0.7, "mid": 0.2, "treble": 0.9, "beat": 3.0},
          semantics="wgsl")


def test_wgsl_semantics_tables_and_sizes(game, oracle):
    """WebGPU voronoi reads the lattice-hash table and the WebGPU simplex the feature-point table: large scales push
    cells outside both tables, so table and arithmetic paths meet inside one frame; ragged size, four pixels per lane."""
    src = ('cinematic "w" { layer a { fn: voronoi(900.0) | glow(1.2) | tint(0.9, 0.5, 0.2) } '
           'layer b { fn: simplex(700.0) | glow(0.8) | tint(0.2, 0.6, 0.9) } '
           'layer c { fn: repeat(3.0) | star(5.0, 0.3, 0.15) | glow(2.0) | tonemap(1.4) | vignette(0.5, 0.6) } }')
    check(game, oracle, src, 257, 67, [0.25, 7.0], semantics="wgsl")
    check(game, oracle, C3_SHOWCASE, 128, 72, [0.5], audio={"bass": 0.8, "mid": 0.6, "treble": 0.4}, semantics="wgsl")


def test_page_locked_readback_equals_pageable(game):
    cin = game.compile(C1_HELLO).cinematics[0]
    rows = cin.uniform_rows([0.0, 0.5, 1.0], 200, 120)
    r = cin.renderer(200, 120)
    a, b = r.render(rows), r.render(rows, pinned=True)
    r.close()
    assert np.array_equal(a, b)


def test_arc_timeline_drives_the_frames(game, oracle):
    """Rows animated by the cinematic's own arc timeline: the frames change over time and stay bit-exact."""
    src = ('cinematic "a" { layer { fn: circle(radius) | glow(brightness) | tint(0.2, 0.7, 0.9)\n radius: 0.05\n brightness: 0.5 } '
           'arc { radius: 0.05 -> 0.4 over 2s expo_out\n brightness: 0.5 -> 3.0 over 1s smooth } }')
    cin = game.compile(src).cinematics[0]
    rows = cin.uniform_rows([0.0, 0.5, 1.0, 2.5, 4.0], 96, 54, arcs=True)
    r = cin.renderer(96, 54)
    got = r.render(rows)
    r.close()
    want = oracle.render_sequence(cin.blob, 96, 54, rows, False)
    assert np.array_equal(got, want)
    assert not np.array_equal(got[0], got[2]) and not np.array_equal(got[2], got[4])


def test_gallery_memory_sequence(game, oracle):
    check(game, oracle, GALLERY["memory_multi"], 64, 36, [k / 30 for k in range(6)])


# ── ragged sizes, bands, ring wrap ──────────────────────────────────────────────────────
def test_band_rendering(game, oracle):
    cin = game.compile(C2B_NOISE).cinematics[0]
    w, h = 200, 117
    rows = cin.uniform_rows([0.75], w, h)
    want = oracle.render(cin.blob, w, h, rows[0])
    r = cin.renderer(w, h)
    full = r.render(rows)[0]
    assert np.array_equal(full, want)
    parts = [r.render_band(rows, r0, n)[0] for r0, n in ((0, 8), (8, 1), (9, 64), (73, 44))]
    assert np.array_equal(np.concatenate(parts), want)
    with pytest.raises(game.GameError):
        r.render_band(rows, 100, 18)           # runs past the last row
    r.close()


def test_ring_wraps_and_many_frames(game, oracle):
    cin = game.compile(GALLERY["waves"]).cinematics[0]
    w, h, n = 64, 36, 11
    rows = cin.uniform_rows([k / 7 for k in range(n)], w, h)
    r = cin.renderer(w, h)
    r.set_ring(3)
    got = r.render(rows)                       # 11 frames through a 3-slot ring
    r.close()
    want = oracle.render_sequence(cin.blob, w, h, rows, False)
    assert np.array_equal(got, want)


def test_bad_arguments_are_rejected(game):
    cin = game.compile(C1_HELLO).cinematics[0]
    with pytest.raises(game.GameError) as ei:
        cin.renderer(0, 16)
    assert ei.value.code == 80
    with pytest.raises(game.GameError):
        cin.renderer(16, 16, device=99)
    r = cin.renderer(16, 16)
    with pytest.raises(ValueError):
        r.render(np.zeros((1, 3), np.float32))
    with pytest.raises(game.GameError) as ei:
        r.particles_seed(16)
    assert ei.value.code == 83                 # no gravity block
    r.close()


# ── full benchmark sizes: bands against the oracle + size-independent properties ────────
def test_c2_4k_bands_and_properties(game, oracle):
    cin = game.compile(C2_FBM).cinematics[0]
    w, h = 3840, 2160
    rows = cin.uniform_rows([0.0, 1.0, 2.0], w, h)
    r = cin.renderer(w, h)
    frames = r.render(rows)
    assert np.array_equal(frames[0], frames[1]) and np.array_equal(frames[0], frames[2])    # time-invariant cinematic
    again = r.render(rows[:1])[0]
    assert np.array_equal(again, frames[0])                                                 # deterministic
    for r0, n in ((0, 2), (1079, 2), (2158, 2)):
        want = oracle.render(cin.blob, w, h, rows[0], row0=r0, rows=n)
        assert np.array_equal(frames[0][r0:r0 + n], want)
    band = r.render_band(rows[:1], 1000, 160)[0]
    assert np.array_equal(band, frames[0][1000:1160])
    r.close()


def test_c3_and_c3b_4k_bands_and_properties(game, oracle):
    """The showcase chain and the same chain with the tonemap stage appended (workloads c3, c3b) at 3840 x 2160: bands
    against the oracle; the tonemap maps every channel c to c / (c + 1) rounded, so c3b is never brighter than c3."""
    W = game.workloads
    w, h = 3840, 2160
    frames = {}
    for key in ("c3", "c3b"):
        cin = game.compile(W.CONFIGS[key]["src"]).cinematics[0]
        rows = W.frame_rows(cin, key, [0, 37], w, h)
        r = cin.renderer(w, h)
        got = r.render(rows)
        assert not np.array_equal(got[0], got[1])                                           # grain reads the time
        for r0, n in ((0, 2), (1079, 2), (2158, 2)):
            assert np.array_equal(got[1][r0:r0 + n], oracle.render(cin.blob, w, h, rows[1], row0=r0, rows=n))
        assert np.array_equal(r.render_band(rows[1:], 1000, 96)[0], got[1][1000:1096])
        r.close()
        frames[key] = got[1]
    assert (frames["c3b"][..., :3] <= frames["c3"][..., :3]).all() and (frames["c3b"][..., :3] < frames["c3"][..., :3]).any()
    assert (frames["c3b"][..., 3] == 255).all()


def test_c4_8k_feedback_recurrence(game, oracle):
    cin = game.compile(C4_TRAILS).cinematics[0]
    w, h, n = 7680, 4320, 3
    bass = [0.5 + 0.5 * np.sin(2 * np.pi * 0.5 * k / 60) for k in range(n)]
    rows = cin.uniform_rows([k / 60 for k in range(n)], w, h, audio={"bass": bass})
    r = cin.renderer(w, h)
    r0, nb = 2100, 4                                  # rows through the glowing core
    bands = r.render_band(rows, r0, nb)               # only the band is ever computed or copied
    r.close()
    prev = None
    for k in range(n):
        want = oracle.render(cin.blob, w, h, rows[k], prev=prev, row0=r0, rows=nb)
        assert np.array_equal(bands[k], want), k
        prev = want


# ── particles ───────────────────────────────────────────────────────────────────────────
def _same_bits_or_nan(a, b):
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


@pytest.mark.parametrize("bounds", ["none", "reflect"])
def test_gravity_wild_positions(game, oracle, bounds):
    """Coincident particles, distances below the 0.001 floor, huge / infinite / NaN coordinates: the tile
    loops that keep the IEEE results for an infinite squared distance must agree with the oracle."""
    src = 'cinematic "t" { gravity { rule: 0.5 / (dist + 0.25), damping: 0.9, bounds: %s } layer s { circle() | glow() } }' % bounds
    cin = game.compile(src).cinematics[0]
    rng = np.random.default_rng(21)
    n = 700                                      # three tiles: the wild values sit in the middle one
    pos = (rng.random((n, 2), dtype=np.float32) * 2 - 1).astype(np.float32)
    pos[300] = pos[10]                           # coincident pair
    pos[301] = pos[11] + np.float32(1e-5)        # closer than the floor
    pos[302] = (1e30, -1e30)
    pos[303] = (3e38, 3e38)                      # squared distance overflows
    pos[304] = (np.inf, 0.25)
    pos[305] = (np.nan, 0.5)
    pos[306] = (2.0 ** 61, 0.0)                  # beyond the 2^60 guard, finite squared distance
    p0 = np.concatenate([pos, np.zeros_like(pos)], axis=1)
    r = cin.renderer(16, 16)
    r.particles_init(pos)
    r.particles_step(1 / 60, 2)
    got_pos, got_vel = r.particles_read()
    want, _ = oracle.gravity(cin.blob, p0, 1 / 60, 2)
    assert _same_bits_or_nan(got_pos, want[:, :2]) and _same_bits_or_nan(got_vel, want[:, 2:])
    r.close()


def test_gravity_default_law_bit_exact(game, oracle):
    cin = game.compile(C5_GRAVITY).cinematics[0]
    n = 4096
    p0 = oracle.seed_particles(n)
    r = cin.renderer(16, 16)
    r.particles_seed(n)
    pos, vel = r.particles_read()
    assert np.array_equal(pos, p0[:, :2]) and np.all(vel == 0)          # same SplitMix64 initial state
    r.particles_step(1 / 60, 20)
    pos, vel = r.particles_read()
    want, t = oracle.gravity(cin.blob, p0, 1 / 60, 20)
    assert np.array_equal(pos, want[:, :2]) and np.array_equal(vel, want[:, 2:])
    assert r.particles_time() == t
    r.close()


def test_gravity_100_steps_tolerance(game, oracle):
    """The stated bar: positions within 1e-4 relative after 100 steps (met exactly here)."""
    cin = game.compile(C5_GRAVITY).cinematics[0]
    n = 2048
    p0 = oracle.seed_particles(n)
    r = cin.renderer(16, 16)
    r.particles_init(p0[:, :2], p0[:, 2:])
    r.particles_step(1 / 60, 100)
    pos, _ = r.particles_read()
    r.close()
    want, _ = oracle.gravity(cin.blob, p0, 1 / 60, 100)
    rel = np.abs(pos - want[:, :2]) / np.maximum(np.abs(want[:, :2]), 1e-3)
    assert rel.max() <= 1e-4
    assert np.array_equal(pos, want[:, :2])


@pytest.mark.parametrize("law,bounds", [
    ("1.0 / (dist * dist + 0.05)", "wrap"),
    ("-(0.2 / dist) * (dist < 0.3) + 0.1 * (dist > 0.3)", "reflect"),
    ("sin(dist * 6.0 + time) * 0.5", "none"),
    ("dist ^ 1.5 - max(0.2, min(dist, 0.7)) + abs(cos(time)) * sqrt(dist)", "wrap"),
    ("exp(-dist * 3.0) + log(1.0 + dist) - clamp(dist, 0.1, 0.5) + mix(0.1, 0.2, dist) + smoothstep(0.1, 0.9, dist) * step(0.5, dist) + floor(dist * 4.0) * 0.1 + fract(dist * 3.0) * sign(dist - 0.5)", "reflect"),
])
def test_gravity_force_laws(game, oracle, law, bounds):
    src = 'cinematic "t" { gravity { rule: %s, damping: 0.98, bounds: %s } layer s { circle() | glow() } }' % (law, bounds)
    cin = game.compile(src).cinematics[0]
    assert not (cin.flags & 256)
    n = 777                                     # not a multiple of the 256-particle tile
    p0 = oracle.seed_particles(n, seed=12345)
    r = cin.renderer(16, 16)
    r.particles_init(p0[:, :2], p0[:, 2:])
    r.particles_step(1 / 60, 6)
    pos, vel = r.particles_read()
    r.close()
    want, _ = oracle.gravity(cin.blob, p0, 1 / 60, 6)
    assert np.array_equal(pos, want[:, :2]) and np.array_equal(vel, want[:, 2:])


def test_gravity_index_sharding(game, oracle):
    """Two index ranges stepped separately (as two ranks would) equal the whole-set step."""
    cin = game.compile(C5_GRAVITY).cinematics[0]
    n = 1000
    p0 = oracle.seed_particles(n, seed=99)
    want, _ = oracle.gravity(cin.blob, p0, 1 / 60, 3)
    a = cin.renderer(16, 16)
    b = cin.renderer(16, 16)
    pa, va = p0[:, :2].copy(), p0[:, 2:].copy()
    for _ in range(3):
        a.particles_init(pa, va); b.particles_init(pa, va)
        a.particles_step(1 / 60, 1, first=0, count=600)
        b.particles_step(1 / 60, 1, first=600, count=400)
        qa, wa = a.particles_read(); qb, wb = b.particles_read()
        pa = np.concatenate([qa[:600], qb[600:]]); va = np.concatenate([wa[:600], wb[600:]])   # the all-gather
    a.close(); b.close()
    assert np.array_equal(pa, want[:, :2]) and np.array_equal(va, want[:, 2:])


@pytest.mark.parametrize("ppl", ["1", "2"])
def test_gravity_particles_per_lane_builds(game, oracle, ppl, monkeypatch):
    """The register-tiled build of the step kernel (two particles per lane) walks every particle's pairs in the
    same order as the one-particle build: same bits for whole sets, ragged counts and index ranges."""
    monkeypatch.setenv("GAME_HIP_GRAV_PPL", ppl)
    cin = game.compile(C5_GRAVITY).cinematics[0]
    for n, first, count in ((777, 0, 777), (2048, 0, 2048), (1500, 300, 900), (1025, 1024, 1)):
        p0 = oracle.seed_particles(n, seed=n)
        r = cin.renderer(16, 16)
        r.particles_init(p0[:, :2], p0[:, 2:])
        r.particles_step(1 / 60, 1, first=first, count=count)
        pos, vel = r.particles_read()
        r.close()
        want, _ = oracle.gravity(cin.blob, p0, 1 / 60, 1)
        # the buffer the step wrote holds new values for the range and the initial positions elsewhere
        want[:first] = p0[:first]
        want[first + count:] = p0[first + count:]
        assert np.array_equal(pos, want[:, :2]) and np.array_equal(vel, want[:, 2:]), (n, first, count)
    src = 'cinematic "t" { gravity { rule: 0.02 / (dist + 0.1) * (dist > 0.05), damping: 0.9, bounds: wrap } layer s { circle() | glow() } }'
    cin = game.compile(src).cinematics[0]
    p0 = oracle.seed_particles(1300, seed=4)
    p0[3, 0] = np.nan                      # one wild particle: its tiles take the IEEE build of the loop
    p0[700, 1] = 3e38
    r = cin.renderer(16, 16)
    r.particles_init(p0[:, :2], p0[:, 2:])
    r.particles_step(1 / 60, 3)
    pos, vel = r.particles_read()
    r.close()
    want, _ = oracle.gravity(cin.blob, p0, 1 / 60, 3)
    assert _same_bits_or_nan(pos, want[:, :2]) and _same_bits_or_nan(vel, want[:, 2:])


def test_gravity_65536_particles_100_steps(game):
    """The particle workload at its parity size against the committed oracle run (tests/golden/make_golden.py
    --gravity-65536): the default build bit for bit, the fast build (root and reciprocal from one hardware
    reciprocal-square-root) within the stated tolerance, 1e-4 relative position after 100 steps."""
    import os
    want = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gravity_65536_100.npy"))
    cin = game.compile(C5_GRAVITY).cinematics[0]
    r = cin.renderer(16, 16)
    r.particles_seed(65536)
    r.particles_step(1 / 60, 100)
    pos, vel = r.particles_read()
    r.close()
    assert np.array_equal(pos, want[:, :2]) and np.array_equal(vel, want[:, 2:])


@pytest.mark.timeout(900)
def test_gravity_16m_particles_sampled_ranges(game, oracle):
    """The particle workload at its full size, 2^24 particles: a whole step costs 2.8e14 pair evaluations, so parity is
    checked on index ranges - the step of a few hundred particles against all 2^24 positions on the GPU
    (game_hip_particles_step with first / count, what one rank of an index-sharded run computes) against the oracle
    evaluating just those particles against all positions, bit for bit. Every range starts from the state the GPU holds
    at that moment (the previous ranges' particles have moved), so later ranges also read positions earlier steps
    wrote. Ranges: the start, one straddling a 2^20 launch boundary, a ragged one, the end."""
    n = 1 << 24
    cin = game.compile(C5_GRAVITY).cinematics[0]
    r = cin.renderer(16, 16)
    r.particles_seed(n)
    pos, vel = r.particles_read()
    state = np.concatenate([pos, vel], axis=1)
    del pos, vel
    assert np.array_equal(state[:4096], oracle.seed_particles(4096))
    t = 0.0
    for first, count in ((0, 256), ((1 << 20) * 7 - 100, 256), (12345677, 300), (n - 256, 256)):
        r.particles_step(1 / 60, 1, first=first, count=count)
        t = np.float32(np.float32(t) + np.float32(1 / 60))
        want = oracle.gravity_range(cin.blob, state, 1 / 60, float(t), first, count, threads=16)
        pos, vel = r.particles_read()
        got = np.concatenate([pos[first:first + count], vel[first:first + count]], axis=1)
        assert np.array_equal(got, want), (first, count)
        # outside the range the buffer the step wrote holds what the other buffer held: the state before this step,
        # except for ranges stepped one call earlier (double buffering) - so only the range itself is taken over
        state[first:first + count] = got
        r.particles_init(state[:, :2], state[:, 2:])
        r.particles_set_time(float(t))
    r.close()


def test_gravity_fast_build_tolerance(game, oracle):
    """The fast build (root and reciprocal of the pair distance from one hardware reciprocal-square-root, about 1 ulp
    each) is another arithmetic and is NOT a parity mode. What it holds: one step of the particle workload within 1e-4
    relative of the default build, and a smooth force law within 1e-5 after 10 steps. The workload itself (rule 1.0:
    65 536 unit forces per particle) throws 93 % of the particles against the walls in its first step and is chaotic
    from there on, so over 100 steps any arithmetic that is not bit-identical ends with a few percent of the
    coordinates on another wall (2.4 % measured); the test states that as a bound on the share, not as agreement."""
    import os
    want100 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gravity_65536_100.npy"))
    cin = game.compile(C5_GRAVITY).cinematics[0]
    n = 65536
    a, f = cin.renderer(16, 16), cin.renderer(16, 16, gravity_fast=True)
    a.particles_seed(n); f.particles_seed(n)
    a.particles_step(1 / 60, 1); f.particles_step(1 / 60, 1)
    pa, _ = a.particles_read(); pf, _ = f.particles_read()
    rel = np.abs(pf - pa) / np.maximum(np.abs(pa), 1e-3)
    assert rel.max() <= 1e-4, rel.max()
    assert not np.array_equal(pf, pa)                 # it IS another arithmetic; equality would mean the flag did nothing
    f.particles_step(1 / 60, 99)
    pf, _ = f.particles_read()
    rel = np.abs(pf - want100[:, :2]) / np.maximum(np.abs(want100[:, :2]), 1e-3)
    assert (rel > 1e-4).mean() < 0.05, (rel > 1e-4).mean()
    a.close(); f.close()
    soft = 'cinematic "t" { gravity { rule: 0.00002 / (dist * dist + 0.01), damping: 0.995, bounds: reflect } layer s { circle() | glow() } }'
    cin = game.compile(soft).cinematics[0]
    p0 = oracle.seed_particles(8192)
    f = cin.renderer(16, 16, gravity_fast=True)
    f.particles_init(p0[:, :2], p0[:, 2:])
    f.particles_step(1 / 60, 10)
    pf, _ = f.particles_read()
    f.close()
    want, _ = oracle.gravity(cin.blob, p0, 1 / 60, 10)
    rel = np.abs(pf - want[:, :2]) / np.maximum(np.abs(want[:, :2]), 1e-3)
    assert rel.max() <= 1e-5, rel.max()


@pytest.mark.gpu
def test_fbm_table_range_test_edges(game, oracle):
    """fbm decides table vs arithmetic with one norm test for all octaves: frames that straddle the table's edge,
    and frequencies that are negative, shrinking, zero, infinite or NaN, must still equal the oracle (and must
    never index the table with a cell outside it)."""
    src = 'cinematic "e" { layer a { fn: translate(tx, 0.0) | fbm(sc, 5.0, pe, la) | glow(1.5) | tint(0.7, 0.9, 1.0) } }'
    cin = game.compile(src).cinematics[0]
    w, h = 192, 108
    nan, inf = float("nan"), float("inf")
    cases = [  # sc, pe, la, tx
        (3.0, 0.5, 2.0, 0.0), (30.0, 0.5, 2.0, 0.0), (17.9, 0.5, 2.0, 0.0), (18.1, 0.5, 2.0, 0.0), (200.0, 0.5, 2.0, 0.0),
        (3.0, 0.5, -2.0, 0.0), (300.0, 0.5, 0.5, 0.0), (500.0, 0.5, 0.25, 1.0), (3.0, 0.5, 0.0, 0.0), (3.0, 0.5, inf, 0.0),
        (3.0, 0.5, nan, 0.0), (3.0, nan, 2.0, 0.0), (nan, 0.5, 2.0, 0.0), (inf, 0.5, 2.0, 0.0), (3.0, 0.5, 2.0, 1e30),
        (511.0, 0.5, 1.0, 0.0), (287.0, 0.5, 1.0, 0.0), (3.0, 0.5, 1e-30, 0.0), (-31.0, 0.5, 2.0, 0.3),
    ]
    rows = cin.uniform_rows([0.0] * len(cases), w, h)
    names = [u["name"] for u in cin.ir["uniforms"]]
    for k, (sc, pe, la, tx) in enumerate(cases):
        for name, v in (("sc", sc), ("pe", pe), ("la", la), ("tx", tx)):
            rows[k, 10 + names.index(name)] = v
    r = cin.renderer(w, h)
    got = r.render(rows)
    r.close()
    want = oracle.render_sequence(cin.blob, w, h, rows, False)
    for k in range(len(cases)):
        assert np.array_equal(got[k], want[k]), f"case {cases[k]}"


@pytest.mark.parametrize("w,h", [(65536, 2), (4, 65535), (65535, 1), (1, 4097)])
def test_extreme_frame_shapes(game, oracle, w, h):
    """The widest and tallest frames the ABI accepts (width <= 65536, height <= 65535), with and without feedback."""
    for src in (C2_FBM, C4_TRAILS):
        cin = game.compile(src).cinematics[0]
        rows = cin.uniform_rows([0.0, 0.25], w, h, audio={"bass": 0.4})
        r = cin.renderer(w, h)
        got = r.render(rows)
        r.close()
        want = oracle.render_sequence(cin.blob, w, h, rows, cin.uses_memory)
        assert np.array_equal(got, want)


def test_checkpoint_resume_of_feedback_and_particles(game, oracle):
    """A run cut in two and resumed in a fresh context from what the first half returned equals the uncut run."""
    cin = game.compile(C4_TRAILS).cinematics[0]
    w, h = 96, 54
    rows = cin.uniform_rows([k / 30 for k in range(8)], w, h, audio={"bass": 0.6})
    r = cin.renderer(w, h)
    whole = r.render(rows)
    r.close()
    r = cin.renderer(w, h)
    first = r.render(rows[:5])
    r.close()
    r = cin.renderer(w, h)
    r.set_memory(first[-1])
    second = r.render(rows[5:])
    # a band restored on top of an existing state replaces just those rows
    r.set_memory(whole[3][10:20], row0=10)
    mixed_prev = second[-1].copy()
    mixed_prev[10:20] = whole[3][10:20]
    nxt = r.render(rows[7:8])
    r.close()
    assert np.array_equal(np.concatenate([first, second]), whole)
    assert np.array_equal(nxt[0], oracle.render(cin.blob, w, h, rows[7], prev=mixed_prev))

    src = 'cinematic "t" { gravity { rule: sin(time * 40.0) / (dist + 0.2), damping: 0.99, bounds: wrap } layer s { circle() | glow() } }'
    cin = game.compile(src).cinematics[0]
    p0 = oracle.seed_particles(600, seed=5)
    r = cin.renderer(16, 16)
    r.particles_init(p0[:, :2], p0[:, 2:])
    r.particles_step(1 / 60, 7)
    want_pos, want_vel = r.particles_read()
    r.close()
    r = cin.renderer(16, 16)
    r.particles_init(p0[:, :2], p0[:, 2:])
    r.particles_step(1 / 60, 3)
    pos, vel = r.particles_read()
    t = r.particles_time()
    r.close()
    r = cin.renderer(16, 16)
    r.particles_init(pos, vel)
    r.particles_set_time(t)
    r.particles_step(1 / 60, 4)
    got_pos, got_vel = r.particles_read()
    r.close()
    assert np.array_equal(got_pos, want_pos) and np.array_equal(got_vel, want_vel)


def test_timing_section_sums_queued_calls(game):
    cin = game.compile(C2_FBM).cinematics[0]
    r = cin.renderer(320, 180)
    r.set_ring(4)
    rows = cin.uniform_rows([0.0, 0.1, 0.2, 0.3], 320, 180)
    r.render(rows, readback=False)
    one, n1 = r.last_kernel_ms()
    r.timing_begin()
    for _ in range(5):
        r.render(rows, readback=False)
    total, n = r.timing_end()
    assert n == 5 * n1 and total > 0 and 2.0 * one < total < 20.0 * one + 1.0
    with pytest.raises(game.GameError):
        r.timing_end()
    r.render(rows, readback=False)           # outside a section the single-call query still works
    assert r.last_kernel_ms()[1] == n1
    r.close()
