"""The resolved IR as data: game_ir_from_blob takes the flat blob a host's own compiler front would write (layout in
include/game_hip.h) in place of .game text. The loader must give back the IR the front end lowered (same blob bytes, same
generated kernels, same derived values) and refuse every blob that names something that does not exist."""
import random
import struct

import numpy as np
import pytest

from conftest import C1_HELLO, C2_FBM, C2B_NOISE, C3_SHOWCASE, C4_TRAILS, C5_GRAVITY
from gallery import GALLERY
from test_fuzz_parity import make_cinematic

HEADER = 112   # magic, total, 4 counts, flags, 3 booleans, damping, bounds, name[64]
LAYER, STAGE, UNIFORM, LAWOP = 32, 80, 64, 16

EXPRESSIVE = '''cinematic "e" {
  layer a memory: 0.9 { fn: translate(sin(time) * 0.2, knob) | circle(0.2 + 0.1 * audio.bass) | glow(length(p) + 1.0) | tint(1.0, 0.5, dial) }
  lens { post: vignette(0.4, 0.8) | grain(mod(time, 0.3) * 0.1) }
}
project dome (fisheye: 150) { source: e }'''


def sources(game):
    out = [C1_HELLO, C2_FBM, C2B_NOISE, C3_SHOWCASE, C4_TRAILS, C5_GRAVITY, EXPRESSIVE] + list(GALLERY.values())
    # a memory clause on a layer that only declares parameters still marks the cinematic as using memory
    out.append('cinematic "m" { layer cfg memory: 0.9 { speed: 1.0 } layer a { fn: circle(0.2) | glow(2.0) } }')
    for seed in range(40):
        out.append(make_cinematic(game, random.Random(1000 + seed), expressions=seed % 2 == 1)[0])
    return out


def test_blob_round_trip(game):
    n = 0
    for src in sources(game):
        for cin in game.compile(src).cinematics:
            blob = cin.blob
            back = game.load_blob(blob).cinematics[0]
            assert back.blob == blob
            assert back.name == cin.name and back.uniforms == [(k, float(np.float32(v))) for k, v in cin.uniforms]
            assert (back.uses_memory, back.has_gravity, back.flags, back.frame_floats) == (cin.uses_memory, cin.has_gravity, cin.flags, cin.frame_floats)
            assert (back.derived_count, back.pixel_expr_count, back.project_mode) == (cin.derived_count, cin.pixel_expr_count, cin.project_mode)
            rows = cin.uniform_rows([0.0, 0.7, 130.5], 64, 36, audio={"bass": 0.4, "beat": 1.0})
            assert np.array_equal(back.eval_derived(rows), cin.eval_derived(rows), equal_nan=True)
            n += 1
    assert n >= 80


@pytest.mark.parametrize("flags", [dict(), dict(semantics="wgsl"), dict(animate=True, lens_post=True, project=True)], ids=["glsl", "wgsl", "animate"])
def test_loaded_ir_generates_the_same_kernels(game, tmp_path, flags):
    for i, src in enumerate([C2_FBM, C3_SHOWCASE, C4_TRAILS, C5_GRAVITY, EXPRESSIVE]):
        cin = game.compile(src).cinematics[0]
        back = game.load_blob(cin.blob).cinematics[0]
        assert back.compile_only(**flags) == cin.compile_only(**flags) > 0
        if not flags:
            cin.dump_code(str(tmp_path / f"a{i}"))
            back.dump_code(str(tmp_path / f"b{i}"))
            for kind in ("frame", "gravity"):
                a, b = tmp_path / f"a{i}_{kind}.hip", tmp_path / f"b{i}_{kind}.hip"
                assert a.exists() == b.exists()
                if a.exists():
                    assert a.read_text() == b.read_text()


def put(blob, offset, fmt, *values):
    b = bytearray(blob)
    struct.pack_into(fmt, b, offset, *values)
    return bytes(b)


def resized(blob, new_len):
    b = bytearray(blob[:new_len]) + bytes(max(0, new_len - len(blob)))
    struct.pack_into("<I", b, 4, len(b))
    return bytes(b)


def test_damaged_blobs_are_refused(game):
    cin = game.compile(EXPRESSIVE).cinematics[0]
    blob = cin.blob
    n_layers, n_stages, n_uniforms, n_law = struct.unpack_from("<4I", blob, 8)
    stage0 = HEADER + n_layers * LAYER
    uniform0 = stage0 + n_stages * STAGE
    pix0 = uniform0 + n_uniforms * UNIFORM + n_law * LAWOP
    assert struct.unpack_from("<I", blob, pix0)[0] == cin.pixel_expr_count >= 1
    cases = {
        "magic": put(blob, 0, "<I", 0x31524948),
        "length field": put(blob, 4, "<I", len(blob) + 4),
        "truncated": blob[:-4],
        "truncated, length fixed": resized(blob, len(blob) - 4),
        "padded, length fixed": resized(blob, len(blob) + 16),
        "layer count": put(blob, 8, "<I", 5000),
        "stage count": put(blob, 12, "<I", n_stages + 1),
        "bounds mode": put(blob, 44, "<I", 3),
        "damping NaN": put(blob, 40, "<I", 0x7fc00000),
        "name unterminated": blob[:48] + b"x" * 64 + blob[112:],
        "layer stage range": put(blob, HEADER + 8, "<I", n_stages + 1),
        "layer end state": put(blob, HEADER + 20, "<I", 3),
        "decay inf": put(blob, HEADER + 16, "<I", 0x7f800000),
        "uses_memory flag": put(blob, 32, "<I", 0),
        "builtin id": put(blob, stage0, "<I", 38),
        "argument count": put(blob, stage0 + 4, "<I", 5),
        "argument kind": put(blob, stage0 + 16, "<I", 3),
        "user slot": put(blob, stage0 + 16, "<II", 1, n_uniforms),
        "builtin slot": put(blob, stage0 + 16, "<II", 2, 10),
        "literal NaN": put(blob, stage0 + 24, "<I", 0xffc00000),
        "derived index": put(blob, stage0 + 28, "<I", cin.derived_count + 1),
        "pixel index": put(blob, stage0 + 28, "<I", (cin.pixel_expr_count + 1) << 16),
        "uniform name unterminated": blob[:uniform0 + 4] + b"y" * 60 + blob[uniform0 + 64:],
        "pixel expression count": put(blob, pix0, "<I", 257),
        "pixel expression length": put(blob, pix0 + 4, "<I", 129),
        "unknown operation": put(blob, pix0 + 8, "<I", 19),
        "stack underflow": put(blob, pix0 + 8, "<I", 3),                 # the program now starts with `+`
        "function arity": put(blob, pix0 + 8, "<IfI", 11, 0.0, 12),       # one-argument call of max
        "frame slot": put(blob, pix0 + 8, "<IfI", 14, 0.0, cin.frame_floats),
        "dist outside a force law": put(blob, pix0 + 8, "<I", 1),
        # the last operation of the first pixel expression becomes a push: two values are left where one is the result
        # (the evaluators would disagree on which of them that is)
        "two values left": put(blob, pix0 + 8 + 16 * (struct.unpack_from("<I", blob, pix0 + 4)[0] - 1), "<IfI", 0, 1.0, 0),
    }
    assert struct.unpack_from("<I", blob, pix0 + 8 + 16 * (struct.unpack_from("<I", blob, pix0 + 4)[0] - 1))[0] in (3, 4, 5, 6)   # a binary operator
    for what, bad in cases.items():
        with pytest.raises(game.GameError) as ei:
            game.load_blob(bad)
        assert ei.value.code == 80 and "game_ir_from_blob" in str(ei.value), what   # GAME_E_ARGUMENT
    with pytest.raises(game.GameError):
        game.load_blob(b"")
    # the oracle's own validator refuses the same program
    from conftest import Oracle
    with pytest.raises(AssertionError):
        Oracle().render(cases["two values left"], 8, 8, cin.uniform_rows([0.0], 8, 8)[0])
    # the force law of a gravity block: no frame-row reads, no pixel reads
    g = game.compile(C5_GRAVITY.replace("damping: 0.995,", "damping: 0.995,\n    rule: 1.0 / (dist * dist + 0.01),")).cinematics[0]
    gb = g.blob
    gl, gs, gu, gn = struct.unpack_from("<4I", gb, 8)
    law0 = HEADER + gl * LAYER + gs * STAGE + gu * UNIFORM
    assert gn >= 3 and game.load_blob(gb).cinematics[0].blob == gb
    for op in (14, 15):
        with pytest.raises(game.GameError):
            game.load_blob(put(gb, law0, "<IfI", op, 0.0, 0))


def test_hand_written_blob(game):
    """A blob assembled field by field from the layout, as a foreign host would: circle(0.25) | glow(2.0) | tint(u, 0.5, time)."""
    def arg(kind=0, slot=0, lit=0.0):
        return struct.pack("<IIfI", kind, slot, lit, 0)
    ids = {d["name"]: i for i, d in enumerate(game.list_builtins())}
    stages = b"".join(struct.pack("<4I", ids[name], len(args), 0, 0) + b"".join(args) + arg() * (4 - len(args))
                      for name, args in (("circle", [arg(lit=0.25)]), ("glow", [arg(lit=2.0)]), ("tint", [arg(1, 0), arg(lit=0.5), arg(2, 0)])))
    body = (struct.pack("<8I", 1, 0, 3, 0, 0, 2, 0, 0) + stages + struct.pack("<f60s", 0.75, b"u") +
            struct.pack("<I", 0) + struct.pack("<IfII", 0, 180.0, 60, 0) + struct.pack("<I", 0))
    head = struct.pack("<10IfI64s", 0x31524947, HEADER + len(body), 1, 3, 1, 0, 0, 0, 0, 0, 0.99, 0, b"by-hand")
    cin = game.load_blob(head + body).cinematics[0]
    ref = game.compile('cinematic "by-hand" { layer l { fn: circle(0.25) | glow(2.0) | tint(u, 0.5, time) } }').cinematics[0]
    assert cin.name == "by-hand" and cin.uniforms == [("u", 0.75)] and cin.frame_floats == 11
    assert cin.compile_only() > 0
    assert [s["id"] for s in cin.ir["layers"][0]["stages"]] == [s["id"] for s in ref.ir["layers"][0]["stages"]]


# ── on the GPU: a context made from a loaded blob renders what the compiled cinematic renders ─────────────────
@pytest.mark.gpu
def test_loaded_ir_renders_like_the_compiled_one(game, oracle):
    for src, w, h in ((C2_FBM, 96, 54), (C4_TRAILS, 64, 36), (EXPRESSIVE, 80, 45)):
        cin = game.compile(src).cinematics[0]
        back = game.load_blob(cin.blob).cinematics[0]
        rows = cin.uniform_rows([k / 7 for k in range(4)], w, h, audio={"bass": 0.6}, user={n: 0.3 for n, _ in cin.uniforms})
        flags = dict(animate=True, lens_post=True, project=True) if src is EXPRESSIVE else {}
        a, b = cin.renderer(w, h, **flags), back.renderer(w, h, **flags)
        assert a.kernel_hash == b.kernel_hash
        got_a, got_b = a.render(rows), b.render(rows)
        a.close()
        b.close()
        assert np.array_equal(got_a, got_b)
        want = oracle.render_sequence(cin.blob, w, h, rows, cin.uses_memory, derived=cin.eval_derived(rows) if flags else None,
                                      lens_post=bool(flags), project=bool(flags))
        assert np.array_equal(got_b, want)
    cin = game.compile(C5_GRAVITY).cinematics[0]
    back = game.load_blob(cin.blob).cinematics[0]
    p0 = oracle.seed_particles(1500)
    r = back.renderer(16, 16)
    r.particles_init(p0[:, :2], p0[:, 2:])
    r.particles_step(1 / 60, 5)
    pos, vel = r.particles_read()
    r.close()
    want, _ = oracle.gravity(cin.blob, p0, 1 / 60, 5)
    assert np.array_equal(pos, want[:, :2]) and np.array_equal(vel, want[:, 2:])


def corpus():
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "corpus_blobs.npz")
    with np.load(path) as z:
        return {k: z[k].tobytes() for k in z.files}


def test_corpus_blobs_load(game, oracle):
    """CPU: every blob of the fixture (tests/golden/make_corpus_blobs.py) loads, serialises back to the same bytes and is
    accepted by the oracle's reader."""
    blobs = corpus()
    assert len(blobs) >= 80
    for name, blob in blobs.items():
        cin = game.load_blob(blob).cinematics[0]
        assert cin.blob == blob, name
        rows = cin.uniform_rows([0.5], 16, 9)
        oracle.render_sequence(blob, 16, 9, rows, cin.uses_memory, derived=cin.eval_derived(rows), lens_post=True, project=True, threads=1)


@pytest.mark.gpu
@pytest.mark.parametrize("part", range(4))
def test_reference_corpus_renders_like_the_oracle(game, oracle, part):
    """Every cinematic of the reference's .game corpus (examples/, presets/, game-compiler/examples/; IR blobs made
    here by tests/golden/make_corpus_blobs.py), rendered from its blob: the WebGL2 wording as the reference emits it, its
    WebGPU wording, and with the opt-in extensions (expression-valued arguments, lens post chain, projection); particle
    blocks take three steps. Every frame and every particle must equal the oracle's bit for bit."""
    rng = random.Random(5 + part)
    frames = 0
    for name, blob in sorted(corpus().items())[part::4]:
        cin = game.load_blob(blob).cinematics[0]
        w, h = rng.choice([(96, 54), (64, 36), (50, 31)])
        times = [k / 10 for k in range(4)] if cin.uses_memory else [0.0, 1.7]
        rows = cin.uniform_rows(times, w, h, audio={"bass": 0.8, "mid": 0.5, "treble": 0.3, "energy": 0.6, "beat": 1.0})
        rows[:, 8:10] = (0.3, 0.7)
        derived = cin.eval_derived(rows)
        for flags in (dict(), dict(semantics="wgsl"), dict(animate=True, lens_post=True, project=True)):
            r = cin.renderer(w, h, **flags)
            got = r.render(rows)
            r.close()
            ext = "animate" in flags
            want = oracle.render_sequence(blob, w, h, rows, cin.uses_memory, semantics=flags.get("semantics", "glsl"),
                                          derived=derived if ext else None, lens_post=ext, project=ext)
            assert np.array_equal(got, want), (name, flags)
            frames += len(rows)
        if cin.has_gravity:
            p0 = oracle.seed_particles(700)
            r = cin.renderer(16, 16)
            r.particles_init(p0[:, :2], p0[:, 2:])
            r.particles_step(1 / 60, 3)
            pos, vel = r.particles_read()
            r.close()
            want, _ = oracle.gravity(blob, p0, 1 / 60, 3)
            assert np.array_equal(pos, want[:, :2], equal_nan=True) and np.array_equal(vel, want[:, 2:], equal_nan=True), name
    assert frames >= 120
