"""Host-side uniform animation: `~` modulation, the temporal operators and resonance. The reference emits them as
JavaScript (game-compiler/src/codegen/temporal.rs:31-82 and :130-172, codegen/resonance.rs:20-65) that its
renderer never calls; the product applies them to the user uniforms of each frame row. The checker is a Python
statement of those JS classes (float64 = JS numbers, Float32Array storage for the ring buffer)."""
import math

import numpy as np
import pytest

BU = 10


class RingBuffer:            # temporal.rs:32-45
    def __init__(self, secs, fps=60):
        self.size = max(1, math.ceil(secs * fps))
        self.buf = np.zeros(self.size, dtype=np.float32)
        self.head = 0

    def update(self, v):
        self.buf[self.head] = v
        self.head = (self.head + 1) % self.size
        return float(self.buf[self.head])


class EMA:                   # temporal.rs:51-62
    def __init__(self, secs, fps=60):
        samples = max(1, secs * fps)
        self.alpha = 2.0 / (samples + 1.0)
        self.value = None

    def update(self, v):
        if self.value is None:
            self.value = v
            return v
        self.value += self.alpha * (v - self.value)
        return self.value


class Edge:                  # temporal.rs:68-81
    def __init__(self, secs, fps=60):
        self.rate = 1.0 / max(1, secs * fps)
        self.prev = 0
        self.env = 0

    def update(self, v):
        delta = v - self.prev
        self.prev = v
        if delta > 0.01:
            self.env = 1.0
        else:
            self.env = max(0, self.env - self.rate)
        return self.env


class Clamp:                 # temporal.rs:165-170
    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi

    def update(self, v):
        return min(max(v, self.lo), self.hi)


SRC = '''cinematic "temporal" {
  layer reactive {
    bass: 0.5 ~ audio.bass <> 50ms >> 200ms .. [0.0, 1.0]
    beat: 0.0 ~ audio.beat !! 300ms
    wobble: 0.2 ~ sin(time * 2.0) * 0.08
    plain: 0.7
    lag: 1.0 ~ mouse.x * 2.0 - 1.0 >> 1bars
  }
  layer orb { fn: circle(bass) | glow(beat) | tint(wobble, plain, lag) }
  resonate {
    kick -> reactive.plain * 0.3
    snare -> orb.wobble * -1.5
    hat -> nowhere.nothing * 2.0
  }
}'''


def audio_track(n):
    rng = np.random.default_rng(7)
    bass = np.clip(0.5 + 0.5 * np.sin(np.arange(n) * 0.3) + rng.normal(0, 0.1, n), 0, 1)
    beat = (np.arange(n) % 15 == 0).astype(np.float64)
    mouse = np.linspace(0.0, 1.0, n)
    return bass.astype(np.float32), beat.astype(np.float32), mouse.astype(np.float32)


def test_ir_lists_the_animated_parameters(game):
    cin = game.compile(SRC).cinematics[0]
    ir = cin.ir
    names = [u["name"] for u in ir["uniforms"]]
    assert names == ["bass", "beat", "wobble", "plain", "lag"]
    anims = {a["name"]: a for a in ir["anims"]}
    assert set(anims) == {"bass", "beat", "wobble", "lag"}
    assert [o["kind"] for o in anims["bass"]["ops"]] == ["smooth", "delay", "range"]
    assert anims["bass"]["ops"][0]["secs"] == 0.05 and anims["bass"]["ops"][1]["secs"] == 0.2
    assert anims["lag"]["ops"][0]["secs"] == 2.0                      # 1 bar = 2 s at the default 120 BPM
    assert anims["wobble"]["modulation"] == "(sin((time * 2.000000)) * 0.080000)" and anims["wobble"]["evaluated"]
    res = ir["resonances"]
    assert [(r["source"], r["idx"]) for r in res] == [("kick", 3), ("snare", 2), ("hat", -1)]
    assert res[1]["weight"] == -1.5


@pytest.mark.parametrize("fps", [60.0, 24.0])
def test_modulation_and_temporal_operators_follow_the_js_classes(game, fps):
    cin = game.compile(SRC).cinematics[0]
    n = 90
    times = [k / fps for k in range(n)]
    bass, beat, mouse = audio_track(n)
    rows = cin.uniform_rows(times, 64, 36)
    rows[:, 1] = bass
    rows[:, 5] = beat
    rows[:, 8] = mouse
    want = rows.copy()
    an = cin.animator(fps)
    an.apply(rows, times, what=an.MODULATION)
    an.close()
    f32 = lambda x: float(np.float32(x))
    ops = {"bass": [EMA(0.05, fps), RingBuffer(0.2, fps), Clamp(0.0, 1.0)], "beat": [Edge(0.3, fps)], "wobble": [], "lag": [RingBuffer(2.0, fps)]}
    for k in range(n):
        r = want[k]
        t = float(r[0])
        mods = {"bass": float(r[1]), "beat": float(r[5]), "wobble": math.sin(t * f32(2.0)) * f32(0.08),
                "lag": float(r[8]) * f32(2.0) - f32(1.0)}
        for name, idx in (("bass", 0), ("beat", 1), ("wobble", 2), ("lag", 4)):
            v = float(r[BU + idx]) + mods[name]
            for op in ops[name]:
                v = op.update(v)
            r[BU + idx] = v
    assert np.array_equal(rows.view(np.uint32), want.view(np.uint32))
    assert np.all(rows[:, BU + 3] == np.float32(0.7))                  # untouched parameter
    assert rows[:, BU + 1].max() == 1.0 and 0.0 < rows[2, BU + 1] < 1.0    # the edge envelope fires and decays


def test_reset_returns_to_the_first_frame(game):
    cin = game.compile(SRC).cinematics[0]
    times = [k / 60 for k in range(30)]
    bass, beat, mouse = audio_track(30)
    base = cin.uniform_rows(times, 64, 36)
    base[:, 1], base[:, 5], base[:, 8] = bass, beat, mouse
    an = cin.animator()
    a = an.apply(base.copy(), times, what=2)
    an.reset()
    b = an.apply(base.copy(), times, what=2)
    an.close()
    assert np.array_equal(a, b)


def test_resonance_adds_signal_times_weight_times_dt(game):
    cin = game.compile(SRC).cinematics[0]
    row = cin.uniform_rows([0.0], 64, 36)[0]
    an = cin.animator()
    an.step(0.0, 0.25, row, what=an.RESONANCE, signals={"kick": 0.8, "snare": 0.5, "hat": 9.0})
    an.close()
    assert row[BU + 3] == np.float32(float(np.float32(0.7)) + 0.8 * 0.3 * 0.25)
    assert row[BU + 2] == np.float32(float(np.float32(0.2)) + 0.5 * -1.5 * 0.25)
    # a missing signal counts as 0 (`signals[name] || 0`)
    row2 = cin.uniform_rows([0.0], 64, 36)[0]
    an = cin.animator()
    an.step(0.0, 0.25, row2, what=4, signals={"kick": 0.8})
    an.close()
    assert row2[BU + 2] == np.float32(0.2)


def test_all_passes_compose_in_order(game):
    src = '''cinematic "c" {
      layer l { fn: circle(r) | glow(2.0)
        r: 0.1 ~ audio.bass * 0.5 .. [0.0, 0.45] }
      arc { r: 0.2 -> 0.4 over 2s }
      resonate { kick -> l.r * 1.0 }
    }'''
    cin = game.compile(src).cinematics[0]
    row = cin.uniform_rows([1.0], 64, 36, audio={"bass": 0.5})[0]
    an = cin.animator()
    an.step(1.0, 0.5, row, signals={"kick": 0.1})
    an.close()
    arc = 0.2 + (0.4 - 0.2) * 0.5                                       # arcs first
    v = min(max(float(np.float32(arc)) + 0.5 * float(np.float32(0.5)), 0.0), 0.45)   # then modulation and the clamp
    v = float(np.float32(v)) + 0.1 * 1.0 * 0.5                          # then resonance
    assert row[BU] == np.float32(v)


def test_modulation_over_unknown_signals_is_left_out(game):
    cin = game.compile('cinematic "c" { layer l { fn: circle(r) | glow(2.0)\n r: 0.3 ~ data.progress * 2.0 } }').cinematics[0]
    a = cin.ir["anims"][0]
    assert a["modulation"] == "(data.progress * 2.000000)" and not a["evaluated"]
    row = cin.uniform_rows([0.5], 32, 32)[0]
    an = cin.animator()
    an.step(0.5, 1 / 60, row)
    an.close()
    assert row[BU] == np.float32(0.3)


def test_bad_arguments(game):
    cin = game.compile(SRC).cinematics[0]
    with pytest.raises(game.GameError):
        cin.animator(fps=-1.0)
    an = cin.animator()
    row = cin.uniform_rows([0.0], 8, 8)[0]
    with pytest.raises(game.GameError):
        an.step(0.0, 0.1, row, what=8)
    an.close()


@pytest.mark.gpu
def test_gpu_animated_uniforms_drive_the_frames(game, oracle):
    cin = game.compile(SRC).cinematics[0]
    n, w, h = 12, 96, 54
    times = [k / 60 for k in range(n)]
    bass, beat, mouse = audio_track(n)
    rows = cin.uniform_rows(times, w, h)
    rows[:, 1], rows[:, 5], rows[:, 8] = bass, beat, mouse
    an = cin.animator()
    an.apply(rows, times, signals=[{"kick": float(b)} for b in beat])
    an.close()
    r = cin.renderer(w, h)
    got = r.render(rows)
    r.close()
    want = oracle.render_sequence(cin.blob, w, h, rows, False)
    assert np.array_equal(got, want)
    assert not np.array_equal(got[0], got[5])


def test_round_and_mod_follow_the_wording_of_their_side(game):
    """The same expression language has two wordings in the reference: stage arguments are shader expressions
    (codegen/expr.rs:50-144, WGSL: round() goes to the nearest even integer on a tie), modulations are JavaScript
    (expr.rs:147-260: Math.round = floor(x + 0.5)). `%` is the exact truncated remainder in both."""
    src = '''cinematic "r" { layer a { fn: circle(round(time * 2.5) * 0.1) | glow(mod(0.0 - time * 7.5, 2.0) + 2.0)
      x: 0.0 ~ round(time * 2.5)
      y: 0.0 ~ mod(0.0 - time * 7.5, 2.0) } }'''
    cin = game.compile(src).cinematics[0]
    assert cin.derived_count == 2
    rows = cin.uniform_rows([1.0, 0.6], 16, 9)
    d = cin.eval_derived(rows)
    assert d[0, 0] == np.float32(2.0 * float(np.float32(0.1)))        # round(2.5) = 2: ties to even
    assert d[1, 0] == np.float32(2.0 * float(np.float32(0.1)))        # round(1.5) = 2
    assert d[0, 1] == np.float32(-1.5 + 2.0) and d[1, 1] == np.float32(math.fmod(0.0 - float(np.float32(0.6)) * 7.5, 2.0) + 2.0)
    an = cin.animator(60.0)
    an.apply(rows, [1.0, 0.6], what=an.MODULATION)
    an.close()
    assert rows[0, BU] == 3.0 and rows[1, BU] == 2.0                   # Math.round(2.5) = 3, Math.round(1.5) = 2
    assert rows[0, BU + 1] == np.float32(-1.5)
