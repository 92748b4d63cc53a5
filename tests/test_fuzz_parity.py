"""Randomised cinematics (seeded): pipelines drawn from all 38 builtins with the type flow Position -> Sdf -> Color
(game-compiler/src/builtins.rs:43-293), arguments from ordinary and awkward values, one to three layers, sometimes
memory, time / audio / mouse / user uniforms as arguments. The GPU frames must equal the CPU oracle's bit for bit, in
both wordings of the builtins. The generator is deterministic, so a failure names its seed."""
import random

import numpy as np
import pytest

ORDINARY = {
    "radius": (0.05, 0.6), "width": (0.005, 0.2), "points": (3.0, 9.0), "inner": (0.05, 0.3), "scale": (0.5, 9.0),
    "octaves": (1.0, 7.0), "persistence": (0.2, 0.8), "lacunarity": (1.5, 3.0), "intensity": (0.2, 5.0),
    "r": (0.0, 1.2), "g": (0.0, 1.2), "b": (0.0, 1.2), "threshold": (0.0, 1.0), "strength": (0.0, 2.5), "amount": (0.0, 1.5),
    "x": (-0.8, 0.8), "y": (-0.8, 0.8), "angle": (-7.0, 7.0), "s": (0.3, 3.0), "cutoff": (-0.2, 0.6), "axis": (0.0, 1.0),
    "count": (1.0, 9.0), "factor": (0.0, 1.0), "thickness": (0.005, 0.1), "freq": (0.5, 6.0), "frequency": (0.5, 300.0),
    "amplitude": (0.0, 0.5), "w": (0.05, 0.6), "h": (0.05, 0.6), "sides": (3.0, 9.0), "exposure": (0.3, 3.0), "offset": (0.0, 0.05),
    "bass": (0.0, 1.0), "mid": (0.0, 1.0), "treble": (0.0, 1.0), "color_a": (0.0, 1.0), "color_b": (0.0, 1.0), "mode": (0.0, 1.0),
}
AWKWARD = [0.0, -0.0 + 0.0, 1.0, -1.0, 0.5, 2.0, 1e-6, -1e-6, 1e6, -1e6, 0.333333, 255.0, 1e-30, 3.999999, 4.0, 64.0]
PALETTES = ["fire", "ice", "rainbow", "ocean", "forest", "neon", "sunset", "plasma", "nosuchname"]
SIGNALS = ["time", "audio.bass", "audio.mid", "audio.treble", "audio.energy", "audio.beat", "mouse.x", "mouse.y", "knob", "dial"]


EXPRESSIONS = ["time * 0.5 + 0.1", "sin(time * 2.0) * 0.2 + 0.3", "audio.bass * 0.5", "0.2 + mouse.x * 0.3", "abs(cos(time)) * knob",
               "mix(0.1, 0.6, audio.beat)", "-(mouse.y - 0.5)", "2.0 ^ audio.mid", "min(time, 1.5)", "p.x * 2.0", "nothing * 3.0",
               # vector-valued subexpressions and the scalar functions added with them
               "length(p) * 0.5", "dot(uv, [0.3, 0.4]) + 0.2", "distance(p, [0.2 * knob, -0.1])", "dot(normalize(p), [0.6, 0.8])",
               "length(p * 2.0 - uv) * 0.3", "mod(p.x * 3.0 + time, 0.7)", "atan2(p.y, p.x) * 0.2", "round(uv.y * 4.0) * 0.1",
               "length(abs(p) - [0.3, 0.2])", "p * 2.0"]


def make_cinematic(game, rng, expressions=False):
    table = {}
    for b in game.list_builtins():
        table.setdefault((b["input"], b["output"]), []).append((b["name"], b["params"]))

    def arg(stage, param):
        roll = rng.random()
        if stage == "palette":
            return rng.choice(PALETTES)
        if stage == "fbm" and param == "octaves":
            return f"{float(rng.randint(1, 7)):.1f}" if roll < 0.9 else rng.choice(["0.0", "40.0", "2.7", "knob"])
        if expressions and roll > 0.8:
            return rng.choice(EXPRESSIONS)
        if roll < 0.12:
            return rng.choice(SIGNALS)
        if roll < 0.22:
            return f"{rng.choice(AWKWARD):.6f}"
        lo, hi = ORDINARY.get(param, (0.0, 1.0))
        return f"{rng.uniform(lo, hi):.6f}"

    def call(entry):
        name, params = entry
        given = params[: rng.randint(0, len(params))] if rng.random() < 0.2 else params     # trailing defaults sometimes
        return f"{name}({', '.join(arg(name, p) for p in given)})"

    layers = []
    uses_memory = False
    for li in range(rng.randint(1, 3)):
        stages = [call(rng.choice(table[("Position", "Position")])) for _ in range(rng.randint(0, 3))]
        if rng.random() < 0.15:
            stages.append(call(rng.choice(table[("Position", "Color")])))
        else:
            stages.append(call(rng.choice(table[("Position", "Sdf")])))
            stages += [call(rng.choice(table[("Sdf", "Sdf")])) for _ in range(rng.randint(0, 2))]
            stages.append(call(rng.choice(table[("Sdf", "Color")])))
        stages += [call(rng.choice(table[("Color", "Color")])) for _ in range(rng.randint(0, 4))]
        mem = ""
        if rng.random() < 0.25:
            mem = f" memory: {rng.choice([0.5, 0.9, 0.97, 0.0, 1.0])}"
            uses_memory = True
        layers.append(f"  layer l{li}{mem} {{ fn: {' | '.join(stages)} }}")
    tail = ""
    if expressions:   # a lens post chain too, and sometimes a projection block
        post = [call(rng.choice(table[("Color", "Color")])) for _ in range(rng.randint(1, 3))]
        layers.append(f"  lens {{ post: {' | '.join(post)} }}")
        if rng.random() < 0.4:
            tail = "\nproject " + rng.choice(["dome (fisheye: 150)", "dome", "led (count: 12)", "cube", "flat"]) + " { source: fuzz }"
    return 'cinematic "fuzz" {\n' + "\n".join(layers) + "\n}" + tail, uses_memory


def frames_and_rows(game, src, rng, w, h):
    cin = game.compile(src).cinematics[0]
    times = [0.0, 0.7, 121.3] if not cin.uses_memory else [k / 10 for k in range(4)]
    rows = cin.uniform_rows(times, w, h, audio={"bass": rng.random(), "mid": rng.random(), "treble": rng.random(), "energy": rng.random(),
                                               "beat": float(rng.random() < 0.5)},
                            user={u["name"]: rng.uniform(0.0, 2.0) for u in cin.ir["uniforms"]})
    rows[:, 8] = rng.random()
    rows[:, 9] = rng.random()
    return cin, rows


def test_generator_is_deterministic_and_covers_every_builtin(game):
    seen = set()
    for seed in range(60):
        src, _ = make_cinematic(game, random.Random(seed))
        again, _ = make_cinematic(game, random.Random(seed))
        assert src == again
        cin = game.compile(src).cinematics[0]
        seen |= {s["name"] for l in cin.ir["layers"] for s in l["stages"]}
    assert seen == {b["name"] for b in game.list_builtins()}


@pytest.mark.parametrize("seed", range(8))
def test_oracle_renders_every_generated_cinematic(game, oracle, seed):
    """CPU only: the oracle accepts the generated IR and both wordings give finite, reproducible frames."""
    rng = random.Random(seed)
    src, _ = make_cinematic(game, rng)
    cin, rows = frames_and_rows(game, src, rng, 40, 24)
    for sem in ("glsl", "wgsl"):
        a = oracle.render_sequence(cin.blob, 40, 24, rows, cin.uses_memory, semantics=sem, threads=1)
        b = oracle.render_sequence(cin.blob, 40, 24, rows, cin.uses_memory, semantics=sem, threads=4)
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(60))
def test_gpu_matches_oracle_on_generated_cinematics(game, oracle, seed):
    rng = random.Random(seed)
    src, _ = make_cinematic(game, rng)
    w, h = rng.choice([(96, 54), (100, 37), (64, 64), (33, 17)])
    cin, rows = frames_and_rows(game, src, rng, w, h)
    for sem in ("glsl", "wgsl"):
        r = cin.renderer(w, h, semantics=sem)
        got = r.render(rows)
        r.close()
        want = oracle.render_sequence(cin.blob, w, h, rows, cin.uses_memory, semantics=sem)
        if not np.array_equal(got, want):
            bad = np.argwhere(got != want)
            raise AssertionError(f"seed {seed} ({sem}): {len(bad)} differing channel values, first at {bad[0].tolist()}: "
                                 f"gpu {got[tuple(bad[0])]} oracle {want[tuple(bad[0])]}\n{src}")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(100, 130))
def test_gpu_matches_oracle_with_expression_arguments_and_lens_post(game, oracle, seed):
    """The opt-in extensions together: expression-valued arguments evaluated per frame (scalar and vector-valued
    subexpressions), the lens post chain and, for some seeds, a projection block."""
    rng = random.Random(seed)
    src, _ = make_cinematic(game, rng, expressions=True)
    w, h = rng.choice([(96, 54), (100, 37), (64, 64)])
    cin, rows = frames_and_rows(game, src, rng, w, h)
    derived = cin.eval_derived(rows)
    sem = rng.choice(["glsl", "wgsl"])
    r = cin.renderer(w, h, semantics=sem, animate=True, lens_post=True, project=True)
    got = r.render(rows)
    r.close()
    want = oracle.render_sequence(cin.blob, w, h, rows, cin.uses_memory, semantics=sem, derived=derived, lens_post=True, project=True)
    assert np.array_equal(got, want), f"seed {seed} ({sem})\n{src}"
