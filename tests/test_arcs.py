"""Arc timelines (game-compiler/src/runtime/arc.rs): the reference emits them as a JS function arcUpdate(time,
params) that nothing calls; here the host applies the same timeline to the user uniforms of each frame row.
The checker is a Python statement of that JS (float64 = JS numbers), written from the emitted text."""
import math

import numpy as np
import pytest

from conftest import REFERENCE, reference_available

BU_COUNT = 10


def js_ease(name, t):   # arc.rs:22-30
    if name == "expo_in":
        return 0 if t == 0 else 2 ** (10 * (t - 1))
    if name == "expo_out":
        return 1 if t == 1 else 1 - 2 ** (-10 * t)
    if name == "cubic_in_out":
        return 4 * t * t * t if t < 0.5 else 1 - (-2 * t + 2) ** 3 / 2
    if name == "smooth":
        return t * t * (3 - 2 * t)
    if name == "elastic":
        return 0 if t == 0 else 1 if t == 1 else -(2 ** (10 * (t - 1))) * math.sin((t - 1.1) * 5 * math.pi)
    if name == "bounce":
        n, d = 7.5625, 2.75
        if t < 1 / d:
            return n * t * t
        if t < 2 / d:
            t -= 1.5 / d
            return n * t * t + 0.75
        if t < 2.5 / d:
            t -= 2.25 / d
            return n * t * t + 0.9375
        t -= 2.625 / d
        return n * t * t + 0.984375
    return t            # linear, and `_ease[a.ease] || _ease.linear` for unknown names


def js_arc_update(timeline, time, params):   # arc.rs:110-122
    for a in timeline:
        if a["idx"] < 0 or time < a["start"]:
            continue
        if time >= a["start"] + a["dur"]:
            params[a["idx"]] = a["to"]
            continue
        t = (time - a["start"]) / a["dur"]
        params[a["idx"]] = a["from"] + (a["to"] - a["from"]) * js_ease(a["ease"], t)


SRC = '''cinematic "arcs" {
  layer orb { fn: circle(radius) | glow(brightness) | tint(hue, 0.5, 0.2)
    radius: 0.1
    brightness: 1.0
    hue: 0.3 }
  arc {
    radius: 0.05 -> 0.4 over 2s expo_out
    brightness: 0.5 -> 3.0 over 1500ms smooth
    hue: 0.0 -> 1.0 over 1bars elastic
    radius: 0.4 -> 0.1 over 2s bounce
    nothing.here: 0.0 -> 1.0 over 1s linear
    brightness: 3.0 -> pi / 2 over 2s cubic_in_out
  }
  arc {
    hue: 1.0 -> 2 ^ 0.5 - 1 over 3s expo_in
    radius: 0.1 -> 0.2 over 1s wobble
  }
}'''


def test_timeline_entries_are_sequential_and_resolved(game):
    cin = game.compile(SRC).cinematics[0]
    arcs = cin.ir["arcs"]
    assert cin.arc_count == len(arcs) == 8
    # arc.rs:62-93: the cursor advances by every entry's duration, across blocks; bars are 2 s each
    assert [a["start"] for a in arcs] == [0.0, 2.0, 3.5, 5.5, 7.5, 8.5, 10.5, 13.5]
    assert [a["dur"] for a in arcs] == [2.0, 1.5, 2.0, 2.0, 1.0, 2.0, 3.0, 1.0]
    assert [a["idx"] for a in arcs] == [0, 1, 2, 0, -1, 1, 2, 0]            # unknown target -> skipped by the JS
    assert arcs[5]["to"] == 3.14159265358979 / 2 and arcs[6]["to"] == 2 ** 0.5 - 1   # expr.rs constants, JS `**`
    assert arcs[7]["ease"] == "wobble"                                       # unknown easing: falls back to linear


def test_entry_without_easing_swallows_the_next_target(game):
    """parser.rs:788-792 takes any identifier after the duration as the easing, so an entry without one must be
    the last of its block; the next target is otherwise read as an easing name and the colon is an error."""
    from conftest import load_product
    GameError = load_product().GameError
    with pytest.raises(GameError) as err:
        game.compile('cinematic "t" { layer { fn: circle(r) | glow(1.0)\n r: 0.1 } arc { r: 0.0 -> 1.0 over 1s\n r: 1.0 -> 0.0 over 1s } }')
    assert "expected identifier, found `:`" in str(err.value)
    ok = game.compile('cinematic "t" { layer { fn: circle(r) | glow(1.0)\n r: 0.1 } arc { r: 0.0 -> 1.0 over 1s } }').cinematics[0]
    assert ok.ir["arcs"][0]["ease"] == "linear"


def test_arc_update_matches_the_emitted_javascript(game):
    cin = game.compile(SRC).cinematics[0]
    arcs = cin.ir["arcs"]
    times = np.concatenate([np.linspace(0.0, 16.0, 641), [2.0, 3.5, 5.5, 7.5, 8.5, 10.5, 13.5, 14.5]]).astype(np.float32)
    rows = cin.uniform_rows(times, 64, 36, arcs=True)
    plain = cin.uniform_rows(times, 64, 36)
    assert np.array_equal(rows[:, :BU_COUNT], plain[:, :BU_COUNT])          # built-in uniforms untouched
    for k, t in enumerate(times):
        params = [float(np.float32(d)) for _, d in cin.uniforms]            # defaults, as the Float32 row holds them
        js_arc_update(arcs, float(t), params)
        want = np.array(params, dtype=np.float32)
        assert np.array_equal(rows[k, BU_COUNT:], want), (t, rows[k, BU_COUNT:], want)
    assert np.array_equal(rows[0, BU_COUNT:], np.array([0.05, 1.0, 0.3], dtype=np.float32))   # only the first entry has begun
    assert rows[-1, BU_COUNT] == np.float32(0.2)                            # the last entry holds its end value


def test_user_values_then_arcs(game):
    cin = game.compile(SRC).cinematics[0]
    rows = cin.uniform_rows([1.0, 1.0], 8, 8, user={"hue": [0.7, 0.9]}, arcs=True)
    assert list(rows[:, BU_COUNT + 2]) == [np.float32(0.7), np.float32(0.9)]   # hue's first entry starts at 3.5 s


def test_without_arcs_nothing_changes(game):
    cin = game.compile('cinematic "t" { layer { fn: circle(r) | glow(1.0)\n r: 0.2 } }').cinematics[0]
    assert cin.arc_count == 0
    assert np.array_equal(cin.uniform_rows([3.0], 8, 8, arcs=True), cin.uniform_rows([3.0], 8, 8))


@pytest.mark.skipif(not reference_available(), reason="reference corpus not present")
def test_reference_tutorial_timeline_as_emitted(game):
    """tutorial-10: timestamps are labels only (generate_arc_js ignores them), `param -> v` starts from 0."""
    src = open(f"{REFERENCE}/examples/tutorial-10-arc-timeline.game").read()
    cin = game.compile(src).cinematics[0]
    arcs = cin.ir["arcs"]
    assert [(a["target"], a["from"], a["to"], a["start"], a["dur"], a["ease"]) for a in arcs[:4]] == [
        ("radius", 0.05, 0.05, 0.0, 0.0, "linear"), ("brightness", 0.3, 0.3, 0.0, 0.0, "linear"),
        ("radius", 0.0, 0.25, 0.0, 2.0, "expo_out"), ("brightness", 0.0, 3.0, 2.0, 2.0, "smooth")]
