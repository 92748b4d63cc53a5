"""The allocation-free entry point (game_hip_render_into: uniform staging in a static device array) against the
ordinary one (game_hip_render), through the C ABI: same bytes, on two calls in a row, the second smaller."""
import numpy as np
import pytest

from conftest import C2_FBM, C3_SHOWCASE, C4_TRAILS

pytestmark = pytest.mark.gpu


def _dev_buffer(torch, nbytes):
    buf = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()          # the fill ran on torch's stream, the render runs on the context's
    return buf


@pytest.mark.parametrize("name,src,w,h", [("fbm", C2_FBM, 256, 144), ("showcase-4px-per-lane", C3_SHOWCASE, 320, 90),
                                           ("memory", C4_TRAILS, 128, 72), ("ragged", C2_FBM, 203, 31)],
                         ids=["fbm", "showcase-4px-per-lane", "memory", "ragged"])
def test_render_into_matches_render(game, name, src, w, h):
    import torch
    cin = game.compile(src).cinematics[0]
    ref, dut = cin.renderer(w, h), cin.renderer(w, h)
    audio = lambda ks: {"bass": [0.1 + 0.07 * k for k in ks], "mid": 0.6, "treble": 0.4}
    for ks in (list(range(6)), list(range(6, 9))):          # two calls in a row, the second with fewer frames
        rows = cin.uniform_rows([k / 30.0 for k in ks], w, h, audio=audio(ks))
        want = ref.render(rows)                              # the existing entry point (frames via its own ring)
        buf = _dev_buffer(torch, len(ks) * h * w * 4)
        dut.render_into(rows, buf.data_ptr())
        dut.sync()
        got = buf.cpu().numpy().reshape(len(ks), h, w, 4)
        assert np.array_equal(got, want), f"{name}: frames {ks} differ"
    # the feedback state carries over between the two kinds of call
    rows = cin.uniform_rows([9 / 30.0, 10 / 30.0], w, h, audio=audio([9, 10]))
    assert np.array_equal(dut.render(rows), ref.render(rows))
    ref.close()
    dut.close()


def test_render_into_rejects_what_does_not_fit(game):
    import torch
    cin = game.compile(C2_FBM).cinematics[0]
    r = cin.renderer(16, 8)
    floats = cin.frame_floats + 2
    too_many = 262144 // (floats * 4) + 1
    rows = cin.uniform_rows(np.zeros(too_many), 16, 8)
    buf = _dev_buffer(torch, 16 * 8 * 4)
    with pytest.raises(game.GameError, match="exceed the static scratch"):
        r.render_into(rows, buf.data_ptr())
    with pytest.raises(game.GameError, match="16-byte aligned"):
        r.render_into(rows[:1], buf.data_ptr() + 4)
    r.render_into(rows[:1], buf.data_ptr())                 # still usable after the refusals
    r.sync()
    assert np.array_equal(buf.cpu().numpy().reshape(1, 8, 16, 4), r.render(rows[:1]))
    r.close()
