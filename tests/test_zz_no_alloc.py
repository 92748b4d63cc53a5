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


def test_two_contexts_alternate_without_a_host_wait(game):
    """Every context of a GPU stages its uniform rows in the same static array. Two renderers driven back to back from
    one thread, with no sync in between, must not overwrite each other's rows while a kernel still reads them: each
    call's refill waits on the device for the previous reader."""
    import torch
    a_cin, b_cin = game.compile(C2_FBM).cinematics[0], game.compile(C3_SHOWCASE).cinematics[0]
    w, h, n = 512, 288, 12
    ra, rb = a_cin.renderer(w, h), b_cin.renderer(w, h)
    ref_a, ref_b = a_cin.renderer(w, h), b_cin.renderer(w, h)
    bufs, wants = [], []
    for rnd in range(4):
        ta = [(rnd * n + k) / 24.0 for k in range(n)]
        rows_a = a_cin.uniform_rows(ta, w, h)
        rows_b = b_cin.uniform_rows(ta, w, h, audio={"bass": [0.1 + 0.05 * k for k in range(n)], "mid": 0.6, "treble": 0.3})
        wants += [ref_a.render(rows_a), ref_b.render(rows_b)]
        bufs += [_dev_buffer(torch, n * h * w * 4), _dev_buffer(torch, n * h * w * 4)]
    for rnd in range(4):                                   # queue everything, wait once
        ta = [(rnd * n + k) / 24.0 for k in range(n)]
        ra.render_into(a_cin.uniform_rows(ta, w, h), bufs[2 * rnd].data_ptr())
        rb.render_into(b_cin.uniform_rows(ta, w, h, audio={"bass": [0.1 + 0.05 * k for k in range(n)], "mid": 0.6, "treble": 0.3}),
                       bufs[2 * rnd + 1].data_ptr())
    ra.sync(); rb.sync()
    for i, (buf, want) in enumerate(zip(bufs, wants)):
        assert np.array_equal(buf.cpu().numpy().reshape(n, h, w, 4), want), f"call {i}"
    for r in (ra, rb, ref_a, ref_b):
        r.close()


def test_render_into_is_refused_while_a_frame_filter_is_set(game):
    import torch
    cin = game.compile(C4_TRAILS).cinematics[0]
    w, h = 128, 72
    r = cin.renderer(w, h)
    rows = cin.uniform_rows([0.0, 0.1], w, h, audio={"bass": 0.4})
    buf = _dev_buffer(torch, 2 * h * w * 4)
    r.set_blur(2)
    with pytest.raises(game.GameError, match="frame filter") as ei:
        r.render_into(rows, buf.data_ptr())
    assert ei.value.code == 83                              # GAME_E_UNSUPPORTED
    filtered = r.render(rows)                               # the filtered path itself is untouched by the refusal
    r.set_blur(0)
    r.reset_memory()
    r.render_into(rows, buf.data_ptr())
    r.sync()
    plain = buf.cpu().numpy().reshape(2, h, w, 4)
    ref = cin.renderer(w, h)
    assert np.array_equal(plain, ref.render(rows)) and not np.array_equal(plain, filtered)
    ref.close()
    r.close()


def test_render_into_allocates_no_device_memory(game):
    """Free device memory as the driver reports it before and after a run of calls (the first call included)."""
    import torch
    cin = game.compile(C4_TRAILS).cinematics[0]
    w, h, n = 256, 144, 8
    r = cin.renderer(w, h)
    rows = cin.uniform_rows([k / 30.0 for k in range(n)], w, h, audio={"bass": 0.3})
    buf = _dev_buffer(torch, n * h * w * 4)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(5):
        r.render_into(rows, buf.data_ptr())
    r.sync()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, (free0, free1)
    r.close()
