"""The `game` command-line host: check/ir/builtins on the CPU, render on the GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import C1_HELLO, C4_TRAILS, ROOT

GAME = os.path.join(ROOT, "game-engine_amd", "game")
pytestmark = pytest.mark.skipif(not os.path.exists(GAME), reason="game-engine_amd/game is not built (make -C game-engine_amd)")


def run(*args, **kw):
    return subprocess.run([GAME, *args], capture_output=True, text=True, timeout=120, **kw)


def test_check_ok_and_error_exit_status(tmp_path):
    good = tmp_path / "good.game"
    good.write_text(C1_HELLO)
    bad = tmp_path / "bad.game"
    bad.write_text('cinematic "x" { layer { fn: circle(0.3) | blorp(1) } }')
    r = run("check", str(good))
    assert r.returncode == 0 and f"[game] {good}: ok" in r.stderr
    r = run("check", str(good), str(bad))
    # main.rs:213-257: errors are rendered with source context and the exit status is 1
    assert r.returncode == 1
    assert "error[E001]" in r.stderr and "unknown stage function: 'blorp'" in r.stderr and "did you mean 'bloom'" in r.stderr


def test_ir_and_builtins(tmp_path):
    f = tmp_path / "m.game"
    f.write_text(C4_TRAILS)
    r = run("ir", str(f))
    assert r.returncode == 0
    ir = json.loads(r.stdout)
    assert len(ir) == 1 and ir[0]["uses_memory"] is True
    r = run("builtins")
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 38 and lines[0] == "circle|Position|Sdf|radius"


def test_ir_blob_out_and_back_in(tmp_path, game):
    """`game ir FILE --out X.gir` writes the resolved-IR blob; every command that takes a .game file takes the blob too."""
    f = tmp_path / "m.game"
    f.write_text(C4_TRAILS)
    gir = tmp_path / "m.gir"
    assert run("ir", str(f), "--out", str(gir)).returncode == 0
    assert gir.read_bytes() == game.compile(C4_TRAILS).cinematics[0].blob
    r = run("ir", str(gir))
    assert r.returncode == 0 and json.loads(r.stdout)[0]["uses_memory"] is True
    a, b = run("kernel", str(f), "--out", str(tmp_path / "ka")), run("kernel", str(gir), "--out", str(tmp_path / "kb"))
    assert a.returncode == 0 and a.stdout == b.stdout and "gb_memory_mix" in a.stdout
    bad = tmp_path / "bad.gir"
    bad.write_bytes(gir.read_bytes()[:-8])
    r = run("ir", str(bad))
    assert r.returncode == 1 and "game_ir_from_blob" in r.stderr


def test_render_requires_backend_flag(tmp_path):
    f = tmp_path / "g.game"
    f.write_text(C1_HELLO)
    r = run("render", str(f))
    assert r.returncode == 2 and "--hip" in r.stderr


def test_compare_reports_similarity_max_diff_and_psnr(tmp_path):
    a = np.random.default_rng(0).integers(0, 256, (8, 12, 3), dtype=np.uint8)
    b = a.copy()
    b[3, 4, 1] ^= 4                                   # one channel of one pixel off by 4
    for name, img in (("a", a), ("b", b)):
        (tmp_path / f"{name}.ppm").write_bytes(b"P6\n12 8\n255\n" + img.tobytes())
    r = run("compare", str(tmp_path / "a.ppm"), str(tmp_path / "a.ppm"))
    assert r.returncode == 0 and "similarity 100.0000 %" in r.stdout and "max abs diff 0" in r.stdout
    r = run("compare", str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm"))
    # snapshot.rs:240-258: a pixel matches when every RGB channel differs by at most the threshold
    assert r.returncode == 1 and "similarity 98.9583 %" in r.stdout and "max abs diff 4" in r.stdout
    mse = 16.0 / (8 * 12 * 3)
    assert f"PSNR {10 * np.log10(255.0 ** 2 / mse):.2f} dB" in r.stdout
    r = run("compare", str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm"), "--threshold", "4")
    assert "similarity 100.0000 %" in r.stdout and r.returncode == 1


def _decode_png(path):
    """Minimal PNG reader (8-bit RGBA, filter 0 rows) through zlib, independent of the tool's own reader."""
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    o, idat, w, h = 8, b"", 0, 0
    while o < len(data):
        n, kind = struct.unpack(">I4s", data[o:o + 8])
        body = data[o + 8:o + 8 + n]
        assert struct.unpack(">I", data[o + 8 + n:o + 12 + n])[0] == zlib.crc32(kind + body)
        if kind == b"IHDR":
            w, h, depth, colour = struct.unpack(">IIBB", body[:10])
            assert (depth, colour) == (8, 6)
        elif kind == b"IDAT":
            idat += body
        o += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, w * 4 + 1)
    assert np.all(raw[:, 0] == 0)
    return raw[:, 1:].reshape(h, w, 4)


@pytest.mark.gpu
def test_render_png_sink_and_compare(tmp_path, game):
    f = tmp_path / "g.game"
    f.write_text(C1_HELLO)
    out = str(tmp_path / "p")
    assert run("render", str(f), "--hip", "--size", "300x70", "--frames", "2", "--out", out, "--png").returncode == 0
    assert run("render", str(f), "--hip", "--size", "300x70", "--frames", "2", "--out", out).returncode == 0
    cin = game.compile(C1_HELLO).cinematics[0]
    want = cin.renderer(300, 70).render(cin.uniform_rows([0.0, 1 / 60.0], 300, 70))
    for k in range(2):
        assert np.array_equal(_decode_png(f"{out}_{k:05d}.png"), want[k])
        r = run("compare", f"{out}_{k:05d}.png", f"{out}_{k:05d}.ppm")
        assert r.returncode == 0 and "max abs diff 0" in r.stdout


@pytest.mark.gpu
def test_render_writes_frames_equal_to_the_library(tmp_path, game):
    f = tmp_path / "g.game"
    f.write_text(C1_HELLO)
    out = str(tmp_path / "frame")
    r = run("render", str(f), "--hip", "--size", "96x64", "--frames", "3", "--fps", "60", "--batch", "2", "--out", out)
    assert r.returncode == 0, r.stderr
    assert "3 frames at 96x64" in r.stderr
    rj = run("render", str(f), "--hip", "--size", "96x64", "--frames", "3", "--json")
    rec = json.loads(rj.stderr.strip().splitlines()[-1])
    assert rec["frames"] == 3 and rec["width"] == 96 and rec["device_mpixel_s"] > 0
    cin = game.compile(C1_HELLO).cinematics[0]
    want = cin.renderer(96, 64).render(cin.uniform_rows([k / 60.0 for k in range(3)], 96, 64))
    for k in range(3):
        data = open(f"{out}_{k:05d}.ppm", "rb").read()
        header = b"P6\n96 64\n255\n"
        assert data.startswith(header)
        rgb = np.frombuffer(data[len(header):], dtype=np.uint8).reshape(64, 96, 3)
        assert np.array_equal(rgb, want[k][..., :3])


@pytest.mark.gpu
def test_render_from_ir_blob_equals_render_from_source(tmp_path):
    f, gir = tmp_path / "t.game", tmp_path / "t.gir"
    f.write_text(C4_TRAILS)
    assert run("ir", str(f), "--out", str(gir)).returncode == 0
    for src, out in ((f, "s"), (gir, "b")):
        assert run("render", str(src), "--hip", "--size", "80x48", "--frames", "4", "--out", str(tmp_path / out)).returncode == 0
    for k in range(4):
        assert (tmp_path / f"s_{k:05d}.ppm").read_bytes() == (tmp_path / f"b_{k:05d}.ppm").read_bytes()


@pytest.mark.gpu
def test_render_animate_runs_modulation_and_expression_arguments(tmp_path, game):
    src = '''cinematic "pulse" {
      layer orb { fn: rotate(time * 2.0) | star(5.0, radius, 0.1) | glow(2.0) | tint(1.0, 0.7, 0.3)
        radius: 0.25 ~ sin(time * 6.0) * 0.1 <> 100ms }
    }'''
    f = tmp_path / "p.game"
    f.write_text(src)
    out = str(tmp_path / "a")
    r = run("render", str(f), "--hip", "--animate", "--size", "96x64", "--frames", "4", "--fps", "30", "--batch", "2", "--out", out)
    assert r.returncode == 0, r.stderr
    cin = game.compile(src).cinematics[0]
    times = [k / 30.0 for k in range(4)]
    rows = cin.uniform_rows(times, 96, 64)
    an = cin.animator(30.0)
    an.apply(rows, times)
    an.close()
    rd = cin.renderer(96, 64, animate=True)
    want = rd.render(rows)
    rd.close()
    header = b"P6\n96 64\n255\n"
    frames = []
    for k in range(4):
        data = open(f"{out}_{k:05d}.ppm", "rb").read()
        frames.append(np.frombuffer(data[len(header):], dtype=np.uint8).reshape(64, 96, 3))
        assert np.array_equal(frames[k], want[k][..., :3])
    assert not np.array_equal(frames[0], frames[3])


@pytest.mark.gpu
def test_render_raw_frames_to_stdout(tmp_path, game):
    """`--out -` writes the frames as raw RGBA8 to stdout (the form an external encoder takes from a pipe)."""
    f = tmp_path / "g.game"
    f.write_text(C4_TRAILS)
    r = subprocess.run([GAME, "render", str(f), "--hip", "--size", "64x36", "--frames", "3", "--fps", "30", "--out", "-"],
                       capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert len(r.stdout) == 3 * 64 * 36 * 4
    cin = game.compile(C4_TRAILS).cinematics[0]
    rd = cin.renderer(64, 36)
    want = rd.render(cin.uniform_rows([k / 30.0 for k in range(3)], 64, 36))
    rd.close()
    assert np.array_equal(np.frombuffer(r.stdout, dtype=np.uint8).reshape(3, 36, 64, 4), want)


@pytest.mark.gpu
def test_render_with_imports_and_lens_post(tmp_path, game):
    (tmp_path / "lib.game").write_text('cinematic "lib" { define orb(size) { circle(size) | glow(2.0) } }')
    src = 'import "lib" expose orb\ncinematic "m" { layer a { fn: orb(0.25) | tint(1.0, 0.6, 0.2) }\n lens { post: vignette(0.4, 0.7) | grain(0.02) } }'
    f = tmp_path / "m.game"
    f.write_text(src)
    out = str(tmp_path / "o")
    r = run("render", str(f), "--hip", "--size", "64x36", "--frames", "1")
    assert r.returncode == 1 and "unknown stage function: 'orb'" in r.stderr          # imports are opt-in
    r = run("render", str(f), "--hip", "--imports", "--lens-post", "--size", "64x36", "--frames", "1", "--out", out)
    assert r.returncode == 0, r.stderr
    cin = game.compile_file(str(f)).cinematics[0]
    rd = cin.renderer(64, 36, lens_post=True)
    want = rd.render(cin.uniform_rows([0.0], 64, 36))
    rd.close()
    data = open(out + "_00000.ppm", "rb").read()
    header = b"P6\n64 36\n255\n"
    assert np.array_equal(np.frombuffer(data[len(header):], dtype=np.uint8).reshape(36, 64, 3), want[0][..., :3])


@pytest.mark.gpu
def test_render_split_over_gpu_workers_writes_the_same_frames(tmp_path, game):
    """--gpus N: one context and one host thread per GPU, frame k -> worker k mod N. With one GPU on the box the workers
    share it (test hook); the files must equal the single-worker run's, animated parameters included."""
    src = '''cinematic "pulse" {
      layer orb { fn: rotate(time * 2.0) | star(5.0, radius, 0.1) | glow(2.0) | tint(1.0, 0.7, 0.3)
        radius: 0.25 ~ sin(time * 6.0) * 0.1 <> 100ms }
    }'''
    f = tmp_path / "p.game"
    f.write_text(src)
    one, three = str(tmp_path / "one"), str(tmp_path / "three")
    common = ["render", str(f), "--hip", "--animate", "--size", "96x64", "--frames", "7", "--fps", "30", "--batch", "2"]
    r = run(*common, "--out", one)
    assert r.returncode == 0, r.stderr
    r = run(*common, "--gpus", "3", "--out", three, env={**os.environ, "GAME_CLI_SHARE_DEVICE": "1"})
    assert r.returncode == 0, r.stderr
    for k in range(7):
        assert open(f"{one}_{k:05d}.ppm", "rb").read() == open(f"{three}_{k:05d}.ppm", "rb").read(), k
    # more workers than devices is refused without the hook; memory cinematics are refused outright
    r = run(*common, "--gpus", "64", "--out", three)
    assert r.returncode == 1 and "devices" in r.stderr
    m = tmp_path / "m.game"
    m.write_text(C4_TRAILS)
    r = run("render", str(m), "--hip", "--size", "64x36", "--frames", "2", "--gpus", "2", env={**os.environ, "GAME_CLI_SHARE_DEVICE": "1"})
    assert r.returncode == 1 and "frame-serial" in r.stderr
