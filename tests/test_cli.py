"""The `game` command-line host: check/ir/builtins on the CPU, render on the GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import C1_HELLO, C4_TRAILS, ROOT

GAME = os.path.join(ROOT, "game-engine_amd", "game")


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


def test_render_requires_backend_flag(tmp_path):
    f = tmp_path / "g.game"
    f.write_text(C1_HELLO)
    r = run("render", str(f))
    assert r.returncode == 2 and "--hip" in r.stderr


@pytest.mark.gpu
def test_render_writes_frames_equal_to_the_library(tmp_path, game):
    f = tmp_path / "g.game"
    f.write_text(C1_HELLO)
    out = str(tmp_path / "frame")
    r = run("render", str(f), "--hip", "--size", "96x64", "--frames", "3", "--fps", "60", "--batch", "2", "--out", out)
    assert r.returncode == 0, r.stderr
    assert "3 frames at 96x64" in r.stderr
    cin = game.compile(C1_HELLO).cinematics[0]
    want = cin.renderer(96, 64).render(cin.uniform_rows([k / 60.0 for k in range(3)], 96, 64))
    for k in range(3):
        data = open(f"{out}_{k:05d}.ppm", "rb").read()
        header = b"P6\n96 64\n255\n"
        assert data.startswith(header)
        rgb = np.frombuffer(data[len(header):], dtype=np.uint8).reshape(64, 96, 3)
        assert np.array_equal(rgb, want[k][..., :3])
