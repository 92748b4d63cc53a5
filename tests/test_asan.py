"""AddressSanitizer + UndefinedBehaviorSanitizer pass over the host side: the front end and the kernel generator
(`make -C game-engine_amd asan`, no HIP) run every .game file of the reference corpus (when the reference tree is
present), the benchmark cinematics, the builtin gallery and the seeded generator of the parity fuzz test; the IR blobs
they produce then go, whole and damaged at random, through the library's own blob loader (game_ir_from_blob, then the
kernel generators again) and through the oracle's blob reader (`make -C oracle asan`). Any out-of-bounds
access, use after free or undefined operation aborts the binaries (-fno-sanitize-recover)."""
import glob
import os
import random
import subprocess

import pytest

from conftest import C1_HELLO, C2_FBM, C2B_NOISE, C3_SHOWCASE, C4_TRAILS, C5_GRAVITY, REFERENCE, ROOT, reference_available
from gallery import GALLERY
from test_fuzz_parity import make_cinematic

ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.fixture(scope="module")
def binaries():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "game-engine_amd"), "asan"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    return os.path.join(ROOT, "game-engine_amd", "build", "game_asan"), os.path.join(ROOT, "oracle", "_asan", "game_oracle_fuzz")


def run(cmd):
    p = subprocess.run(cmd, env=ENV, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "no fault" in p.stdout, (p.stdout[-2000:], p.stderr[-6000:])
    return p.stdout


def test_host_side_is_clean_under_sanitizers(binaries, tmp_path, game):
    driver, fuzz = binaries
    srcdir, blobdir = tmp_path / "src", tmp_path / "blobs"
    srcdir.mkdir()
    blobdir.mkdir()
    sources = [C1_HELLO, C2_FBM, C2B_NOISE, C3_SHOWCASE, C4_TRAILS, C5_GRAVITY] + list(GALLERY.values())
    for seed in range(90):
        sources.append(make_cinematic(game, random.Random(seed), expressions=seed >= 60)[0])
    # inputs that must be refused, not crashed on
    sources += ["", "cinematic", 'cinematic "x" { layer { fn: circle( } }', "\x00\xff\xfe", 'import "nowhere.game" expose ALL\ncinematic "a" { layer l { fn: circle() | glow() } }',
                'cinematic "d" { layer l { fn: ' + "(" * 400 + "1" + ")" * 400 + " } }",
                'cinematic "p" { layer l { fn: circle(length(p * [1.0, 2.0]) + dot(uv, normalize(p))) | glow(mod(p.x, 0.3)) } }\nproject dome (fisheye: 90) { source: p }']
    for i, s in enumerate(sources):
        (srcdir / f"{i:03d}.game").write_bytes(s.encode("utf-8", "surrogateescape") if isinstance(s, str) else s)
    files = sorted(str(p) for p in srcdir.glob("*.game"))
    libs = []
    if reference_available():
        for d in ("examples", "presets", "stdlib", "game-compiler/examples"):
            files += sorted(glob.glob(os.path.join(REFERENCE, d, "*.game")))
        libs = ["--lib", os.path.join(REFERENCE, "stdlib")]
    # every blob the corpus produces also goes back in through game_ir_from_blob, whole and damaged 25 times
    out = run([driver, *libs, "--blobs", str(blobdir), "--mutate", "25", "--seed", "20261111", *files])
    compiled = int(out.split(" files compiled")[0].split()[-1])
    loaded, refused = int(out.split(" damaged blobs loaded")[0].split()[-1]), int(out.split(" refused, no fault")[0].split()[-1])
    assert loaded > 0 and refused > 0 and loaded + refused >= 25 * (len(sources) - 8)
    assert compiled >= len(sources) - 8
    blobs = sorted(str(p) for p in blobdir.glob("*.gir"))
    assert len(blobs) >= len(sources) - 8          # library files compile to no cinematic at all
    out = run([fuzz, "20261111", "200", *blobs])
    assert int(out.split()[0]) == 200 * len(blobs)
