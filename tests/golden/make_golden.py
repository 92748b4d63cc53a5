"""Regenerates the regression fixtures in this directory from the CPU oracle.

These are NOT reference outputs: the reference cannot execute pixels and ships no golden images.
They pin the oracle against accidental change; run from the repository root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest as T  # noqa: E402
from gallery import GALLERY  # noqa: E402
from test_oracle import GOLDEN_CASES  # noqa: E402

game = T.load_product()
oracle = T.Oracle()


def frame(src, w, h, t, audio, semantics="glsl"):
    cin = game.compile(src).cinematics[0]
    rows = cin.uniform_rows([t], w, h, audio=audio)
    return oracle.render(cin.blob, w, h, rows[0], threads=4, semantics=semantics)


for name, (src, w, h, t, audio) in GOLDEN_CASES.items():
    np.save(os.path.join(HERE, name + ".npy"), frame(src, w, h, t, audio))
np.savez_compressed(os.path.join(HERE, "gallery.npz"),
                    **{n: frame(s, 48, 27, 2.5, {"bass": 0.7, "mid": 0.2, "treble": 0.9}) for n, s in GALLERY.items()})
np.savez_compressed(os.path.join(HERE, "gallery_wgsl.npz"),
                    **{n: frame(s, 48, 27, 2.5, {"bass": 0.7, "mid": 0.2, "treble": 0.9}, "wgsl") for n, s in GALLERY.items()})
src = 'cinematic "t" { gravity { rule: 0.01 / (dist * dist + 0.01) * sin(time * 3.0 + 1.0), damping: 0.98, bounds: wrap } layer s { circle() | glow() } }'
cin = game.compile(src).cinematics[0]
p, _ = oracle.gravity(cin.blob, oracle.seed_particles(257), 1 / 60, 5, threads=4)
np.save(os.path.join(HERE, "gravity_257.npy"), p)
print("wrote goldens to", HERE)

# The particle workload at its parity size: 65 536 particles of examples/010-gravity-particles.game (default rule 1.0,
# damping 0.995, reflect), SplitMix64 seed, 100 steps of 1/60 s. About five minutes on eight threads (the oracle's
# blocked constant-law loop); skipped unless asked for, because the other fixtures take seconds.
if "--gravity-65536" in sys.argv:
    cin = game.compile(T.C5_GRAVITY).cinematics[0]
    p, t = oracle.gravity(cin.blob, oracle.seed_particles(65536), 1 / 60, 100, threads=os.cpu_count() or 1)
    np.save(os.path.join(HERE, "gravity_65536_100.npy"), p)
    print("wrote gravity_65536_100.npy, sim time", t)
