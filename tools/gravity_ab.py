#!/usr/bin/env python3
"""Times the gravity step kernel builds side by side on the GPU box: particles per lane (1 / 2 / 4) x exact / fast
arithmetic at several particle counts.

    python tools/gravity_ab.py [n ...]        (default 262144 1048576)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest as T  # noqa: E402

game = T.load_product()
cin = game.compile(T.C5_GRAVITY).cinematics[0]
for n in [int(a) for a in sys.argv[1:]] or [262144, 1048576]:
    for fast in (False, True):
        for ppl in ("1", "2", "4"):
            os.environ["GAME_HIP_GRAV_PPL"] = ppl
            r = cin.renderer(16, 16, gravity_fast=fast)
            r.particles_seed(n)
            r.particles_step(1 / 60, 1)
            r.sync()
            steps = 2 if n > 500000 else 8
            r.particles_step(1 / 60, steps)
            ms, launches = r.last_kernel_ms()
            r.close()
            per = ms / launches
            pairs = float(n) * (n - 1)
            print(f"n={n:8d} {'fast ' if fast else 'exact'} ppl={ppl}  {per:9.3f} ms/step  {pairs / per / 1e9:8.1f} Gpair/s  "
                  f"{20 * pairs / per / 1e9 / 157.3e3:.3f} of peak by the 20-flop convention", flush=True)
