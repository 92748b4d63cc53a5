#!/usr/bin/env python3
"""One-off wide run of the seeded parity fuzzer (tests/test_fuzz_parity.py) on the GPU.
usage: tools/fuzz_sweep.py FIRST LAST   -> prints every seed whose GPU frames differ from the oracle's."""
import os, random, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import conftest as T
import test_fuzz_parity as F
g = T.load_product()
oracle = T.Oracle()
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, last):
    rng = random.Random(seed)
    ext = seed % 3 == 0
    src, _ = F.make_cinematic(g, rng, expressions=ext)
    w, h = rng.choice([(96, 54), (100, 37), (64, 64), (33, 17)])
    try:
        cin, rows = F.frames_and_rows(g, src, rng, w, h)
    except Exception as e:
        print("seed", seed, "compile failed:", str(e)[:200], flush=True)
        bad += 1
        continue
    derived = cin.eval_derived(rows) if ext else None
    for sem in ("glsl", "wgsl"):
        r = cin.renderer(w, h, semantics=sem, animate=ext, lens_post=ext, project=ext)
        got = r.render(rows)
        r.close()
        want = oracle.render_sequence(cin.blob, w, h, rows, cin.uses_memory, semantics=sem, derived=derived, lens_post=ext, animate=ext, project=ext)
        if not np.array_equal(got, want):
            bad += 1
            print("MISMATCH seed", seed, sem, int((got != want).sum()), "values\n" + src, flush=True)
    if seed % 50 == 0:
        print("seed", seed, "ok so far, mismatches:", bad, flush=True)
print("done", first, last, "mismatches", bad)
