#!/usr/bin/env python3
"""One-off: random force laws (gravity.rs:10-41 grammar: dist, time, + - * / ^ > <, scalar functions) on random
particles, GPU step against the oracle's.  usage: tools/gravity_fuzz.py FIRST LAST"""
import os, random, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import conftest as T
g = T.load_product()
oracle = T.Oracle()
F1 = ["sin", "cos", "abs", "sqrt", "exp", "log", "floor", "ceil", "fract", "sign", "tan"]
F2 = ["min", "max", "pow", "step"]
F3 = ["clamp", "mix", "smoothstep"]


def expr(rng, depth):
    r = rng.random()
    if depth <= 0 or r < 0.25:
        return rng.choice(["dist", "dist", "time", f"{rng.uniform(0.01, 3.0):.4f}", f"{rng.uniform(-2.0, 2.0):.4f}"])
    if r < 0.6:
        return f"({expr(rng, depth - 1)} {rng.choice(['+', '-', '*', '/', '^', '>', '<'])} {expr(rng, depth - 1)})"
    if r < 0.8:
        return f"{rng.choice(F1)}({expr(rng, depth - 1)})"
    if r < 0.9:
        return f"{rng.choice(F2)}({expr(rng, depth - 1)}, {expr(rng, depth - 1)})"
    return f"{rng.choice(F3)}({expr(rng, depth - 1)}, {expr(rng, depth - 1)}, {expr(rng, depth - 1)})"


def same(a, b):
    return bool((((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))).all())


first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, last):
    rng = random.Random(seed)
    law = expr(rng, 4)
    bounds = rng.choice(["reflect", "wrap", "none"])
    src = 'cinematic "t" { gravity { rule: %s, damping: %.3f, bounds: %s } layer s { circle() | glow() } }' % (law, rng.uniform(0.9, 1.0), bounds)
    try:
        cin = g.compile(src).cinematics[0]
    except Exception as e:
        print("seed", seed, "compile failed", str(e)[:120], flush=True)
        continue
    n = rng.choice([300, 512, 777])
    p0 = oracle.seed_particles(n, seed=seed + 1)
    r = cin.renderer(16, 16)
    r.particles_init(p0[:, :2], p0[:, 2:])
    r.particles_step(1 / 60, 3)
    pos, vel = r.particles_read()
    r.close()
    want, _ = oracle.gravity(cin.blob, p0, 1 / 60, 3)
    if not (same(pos, np.ascontiguousarray(want[:, :2])) and same(vel, np.ascontiguousarray(want[:, 2:]))):
        bad += 1
        print("MISMATCH seed", seed, bounds, law, flush=True)
    if seed % 20 == 0:
        print("seed", seed, "mismatches so far", bad, flush=True)
print("done", first, last, "mismatches", bad)
