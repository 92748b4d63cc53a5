import sys, os
import numpy as np
sys.path.insert(0, "tests")
import conftest as T
game = T.load_product()
laws = {"default": T.C5_GRAVITY,
        "soft": 'cinematic "t" { gravity { rule: 0.00002 / (dist * dist + 0.01), damping: 0.995, bounds: reflect } layer s { circle() | glow() } }'}
for name, src in laws.items():
    cin = game.compile(src).cinematics[0]
    n = 65536
    a = cin.renderer(16, 16); b = cin.renderer(16, 16, gravity_fast=True)
    a.particles_seed(n); b.particles_seed(n)
    done = 0
    for upto in (1, 10, 100):
        a.particles_step(1 / 60, upto - done); b.particles_step(1 / 60, upto - done); done = upto
        pa, va = a.particles_read(); pb, vb = b.particles_read()
        rel = np.abs(pb - pa) / np.maximum(np.abs(pa), 1e-3)
        print(name, "steps", upto, "max rel", rel.max(), "frac > 1e-4", (rel > 1e-4).mean(), "at wall", (np.abs(pa) == 1).mean(), "vmax", np.abs(va).max(), flush=True)
    a.close(); b.close()
