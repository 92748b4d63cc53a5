#!/usr/bin/env python3
"""PCIe-inclusive rate of the render boundary: frames rendered AND copied to host memory
(game_hip_render with a host buffer), for the note next to the HBM-resident benchmark value."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conftest as T, bench
g = T.load_product()
key = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = bench.CONFIGS[key]
cin = g.compile(cfg["src"]).cinematics[0]
w, h, batch = cfg["w"], cfg["h"], 8
r = cin.renderer(w, h); r.set_ring(2 if cin.uses_memory else batch)
rows = cin.uniform_rows([k / 60 for k in range(batch)], w, h)
for pinned in (False, True):
    r.render(rows, pinned=pinned)       # warm-up
    t0 = time.perf_counter(); n = 3
    for _ in range(n): out = r.render(rows, pinned=pinned)
    dt = time.perf_counter() - t0
    print(f"{key} [{'page-locked' if pinned else 'pageable'} host buffer]: {n*batch} frames {w}x{h} rendered and read back in {dt:.3f} s -> "
          f"{n*batch*w*h/dt/1e6:.0f} Mpixel/s, {n*batch*w*h*4/dt/1e9:.2f} GB/s to host")
