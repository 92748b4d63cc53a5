#!/usr/bin/env python3
"""Times build/launch variants of the frame kernel of each bench config in one process (all variants produce
the same pixels; the first rows of each are compared).  usage: tools/variant_sweep.py [config ...]"""
import itertools, os, statistics, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
import conftest as T
import bench
g = T.load_product()
keys = sys.argv[1:] or ["c1", "c2", "c2b", "c3", "c4"]
for key in keys:
    cfg = bench.CONFIGS[key]
    cin = g.compile(cfg["src"]).cinematics[0]
    w, h = cfg["w"], cfg["h"]
    batch = 1 if cin.uses_memory else {"c1": 100}.get(key, 10)
    rows = cin.uniform_rows([k / 60 for k in range(max(batch, 8))], w, h, audio={"bass": 0.8, "mid": 0.6, "treble": 0.4})
    ref = None
    results = []
    for guards, ppl, lds in itertools.product(["", "-DGD_WAVE_GUARDS"], [1, 4, 8], [0, 40960, 81920]):
        os.environ["GAME_HIP_FLAGS"] = guards
        os.environ["GAME_HIP_PPL"] = str(ppl)
        os.environ["GAME_HIP_LDS_BYTES"] = str(lds)
        r = cin.renderer(w, h)
        r.set_ring(max(2, batch))
        band = r.render_band(rows[:1], h // 2, 4)[0]
        if ref is None:
            ref = band
        same = np.array_equal(band, ref)
        r.reset_memory()
        ts = []
        for rep in range(6):
            if cin.uses_memory:
                r.render(rows[:8], readback=False)
                ms, n = r.last_kernel_ms()
                ts.append(ms / 8)
            else:
                r.render(rows[:batch], readback=False)
                ms, n = r.last_kernel_ms()
                ts.append(ms / batch)
        r.close()
        med = statistics.median(ts[1:])
        results.append((med, guards or "lane-guards", ppl, lds, same))
        print(f"{key} {guards or 'lane-guards':18s} ppl={ppl} lds={lds:6d}: {med*1e3:9.2f} us/frame {w*h/med/1e3:10.0f} Mpixel/s same={same}", flush=True)
    best = min(results)
    print(f"== {key} best: {best[1]} ppl={best[2]} lds={best[3]} {w*h/best[0]/1e3:.0f} Mpixel/s", flush=True)
