#!/usr/bin/env python3
"""A/B of kernel build variants on the GPU, interleaved rounds in one process.
usage: tools/ab_flags.py <config> "<flags A>" "<flags B>" ...   ("" = default build)
All variants live in one process and share the GPU; memory cinematics run with a 25-slot ring here. Differences of a
few percent have pointed the wrong way with this tool: confirm them with `bench.py --config <c>` under the same
GAME_HIP_* settings, alternating on one box."""
import os, sys, statistics
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest as T
g = T.load_product()
key = sys.argv[1]
variants = sys.argv[2:] or [""]
HOOKS = ("PPL=", "RPL=", "ROWLDS=", "WG=", "NTLOAD=", "SPECULATE=")     # GAME_HIP_<name> shape hooks of the runtime
cfg = g.workloads.CONFIGS[key]
cin = g.compile(cfg["src"]).cinematics[0]
w, h, batch = cfg["w"], cfg["h"], 25
rows = cin.uniform_rows([k / 60 for k in range(batch)], w, h, audio={"bass": 0.8, "mid": 0.6, "treble": 0.4})
rs = []
ref = None
for v in variants:
    # a variant is hiprtc flags, optionally prefixed by shape settings: "PPL=n,RPL=m;flags" (pixels / rows per lane)
    head, _, fl = v.partition(";") if v.startswith(HOOKS) else ("", "", v)
    os.environ["GAME_HIP_FLAGS"] = fl
    for name in (h[:-1] for h in HOOKS):
        os.environ.pop("GAME_HIP_" + name, None)
    for item in filter(None, head.split(",")):
        name, _, val = item.partition("=")
        os.environ["GAME_HIP_" + name] = val
    r = cin.renderer(w, h); r.set_ring(max(2, batch))
    band = r.render_band(rows[:1], h // 2, 8)[0]
    if ref is None: ref = band
    same = np.array_equal(band, ref)
    r.reset_memory()
    rs.append((v, r, same))
times = {v: [] for v in variants}
for rnd in range(7):
    for v, r, _ in rs:
        r.render(rows, readback=False)
        ms, n = r.last_kernel_ms()
        times[v].append(ms / (1 if not cin.uses_memory else 1))
for v, r, same in rs:
    t = times[v][1:]
    med = statistics.median(t)
    print(f"{key} [{v or 'default'}]: median {med:.3f} ms min {min(t):.3f} ms -> {batch*w*h/med/1e3:.0f} Mpixel/s  same_pixels_as_first={same}", flush=True)
