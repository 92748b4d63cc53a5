#!/usr/bin/env python3
"""Frame filter throughput: the feedback cinematic at 8K with and without the blur (radius 1..4), frames left in HBM.
Run it under `rocprofv3 --kernel-trace --stats` for the blur kernel's own duration (the difference printed here also
holds the launch gap of the second kernel); its algorithmic traffic is one read and one write of the frame, 265 MB."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest as T  # noqa: E402

game = T.load_product()
W = game.workloads
cin = game.compile(W.C4_TRAILS).cinematics[0]
w, h = 7680, 4320
radii = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3, 4]     # e.g. "0,4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rows = W.frame_rows(cin, "c4", range(n), w, h)
base = None
for radius in radii:
    r = cin.renderer(w, h)
    r.set_blur(radius)
    r.render(rows[:10], readback=False)
    r.sync()
    r.render(rows, readback=False)
    ms, launches = r.last_kernel_ms()
    r.close()
    per = ms / n
    if radius == 0 or base is None:
        base = per if radius == 0 else 0.0
        print(f"no filter: {per * 1e3:.1f} us/frame", flush=True)
    else:
        blur = per - base
        print(f"radius {radius}: {per * 1e3:.1f} us/frame, blur alone ~{blur * 1e3:.1f} us = {2 * w * h * 4 / blur / 1e6:.0f} GB/s of {w * h * 8 / 1e6:.0f} MB", flush=True)
