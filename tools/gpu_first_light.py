"""First GPU run: hiprtc on the box, parity of a few cinematics against the oracle, rough throughput."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest as T
g = T.load_product()
o = T.Oracle()
print("devices", g.device_count(), flush=True)
for which in (0, 1, 2):
    t0 = time.time(); m, f = g.check_exhaustive(which); print("exhaustive", which, "mismatches", m, "first", hex(f), "%.2fs" % (time.time() - t0), flush=True)
def parity(name, src, w, h, times, audio=None):
    cin = g.compile(src).cinematics[0]
    rows = cin.uniform_rows(times, w, h, audio=audio)
    r = cin.renderer(w, h)
    got = r.render(rows)
    want = o.render_sequence(cin.blob, w, h, rows, cin.uses_memory)
    diff = np.abs(got.astype(int) - want.astype(int))
    print(f"{name:28s} {w}x{h}x{len(times)} max|d|={diff.max()} mismatching={np.count_nonzero(diff.max(axis=-1))} flags={cin.flags}", flush=True)
    r.close()
parity("C1", T.C1_HELLO, 256, 256, [0.0, 0.5])
parity("C2", T.C2_FBM, 320, 180, [0.0])
parity("C2b", T.C2B_NOISE, 320, 180, [0.0, 1.0])
parity("C3", T.C3_SHOWCASE, 320, 180, [0.0, 1.0 / 60, 7.3], audio={"bass": .8, "mid": .6, "treble": .4, "energy": .6})
parity("C4", T.C4_TRAILS, 256, 144, [k / 60 for k in range(12)], audio={"bass": [0.5 + 0.5 * np.sin(2 * np.pi * 0.5 * k / 60) for k in range(12)]})
# throughput, C2 at 4K
cin = g.compile(T.C2_FBM).cinematics[0]
W, H = 3840, 2160
r = cin.renderer(W, H); r.set_ring(8)
rows = cin.uniform_rows([k / 60 for k in range(8)], W, H)
r.render(rows, readback=False); r.sync()
t0 = time.time(); n = 5
for _ in range(n): r.render(rows, readback=False)
r.sync(); dt = time.time() - t0
ms, launches = r.last_kernel_ms()
print(f"C2 4K: {n*8*W*H/dt/1e6:.0f} Mpixel/s wall; last launch {ms:.3f} ms for 8 frames -> {8*W*H/ms/1e3:.0f} Mpixel/s", flush=True)
