#!/usr/bin/env python3
"""How far do the pixels move when sin / cos return a neighbouring float?

GLSL ES and WGSL do not fix the last bits of sin and cos, and the reference's real executor (a browser's shader
compiler) is not available, so "the same pixels" can only mean: the same as this build's fixed algorithm. This tool
states the other half - which cinematics would still agree with ANY correct sine to within the image tolerance
(1/255 per channel, PSNR >= 50 dB) and which cannot, because they feed a sine into fract(sin(x) * 43758.5453), where
one unit in the last place of the sine moves the hash by up to 43758 * 2^-24 = 0.0026 of its range... and far more once
the argument is large. Every gallery cinematic (both wordings) and the benchmark cinematics c2b / c3 are rendered
three times: as built, and with the kernels compiled with -DGD_SIN_NUDGE=1 / =-1 (sine and cosine one unit in the last
place away from / toward zero; lattice tables off so that table entries are computed with the nudged sine too).
Writes a markdown table.  usage (GPU box): python tools/sin_sensitivity.py [out.md]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest as T  # noqa: E402
from gallery import GALLERY  # noqa: E402

game = T.load_product()
W = game.workloads
SIN_BUILTINS = {"glsl": ("star", "polygon", "concentric_waves", "rotate", "twist", "repeat", "palette", "grain", "scanlines", "glitch", "voronoi"),
                "wgsl": ("star", "polygon", "concentric_waves", "rotate", "twist", "palette", "grain", "scanlines", "glitch", "simplex")}


def render(src, w, h, semantics, flags, times, audio):
    os.environ["GAME_HIP_FLAGS"] = flags
    if flags:
        os.environ["GAME_HIP_NO_LUT"] = "1"
    else:
        os.environ.pop("GAME_HIP_NO_LUT", None)
    cin = game.compile(src).cinematics[0]
    r = cin.renderer(w, h, semantics=semantics)
    rows = cin.uniform_rows(times, w, h, audio=audio)
    out = r.render(rows)
    r.close()
    os.environ.pop("GAME_HIP_FLAGS")
    os.environ.pop("GAME_HIP_NO_LUT", None)
    return cin, out


def compare(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    mse = float((d.astype(np.float64) ** 2).mean())
    psnr = float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)
    return int(d.max()), psnr, float((d > 1).mean())


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "sin_sensitivity.md")
    cases = [(n, s, 192, 108) for n, s in GALLERY.items()]
    cases += [("c2b tutorial-05-noise", W.C2B_NOISE, 960, 540), ("c3 showcase", W.C3_SHOWCASE, 960, 540), ("c2 fbm showcase", W.C2_FBM, 960, 540),
              ("c1 hello", W.C1_HELLO, 512, 512), ("c4 memory trails (frame 8)", W.C4_TRAILS, 480, 270)]
    audio = {"bass": 0.7, "mid": 0.2, "treble": 0.9}
    lines = ["| cinematic | wording | sine-dependent builtins | max abs diff (of 255) +1 / -1 ulp | PSNR dB +1 / -1 | pixels off by > 1 | inside 1/255 and 50 dB |",
             "|---|---|---|---|---|---|---|"]
    outside = []
    for name, src, w, h in cases:
        for sem in ("glsl", "wgsl"):
            times = [k / 30.0 for k in range(9)] if "memory" in name else [2.5]
            cin, base = render(src, w, h, sem, "", times, audio)
            res = []
            for nudge in ("1", "-1"):
                _, other = render(src, w, h, sem, "-DGD_SIN_NUDGE=" + nudge, times, audio)
                res.append(compare(base[-1], other[-1]))
            used = sorted({st["name"] for layer in cin.ir["layers"] for st in layer["stages"]} & set(SIN_BUILTINS[sem]))
            ok = all(m <= 1 and p >= 50.0 for m, p, _ in res)
            fmt = lambda p: "identical" if p == float("inf") else f"{p:.1f}"
            lines.append(f"| {name} | {sem} | {', '.join(used) or '-'} | {res[0][0]} / {res[1][0]} | {fmt(res[0][1])} / {fmt(res[1][1])} | "
                         f"{max(res[0][2], res[1][2]) * 100:.2f} % | {'yes' if ok else '**no**'} |")
            if not ok:
                outside.append((name, sem, used))
            print(lines[-1], flush=True)
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n\noutside the tolerance: " + "; ".join(f"{n} ({s}: {', '.join(u)})" for n, s, u in outside) + "\n")
    print("wrote", out_path)


if __name__ == "__main__":
    main()
