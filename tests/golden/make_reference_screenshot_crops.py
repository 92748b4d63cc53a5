#!/usr/bin/env python3
"""Writes tests/golden/reference_screenshots.npz: the canvas regions of three screenshots the reference ships under
screenshots/ - its dev server showing presets/dashboard-gauge.game (01-load.png) and the "layered-scene" cinematic in
the first, un-animated moment of its arc timeline (arc-01-load.png), each as the 200 x 200 component canvas and as the
preview pane, and the component canvas of the same cinematic 17 ms into the second moment of that timeline
(arc-09-params.png). The other five screenshots show the same renders again (06-slider.png: identical canvas) or text
tabs and overlays. They are the only pixels in the reference tree that its own emitted shaders produced (in a browser, by an
older generation of the compiler: dist/dashboard-gauge.js and dist/layered-scene.js hold that generation's shader text).
Data only: RGB bytes cut out of the PNG files, no scaling or filtering.

    python tests/golden/make_reference_screenshot_crops.py        (needs /root/reference and PIL)
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SHOTS = "/root/reference/screenshots"

# name: (file, x0, y0, width, height) of a canvas inside the 1400 x 900 screenshot
CROPS = {
    "gauge_component": ("01-load.png", 951, 266, 200, 200),
    "gauge_pane": ("01-load.png", 0, 36, 698, 864),
    "layered_component": ("arc-01-load.png", 951, 264, 200, 200),
    "layered_pane": ("arc-01-load.png", 0, 36, 698, 794),
    # the same cinematic two seconds into its arc timeline, just after the "ignite" moment has begun (arc-09-params.png)
    "layered_component_ignite": ("arc-09-params.png", 951, 264, 200, 200),
}


def main():
    out = {}
    for name, (f, x0, y0, w, h) in CROPS.items():
        rgb = np.asarray(Image.open(os.path.join(SHOTS, f)).convert("RGB"))
        out[name] = np.ascontiguousarray(rgb[y0:y0 + h, x0:x0 + w])
        assert out[name].shape == (h, w, 3)
    np.savez_compressed(os.path.join(HERE, "reference_screenshots.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
