"""The benchmark cinematics and their synthetic inputs.

Sources are inline because the machine that renders has no copy of the reference tree: C1, C3, C4 and C5 are the
reference's examples/001-hello.game, 007-showcase.game, 003-memory-trails.game and 010-gravity-particles.game
verbatim; C2 is the fbm showcase the headline metric is quoted on; C2B is examples/tutorial-05-noise.game without
its comments; C3B is C3 with a tonemap stage appended (the file has none). Frame k of every workload has u_time = k / 60 s, mouse = 0 and resolution = (w, h)."""
import numpy as np

C1_HELLO = '''cinematic "hello-glow" {
  layer main {
    circle(0.3) | glow(2.0) | tint(0.831, 0.686, 0.216)
  }
}'''
C2_FBM = '''cinematic "fbm-showcase" {
  layer {
    fn: domain_warp(0.1, 3.0) | fbm(3.0, octaves: 6, persistence: 0.5, lacunarity: 2.0) | glow(1.5) | tint(0.686, 0.878, 0.953)
  }
}'''
C2B_NOISE = '''cinematic "Noise" {
  layer clouds { fn: fbm(3.0, 4, 0.5, 2.0) | glow(1.5) | tint(frost) }
  layer flow { fn: simplex(4.0) | glow(1.2) | tint(ember) }
  layer cells { fn: voronoi(6.0) | glow(1.0) | tint(violet) }
}'''
C3_SHOWCASE = '''cinematic "Showcase" {
  layer {
    fn: spectrum(bass, mid, treble)
      | bloom(0.4, 2.0)
      | chromatic(0.6)
      | vignette(0.35)
      | grain(0.025)
    bass: 0.0 ~ audio.bass * 1.5
    mid: 0.0 ~ audio.mid * 1.5
    treble: 0.0 ~ audio.treble * 1.5
  }
}'''
# 007-showcase.game ends at grain; the same chain with the Reinhard tonemap (glsl.rs:473-478) appended
C3B_SHOWCASE_TONEMAP = C3_SHOWCASE.replace("| grain(0.025)", "| grain(0.025)\n      | tonemap(1.0)")
C4_TRAILS = '''cinematic "trails" {
  layer glow_core memory: 0.97 {
    translate(audio.bass, 0.0) | circle(0.1) | glow(2.5) | tint(1.0, 0.5, 0.2)
  }
}'''
C5_GRAVITY = '''cinematic "n-body" {
  gravity {
    damping: 0.995,
    bounds: reflect
  }
  layer stars {
    circle(0.005) | glow(1.0) | tint(0.8, 0.9, 1.0)
  }
}'''


# Algorithmic work per pixel (flop convention: add/sub/mul/div/min/max/abs/floor/fract/compare/select = 1,
# sin/cos/sqrt/atan = 1, mul+add pair = 2), summed over the stages of each cinematic; bytes per pixel = the RGBA8
# store (+ the previous frame's texel for memory cinematics). `frames` is the length of the whole job, `batch`
# the frames per launch (= ring slots) for stateless cinematics. `bound`: what limits the kernel - FP32 vector issue for
# the stateless cinematics (4 B per pixel is far below what HBM carries), HBM for the feedback path; `tables`: the kernel
# reads lattice tables instead of evaluating every hash, so its executed work is below the flop convention's count.
CONFIGS = {
    "c1": dict(src=C1_HELLO, w=1024, h=1024, flops_px=47, bytes_px=4, bound="valu", frames=600, batch=200,
               workload="examples/001-hello.game circle|glow|tint, 1024x1024 x 600 frames"),
    "c2": dict(src=C2_FBM, w=3840, h=2160, flops_px=1006, bytes_px=4, bound="valu", tables=True, frames=1000, batch=25,
               workload="fbm showcase: domain_warp(0.1,3.0)|fbm(3.0,octaves:6,0.5,2.0)|glow(1.5)|tint, 3840x2160 x 1000 frames, frame-sharded"),
    "c2b": dict(src=C2B_NOISE, w=3840, h=2160, flops_px=900, bytes_px=4, bound="valu", tables=True, frames=200, batch=25,
                workload="examples/tutorial-05-noise.game (fbm4 + simplex + voronoi, 3 layers), 3840x2160"),
    "c3": dict(src=C3_SHOWCASE, w=3840, h=2160, flops_px=95, bytes_px=4, bound="valu", frames=100, batch=25,
               workload="examples/007-showcase.game spectrum|bloom|chromatic|vignette|grain, 3840x2160 x 100 frames"),
    "c3b": dict(src=C3B_SHOWCASE_TONEMAP, w=3840, h=2160, flops_px=102, bytes_px=4, bound="valu", frames=100, batch=25,
                workload="examples/007-showcase.game with | tonemap(1.0) appended, 3840x2160 x 100 frames"),
    "c4": dict(src=C4_TRAILS, w=7680, h=4320, flops_px=59, bytes_px=8, bound="hbm", frames=300, batch=300,
               workload="examples/003-memory-trails.game feedback memory, 7680x4320 x 300 frames, frame-serial, band-sharded"),
}


def frame_rows(cin, key, frame_indices, w, h):
    """Uniform rows of the given frames of workload `key`: u_time = k / 60; c3 holds the audio bands at
    (.8, .6, .4, energy .6); c4 drives audio_bass(k) = .5 + .5 sin(2 pi .5 k / 60)."""
    ks = np.asarray(frame_indices, dtype=np.float64)
    audio = None
    if key == "c4":
        audio = {"bass": 0.5 + 0.5 * np.sin(2 * np.pi * 0.5 * ks / 60.0)}
    elif key in ("c3", "c3b"):
        audio = {"bass": 0.8, "mid": 0.6, "treble": 0.4, "energy": 0.6}
    return cin.uniform_rows(ks / 60.0, w, h, audio=audio)
