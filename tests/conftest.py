"""Shared test plumbing: loads the product binding (game-engine_amd/__init__.py -> libgame_hip.so)
and the CPU oracle (oracle/libgame_oracle.so, test infrastructure only)."""
import ctypes
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_product():
    if "game_engine_amd" in sys.modules:
        return sys.modules["game_engine_amd"]
    path = os.path.join(ROOT, "game-engine_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("game_engine_amd", path,
                                                  submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["game_engine_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


class Oracle:
    """ctypes view of oracle/libgame_oracle.so."""

    def __init__(self):
        so = os.path.join(ROOT, "oracle", "libgame_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = ctypes.CDLL(so)
        c = ctypes
        L.game_oracle_render.restype = c.c_int
        L.game_oracle_render.argtypes = [c.c_void_p, c.c_size_t, c.c_uint32, c.c_uint32, c.c_uint32, c.c_uint32,
                                         c.c_void_p, c.c_void_p, c.c_void_p, c.c_int]
        L.game_oracle_render_ex.restype = c.c_int
        L.game_oracle_render_ex.argtypes = L.game_oracle_render.argtypes + [c.c_int]
        L.game_oracle_render_ex2.restype = c.c_int
        L.game_oracle_render_ex2.argtypes = L.game_oracle_render.argtypes + [c.c_int, c.c_void_p]
        L.game_oracle_gravity.restype = c.c_int
        L.game_oracle_gravity.argtypes = [c.c_void_p, c.c_size_t, c.c_uint32, c.c_void_p, c.c_float, c.c_uint32,
                                          c.POINTER(c.c_float), c.c_int]
        L.game_oracle_seed_particles.restype = None
        L.game_oracle_seed_particles.argtypes = [c.c_uint32, c.c_uint64, c.c_void_p]
        for name, nargs in (("sin", 1), ("cos", 1), ("atan2", 2), ("fract", 1), ("hash2", 2), ("noise2", 2),
                            ("simplex2d", 2), ("voronoi2d", 3), ("pow", 2), ("exp", 1), ("log", 1)):
            fn = getattr(L, "game_oracle_" + name)
            fn.restype = c.c_float
            fn.argtypes = [c.c_float] * nargs
        L.game_oracle_fbm2.restype = c.c_float
        L.game_oracle_fbm2.argtypes = [c.c_float, c.c_float, c.c_int, c.c_float, c.c_float]
        self.L = L

    def render(self, blob, w, h, row, prev=None, row0=0, rows=None, threads=8, semantics="glsl", derived=None,
               lens_post=False, animate=False):
        rows = h if rows is None else rows
        row = np.ascontiguousarray(row, dtype=np.float32)
        out = np.zeros((rows, w, 4), dtype=np.uint8)
        if prev is not None:
            prev = np.ascontiguousarray(prev, dtype=np.uint8)
            assert prev.shape == out.shape
        if derived is not None:
            derived = np.ascontiguousarray(derived, dtype=np.float32)
        rc = self.L.game_oracle_render_ex2(blob, len(blob), w, h, row0, rows, row.ctypes.data,
                                           None if prev is None else prev.ctypes.data, out.ctypes.data, threads,
                                           {"glsl": 0, "wgsl": 1}[semantics] | (4 if lens_post else 0) | (2 if animate or derived is not None else 0),
                                           None if derived is None or derived.size == 0 else derived.ctypes.data)
        assert rc == 0, "oracle rejected the IR blob or arguments"
        return out

    def render_sequence(self, blob, w, h, rows_table, uses_memory, row0=0, rows=None, threads=8, semantics="glsl",
                        derived=None, lens_post=False, animate=False):
        """derived: None, or one row of expression-argument values per frame."""
        frames, prev = [], None
        for k, r in enumerate(rows_table):
            f = self.render(blob, w, h, r, prev=prev if uses_memory else None, row0=row0, rows=rows, threads=threads,
                            semantics=semantics, derived=None if derived is None else derived[k], lens_post=lens_post, animate=animate)
            frames.append(f)
            prev = f
        return np.stack(frames)

    def seed_particles(self, n, seed=0x9E3779B97F4A7C15):
        p = np.zeros((n, 4), dtype=np.float32)
        self.L.game_oracle_seed_particles(n, seed, p.ctypes.data)
        return p

    def gravity(self, blob, particles, dt, steps, sim_time=0.0, threads=8):
        p = np.ascontiguousarray(particles, dtype=np.float32).copy()
        t = ctypes.c_float(sim_time)
        rc = self.L.game_oracle_gravity(blob, len(blob), p.shape[0], p.ctypes.data, dt, steps, ctypes.byref(t), threads)
        assert rc == 0
        return p, t.value


@pytest.fixture(scope="session")
def game():
    return load_product()


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


REFERENCE = "/root/reference"


def reference_available():
    return os.path.isdir(REFERENCE)


# The five benchmark cinematics, inline (the GPU box has no copy of the reference tree).
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
