"""ctypes view of oracle/libgame_oracle.so, the CPU restatement of the render path.

Test infrastructure: only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this
module. The product (game-engine_amd/) never does; it fails when its HIP library or a GPU is missing."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Oracle:
    """ctypes view of oracle/libgame_oracle.so."""

    def __init__(self):
        so = os.path.join(HERE, "libgame_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", HERE])
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
        L.game_oracle_gravity_range.restype = c.c_int
        L.game_oracle_gravity_range.argtypes = [c.c_void_p, c.c_size_t, c.c_uint32, c.c_void_p, c.c_float, c.c_float, c.c_uint32,
                                                c.c_uint32, c.c_void_p, c.c_int]
        L.game_oracle_set_gravity_blocked.restype = None
        L.game_oracle_set_gravity_blocked.argtypes = [c.c_int]
        L.game_oracle_seed_particles.restype = None
        L.game_oracle_seed_particles.argtypes = [c.c_uint32, c.c_uint64, c.c_void_p]
        for name, nargs in (("sin", 1), ("cos", 1), ("atan2", 2), ("fract", 1), ("hash2", 2), ("noise2", 2),
                            ("simplex2d", 2), ("voronoi2d", 3), ("pow", 2), ("exp", 1), ("log", 1), ("rsqrt", 1)):
            fn = getattr(L, "game_oracle_" + name)
            fn.restype = c.c_float
            fn.argtypes = [c.c_float] * nargs
        L.game_oracle_rsqrt_array.restype = None
        L.game_oracle_rsqrt_array.argtypes = [c.c_uint32, c.c_void_p, c.c_void_p]
        L.game_oracle_rsqrt_selfcheck.restype = c.c_uint32
        L.game_oracle_rsqrt_selfcheck.argtypes = [c.POINTER(c.c_uint32)]
        L.game_oracle_fbm2.restype = c.c_float
        L.game_oracle_fbm2.argtypes = [c.c_float, c.c_float, c.c_int, c.c_float, c.c_float]
        self.L = L

    def render(self, blob, w, h, row, prev=None, row0=0, rows=None, threads=8, semantics="glsl", derived=None,
               lens_post=False, animate=False, project=False):
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
                                           {"glsl": 0, "wgsl": 1}[semantics] | (4 if lens_post else 0) | (2 if animate or derived is not None else 0) | (8 if project else 0),
                                           None if derived is None or derived.size == 0 else derived.ctypes.data)
        assert rc == 0, "oracle rejected the IR blob or arguments"
        return out

    def render_sequence(self, blob, w, h, rows_table, uses_memory, row0=0, rows=None, threads=8, semantics="glsl",
                        derived=None, lens_post=False, animate=False, project=False):
        """derived: None, or one row of expression-argument values per frame."""
        frames, prev = [], None
        for k, r in enumerate(rows_table):
            f = self.render(blob, w, h, r, prev=prev if uses_memory else None, row0=row0, rows=rows, threads=threads,
                            semantics=semantics, derived=None if derived is None else derived[k], lens_post=lens_post, animate=animate, project=project)
            frames.append(f)
            prev = f
        return np.stack(frames)

    def seed_particles(self, n, seed=0x9E3779B97F4A7C15):
        p = np.zeros((n, 4), dtype=np.float32)
        self.L.game_oracle_seed_particles(n, seed, p.ctypes.data)
        return p

    def gravity_range(self, blob, particles, dt, time, first, count, threads=8, blocked=True):
        """One step of particles [first, first + count) against all of `particles` ((n, 4) float32, not copied);
        `time` is the dispatch clock of the step (the clock before it + dt). Returns (count, 4)."""
        self.L.game_oracle_set_gravity_blocked(1 if blocked else 0)
        p = np.ascontiguousarray(particles, dtype=np.float32)
        out = np.empty((count, 4), dtype=np.float32)
        rc = self.L.game_oracle_gravity_range(blob, len(blob), p.shape[0], p.ctypes.data, dt, time, first, count, out.ctypes.data, threads)
        assert rc == 0
        return out

    def gravity(self, blob, particles, dt, steps, sim_time=0.0, threads=8, blocked=True):
        """blocked=False forces the one-particle-at-a-time loop even for constant force laws."""
        self.L.game_oracle_set_gravity_blocked(1 if blocked else 0)
        p = np.ascontiguousarray(particles, dtype=np.float32).copy()
        t = ctypes.c_float(sim_time)
        rc = self.L.game_oracle_gravity(blob, len(blob), p.shape[0], p.ctypes.data, dt, steps, ctypes.byref(t), threads)
        assert rc == 0
        return p, t.value
