"""ctypes binding of libgame_hip.so (C ABI: include/game_hip.h).

Thin host-side mirror of the reference's public API (game-compiler/src/lib.rs: compile, check, lex,
list_builtins) plus the GPU render path that replaces its browser runtime. Nothing here computes
pixels: every call goes to the shared library, and a missing library or a missing GPU raises.
"""
import ctypes
import importlib.util
import json
import os
import struct
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgame_hip.so")

BU_COUNT = 10  # built-in floats per frame row: time, 5 audio, res.xy, mouse.xy (runtime/helpers.rs:222-232)
BU_TIME, BU_BASS, BU_MID, BU_TREBLE, BU_ENERGY, BU_BEAT, BU_RES_X, BU_RES_Y, BU_MOUSE_X, BU_MOUSE_Y = range(10)


class GameError(RuntimeError):
    """A non-zero return from the library; .code is the GAME_E_* / E00x value."""

    def __init__(self, code, message, rendered=""):
        super().__init__(message)
        self.code = code
        self.rendered = rendered


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C game-engine_amd` (or __graft_entry__.build())")
    # One HIP runtime per process. PyTorch-ROCm ships a private copy of libamdhip64 that it asks for by a name
    # (libamdhip64.so) the system copy's soname (libamdhip64.so.7) does not satisfy, so a process that loads this
    # library first and torch later ends up with two runtimes, and the second finds no GPU. When torch is installed
    # but not imported yet, its copy is loaded here first: libgame_hip.so then binds to it by soname, and a later
    # `import torch` finds the same file already loaded. Without torch the system runtime is used.
    if "torch" not in sys.modules:
        spec = importlib.util.find_spec("torch")
        shared = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else ""
        if shared and os.path.exists(shared):
            ctypes.CDLL(shared, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, sz, u32, u64, f32 = c.c_void_p, c.c_size_t, c.c_uint32, c.c_uint64, c.c_float
    sig = {
        "game_last_error": (c.c_char_p, []),
        "game_last_error_rendered": (c.c_char_p, []),
        "game_front_compile": (c.c_int, [c.c_char_p, sz, c.POINTER(vp)]),
        "game_program_free": (None, [vp]),
        "game_program_cinematic_count": (sz, [vp]),
        "game_program_ir": (vp, [vp, sz]),
        "game_program_warning_count": (sz, [vp]),
        "game_program_warning": (c.c_char_p, [vp, sz]),
        "game_front_lex": (c.c_int, [c.c_char_p, sz, c.POINTER(sz), vp, sz]),
        "game_front_tokens": (c.c_char_p, [c.c_char_p, sz]),
        "game_front_compile_with_imports": (c.c_int, [c.c_char_p, sz, c.c_char_p, vp, sz, c.POINTER(vp)]),
        "game_front_ast_json": (c.c_char_p, [c.c_char_p, sz, c.c_int]),
        "game_builtin_count": (sz, []),
        "game_builtin_describe": (c.c_char_p, [sz]),
        "game_ir_from_blob": (c.c_int, [vp, sz, c.POINTER(vp)]),
        "game_ir_name": (c.c_char_p, [vp]),
        "game_ir_json": (c.c_char_p, [vp]),
        "game_ir_blob": (sz, [vp, c.POINTER(vp)]),
        "game_ir_uniform_count": (u32, [vp]),
        "game_ir_uniform_name": (c.c_char_p, [vp, u32]),
        "game_ir_uniform_default": (c.c_double, [vp, u32]),
        "game_ir_flags": (u32, [vp]),
        "game_ir_uses_memory": (c.c_int, [vp]),
        "game_ir_has_gravity": (c.c_int, [vp]),
        "game_ir_frame_floats": (u32, [vp]),
        "game_ir_default_uniforms": (None, [vp, f32, u32, u32, vp]),
        "game_ir_derived_count": (u32, [vp]),
        "game_ir_pixel_expr_count": (u32, [vp]),
        "game_ir_project_mode": (c.c_int, [vp]),
        "game_ir_eval_derived": (None, [vp, vp, vp]),
        "game_animator_create": (c.c_int, [vp, c.c_double, c.POINTER(vp)]),
        "game_animator_destroy": (None, [vp]),
        "game_animator_reset": (None, [vp]),
        "game_animator_step": (c.c_int, [vp, c.c_double, c.c_double, vp, u32, vp, vp, u32]),
        "game_ir_arc_count": (u32, [vp]),
        "game_ir_arc_update": (None, [vp, c.c_double, vp]),
        "game_hip_device_count": (c.c_int, [c.POINTER(c.c_int)]),
        "game_hip_create": (c.c_int, [vp, u32, u32, c.c_int, c.POINTER(vp)]),
        "game_hip_create_ex": (c.c_int, [vp, u32, u32, c.c_int, u32, c.POINTER(vp)]),
        "game_hip_destroy": (None, [vp]),
        "game_hip_set_ring": (c.c_int, [vp, u32]),
        "game_hip_render": (c.c_int, [vp, vp, u32, vp]),
        "game_hip_render_band": (c.c_int, [vp, vp, u32, u32, u32, vp]),
        "game_hip_render_into": (c.c_int, [vp, vp, u32, vp]),
        "game_hip_host_alloc": (vp, [sz]),
        "game_hip_host_free": (None, [vp]),
        "game_hip_set_blur": (c.c_int, [vp, u32]),
        "game_hip_render_raw_band": (c.c_int, [vp, vp, u32, u32]),
        "game_hip_raw_rows": (vp, [vp, u32]),
        "game_hip_blur_band": (c.c_int, [vp, u32, u32, vp]),
        "game_hip_reset_memory": (c.c_int, [vp]),
        "game_hip_set_memory": (c.c_int, [vp, vp, u32, u32]),
        "game_hip_particles_set_time": (c.c_int, [vp, c.c_float]),
        "game_hip_sync": (c.c_int, [vp]),
        "game_hip_timing_begin": (c.c_int, [vp]),
        "game_hip_timing_end": (c.c_int, [vp, c.POINTER(c.c_float), c.POINTER(u32)]),
        "game_hip_frame_ptr": (vp, [vp, u32]),
        "game_hip_ring_frames": (u32, [vp]),
        "game_hip_last_kernel_ms": (c.c_int, [vp, c.POINTER(f32), c.POINTER(u32)]),
        "game_hip_kernel_source": (c.c_char_p, [vp]),
        "game_hip_device_header_source": (c.c_char_p, []),
        "game_hip_gravity_source": (c.c_char_p, [vp]),
        "game_hip_stream": (vp, [vp]),
        "game_hip_setup_ms": (c.c_int, [vp, c.POINTER(f32), c.POINTER(f32)]),
        "game_hip_compile_only": (c.c_int, [vp, c.POINTER(sz)]),
        "game_hip_compile_only_ex": (c.c_int, [vp, u32, c.POINTER(sz)]),
        "game_hip_dump_code": (c.c_int, [vp, c.c_char_p]),
        "game_hip_particles_init": (c.c_int, [vp, u32, vp, vp]),
        "game_hip_particles_seed": (c.c_int, [vp, u32, u64]),
        "game_hip_particles_step": (c.c_int, [vp, f32, u32, u32, u32]),
        "game_hip_particles_read": (c.c_int, [vp, vp, vp]),
        "game_hip_particles_positions": (c.c_int, [vp, c.POINTER(vp), c.POINTER(vp)]),
        "game_hip_particles_time": (c.c_int, [vp, c.POINTER(f32)]),
        "game_hip_shutdown": (c.c_int, []),
        "game_hip_eval": (c.c_int, [c.c_int, u32, vp, vp, vp]),
        "game_hip_check_exhaustive": (c.c_int, [c.c_int, c.POINTER(u64), c.POINTER(u32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError here == a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


lib, EXPORTED_SYMBOLS = _load()


SEMANTICS = {"glsl": 0, "wgsl": 1}   # GAME_HIP_SEMANTICS_WGSL
ANIMATE_ARGS = 2                     # GAME_HIP_ANIMATE_ARGS
LENS_POST = 4                        # GAME_HIP_LENS_POST
GRAVITY_FAST = 8                     # GAME_HIP_GRAVITY_FAST
PROJECT = 16                         # GAME_HIP_PROJECT


def _flags(semantics, animate, lens_post, gravity_fast=False, project=False):
    return (SEMANTICS[semantics] | (ANIMATE_ARGS if animate else 0) | (LENS_POST if lens_post else 0) |
            (GRAVITY_FAST if gravity_fast else 0) | (PROJECT if project else 0))


def _check(rc):
    if rc != 0:
        raise GameError(rc, lib.game_last_error().decode("utf-8", "replace"),
                        lib.game_last_error_rendered().decode("utf-8", "replace"))


class Cinematic:
    """One cinematic's resolved IR (reference: codegen::ShaderOutput, game-compiler/src/codegen/mod.rs:35-48)."""

    def __init__(self, program, handle):
        self._program = program   # keeps the owning program alive
        self._h = handle

    @property
    def name(self):
        return lib.game_ir_name(self._h).decode("utf-8", "replace")

    @property
    def ir(self):
        return json.loads(lib.game_ir_json(self._h).decode("utf-8", "replace"))

    @property
    def blob(self):
        p = ctypes.c_void_p()
        n = lib.game_ir_blob(self._h, ctypes.byref(p))
        return ctypes.string_at(p, n)

    @property
    def uniforms(self):
        n = lib.game_ir_uniform_count(self._h)
        return [(lib.game_ir_uniform_name(self._h, i).decode("utf-8", "replace"), lib.game_ir_uniform_default(self._h, i)) for i in range(n)]

    @property
    def flags(self):
        return lib.game_ir_flags(self._h)

    @property
    def uses_memory(self):
        return bool(lib.game_ir_uses_memory(self._h))

    @property
    def has_gravity(self):
        return bool(lib.game_ir_has_gravity(self._h))

    @property
    def frame_floats(self):
        return lib.game_ir_frame_floats(self._h)

    def uniform_rows(self, times, w, h, audio=None, user=None, arcs=False):
        """Frame uniform table: one row per entry of `times` (seconds), reference defaults elsewhere. arcs=True
        applies the cinematic's arc timeline (runtime/arc.rs:110-122) to the user uniforms of every row, after `user`."""
        times = np.asarray(times, dtype=np.float32).reshape(-1)
        rows = np.zeros((len(times), self.frame_floats), dtype=np.float32)
        for k, t in enumerate(times):
            lib.game_ir_default_uniforms(self._h, float(t), w, h, rows[k].ctypes.data)
        if audio:
            for key, col in (("bass", BU_BASS), ("mid", BU_MID), ("treble", BU_TREBLE), ("energy", BU_ENERGY), ("beat", BU_BEAT)):
                if key in audio:
                    rows[:, col] = np.asarray(audio[key], dtype=np.float32)
        if user:
            names = [n for n, _ in self.uniforms]
            for key, val in user.items():
                rows[:, BU_COUNT + names.index(key)] = np.asarray(val, dtype=np.float32)
        if arcs:
            for k, t in enumerate(times):
                lib.game_ir_arc_update(self._h, float(t), rows[k].ctypes.data)
        return rows

    @property
    def arc_count(self):
        return lib.game_ir_arc_count(self._h)

    @property
    def derived_count(self):
        """Number of expression-valued stage arguments kept as per-frame programs."""
        return lib.game_ir_derived_count(self._h)

    @property
    def pixel_expr_count(self):
        """Number of expression-valued stage arguments that read the pixel (evaluated inside the kernel)."""
        return lib.game_ir_pixel_expr_count(self._h)

    def eval_derived(self, rows):
        """Per-frame values of those arguments for an (n, frame_floats) float32 array of uniform rows."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        out = np.zeros((rows.shape[0], self.derived_count), dtype=np.float32)
        for k in range(rows.shape[0]):
            lib.game_ir_eval_derived(self._h, rows[k].ctypes.data, out[k].ctypes.data)
        return out

    def animator(self, fps=60.0):
        """Stateful host-side animation of the user uniforms: arcs, `~` modulation, temporal operators, resonance."""
        return Animator(self, fps)

    @property
    def project_mode(self):
        """None, or the mode of the project block that names this cinematic: "flat", "dome", "cube", "led"."""
        m = lib.game_ir_project_mode(self._h)
        return None if m < 0 else ("flat", "dome", "cube", "led")[m]

    def compile_only(self, semantics="glsl", animate=False, lens_post=False, project=False):
        """hiprtc-compiles the generated kernels for gfx950 without a device; returns code bytes."""
        n = ctypes.c_size_t()
        _check(lib.game_hip_compile_only_ex(self._h, _flags(semantics, animate, lens_post, project=project), ctypes.byref(n)))
        return n.value

    def dump_code(self, path_prefix):
        """Writes the generated kernel source and gfx950 code object next to `path_prefix`."""
        _check(lib.game_hip_dump_code(self._h, path_prefix.encode()))

    def renderer(self, w, h, device=0, semantics="glsl", animate=False, lens_post=False, gravity_fast=False, project=False):
        """semantics: "glsl" = the reference's WebGL2 wording of the builtins (default), "wgsl" = its WebGPU wording.
        animate: expression-valued stage arguments take their per-frame value instead of the parameter default.
        lens_post: the `lens { post: ... }` chain runs on the composited colour (the reference ignores it)."""
        return Renderer(self, w, h, device, semantics, animate, lens_post, gravity_fast, project)


class Program:
    """Result of compiling .game source (reference: game_compiler::compile, game-compiler/src/lib.rs:117)."""

    def __init__(self, source, base_dir=None, lib_dirs=(), blob=None):
        if isinstance(source, str):
            source = source.encode("utf-8")
        h = ctypes.c_void_p()
        if blob is not None:    # a resolved IR handed over as data (game_ir_from_blob)
            blob = bytes(blob)
            _check(lib.game_ir_from_blob(blob, len(blob), ctypes.byref(h)))
        elif base_dir is None:
            _check(lib.game_front_compile(source, len(source), ctypes.byref(h)))
        else:       # file imports resolved against base_dir, then lib_dirs (reference: resolver::resolve_imports)
            dirs = (ctypes.c_char_p * max(len(lib_dirs), 1))(*[os.fsencode(d) for d in lib_dirs])
            _check(lib.game_front_compile_with_imports(source, len(source), os.fsencode(base_dir), dirs, len(lib_dirs), ctypes.byref(h)))
        self._h = h
        n = lib.game_program_cinematic_count(h)
        self.cinematics = [Cinematic(self, lib.game_program_ir(h, i)) for i in range(n)]
        self.warnings = [lib.game_program_warning(h, i).decode() for i in range(lib.game_program_warning_count(h))]

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.game_program_free(h)
            self._h = None


def compile(source, base_dir=None, lib_dirs=()):
    """base_dir=None: file imports are left alone, as the reference's compile() does; otherwise they are resolved."""
    return Program(source, base_dir, lib_dirs)


def load_blob(blob):
    """A program of one cinematic from a resolved-IR blob (layout in include/game_hip.h), validated; raises GameError
    for a blob that is damaged or names things that do not exist."""
    return Program(None, blob=blob)


def compile_file(path, lib_dirs=()):
    with open(path, "rb") as f:
        return Program(f.read(), os.path.dirname(os.path.abspath(path)), lib_dirs)


def lex(source):
    """Token spans (start, end) of `source` (reference: game_compiler::lex)."""
    if isinstance(source, str):
        source = source.encode("utf-8")
    n = ctypes.c_size_t()
    _check(lib.game_front_lex(source, len(source), ctypes.byref(n), None, 0))
    spans = (ctypes.c_size_t * (2 * max(n.value, 1)))()
    _check(lib.game_front_lex(source, len(source), ctypes.byref(n), spans, n.value))
    return [(spans[2 * i], spans[2 * i + 1]) for i in range(n.value)]


def tokens(source):
    """[(start, end, display text)] per token (reference: lexer::lex + Token's Display)."""
    if isinstance(source, str):
        source = source.encode("utf-8")
    n = ctypes.c_size_t()
    _check(lib.game_front_lex(source, len(source), ctypes.byref(n), None, 0))
    r = lib.game_front_tokens(source, len(source))
    out = []
    for rec in r.decode("utf-8").split("\x1e"):
        if rec:
            a, b, text = rec.split("\x1f", 2)
            out.append((int(a), int(b), text))
    return out


def parse(source, optimized=False):
    """AST as a dict (reference: game_compiler::compile_to_ast); optimized=True also runs
    expand_defines + optimize_cinematic."""
    if isinstance(source, str):
        source = source.encode("utf-8")
    r = lib.game_front_ast_json(source, len(source), 1 if optimized else 0)
    if r is None:
        raise GameError(3, lib.game_last_error().decode("utf-8", "replace"))
    return json.loads(r.decode("utf-8"))


def list_builtins():
    out = []
    for i in range(lib.game_builtin_count()):
        name, inp, outp, params = lib.game_builtin_describe(i).decode().split("|")
        out.append({"name": name, "input": inp, "output": outp, "params": [p for p in params.split(",") if p]})
    return out


def device_count():
    n = ctypes.c_int()
    _check(lib.game_hip_device_count(ctypes.byref(n)))
    return n.value


def shutdown():
    """Frees the per-device lattice tables (every Renderer must be closed first)."""
    _check(lib.game_hip_shutdown())


def pinned_array(shape):
    """uint8 array in page-locked host memory (game_hip_host_alloc), released when the array is collected."""
    import weakref
    n = int(np.prod(shape))
    ptr = lib.game_hip_host_alloc(n)
    if not ptr:
        raise GameError(81, lib.game_last_error().decode("utf-8", "replace"), "")
    buf = (ctypes.c_ubyte * n).from_address(ptr)
    weakref.finalize(buf, lib.game_hip_host_free, ptr)
    return np.frombuffer(buf, dtype=np.uint8).reshape(shape)


class Animator:
    """game_animator: call step() once per frame, in frame order; the row's user uniforms are updated in place."""
    ARCS, MODULATION, RESONANCE = 1, 2, 4

    def __init__(self, cinematic, fps=60.0):
        h_ = ctypes.c_void_p()
        _check(lib.game_animator_create(cinematic._h, float(fps), ctypes.byref(h_)))
        self._h = h_
        self.fps = float(fps) if fps else 60.0

    def step(self, time_s, dt, row, what=0, signals=None):
        assert row.dtype == np.float32 and row.flags["C_CONTIGUOUS"]
        signals = signals or {}
        n = len(signals)
        names = (ctypes.c_char_p * max(n, 1))(*[k.encode() for k in signals])
        values = (ctypes.c_double * max(n, 1))(*[float(v) for v in signals.values()])
        _check(lib.game_animator_step(self._h, float(time_s), float(dt), row.ctypes.data, what, names, values, n))
        return row

    def apply(self, rows, times, what=0, signals=None):
        """Steps through (n, frame_floats) rows in order; dt is the time since the previous frame (1/fps for the first).
        signals: None, or one dict of named signal values per frame."""
        prev = None
        for k, t in enumerate(times):
            dt = 1.0 / self.fps if prev is None else float(t) - prev
            self.step(t, dt, rows[k], what, None if signals is None else signals[k])
            prev = float(t)
        return rows

    def reset(self):
        lib.game_animator_reset(self._h)

    def close(self):
        if self._h:
            lib.game_animator_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Renderer:
    """GPU context for one cinematic at one resolution (reference: GameRendererGL, runtime/helpers.rs:131-241)."""

    def __init__(self, cinematic, w, h, device=0, semantics="glsl", animate=False, lens_post=False, gravity_fast=False, project=False):
        self.cinematic = cinematic
        self.w, self.h = w, h
        h_ = ctypes.c_void_p()
        _check(lib.game_hip_create_ex(cinematic._h, w, h, device, _flags(semantics, animate, lens_post, gravity_fast, project), ctypes.byref(h_)))
        self._h = h_

    def close(self):
        if getattr(self, "_h", None):
            lib.game_hip_destroy(self._h)
            self._h = None

    __del__ = close

    def set_ring(self, frames):
        _check(lib.game_hip_set_ring(self._h, frames))

    @property
    def ring_frames(self):
        return lib.game_hip_ring_frames(self._h)

    def _rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim == 1:
            rows = rows[None, :]
        if rows.shape[1] != self.cinematic.frame_floats:
            raise ValueError(f"uniform rows need {self.cinematic.frame_floats} floats, got {rows.shape[1]}")
        return rows

    def render(self, rows, readback=True, pinned=False, out=None):
        """Renders len(rows) frames; returns (nframes, h, w, 4) uint8 (top row first) or None. pinned=True
        reads back into page-locked host memory (freed with the returned array); out= a caller's array of that shape."""
        rows = self._rows(rows)
        shape = (rows.shape[0], self.h, self.w, 4)
        if out is not None:
            assert out.dtype == np.uint8 and out.shape == shape and out.flags["C_CONTIGUOUS"]
        elif readback:
            out = pinned_array(shape) if pinned else np.empty(shape, dtype=np.uint8)
        _check(lib.game_hip_render(self._h, rows.ctypes.data, rows.shape[0], None if out is None else out.ctypes.data))
        return out

    def render_into(self, rows, dev_ptr):
        """Allocation-free variant: frames go to the caller's device buffer (len(rows) * h * w * 4 bytes at dev_ptr)."""
        rows = self._rows(rows)
        _check(lib.game_hip_render_into(self._h, rows.ctypes.data, rows.shape[0], dev_ptr))

    def render_band(self, rows, row0, nrows, readback=True):
        rows = self._rows(rows)
        out = np.empty((rows.shape[0], nrows, self.w, 4), dtype=np.uint8) if readback else None
        _check(lib.game_hip_render_band(self._h, rows.ctypes.data, rows.shape[0], row0, nrows,
                                        out.ctypes.data if readback else None))
        return out

    def set_blur(self, radius):
        """Frame filter: separable binomial blur of radius 1..4 on every frame before it is stored (0 = off)."""
        _check(lib.game_hip_set_blur(self._h, radius))

    def render_raw_band(self, row, row0, nrows):
        """Phase one of a band with the frame filter on: one frame's band into the raw frame."""
        row = self._rows(row)
        assert row.shape[0] == 1
        _check(lib.game_hip_render_raw_band(self._h, row.ctypes.data, row0, nrows))

    def raw_rows_ptr(self, row):
        """Device pointer of a row of the raw frame (for halo exchange between bands)."""
        return lib.game_hip_raw_rows(self._h, row)

    def blur_band(self, row0, nrows, readback=True):
        """Phase two: blur the band (reading its halo rows in the raw frame) into the next ring slot."""
        out = np.empty((nrows, self.w, 4), dtype=np.uint8) if readback else None
        _check(lib.game_hip_blur_band(self._h, row0, nrows, out.ctypes.data if readback else None))
        return out

    def reset_memory(self):
        _check(lib.game_hip_reset_memory(self._h))

    def set_memory(self, frame, row0=0):
        """Restores the feedback state from an RGBA8 frame (or a band of it starting at row0): checkpoint / resume."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        assert frame.ndim == 3 and frame.shape[1] == self.w and frame.shape[2] == 4
        _check(lib.game_hip_set_memory(self._h, frame.ctypes.data, row0, frame.shape[0]))

    def sync(self):
        _check(lib.game_hip_sync(self._h))

    def timing_begin(self):
        """Calls until timing_end() each record their own event pair; nothing waits in between."""
        _check(lib.game_hip_timing_begin(self._h))

    def timing_end(self):
        """(summed device ms, launches) of the calls since timing_begin(); waits for them."""
        ms, n = ctypes.c_float(), ctypes.c_uint32()
        _check(lib.game_hip_timing_end(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def last_kernel_ms(self):
        ms, n = ctypes.c_float(), ctypes.c_uint32()
        _check(lib.game_hip_last_kernel_ms(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def frame_ptr(self, slot):
        return lib.game_hip_frame_ptr(self._h, slot)

    @property
    def kernel_source(self):
        return lib.game_hip_kernel_source(self._h).decode()

    @property
    def kernel_hash(self):
        """Identifies the kernel build: sha256 over the generated source and the device-function header it includes."""
        import hashlib
        return hashlib.sha256(lib.game_hip_kernel_source(self._h) + b"\0" + lib.game_hip_gravity_source(self._h) + b"\0" +
                              lib.game_hip_device_header_source()).hexdigest()[:16]

    def stream_ptr(self):
        """The context's hipStream_t as an integer (for torch.cuda.ExternalStream)."""
        return lib.game_hip_stream(self._h)

    def setup_ms(self):
        """(hiprtc compile ms, lattice-table build ms) spent creating this context."""
        a, b = ctypes.c_float(), ctypes.c_float()
        _check(lib.game_hip_setup_ms(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    # particles (reference: GameGravitySim, game-compiler/src/codegen/gravity.rs:123-191)
    def particles_init(self, pos_xy, vel_xy=None):
        pos = np.ascontiguousarray(pos_xy, dtype=np.float32).reshape(-1, 2)
        vel = None if vel_xy is None else np.ascontiguousarray(vel_xy, dtype=np.float32).reshape(-1, 2)
        _check(lib.game_hip_particles_init(self._h, pos.shape[0], pos.ctypes.data, None if vel is None else vel.ctypes.data))
        self._n = pos.shape[0]

    def particles_seed(self, n, seed=0x9E3779B97F4A7C15):
        _check(lib.game_hip_particles_seed(self._h, n, seed))
        self._n = n

    def particles_step(self, dt, steps=1, first=0, count=None):
        _check(lib.game_hip_particles_step(self._h, dt, steps, first, self._n if count is None else count))

    def particles_read(self):
        pos = np.empty((self._n, 2), dtype=np.float32)
        vel = np.empty((self._n, 2), dtype=np.float32)
        _check(lib.game_hip_particles_read(self._h, pos.ctypes.data, vel.ctypes.data))
        return pos, vel

    def particles_positions(self):
        x, y = ctypes.c_void_p(), ctypes.c_void_p()
        _check(lib.game_hip_particles_positions(self._h, ctypes.byref(x), ctypes.byref(y)))
        return x.value, y.value

    def particles_set_time(self, t):
        _check(lib.game_hip_particles_set_time(self._h, float(t)))

    def particles_time(self):
        t = ctypes.c_float()
        _check(lib.game_hip_particles_time(self._h, ctypes.byref(t)))
        return t.value


EVAL_OPS = {"fract": 0, "sin": 1, "cos": 2, "atan2": 3, "hash2": 4, "noise2": 5, "simplex2d": 6, "voronoi2d": 7,
            "exp2": 8, "log2": 9, "pow": 10, "max": 11, "min": 12, "fbm6": 13, "sqrt": 14, "div": 15,
            "smoothstep_glow": 16, "mod": 17, "unorm8": 18, "noise2_lut": 19, "fbm6_lut": 20, "voronoi2d_lut": 21, "simplex2d_lut": 22, "pack_unorm8": 23, "rsqrt": 24}


def device_eval(op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.float32).reshape(-1)
    out = np.empty_like(a)
    _check(lib.game_hip_eval(EVAL_OPS[op], a.size, a.ctypes.data, None if bb is None else bb.ctypes.data, out.ctypes.data))
    return out


def check_exhaustive(which):
    m, f = ctypes.c_uint64(), ctypes.c_uint32()
    _check(lib.game_hip_check_exhaustive(which, ctypes.byref(m), ctypes.byref(f)))
    return m.value, f.value


def parse_blob(blob):
    """Decodes the flat IR blob (layout in include/game_hip.h) into a dict; used by tests."""
    hdr = struct.unpack_from("<10IfI64s", blob, 0)
    keys = ("magic", "total", "n_layers", "n_stages", "n_uniforms", "n_law", "flags", "multi_layer", "uses_memory",
            "gravity_present", "damping", "bounds", "name")
    out = dict(zip(keys, hdr))
    out["name"] = out["name"].split(b"\0")[0].decode()
    return out


from . import distributed  # noqa: E402,F401  (partition rules and exchange steps for N > 1 GPUs)
from . import workloads  # noqa: E402,F401  (benchmark cinematics and their synthetic inputs)
