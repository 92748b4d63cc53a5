/* game_hip.h — C ABI of libgame_hip.so, the MI355X (gfx950) render path for GAME cinematics.
 *
 * The reference compiler (runyourempire/game-engine, Rust) has no FFI of its own: its seam is the
 * Rust function boundary `game_compiler::compile` -> `codegen::generate` -> `ShaderOutput`, whose
 * GLSL text a browser GPU then executes once per pixel. This library replaces everything below
 * that seam: it takes .game source (or the resolved IR), and renders RGBA8 frames, memory-feedback
 * frames and gravity particle steps on the GPU. Each entry point names the reference interface it
 * stands in for. All functions return 0 on success; see GAME_E_* for failures. After a non-zero
 * return `game_last_error()` holds the message (thread-local, valid until the next call on the
 * same thread).
 *
 * Threading: a game_program / game_ir is immutable after creation and may be shared. A
 * game_hip_ctx belongs to one GPU and must be driven by one host thread at a time; distinct
 * contexts may be driven concurrently (one per GPU), which mirrors the reference's single-threaded
 * render() per renderer instance (game-compiler/src/runtime/helpers.rs:214).
 */
#ifndef GAME_HIP_H
#define GAME_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct game_program game_program; /* parsed + optimised + validated source: N cinematics */
typedef struct game_ir game_ir;           /* one cinematic's resolved IR (owned by its program)  */
typedef struct game_hip_ctx game_hip_ctx; /* device state for one cinematic at one resolution    */

/* Return codes. 1..10 are the reference's ErrorCode E001..E010 (game-compiler/src/error.rs:90-111). */
enum {
    GAME_OK = 0,
    GAME_E001_UNKNOWN_STAGE = 1,
    GAME_E002_TYPE_MISMATCH = 2,
    GAME_E003_PARSE = 3,
    GAME_E_LEX = 64,        /* CompileError::LexError (carries no code in the reference)          */
    GAME_E_VALIDATION = 65, /* CompileError::ValidationError without a code (cast errors)          */
    GAME_E_CODEGEN = 66,    /* CompileError::CodegenError                                          */
    GAME_E_ARGUMENT = 80,   /* bad argument to this library                                        */
    GAME_E_HIP = 81,        /* HIP runtime or hiprtc failure                                       */
    GAME_E_NO_DEVICE = 82,  /* no usable GPU / HIP runtime not initialisable                       */
    GAME_E_UNSUPPORTED = 83 /* cinematic has no such part (e.g. particles without a gravity block) */
};

const char* game_last_error(void);

/* ── front end ─────────────────────────────────────────────────────────────────────────────
 * game_front_compile replaces game_compiler::compile (game-compiler/src/lib.rs:117) up to and
 * including codegen::generate's validation (game-compiler/src/codegen/mod.rs:138-141): lex, parse,
 * expand_defines, optimize_cinematic, validate, extract_uniforms. The error text in
 * game_last_error() is the reference's Display text for the same failure. */
int game_front_compile(const char* src, size_t len, game_program** out);
/* The same with file imports resolved first (game-compiler/src/resolver.rs:18-147, a library function the
 * reference's own compile() does not call yet: lib.rs:150 "standard file import - future work"): each
 * `import "path" ...` is looked up relative to base_dir, then in each library directory, as written and with
 * ".game" appended; the defines of the imported file (and, transitively, of its imports) are merged into the
 * first cinematic - all of them for `expose ALL`, the named ones for `expose a, b`, and under the name
 * "alias.name" for `as alias`. Errors keep the reference's messages ("import not found: 'x'", "circular import
 * detected: 'x'", ...). URI imports (shadertoy://, midi://, osc://, camera://) are not files and stay inert. */
int game_front_compile_with_imports(const char* src, size_t len, const char* base_dir, const char* const* lib_dirs,
                                    size_t n_lib_dirs, game_program** out);
void game_program_free(game_program* p);
size_t game_program_cinematic_count(const game_program* p);
const game_ir* game_program_ir(const game_program* p, size_t index);
/* game_compiler::check (lib.rs:83): semantic warnings, run on the parsed (unexpanded) program. */
size_t game_program_warning_count(const game_program* p);
const char* game_program_warning(const game_program* p, size_t index);
/* error.rs:296 render_with_source for the last front-end error of this thread ("" if none). */
const char* game_last_error_rendered(void);

/* game_compiler::lex (lib.rs:76): number of tokens, and per token its byte span and display text. */
int game_front_lex(const char* src, size_t len, size_t* ntokens, size_t* spans /* 2*cap */, size_t cap);
/* Token records "<start>\x1f<end>\x1f<display>" separated by \x1e, with the display text of
 * token.rs:82-149 (keywords and punctuation as written, numbers in shortest form, units
 * re-attached). NULL on a lex error. Thread-local storage, valid until the next call. */
const char* game_front_tokens(const char* src, size_t len);
/* game_compiler::compile_to_ast (lib.rs:66) as JSON: stage 0 = as parsed; stage 1 = after
 * expand_defines + optimize_cinematic (lib.rs:121-129). NULL on a lex/parse error. */
const char* game_front_ast_json(const char* src, size_t len, int stage);
/* game_compiler::list_builtins (lib.rs:102): "name|Input|Output|param,param,..." per builtin. */
size_t game_builtin_count(void);
const char* game_builtin_describe(size_t index);

/* ── IR accessors ──────────────────────────────────────────────────────────────────────────
 * The IR is what codegen::generate would have frozen into shader text: ShaderOutput.name,
 * .uniforms (codegen/mod.rs:29-48), the per-stage argument text of stages.rs:25 get_arg, the
 * memory decay literal (memory.rs:38-45) and the gravity force law (gravity.rs:10-41). */
const char* game_ir_name(const game_ir* ir);
const char* game_ir_json(const game_ir* ir);                /* full dump, UTF-8 JSON */
size_t game_ir_blob(const game_ir* ir, const void** data);  /* flat blob, layout below */
uint32_t game_ir_uniform_count(const game_ir* ir);
const char* game_ir_uniform_name(const game_ir* ir, uint32_t i);
double game_ir_uniform_default(const game_ir* ir, uint32_t i);
uint32_t game_ir_flags(const game_ir* ir);                  /* GAME_IRF_* */
int game_ir_uses_memory(const game_ir* ir);
int game_ir_has_gravity(const game_ir* ir);
/* Floats per frame in the uniform table: 10 built-ins + user uniforms
 * [time, audio_bass, audio_mid, audio_treble, audio_energy, audio_beat, res.x, res.y, mouse.x,
 *  mouse.y, user...] — the order GameRendererGL.render() uploads them (runtime/helpers.rs:222-232). */
uint32_t game_ir_frame_floats(const game_ir* ir);
/* Fills one frame's uniform row with the reference's defaults: time = t, audio = 0, resolution =
 * (w, h), mouse = 0, user uniforms = their declared defaults. */
void game_ir_default_uniforms(const game_ir* ir, float time_s, uint32_t w, uint32_t h, float* row);

/* Expression-valued stage arguments (`rotate(time * 1.5)`, `circle(0.15 + sin(time * 2.0) * 0.03)`): the
 * reference's emitters replace them by the parameter's default (codegen/stages.rs:8-22) although its
 * expression compiler (codegen/expr.rs:50-144) exists for them. The IR keeps every such expression that reads
 * only frame constants (numbers, pi/tau/e/phi, `time`, audio.*, mouse.*, user uniforms; + - * / ^ > <, the
 * scalar math functions) as a "derived value". game_ir_eval_derived computes them for one frame row in f32;
 * a context created with GAME_HIP_ANIMATE_ARGS does so itself for every frame and feeds them to the kernel
 * as extra uniforms. Without that flag the defaults are used, as the reference does.
 * An expression that also reads the pixel (`p.x`, `p.y`, `uv.x`, `uv.y`: `circle(0.2 + 0.05 * sin(p.x * 20.0))`)
 * cannot be a frame constant; the IR keeps it as a "pixel expression" (game_ir_pixel_expr_count) and a context
 * with GAME_HIP_ANIMATE_ARGS evaluates it inside the kernel, in f32 with the device functions, at the point
 * where its stage runs (`p` is the position the earlier stages of the layer have produced). */
uint32_t game_ir_derived_count(const game_ir* ir);
uint32_t game_ir_pixel_expr_count(const game_ir* ir);
void game_ir_eval_derived(const game_ir* ir, const float* row, float* out /* game_ir_derived_count floats */);

/* Host-side animation of the user uniforms, one call per frame, in frame order. The reference emits this as
 * JavaScript that nothing calls: the modulation `base ~ expr` (LANGUAGE.md section 8: the value of `expr` is
 * added to the base every frame), the temporal operators `>>` `<>` `!!` `..` as the classes GameRingBuffer,
 * GameEMAFilter and GameEdgeDetector (game-compiler/src/codegen/temporal.rs:31-82, applied left to right as
 * :130-172), `resonanceUpdate(params, signals, dt)` (codegen/resonance.rs:20-65) and `arcUpdate(time, params)`
 * (runtime/arc.rs:109-122). game_animator_step applies, in this order and to the user uniforms of `row`
 * (game_ir_frame_floats floats, in/out): bit 0 of `what` the arc timeline at time_s; bit 1 modulation (over
 * the row's own time/audio/mouse/user values; `time` here is the raw u_time, as the JS sees it) followed by
 * the parameter's temporal operators; bit 2 resonance with the caller's named signals. what = 0 means all
 * three. Arithmetic is binary64 (JS numbers), ring buffers hold f32 (Float32Array), results are stored as
 * f32. fps sizes the temporal operators (0 = 60, the classes' default). The animator keeps the operators'
 * state between calls; game_animator_reset returns it to the first frame. */
typedef struct game_animator game_animator;
int game_animator_create(const game_ir* ir, double fps, game_animator** out);
void game_animator_destroy(game_animator* an);
void game_animator_reset(game_animator* an);
int game_animator_step(game_animator* an, double time_s, double dt, float* row, uint32_t what,
                       const char* const* signal_names, const double* signal_values, uint32_t nsignals);

/* Arc timeline (game-compiler/src/runtime/arc.rs:52-125, emitted by the reference as the JS function
 * arcUpdate(time, params) that its own renderer never calls): entries `target: from -> to over duration
 * [easing]` run back to back; game_ir_arc_update applies the timeline at time_s, in JS-number (double)
 * arithmetic, to the user-uniform part of one frame row (values are then Float32, as uploaded). Seven
 * easings: linear, expo_in, expo_out, cubic_in_out, smooth, elastic, bounce. Entries whose target names no
 * uniform or whose endpoints are not constant numbers are skipped, as the emitted JS skips idx < 0. */
uint32_t game_ir_arc_count(const game_ir* ir);
void game_ir_arc_update(const game_ir* ir, double time_s, float* row);

/* `project flat|dome|cube|led (params) { source: name }` (game-compiler/src/codegen/project.rs, applied at lib.rs:187-192):
 * the reference swaps the WebGPU vertex stage of the cinematic named by `source`, which changes the v_uv its fragments
 * start from (and leaves its WebGL2 output alone). game_ir_project_mode: -1 = no project block names this cinematic,
 * 0 flat, 1 dome, 2 cube, 3 led. A context created with GAME_HIP_PROJECT applies the block as a per-pixel warp of
 * v_uv before the layers run: dome = the fisheye mapping of project.rs:62-69 (evaluated per pixel; the reference
 * evaluates it at the three vertices of its full-screen triangle, where it degenerates to an affine map), led = every
 * pixel of LED k = floor(v_uv.x * count) samples (k / count, 0.5) (project.rs:128-137), flat and cube = unchanged. The
 * memory mix still reads the previous frame at the pixel itself. */
int game_ir_project_mode(const game_ir* ir);

/* Places where the reference's emitted GLSL does not compile or leaves a value undefined; the IR
 * then carries the evident intent (see front.hpp IrFlag for the per-bit meaning). */
enum {
    GAME_IRF_DOTTED_ARG = 1u << 0, GAME_IRF_TIME_IDENT = 1u << 1, GAME_IRF_BUILTIN_IDENT = 1u << 2,
    GAME_IRF_BLOCK_SCOPED_SDF = 1u << 3, GAME_IRF_REDECLARED = 1u << 4, GAME_IRF_NO_COLOR = 1u << 5,
    GAME_IRF_UNKNOWN_DOTTED = 1u << 6, GAME_IRF_NO_OUTPUT = 1u << 7, GAME_IRF_GRAVITY_IDENT = 1u << 8,
    GAME_IRF_USE_BEFORE_DECL = 1u << 9
};

/* Blob layout (little-endian, every field 4 bytes unless noted):
 *   header : magic 'GIR1', total_bytes, n_layers, n_stages, n_uniforms, n_law_ops, flags,
 *            multi_layer, uses_memory, gravity_present, gravity_damping(f32), gravity_bounds
 *            (0 reflect, 1 wrap, 2 none), name[64]
 *   layer  : is_pipeline, stage_begin, stage_count, has_memory, decay(f32), end_state
 *            (0 Position, 1 Sdf, 2 Color), post (1 = the lens post chain), 0      (32 bytes)
 *   stage  : builtin_id, nargs, aux(palette index), 0, then 4 x {kind, slot, lit(f32), expr}
 *            kind 0 = literal, 1 = user uniform slot, 2 = built-in uniform slot;
 *            expr = (derived index + 1) | (pixel expression index + 1) << 16, 0 = none      (80 bytes)
 *   uniform: default(f32), name[60]                                               (64 bytes)
 *   law op : op, value(f32), fn, 0                                                (16 bytes)
 *   then   : n_pixel_expressions, and per expression its length and that many law ops (ops 14..18 push a frame-row
 *            slot (fn), p.x, p.y, uv.x, uv.y)
 *   then   : project: mode + 1 (0 = none; 1 flat, 2 dome, 3 cube, 4 led), fov_deg(f32), led count, 0     (16 bytes)
 *   then   : n_derived_values, and per value its length and that many law ops (the frame-level programs of
 *            expression-valued stage arguments: numbers, frame-row slots, operators and calls only)
 * Law ops: 0 number(value), 1 dist, 2 simulation time (force laws only), 3 + 4 - 5 * 6 / 7 pow 8 > 9 <, 10 negate,
 * 11 / 12 / 13 call of a 1 / 2 / 3 argument function fn (0 sin cos tan abs sqrt exp log floor ceil fract 10 sign |
 * 11 min max pow 14 step | 15 clamp mix 17 smoothstep | 18 exp2 log2 20 round | 21 mod atan2 23 length(x, y) |
 * 24 select(c, t, f) = c is non-zero and not NaN ? t : f, three arguments), a program leaves exactly one value;
 * 14 frame-row slot fn, 15..18 p.x p.y uv.x uv.y (pixel expressions only).
 * Built-in uniform slot 0 used as a stage argument means the shader's wrapped time
 * `fract(u_time / 120) * 120` (glsl.rs:56), not the raw u_time. */

/* The resolved IR as data. A host that runs the reference's own lexer, parser and optimizer
 * (game_compiler::compile up to codegen::generate's validation, game-compiler/src/lib.rs:117-141, codegen/mod.rs:110-141)
 * writes what `generate` would have frozen into shader text - the pipe-chain stage sequence with its arguments, the
 * uniform table (codegen/mod.rs:54-107), memory decay, force law - into the blob layout above and hands it over here
 * instead of `.game` text. Every count, index, string and program in the blob is validated; a blob that fails is refused
 * with GAME_E_ARGUMENT and a reason in game_last_error(). The program holds one cinematic (game_program_ir(p, 0));
 * layer names, source texts, arcs and modulation are not part of the blob and are empty. game_ir_blob of the result
 * returns the same bytes. */
int game_ir_from_blob(const void* data, size_t len, game_program** out);

/* ── GPU render path ───────────────────────────────────────────────────────────────────────
 * game_hip_create replaces GameRendererGL.init (runtime/helpers.rs:148-183): where the browser
 * compiles the emitted GLSL, this assembles one fused per-pixel kernel for the cinematic from
 * hand-written gfx950 device functions and compiles it with hiprtc. `device` is a HIP ordinal. */
int game_hip_device_count(int* count);
int game_hip_create(const game_ir* ir, uint32_t width, uint32_t height, int device, game_hip_ctx** out);
/* Same with flags. GAME_HIP_SEMANTICS_WGSL: the sixteen builtins that the reference's WebGPU emitter
 * (game-compiler/src/codegen/wgsl.rs:268-540) words differently from its WebGL2 emitter (star, polygon,
 * simplex, voronoi, concentric_waves, threshold, vignette, chromatic, tonemap, scanlines, glitch, mirror,
 * repeat, curl_noise, displace, spectrum) follow the WebGPU text; flags = 0 is game_hip_create. */
enum {
    GAME_HIP_SEMANTICS_WGSL = 1u,
    GAME_HIP_ANIMATE_ARGS = 2u,   /* see game_ir_eval_derived */
    /* `lens { post: bloom(0.4, 1.5) | grain(0.015) }` is parsed by the reference (parser.rs:1197-1234) and then ignored
     * by both of its emitters. With this flag the Color -> Color builtins of the chain run, in order, on the colour
     * composited from the layers, before the frame is stored (and so before it becomes the next frame's memory). */
    GAME_HIP_LENS_POST = 4u,
    /* Gravity step with the pair distance and its reciprocal taken from one hardware reciprocal-square-root
     * (about 1 ulp each) instead of the correctly rounded root and reciprocal of the default build. Not bit-reproducible
     * on a CPU; positions stay within the stated tolerance of the default build (1e-4 relative after 100 steps). */
    GAME_HIP_GRAVITY_FAST = 8u,
    GAME_HIP_PROJECT = 16u        /* see game_ir_project_mode */
};
int game_hip_create_ex(const game_ir* ir, uint32_t width, uint32_t height, int device, uint32_t flags,
                       game_hip_ctx** out);
void game_hip_destroy(game_hip_ctx* ctx);

/* game_hip_render replaces GameRendererGL.render (runtime/helpers.rs:214-235) for `nframes`
 * consecutive frames. `uniforms` holds nframes rows of game_ir_frame_floats() floats (host memory).
 * Frames are RGBA8, row 0 = TOP row of the image (the GL framebuffer's row 0 is the bottom row;
 * the flip happens when the kernel stores a pixel). Frames land in a device ring of `ring_frames`
 * slots (frame k of this call -> slot (first_slot + k) % ring_frames; a call may render more frames
 * than the ring holds, in ring-sized launches); when rgba8_host is non-NULL every frame is also copied
 * to host memory in stream order, before its slot is reused (nframes * w * h * 4 bytes). Cinematics with `memory:` layers are frame-serial: each frame reads the previous
 * frame's RGBA8 output (zero before the first frame; see game_hip_reset_memory). */
int game_hip_set_ring(game_hip_ctx* ctx, uint32_t ring_frames);
int game_hip_render(game_hip_ctx* ctx, const float* uniforms, uint32_t nframes, uint8_t* rgba8_host);
/* Same, restricted to image rows [row0, row0 + rows) (top-first): the unit of work of one rank
 * when a memory/feedback cinematic is sharded into horizontal bands. Rows outside the band are
 * not touched. rgba8_host, when non-NULL, receives only the band (nframes * rows * w * 4). */
int game_hip_render_band(game_hip_ctx* ctx, const float* uniforms, uint32_t nframes, uint32_t row0,
                         uint32_t rows, uint8_t* rgba8_host);
/* The same frames as game_hip_render, without any allocation. This entry point has NO counterpart in the reference: it
 * is for callers that cannot allocate device memory while they render - code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc. The caller passes the input rows (host) and the output (rgba8_dev: DEVICE
 * memory, 16-byte aligned, nframes * w * h * 4 bytes, frame k at offset k * w * h * 4, top row first); the staging copy
 * of the uniform rows, which the other render calls keep in a buffer of their own, lives in a file-scope device array
 * of GAME_HIP_SCRATCH_BYTES = 262144 bytes that every GPU the library runs on holds until the process ends. The call
 * makes no hipMalloc / hipFree / hipHostMalloc / event creation; everything it needs was made by game_hip_create. It
 * returns when the work is queued on the context's stream (game_hip_sync waits). Limits: nframes * (frame floats + 2
 * + derived values) * 4 <= 262144 and nframes <= 65535, GAME_E_ARGUMENT beyond. Memory cinematics continue from, and
 * update, the context's feedback state exactly as game_hip_render does. A context with a frame filter set
 * (game_hip_set_blur) is refused with GAME_E_UNSUPPORTED: filtered frames need the raw frame and go through
 * game_hip_render. All contexts of one GPU share that GPU's array, which keeps the last call's contents. Calls made one
 * after the other - from one thread, or from several with the caller's own lock - are ordered on the device without a
 * host wait: each call makes its stream wait for a per-GPU event that the previous call recorded behind its kernels, so
 * two contexts may alternate without game_hip_sync in between. Calls that overlap in time on one GPU (distinct contexts
 * driven concurrently from distinct threads) race on the array and on that event - serialise them, or use
 * game_hip_render. Inside a timing section the call is not timed. */
int game_hip_render_into(game_hip_ctx* ctx, const float* uniforms, uint32_t nframes, void* rgba8_dev);
/* Page-locked host memory for the rgba8_host argument: frames then leave the GPU by DMA at PCIe rate
 * instead of through the staged copy a pageable buffer gets (about 4x faster on the tested host). Any host
 * pointer is accepted by the render calls; this pair is a convenience around hipHostMalloc / hipHostFree. */
void* game_hip_host_alloc(size_t bytes);
void game_hip_host_free(void* p);
/* Frame filter (an extension: the reference's post passes are all pointwise, game-compiler/src/codegen/glsl.rs:440-505,
 * and it has no blur). radius 1..4 turns on a separable binomial blur of every rendered frame before it is stored
 * in the ring - and so before it becomes the next frame's memory: weights C(2r, k), each pass
 * (sum of weight * byte + 4^r / 2) >> 2r per channel, coordinates clamped to the frame (the memory texture of
 * memory.rs:114-115 is CLAMP_TO_EDGE); radius 0 turns it off. With the filter on, frames are rendered one launch pair at
 * a time (fused kernel into a raw frame, blur into the ring slot).
 * Band sharding with the filter needs `radius` halo rows from the neighbouring bands, and they must be the neighbours'
 * RAW rows of the same frame. The two-phase calls leave room for that exchange, device to device:
 *   game_hip_render_raw_band  renders one frame's band into the raw frame (does not advance the ring);
 *   game_hip_raw_rows         device pointer of row `row` of the raw frame (w RGBA8 words per row): send the band's
 *                             first / last `radius` rows to the ranks above / below and receive theirs into the rows
 *                             just outside the band (RCCL send/recv, or a peer copy), queued on game_hip_stream;
 *   game_hip_blur_band        blurs the band into the next ring slot (reading the halo rows) and advances the ring.
 * game_hip_render / _render_band with the filter on do both phases back to back and read whatever the raw frame
 * holds outside the band. */
int game_hip_set_blur(game_hip_ctx* ctx, uint32_t radius);
int game_hip_render_raw_band(game_hip_ctx* ctx, const float* uniforms /* one row */, uint32_t row0, uint32_t rows);
void* game_hip_raw_rows(game_hip_ctx* ctx, uint32_t row);
int game_hip_blur_band(game_hip_ctx* ctx, uint32_t row0, uint32_t rows, uint8_t* rgba8_host /* band or NULL */);
int game_hip_reset_memory(game_hip_ctx* ctx);          /* previous-frame store := 0 */
/* Checkpoint / resume of the feedback state (the reference keeps it in a browser texture and has no such call):
 * loads rows [row0, row0 + rows) of an RGBA8 frame, top row first - typically the last frame (or band) a previous
 * run returned - as the frame the next rendered frame reads as its previous one. Synchronous. */
int game_hip_set_memory(game_hip_ctx* ctx, const uint8_t* rgba8_frame, uint32_t row0, uint32_t rows);
int game_hip_sync(game_hip_ctx* ctx);                  /* wait for the context's stream */
/* The context's HIP stream (a hipStream_t): every launch and copy of the context is queued on it, so work a caller
 * queues on the same stream - a collective over the particle arrays, a consumer of a ring slot - is ordered against
 * the context's kernels without a host synchronisation. */
void* game_hip_stream(game_hip_ctx* ctx);
void* game_hip_frame_ptr(game_hip_ctx* ctx, uint32_t slot);   /* device pointer of a ring slot */
uint32_t game_hip_ring_frames(const game_hip_ctx* ctx);
/* Device time of the kernels of the last render call, from HIP events recorded on the context's
 * stream around the launches (milliseconds), and the number of launches it covers. */
int game_hip_last_kernel_ms(game_hip_ctx* ctx, float* ms, uint32_t* launches);
/* The same over many calls without waiting in between: every render / particle-step call between _begin and
 * _end records an event pair of its own on the render stream; _end waits for the last of them and returns the
 * summed device time and the number of kernel launches. (game_hip_last_kernel_ms after a call inside the section
 * still works and waits for that call.) */
int game_hip_timing_begin(game_hip_ctx* ctx);
int game_hip_timing_end(game_hip_ctx* ctx, float* total_ms, uint32_t* launches);
/* Source text of the generated kernel (for inspection and tests), and of the device-function header it includes
 * (the two together are everything hiprtc compiles; a hash of both identifies a kernel build). */
const char* game_hip_kernel_source(const game_hip_ctx* ctx);
const char* game_hip_gravity_source(const game_hip_ctx* ctx);   /* "" without a gravity block */
const char* game_hip_device_header_source(void);
/* Host milliseconds game_hip_create spent (a) in hiprtc for this context's kernels - 0 when the process had already
 * compiled the same source - and (b) building the per-device lattice tables - 0 when the device already had them.
 * The browser's counterpart is shader compilation inside GameRendererGL.init (runtime/helpers.rs:148-183). */
int game_hip_setup_ms(const game_hip_ctx* ctx, float* compile_ms, float* tables_ms);
/* Compile the generated kernel for gfx950 without touching a device (build-time check). */
int game_hip_compile_only(const game_ir* ir, size_t* code_bytes);
int game_hip_compile_only_ex(const game_ir* ir, uint32_t flags, size_t* code_bytes);
/* Same, and writes <prefix>_frame.hip / <prefix>_frame.co (and _gravity.*): the generated source
 * and the gfx950 code object, for disassembly and instruction-mix studies. */
int game_hip_dump_code(const game_ir* ir, const char* path_prefix);

/* ── gravity particles ─────────────────────────────────────────────────────────────────────
 * Replaces GameGravitySim.init/dispatch (game-compiler/src/codegen/gravity.rs:123-191) and the
 * compute shader of gravity.rs:43-120. Particles are stored SoA on the device (x, y, vx, vy).
 * `first`/`count` select the index range this context integrates (all-pairs against all n
 * positions); with count < n the caller exchanges positions between ranks after each step
 * (game_hip_particles_positions gives the device arrays to all-gather). */
int game_hip_particles_init(game_hip_ctx* ctx, uint32_t n, const float* pos_xy_host /* 2n */,
                            const float* vel_xy_host /* 2n or NULL = 0 */);
int game_hip_particles_seed(game_hip_ctx* ctx, uint32_t n, uint64_t seed); /* SplitMix64 init */
int game_hip_particles_step(game_hip_ctx* ctx, float dt, uint32_t steps, uint32_t first, uint32_t count);
int game_hip_particles_read(game_hip_ctx* ctx, float* pos_xy_host, float* vel_xy_host);
int game_hip_particles_positions(game_hip_ctx* ctx, float** x_dev, float** y_dev);
int game_hip_particles_time(game_hip_ctx* ctx, float* sim_time);
/* Resume: the simulation time a restored particle state continues from (particles_init resets it to 0). */
int game_hip_particles_set_time(game_hip_ctx* ctx, float sim_time);

/* Releases what the library keeps per device for the life of the process (the lattice tables of game_device.h). Every
 * context must have been destroyed first - while one exists the call fails with GAME_E_ARGUMENT and frees nothing (a
 * context holds pointers into those tables); later contexts rebuild what they need. The browser counterpart is the
 * page going away. */
int game_hip_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* GAME_HIP_H */
