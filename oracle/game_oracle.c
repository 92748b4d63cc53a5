/* game_oracle.c — TEST INFRASTRUCTURE. CPU restatement of the per-pixel fragment pipeline,
 * memory feedback and gravity step that the reference compiler emits as shader text and a browser
 * GPU executes. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this;
 * the shipped render path never does.
 *
 * PARITY UNPINNED: the reference holds no pixel or particle golden vectors (its tests assert on
 * emitted text only) and its executor (a browser GLSL/WGSL implementation) cannot run here. The only
 * pixels of the reference's own in its tree, two dev-server screenshots rendered from an older
 * generation's shader text, pin the prologue (pixel centres, uv, aspect, p), the expression
 * functions and the RGBA8 store - not the builtins of today's emitter
 * (tests/test_reference_screenshots.py: within one 8-bit level everywhere). This
 * file follows the emitted GLSL ES 3.00 text operation by operation, in scalar FP32, one rounding
 * per GLSL operator (compile with -ffp-contract=off), with the function definitions of gm_math.h.
 *
 *   fragment prologue ............ game-compiler/src/codegen/glsl.rs:53-61, vertex :541-555
 *   hash2 / noise2 / fbm2 ........ glsl.rs:134-150, :155-169
 *   simplex2d / voronoi2d ........ glsl.rs:172-199, :201-220
 *   stages ....................... glsl.rs:261-531 (cited per case below)
 *   layer compositing ............ glsl.rs:66-77, :228-259
 *   memory mix ................... game-compiler/src/codegen/memory.rs:38-45, runtime :102-147
 *   gravity step ................. game-compiler/src/codegen/gravity.rs:43-120
 *
 * Input is the flat IR blob produced by the front end (layout in include/game_hip.h).
 */
#include "gm_math.h"
#include <immintrin.h>
#include <pthread.h>
#include <stdlib.h>
#include <stdio.h>

/* builtin ids: order of the reference's BUILTINS registry (builtins.rs:221-293) */
enum {
    B_CIRCLE, B_RING, B_STAR, B_FBM, B_SIMPLEX, B_GLOW, B_SHADE, B_EMISSIVE, B_TINT, B_BLOOM, B_GRAIN,
    B_TRANSLATE, B_ROTATE, B_SCALE, B_MASK_ARC, B_THRESHOLD, B_TWIST, B_MIRROR, B_REPEAT, B_GRADIENT,
    B_BLEND, B_VIGNETTE, B_VORONOI, B_ONION, B_DOMAIN_WARP, B_CURL_NOISE, B_BOX, B_POLYGON,
    B_CONCENTRIC_WAVES, B_ROUND, B_TONEMAP, B_SCANLINES, B_CHROMATIC, B_SATURATE_COLOR, B_GLITCH,
    B_DISPLACE, B_SPECTRUM, B_PALETTE
};
enum { BU_TIME = 0, BU_BASS, BU_MID, BU_TREBLE, BU_ENERGY, BU_BEAT, BU_RES_X, BU_RES_Y, BU_MOUSE_X, BU_MOUSE_Y, BU_COUNT };

typedef struct { uint32_t kind, slot; float lit; uint32_t pad; } ir_arg;
typedef struct { uint32_t builtin, nargs, aux, pad; ir_arg args[4]; } ir_stage;
typedef struct { uint32_t is_pipeline, stage_begin, stage_count, has_memory; float decay; uint32_t end_state, pad0, pad1; } ir_layer;
typedef struct { uint32_t op; float val; uint32_t fn, pad; } ir_law;
typedef struct {
    uint32_t magic, total, n_layers, n_stages, n_uniforms, n_law, flags, multi_layer, uses_memory, gravity_present;
    float damping; uint32_t bounds; char name[64];
} ir_header;

enum { LAW_NUM = 0, LAW_DIST, LAW_TIME, LAW_ADD, LAW_SUB, LAW_MUL, LAW_DIV, LAW_POW, LAW_GT, LAW_LT, LAW_NEG,
       LAW_CALL1, LAW_CALL2, LAW_CALL3, LAW_UNIFORM, LAW_PX, LAW_PY, LAW_UVX, LAW_UVY };
enum { LF_SIN = 0, LF_COS, LF_TAN, LF_ABS, LF_SQRT, LF_EXP, LF_LOG, LF_FLOOR, LF_CEIL, LF_FRACT, LF_SIGN,
       LF_MIN, LF_MAX, LF_POW, LF_STEP, LF_CLAMP, LF_MIX, LF_SMOOTHSTEP, LF_EXP2, LF_LOG2,
       LF_ROUND, LF_MOD, LF_ATAN2, LF_LENGTH2, LF_SELECT };

#define MAX_PIXEL_EXPRS 256
typedef struct {
    const ir_header* h;
    const ir_layer* layers;
    const ir_stage* stages;
    const ir_law* law;
    uint32_t n_pix;                              /* stage-argument expressions that read the pixel */
    const ir_law* pix[MAX_PIXEL_EXPRS];
    uint32_t pix_len[MAX_PIXEL_EXPRS];
    uint32_t project_mode;                       /* 0 none, 1 flat, 2 dome, 3 cube, 4 led (codegen/project.rs) */
    float project_fov_deg;
    uint32_t project_count;
} ir_view;

/* A reverse Polish program is well formed when every operation is known, finds its operands on the stack, the stack
 * stays within the evaluator's 64 slots and one value is left; `frame_floats` = floats of a frame row it may read
 * (0: it may read none, as a gravity force law). */
static int prog_ok(const ir_law* prog, uint32_t n, uint32_t frame_floats, int pixel)
{
    int sp = 0;
    for (uint32_t i = 0; i < n; i++) {
        const ir_law* o = &prog[i];
        int need = 0, push = 1;
        switch (o->op) {
        case LAW_NUM: case LAW_DIST: case LAW_TIME: break;
        case LAW_UNIFORM: if (o->fn >= frame_floats) return 0; break;
        case LAW_PX: case LAW_PY: case LAW_UVX: case LAW_UVY: if (!pixel) return 0; break;
        case LAW_ADD: case LAW_SUB: case LAW_MUL: case LAW_DIV: case LAW_POW: case LAW_GT: case LAW_LT: need = 2; break;
        case LAW_NEG: need = 1; break;
        /* a call names a function of that many arguments */
        case LAW_CALL1: need = 1; if (!(o->fn <= LF_SIGN || (o->fn >= LF_EXP2 && o->fn <= LF_ROUND))) return 0; break;
        case LAW_CALL2: need = 2; if (!((o->fn >= LF_MIN && o->fn <= LF_STEP) || (o->fn >= LF_MOD && o->fn <= LF_LENGTH2))) return 0; break;
        case LAW_CALL3: need = 3; if (!((o->fn >= LF_CLAMP && o->fn <= LF_SMOOTHSTEP) || o->fn == LF_SELECT)) return 0; break;
        default: return 0;
        }
        if (sp < need) return 0;
        sp = sp - need + push;
        if (sp > 60) return 0;
    }
    return n == 0 || sp == 1;   /* the result is the one value left on the stack */
}

/* Everything the interpreter indexes with a value read from the blob is checked here, once, so that a damaged or
 * hostile blob is refused instead of read out of bounds. */
static int ir_validate(const ir_view* v)
{
    const ir_header* h = v->h;
    if (h->bounds > 2) return -1;
    for (uint32_t i = 0; i < h->n_layers; i++) {
        const ir_layer* L = &v->layers[i];
        if ((uint64_t)L->stage_begin + L->stage_count > h->n_stages || L->end_state > 2) return -1;
    }
    for (uint32_t i = 0; i < h->n_stages; i++) {
        const ir_stage* S = &v->stages[i];
        if (S->builtin > B_PALETTE || S->nargs > 4) return -1;
        for (int k = 0; k < 4; k++) {
            const ir_arg* a = &S->args[k];
            if (a->kind > 2 || (a->kind == 1 && a->slot >= h->n_uniforms) || (a->kind == 2 && a->slot >= BU_COUNT)) return -1;
            if ((a->pad >> 16) > v->n_pix) return -1;
        }
    }
    if (!prog_ok(v->law, h->n_law, 0, 0)) return -1;
    for (uint32_t i = 0; i < v->n_pix; i++)
        if (!prog_ok(v->pix[i], v->pix_len[i], BU_COUNT + h->n_uniforms, 1)) return -1;
    return 0;
}

static int ir_open(const void* blob, size_t len, ir_view* v)
{
    if (len < sizeof(ir_header)) return -1;
    const ir_header* h = (const ir_header*)blob;
    if (h->magic != 0x31524947u || h->total != len) return -1;
    if (h->n_layers > 4096 || h->n_stages > 65536 || h->n_uniforms > 4096 || h->n_law > 4096) return -1;   /* also keeps the sum from wrapping */
    size_t need = sizeof(ir_header) + (size_t)h->n_layers * sizeof(ir_layer) + (size_t)h->n_stages * sizeof(ir_stage) +
                  (size_t)h->n_uniforms * 64 + (size_t)h->n_law * sizeof(ir_law);
    if (need + 4 > len) return -1;
    const char* p = (const char*)blob + sizeof(ir_header);
    v->h = h;
    v->layers = (const ir_layer*)p; p += (size_t)h->n_layers * sizeof(ir_layer);
    v->stages = (const ir_stage*)p; p += (size_t)h->n_stages * sizeof(ir_stage);
    p += (size_t)h->n_uniforms * 64;
    v->law = (const ir_law*)p; p += (size_t)h->n_law * sizeof(ir_law);
    /* trailing section: count, then per program its length and operations */
    const char* end = (const char*)blob + len;
    memcpy(&v->n_pix, p, 4); p += 4;
    if (v->n_pix > MAX_PIXEL_EXPRS) return -1;
    for (uint32_t i = 0; i < v->n_pix; i++) {
        if (p + 4 > end) return -1;
        memcpy(&v->pix_len[i], p, 4); p += 4;
        if (v->pix_len[i] > 128 || p + (size_t)v->pix_len[i] * sizeof(ir_law) > end) return -1;
        v->pix[i] = (const ir_law*)p; p += (size_t)v->pix_len[i] * sizeof(ir_law);
    }
    /* trailing section: the projection block (mode + 1, fov_deg, led count, pad) */
    if (p + 20 > end) return -1;
    memcpy(&v->project_mode, p, 4); memcpy(&v->project_fov_deg, p + 4, 4); memcpy(&v->project_count, p + 8, 4);
    if (v->project_mode > 4) return -1;
    p += 16;
    /* trailing section: the frame-level programs of expression-valued stage arguments. Their values reach the
     * interpreter through the frame rows, so the programs are only checked and skipped. */
    uint32_t n_derived;
    memcpy(&n_derived, p, 4); p += 4;
    if (n_derived > 65535) return -1;
    for (uint32_t i = 0; i < n_derived; i++) {
        uint32_t n;
        if (p + 4 > end) return -1;
        memcpy(&n, p, 4); p += 4;
        if (n > 4096 || (size_t)(end - p) < (size_t)n * sizeof(ir_law)) return -1;
        if (!prog_ok((const ir_law*)p, n, BU_COUNT + h->n_uniforms, 0)) return -1;
        p += (size_t)n * sizeof(ir_law);
    }
    if (p != end) return -1;
    return ir_validate(v);
}

/* ── noise helpers ─────────────────────────────────────────────────────────────────────── */
static float hash2(float px, float py)   /* glsl.rs:134-138 */
{
    float x = gm_fract(px * 0.1031f);
    float y = gm_fract(py * 0.1031f);
    float z = gm_fract(px * 0.1031f);
    float d = x * (y + 33.33f) + y * (z + 33.33f) + z * (x + 33.33f);   /* dot(p3, p3.yzx + 33.33) */
    x = x + d; y = y + d; z = z + d;
    return gm_fract((x + y) * z);
}

static float noise2(float px, float py)   /* glsl.rs:141-150 */
{
    float ix = gm_floor(px), iy = gm_floor(py);
    float fx = gm_fract(px), fy = gm_fract(py);
    float ux = fx * fx * fmaf(-2.0f, fx, 3.0f);
    float uy = fy * fy * fmaf(-2.0f, fy, 3.0f);
    float a = hash2(ix, iy);
    float b = hash2(ix + 1.0f, iy + 0.0f);
    float c = hash2(ix + 0.0f, iy + 1.0f);
    float d = hash2(ix + 1.0f, iy + 1.0f);
    return fmaf(gm_mix(gm_mix(a, b, ux), gm_mix(c, d, ux), uy), 2.0f, -1.0f);
}

static int to_int_octaves(float o)   /* int(x): truncation; out-of-range values are clamped to [0, 32] */
{
    if (!(o >= 1.0f)) return 0;
    if (o >= 32.0f) return 32;
    return (int)o;
}

static float fbm2(float px, float py, int octaves, float persistence, float lacunarity)   /* glsl.rs:155-169 */
{
    float value = 0.0f, amplitude = 1.0f, frequency = 1.0f, max_val = 0.0f;
    float qx = px, qy = py;
    for (int i = 0; i < octaves; i++) {
        value = fmaf(noise2(qx * frequency, qy * frequency), amplitude, value);
        max_val = max_val + amplitude;
        amplitude = amplitude * persistence;
        frequency = frequency * lacunarity;
        float nx = fmaf(qx, 0.8f, -(qy * 0.6f));
        float ny = fmaf(qx, 0.6f, qy * 0.8f);
        qx = nx; qy = ny;
    }
    return value * gm_rcp(max_val);
}

static float permute1(float x) { return gm_mod((x * 34.0f + 1.0f) * x, 289.0f); }   /* glsl.rs:173-175 */

static float simplex2d(float vx, float vy)   /* glsl.rs:176-198 */
{
    const float Cx = 0.211324865405187f, Cy = 0.366025403784439f, Cz = -0.577350269189626f, Cw = 0.024390243902439f;
    float dv = vx * Cy + vy * Cy;
    float ix = gm_floor(vx + dv), iy = gm_floor(vy + dv);
    float di = ix * Cx + iy * Cx;
    float x0x = vx - ix + di, x0y = vy - iy + di;
    float i1x = (x0x > x0y) ? 1.0f : 0.0f;
    float i1y = (x0x > x0y) ? 0.0f : 1.0f;
    float x12x = x0x + Cx, x12y = x0y + Cx, x12z = x0x + Cz, x12w = x0y + Cz;
    x12x = x12x - i1x; x12y = x12y - i1y;
    ix = gm_mod(ix, 289.0f); iy = gm_mod(iy, 289.0f);
    float p0 = permute1(permute1(iy + 0.0f) + ix + 0.0f);
    float p1 = permute1(permute1(iy + i1y) + ix + i1x);
    float p2 = permute1(permute1(iy + 1.0f) + ix + 1.0f);
    float m0 = gm_max(0.5f - (x0x * x0x + x0y * x0y), 0.0f);
    float m1 = gm_max(0.5f - (x12x * x12x + x12y * x12y), 0.0f);
    float m2 = gm_max(0.5f - (x12z * x12z + x12w * x12w), 0.0f);
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    float xv0 = 2.0f * gm_fract(p0 * Cw) - 1.0f;
    float xv1 = 2.0f * gm_fract(p1 * Cw) - 1.0f;
    float xv2 = 2.0f * gm_fract(p2 * Cw) - 1.0f;
    float h0 = gm_abs(xv0) - 0.5f, h1 = gm_abs(xv1) - 0.5f, h2 = gm_abs(xv2) - 0.5f;
    float ox0 = gm_floor(xv0 + 0.5f), ox1 = gm_floor(xv1 + 0.5f), ox2 = gm_floor(xv2 + 0.5f);
    float a00 = xv0 - ox0, a01 = xv1 - ox1, a02 = xv2 - ox2;
    m0 = m0 * (1.79284291400159f - 0.85373472095314f * (a00 * a00 + h0 * h0));
    m1 = m1 * (1.79284291400159f - 0.85373472095314f * (a01 * a01 + h1 * h1));
    m2 = m2 * (1.79284291400159f - 0.85373472095314f * (a02 * a02 + h2 * h2));
    float g0 = a00 * x0x + h0 * x0y;
    float g1 = a01 * x12x + h1 * x12y;
    float g2 = a02 * x12z + h2 * x12w;
    return 130.0f * (m0 * g0 + m1 * g1 + m2 * g2);
}

static float voronoi2d(float px, float py, float scale)   /* glsl.rs:202-219 */
{
    float psx = px * scale, psy = py * scale;
    float ipx = gm_floor(psx), ipy = gm_floor(psy);
    float fpx = gm_fract(psx), fpy = gm_fract(psy);
    float d = 1.0f;
    for (int y = -1; y <= 1; y++) {
        for (int x = -1; x <= 1; x++) {
            float nx = (float)x, ny = (float)y;
            float cx = ipx + nx, cy = ipy + ny;
            float hx = gm_fract(gm_sin(cx * 127.1f + cy * 311.7f) * 43758.5453f);
            float hy = gm_fract(gm_sin(cx * 269.5f + cy * 183.3f) * 43758.5453f);
            float ptx = nx + hx - fpx, pty = ny + hy - fpy;
            d = gm_min(d, ptx * ptx + pty * pty);
        }
    }
    return sqrtf(d);
}

static float apply_glow(float d, float intensity)   /* glsl.rs:102-107 */
{
    float core = gm_smoothstep(0.005f, -0.005f, d);
    float halo = intensity * gm_rcp(fmaf(gm_max(d, 0.0f) * gm_max(d, 0.0f) * intensity * intensity, 16.0f, 1.0f));   /* reciprocal, multiply */
    return core + halo;
}

/* palette table (glsl.rs:424-434): a, b, c, d per named set */
static const float PALETTES[8][12] = {
    /* fire    */ {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 1.0f, 1.0f, 1.0f, 0.00f, 0.10f, 0.20f},
    /* ice     */ {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 1.0f, 1.0f, 1.0f, 0.30f, 0.20f, 0.20f},
    /* rainbow */ {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 1.0f, 1.0f, 1.0f, 0.00f, 0.33f, 0.67f},
    /* ocean   */ {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 1.0f, 0.7f, 0.4f, 0.00f, 0.15f, 0.20f},
    /* forest  */ {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 1.0f, 1.0f, 0.5f, 0.80f, 0.90f, 0.30f},
    /* neon    */ {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 2.0f, 1.0f, 0.0f, 0.50f, 0.20f, 0.25f},
    /* sunset  */ {0.8f, 0.5f, 0.4f, 0.2f, 0.4f, 0.2f, 2.0f, 1.0f, 1.0f, 0.00f, 0.25f, 0.25f},
    /* plasma  */ {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 1.0f, 1.0f, 1.0f, 0.00f, 0.10f, 0.20f},
};

/* ── one pixel ─────────────────────────────────────────────────────────────────────────── */
typedef struct {
    int wgsl;        /* builtins as worded by the WebGPU emitter (wgsl.rs:268-540) where they differ */
    int project;     /* apply the projection block to v_uv (game-compiler/src/codegen/project.rs) */
    int lens_post;   /* run the lens post chain (the layer marked pad0 = 1) on the composited colour; the reference's
                        emitters ignore the chain, so 0 is its behaviour */
    float time;      /* fract(u_time / 120) * 120 */
    float aspect;    /* u_resolution.x / u_resolution.y */
    const float* row;/* this frame's uniform row: 10 built-ins then user uniforms */
    const float* derived; /* per-frame values of expression-valued arguments, or NULL: the reference's behaviour
                             (stages.rs:8-22 substitutes the parameter default, which kind/slot/lit hold) */
    int animate;     /* expression-valued arguments are live: frame-level ones read `derived`, those that also
                        read the pixel are evaluated here, in f32, where the stage runs */
} frame_consts;

static float law_call(uint32_t fn, float a, float b, float c)
{
    switch (fn) {
    case LF_SIN: return gm_sin(a); case LF_COS: return gm_cos(a); case LF_TAN: return gm_tan(a);
    case LF_ABS: return gm_abs(a); case LF_SQRT: return sqrtf(a); case LF_EXP: return gm_exp(a);
    case LF_LOG: return gm_log(a); case LF_FLOOR: return gm_floor(a); case LF_CEIL: return -gm_floor(-a);
    case LF_FRACT: return gm_fract(a); case LF_SIGN: return a > 0.0f ? 1.0f : (a < 0.0f ? -1.0f : 0.0f);
    case LF_EXP2: return gm_exp2(a); case LF_LOG2: return gm_log2(a);
    case LF_ROUND: return rintf(a);                                   /* WGSL round: to nearest, ties to even */
    case LF_MOD: return a - b * truncf(a * gm_rcp(b));                /* expr.rs:103-109 `%`: truncated remainder, quotient as reciprocal-multiply */
    case LF_ATAN2: return gm_atan2(a, b);
    case LF_LENGTH2: return gm_length2(a, b);                         /* length(vec2(a, b)) */
    case LF_MIN: return gm_min(a, b); case LF_MAX: return gm_max(a, b); case LF_POW: return gm_pow(a, b);
    case LF_STEP: return gm_step(a, b);
    case LF_CLAMP: return gm_clamp(a, b, c); case LF_MIX: return gm_mix(a, b, c);
    case LF_SMOOTHSTEP: return gm_smoothstep(a, b, c);
    case LF_SELECT: return (a != 0.0f && a == a) ? b : c;             /* expr.rs:132-142 `c ? t : f` = select(f, t, c) */
    default: return 0.0f;
    }
}

/* What a reverse Polish program may read: the gravity force law uses dist and time (gravity.rs:10-41); a stage
 * argument that reads the pixel uses the frame row (slot 0 = the wrapped time), p and uv. */
typedef struct { float dist, time; const float* row; float wtime, px, py, uvx, uvy; } law_vars;

static float eval_rpn(const ir_law* law, uint32_t n, const law_vars* V)
{
    float st[64];
    int sp = 0;
    if (n == 0) return 1.0f;
    for (uint32_t i = 0; i < n; i++) {
        const ir_law* o = &law[i];
        switch (o->op) {
        case LAW_NUM: st[sp++] = o->val; break;
        case LAW_DIST: st[sp++] = V->dist; break;
        case LAW_TIME: st[sp++] = V->time; break;
        case LAW_UNIFORM: st[sp++] = o->fn == BU_TIME ? V->wtime : V->row[o->fn]; break;
        case LAW_PX: st[sp++] = V->px; break;
        case LAW_PY: st[sp++] = V->py; break;
        case LAW_UVX: st[sp++] = V->uvx; break;
        case LAW_UVY: st[sp++] = V->uvy; break;
        case LAW_ADD: sp--; st[sp - 1] = st[sp - 1] + st[sp]; break;
        case LAW_SUB: sp--; st[sp - 1] = st[sp - 1] - st[sp]; break;
        case LAW_MUL: sp--; st[sp - 1] = st[sp - 1] * st[sp]; break;
        case LAW_DIV: sp--; st[sp - 1] = st[sp - 1] / st[sp]; break;
        case LAW_POW: sp--; st[sp - 1] = gm_pow(st[sp - 1], st[sp]); break;
        case LAW_GT: sp--; st[sp - 1] = st[sp - 1] > st[sp] ? 1.0f : 0.0f; break;
        case LAW_LT: sp--; st[sp - 1] = st[sp - 1] < st[sp] ? 1.0f : 0.0f; break;
        case LAW_NEG: st[sp - 1] = -st[sp - 1]; break;
        case LAW_CALL1: st[sp - 1] = law_call(o->fn, st[sp - 1], 0.0f, 0.0f); break;
        case LAW_CALL2: sp--; st[sp - 1] = law_call(o->fn, st[sp - 1], st[sp], 0.0f); break;
        case LAW_CALL3: sp -= 2; st[sp - 1] = law_call(o->fn, st[sp - 1], st[sp], st[sp + 1]); break;
        default: break;
        }
        if (sp >= 60) return 0.0f;
    }
    return st[0];
}

static float eval_law(const ir_law* law, uint32_t n, float dist, float time)
{
    law_vars V = {dist, time, NULL, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    return eval_rpn(law, n, &V);
}

static inline float arg_value(const ir_arg* a, const frame_consts* fc, const ir_view* ir, float px, float py, float uvx, float uvy)
{
    if (fc->animate && a->pad) {   /* low half: derived index + 1; high half: pixel-expression index + 1 */
        const uint32_t pix = a->pad >> 16, der = a->pad & 0xffffu;
        if (pix && pix <= ir->n_pix) {
            law_vars V = {0.0f, 0.0f, fc->row, fc->time, px, py, uvx, uvy};
            return eval_rpn(ir->pix[pix - 1], ir->pix_len[pix - 1], &V);
        }
        if (der && fc->derived) return fc->derived[der - 1];
    }
    if (a->kind == 0) return a->lit;
    if (a->kind == 1) return fc->row[BU_COUNT + a->slot];
    return a->slot == BU_TIME ? fc->time : fc->row[a->slot];
}

static inline uint32_t pack_unorm8(float c)   /* framebuffer write: c * 255 to the nearest integer (ties to even), saturated; NaN -> 0 */
{
    float t = rintf(c * 255.0f);              /* the default rounding mode is to nearest even */
    if (!(t > 0.0f)) return 0u;
    return t >= 255.0f ? 255u : (uint32_t)t;
}
static inline float unorm8_to_float(uint32_t b) { return (float)b * 0x1.010102p-8f; }   /* texel fetch: byte times RN(1 / 255) */

/* ── WGSL-semantics variants (game-compiler/src/codegen/wgsl.rs) ───────────────────────── */
static inline float w_rem(float x, float y) { return x - y * truncf(x * gm_rcp(y)); }   /* WGSL `%` on f32 */
static inline float w_step(float edge, float x) { return edge <= x ? 1.0f : 0.0f; }
static inline float w_fold(float a, float seg) { return w_rem(w_rem(a, seg) + seg, seg); }   /* wgsl.rs:291 */

static void w_hash2v(float x, float y, float* hx, float* hy)   /* wgsl.rs:206-209 */
{
    float fx = gm_fract(gm_sin(x * 127.1f + y * 311.7f) * 43758.5453f);
    float fy = gm_fract(gm_sin(x * 269.5f + y * 183.3f) * 43758.5453f);
    *hx = -1.0f + 2.0f * fx; *hy = -1.0f + 2.0f * fy;
}

static float w_simplex2(float px, float py)   /* wgsl.rs:211-222 */
{
    const float K1 = 0.366025404f, K2 = 0.211324865f;
    float sk = (px + py) * K1;
    float ix = gm_floor(px + sk), iy = gm_floor(py + sk);
    float t = (ix + iy) * K2;
    float ax = px - ix + t, ay = py - iy + t;
    float ox = w_step(ay, ax), oy = w_step(ax, ay);
    float bx = ax - ox + K2, by = ay - oy + K2;
    float cx = ax - 1.0f + 2.0f * K2, cy = ay - 1.0f + 2.0f * K2;
    float h0 = gm_max(0.5f - (ax * ax + ay * ay), 0.0f);
    float h1 = gm_max(0.5f - (bx * bx + by * by), 0.0f);
    float h2 = gm_max(0.5f - (cx * cx + cy * cy), 0.0f);
    float g0x, g0y, g1x, g1y, g2x, g2y;
    w_hash2v(ix, iy, &g0x, &g0y);
    w_hash2v(ix + ox, iy + oy, &g1x, &g1y);
    w_hash2v(ix + 1.0f, iy + 1.0f, &g2x, &g2y);
    float n0 = h0 * h0 * h0 * h0 * (ax * g0x + ay * g0y);
    float n1 = h1 * h1 * h1 * h1 * (bx * g1x + by * g1y);
    float n2 = h2 * h2 * h2 * h2 * (cx * g2x + cy * g2y);
    return n0 * 70.0f + n1 * 70.0f + n2 * 70.0f;
}

static float w_voronoi2(float px, float py, float scale)   /* wgsl.rs:187-201 */
{
    float spx = px * scale, spy = py * scale;
    float ipx = gm_floor(spx), ipy = gm_floor(spy);
    float fpx = gm_fract(spx), fpy = gm_fract(spy);
    float d = 1.0f;
    for (int y = -1; y <= 1; y++) {
        for (int x = -1; x <= 1; x++) {
            float nx = (float)x, ny = (float)y;
            float pv = hash2(ipx + nx, ipy + ny);
            float ptx = nx + pv - fpx, pty = ny + gm_fract(pv * 17.0f) - fpy;
            d = gm_min(d, ptx * ptx + pty * pty);
        }
    }
    return sqrtf(d);
}

static inline float w_aces(float x)   /* wgsl.rs:409-419 */
{
    return gm_clamp((x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f), 0.0f, 1.0f);
}

typedef struct { float px, py, sdf, cr, cg, cb, ca; } pix_state;

/* Applies stage S in the WebGPU wording when that differs from the WebGL2 one; returns 0 when the
 * builtin is shared and the caller's switch must run. */
static int wgsl_stage(uint32_t builtin, pix_state* s, float a0, float a1, float a2, float uvx, float uvy, float time)
{
    switch (builtin) {
    case B_STAR: {                                                                                 /* wgsl.rs:282-294 */
        float angle = gm_atan2(s->py, s->px);
        float r = gm_length2(s->px, s->py);
        float seg = 6.28318f * gm_rcp(a0);
        float a = gm_abs(w_fold(angle, seg) - seg * 0.5f);
        s->sdf = r * gm_cos(a) - a1 + (r * gm_sin(a) - a2) * 0.5f;
        return 1;
    }
    case B_POLYGON: {                                                                              /* wgsl.rs:301-312 */
        float a = gm_atan2(s->py, s->px);
        float r = gm_length2(s->px, s->py);
        float seg = 6.28318f * gm_rcp(a0);
        float ha = gm_abs(w_fold(a, seg) - seg * 0.5f);
        s->sdf = r * gm_cos(ha) - a1;
        return 1;
    }
    case B_SIMPLEX: s->sdf = w_simplex2(s->px * a0, s->py * a0); return 1;                         /* wgsl.rs:320-323 */
    case B_VORONOI: s->sdf = w_voronoi2(s->px, s->py, a0); return 1;                               /* wgsl.rs:324-327 */
    case B_CONCENTRIC_WAVES: {                                                                     /* wgsl.rs:328-334 */
        float len = gm_length2(s->px, s->py);
        float r = len * a2;
        s->sdf = gm_sin(r - time * 2.0f) * a0 * gm_exp(-len * a1);
        return 1;
    }
    case B_THRESHOLD: s->sdf = w_step(a0, s->sdf); return 1;                                       /* wgsl.rs:501-504 */
    case B_VIGNETTE: {                                                                             /* wgsl.rs:392-398 */
        float d = gm_length2(uvx, uvy);
        float vig = gm_smoothstep(a1 + 0.3f, a1 - 0.2f, d);
        float f = gm_mix(1.0f, vig, a0);
        s->cr = s->cr * f; s->cg = s->cg * f; s->cb = s->cb * f;
        return 1;
    }
    case B_CHROMATIC: {                                                                            /* wgsl.rs:399-408 */
        float d = gm_length2(uvx, uvy) * a0;
        s->cr = s->cr + d * 0.5f;
        s->cb = s->cb - d * 0.5f;
        return 1;
    }
    case B_TONEMAP: s->cr = w_aces(s->cr * a0); s->cg = w_aces(s->cg * a0); s->cb = w_aces(s->cb * a0); return 1;
    case B_SCANLINES: {                                                                            /* wgsl.rs:420-425 */
        float scan = gm_sin(uvy * a0 * 3.14159f) * 0.5f + 0.5f;
        float f = 1.0f - a1 * (1.0f - scan);
        s->cr = s->cr * f; s->cg = s->cg * f; s->cb = s->cb * f;
        return 1;
    }
    case B_GLITCH: {                                                                               /* wgsl.rs:431-437 */
        float gt = gm_floor(time * 8.0f);
        float off = (gm_fract(gm_sin(gt * 12.9898f) * 43758.5453f) - 0.5f) * a0 * 0.1f;
        float block = w_step(0.9f - a0 * 0.3f, gm_fract(gm_sin(uvy * 20.0f * 12.9898f + gt * 78.233f) * 43758.5453f));
        s->cr = gm_mix(s->cr, s->cr + off, block);
        s->cg = gm_mix(s->cg, s->cg - off * 0.5f, block);
        s->cb = gm_mix(s->cb, s->cb, block);
        return 1;
    }
    case B_MIRROR:                                                                                 /* wgsl.rs:464-468 */
        if (a0 > 0.5f) s->py = gm_abs(s->py); else s->px = gm_abs(s->px);
        return 1;
    case B_REPEAT: {                                                                               /* wgsl.rs:469-475 */
        float rep = 2.0f * gm_rcp(a0);
        s->px = w_fold(s->px, rep) - rep * 0.5f;
        s->py = w_fold(s->py, rep) - rep * 0.5f;
        return 1;
    }
    case B_CURL_NOISE: {                                                                           /* wgsl.rs:481-488 */
        const float eps = 0.01f;
        float a = noise2((s->px + eps) * a0, (s->py + 0.0f) * a0), b = noise2((s->px - eps) * a0, (s->py - 0.0f) * a0);
        float c = noise2((s->px + 0.0f) * a0, (s->py + eps) * a0), d = noise2((s->px - 0.0f) * a0, (s->py - eps) * a0);
        float cnx = a - b, cny = c - d;
        const float inv = gm_rcp(2.0f * eps);
        s->px = s->px + cny * a1 * inv;
        s->py = s->py + -cnx * a1 * inv;
        return 1;
    }
    case B_DISPLACE: {                                                                             /* wgsl.rs:489-492 */
        float n0 = noise2(s->px * 3.0f, s->py * 3.0f);
        float n1 = noise2(s->px * 3.0f + 5.0f, s->py * 3.0f + 3.0f);
        s->px = s->px + n0 * a0; s->py = s->py + n1 * a0;
        return 1;
    }
    case B_SPECTRUM: {                                                                             /* wgsl.rs:520-534 */
        float d = gm_length2(s->px, s->py);
        float sb = gm_smoothstep(0.35f, 0.15f, d) * a0;
        float sm = gm_smoothstep(0.55f, 0.35f, d) * gm_smoothstep(0.15f, 0.35f, d) * a1;
        float st = gm_smoothstep(0.75f, 0.55f, d) * gm_smoothstep(0.35f, 0.55f, d) * a2;
        s->cr = sb * 1.0f + sm * 0.1f + st * 0.2f;
        s->cg = sb * 0.2f + sm * 1.0f + st * 0.3f;
        s->cb = sb * 0.1f + sm * 0.3f + st * 1.0f;
        s->ca = 1.0f;
        return 1;
    }
    default:
        return 0;
    }
}

/* uvx, uvy: `uv` of glsl.rs:54. prev: previous frame's RGBA8 texel at this pixel. out: RGBA floats. */
static void shade_pixel(const ir_view* ir, const frame_consts* fc, float uvx, float uvy, uint32_t prev, float out[4])
{
    const float time = fc->time;
    float fin_r = 0.0f, fin_g = 0.0f, fin_b = 0.0f, fin_a = 1.0f;   /* final_color (glsl.rs:68) */
    float frag[4] = {0.0f, 0.0f, 0.0f, 1.0f};
    const int multi = (int)ir->h->multi_layer;

    for (uint32_t li = 0; li < ir->h->n_layers; li++) {
        const ir_layer* L = &ir->layers[li];
        if (!L->is_pipeline) continue;                         /* glsl.rs:229-232 */
        const int post = L->pad0 == 1u;
        if (post && !fc->lens_post) continue;
        float px = uvx * fc->aspect, py = uvy;                 /* glsl.rs:240 */
        float sdf = 0.0f;
        float cr = 0.0f, cg = 0.0f, cb = 0.0f, ca = 0.0f;
        if (post) {                                            /* starts from the frame's colour so far */
            if (multi) { cr = fin_r; cg = fin_g; cb = fin_b; ca = fin_a; }
            else { cr = frag[0]; cg = frag[1]; cb = frag[2]; ca = frag[3]; }
        }
        for (uint32_t si = 0; si < L->stage_count; si++) {
            const ir_stage* S = &ir->stages[L->stage_begin + si];
            const float a0 = arg_value(&S->args[0], fc, ir, px, py, uvx, uvy), a1 = arg_value(&S->args[1], fc, ir, px, py, uvx, uvy);
            const float a2 = arg_value(&S->args[2], fc, ir, px, py, uvx, uvy), a3 = arg_value(&S->args[3], fc, ir, px, py, uvx, uvy);
            if (fc->wgsl) {
                pix_state st = {px, py, sdf, cr, cg, cb, ca};
                if (wgsl_stage(S->builtin, &st, a0, a1, a2, uvx, uvy, time)) {
                    px = st.px; py = st.py; sdf = st.sdf; cr = st.cr; cg = st.cg; cb = st.cb; ca = st.ca;
                    continue;
                }
            }
            switch (S->builtin) {
            /* SDF generators */
            case B_CIRCLE: sdf = gm_length2(px, py) - a0; break;                                   /* :265-268, :96-100 */
            case B_RING: sdf = gm_abs(gm_length2(px, py) - a0) - a1; break;                        /* :269-273 */
            case B_STAR: {                                                                         /* :274-288 */
                float an = 3.14159265359f / a0;
                float a = gm_atan2(py, px);
                float sector = gm_mod(a + an, 2.0f * an) - an;
                float len = gm_length2(px, py);
                float spx = gm_cos(sector) * len, spy = gm_abs(gm_sin(sector)) * len;
                float apx = a1, apy = 0.0f;
                float bpx = a2 * gm_cos(an), bpy = a2 * gm_sin(an);
                float bax = bpx - apx, bay = bpy - apy;
                float t = gm_clamp(((spx - apx) * bax + (spy - apy) * bay) / (bax * bax + bay * bay), 0.0f, 1.0f);
                sdf = gm_length2(spx - apx - bax * t, spy - apy - bay * t);
                break;
            }
            case B_BOX: {                                                                          /* :289-294 */
                float dx = gm_abs(px) - a0, dy = gm_abs(py) - a1;
                sdf = gm_length2(gm_max(dx, 0.0f), gm_max(dy, 0.0f)) + gm_min(gm_max(dx, dy), 0.0f);
                break;
            }
            case B_POLYGON: {                                                                      /* :295-304 */
                float n = a0;
                float a = gm_atan2(py, px);
                float r = gm_length2(px, py);
                float an = 6.28318530718f / n;
                float sector = gm_cos(gm_floor(0.5f + a / an) * an - a) * r;
                sdf = sector - a1;
                break;
            }
            case B_SIMPLEX: sdf = simplex2d(px * a0, py * a0); break;                              /* :305-308 */
            case B_VORONOI: sdf = voronoi2d(px, py, a0); break;                                    /* :309-312 */
            case B_CONCENTRIC_WAVES: {                                                             /* :313-319 */
                float dist = gm_length2(px, py);
                sdf = a0 * gm_sin(dist * a2 * 6.28318530718f - time * a1);
                break;
            }
            case B_FBM: sdf = fbm2(px * a0, py * a0, to_int_octaves(a1), a2, a3); break;            /* :320-326 */

            /* transforms */
            case B_ROTATE: {                                                                       /* :329-334 */
                float rc = gm_cos(a0), rs = gm_sin(a0);
                float nx = px * rc - py * rs, ny = px * rs + py * rc;
                px = nx; py = ny;
                break;
            }
            case B_TRANSLATE: px = px - a0; py = py - a1; break;                                   /* :335-339 */
            case B_SCALE: { float r = gm_rcp(a0); px = px * r; py = py * r; } break;                                       /* :340-343 */
            case B_TWIST: {                                                                        /* :344-349 */
                float tw = py * a0;
                float c = gm_cos(tw), s = gm_sin(tw);
                float nx = px * c - py * s, ny = px * s + py * c;
                px = nx; py = ny;
                break;
            }
            case B_MIRROR:                                                                         /* :350-354 */
                if (a0 < 0.5f) px = gm_abs(px); else py = gm_abs(py);
                break;
            case B_REPEAT: {                                                                       /* :355-361 */
                float ra = 6.28318530718f / a0;
                float a = gm_atan2(py, px);
                a = gm_mod(a + ra * 0.5f, ra) - ra * 0.5f;
                float len = gm_length2(px, py);
                px = gm_cos(a) * len; py = gm_sin(a) * len;
                break;
            }
            case B_DOMAIN_WARP: {                                                                  /* :362-366 */
                float n0 = noise2(px * a1, py * a1);
                float n1 = noise2(px * a1 + 5.2f, py * a1 + 1.3f);
                px = px + n0 * a0; py = py + n1 * a0;
                break;
            }
            case B_CURL_NOISE: {                                                                   /* :367-377 */
                const float eps = 0.001f;
                float n0 = noise2(px * a0 + 0.0f, py * a0 + eps);
                float n1 = noise2(px * a0 - 0.0f, py * a0 - eps);
                float n2 = noise2(px * a0 + eps, py * a0 + 0.0f);
                float n3 = noise2(px * a0 - eps, py * a0 - 0.0f);
                float cx = (n0 - n1) / (2.0f * eps);
                float cy = -(n2 - n3) / (2.0f * eps);
                float len = gm_length2(cx, cy);
                px = px + cx / len * a1; py = py + cy / len * a1;                                  /* normalize = v / length(v) */
                break;
            }
            case B_DISPLACE: {                                                                     /* :378-381 */
                float t5 = time * 0.5f;
                float n0 = noise2(px * 3.0f + t5, py * 3.0f + t5);
                float n1 = noise2(px * 3.0f + 5.2f + t5, py * 3.0f + 1.3f + t5);
                px = px + n0 * a0; py = py + n1 * a0;
                break;
            }

            /* SDF modifiers */
            case B_MASK_ARC: {                                                                     /* :384-389 */
                float theta = gm_atan2(px, py) + 3.14159265359f;
                sdf = (theta < a0) ? sdf : 999.0f;
                break;
            }
            case B_THRESHOLD: sdf = (sdf < a0) ? -1.0f : 1.0f; break;                              /* :390-393 */
            case B_ONION: sdf = gm_abs(sdf) - a0; break;                                           /* :394-397 */
            case B_ROUND: sdf = sdf - a0; break;                                                   /* :398-401 */

            /* bridges */
            case B_GLOW: { float g = apply_glow(sdf, a0); cr = cg = cb = g; ca = 1.0f; break; }    /* :404-408 */
            case B_SHADE: {                                                                        /* :409-416 */
                float m = gm_smoothstep(0.005f, -0.005f, sdf);
                cr = a0 * m; cg = a1 * m; cb = a2 * m; ca = 1.0f;
                break;
            }
            case B_EMISSIVE: { float g = apply_glow(sdf, a0); cr = cg = cb = g; ca = g; break; }   /* :417-421 */
            case B_PALETTE: {                                                                      /* :422-437, :222-226 */
                const float* P = PALETTES[S->aux < 8 ? S->aux : 2];
                float t = gm_clamp(sdf * 0.5f + 0.5f, 0.0f, 1.0f);
                cr = P[0] + P[3] * gm_cos(6.28318f * (P[6] * t + P[9]));
                cg = P[1] + P[4] * gm_cos(6.28318f * (P[7] * t + P[10]));
                cb = P[2] + P[5] * gm_cos(6.28318f * (P[8] * t + P[11]));
                ca = 1.0f;
                break;
            }

            /* colour processors */
            case B_TINT: cr = cr * a0; cg = cg * a1; cb = cb * a2; ca = 1.0f; break;               /* :440-445 */
            case B_BLOOM: {                                                                        /* :446-452 */
                float lum = cr * 0.299f + cg * 0.587f + cb * 0.114f;
                float add = gm_max(lum - a0, 0.0f) * a1;
                cr = cr + add; cg = cg + add; cb = cb + add; ca = 1.0f;
                break;
            }
            case B_GRAIN: {                                                                        /* :453-457 */
                float n = gm_fract(gm_sin(px * 12.9898f + py * 78.233f + time) * 43758.5453f);
                float add = (n - 0.5f) * a0;
                cr = cr + add; cg = cg + add; cb = cb + add;
                break;
            }
            case B_VIGNETTE: {                                                                     /* :458-464 */
                float dist = gm_length2(uvx, uvy);
                float f = 1.0f - gm_smoothstep(a1, a1 + a0, dist);
                cr = cr * f; cg = cg * f; cb = cb * f;
                break;
            }
            case B_CHROMATIC: {                                                                    /* :465-472 */
                float dirx = uvx * gm_rsqrt(fmaf(uvx, uvx, uvy * uvy)) * a0;                       /* normalize(uv) = uv * inversesqrt(dot(uv, uv)) */
                float r = cr, b = cb;
                cr = r * (1.0f + dirx);
                cb = b * (1.0f - dirx);
                break;
            }
            case B_TONEMAP: {                                                                      /* :473-478 */
                float r = cr * a0, g = cg * a0, b = cb * a0;
                cr = r / (r + 1.0f); cg = g / (g + 1.0f); cb = b / (b + 1.0f);
                break;
            }
            case B_SCANLINES: {                                                                    /* :479-484 */
                float v = gm_sin(uvy * a0 * 3.14159265359f) * a1;
                float f = 1.0f - v * 0.5f;
                cr = cr * f; cg = cg * f; cb = cb * f;
                break;
            }
            case B_SATURATE_COLOR: {                                                               /* :485-489 */
                float lum = cr * 0.299f + cg * 0.587f + cb * 0.114f;
                cr = gm_mix(lum, cr, a0); cg = gm_mix(lum, cg, a0); cb = gm_mix(lum, cb, a0);
                break;
            }
            case B_GLITCH: {                                                                       /* :490-501 */
                float gt = gm_floor(time * 10.0f);
                float n = gm_fract(gm_sin(gt * 43758.5453f + py * 100.0f) * 43758.5453f);
                float shift = (n - 0.5f) * a0 * 0.1f;
                float block = gm_step(1.0f - a0 * 0.3f, gm_fract(gm_sin(gt * 12.9898f + gm_floor(py * 20.0f)) * 43758.5453f));
                cr = cr + shift * block;
                cb = cb - shift * block;
                break;
            }
            case B_BLEND: cr = cr * a0; cg = cg * a0; cb = cb * a0; break;                         /* :502-505 */

            /* full-screen generators */
            case B_GRADIENT: { float t = uvy * 0.5f + 0.5f; cr = cg = cb = t; ca = 1.0f; break; }  /* :508-515 */
            case B_SPECTRUM: {                                                                     /* :516-525 */
                float sx = uvx * 0.5f + 0.5f;
                cr = gm_smoothstep(0.0f, 0.33f, sx) * (fc->row[BU_BASS] + a0);
                cg = gm_smoothstep(0.33f, 0.66f, sx) * (fc->row[BU_MID] + a1);
                cb = gm_smoothstep(0.66f, 1.0f, sx) * (fc->row[BU_TREBLE] + a2);
                ca = 1.0f;
                break;
            }
            default: break;
            }
            (void)a3;
        }
        if (L->end_state != 2) continue;   /* no color_result: the layer contributes nothing */
        if (post) {
            if (multi) { fin_r = cr; fin_g = cg; fin_b = cb; fin_a = ca; }
            else { frag[0] = cr; frag[1] = cg; frag[2] = cb; frag[3] = ca; }
            continue;
        }
        if (L->has_memory) {               /* memory.rs:38-45 */
            float pr = unorm8_to_float(prev & 0xffu), pg = unorm8_to_float((prev >> 8) & 0xffu);
            float pb = unorm8_to_float((prev >> 16) & 0xffu), pa = unorm8_to_float(prev >> 24);
            cr = gm_mix(cr, pr, L->decay); cg = gm_mix(cg, pg, L->decay);
            cb = gm_mix(cb, pb, L->decay); ca = gm_mix(ca, pa, L->decay);
        }
        if (multi) {                       /* glsl.rs:251-255 screen blend */
            fin_r = fin_r + cr - fin_r * cr;
            fin_g = fin_g + cg - fin_g * cg;
            fin_b = fin_b + cb - fin_b * cb;
            fin_a = 1.0f;
        } else {                           /* glsl.rs:257 */
            frag[0] = cr; frag[1] = cg; frag[2] = cb; frag[3] = ca;
        }
    }
    if (multi) { out[0] = fin_r; out[1] = fin_g; out[2] = fin_b; out[3] = fin_a; }
    else { out[0] = frag[0]; out[1] = frag[1]; out[2] = frag[2]; out[3] = frag[3]; }
}

/* ── frame ─────────────────────────────────────────────────────────────────────────────── */
typedef struct {
    const ir_view* ir;
    frame_consts fc;
    uint32_t w, h, row_begin, row_end;     /* image rows, top-first */
    uint32_t band_row0;                    /* first row stored in out/prev */
    const uint32_t* prev;                  /* previous frame, same band layout, or NULL */
    uint32_t* out;
} frame_job;

static void render_rows(const frame_job* J)
{
    const float inv_w = 1.0f / (float)J->w, inv_h = 1.0f / (float)J->h;
    for (uint32_t ty = J->row_begin; ty < J->row_end; ty++) {
        uint32_t gy = J->h - 1u - ty;                                  /* GL row index, 0 = bottom */
        float vy = ((float)gy + 0.5f) * inv_h;                         /* v_uv (vertex :541-555, pixel centre) */
        float uvy = vy * 2.0f - 1.0f;                                  /* glsl.rs:54 */
        for (uint32_t x = 0; x < J->w; x++) {
            float vx = ((float)x + 0.5f) * inv_w;
            float uvx = vx * 2.0f - 1.0f;
            float puvy = uvy;
            if (J->fc.project && J->ir->project_mode == 2) {           /* dome: project.rs:62-69, per pixel */
                float fov_rad = J->ir->project_fov_deg * 3.14159265359f / 180.0f;
                float cx = vx - 0.5f, cy = vy - 0.5f;
                float r = gm_length2(cx, cy) * 2.0f;
                float theta = gm_atan2(cy, cx);
                float phi = r * fov_rad * 0.5f;
                float sp = gm_sin(phi);
                float wvx = sp * gm_cos(theta) * 0.5f + 0.5f, wvy = sp * gm_sin(theta) * 0.5f + 0.5f;
                uvx = wvx * 2.0f - 1.0f; puvy = wvy * 2.0f - 1.0f;
            } else if (J->fc.project && J->ir->project_mode == 4) {    /* led: project.rs:128-137 */
                float count = (float)J->ir->project_count;
                float wvx = gm_floor(vx * count) / count;              /* t = f32(led_idx) / f32(count) */
                uvx = wvx * 2.0f - 1.0f; puvy = 0.5f * 2.0f - 1.0f;
            }
            size_t idx = (size_t)(ty - J->band_row0) * J->w + x;
            uint32_t prev = J->prev ? J->prev[idx] : 0u;
            float c[4];
            shade_pixel(J->ir, &J->fc, uvx, puvy, prev, c);
            J->out[idx] = pack_unorm8(c[0]) | (pack_unorm8(c[1]) << 8) | (pack_unorm8(c[2]) << 16) | (pack_unorm8(c[3]) << 24);
        }
    }
}

static void* render_thread(void* p) { render_rows((const frame_job*)p); return NULL; }

/* Renders image rows [row0, row0+rows) of one frame. uniforms: one row (10 + n_uniforms floats).
 * prev/out hold only the band, top row first, RGBA8 (R in the low byte). prev may be NULL (= all
 * zero). nthreads <= 1 renders on the calling thread. Returns 0, or -1 for a malformed blob. */
int game_oracle_render_ex(const void* blob, size_t blob_len, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows,
                          const float* uniforms, const uint8_t* prev_rgba8, uint8_t* out_rgba8, int nthreads, int semantics);
int game_oracle_render_ex2(const void* blob, size_t blob_len, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows,
                           const float* uniforms, const uint8_t* prev_rgba8, uint8_t* out_rgba8, int nthreads, int semantics,
                           const float* derived);

int game_oracle_render(const void* blob, size_t blob_len, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows,
                       const float* uniforms, const uint8_t* prev_rgba8, uint8_t* out_rgba8, int nthreads)
{
    return game_oracle_render_ex(blob, blob_len, w, h, row0, rows, uniforms, prev_rgba8, out_rgba8, nthreads, 0);
}

/* semantics, bit 0: 0 = the WebGL2 wording of every builtin (glsl.rs), 1 = the WebGPU wording where it differs
 * (wgsl.rs). Bit 1 (value 2): expression-valued stage arguments are live (implied by a non-NULL `derived`). Bit 2
 * (value 4): also run the lens post chain. */
int game_oracle_render_ex(const void* blob, size_t blob_len, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows,
                          const float* uniforms, const uint8_t* prev_rgba8, uint8_t* out_rgba8, int nthreads, int semantics)
{
    return game_oracle_render_ex2(blob, blob_len, w, h, row0, rows, uniforms, prev_rgba8, out_rgba8, nthreads, semantics, NULL);
}

/* derived: NULL, or this frame's value of every expression-valued stage argument, in the order the IR numbers
 * them (the caller computes them; the stage arguments that carry an index read them instead of their default). */
int game_oracle_render_ex2(const void* blob, size_t blob_len, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows,
                           const float* uniforms, const uint8_t* prev_rgba8, uint8_t* out_rgba8, int nthreads, int semantics,
                           const float* derived)
{
    ir_view ir;
    if (ir_open(blob, blob_len, &ir) != 0 || !uniforms || !out_rgba8 || w == 0 || h == 0 || row0 + rows > h) return -1;
    frame_consts fc;
    fc.wgsl = (semantics & 1) != 0;
    fc.lens_post = (semantics & 4) != 0;
    fc.project = (semantics & 8) != 0;
    fc.animate = (semantics & 2) != 0 || derived != NULL;
    fc.row = uniforms;
    fc.derived = derived;
    fc.aspect = uniforms[BU_RES_X] / uniforms[BU_RES_Y];               /* glsl.rs:55 */
    fc.time = gm_fract(uniforms[BU_TIME] / 120.0f) * 120.0f;           /* glsl.rs:56 */
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > rows) nthreads = (int)(rows ? rows : 1);
    frame_job* jobs = (frame_job*)calloc((size_t)nthreads, sizeof(frame_job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    if (!jobs || !th) { free(jobs); free(th); return -1; }
    for (int t = 0; t < nthreads; t++) {
        jobs[t].ir = &ir; jobs[t].fc = fc; jobs[t].w = w; jobs[t].h = h;
        jobs[t].row_begin = row0 + (uint32_t)((uint64_t)rows * (uint32_t)t / (uint32_t)nthreads);
        jobs[t].row_end = row0 + (uint32_t)((uint64_t)rows * (uint32_t)(t + 1) / (uint32_t)nthreads);
        jobs[t].band_row0 = row0;
        jobs[t].prev = (const uint32_t*)prev_rgba8;
        jobs[t].out = (uint32_t*)out_rgba8;
    }
    if (nthreads == 1) render_rows(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, render_thread, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    free(jobs); free(th);
    return 0;
}

/* ── gravity (gravity.rs:43-120) ───────────────────────────────────────────────────────── */
typedef struct {
    const ir_view* ir;
    const float* in;   /* n x {pos.x, pos.y, vel.x, vel.y} */
    float* out;
    uint32_t n, begin, end;
    uint32_t out_base;   /* particle idx is written to out[4 * (idx - out_base)] */
    float dt, time;
} grav_job;

static void gravity_range(const grav_job* J)
{
    const ir_view* ir = J->ir;
    for (uint32_t idx = J->begin; idx < J->end; idx++) {
        float sx = J->in[4 * idx], sy = J->in[4 * idx + 1];
        float fx = 0.0f, fy = 0.0f;
        for (uint32_t j = 0; j < J->n; j++) {                                   /* gravity.rs:78-89 */
            if (j == idx) continue;
            float dx = J->in[4 * j] - sx, dy = J->in[4 * j + 1] - sy;
            float d2 = fmaf(dx, dx, dy * dy);
            float dist = gm_max(sqrtf(d2), 0.001f);
            /* dir = diff / dist as diff * inversesqrt(max(d2, c)), c = the smallest float whose root rounds to 0.001f
             * (so sqrt(max(d2, c)) is dist for every d2): the reciprocal root is rounded once */
            float inv = gm_rsqrt(gm_max(d2, 0x1.0c6f7ap-20f));
            float f = eval_law(ir->law, ir->h->n_law, dist, J->time);
            float fs = inv * f;                                                 /* force += dir * law as diff * (law / dist) */
            fx = fmaf(dx, fs, fx); fy = fmaf(dy, fs, fy);                       /* contraction point */
        }
        float vx = J->in[4 * idx + 2] * ir->h->damping + fx * J->dt;            /* gravity.rs:92-96 */
        float vy = J->in[4 * idx + 3] * ir->h->damping + fy * J->dt;
        float px = sx + vx * J->dt, py = sy + vy * J->dt;
        if (ir->h->bounds == 0) {                                               /* reflect :100-106 */
            if (px < -1.0f) { px = -1.0f; vx = -vx; }
            if (px > 1.0f) { px = 1.0f; vx = -vx; }
            if (py < -1.0f) { py = -1.0f; vy = -vy; }
            if (py > 1.0f) { py = 1.0f; vy = -vy; }
        } else if (ir->h->bounds == 1) {                                        /* wrap :107-110 */
            px = gm_fract((px + 1.0f) * 0.5f) * 2.0f - 1.0f;
            py = gm_fract((py + 1.0f) * 0.5f) * 2.0f - 1.0f;
        }
        J->out[4 * (idx - J->out_base)] = px; J->out[4 * (idx - J->out_base) + 1] = py; J->out[4 * (idx - J->out_base) + 2] = vx; J->out[4 * (idx - J->out_base) + 3] = vy;
    }
}
/* The same loop for a force law that is one constant (the default rule 1.0), eight particles side by side: each of
 * them walks j in the same order through the same correctly rounded operations (inversesqrt, fmaf), so every
 * particle gets the bits gravity_range gives it; written with AVX2 / FMA vector operations, each of which is the
 * IEEE operation of its scalar counterpart in every lane.
 * max(d2, c) is written as a select: a NaN d2 yields c as gm_max does, and d2 is never negative. */
#define GRAV_BLOCK 8
/* gm_rsqrt on eight floats. The binary64 quotient rounded to binary32 is the correctly rounded result unless it lies
 * within a few binary64 ulps of the midpoint of two floats; a block with such a lane, with a lane outside
 * [2^-100, inf), or with a lane of the excepted class 4^k (1 - 2^-23) goes through the scalar function lane by lane. */
static inline __m256 gm_rsqrt8(__m256 x)
{
    const __m256d one = _mm256_set1_pd(1.0);
    const __m256d y0 = _mm256_div_pd(one, _mm256_sqrt_pd(_mm256_cvtps_pd(_mm256_castps256_ps128(x))));
    const __m256d y1 = _mm256_div_pd(one, _mm256_sqrt_pd(_mm256_cvtps_pd(_mm256_extractf128_ps(x, 1))));
    const __m256i mask = _mm256_set1_epi64x(0x1fffffff), lo = _mm256_set1_epi64x(0x0ffffff7), hi = _mm256_set1_epi64x(0x10000009);
    const __m256i l0 = _mm256_and_si256(_mm256_castpd_si256(y0), mask), l1 = _mm256_and_si256(_mm256_castpd_si256(y1), mask);
    const __m256i near0 = _mm256_and_si256(_mm256_cmpgt_epi64(l0, lo), _mm256_cmpgt_epi64(hi, l0));
    const __m256i near1 = _mm256_and_si256(_mm256_cmpgt_epi64(l1, lo), _mm256_cmpgt_epi64(hi, l1));
    const __m256 in_range = _mm256_and_ps(_mm256_cmp_ps(x, _mm256_set1_ps(0x1p-100f), _CMP_GE_OQ), _mm256_cmp_ps(x, _mm256_set1_ps(INFINITY), _CMP_LT_OQ));
    const __m256i xb = _mm256_castps_si256(x);
    const __m256i excepted = _mm256_cmpeq_epi32(_mm256_and_si256(xb, _mm256_set1_epi32(0x00ffffff)), _mm256_set1_epi32(0x007ffffe));
    if (_mm256_movemask_ps(in_range) != 0xff || !_mm256_testz_si256(_mm256_or_si256(near0, near1), _mm256_set1_epi64x(-1)) ||
        !_mm256_testz_si256(excepted, excepted)) {
        float v[8];
        _mm256_storeu_ps(v, x);
        for (int k = 0; k < 8; k++) v[k] = gm_rsqrt(v[k]);
        return _mm256_loadu_ps(v);
    }
    return _mm256_set_m128(_mm256_cvtpd_ps(y1), _mm256_cvtpd_ps(y0));
}
static void gravity_range_const(const grav_job* J, float f)
{
    const ir_view* ir = J->ir;
    const __m256 vf = _mm256_set1_ps(f), vc = _mm256_set1_ps(0x1.0c6f7ap-20f);
    for (uint32_t i0 = J->begin; i0 < J->end; i0 += GRAV_BLOCK) {
        uint32_t id[GRAV_BLOCK];
        float sx[GRAV_BLOCK], sy[GRAV_BLOCK], fx[GRAV_BLOCK], fy[GRAV_BLOCK];
        const uint32_t m = J->end - i0 < GRAV_BLOCK ? J->end - i0 : GRAV_BLOCK;
        for (uint32_t k = 0; k < GRAV_BLOCK; k++) {
            id[k] = k < m ? i0 + k : i0;          /* lanes past the range repeat particle i0; their sums are dropped */
            sx[k] = J->in[4 * id[k]]; sy[k] = J->in[4 * id[k] + 1];
        }
        const __m256 vsx = _mm256_loadu_ps(sx), vsy = _mm256_loadu_ps(sy);
        const __m256i vid = _mm256_loadu_si256((const __m256i*)id);
        __m256 vfx = _mm256_setzero_ps(), vfy = _mm256_setzero_ps();
        for (uint32_t j = 0; j < J->n; j++) {
            const __m256 dx = _mm256_sub_ps(_mm256_set1_ps(J->in[4 * j]), vsx), dy = _mm256_sub_ps(_mm256_set1_ps(J->in[4 * j + 1]), vsy);
            const __m256 d2 = _mm256_fmadd_ps(dx, dx, _mm256_mul_ps(dy, dy));                      /* fmaf(dx, dx, dy * dy) */
            const __m256 d2c = _mm256_blendv_ps(vc, d2, _mm256_cmp_ps(d2, vc, _CMP_GT_OQ));          /* d2 > c ? d2 : c */
            const __m256 inv = gm_rsqrt8(d2c);
            const __m256 fs = _mm256_mul_ps(inv, vf);
            const __m256 nx = _mm256_fmadd_ps(dx, fs, vfx), ny = _mm256_fmadd_ps(dy, fs, vfy);
            const __m256 self = _mm256_castsi256_ps(_mm256_cmpeq_epi32(vid, _mm256_set1_epi32((int)j)));
            vfx = _mm256_blendv_ps(nx, vfx, self);                                                /* j == idx: the pair is skipped */
            vfy = _mm256_blendv_ps(ny, vfy, self);
        }
        _mm256_storeu_ps(fx, vfx); _mm256_storeu_ps(fy, vfy);
        for (uint32_t k = 0; k < m; k++) {
            uint32_t idx = i0 + k;
            float vx = J->in[4 * idx + 2] * ir->h->damping + fx[k] * J->dt;
            float vy = J->in[4 * idx + 3] * ir->h->damping + fy[k] * J->dt;
            float px = sx[k] + vx * J->dt, py = sy[k] + vy * J->dt;
            if (ir->h->bounds == 0) {
                if (px < -1.0f) { px = -1.0f; vx = -vx; }
                if (px > 1.0f) { px = 1.0f; vx = -vx; }
                if (py < -1.0f) { py = -1.0f; vy = -vy; }
                if (py > 1.0f) { py = 1.0f; vy = -vy; }
            } else if (ir->h->bounds == 1) {
                px = gm_fract((px + 1.0f) * 0.5f) * 2.0f - 1.0f;
                py = gm_fract((py + 1.0f) * 0.5f) * 2.0f - 1.0f;
            }
            J->out[4 * (idx - J->out_base)] = px; J->out[4 * (idx - J->out_base) + 1] = py; J->out[4 * (idx - J->out_base) + 2] = vx; J->out[4 * (idx - J->out_base) + 3] = vy;
        }
    }
}
static int g_gravity_blocked = 1;
/* 0: always the one-particle-at-a-time loop (tests compare the two). */
void game_oracle_set_gravity_blocked(int on) { g_gravity_blocked = on; }
static void* gravity_thread(void* p)
{
    const grav_job* J = (const grav_job*)p;
    const ir_view* ir = J->ir;
    if (g_gravity_blocked && ir->h->n_law == 1 && ir->law[0].op == LAW_NUM) gravity_range_const(J, ir->law[0].val);
    else gravity_range(J);
    return NULL;
}

/* Advances `steps` steps of dt. particles: n x {pos.x, pos.y, vel.x, vel.y}, updated in place.
 * *sim_time is the dispatch clock (`_time += dt` before each step, gravity.rs:158). */
int game_oracle_gravity(const void* blob, size_t blob_len, uint32_t n, float* particles, float dt, uint32_t steps,
                        float* sim_time, int nthreads)
{
    ir_view ir;
    if (ir_open(blob, blob_len, &ir) != 0 || !ir.h->gravity_present || !particles || !sim_time) return -1;
    float* tmp = (float*)malloc((size_t)n * 16);
    if (!tmp) return -1;
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > n) nthreads = n ? (int)n : 1;
    grav_job* jobs = (grav_job*)calloc((size_t)nthreads, sizeof(grav_job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    float* a = particles;
    float* b = tmp;
    for (uint32_t s = 0; s < steps; s++) {
        *sim_time = *sim_time + dt;
        for (int t = 0; t < nthreads; t++) {
            jobs[t].ir = &ir; jobs[t].in = a; jobs[t].out = b; jobs[t].n = n; jobs[t].dt = dt; jobs[t].time = *sim_time;
            jobs[t].begin = (uint32_t)((uint64_t)n * (uint32_t)t / (uint32_t)nthreads);
            jobs[t].end = (uint32_t)((uint64_t)n * (uint32_t)(t + 1) / (uint32_t)nthreads);
        }
        if (nthreads == 1) gravity_thread(&jobs[0]);
        else {
            for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, gravity_thread, &jobs[t]);
            for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        }
        float* sw = a; a = b; b = sw;
    }
    if (a != particles) memcpy(particles, a, (size_t)n * 16);
    free(tmp); free(jobs); free(th);
    return 0;
}

/* One step for the particles [first, first + count) of an n-particle set, each against all n (what one rank of an
 * index-sharded run computes, and what lets a few particles of a very large set be checked in O(count * n)).
 * particles: n x {pos.x, pos.y, vel.x, vel.y} (read only); out: count x 4; time = the dispatch clock of the step. */
int game_oracle_gravity_range(const void* blob, size_t blob_len, uint32_t n, const float* particles, float dt, float time,
                              uint32_t first, uint32_t count, float* out, int nthreads)
{
    ir_view ir;
    if (ir_open(blob, blob_len, &ir) != 0 || !ir.h->gravity_present || !particles || !out) return -1;
    if (count == 0 || (uint64_t)first + count > n) return -1;
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > count) nthreads = (int)count;
    grav_job* jobs = (grav_job*)calloc((size_t)nthreads, sizeof(grav_job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    if (!jobs || !th) { free(jobs); free(th); return -1; }
    for (int t = 0; t < nthreads; t++) {
        jobs[t].ir = &ir; jobs[t].in = particles; jobs[t].out = out; jobs[t].n = n; jobs[t].dt = dt; jobs[t].time = time;
        jobs[t].out_base = first;
        jobs[t].begin = first + (uint32_t)((uint64_t)count * (uint32_t)t / (uint32_t)nthreads);
        jobs[t].end = first + (uint32_t)((uint64_t)count * (uint32_t)(t + 1) / (uint32_t)nthreads);
    }
    if (nthreads == 1) gravity_thread(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, gravity_thread, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    free(jobs); free(th);
    return 0;
}

/* SplitMix64 particle initialisation used by the synthetic particle workloads:
 * pos = (u >> 40) * 2^-23 - 1 per coordinate, vel = 0. */
void game_oracle_seed_particles(uint32_t n, uint64_t seed, float* particles)
{
    uint64_t s = seed;
    for (uint32_t i = 0; i < n; i++) {
        for (int k = 0; k < 2; k++) {
            s += 0x9E3779B97F4A7C15ull;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z = z ^ (z >> 31);
            particles[4 * i + k] = (float)(z >> 40) * 0x1p-23f - 1.0f;
        }
        particles[4 * i + 2] = 0.0f;
        particles[4 * i + 3] = 0.0f;
    }
}

/* Exposed for unit tests of the math layer. */
float game_oracle_sin(float x) { return gm_sin(x); }
float game_oracle_cos(float x) { return gm_cos(x); }
float game_oracle_atan2(float y, float x) { return gm_atan2(y, x); }
float game_oracle_fract(float x) { return gm_fract(x); }
float game_oracle_hash2(float x, float y) { return hash2(x, y); }
float game_oracle_noise2(float x, float y) { return noise2(x, y); }
float game_oracle_fbm2(float x, float y, int oct, float pers, float lac) { return fbm2(x, y, oct, pers, lac); }
float game_oracle_simplex2d(float x, float y) { return simplex2d(x, y); }
float game_oracle_voronoi2d(float x, float y, float s) { return voronoi2d(x, y, s); }
float game_oracle_pow(float x, float y) { return gm_pow(x, y); }
float game_oracle_exp(float x) { return gm_exp(x); }
float game_oracle_log(float x) { return gm_log(x); }
float game_oracle_rsqrt(float x) { return gm_rsqrt(x); }
void game_oracle_rsqrt_array(uint32_t n, const float* x, float* out) { for (uint32_t i = 0; i < n; i++) out[i] = gm_rsqrt(x[i]); }

/* Self-check of gm_rsqrt_cr against integer arithmetic, over every float of [1, 4) (both exponent parities of
 * every significand; 1 / sqrt(4^k x) = 2^-k / sqrt(x) exactly, so these stand for all normal inputs). For
 * x = X / 2^23 (X in [2^23, 2^25)) the result in units of 2^-24 is sqrt(2^71 / X): F = the largest integer with
 * F^2 X <= 2^71, rounded up when (2F + 1)^2 X < 2^73 - except for the one input the definition excepts,
 * 4 (1 - 2^-23). Returns the number of inputs where gm_rsqrt_cr differs. */
uint32_t game_oracle_rsqrt_selfcheck(uint32_t* first_bad)
{
    typedef unsigned __int128 u128;
    uint32_t bad = 0;
    for (uint32_t X = 1u << 23; X < (1u << 25); X++) {
        float x = gm_from_bits(X < (1u << 24) ? 0x3f800000u + (X - (1u << 23)) : 0x40000000u + ((X - (1u << 24)) >> 1));
        if (X >= (1u << 24) && (X & 1u)) continue;                      /* [2, 4): only even X are floats */
        uint64_t F = (uint64_t)sqrt(0x1p71 / (double)X);
        while ((u128)F * F * X > ((u128)1 << 71)) F--;
        while ((u128)(F + 1) * (F + 1) * X <= ((u128)1 << 71)) F++;
        uint64_t M = 2 * F + 1;
        uint64_t Y = ((u128)M * M * X < ((u128)1 << 73)) ? F + 1 : F;   /* nearest integer to sqrt(2^71 / X) */
        if (X == (1u << 25) - 4u) Y = 1u << 23;                         /* the stated exception: 4 (1 - 2^-23) -> 1 / 2 */
        float want = (float)Y * 0x1p-24f;                                /* Y <= 2^24: exact */
        if (gm_bits(gm_rsqrt_cr(x)) != gm_bits(want)) { if (!bad && first_bad) *first_bad = gm_bits(x); bad++; }
    }
    return bad;
}
