"""Independent numpy (float32, one rounding per operation) restatement of the emitted GLSL of the
benchmark cinematics (the sine of the hash stages as the render path defines it), written from
game-compiler/src/codegen/glsl.rs without looking at the C oracle's structure. Used to cross-check oracle/game_oracle.c bit for bit; numpy
float32 arithmetic is IEEE single precision with no fused operations; the places where the render
path fixes a fused multiply-add (GLSL leaves contraction to the implementation) go through fma()
below, which rounds once by way of float64 and round-to-odd.
"""
import numpy as np

F = np.float32

# literal_form(): evaluate the reference's wording literally instead - no fused multiply-add anywhere, every quotient a
# true IEEE division, the framebuffer write round(clamp(c, 0, 1) * 255) half up, the texel fetch byte / 255. The render
# path deviates from that form where GLSL leaves the choice open (contraction points; reciprocal-multiply quotients,
# within 1.5 ulp of the division GLSL bounds by 2.5 ulp); tests use this mode to bound what the deviations cost.
LITERAL = False


class literal_form:
    def __enter__(self):
        global LITERAL
        self.saved, LITERAL = LITERAL, True

    def __exit__(self, *exc):
        global LITERAL
        LITERAL = self.saved


def quotient(x, y):
    """x / y as the render path defines quotients by a stage argument, a constant or a length: x * (1 / y)."""
    x, y = f(x), f(y)
    return x / y if LITERAL else x * (F(1.0) / y)


def f(x):
    return np.asarray(x, dtype=np.float32)


def rsqrt(x):
    """inversesqrt as the render path defines it: the exact 1 / sqrt(x) rounded ONCE to float32 for
    2^-100 <= x < inf (one excepted class of inputs, below), 1 / sqrt(x) as two operations elsewhere. The float64 quotient rounded to float32 is that value
    unless it sits within a few float64 ulps of the midpoint of two floats; those elements are decided in integers."""
    x = f(x)
    # the one stated exception: the float two below a power of four, 4^k (1 - 2^-23), gives 2^-k
    xb = x.view(np.uint32)
    excepted = (xb & np.uint32(0x00FFFFFF)) == np.uint32(0x007FFFFE)
    exc_value = np.ldexp(F(1.0), -((xb >> np.uint32(23)).astype(np.int32) - 126) // 2).astype(np.float32)   # E = 2 k + 126 -> 2^-k
    with np.errstate(divide="ignore", invalid="ignore"):
        plain = (F(1.0) / np.sqrt(x)).astype(np.float32)
        yd = 1.0 / np.sqrt(x.astype(np.float64))
    out = yd.astype(np.float32)
    in_range = (x >= F(2.0 ** -100)) & np.isfinite(x)
    low = yd.view(np.int64) & 0x1FFFFFFF
    for i in np.flatnonzero((in_range & (np.abs(low - 0x10000000) <= 8)).reshape(-1)):
        from fractions import Fraction
        xv = Fraction(float(x.reshape(-1)[i]))
        a = np.float32(yd.reshape(-1)[i])
        cands = [np.nextafter(a, F(0)), a, np.nextafter(a, F(np.inf))]
        # nearest candidate to 1 / sqrt(x): compare squared distances without taking a root: |c - y| smallest, y^2 = 1 / x
        lo, mid, hi = (Fraction(float(c)) for c in cands)
        m1, m2 = (lo + mid) / 2, (mid + hi) / 2
        pick = cands[0] if m1 * m1 * xv > 1 else (cands[1] if m2 * m2 * xv > 1 else cands[2])
        out.reshape(-1)[i] = pick
    return np.where(in_range, np.where(excepted, exc_value, out), plain).astype(np.float32)


def fma(a, b, c):
    """Correctly rounded float32 a*b+c: the product is exact in float64, the sum is rounded to odd
    (53 >= 2*24+2 bits), and the final float64 -> float32 conversion rounds once."""
    if LITERAL:
        return (f(a) * f(b) + f(c)).astype(np.float32)
    a, b, c = (np.asarray(v, dtype=np.float32).astype(np.float64) for v in (a, b, c))
    a, b, c = np.broadcast_arrays(a, b, c)
    p = a * b
    s = p + c
    bb = s - p
    e = (p - (s - bb)) + (c - bb)
    bits = s.copy().view(np.int64)
    need = np.isfinite(s) & np.isfinite(e) & (e != 0) & ((bits & 1) == 0)
    same = (e > 0) == (s > 0)
    bits = np.where(need, np.where(same, bits + 1, bits - 1), bits)
    return bits.view(np.float64).astype(np.float32)


def fract(x):
    r = x - np.floor(x)
    return np.where(r < F(1.0), r, F(np.nextafter(np.float32(1.0), np.float32(0.0)))).astype(np.float32)


def mix(a, b, t):
    """a + t (b - a) with one fused multiply-add; literally, GLSL's x (1 - t) + y t."""
    if LITERAL:
        return (f(a) * (F(1.0) - f(t)) + f(b) * f(t)).astype(np.float32)
    return fma(t, f(b) - f(a), a)


def clamp(x, lo, hi):
    return np.minimum(np.maximum(x, F(lo)), F(hi))


def smoothstep(e0, e1, x):
    t = clamp(quotient(x - F(e0), F(e1) - F(e0)), 0.0, 1.0)   # one reciprocal, one multiply
    return t * t * fma(F(-2.0), t, F(3.0))


def hash2(px, py):  # glsl.rs:134-138
    x = fract(px * F(0.1031))
    y = fract(py * F(0.1031))
    z = fract(px * F(0.1031))
    d = x * (y + F(33.33)) + y * (z + F(33.33)) + z * (x + F(33.33))
    x, y, z = x + d, y + d, z + d
    return fract((x + y) * z)


def noise2(px, py):  # glsl.rs:141-150
    ix, iy = np.floor(px), np.floor(py)
    fx, fy = fract(px), fract(py)
    ux = fx * fx * fma(F(-2.0), fx, F(3.0))
    uy = fy * fy * fma(F(-2.0), fy, F(3.0))
    a = hash2(ix, iy)
    b = hash2(ix + F(1.0), iy + F(0.0))
    c = hash2(ix + F(0.0), iy + F(1.0))
    d = hash2(ix + F(1.0), iy + F(1.0))
    return fma(mix(mix(a, b, ux), mix(c, d, ux), uy), F(2.0), F(-1.0))


def fbm2(px, py, octaves, persistence, lacunarity):  # glsl.rs:155-169
    value = np.zeros_like(px)
    amplitude, frequency, max_val = F(1.0), F(1.0), F(0.0)
    qx, qy = px, py
    for _ in range(octaves):
        value = fma(noise2(qx * frequency, qy * frequency), amplitude, value)
        max_val = F(max_val + amplitude)
        amplitude = F(amplitude * F(persistence))
        frequency = F(frequency * F(lacunarity))
        qx, qy = fma(qx, F(0.8), -(qy * F(0.6))), fma(qx, F(0.6), qy * F(0.8))
    return quotient(value, max_val)


def sin32(x):
    """sin as the render path defines it for |x| <= 32768 (GLSL leaves its precision open): k = rint(x / pi),
    r = x - k pi by a three-constant Cody-Waite reduction with fma, sin = (-1)^k (r + r^3 P(r^2)), P of degree 3."""
    x = f(x)
    assert np.all(np.abs(x) <= F(32768.0))
    k = np.rint(x * F(0.318309886))
    r = fma(k, F(-3.140625), x)
    r = fma(k, F(-9.67502593994140625e-4), r)
    r = fma(k, F(-1.509957990978376432e-7), r)
    z = r * r
    ps = fma(fma(fma(F(float.fromhex("0x1.5c33e8p-19")), z, F(float.fromhex("-0x1.9f5196p-13"))), z, F(float.fromhex("0x1.110e1ap-7"))), z,
             F(float.fromhex("-0x1.555546p-3")))
    sn = fma(ps * z, r, r)
    odd = (k.astype(np.int64) & 1).astype(np.uint32) << np.uint32(31)
    return (sn.view(np.uint32) ^ odd).view(np.float32)


def mod(x, y):
    """GLSL mod(x, y) = x - y floor(x / y), the quotient by a constant as one reciprocal and one multiply."""
    return (f(x) - F(y) * np.floor(quotient(x, F(y)))).astype(np.float32)


def voronoi2d(px, py, scale):  # glsl.rs:201-220
    psx, psy = px * F(scale), py * F(scale)
    ipx, ipy = np.floor(psx), np.floor(psy)
    fpx, fpy = fract(psx), fract(psy)
    d = np.ones_like(px)
    for y in (-1.0, 0.0, 1.0):
        for x in (-1.0, 0.0, 1.0):
            cx, cy = ipx + F(x), ipy + F(y)
            hx = fract(sin32(cx * F(127.1) + cy * F(311.7)) * F(43758.5453))
            hy = fract(sin32(cx * F(269.5) + cy * F(183.3)) * F(43758.5453))
            ptx, pty = F(x) + hx - fpx, F(y) + hy - fpy
            d = np.minimum(d, ptx * ptx + pty * pty)
    return np.sqrt(d)


def simplex2d(vx, vy):  # glsl.rs:172-199
    Cx, Cy, Cz, Cw = F(0.211324865405187), F(0.366025403784439), F(-0.577350269189626), F(0.024390243902439)

    def permute(x):
        return mod((x * F(34.0) + F(1.0)) * x, 289.0)
    s = vx * Cy + vy * Cy
    ix, iy = np.floor(vx + s), np.floor(vy + s)
    t = ix * Cx + iy * Cx
    x0x, x0y = vx - ix + t, vy - iy + t
    i1x = np.where(x0x > x0y, F(1.0), F(0.0)).astype(np.float32)
    i1y = np.where(x0x > x0y, F(0.0), F(1.0)).astype(np.float32)
    x1x, x1y = x0x + Cx - i1x, x0y + Cx - i1y
    x2x, x2y = x0x + Cz, x0y + Cz
    ix, iy = mod(ix, 289.0), mod(iy, 289.0)
    pm = [permute(permute(iy + oy) + ix + ox) for oy, ox in ((F(0.0), F(0.0)), (i1y, i1x), (F(1.0), F(1.0)))]
    m = [np.maximum(F(0.5) - (ax * ax + ay * ay), F(0.0)) for ax, ay in ((x0x, x0y), (x1x, x1y), (x2x, x2y))]
    m = [mm * mm for mm in m]
    m = [mm * mm for mm in m]
    g = []
    for k, (ax, ay) in enumerate(((x0x, x0y), (x1x, x1y), (x2x, x2y))):
        xv = F(2.0) * fract(pm[k] * Cw) - F(1.0)
        hh = np.abs(xv) - F(0.5)
        a0 = xv - np.floor(xv + F(0.5))
        m[k] = m[k] * (F(1.79284291400159) - F(0.85373472095314) * (a0 * a0 + hh * hh))
        g.append(a0 * ax + hh * ay)
    return F(130.0) * (m[0] * g[0] + m[1] * g[1] + m[2] * g[2])


def apply_glow(d, intensity):  # glsl.rs:102-107
    i = F(intensity)
    core = smoothstep(0.005, -0.005, d)
    md = np.maximum(d, F(0.0))
    halo = quotient(i, fma(md * md * i * i, F(16.0), F(1.0)))   # one reciprocal, one multiply
    return core + halo


def pixel_grid(w, h):
    """uv per pixel, top row first: v_uv = ((i + .5)/W, (j + .5)/H) with j counted from the bottom."""
    inv_w, inv_h = F(1.0) / F(w), F(1.0) / F(h)
    x = (np.arange(w, dtype=np.float32) + F(0.5)) * inv_w
    y = (np.arange(h - 1, -1, -1, dtype=np.float32) + F(0.5)) * inv_h
    uvx = np.broadcast_to(x * F(2.0) - F(1.0), (h, w)).astype(np.float32)
    uvy = np.broadcast_to((y * F(2.0) - F(1.0))[:, None], (h, w)).astype(np.float32)
    return uvx, uvy


def pack(r, g, b, a):
    def u8(c):
        if LITERAL:
            return np.floor(clamp(np.where(np.isnan(c), F(0.0), c), 0.0, 1.0) * F(255.0) + F(0.5)).astype(np.uint32).astype(np.uint8)
        t = np.rint(c * F(255.0))                      # to the nearest integer, ties to even
        t = np.where(np.isnan(t), F(0.0), t)
        return np.minimum(np.maximum(t, F(0.0)), F(255.0)).astype(np.uint32).astype(np.uint8)
    return np.stack([u8(r), u8(g), u8(b), u8(np.broadcast_to(f(a), r.shape))], axis=-1)


def render_c1(w, h):
    """circle(0.3) | glow(2.0) | tint(0.831, 0.686, 0.216)"""
    uvx, uvy = pixel_grid(w, h)
    aspect = F(w) / F(h)
    px, py = uvx * aspect, uvy
    sdf = np.sqrt(fma(px, px, py * py)) - F(0.3)
    g = apply_glow(sdf, 2.0)
    return pack(g * F(0.831), g * F(0.686), g * F(0.216), 1.0)


def render_c2(w, h):
    """domain_warp(0.1, 3.0) | fbm(3.0, 6, 0.5, 2.0) | glow(1.5) | tint(0.686, 0.878, 0.953)"""
    uvx, uvy = pixel_grid(w, h)
    aspect = F(w) / F(h)
    px, py = uvx * aspect, uvy
    n0 = noise2(px * F(3.0), py * F(3.0))
    n1 = noise2(px * F(3.0) + F(5.2), py * F(3.0) + F(1.3))
    px, py = px + n0 * F(0.1), py + n1 * F(0.1)
    sdf = fbm2(px * F(3.0), py * F(3.0), 6, 0.5, 2.0)
    g = apply_glow(sdf, 1.5)
    return pack(g * F(0.686), g * F(0.878), g * F(0.953), 1.0)


def render_c3_no_grain(w, h, bass, mid, treble):
    """spectrum(0,0,0) | bloom(0.4, 2.0) | chromatic(0.6) | vignette(0.35) with audio uniforms."""
    uvx, uvy = pixel_grid(w, h)
    sx = uvx * F(0.5) + F(0.5)
    r = smoothstep(0.0, 0.33, sx) * (F(bass) + F(0.0))
    g = smoothstep(0.33, 0.66, sx) * (F(mid) + F(0.0))
    b = smoothstep(0.66, 1.0, sx) * (F(treble) + F(0.0))
    lum = r * F(0.299) + g * F(0.587) + b * F(0.114)
    add = np.maximum(lum - F(0.4), F(0.0)) * F(2.0)
    r, g, b = r + add, g + add, b + add
    d2 = fma(uvx, uvx, uvy * uvy)
    ln = np.sqrt(d2)
    dirx = (uvx / ln if LITERAL else uvx * rsqrt(d2)) * F(0.6)     # normalize(uv) = uv * inversesqrt(dot(uv, uv))
    r, b = r * (F(1.0) + dirx), b * (F(1.0) - dirx)
    fac = F(1.0) - smoothstep(0.8, F(0.8) + F(0.35), ln)
    return pack(r * fac, g * fac, b * fac, 1.0)


def render_c3(w, h, bass, mid, treble, time, tonemap=None):
    """render_c3_no_grain followed by grain(0.025) (glsl.rs:453-457): every channel plus
    (fract(sin(dot(p, (12.9898, 78.233)) + time) * 43758.5453) - 0.5) * amount; `time` is the wrapped shader time.
    tonemap: exposure of a trailing Reinhard stage (glsl.rs:473-478), c * e / (c * e + 1) with a true division."""
    uvx, uvy = pixel_grid(w, h)
    sx = uvx * F(0.5) + F(0.5)
    r = smoothstep(0.0, 0.33, sx) * (F(bass) + F(0.0))
    g = smoothstep(0.33, 0.66, sx) * (F(mid) + F(0.0))
    b = smoothstep(0.66, 1.0, sx) * (F(treble) + F(0.0))
    lum = r * F(0.299) + g * F(0.587) + b * F(0.114)
    add = np.maximum(lum - F(0.4), F(0.0)) * F(2.0)
    r, g, b = r + add, g + add, b + add
    d2 = fma(uvx, uvx, uvy * uvy)
    ln = np.sqrt(d2)
    dirx = (uvx * rsqrt(d2)) * F(0.6)
    r, b = r * (F(1.0) + dirx), b * (F(1.0) - dirx)
    fac = F(1.0) - smoothstep(0.8, F(0.8) + F(0.35), ln)
    r, g, b = r * fac, g * fac, b * fac
    px, py = uvx * (F(w) / F(h)), uvy
    n = (fract(sin32(px * F(12.9898) + py * F(78.233) + F(time)) * F(43758.5453)) - F(0.5)) * F(0.025)
    r, g, b = r + n, g + n, b + n
    if tonemap is not None:
        with np.errstate(invalid="ignore", divide="ignore"):
            r, g, b = ((c * F(tonemap)) / (c * F(tonemap) + F(1.0)) for c in (r, g, b))
    return pack(r, g, b, 1.0)


def render_c2b(w, h):
    """examples/tutorial-05-noise.game: three layers, screen-blended (glsl.rs:251-258), whose tint colours are user
    uniforms at their default 0 with the builtin's defaults 1, 1 for the other two channels:
    fbm(3.0, 4, 0.5, 2.0) | glow(1.5); simplex(4.0) | glow(1.2); voronoi(6.0) | glow(1.0); each | tint(0, 1, 1)."""
    uvx, uvy = pixel_grid(w, h)
    px, py = uvx * (F(w) / F(h)), uvy
    glows = [apply_glow(fbm2(px * F(3.0), py * F(3.0), 4, 0.5, 2.0), 1.5),
             apply_glow(simplex2d(px * F(4.0), py * F(4.0)), 1.2),
             apply_glow(voronoi2d(px, py, 6.0), 1.0)]
    fin = [np.zeros_like(px) for _ in range(3)]
    for gl in glows:
        lc = (gl * F(0.0), gl * F(1.0), gl * F(1.0))
        fin = [fc + c - fc * c for fc, c in zip(fin, lc)]
    return pack(fin[0], fin[1], fin[2], 1.0)


def voronoi2d_wgsl(px, py, scale):  # wgsl.rs:187-201
    """The WebGPU emitter's cell noise: one arithmetic hash per cell gives the feature point (h, fract(17 h))."""
    spx, spy = px * F(scale), py * F(scale)
    ipx, ipy = np.floor(spx), np.floor(spy)
    fpx, fpy = fract(spx), fract(spy)
    d = np.ones_like(px)
    for y in (-1.0, 0.0, 1.0):
        for x in (-1.0, 0.0, 1.0):
            v = hash2(ipx + F(x), ipy + F(y))
            ptx, pty = F(x) + v - fpx, F(y) + fract(v * F(17.0)) - fpy
            d = np.minimum(d, ptx * ptx + pty * pty)
    return np.sqrt(d)


def simplex2d_wgsl(px, py):  # wgsl.rs:206-222
    """The WebGPU emitter's simplex noise: gradients -1 + 2 fract(sin(dot(cell, k)) 43758.5453), fourth-power falloff, times 70."""
    K1, K2 = F(0.366025404), F(0.211324865)

    def grad(cx, cy):
        qx = cx * F(127.1) + cy * F(311.7)
        qy = cx * F(269.5) + cy * F(183.3)
        return (F(-1.0) + F(2.0) * fract(sin32(qx) * F(43758.5453)), F(-1.0) + F(2.0) * fract(sin32(qy) * F(43758.5453)))
    s = (px + py) * K1
    ix, iy = np.floor(px + s), np.floor(py + s)
    t = (ix + iy) * K2
    ax, ay = px - ix + t, py - iy + t
    ox = np.where(ax >= ay, F(1.0), F(0.0)).astype(np.float32)         # step(a.yx, a): 1 where the edge <= the value
    oy = np.where(ay >= ax, F(1.0), F(0.0)).astype(np.float32)
    bx, by = ax - ox + K2, ay - oy + K2
    cx, cy = ax - F(1.0) + F(2.0) * K2, ay - F(1.0) + F(2.0) * K2
    n = []
    for (vx, vy), (gx, gy) in (((ax, ay), grad(ix, iy)), ((bx, by), grad(ix + ox, iy + oy)), ((cx, cy), grad(ix + F(1.0), iy + F(1.0)))):
        hh = np.maximum(F(0.5) - (vx * vx + vy * vy), F(0.0))
        n.append(hh * hh * hh * hh * (vx * gx + vy * gy))
    return n[0] * F(70.0) + n[1] * F(70.0) + n[2] * F(70.0)


def render_c2b_wgsl(w, h):
    """render_c2b in the wording of the WebGPU emitter: same fbm and glow, the two noises above, same screen blend (wgsl.rs:259)."""
    uvx, uvy = pixel_grid(w, h)
    px, py = uvx * (F(w) / F(h)), uvy
    glows = [apply_glow(fbm2(px * F(3.0), py * F(3.0), 4, 0.5, 2.0), 1.5),
             apply_glow(simplex2d_wgsl(px * F(4.0), py * F(4.0)), 1.2),
             apply_glow(voronoi2d_wgsl(px, py, 6.0), 1.0)]
    fin = [np.zeros_like(px) for _ in range(3)]
    for gl in glows:
        lc = (gl * F(0.0), gl * F(1.0), gl * F(1.0))
        fin = [fc + c - fc * c for fc, c in zip(fin, lc)]
    return pack(fin[0], fin[1], fin[2], 1.0)


def memory_mix(cur_rgba_float, prev_rgba8, decay):
    """memory.rs:38-45 on float colours; prev is RGBA8."""
    prev = quotient(prev_rgba8.astype(np.float32), F(255.0))   # byte times RN(1 / 255)
    return mix(cur_rgba_float, prev, F(decay))


def gravity_step(p, dt, damping, law, bounds, time):
    """One step of gravity.rs:70-117 on an (n, 4) float32 array {pos.x, pos.y, vel.x, vel.y}: every
    particle sums, in index order, dir * law(dist) over the others with dist = max(length(diff), 0.001),
    force += diff * (law * inversesqrt(max(|diff|^2, c))), an fma per component; then damped velocity, position, bounds.
    `law(dist, time)` works on float32 arrays. Vectorised over the receiving particle, sequential in j."""
    p = np.asarray(p, dtype=np.float32)
    n = p.shape[0]
    sx, sy = p[:, 0], p[:, 1]
    fx, fy = np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32)
    idx = np.arange(n)
    for j in range(n):
        dx, dy = p[j, 0] - sx, p[j, 1] - sy
        d2 = fma(dx, dx, dy * dy)
        dist = np.maximum(np.sqrt(d2), F(0.001))
        f = np.broadcast_to(np.asarray(law(dist, F(time)), dtype=np.float32), dist.shape)
        if LITERAL:
            nfx, nfy = fma(dx / dist, f, fx), fma(dy / dist, f, fy)
        else:   # diff * (law * inversesqrt(max(d2, c))); c = the smallest float whose root rounds to 0.001
            inv = rsqrt(np.where(d2 > F(float.fromhex("0x1.0c6f7ap-20")), d2, F(float.fromhex("0x1.0c6f7ap-20"))))
            fs = inv * f
            nfx, nfy = fma(dx, fs, fx), fma(dy, fs, fy)
        keep = idx == j
        fx, fy = np.where(keep, fx, nfx), np.where(keep, fy, nfy)
    vx = p[:, 2] * F(damping) + fx * F(dt)
    vy = p[:, 3] * F(damping) + fy * F(dt)
    px, py = sx + vx * F(dt), sy + vy * F(dt)
    if bounds == "reflect":
        lo = px < F(-1.0); px = np.where(lo, F(-1.0), px); vx = np.where(lo, -vx, vx)
        hi = px > F(1.0); px = np.where(hi, F(1.0), px); vx = np.where(hi, -vx, vx)
        lo = py < F(-1.0); py = np.where(lo, F(-1.0), py); vy = np.where(lo, -vy, vy)
        hi = py > F(1.0); py = np.where(hi, F(1.0), py); vy = np.where(hi, -vy, vy)
    elif bounds == "wrap":
        px = fract((px + F(1.0)) * F(0.5)) * F(2.0) - F(1.0)
        py = fract((py + F(1.0)) * F(0.5)) * F(2.0) - F(1.0)
    return np.stack([px, py, vx, vy], axis=1).astype(np.float32)


def render_wgsl_post_chain(w, h):
    """spectrum(0.9, 0.7, 0.5) | bloom(0.4, 2.0) | chromatic(0.6) | vignette(0.35, 0.8) | tonemap(1.2) in the
    wording of the reference's WebGPU emitter (wgsl.rs:520-534, :381-386, :399-408, :392-398, :409-419)."""
    uvx, uvy = pixel_grid(w, h)
    aspect = F(w) / F(h)
    px, py = uvx * aspect, uvy
    d = np.sqrt(fma(px, px, py * py))
    sb = smoothstep(0.35, 0.15, d) * F(0.9)
    sm = smoothstep(0.55, 0.35, d) * smoothstep(0.15, 0.35, d) * F(0.7)
    st = smoothstep(0.75, 0.55, d) * smoothstep(0.35, 0.55, d) * F(0.5)
    r = sb * F(1.0) + sm * F(0.1) + st * F(0.2)
    g = sb * F(0.2) + sm * F(1.0) + st * F(0.3)
    b = sb * F(0.1) + sm * F(0.3) + st * F(1.0)
    lum = r * F(0.299) + g * F(0.587) + b * F(0.114)
    add = np.maximum(lum - F(0.4), F(0.0)) * F(2.0)
    r, g, b = r + add, g + add, b + add
    ln = np.sqrt(fma(uvx, uvx, uvy * uvy))
    chr_d = ln * F(0.6)
    r, b = r + chr_d * F(0.5), b - chr_d * F(0.5)
    vig = smoothstep(F(0.8) + F(0.3), F(0.8) - F(0.2), ln)
    f = mix(F(1.0), vig, F(0.35))
    r, g, b = r * f, g * f, b * f

    def aces(x):
        x = x * F(1.2)
        return clamp((x * (F(2.51) * x + F(0.03))) / (x * (F(2.43) * x + F(0.59)) + F(0.14)), 0.0, 1.0)
    return pack(aces(r), aces(g), aces(b), 1.0)


def blur_rgba8(frame, radius):
    """The frame filter of game_hip_set_blur on an (h, w, 4) uint8 frame: binomial weights C(2r, k), horizontal pass
    then vertical pass, each (sum of weight * byte + 4^r / 2) >> 2r per channel with coordinates clamped to the frame."""
    from math import comb
    wts = [comb(2 * radius, k) for k in range(2 * radius + 1)]

    def one_pass(a, axis):
        n = a.shape[axis]
        acc = np.zeros(a.shape, dtype=np.uint32)
        for k, wk in enumerate(wts):
            idx = np.clip(np.arange(n) + k - radius, 0, n - 1)
            acc += np.uint32(wk) * np.take(a, idx, axis=axis).astype(np.uint32)
        return ((acc + np.uint32(1 << (2 * radius - 1))) >> np.uint32(2 * radius)).astype(np.uint8)

    return one_pass(one_pass(frame, 1), 0)
