"""Small cinematics, written for these tests, that together exercise every builtin, every
argument kind (literal, user uniform, built-in uniform, defaulted expression) and the layer
compositing rules. Used by the oracle regression goldens and by the GPU parity tests."""


def _c(body):
    return 'cinematic "g" { %s }' % body


GALLERY = {
    # SDF generators
    "circle_ring": _c("layer a { ring(0.35, 0.03) | glow(1.2) | tint(0.2, 0.9, 0.7) } layer b { translate(0.3, -0.2) | circle(0.12) | shade(0.9, 0.4, 0.1) }"),
    "star": _c("layer a { rotate(0.3) | star(5.0, 0.4, 0.18) | glow(1.5) | tint(1.0, 0.8, 0.2) }"),
    "star7_shade": _c("layer a { star(7.0, 0.5, 0.3) | onion(0.02) | shade(0.3, 0.6, 1.0) }"),
    "box_round": _c("layer a { box(0.4, 0.2) | round(0.05) | glow(2.0) | tint(0.9, 0.3, 0.5) }"),
    "polygon": _c("layer a { scale(1.3) | polygon(6.0, 0.35) | glow(1.8) } layer b { polygon(3.0, 0.2) | threshold(0.0) | shade(0.2, 0.2, 0.9) }"),
    "simplex": _c("layer a { simplex(5.0) | glow(1.1) | tint(0.8, 0.5, 0.9) }"),
    "voronoi": _c("layer a { voronoi(7.0) | glow(0.9) | tint(0.4, 0.9, 0.6) }"),
    "waves": _c("layer a { concentric_waves(0.8, 0.7, 4.0) | glow(1.3) | tint(0.3, 0.6, 1.0) }"),
    "fbm_defaults": _c("layer a { fbm() | glow() | tint() }"),
    "fbm_uniform_octaves": _c("layer cfg { oct: 3.0 sc: 2.5 } layer a { fbm(sc, oct, 0.6, 2.2) | glow(1.4) }"),
    # transforms
    "twist_mirror": _c("layer a { mirror(0.0) | twist(2.5) | box(0.3, 0.1) | glow(1.5) } layer b { mirror(1.0) | translate(0.0, 0.4) | circle(0.1) | glow(1.0) }"),
    "repeat": _c("layer a { repeat(6.0) | translate(0.45, 0.0) | circle(0.08) | glow(2.0) | tint(1.0, 0.5, 0.8) }"),
    "domain_warp": _c("layer a { domain_warp(0.15, 4.0) | ring(0.4, 0.05) | glow(1.5) }"),
    "curl_noise": _c("layer a { curl_noise(3.0, 0.08) | circle(0.3) | glow(1.7) | tint(0.6, 1.0, 0.4) }"),
    "displace": _c("layer a { displace(0.12) | box(0.3, 0.3) | glow(1.4) | tint(1.0, 0.7, 0.3) }"),
    "rotate_time": _c("layer a { rotate(time) | box(0.4, 0.1) | glow(1.5) }"),
    # sdf modifiers
    "mask_arc": _c("layer a { ring(0.4, 0.04) | mask_arc(4.0) | glow(1.6) | tint(0.2, 0.8, 1.0) }"),
    "mask_arc_audio": _c("layer a { ring(0.3, 0.05) | mask_arc(audio.beat) | glow(1.6) }"),
    "threshold_onion": _c("layer a { fbm(4.0, 3, 0.5, 2.0) | threshold(0.1) | glow(0.8) } layer b { circle(0.3) | onion(0.04) | onion(0.01) | glow(1.2) | tint(1.0, 0.2, 0.2) }"),
    # bridges
    "emissive_single": _c("layer a { circle(0.3) | emissive(0.4) }"),
    "emissive_multi": _c("layer a { circle(0.3) | emissive(0.4) } layer b { ring(0.5, 0.02) | glow(1.0) }"),
    "palette_fire": _c("layer a { fbm(3.0, 4, 0.5, 2.0) | palette(fire) }"),
    "palette_sunset": _c("layer a { voronoi(4.0) | palette(sunset) | tint(0.9, 0.9, 1.0) }"),
    "palette_default": _c("layer a { simplex(2.0) | palette(whatever) }"),
    # colour processors
    "bloom_vignette": _c("layer a { circle(0.25) | glow(2.5) | tint(0.9, 0.6, 0.3) | bloom(0.5, 1.5) | vignette(0.4, 0.6) }"),
    "grain": _c("layer a { gradient() | grain(0.2) }"),
    "chromatic_tonemap": _c("layer a { ring(0.4, 0.08) | glow(3.0) | tint(1.0, 0.8, 0.6) | chromatic(0.3) | tonemap(1.5) }"),
    "scanlines_saturate": _c("layer a { fbm(2.0, 4, 0.5, 2.0) | palette(neon) | scanlines(120.0, 0.5) | saturate_color(0.3) }"),
    "glitch": _c("layer a { box(0.5, 0.2) | glow(1.5) | tint(0.5, 1.0, 0.5) | glitch(0.8) }"),
    "blend": _c("layer a { circle(0.4) | glow(2.0) | blend(0.35) } layer b { gradient() | blend(0.5) }"),
    "spectrum": _c("layer a { fn: spectrum(b, m, t) | bloom(0.3, 1.0) b: 0.1 m: 0.2 t: 0.3 }"),
    "spectrum_post": _c("layer a { spectrum(0.2, 0.1, 0.0) | chromatic(0.6) | vignette(0.35) | grain(0.025) | tonemap(1.0) }"),
    # state resets and stage order
    "reset_chain": _c("layer a { fbm(2.0) | displace(0.1) | circle(0.2) | glow(1.5) | grain(0.05) }"),
    "regenerate_after_color": _c("layer a { circle(0.3) | glow(1.5) | rotate(1.0) | scale(2.0) | box(0.3, 0.1) | shade(0.2, 0.9, 0.4) | grain(0.3) }"),
    "memory_multi": _c("layer a memory: 0.9 { circle(0.2) | glow(2.0) | tint(0.5, 0.7, 1.0) } layer b { ring(0.4, 0.01) | glow(0.8) }"),
    "no_color_layer": _c("layer a cast sdf { circle(0.3) } layer b { ring(0.4, 0.02) | glow(1.0) }"),
    "uniform_args": _c("layer cfg { r: 0.3 g: 0.8 } layer a { circle(r) | glow(2.0) | tint(g, g, mouse.x) }"),
    "define_expand": _c("define halo(r, i) { ring(r, 0.02) | glow(i) } layer a { halo(0.3, 2.0) | tint(0.9, 0.9, 0.2) }"),
}
