/* fuzz_main.c — TEST INFRASTRUCTURE. Mutated-blob fuzz of the oracle's IR reader, meant for the sanitizer build
 * (make -C oracle asan): every blob file named on the command line is rendered as it is, then `count` damaged copies
 * of it (flipped bytes, overwritten words, changed length fields, truncations) go through the same entry points. A
 * damaged blob may be refused (-1) or rendered; it must never be read out of bounds.
 *
 *     game_oracle_fuzz <seed> <count> blob...
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int game_oracle_render_ex2(const void* blob, size_t blob_len, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows,
                           const float* uniforms, const uint8_t* prev_rgba8, uint8_t* out_rgba8, int nthreads, int semantics,
                           const float* derived);
int game_oracle_gravity(const void* blob, size_t blob_len, uint32_t n, float* particles, float dt, uint32_t steps,
                        float* sim_time, int nthreads);
void game_oracle_seed_particles(uint32_t n, uint64_t seed, float* particles);

static uint64_t rng_state;
static uint32_t rnd(void)
{
    rng_state += 0x9E3779B97F4A7C15ull;
    uint64_t z = rng_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 16);
}

enum { W = 9, H = 5, ROW_FLOATS = 10 + 4096, N_DERIVED = 65536 };

static int run(const uint8_t* blob, size_t len, const float* row, const float* derived)
{
    static uint8_t out[W * H * 4], prev[W * H * 4];
    int accepted = 0;
    for (int flags = 0; flags < 16; flags += 5)   /* 0, wgsl | lens post, animate | project, all */
        accepted += game_oracle_render_ex2(blob, len, W, H, 0, H, row, flags & 1 ? prev : NULL, out, 1, flags, derived) == 0;
    float p[16 * 4], t = 0.0f;
    game_oracle_seed_particles(16, 3, p);
    accepted += game_oracle_gravity(blob, len, 16, p, 1.0f / 60.0f, 2, &t, 1) == 0;
    return accepted;
}

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s seed count blob...\n", argv[0]); return 2; }
    rng_state = strtoull(argv[1], NULL, 0);
    long count = strtol(argv[2], NULL, 0);
    float* row = (float*)calloc(ROW_FLOATS, sizeof(float));
    float* derived = (float*)calloc(N_DERIVED, sizeof(float));
    if (!row || !derived) return 2;
    row[0] = 0.5f; row[1] = 0.3f; row[6] = (float)W; row[7] = (float)H;
    long total = 0, accepted = 0;
    for (int a = 3; a < argc; a++) {
        FILE* f = fopen(argv[a], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
        fseek(f, 0, SEEK_END);
        long len = ftell(f);
        fseek(f, 0, SEEK_SET);
        uint8_t* blob = (uint8_t*)malloc((size_t)len + 64);
        if (!blob || fread(blob, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        fclose(f);
        if (run(blob, (size_t)len, row, derived) == 0) { fprintf(stderr, "%s: the undamaged blob was refused\n", argv[a]); return 1; }
        for (long k = 0; k < count; k++) {
            /* every damaged copy lives in an allocation of exactly its length, so that the sanitizer sees any read past it */
            size_t mlen = (size_t)len;
            uint32_t kind = rnd() % 8u;
            if (kind == 0 && len > 8) mlen = rnd() % (uint32_t)len;                 /* truncation */
            uint8_t* m = (uint8_t*)malloc(mlen ? mlen : 1);
            memcpy(m, blob, mlen);
            if (mlen >= 8 && kind != 0) {
                uint32_t edits = 1u + rnd() % 4u;
                for (uint32_t e = 0; e < edits; e++) {
                    size_t at = rnd() % mlen;
                    switch (rnd() % 4u) {
                    case 0: m[at] ^= (uint8_t)(1u << (rnd() % 8u)); break;
                    case 1: m[at] = (uint8_t)rnd(); break;
                    case 2: {                                                       /* a whole word: small, huge or random */
                        size_t w4 = (at & ~(size_t)3);
                        if (w4 + 4 <= mlen) { uint32_t v = rnd() % 3u == 0 ? 0xffffffffu : rnd() % 2u ? rnd() % 300u : rnd(); memcpy(m + w4, &v, 4); }
                        break;
                    }
                    default: {                                                      /* one of the header's count fields */
                        size_t w4 = 8 + 4 * (rnd() % 4u);
                        if (w4 + 4 <= mlen) { uint32_t v = rnd() % 2u ? rnd() % 64u : rnd(); memcpy(m + w4, &v, 4); }
                        break;
                    }
                    }
                }
                if (kind == 1 && mlen >= 8) { uint32_t v = (uint32_t)mlen; memcpy(m + 4, &v, 4); }   /* keep `total` honest */
            }
            accepted += run(m, mlen, row, derived) > 0;
            total++;
            free(m);
        }
        free(blob);
    }
    printf("%ld damaged blobs, %ld still accepted, no fault\n", total, accepted);
    free(row); free(derived);
    return 0;
}
