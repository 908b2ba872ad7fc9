/*
 * pcg64.h — ORACLE / TEST INFRASTRUCTURE (plain C restatement, host only).
 *
 * gymnasium seeds `env.np_random` as numpy.random.Generator(PCG64(SeedSequence(seed)))
 * (reference: WindGym/Wind_Farm_Env.py:689 `super().reset(seed=seed)`; draws at :564-568 via
 * WindGym/WindEnv.py:24-35).  To sample the *same* wind conditions as the reference for a given seed we
 * restate numpy's published algorithms:
 *   - SeedSequence entropy mixing + generate_state  (numpy/random/bit_generator.pyx, numpy 1.17+)
 *   - PCG64 (pcg_setseq_128, XSL-RR 128/64 output)   (numpy/random/src/pcg64/pcg64.h)
 *   - Generator.uniform = low + (high-low) * next_double, next_double = (u64 >> 11) * 2^-53
 *   - Generator.integers(0, n) for n-1 < 2^32: Lemire rejection on buffered 32-bit halves
 * Pinned in tests/test_rng.py against numpy itself (numpy is importable everywhere the tests run).
 */
#ifndef WGO_PCG64_H
#define WGO_PCG64_H
#include <stdint.h>

typedef unsigned __int128 wgo_u128;

typedef struct wgo_pcg64 {
    wgo_u128 state, inc;
    int has_uint32;
    uint32_t uinteger;
} wgo_pcg64;

#define WGO_PCG_MULT ((((wgo_u128)2549297995355413924ULL) << 64) | (wgo_u128)4865540595714422341ULL)

static inline uint32_t wgo_ss_hashmix(uint32_t value, uint32_t* hash_const) {
    value ^= *hash_const;
    *hash_const *= 0x931e8875u;
    value *= *hash_const;
    value ^= value >> 16;
    return value;
}
static inline uint32_t wgo_ss_mix(uint32_t x, uint32_t y) {
    uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
    r ^= r >> 16;
    return r;
}
/* SeedSequence(seed).generate_state(4, uint64) for a non-negative integer seed < 2^64 */
static inline void wgo_seedseq_state4(uint64_t seed, uint64_t out[4]) {
    uint32_t ent[2];
    int n_ent = 1;
    ent[0] = (uint32_t)(seed & 0xffffffffu);
    ent[1] = (uint32_t)(seed >> 32);
    if (ent[1] != 0) n_ent = 2;
    uint32_t pool[4];
    uint32_t hc = 0x43b0d7e5u;
    for (int i = 0; i < 4; ++i) pool[i] = wgo_ss_hashmix(i < n_ent ? ent[i] : 0u, &hc);
    for (int s = 0; s < 4; ++s)
        for (int d = 0; d < 4; ++d)
            if (s != d) pool[d] = wgo_ss_mix(pool[d], wgo_ss_hashmix(pool[s], &hc));
    uint32_t hb = 0x8b51f9ddu;
    uint32_t w[8];
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pool[i & 3];
        v ^= hb;
        hb *= 0x58f38dedu;
        v *= hb;
        v ^= v >> 16;
        w[i] = v;
    }
    for (int i = 0; i < 4; ++i) out[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}
static inline void wgo_pcg64_seed(wgo_pcg64* g, uint64_t seed) {
    uint64_t s[4];
    wgo_seedseq_state4(seed, s);
    wgo_u128 initstate = ((wgo_u128)s[0] << 64) | s[1];
    wgo_u128 initseq = ((wgo_u128)s[2] << 64) | s[3];
    g->state = 0;
    g->inc = (initseq << 1) | 1;
    g->state = g->state * WGO_PCG_MULT + g->inc;
    g->state += initstate;
    g->state = g->state * WGO_PCG_MULT + g->inc;
    g->has_uint32 = 0;
    g->uinteger = 0;
}
static inline uint64_t wgo_pcg64_next64(wgo_pcg64* g) {
    g->state = g->state * WGO_PCG_MULT + g->inc;
    uint64_t hi = (uint64_t)(g->state >> 64), lo = (uint64_t)g->state;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(g->state >> 122);
    return (x >> rot) | (x << ((-rot) & 63));
}
static inline uint32_t wgo_pcg64_next32(wgo_pcg64* g) {
    if (g->has_uint32) {
        g->has_uint32 = 0;
        return g->uinteger;
    }
    uint64_t n = wgo_pcg64_next64(g);
    g->has_uint32 = 1;
    g->uinteger = (uint32_t)(n >> 32);
    return (uint32_t)(n & 0xffffffffu);
}
static inline double wgo_pcg64_double(wgo_pcg64* g) {
    return (double)(wgo_pcg64_next64(g) >> 11) * (1.0 / 9007199254740992.0);
}
static inline double wgo_pcg64_uniform(wgo_pcg64* g, double low, double high) {
    return low + (high - low) * wgo_pcg64_double(g);
}
/* Generator.integers(0, high) with high-1 < 2^32 - 1 (numpy: buffered_bounded_lemire_uint32) */
static inline uint32_t wgo_pcg64_integers(wgo_pcg64* g, uint32_t high) {
    uint32_t rng = high - 1;
    if (rng == 0) return 0;
    uint32_t rng_excl = rng + 1;
    uint64_t m = (uint64_t)wgo_pcg64_next32(g) * (uint64_t)rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        uint32_t threshold = (0xffffffffu - rng) % rng_excl;
        while (leftover < threshold) {
            m = (uint64_t)wgo_pcg64_next32(g) * (uint64_t)rng_excl;
            leftover = (uint32_t)m;
        }
    }
    return (uint32_t)(m >> 32);
}
#endif
