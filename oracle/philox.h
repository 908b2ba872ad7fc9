/*
 * philox.h — ORACLE / TEST INFRASTRUCTURE.  Philox4x32-10 (Salmon et al., SC'11) + Box-Muller.
 *
 * The reference draws sensor noise from an *unseeded* generator (MesClass.farm_mes is its own gym.Env,
 * WindGym/WindEnv.py:37-42, MesClass.py:574-577), so its noise is not reproducible from the env seed.
 * The build replaces it by a counter-based stream: normal(key = env noise key, counter = (push index,
 * turbine, channel, episode)).  DESIGN.md §3.4 documents the deviation.
 */
#ifndef WGO_PHILOX_H
#define WGO_PHILOX_H
#include <math.h>
#include <stdint.h>

static inline void wgo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* standard normal from the top 24 bits of two 32-bit words: u1 in (0,1], u2 in [0,1) */
static inline double wgo_normal_from_u32(uint32_t a, uint32_t b) {
    double u1 = ((double)(a >> 8) + 1.0) * (1.0 / 16777216.0);
    double u2 = (double)(b >> 8) * (1.0 / 16777216.0);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}
static inline double wgo_noise_normal(uint64_t key, uint32_t push_idx, uint32_t turbine, uint32_t channel,
                                      uint32_t episode) {
    uint32_t ctr[4] = {push_idx, turbine, channel, episode};
    uint32_t k[2] = {(uint32_t)key, (uint32_t)(key >> 32)};
    uint32_t o[4];
    wgo_philox4x32_10(ctr, k, o);
    return wgo_normal_from_u32(o[0], o[1]);
}
/* inflow streams of model M0: standard normal keyed by the episode's turbulence seed */
static inline double wgo_turb_normal(uint32_t turb_seed, uint32_t step, uint32_t index, uint32_t comp, uint32_t kind) {
    uint32_t ctr[4] = {step, index, comp, kind};
    uint32_t k[2] = {turb_seed, 0x57474d30u};
    uint32_t o[4];
    wgo_philox4x32_10(ctr, k, o);
    return wgo_normal_from_u32(o[0], o[1]);
}
/* horizontal offset (fractions of the box length) of an episode into the shared frozen box */
static inline void wgo_turb_offset(uint32_t turb_seed, double* fx, double* fy) {
    uint32_t ctr[4] = {0, 0, 0, 0x4f};
    uint32_t k[2] = {turb_seed, 0x57474d30u};
    uint32_t o[4];
    wgo_philox4x32_10(ctr, k, o);
    *fx = (double)o[0] * (1.0 / 4294967296.0);
    *fy = (double)o[1] * (1.0 / 4294967296.0);
}
#endif
