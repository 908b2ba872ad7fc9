/* ORACLE / TEST INFRASTRUCTURE: thin exports of pcg64.h / philox.h so tests can pin them against numpy. */
#include "pcg64.h"
#include "philox.h"

void wgo_rng_uniform(uint64_t seed, int n, double low, double high, double* out) {
    wgo_pcg64 g;
    wgo_pcg64_seed(&g, seed);
    for (int i = 0; i < n; ++i) out[i] = wgo_pcg64_uniform(&g, low, high);
}
/* interleaved stream like the reference's reset: 3 uniforms, one integers(0, hi), n uniforms */
void wgo_rng_mixed(uint64_t seed, int n_rounds, uint32_t hi, int n_tail, double* out_u, uint32_t* out_i) {
    wgo_pcg64 g;
    wgo_pcg64_seed(&g, seed);
    int k = 0;
    for (int r = 0; r < n_rounds; ++r) {
        for (int i = 0; i < 3; ++i) out_u[k++] = wgo_pcg64_double(&g);
        out_i[r] = wgo_pcg64_integers(&g, hi);
        for (int i = 0; i < n_tail; ++i) out_u[k++] = wgo_pcg64_double(&g);
    }
}
void wgo_rng_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    wgo_philox4x32_10(ctr, key, out);
}
double wgo_rng_noise(uint64_t key, uint32_t push_idx, uint32_t turbine, uint32_t channel, uint32_t episode) {
    return wgo_noise_normal(key, push_idx, turbine, channel, episode);
}

/* the Philox stream of wg_generate_mann_box (windgym_amd/csrc/wg_mann.hip: mann_noise): complex standard normal of
 * (component c, cell idx), E|n|^2 = 1 — out[(c * cells + idx) * 2 + {0, 1}] = (re, im); float arithmetic like the kernel's */
void wgo_mann_noise(uint64_t seed, uint64_t cells, float* out) {
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (uint32_t c = 0; c < 3; ++c)
        for (uint64_t idx = 0; idx < cells; ++idx) {
            const uint32_t ctr[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), c, 0x4d414e4eu};
            uint32_t o[4];
            wgo_philox4x32_10(ctr, key, o);
            const float u1 = ((float)(o[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
            const float u2 = (float)(o[1] >> 8) * (1.0f / 16777216.0f);
            const float r = sqrtf(-logf(u1));
            out[(c * cells + idx) * 2] = r * cosf(6.2831853071795864f * u2);
            out[(c * cells + idx) * 2 + 1] = r * sinf(6.2831853071795864f * u2);
        }
}
