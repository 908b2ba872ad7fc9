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
