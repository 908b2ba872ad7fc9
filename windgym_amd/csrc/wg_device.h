// wg_device.h — device helpers shared by the kernels of libwindgym_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include "wg_state.h"

#define WG_PI_D 3.14159265358979323846
#define WG_DEG2RAD_F 0.017453292519943295f
#define WG_RAD2DEG_F 57.29577951308232f

// ---------------------------------------------------------------------------------------------------
// PCG64 + numpy Generator.uniform / .integers (gymnasium's np_random behind the reference's sampling,
// Wind_Farm_Env.py:557-568, WindEnv.py:24-35)
// ---------------------------------------------------------------------------------------------------
#define WG_PCG_MULT ((((wg_u128)2549297995355413924ULL) << 64) | (wg_u128)4865540595714422341ULL)

__device__ inline uint32_t wg_ss_hashmix(uint32_t value, uint32_t& hc) {
    value ^= hc;
    hc *= 0x931e8875u;
    value *= hc;
    value ^= value >> 16;
    return value;
}
__device__ inline uint32_t wg_ss_mix(uint32_t x, uint32_t y) {
    uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
    r ^= r >> 16;
    return r;
}
// np.random.default_rng(seed): SeedSequence(seed).generate_state(4, uint64) -> PCG64 srandom
__device__ inline void wg_pcg_seed(WgEnv& e, uint64_t seed) {
    uint32_t ent[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
    int n_ent = ent[1] != 0 ? 2 : 1;
    uint32_t pool[4];
    uint32_t hc = 0x43b0d7e5u;
    for (int i = 0; i < 4; ++i) pool[i] = wg_ss_hashmix(i < n_ent ? ent[i] : 0u, hc);
    for (int s = 0; s < 4; ++s)
        for (int dd = 0; dd < 4; ++dd)
            if (s != dd) pool[dd] = wg_ss_mix(pool[dd], wg_ss_hashmix(pool[s], hc));
    uint32_t hb = 0x8b51f9ddu;
    uint64_t st[4];
    for (int i = 0; i < 4; ++i) {
        uint32_t w2[2];
        for (int h = 0; h < 2; ++h) {
            uint32_t v = pool[(2 * i + h) & 3];
            v ^= hb;
            hb *= 0x58f38dedu;
            v *= hb;
            v ^= v >> 16;
            w2[h] = v;
        }
        st[i] = (uint64_t)w2[0] | ((uint64_t)w2[1] << 32);
    }
    wg_u128 initstate = ((wg_u128)st[0] << 64) | st[1];
    wg_u128 initseq = ((wg_u128)st[2] << 64) | st[3];
    e.rng_inc = (initseq << 1) | 1;
    e.rng_state = (wg_u128)0 * WG_PCG_MULT + e.rng_inc;
    e.rng_state += initstate;
    e.rng_state = e.rng_state * WG_PCG_MULT + e.rng_inc;
    e.rng_has32 = 0;
    e.rng_u32 = 0;
}
__device__ inline uint64_t wg_pcg_next64(WgEnv& e) {
    e.rng_state = e.rng_state * WG_PCG_MULT + e.rng_inc;
    uint64_t hi = (uint64_t)(e.rng_state >> 64), lo = (uint64_t)e.rng_state;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(e.rng_state >> 122);
    return (x >> rot) | (x << ((-rot) & 63));
}
__device__ inline uint32_t wg_pcg_next32(WgEnv& e) {
    if (e.rng_has32) {
        e.rng_has32 = 0;
        return e.rng_u32;
    }
    uint64_t n = wg_pcg_next64(e);
    e.rng_has32 = 1;
    e.rng_u32 = (uint32_t)(n >> 32);
    return (uint32_t)(n & 0xffffffffu);
}
__device__ inline double wg_pcg_uniform(WgEnv& e, double low, double high) {
    double u = (double)(wg_pcg_next64(e) >> 11) * (1.0 / 9007199254740992.0);
    return low + (high - low) * u;
}
__device__ inline uint32_t wg_pcg_integers(WgEnv& e, uint32_t high) {
    uint32_t rng = high - 1;
    if (rng == 0) return 0;
    uint32_t rng_excl = rng + 1;
    uint64_t m = (uint64_t)wg_pcg_next32(e) * (uint64_t)rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        uint32_t threshold = (0xffffffffu - rng) % rng_excl;
        while (leftover < threshold) {
            m = (uint64_t)wg_pcg_next32(e) * (uint64_t)rng_excl;
            leftover = (uint32_t)m;
        }
    }
    return (uint32_t)(m >> 32);
}

// ---------------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller: sensor noise stream (replaces the reference's unseeded generator,
// MesClass.py:574-577)
// ---------------------------------------------------------------------------------------------------
__device__ inline float wg_noise_normal(uint64_t key, uint32_t push_idx, uint32_t turbine, uint32_t channel,
                                        uint32_t episode) {
    uint32_t c0 = push_idx, c1 = turbine, c2 = channel, c3 = episode;
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
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
    // 24-bit uniforms keep everything in float32: u1 in (0,1], u2 in [0,1)
    const float u1 = ((float)(c0 >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(c1 >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853071795864f * u2);
}

// inflow streams of model M0: standard normal keyed by the episode's turbulence seed
__device__ inline void wg_philox_turb(uint32_t turb_seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                      uint32_t& o0, uint32_t& o1) {
    uint32_t k0 = turb_seed, k1 = 0x57474d30u;
#pragma unroll
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
    o0 = c0; o1 = c1;
}
__device__ inline float wg_turb_normal(uint32_t turb_seed, uint32_t step, uint32_t index, uint32_t comp, uint32_t kind) {
    uint32_t a, b;
    wg_philox_turb(turb_seed, step, index, comp, kind, a, b);
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853071795864f * u2);
}

// ---------------------------------------------------------------------------------------------------
// tabular turbine: linear interpolation, 0 outside the table
// ---------------------------------------------------------------------------------------------------
template <typename T>
__device__ inline T wg_tab_interp(const T* xs, const T* ys, int n, T x) {
    if (!(x >= xs[0]) || x > xs[n - 1]) return (T)0;
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (xs[mid] <= x) lo = mid; else hi = mid;
    }
    T f = (x - xs[lo]) / (xs[hi] - xs[lo]);
    return ys[lo] + f * (ys[hi] - ys[lo]);
}

// ---------------------------------------------------------------------------------------------------
// wave64 reductions
// ---------------------------------------------------------------------------------------------------
__device__ inline float wg_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline int wg_wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wg_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline double wg_wave_min_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline double wg_wave_max_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------------
// MesClass (WindGym/MesClass.py)
// ---------------------------------------------------------------------------------------------------
struct WgRing {
    const float* data;
    int n_pushed, hlen;
    int n_avail, start;      // start = physical slot of the oldest sample (one modulo per ring, not per element)
    __device__ WgRing() {}
    __device__ WgRing(const float* d_, int n_pushed_, int hlen_) : data(d_), n_pushed(n_pushed_), hlen(hlen_) {
        n_avail = n_pushed < hlen ? n_pushed : hlen;
        start = (n_pushed - n_avail) % hlen;
    }
    __device__ int avail() const { return n_avail; }
    __device__ float at(int q) const {   // q = 0 oldest
        int phys = start + q;
        if (phys >= hlen) phys -= hlen;
        return data[phys];
    }
};

// turb_mes._scale_val in float32 (MesClass.py:324-326)
__device__ inline float wg_scale(float v, float mn, float rng) {
    float t = v - mn;
    t = 2.0f * t;
    t = t / rng;
    return t - 1.0f;
}

// Mes.get_measurements (MesClass.py:70-125): appends `current` + history_N window means, scaled
__device__ inline int wg_mes_get(const wg_channel& c, bool cur_on, bool rol_on, const WgRing& r, float mn,
                                 float rng, float* out) {
    int n = 0;
    const int avail = r.avail();
    if (avail == 0) return 0;
    if (cur_on) out[n++] = wg_scale(r.at(avail - 1), mn, rng);
    if (rol_on) {
        const int W = c.window_len;
        for (int i = 0; i < c.history_n; ++i) {
            int lo, hi;
            if (i == 0) {
                lo = avail - W; if (lo < 0) lo = 0; hi = avail;
            } else if (i == c.history_n - 1 && avail >= W) {
                lo = 0; hi = W;
            } else if (avail < W) {
                lo = 0; hi = avail;
            } else {
                int spacing = (avail - W) / (c.history_n - 1);
                if (spacing < 1) spacing = 1;
                int pos = i * spacing;
                if (pos > avail - W) pos = avail - W;
                lo = pos; hi = pos + W;
            }
            float s = 0.f;
            for (int q = lo; q < hi; ++q) s += r.at(q);
            out[n++] = wg_scale(s / (float)(hi - lo), mn, rng);
        }
    }
    return n;
}

// turb_mes.calc_TI (MesClass.py:220-237), unscaled
__device__ inline float wg_calc_ti(const WgRing& r) {
    const int avail = r.avail();
    float U = 0.f;
    for (int q = 0; q < avail; ++q) U += r.at(q);
    U /= (float)avail;
    float m2 = 0.f;
    for (int q = 0; q < avail; ++q) {
        float dv = r.at(q) - U;
        m2 += dv * dv;
    }
    return sqrtf(m2 / (float)avail) / U;
}

__device__ inline float wg_clip1(float v) { return fminf(fmaxf(v, -1.0f), 1.0f); }

// one turb_mes.get_measurements(scaled=True) block (MesClass.py:328-340).  farm_level selects the
// farm_mes.farm_mes object used by the PettingZoo facade (WindEnvMulti.py:90-92).
// (rbase / fbase: the context's turbine / farm rings — global memory or a staged LDS copy)
__device__ inline int wg_turb_block_b(const WgParams& p, const float* rbase, const float* fbase, int n_pushed, int t,
                                      bool farm_level, float* out) {
    int n = 0;
    for (int ch = 0; ch < WG_N_CH; ++ch) {
        if (ch == WG_CH_POWER) {
            bool ti_on = farm_level ? p.farm_ti : p.turb_ti;
            if (ti_on) {
                const int H = p.ch[WG_CH_WS].history_len;
                WgRing r{farm_level ? fbase + p.fring_off[WG_CH_WS] : rbase + p.ring_off[WG_CH_WS] + (size_t)t * H,
                         n_pushed, H};
                out[n++] = wg_scale(wg_calc_ti(r), p.ti_min_f, p.ti_rng_f);
            }
        }
        const int H = p.ch[ch].history_len;
        bool on = farm_level ? (p.farm_on[ch] != 0) : (p.turb_on[ch] != 0);
        WgRing r;
        if (farm_level) {
            // the farm object's yaw deque is never filled: the flags would allow it, the empty deque yields []
            r = WgRing(fbase + p.fring_off[ch], (ch == WG_CH_YAW) ? 0 : n_pushed, H);
            if (ch == WG_CH_YAW) on = true;
        } else {
            r = WgRing(rbase + p.ring_off[ch] + (size_t)t * H, n_pushed, H);
        }
        float rng = (farm_level && ch == WG_CH_POWER) ? p.sc_rng_farm_power : p.sc_rng[ch];
        n += wg_mes_get(p.ch[ch], p.ch[ch].current && on, p.ch[ch].rolling_mean && on, r, p.sc_min[ch], rng,
                        out + n);
    }
    return n;
}
__device__ inline int wg_turb_block(const WgParams& p, const WgPtrs& d, int ctx_id, int n_pushed, int t,
                                    bool farm_level, float* out) {
    return wg_turb_block_b(p, d.ring + (size_t)ctx_id * p.ring_stride, d.fring + (size_t)ctx_id * p.fring_stride,
                           n_pushed, t, farm_level, out);
}
