// The observation builder (farm_mes.get_measurements(scaled=True) + clip) — shared by the glue kernels (wg_kernels.hip)
// and by k_flow's epilogue, which builds a background episode's FIRST observation when its development completes
// (wg_flow.hip: wg_first_obs).
#pragma once
#include "wg_device.h"

// ===================================================================================================
// observation (farm_mes.get_measurements(scaled=True) + clip, MesClass.py:679-703, Wind_Farm_Env.py:513-520)
// ===================================================================================================
// `raw`: unscaled, unclipped sensor values in the same layout (the "... measured" entries of the info dict,
// farm_measurements.get_*_turb() / get_*_farm(), Wind_Farm_Env.py:529-537)
#define WG_OBSV(v, mn, rng) (raw ? (v) : wg_clip1(wg_scale((v), (mn), (rng))))
// `obs_m` (optional): this env's slice of the per-agent observation buffer [N][obs_dim_multi] of the PettingZoo facade
// (WindEnvMulti._get_obs_multi, WindEnvMulti.py:79-103): agent t = its own turbine block (the values just computed) ++
// the farm_mes.farm_mes block, which differs from the single-agent farm block in its TI entry
// Group-parallel window sums: the L = 2^k lanes of a group (adjacent lanes, sub = lane % L) share one turbine; lane
// `sub` sums the window's elements lo + sub, lo + sub + L, ... and the partial sums are combined across the group.
// (One lane per turbine walked its 25 + 10 window elements as a chain of dependent LDS round trips on 16 of the
// wave's 64 lanes: 6.6 us of k_glue's 18 on cfg2.  Summation order differs from the oracle's sequential sum by
// float rounding only — covered by the 2e-4 observation tolerance.)
template <int L>
__device__ inline float wg_group_sum(float s) {
#pragma unroll
    for (int o = 1; o < L; o <<= 1) s += __shfl_xor(s, o, 64);
    return s;
}
// (reads in flight per lane: eight when a lane walks a whole window, four when it walks a quarter or less of it)
template <int L>
__device__ inline float wg_ring_sum_g(const WgRing& r, const int lo, const int hi, const int sub) {
    constexpr int U = L == 1 ? 8 : 4;
    float s = 0.f;
#pragma nounroll
    for (int q = lo + sub; q < hi; q += U * L) {
        float v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = r.at(min(q + k * L, hi - 1));
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (q + k * L < hi) s += v[k];
    }
    return wg_group_sum<L>(s);
}
// turb_mes.calc_TI (MesClass.py:220-237), unscaled, across the group
template <int L>
__device__ inline float wg_calc_ti_g(const WgRing& r, const int sub) {
    constexpr int U = L == 1 ? 8 : 4;
    const int avail = r.avail();
    const float U_ = wg_ring_sum_g<L>(r, 0, avail, sub) / (float)avail;
    float m2 = 0.f;
#pragma nounroll
    for (int q = sub; q < avail; q += U * L) {
        float v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = r.at(min(q + k * L, avail - 1));
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (q + k * L < avail) { const float dv = v[k] - U_; m2 += dv * dv; }
    }
    m2 = wg_group_sum<L>(m2);
    return sqrtf(m2 / (float)avail) / U_;
}

// L (1, 2, 4 or 8; N * L <= 64 unless L == 1): lanes per turbine, see above
template <int L>
__device__ inline void build_obs(const WgParams& p, const WgPtrs& d, int ctx_id, int lane, float* __restrict__ obs,
                                 float* __restrict__ obs2, const float* rbase, const float* fbase,
                                 const bool raw = false, float* __restrict__ obs_m = nullptr,
                                 const int n_pushed_known = -1, float* mscratch = nullptr) {
    const int N = p.N;
    // (callers that already hold the context header pass it: one dependent load less on the kernel's latency chain)
    const int n_pushed = n_pushed_known >= 0 ? n_pushed_known : d.ctx[ctx_id].n_pushed;
    float ti_sum = 0.f;
    float buf[8];
    const int sub = lane & (L - 1), per_pass = WG_WAVE / L;
    const bool wr = sub == 0;
    int n_turb_vals = 0;      // values per turbine block (0 before the first push)
    for (int t = lane / L; t < N; t += per_pass) {
        // turbine block written straight to its place (block length is fixed = turb_obs)
        float* o = obs + (size_t)t * p.turb_obs;
        int n = 0;
        for (int ch = 0; ch < WG_N_CH; ++ch) {
            const int H = p.ch[ch].history_len;
            if (ch == WG_CH_POWER && p.turb_ti) {
                WgRing r(rbase + p.ring_off[WG_CH_WS] + t, n_pushed, p.ch[WG_CH_WS].history_len, N);
                float v = WG_OBSV(wg_calc_ti_g<L>(r, sub), p.ti_min_f, p.ti_rng_f);
                if (wr) {
                    o[n] = v; if (obs2) obs2[(size_t)t * p.turb_obs + n] = v;
                    if (obs_m) obs_m[(size_t)t * p.obs_dim_multi + n] = v;
                }
                ++n;
            }
            WgRing r(rbase + p.ring_off[ch] + t, n_pushed, H, N);
            const bool on = p.turb_on[ch] != 0;
            const bool cur_on = p.ch[ch].current && on, rol_on = p.ch[ch].rolling_mean && on;
            // stream the values out one at a time (window count is unbounded: history_N up to 100s)
            const int avail = r.avail();
            if (avail == 0) continue;
            if (cur_on) {
                float v = WG_OBSV(r.at(avail - 1), p.sc_min[ch], p.sc_rng[ch]);
                if (wr) {
                    o[n] = v; if (obs2) obs2[(size_t)t * p.turb_obs + n] = v;
                    if (obs_m) obs_m[(size_t)t * p.obs_dim_multi + n] = v;
                }
                ++n;
            }
            if (rol_on) {
                const int W = p.ch[ch].window_len, HN = p.ch[ch].history_n;
                for (int i = 0; i < HN; ++i) {
                    int lo, hi;
                    if (i == 0) { lo = avail - W; if (lo < 0) lo = 0; hi = avail; }
                    else if (i == HN - 1 && avail >= W) { lo = 0; hi = W; }
                    else if (avail < W) { lo = 0; hi = avail; }
                    else {
                        int spacing = (avail - W) / (HN - 1); if (spacing < 1) spacing = 1;
                        int pos = i * spacing; if (pos > avail - W) pos = avail - W;
                        lo = pos; hi = pos + W;
                    }
                    const float s = wg_ring_sum_g<L>(r, lo, hi, sub);
                    float v = WG_OBSV(s / (float)(hi - lo), p.sc_min[ch], p.sc_rng[ch]);
                    if (wr) {
                        o[n] = v; if (obs2) obs2[(size_t)t * p.turb_obs + n] = v;
                        if (obs_m) obs_m[(size_t)t * p.obs_dim_multi + n] = v;
                    }
                    ++n;
                }
            }
        }
        n_turb_vals = n;
        if (p.farm_ti) {   // farm TI = mean of the *scaled* turbine TIs (MesClass.py:670-673)
            WgRing r(rbase + p.ring_off[WG_CH_WS] + t, n_pushed, p.ch[WG_CH_WS].history_len, N);
            const float ti = wg_calc_ti_g<L>(r, sub);
            if (wr) ti_sum += raw ? ti : wg_scale(ti, p.ti_min_f, p.ti_rng_f);
        }
    }
    if (obs_m) {
        // the agents' farm_mes.farm_mes block is the same for every agent: computed once (lane 0, into the wave's LDS
        // scratch), clipped and copied behind each agent's turbine block by the whole wave.  (Every agent's lane used
        // to recompute it in place — N identical serial chains and the code that set the kernel's register pressure.)
        const int nt = __shfl(n_turb_vals, 0, 64);
        int m = 0;
        if (lane == 0) m = wg_turb_block_b(p, rbase, fbase, n_pushed, 0, true, mscratch);
        m = __shfl(m, 0, 64);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // single wave: program order suffices
        for (int i = lane; i < N * m; i += WG_WAVE) {
            const int t = i / m, k = i - t * m;
            obs_m[(size_t)t * p.obs_dim_multi + nt + k] = wg_clip1(mscratch[k]);
        }
    }
    if (p.farm_ti) ti_sum = wg_wave_sum(ti_sum);
    if (lane == 0 && p.farm_obs > 0) {
        float* o = obs + (size_t)N * p.turb_obs;
        float* o2 = obs2 ? obs2 + (size_t)N * p.turb_obs : nullptr;
        int n = 0;
        const int chs[3] = {WG_CH_WS, WG_CH_WD, WG_CH_POWER};
        for (int ci = 0; ci < 3; ++ci) {
            const int ch = chs[ci];
            if (ch == WG_CH_POWER && p.farm_ti) {
                float v = raw ? ti_sum / (float)N : wg_clip1(ti_sum / (float)N);
                o[n] = v; if (o2) o2[n] = v;
                ++n;
            }
            if (!p.farm_on[ch]) continue;
            WgRing r(fbase + p.fring_off[ch], n_pushed, p.ch[ch].history_len);
            const float rng = ch == WG_CH_POWER ? p.sc_rng_farm_power : p.sc_rng[ch];
            // farm windows are few: reuse the generic helper through a small buffer when it fits
            const int cnt = (p.ch[ch].current ? 1 : 0) + (p.ch[ch].rolling_mean ? p.ch[ch].history_n : 0);
            if (cnt <= 8 && !raw) {
                int m = wg_mes_get(p.ch[ch], p.ch[ch].current, p.ch[ch].rolling_mean, r, p.sc_min[ch], rng, buf);
                for (int i = 0; i < m; ++i) { float v = wg_clip1(buf[i]); o[n] = v; if (o2) o2[n] = v; ++n; }
            } else {
                const int avail = r.avail();
                if (avail == 0) continue;
                if (p.ch[ch].current) { float v = WG_OBSV(r.at(avail - 1), p.sc_min[ch], rng); o[n] = v; if (o2) o2[n] = v; ++n; }
                if (p.ch[ch].rolling_mean) {
                    const int W = p.ch[ch].window_len, HN = p.ch[ch].history_n;
                    for (int i = 0; i < HN; ++i) {
                        int lo, hi;
                        if (i == 0) { lo = avail - W; if (lo < 0) lo = 0; hi = avail; }
                        else if (i == HN - 1 && avail >= W) { lo = 0; hi = W; }
                        else if (avail < W) { lo = 0; hi = avail; }
                        else {
                            int spacing = (avail - W) / (HN - 1); if (spacing < 1) spacing = 1;
                            int pos = i * spacing; if (pos > avail - W) pos = avail - W;
                            lo = pos; hi = pos + W;
                        }
                        const float s = wg_ring_sum(r, lo, hi);
                        float v = WG_OBSV(s / (float)(hi - lo), p.sc_min[ch], rng);
                        o[n] = v; if (o2) o2[n] = v; ++n;
                    }
                }
            }
        }
    }
}

// ===================================================================================================
// Sums mode (WgParams::sums_mode): the same observation from the running window sums the flow kernels maintain
// (WgPtrs::wsum / wcur) — per turbine a handful of values at addresses that depend on the context only, so the glue
// kernel requests them together with the context header instead of staging the rings behind it.
// ===================================================================================================
struct ObsIn {
    double S[WG_N_SUMS];
    float cur[WG_N_CH];
};
// entity `ent`: turbine t, or N = the farm-level deques
__device__ inline ObsIn wg_obs_load(const WgParams& p, const WgPtrs& d, const int ctx_id, const int ent) {
    ObsIn o;
    const int NS = p.N + 1;
    const double* ws_ = d.wsum + (size_t)ctx_id * WG_N_SUMS * NS + ent;
    const float* wc_ = d.wcur + (size_t)ctx_id * WG_N_CH * NS + ent;
    const unsigned sm = ent == p.N ? p.sum_mask_f : p.sum_mask_t, cm = ent == p.N ? p.cur_mask_f : p.cur_mask_t;
#pragma unroll
    for (int s = 0; s < WG_N_SUMS; ++s) o.S[s] = ((sm >> s) & 1u) ? ws_[(size_t)s * NS] : 0.0;
#pragma unroll
    for (int ch = 0; ch < WG_N_CH; ++ch) o.cur[ch] = ((cm >> ch) & 1u) ? wc_[(size_t)ch * NS] : 0.f;
    return o;
}
// mean of the newest min(window, available) samples: Mes.get_measurements' window 0 (MesClass.py:85-91)
__device__ inline float wg_sums_mean(const WgParams& p, const ObsIn& o, const int ch, const int n_pushed) {
    const int H = p.ch[ch].history_len, W = p.ch[ch].window_len;
    const int avail = n_pushed < H ? n_pushed : H;
    const int cnt = W < avail ? W : avail;
    return (float)o.S[ch] / (float)cnt;
}
// turb_mes.calc_TI (MesClass.py:220-237) from the deque's sum and sum of squares
__device__ inline float wg_sums_ti(const WgParams& p, const ObsIn& o, const int n_pushed) {
    const int H = p.ch[WG_CH_WS].history_len;
    const double n = (double)(n_pushed < H ? n_pushed : H);
    const double U = o.S[WG_SUM_TI1] / n;
    double var = o.S[WG_SUM_TI2] / n - U * U;
    if (!(var > 0.0)) var = 0.0;
    return (float)(sqrt(var) / U);
}

// `first`: wg_obs_load(ctx, lane) requested early by the caller (entity = this lane's first turbine; ignored for lanes >= N)
template <bool MULTI>
__device__ inline void build_obs_sums(const WgParams& p, const WgPtrs& d, const int ctx_id, const int lane,
                                      float* __restrict__ obs, float* __restrict__ obs2, float* __restrict__ obs_m,
                                      const int n_pushed, float* mscratch, const ObsIn& first) {
    const int N = p.N;
    float ti_sum = 0.f;
    int n_turb_vals = 0;
#define WG_EMIT(v_)                                                         \
    do {                                                                    \
        const float _v = (v_);                                              \
        o[n] = _v;                                                          \
        if (obs2) obs2[(size_t)t * p.turb_obs + n] = _v;                    \
        if (MULTI && obs_m) obs_m[(size_t)t * p.obs_dim_multi + n] = _v;    \
        ++n;                                                                \
    } while (0)
    for (int t = lane; t < N; t += WG_WAVE) {
        const ObsIn oi = t == lane ? first : wg_obs_load(p, d, ctx_id, t);
        float* o = obs + (size_t)t * p.turb_obs;
        int n = 0;
        const float ti = (p.turb_ti || p.farm_ti) ? wg_sums_ti(p, oi, n_pushed) : 0.f;
#pragma unroll
        for (int ch = 0; ch < WG_N_CH; ++ch) {
            if (ch == WG_CH_POWER && p.turb_ti) WG_EMIT(wg_clip1(wg_scale(ti, p.ti_min_f, p.ti_rng_f)));
            const bool on = p.turb_on[ch] != 0;
            if (n_pushed == 0) continue;
            if (p.ch[ch].current && on) WG_EMIT(wg_clip1(wg_scale(oi.cur[ch], p.sc_min[ch], p.sc_rng[ch])));
            if (p.ch[ch].rolling_mean && on) WG_EMIT(wg_clip1(wg_scale(wg_sums_mean(p, oi, ch, n_pushed), p.sc_min[ch], p.sc_rng[ch])));
        }
        n_turb_vals = n;
        if (p.farm_ti) ti_sum += wg_scale(ti, p.ti_min_f, p.ti_rng_f);   // mean of the *scaled* turbine TIs (MesClass.py:670-673)
    }
#undef WG_EMIT
    const bool farm_any = p.farm_obs > 0 || (MULTI && obs_m != nullptr);
    if (!farm_any) return;
    if (p.farm_ti) ti_sum = wg_wave_sum(ti_sum);
    // farm-level entity (lane 0): the farm deques' sums
    ObsIn of{};
    if (lane == 0 && (p.sum_mask_f | p.cur_mask_f)) of = wg_obs_load(p, d, ctx_id, N);
    if (MULTI && obs_m) {
        // the agents' farm_mes.farm_mes block (WindEnvMulti.py:90-92; wg_turb_block_b with farm_level): same for every
        // agent — computed once into the wave's LDS scratch, clipped and copied behind each agent's turbine block
        const int nt = __shfl(n_turb_vals, 0, 64);
        int m = 0;
        if (lane == 0) {
#pragma unroll
            for (int ch = 0; ch < WG_N_CH; ++ch) {
                if (ch == WG_CH_POWER && p.farm_ti) mscratch[m++] = wg_scale(wg_sums_ti(p, of, n_pushed), p.ti_min_f, p.ti_rng_f);
                if (ch == WG_CH_YAW || !p.farm_on[ch] || n_pushed == 0) continue;     // (the farm object's yaw deque is never filled)
                const float rng = ch == WG_CH_POWER ? p.sc_rng_farm_power : p.sc_rng[ch];
                if (p.ch[ch].current) mscratch[m++] = wg_scale(of.cur[ch], p.sc_min[ch], rng);
                if (p.ch[ch].rolling_mean) mscratch[m++] = wg_scale(wg_sums_mean(p, of, ch, n_pushed), p.sc_min[ch], rng);
            }
        }
        m = __shfl(m, 0, 64);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // single wave: program order suffices
        for (int i = lane; i < N * m; i += WG_WAVE) {
            const int t = i / m, k = i - t * m;
            obs_m[(size_t)t * p.obs_dim_multi + nt + k] = wg_clip1(mscratch[k]);
        }
    }
    if (lane == 0 && p.farm_obs > 0) {
        float* o = obs + (size_t)N * p.turb_obs;
        float* o2 = obs2 ? obs2 + (size_t)N * p.turb_obs : nullptr;
        int n = 0;
        const int chs[3] = {WG_CH_WS, WG_CH_WD, WG_CH_POWER};
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const int ch = chs[ci];
            if (ch == WG_CH_POWER && p.farm_ti) { const float v = wg_clip1(ti_sum / (float)N); o[n] = v; if (o2) o2[n] = v; ++n; }
            if (!p.farm_on[ch] || n_pushed == 0) continue;
            const float rng = ch == WG_CH_POWER ? p.sc_rng_farm_power : p.sc_rng[ch];
            if (p.ch[ch].current) { const float v = wg_clip1(wg_scale(of.cur[ch], p.sc_min[ch], rng)); o[n] = v; if (o2) o2[n] = v; ++n; }
            if (p.ch[ch].rolling_mean) {
                const float v = wg_clip1(wg_scale(wg_sums_mean(p, of, ch, n_pushed), p.sc_min[ch], rng));
                o[n] = v; if (o2) o2[n] = v; ++n;
            }
        }
    }
}
