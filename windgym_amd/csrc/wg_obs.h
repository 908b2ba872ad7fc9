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
                WgRing r(rbase + p.ring_off[WG_CH_WS] + t, n_pushed, p.ch[WG_CH_WS].history_len, N, p.ring_cap[WG_CH_WS]);
                float v = WG_OBSV(wg_calc_ti_g<L>(r, sub), p.ti_min_f, p.ti_rng_f);
                if (wr) {
                    o[n] = v; if (obs2) obs2[(size_t)t * p.turb_obs + n] = v;
                    if (obs_m) obs_m[(size_t)t * p.obs_dim_multi + n] = v;
                }
                ++n;
            }
            WgRing r(rbase + p.ring_off[ch] + t, n_pushed, H, N, p.ring_cap[ch]);
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
            WgRing r(rbase + p.ring_off[WG_CH_WS] + t, n_pushed, p.ch[WG_CH_WS].history_len, N, p.ring_cap[WG_CH_WS]);
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
            WgRing r(fbase + p.fring_off[ch], n_pushed, p.ch[ch].history_len, 1, p.ring_cap[ch]);
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
// Sums mode (WgParams::sums_mode): the same observation from running window sums (WgPtrs::wsum, kept by k_glue_lean).
// ===================================================================================================
struct ObsIn {              // one entity (turbine t, or N = the farm-level deques): window sums + newest samples
    double S[WG_N_SUMS];
    float cur[WG_N_CH];
};
struct SumsRaw {            // what one step's update of an entity reads: old sums, newest sample per channel, leaving sample per slot
    double S[WG_N_SUMS];
    float nw[WG_N_CH];
    float lv[WG_N_SUMS];
};
// n % H for 0 <= n, n * H < 2^32, magic = floor(2^32 / H) + 1 (host: WgParams::ring_magic) — integer only, stays on the
// scalar unit for wave-uniform n
__device__ __forceinline__ int wg_umod(int n, int H, unsigned magic) {
    if (H == 1) return 0;
    int r = n - (int)__umulhi((unsigned)n, magic) * H;
    if (r >= H) r -= H;
    if (r < 0) r += H;
    return r;
}
struct SumsEnt {            // where entity `ent` lives
    const float* rb; int stride; bool farm; unsigned sm, cm;
};
__device__ __forceinline__ SumsEnt wg_sums_ent(const WgParams& p, const WgPtrs& d, const int ctx_id, const int ent) {
    SumsEnt q;
    q.farm = ent == p.N;
    q.rb = q.farm ? d.fring + (size_t)ctx_id * p.fring_stride : d.ring + (size_t)ctx_id * p.ring_stride + ent;
    q.stride = q.farm ? 1 : p.N;
    q.sm = q.farm ? p.sum_mask_f : p.sum_mask_t;
    q.cm = q.farm ? p.cur_mask_f : p.cur_mask_t;
    return q;
}
// the loads of one step's update: `np` pushes are in the sums, the flow kernel has pushed sample number np since
// (NEWEST = false: only what does not depend on the step in flight — the old sums and the leaving samples; k_flow_env's pass wave
// fetches those while the step runs)
template <bool GEN, bool NEWEST = true>
__device__ inline SumsRaw wg_sums_load(const WgParams& p, const WgPtrs& d, const int e, const int ctx_id, const int ent, const int np) {
    constexpr int NSL = GEN ? WG_N_SUMS : WG_N_CH;        // (without TI entries only the four window slots exist)
    SumsRaw r;
    const SumsEnt q = wg_sums_ent(p, d, ctx_id, ent);
    const int NS = p.N + 1;
    const double* ws_ = d.wsum + (size_t)ctx_id * WG_N_SUMS * NS + ent;
    const unsigned ti = GEN ? q.sm >> WG_SUM_TI1 : 0u;      // (the TI slots read the ws channel)
#pragma unroll
    for (int ch = 0; ch < WG_N_CH; ++ch) {
        const bool need = (((q.sm | q.cm) >> ch) & 1u) || (ch == WG_CH_WS && ti);
        const int off = q.farm ? p.fring_off[ch] : p.ring_off[ch];
        r.nw[ch] = (NEWEST && need) ? q.rb[off + wg_umod(np, p.ring_cap[ch], p.ring_magic[ch]) * q.stride] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < WG_N_SUMS; ++s) { r.S[s] = 0.0; r.lv[s] = 0.f; }
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
        if ((q.sm >> s) & 1u) {
            const int ch = s < WG_N_CH ? s : WG_CH_WS;
            const int off = q.farm ? p.fring_off[ch] : p.ring_off[ch];
            const int Wc = p.sum_w[s];
            r.S[s] = ws_[(size_t)s * NS];
            // (the ring holds one sample more than the deque: the sample pushed Wc <= H pushes before the newest is still there)
            if (np >= Wc) r.lv[s] = q.rb[off + wg_umod(np - Wc, p.ring_cap[ch], p.ring_magic[ch]) * q.stride];
        }
    }
    return r;
}
// S += newest - leaving, stored.  Exact in double for the LINEAR slots as long as the samples' exponents stay within ~29 bits of
// each other (a sum of <= 100 floats then fits the 53-bit mantissa: adding and later subtracting the same float cancels without
// rounding — wind speeds, yaws, powers of one farm do).  NOT exact for WG_SUM_TI2: v * v needs 48 mantissa bits, a window of
// them more than 53, so that slot rounds, depends on the summation order (sequential here, grouped partial sums in
// wg_first_obs / lean_swap) and carries its rounding through an episode — far below the 2e-6 the tests hold TI entries to, but
// configurations WITH TI entries reproduce a checkpointed run to rounding only, not bit for bit.
template <bool GEN>
__device__ inline ObsIn wg_sums_apply(const WgParams& p, const WgPtrs& d, const int ctx_id, const int ent, const SumsRaw& r, const bool store) {
    constexpr int NSL = GEN ? WG_N_SUMS : WG_N_CH;
    ObsIn o;
    const int NS = p.N + 1;
    const unsigned sm = ent == p.N ? p.sum_mask_f : p.sum_mask_t;
    double* ws_ = d.wsum + (size_t)ctx_id * WG_N_SUMS * NS + ent;
#pragma unroll
    for (int s = 0; s < WG_N_SUMS; ++s) o.S[s] = 0.0;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
        if ((sm >> s) & 1u) {
            const int ch = s < WG_N_CH ? s : WG_CH_WS;
            double v = (double)r.nw[ch], l = (double)r.lv[s];
            if (s == WG_SUM_TI2) { v = v * v; l = l * l; }
            o.S[s] = (r.S[s] - l) + v;
            if (store) ws_[(size_t)s * NS] = o.S[s];
        }
    }
#pragma unroll
    for (int ch = 0; ch < WG_N_CH; ++ch) o.cur[ch] = r.nw[ch];
    return o;
}
__device__ __forceinline__ float wg_scale_r(const float v, const float mn, const float inv_rng) {
    return wg_clip1(2.0f * (v - mn) * inv_rng - 1.0f);
}
// mean of the newest min(window, available) samples — Mes.get_measurements' window 0 (MesClass.py:85-91) — from its sum
template <typename OC>
__device__ __forceinline__ float wg_sums_mean(const OC& oc, const double S, const int ch, const int n_pushed) {
    const int H = oc.hlen[ch], W = oc.wlen[ch];
    const int wc = W < H ? W : H;
    if (n_pushed >= wc) return (float)(S * oc.inv_w[ch]);
    return (float)(S / (double)n_pushed);                   // (the window is still filling: rare)
}
// turb_mes.calc_TI (MesClass.py:220-237) from the ws deque's sum and sum of squares
template <typename OC>
__device__ __forceinline__ float wg_sums_ti(const OC& oc, const double S1, const double S2, const int n_pushed) {
    const int H = oc.hlen[WG_CH_WS];
    const double n = (double)(n_pushed < H ? n_pushed : H);
    const double U = S1 / n;
    double var = S2 / n - U * U;
    if (!(var > 0.0)) var = 0.0;
    return (float)(sqrt(var) / U);
}
// One turbine's block of the observation (turb_mes.get_measurements(scaled=True) + clip, MesClass.py:328-340) from its
// window sums S[WG_N_SUMS] and newest samples cur[WG_N_CH].
// o / o2 / om: the block's place in obs, (optional) final_obs and (optional) the agent's row of the per-agent buffer.
// GEN = false: instantiation for configurations without TI entries (the common case: no double-precision sqrt / division
// in the kernel at all).
template <bool GEN, typename OC>
__device__ __forceinline__ int wg_obs_turbine(const OC& oc, const double* S, const float* cur, const int n_pushed,
                                              float* __restrict__ o, float* __restrict__ o2, float* __restrict__ om) {
    int n = 0;
#define WG_EMIT(v_) do { const float _v = (v_); o[n] = _v; if (o2) o2[n] = _v; if (om) om[n] = _v; ++n; } while (0)
#pragma unroll
    for (int ch = 0; ch < WG_N_CH; ++ch) {
        if (GEN && ch == WG_CH_POWER && oc.turb_ti)
            WG_EMIT(wg_scale_r(wg_sums_ti(oc, S[WG_SUM_TI1], S[WG_SUM_TI2], n_pushed), oc.ti_mn, oc.inv_ti_rng));
        if (n_pushed == 0) continue;
        if ((oc.cur_mask >> ch) & 1u) WG_EMIT(wg_scale_r(cur[ch], oc.mn[ch], oc.inv_rng[ch]));
        if ((oc.rol_mask >> ch) & 1u) WG_EMIT(wg_scale_r(wg_sums_mean(oc, S[ch], ch, n_pushed), oc.mn[ch], oc.inv_rng[ch]));
    }
#undef WG_EMIT
    return n;
}

// The whole observation from the sums.  `first`: the sums of entity `lane` (ignored for lanes >= N); get(ent): those of any
// other entity — turbines beyond the 64th, the farm-level deques (step: update + store, swap / reset: summed afresh).
// n_pushed: samples in the deques.
// GEN = false: nothing farm-level and no TI in the observation (wg_launch_glue decides) — the turbine blocks are all of it.
template <bool MULTI, bool GEN, typename Get>
__device__ inline void build_obs_sums(const WgParams& p, const int lane, float* __restrict__ obs, float* __restrict__ obs2,
                                      float* __restrict__ obs_m, const int n_pushed, float* mscratch, const ObsIn& first, Get get) {
    const int N = p.N;
    float ti_sum = 0.f;
    int n_turb_vals = 0;
    for (int t = lane; t < N; t += WG_WAVE) {
        const ObsIn oi = t == lane ? first : get(t);
        n_turb_vals = wg_obs_turbine<GEN>(p.oc, oi.S, oi.cur, n_pushed, obs + (size_t)t * p.turb_obs,
                                          obs2 ? obs2 + (size_t)t * p.turb_obs : nullptr,
                                          (MULTI && obs_m) ? obs_m + (size_t)t * p.obs_dim_multi : nullptr);
        // farm TI = mean of the *scaled* (unclipped) turbine TIs (MesClass.py:670-673)
        if (GEN && p.farm_ti) ti_sum += 2.0f * (wg_sums_ti(p.oc, oi.S[WG_SUM_TI1], oi.S[WG_SUM_TI2], n_pushed) - p.oc.ti_mn) * p.oc.inv_ti_rng - 1.0f;
    }
    if (!GEN) return;
    const bool farm_any = p.farm_obs > 0 || (MULTI && obs_m != nullptr);
    if (!farm_any) return;
    if (p.farm_ti) ti_sum = wg_wave_sum(ti_sum);
    // farm-level entity (lane 0): the farm deques' sums
    ObsIn of{};
    if (lane == 0 && (p.sum_mask_f | p.cur_mask_f)) of = get(N);
    if (MULTI && obs_m) {
        // the agents' farm_mes.farm_mes block (WindEnvMulti.py:90-92; wg_turb_block_b with farm_level): same for every
        // agent — computed once into the wave's LDS scratch and copied behind each agent's turbine block
        const int nt = __shfl(n_turb_vals, 0, 64);
        int m = 0;
        if (lane == 0) {
#pragma unroll
            for (int ch = 0; ch < WG_N_CH; ++ch) {
                if (ch == WG_CH_POWER && p.farm_ti)
                    mscratch[m++] = wg_scale_r(wg_sums_ti(p.oc, of.S[WG_SUM_TI1], of.S[WG_SUM_TI2], n_pushed), p.oc.ti_mn, p.oc.inv_ti_rng);
                if (ch == WG_CH_YAW || !p.farm_on[ch] || n_pushed == 0) continue;     // (the farm object's yaw deque is never filled)
                const float irng = ch == WG_CH_POWER ? 1.0f / p.sc_rng_farm_power : p.oc.inv_rng[ch];
                if (p.ch[ch].current) mscratch[m++] = wg_scale_r(of.cur[ch], p.oc.mn[ch], irng);
                if (p.ch[ch].rolling_mean) mscratch[m++] = wg_scale_r(wg_sums_mean(p.oc, of.S[ch], ch, n_pushed), p.oc.mn[ch], irng);
            }
        }
        m = __shfl(m, 0, 64);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // single wave: program order suffices
        for (int i = lane; i < N * m; i += WG_WAVE) {
            const int t = i / m, k = i - t * m;
            obs_m[(size_t)t * p.obs_dim_multi + nt + k] = mscratch[k];
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
            const float irng = ch == WG_CH_POWER ? 1.0f / p.sc_rng_farm_power : p.oc.inv_rng[ch];
            if (p.ch[ch].current) { const float v = wg_scale_r(of.cur[ch], p.oc.mn[ch], irng); o[n] = v; if (o2) o2[n] = v; ++n; }
            if (p.ch[ch].rolling_mean) {
                const float v = wg_scale_r(wg_sums_mean(p.oc, of.S[ch], ch, n_pushed), p.oc.mn[ch], irng);
                o[n] = v; if (o2) o2[n] = v; ++n;
            }
        }
    }
}
