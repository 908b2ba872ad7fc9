// wg_glue_lean.h — the glue of sums-mode handles as a device function (lean_step), shared by k_glue_lean (wg_kernels.hip: one
// wave per env after the flow launch) and k_step_env (wg_env.hip: the env's own flow wave runs it as its tail, step() as ONE launch).
#pragma once
#include <hip/hip_runtime.h>

#include "wg_device.h"
#include "wg_obs.h"

__device__ inline float deque_at(const float* dq, int n_total, int maxlen, int q) {
    const int n = n_total < maxlen ? n_total : maxlen;
    return dq[(n_total - n + q) % maxlen];
}

// Write the env header back from the wave's register copy.  `env = ev` by lane 0 compiled into ~26 dependent
// single-lane stores and cost 6 us of the kernel's 28; the 11 scalar fields the step path changes are contiguous, so
// lane i stores field i: ONE coalesced store.  The generator state only changes when an episode was initialised.
// (the register copy holds ONLY these scalars: a full `WgEnv ev = env` copy has its address taken by ctx_init /
// the PCG64 helpers and therefore lived in scratch memory — every field access of the hot path was a private-memory
// round trip)
struct EnvHot {
    int live, timestep, episode, done, shadow_iters, farm_pow_n, base_pow_n, steps_done;
    float ep_return, ep_power_sum;
    int ep_len;
};
__device__ inline EnvHot env_load(const WgEnv& env) {
    EnvHot h;
    h.live = env.live; h.timestep = env.timestep; h.episode = env.episode; h.done = env.done;
    h.shadow_iters = env.shadow_iters; h.farm_pow_n = env.farm_pow_n; h.base_pow_n = env.base_pow_n;
    h.steps_done = env.steps_done; h.ep_return = env.ep_return; h.ep_power_sum = env.ep_power_sum; h.ep_len = env.ep_len;
    return h;
}
__device__ inline void env_writeback(WgEnv& env, const EnvHot& ev, const int lane, const bool skip_shadow = false) {
    // one field per lane: independent fire-and-forget stores.  (A select chain "lane i stores field i" is turned by the
    // compiler into an indexed load from a scratch copy of the struct — a private-memory round trip at the very end
    // of the kernel's latency chain.)
    if (lane == 0) env.live = ev.live;
    if (lane == 1) env.timestep = ev.timestep;
    if (lane == 2) env.episode = ev.episode;
    if (lane == 3) env.done = ev.done;
    if (lane == 4 && !skip_shadow) env.shadow_iters = ev.shadow_iters;
    if (lane == 5) env.farm_pow_n = ev.farm_pow_n;
    if (lane == 6) env.base_pow_n = ev.base_pow_n;
    if (lane == 7) env.steps_done = ev.steps_done;
    if (lane == 8) env.ep_return = ev.ep_return;
    if (lane == 9) env.ep_power_sum = ev.ep_power_sum;
    if (lane == 10) env.ep_len = ev.ep_len;
}

struct LeanArgs { WgParams p; WgPtrs d; };      // the first two kernel arguments of k_glue_lean in its kernarg segment
typedef const __attribute__((address_space(4))) LeanArgs* LeanArgsPtr;

// The common swap — prepared episode (wg_first_obs), no TI / farm-level entries, deque no longer than a wave — INLINE at the
// very end of k_glue_lean (nothing runs after it, so it adds no live range to the step path; as a call it cost the
// truncating waves ~1.5 us of call / frame set-up on the kernel's tail).  Every load is issued before the first store: one
// round trip.  (Measured and not kept: the same loads requested at the top of the kernel as LDS-DMA into a per-wave zone —
// k_glue_lean 11.4 -> 12.1 us: 9 KB of LDS per workgroup and the wait-count pass's conservatism cost more than the trip.)
template <bool MULTI, typename P, typename D>
__device__ __forceinline__ void lean_swap_fast(const P& p, const D& d, const int e, const int live, const int lane, const float fp,
                                       const float bp, const int fslot, const int bslot, const int n1f, const int n1b,
                                       const int episode_next, float* obs, float* om, const int pfn, const int pbn,
                                       const int nnp, const int time_max, const float rated) {
    const int nctx = e * 2 + (live ^ 1);
    const int F = p.F, PA = p.power_avg, obs_dim = p.obs_dim;
    float* fq = d.farm_pow + (size_t)e * PA;
    float* bq = d.base_pow + (size_t)e * PA;
    const float* pf = d.pend_farm + (size_t)nctx * PA;
    const float* pb = d.pend_base + (size_t)nctx * PA;
    const float* no = d.next_obs + (size_t)nctx * obs_dim;
    WgEnv& envw = d.env[e];
    const int mf = pfn < PA ? pfn : PA, mb = pbn < PA ? pbn : PA;
    const int n2f = n1f + mf, n2b = n1b + mb;
    const int i = lane < PA ? lane : 0;
    int qf = (i - n1f) % PA; if (qf < 0) qf += PA;
    int qb = (i - n1b) % PA; if (qb < 0) qb += PA;
    // ---- loads ----
    const float f_old = fq[i], f_new = deque_at(pf, pfn > 0 ? pfn : 1, PA, qf < mf ? qf : 0);
    const float b_old = F == 2 ? bq[i] : 0.f, b_new = F == 2 ? deque_at(pb, pbn > 0 ? pbn : 1, PA, qb < mb ? qb : 0) : 0.f;
    float ov[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ov[k] = lane + 64 * k < obs_dim ? no[lane + 64 * k] : 0.f;
    bool unready = false;
    if (lane < F) { const WgSlot& sl = d.slot[nctx * F + lane]; unready = sl.dev_remaining != 0 || sl.fill_remaining != 0; }
    const wg_u128 r_state = envw.rng_state, r_inc = envw.rng_inc;
    const uint32_t r_has32 = envw.rng_has32, r_u32 = envw.rng_u32;
    // ---- stores ----
    // (per-agent buffer of the PettingZoo facade: without farm-level entries agent t's row is its turbine block)
    const int tobs = p.turb_obs, odm = p.obs_dim_multi;
    auto put = [&](const int k, const float v) {
        obs[k] = v;
        if (MULTI && om) { const int t = k / tobs; om[(size_t)t * odm + (k - t * tobs)] = v; }
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) if (lane + 64 * k < obs_dim) put(lane + 64 * k, ov[k]);
    for (int k = lane + 256; k < obs_dim; k += WG_WAVE) put(k, no[k]);
    double sf = 0.0, sb = 0.0;
    {
        float vf = lane == fslot ? fp : f_old;
        if (lane < PA && qf < mf) { vf = f_new; fq[lane] = vf; }
        if (lane < (n2f < PA ? n2f : PA)) sf = (double)vf;
        if (F == 2) {
            float vb = lane == bslot ? bp : b_old;
            if (lane < PA && qb < mb) { vb = b_new; bq[lane] = vb; }
            if (lane < (n2b < PA ? n2b : PA)) sb = (double)vb;
        }
    }
#pragma unroll
    for (int s2 = 32; s2 > 0; s2 >>= 1) { sf += __shfl_xor(sf, s2, 64); sb += __shfl_xor(sb, s2, 64); }
    if (unready) atomicOr(d.status, WG_STATUS_BIT_STATE);
    if (lane == 0) {
        WgCtx& ncx = d.ctx[nctx];
        WgCtx& rcx = d.ctx[e * 2 + live];
        ncx.pend_farm_n = 0; ncx.pend_base_n = 0;
        d.next_obs_ok[nctx] = 0;          // consumed (the context holds a live episode now)
        rcx.init_pending = 1;                 // retire the finished context
        rcx.episode_tag = episode_next;
        rcx.snap_state = r_state; rcx.snap_inc = r_inc;
        rcx.snap_has32 = r_has32; rcx.snap_u32 = r_u32;
    }
    if (lane == 0) envw.live = live ^ 1;
    if (lane == 1) envw.timestep = 0;
    if (lane == 2) envw.steps_done = 0;
    if (lane == 3) envw.farm_pow_n = n2f;
    if (lane == 4) envw.base_pow_n = n2b;
    if (lane == 5) envw.time_max_live = time_max;
    if (lane == 6) envw.rated_live = rated;
    if (lane == 7) envw.fsum_run = sf;
    if (lane == 8) envw.bsum_run = sb;
    if (lane == 9) envw.n_pushed_live = nnp;
    if (lane == 10) envw.shadow_iters = 0;
    return;
}

// Same-step autoreset of a sums-mode env, out of line (a handful of waves per launch take it; inlined, its unrolled loads
// cost every wave of k_glue_lean its register allocation): the next episode — developed in the background — goes live.
//   * its window sums are summed afresh from its rings (nnp samples) into WgPtrs::wsum: Lg = 2^k lanes share a turbine
//     (N * Lg <= 64), lane `sub` of a group sums the window's samples sub, sub + Lg, ... with eight loads in flight, the
//     partial sums meet by shuffles.  Double sums of floats are exact for the linear slots (whoever adds them up, the result is THE sum; the sum of squares behind
//     TI rounds: wg_obs.h).  (One
//     lane per turbine walking its 25 + 10 samples one dependent load at a time made the truncating waves last 20 us.)
//   * !gen (no TI, nothing farm-level): the episode's first observation is written here, straight from the group sums;
//   * the deferred power-deque pushes of its window fill (:766, :796) are merged arithmetically — lane i decides what slot
//     i holds afterwards, stores it if it changed and contributes it to the deque's new running sum: no store -> load trip;
//   * the finished context is retired (set up again by the next k_flow launch, WgCtx::init_pending).
// The parameter blocks come from their device-resident copies.
// Called at the very END of k_glue_lean, after the wave has written the step's results and the env header as if nothing
// were swapped: nothing of the step path is live across the call (values that are get spilled where they are defined —
// scratch stores in every wave), and this function rewrites the header fields the swap changes.
template <bool MULTI, bool GEN>
__device__ __attribute__((noinline)) void lean_swap(const WgParams* gp, const WgPtrs* gd, const int e, const int live, const int lane,
                                                    const float fp, const float bp, const int fslot, const int bslot,
                                                    const int n1f, const int n1b, const int episode_next, float* obs, float* om,
                                                    float* mscr, const int pfn, const int pbn, const int nnp, const int time_max,
                                                    const float rated, const bool prepared, const LeanArgsPtr ka) {
    // (pfn .. rated: the new context's header, `prepared`: the flow kernel has prepared the episode's window sums — in
    // wsum[nctx] — and its single-agent first observation when its development completed, wg_first_obs; all fetched by the
    // caller through the scalar cache at the top of the kernel, where it already knows that the env truncates)
    const int nctx = e * 2 + (live ^ 1);
    const WgParams& p = *gp;
    const WgPtrs& d = *gd;
    const int N = p.N, F = p.F, PA = p.power_avg, NS = N + 1;
    WgCtx& ncx = d.ctx[nctx];
    if (lane < F) {
        const WgSlot& sl = d.slot[nctx * F + lane];
        if (sl.dev_remaining != 0 || sl.fill_remaining != 0) atomicOr(d.status, WG_STATUS_BIT_STATE);
    }
    // ---- power deques: what slot i holds after the deferred pushes ----
    const int mf = pfn < PA ? pfn : PA, mb = pbn < PA ? pbn : PA;
    float* fq = d.farm_pow + (size_t)e * PA;
    float* bq = d.base_pow + (size_t)e * PA;
    const float* pf = d.pend_farm + (size_t)nctx * PA;
    const float* pb = d.pend_base + (size_t)nctx * PA;
    const int n2f = n1f + mf, n2b = n1b + mb;
    double sf = 0.0, sb = 0.0;
    for (int i = lane; i < PA; i += WG_WAVE) {
        int qf = (i - n1f) % PA; if (qf < 0) qf += PA;
        float vf = i == fslot ? fp : fq[i];
        if (qf < mf) { vf = deque_at(pf, pfn, PA, qf); fq[i] = vf; }
        if (i < (n2f < PA ? n2f : PA)) sf += (double)vf;
        if (F == 2) {
            int qb = (i - n1b) % PA; if (qb < 0) qb += PA;
            float vb = i == bslot ? bp : bq[i];
            if (qb < mb) { vb = deque_at(pb, pbn, PA, qb); bq[i] = vb; }
            if (i < (n2b < PA ? n2b : PA)) sb += (double)vb;
        }
    }
    bool obs_done = false;
    if (prepared) {
        if (lane == 0) d.next_obs_ok[nctx] = 0;      // consumed (the context holds a live episode now)
        if (!MULTI && !GEN && obs) {
            const float* no = d.next_obs + (size_t)nctx * p.obs_dim;
            for (int i = lane; i < p.obs_dim; i += WG_WAVE) obs[i] = no[i];
            obs_done = true;
        }
    } else {
        // ---- window sums afresh (+ the first observation when nothing generic is in it) ----
        int Lg = 1;
        while (Lg < 8 && N * (Lg * 2) <= WG_WAVE) Lg *= 2;
        const int sub = lane & (Lg - 1), per_pass = WG_WAVE / Lg;
        const bool farm_ent = GEN && p.sum_mask_f != 0u;
        for (int t0 = 0; t0 < N + (farm_ent ? 1 : 0); t0 += per_pass) {
            const int ent = t0 + lane / Lg;
            const bool have = ent < N || (ent == N && farm_ent);
            const SumsEnt q = wg_sums_ent(p, d, nctx, have ? ent : 0);
            double* ws_ = d.wsum + (size_t)nctx * WG_N_SUMS * NS + (have ? ent : 0);
            float* o = (obs && !GEN && have) ? obs + (size_t)ent * p.turb_obs : nullptr;
            float* o_m = (MULTI && om && !GEN && have) ? om + (size_t)ent * p.obs_dim_multi : nullptr;
            int n = 0;
#pragma nounroll
            for (int sl = 0; sl < (GEN ? WG_N_SUMS : WG_N_CH); ++sl) {
                const bool on = have && ((q.sm >> sl) & 1u);
                const int ch = sl < WG_N_CH ? sl : WG_CH_WS;
                const int off = q.farm ? p.fring_off[ch] : p.ring_off[ch];
                const int cap = p.ring_cap[ch];
                const int cnt = p.sum_w[sl] < nnp ? p.sum_w[sl] : nnp;
                const int r0 = (nnp - cnt) % cap;
                const bool cur_on = o && ((p.oc.cur_mask >> ch) & 1u) && nnp > 0;
                const float cv = cur_on ? q.rb[off + ((nnp - 1) % cap) * q.stride] : 0.f;
                double acc = 0.0;
                if (on) {
#pragma nounroll
                    for (int k = sub; k < cnt; k += 8 * Lg) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            int row = r0 + min(k + u * Lg, cnt - 1); if (row >= cap) row -= cap;
                            v[u] = q.rb[off + row * q.stride];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (k + u * Lg < cnt) acc += sl == WG_SUM_TI2 ? (double)v[u] * (double)v[u] : (double)v[u];
                    }
                }
                for (int s2 = 1; s2 < Lg; s2 <<= 1) acc += __shfl_xor(acc, s2, 64);
                if (on && sub == 0) ws_[(size_t)sl * NS] = acc;
                // first observation, this channel's entries (wg_obs_turbine<false>: `current`, then the rolling mean)
                if (o && sub == 0 && nnp > 0) {
                    if (cur_on) { const float v = wg_scale_r(cv, p.oc.mn[ch], p.oc.inv_rng[ch]); o[n] = v; if (o_m) o_m[n] = v; ++n; }
                    if ((p.oc.rol_mask >> ch) & 1u) {
                        const float v = wg_scale_r(wg_sums_mean(p.oc, acc, ch, nnp), p.oc.mn[ch], p.oc.inv_rng[ch]);
                        o[n] = v; if (o_m) o_m[n] = v; ++n;
                    }
                }
            }
        }
        obs_done = !GEN;
    }
#pragma unroll
    for (int s2 = 32; s2 > 0; s2 >>= 1) { sf += __shfl_xor(sf, s2, 64); sb += __shfl_xor(sb, s2, 64); }
    if (lane == 0) {
        ncx.pend_farm_n = 0; ncx.pend_base_n = 0;
        // retire the finished context
        WgCtx& rcx = d.ctx[e * 2 + live];
        const WgEnv& env = d.env[e];
        rcx.init_pending = 1;
        rcx.episode_tag = episode_next;
        rcx.snap_state = env.rng_state; rcx.snap_inc = env.rng_inc;
        rcx.snap_has32 = env.rng_has32; rcx.snap_u32 = env.rng_u32;
    }
    // the env header of the new live episode (the caller has written the finished one's)
    WgEnv& envw = d.env[e];
    if (lane == 0) envw.live = live ^ 1;
    if (lane == 1) envw.timestep = 0;
    if (lane == 2) envw.steps_done = 0;
    if (lane == 3) envw.farm_pow_n = n2f;
    if (lane == 4) envw.base_pow_n = n2b;
    if (lane == 5) envw.time_max_live = time_max;
    if (lane == 6) envw.rated_live = rated;
    if (lane == 7) envw.fsum_run = sf;
    if (lane == 8) envw.bsum_run = sb;
    if (lane == 9) envw.n_pushed_live = nnp;
    if (lane == 10) envw.shadow_iters = 0;       // (the initialising workgroups of the next k_flow launch plan their own first share)
    if (!obs_done && obs) {
        // first observation from the stored sums (generic layouts; per-agent buffer with prepared sums)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have landed before it reads the sums back
        auto ld = [&](const int ent) {
            ObsIn o;
            const SumsEnt q = wg_sums_ent(p, d, nctx, ent);
            const double* ws_ = d.wsum + (size_t)nctx * WG_N_SUMS * NS + ent;
#pragma unroll
            for (int sl = 0; sl < WG_N_SUMS; ++sl) o.S[sl] = ((q.sm >> sl) & 1u) ? __builtin_nontemporal_load(ws_ + (size_t)sl * NS) : 0.0;
#pragma unroll
            for (int ch = 0; ch < WG_N_CH; ++ch) {
                const int off = q.farm ? p.fring_off[ch] : p.ring_off[ch];
                o.cur[ch] = (nnp > 0 && ((q.cm >> ch) & 1u)) ? q.rb[off + ((nnp - 1) % p.ring_cap[ch]) * q.stride] : 0.f;
            }
            return o;
        };
        build_obs_sums<MULTI, GEN>(p, lane, obs, nullptr, om, nnp, mscr, ld(lane < N ? lane : 0), ld);
    }
}


#ifndef WG_LEAN_ABLATE
#define WG_LEAN_ABLATE 0      // profiling builds: return from k_glue_lean after phase n (tools/glue_ablate.sh)
#endif
// Loads of words this kernel (or, fused, this launch) also WRITES — the env header, the power deques, the contexts' flags — go
// through the GLOBAL address space, never the constant one: memory behind address_space(4) is "never written" to the compiler
// (it may sink such a load below the store or re-execute it there) and is served by the scalar cache, which is not coherent
// with this launch's vector stores (round 5 papered over both with an opaque register pin).  Wave-uniform
// values are moved to scalar registers with v_readlane / v_readfirstlane instead, so the arithmetic on them stays on the
// scalar unit.  What the fused step kernel (k_flow_env / k_flow_envb) hands over in LeanFused are values its flow part
// holds in LDS anyway.
__device__ __forceinline__ int wg_uni(const int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float wg_uni(const float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

struct LeanFused {
    float fp, bp;             // farm power of the step: agent farm, baseline farm (WgPtrs::step_farm_pow / step_base_pow)
    int work;                 // remaining development work of the background episode: max over its farms of dev + K * fill
    int bg_init_pending;      // the background context's set-up flag was pending at the head of this launch
    int plan_elsewhere;       // the background context's own wave plans its next share (WgEnv::shadow_iters) unless the env truncates
    int hw;                   // word (lane & 31) of the env's 128-byte header as the flow part's prologue loaded it: nothing this wave
                              // reads of it has changed since (the header is rewritten only below) — the glue needs no second trip for it
    // k_flow_env with a pass wave: NO load left on the glue's common path.  What the step produced comes in the flow part's
    // registers (the lane's pushed samples, yaw before / after, power: own and baseline turbine), what it did not touch — old
    // window sums, leaving samples, oldest deque entries, metrics — was fetched by the pass wave while the step ran and waits in
    // LDS (`pre`, layout LEAN_PRE_*; nullptr: the glue loads everything itself).
    const float* pre;
    float nw[WG_N_CH], yaw, old_yaw, pw, pwb;
};
// LDS zone of the pre-fetched glue inputs (floats; lane = the glue's lane): S[s] as doubles | lv[s] | metrics | deque entries | flag
#define LEAN_PRE_S(s, lane) ((s) * 64 + (lane))                    // index into (const double*)pre
#define LEAN_PRE_LV(s, lane) (512 + (s) * 64 + (lane))
#define LEAN_PRE_MET(lane) (768 + (lane))
#define LEAN_PRE_FOLD 832
#define LEAN_PRE_BOLD 833
#define LEAN_PRE_FLAG 834
#define LEAN_PRE_BYTES 3584

// One env's glue after its flow step: power deques, window sums -> observation, reward, penalty, truncation, metrics, the
// background episode's next share, and — at truncation — the swap.  One wave; `lane` 0 .. 63.
// GEN = false: no TI and nothing farm-level in the observation — the instantiation of the shipped sensor sets.
// FUSED: called by the env's own flow wave at the end of k_step_env (all of that wave's stores have been waited for).
template <bool MULTI, bool GEN, bool FUSED>
__device__ __forceinline__ void lean_step(const WgParams& p, const WgPtrs& d, const WgParams* gp, const WgPtrs* gd, const int e,
                                          const int lane, float* __restrict__ obs_out, float* __restrict__ reward_out,
                                          uint8_t* __restrict__ trunc_out, float* __restrict__ final_obs_out, float* const mscr,
                                          const LeanFused& fz) {
    const int N = p.N, F = p.F, PA = p.power_avg;
    WgEnv& env = d.env[e];
    float* obs = obs_out ? obs_out + (size_t)e * p.obs_dim : nullptr;
    float* om = MULTI ? d.multi_out + (size_t)e * N * p.obs_dim_multi : nullptr;
    float* fq = d.farm_pow + (size_t)e * PA;
    float* bq = d.base_pow + (size_t)e * PA;
    const int own = lane < N ? lane : 0;          // this lane's first entity (lanes >= N: a valid dummy)

    // The env's 128-byte header in ONE coalesced load: lane i (< 32) fetches 32-bit word i, the fields are broadcast with
    // v_readlane (-> scalar registers).  A plain global load: ordered against this wave's own stores to the header below.
    static_assert(sizeof(WgEnv) == 128, "lean_step loads the env header as 32 words");
    const int hw = FUSED ? fz.hw : reinterpret_cast<const int*>(&env)[lane & 31];
#define WG_HDR_I(f) __builtin_amdgcn_readlane(hw, (int)(offsetof(WgEnv, f) / 4))
#define WG_HDR_F(f) __int_as_float(WG_HDR_I(f))
#define WG_HDR_D(f) __hiloint2double(__builtin_amdgcn_readlane(hw, (int)(offsetof(WgEnv, f) / 4) + 1), WG_HDR_I(f))
    if (WG_HDR_I(done)) {
        if (lane == 0) atomicOr(d.status, WG_STATUS_BIT_STATE);
        return;
    }
    EnvHot ev;
    ev.live = WG_HDR_I(live); ev.timestep = WG_HDR_I(timestep); ev.episode = WG_HDR_I(episode); ev.done = 0; ev.shadow_iters = 0;
    ev.farm_pow_n = WG_HDR_I(farm_pow_n); ev.base_pow_n = WG_HDR_I(base_pow_n); ev.steps_done = WG_HDR_I(steps_done);
    ev.ep_return = WG_HDR_F(ep_return); ev.ep_power_sum = WG_HDR_F(ep_power_sum); ev.ep_len = WG_HDR_I(ep_len);
    int time_max = WG_HDR_I(time_max_live), n_pushed_live = WG_HDR_I(n_pushed_live);
    float rated_power = WG_HDR_F(rated_live);
    double fsum_run = WG_HDR_D(fsum_run), bsum_run = WG_HDR_D(bsum_run);
#undef WG_HDR_D
#undef WG_HDR_F
#undef WG_HDR_I
    const int live = ev.live, nxt = live ^ 1;
#if defined(WG_TIMELINE) && defined(WG_STAMP2)
    if (FUSED) WG_STAMP2(16);
#endif
    const int ctx_id = e * 2 + live;
    if (WG_LEAN_ABLATE == 1) { if (lane == 0 && reward_out) reward_out[e] = (float)(fsum_run + bsum_run) + (float)(ev.timestep + ev.episode + time_max + n_pushed_live); return; }
    // an env that truncates (known from the header alone) fetches the next context's header with everything else
    int nx_pfn = 0, nx_pbn = 0, nx_np = 0, nx_tmax = 0;
    float nx_rated = 0.f;
    bool nx_prep = false;
    if (ev.timestep >= time_max && p.autoreset) {      // (rare: plain loads with a uniform address)
        const WgCtx& nc = d.ctx[e * 2 + nxt];
        nx_pfn = wg_uni(nc.pend_farm_n); nx_pbn = wg_uni(nc.pend_base_n); nx_np = wg_uni(nc.n_pushed); nx_tmax = wg_uni(nc.time_max);
        nx_rated = wg_uni(nc.rated_power);
        nx_prep = d.next_obs_ok != nullptr && wg_uni(d.next_obs_ok[e * 2 + nxt]) != 0;
    }
    // ---- every load of the step, issued together ----
    const bool pre = FUSED && !GEN && fz.pre != nullptr;
    SumsRaw raw;
    // (farms with more than 64 turbines, cfg3: the lane's SECOND turbine is requested with the first — the observation loop
    // below would otherwise take a second memory round trip for it)
    const bool have2 = !FUSED && lane + WG_WAVE < N;
    SumsRaw raw2 = {};
    const size_t tb_a = (size_t)(ctx_id * F) * N;
    float l_yaw = 0.f, l_old = 0.f, l_pow = 0.f, l_powb = 0.f;
    const float fp = FUSED ? fz.fp : wg_uni(d.step_farm_pow[e]);
    const float bp = F == 2 ? (FUSED ? fz.bp : wg_uni(d.step_base_pow[e])) : 0.f;
    const int fslot = ev.farm_pow_n % PA, bslot = ev.base_pow_n % PA;      // (counts run over all episodes)
    float* met = d.metrics + (size_t)e * WG_N_METRICS;
    float f_old, b_old, l_met;
    if (pre) {
        // (the same values the loads below would return: the registers hold what this wave has just stored, the LDS zone what
        // nothing in this launch writes before the glue does)
        const double* const zs = reinterpret_cast<const double*>(fz.pre);
        const unsigned msk = p.sum_mask_t | p.cur_mask_t;
#pragma unroll
        for (int s = 0; s < WG_N_SUMS; ++s) { raw.S[s] = 0.0; raw.lv[s] = 0.f; }
#pragma unroll
        for (int s = 0; s < WG_N_CH; ++s) {
            if ((p.sum_mask_t >> s) & 1u) { raw.S[s] = zs[LEAN_PRE_S(s, lane)]; raw.lv[s] = fz.pre[LEAN_PRE_LV(s, lane)]; }
        }
#pragma unroll
        for (int ch = 0; ch < WG_N_CH; ++ch) raw.nw[ch] = ((msk >> ch) & 1u) ? fz.nw[ch] : 0.f;
        if (lane < N) { l_yaw = fz.yaw; l_old = fz.old_yaw; l_pow = fz.pw; if (F == 2) l_powb = fz.pwb; }
        f_old = wg_uni(fz.pre[LEAN_PRE_FOLD]); b_old = F == 2 ? wg_uni(fz.pre[LEAN_PRE_BOLD]) : 0.f;
        l_met = lane < WG_N_METRICS ? fz.pre[LEAN_PRE_MET(lane)] : 0.f;
    } else {
        raw = wg_sums_load<GEN>(p, d, e, ctx_id, own, n_pushed_live);
        if (have2) raw2 = wg_sums_load<GEN>(p, d, e, ctx_id, lane + WG_WAVE, n_pushed_live);
        if (lane < N) {
            l_yaw = d.yaw[tb_a + lane]; l_old = d.old_yaw[(size_t)e * N + lane]; l_pow = d.power[tb_a + lane];
            if (F == 2) l_powb = d.power[tb_a + N + lane];
        }
        f_old = wg_uni(fq[fslot]); b_old = F == 2 ? wg_uni(bq[bslot]) : 0.f;
        l_met = lane < WG_N_METRICS ? met[lane] : 0.f;
    }
    // background episode: remaining work of its farms (plan of the next step's share), pending set-up flag
    int work = 0;
    if (p.autoreset) {
        if (FUSED) {
            work = fz.work;
            if (fz.bg_init_pending && lane == 0) d.ctx[e * 2 + nxt].init_pending = 0;
        } else {
            const WgSlot* const bs = d.slot + (size_t)(e * 2 + nxt) * F;
            for (int f = 0; f < F; ++f) work = max(work, wg_uni(bs[f].dev_remaining) + p.K * wg_uni(bs[f].fill_remaining));
            if (wg_uni(d.ctx[e * 2 + nxt].init_pending) && lane == 0) d.ctx[e * 2 + nxt].init_pending = 0;
        }
    }
#if defined(WG_TIMELINE) && defined(WG_STAMP2)
    if (FUSED) { __builtin_amdgcn_s_waitcnt(0); WG_STAMP2(17); }
#endif
    if (WG_LEAN_ABLATE == 2) {
        float acc = raw.nw[0] + raw.nw[2] + raw.lv[0] + raw.lv[2] + (float)raw.S[0] + (float)raw.S[2] + l_yaw + l_old + l_pow + l_powb + fp + bp + f_old + b_old + l_met;
        if (reward_out) reward_out[e] = acc + (float)work;
        return;
    }

    // power deques (:975-981): the new sample replaces the oldest one once the deque is full
    const bool f_full = ev.farm_pow_n >= PA, b_full = ev.base_pow_n >= PA;
    fsum_run += (double)fp - (f_full ? (double)f_old : 0.0);
    if (F == 2) bsum_run += (double)bp - (b_full ? (double)b_old : 0.0);
    ev.farm_pow_n++;
    if (F == 2) ev.base_pow_n++;
    const int nf = ev.farm_pow_n < PA ? ev.farm_pow_n : PA, nb = ev.base_pow_n < PA ? ev.base_pow_n : PA;
    if (lane == 0) {
        fq[fslot] = fp;
        if (F == 2) bq[bslot] = bp;
        if (fp != fp) atomicOr(d.status, WG_STATUS_BIT_NAN_POWER);
    }
    const int truncated = ev.timestep >= time_max;                                                        // :1003
    const bool swap_obs = truncated && p.autoreset;
    float* fin = final_obs_out ? final_obs_out + (size_t)e * p.obs_dim : nullptr;
    const int np_step = n_pushed_live;            // pushes the window sums account for before this step's update
    n_pushed_live += 1;
    if (WG_LEAN_ABLATE == 3) {
        float acc = raw.nw[0] + raw.nw[2] + raw.lv[0] + raw.lv[2] + (float)raw.S[0] + (float)raw.S[2] + l_yaw + l_old + l_pow + l_powb + l_met;
        if (reward_out) reward_out[e] = acc + (float)work + (float)(fsum_run + bsum_run) + (float)(nf + nb);
        return;
    }

    // action penalty sums (:804-820) and current farm powers ("Power agent", :539)
    float pen_s = 0.f, pnow = 0.f, pbase = 0.f;
    if (lane < N) {
        pen_s = p.penalty_type == WG_PEN_CHANGE ? fabsf(l_old - l_yaw) : fabsf(l_yaw);
        pnow = l_pow; pbase = l_powb;
    }
    for (int t = lane + WG_WAVE; t < N; t += WG_WAVE) {      // farms with more than 64 turbines
        const float y = d.yaw[tb_a + t];
        pen_s += p.penalty_type == WG_PEN_CHANGE ? fabsf(d.old_yaw[(size_t)e * N + t] - y) : fabsf(y);
        pnow += d.power[tb_a + t];
        if (F == 2) pbase += d.power[tb_a + N + t];
    }
    if (p.action_penalty >= 0.001) pen_s = N <= 16 ? wg_row_sum(pen_s) : wg_wave_sum(pen_s);      // (lanes >= N hold 0)
    pnow = N <= 16 ? wg_row_sum(pnow) : wg_wave_sum(pnow);
    if (F == 2) pbase = N <= 16 ? wg_row_sum(pbase) : wg_wave_sum(pbase);
    pen_s = __shfl(pen_s, 0, 64); pnow = __shfl(pnow, 0, 64); pbase = __shfl(pbase, 0, 64);

    double pr = 0.0;
    switch (p.reward_mode) {
    case WG_REW_BASELINE: pr = (fsum_run * (double)nb) / (bsum_run * (double)nf) - 1.0; break;              // :882-891: mean / mean - 1
    case WG_REW_POWER_AVG: pr = fsum_run / ((double)nf * (double)N * (double)rated_power); break;           // :896-897
    case WG_REW_NONE: pr = 0.0; break;
    case WG_REW_POWER_DIFF: {                                                                              // :904-918
        // mean of the newest tenth minus mean of the oldest tenth of the deque: the one mode that reads the deque
        const int wsz = PA / 10;
        double fl = 0.0, fo = 0.0;
        int nl = 0, no = 0;
        for (int i = lane; i < PA; i += WG_WAVE) {
            const float fv = i == fslot ? fp : fq[i];
            if (i < nf) {
                int q = i - (ev.farm_pow_n - nf) % PA; if (q < 0) q += PA;      // logical index (0 = oldest) of slot i
                if (q >= PA - wsz && q < PA) { fl += (double)fv; nl++; }
                if (q < wsz) { fo += (double)fv; no++; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            fl += __shfl_xor(fl, o, 64); fo += __shfl_xor(fo, o, 64);
            nl += __shfl_xor(nl, o, 64); no += __shfl_xor(no, o, 64);
        }
        pr = (fl / (double)nl - fo / (double)no) / N;
        break;
    }
    }
    double pen = 0.0;
    if (p.action_penalty >= 0.001)
        pen = p.penalty_type == WG_PEN_CHANGE ? p.action_penalty * ((double)pen_s / N) : p.action_penalty * ((double)pen_s / N / p.yaw_max_d);
    const float reward = (float)(pr * p.power_scaling + 0.0 - pen);                                       // :989-996
    ev.timestep += 1 + (p.extra_inc ? 1 : 0);                                                             // :1027
    ev.steps_done += 1;
    ev.ep_return += reward; ev.ep_power_sum += pnow; ev.ep_len += 1;
    if (lane == 0) {
        if (reward_out) reward_out[e] = reward;
        if (trunc_out) trunc_out[e] = (uint8_t)truncated;
        d.last_pow_agent[e] = pnow; d.last_pow_base[e] = pbase;
    }
    if (lane < WG_N_METRICS) {            // lane m owns metric m (recordEpisodeVals.py:31-64)
        const float trf = truncated ? 1.f : 0.f;
        float add = 0.f;
        add = lane == WG_MET_STEP_REWARD_SUM ? reward : add;
        add = lane == WG_MET_FARM_POWER_SUM ? pnow : add;
        add = lane == WG_MET_BASE_POWER_SUM ? pbase : add;
        add = lane == WG_MET_N_STEPS ? 1.f : add;
        add = lane == WG_MET_EP_RETURN_SUM ? trf * ev.ep_return : add;
        add = lane == WG_MET_EP_LENGTH_SUM ? trf * (float)ev.ep_len : add;
        add = lane == WG_MET_EP_MEAN_POWER_SUM ? (truncated ? ev.ep_power_sum / (float)ev.ep_len : 0.f) : add;
        add = lane == WG_MET_N_EPISODES ? trf : add;
        met[lane] = l_met + add;
    }
#if defined(WG_TIMELINE) && defined(WG_STAMP2)
    if (FUSED) WG_STAMP2(18);
#endif
    if (WG_LEAN_ABLATE == 4) {
        float acc = raw.nw[0] + raw.nw[2] + raw.lv[0] + raw.lv[2] + (float)raw.S[0] + (float)raw.S[2];
        if (trunc_out && acc == 123.f) trunc_out[e] = 7;
        return;
    }
    if (truncated) {
        ev.ep_return = 0.f; ev.ep_power_sum = 0.f; ev.ep_len = 0;
        ev.episode += 1;
        if (!p.autoreset) {
            ev.done = 1;
        } else {
            // same-step autoreset: the swap itself is the LAST thing the wave does (lean_swap, below)
        }
    }
    // observation (:983): the window sums of the lane's turbine advance by one sample (S += newest - leaving), then the
    // blocks; an env that truncates with same-step autoreset keeps it as final_obs only (and leaves the sums alone: the new
    // episode's replace them)
    {
        float* o1 = swap_obs ? fin : obs;
        if (o1) {
            auto get = [&](const int ent) {
                if (have2 && ent == lane + WG_WAVE) return wg_sums_apply<GEN>(p, d, ctx_id, ent, raw2, !swap_obs);
                return wg_sums_apply<GEN>(p, d, ctx_id, ent, wg_sums_load<GEN>(p, d, e, ctx_id, ent, np_step), !swap_obs);
            };
            const ObsIn oi = wg_sums_apply<GEN>(p, d, ctx_id, own, raw, lane < N && !swap_obs);
            build_obs_sums<MULTI, GEN>(p, lane, o1, swap_obs ? nullptr : fin, swap_obs ? nullptr : om, np_step + 1, mscr, oi, get);
        }
    }
#if defined(WG_TIMELINE) && defined(WG_STAMP2)
    if (FUSED) WG_STAMP2(19);
#endif
    if (WG_LEAN_ABLATE == 5) return;

    if (!p.autoreset || truncated) ev.shadow_iters = 0;      // (after a swap the initialising workgroups plan their own first share)
    else if (work == 0) ev.shadow_iters = 0;
    else {
        const int inc = 1 + (p.extra_inc ? 1 : 0);
        const long total = (long)((time_max + inc - 1) / inc) + 1;
        ev.shadow_iters = wg_shadow_share(work, total - ev.steps_done, ev.steps_done, e);
    }
    env_writeback(env, ev, lane, FUSED && fz.plan_elsewhere && !truncated);
    if (lane == 11) env.time_max_live = time_max;
    if (lane == 12) env.rated_live = rated_power;
    if (lane == 13) env.fsum_run = fsum_run;
    if (lane == 14) env.bsum_run = bsum_run;
    if (lane == 15) env.n_pushed_live = n_pushed_live;
    // same-step autoreset: the next episode — developed in the background — goes live
    if (swap_obs) {
        if (!GEN && nx_prep && obs && PA <= WG_WAVE)
            lean_swap_fast<MULTI>(p, d, e, live, lane, fp, bp, fslot, bslot, ev.farm_pow_n, ev.base_pow_n, ev.episode + 1, obs, om,
                                  nx_pfn, nx_pbn, nx_np, nx_tmax, nx_rated);
        else
            lean_swap<MULTI, GEN>(gp, gd, e, live, lane, fp, bp, fslot, bslot, ev.farm_pow_n, ev.base_pow_n, ev.episode + 1, obs, om, mscr,
                                  nx_pfn, nx_pbn, nx_np, nx_tmax, nx_rated, nx_prep, (LeanArgsPtr)__builtin_amdgcn_kernarg_segment_ptr());
    }
}
