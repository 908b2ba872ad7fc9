// wg_kernels.hip — HIP kernels of the batched WindGym step() transition for MI355X (gfx950, wave64).
//
//   (k_flow, the dominant kernel, lives in wg_flow.hip)
//   k_glue   one wave per env: power deques, observation (MesClass windows / TI / scaling / clip), reward,
//            penalty, truncation, same-step autoreset by swapping in the pre-developed next episode, and
//            sampling of the episode after that (Wind_Farm_Env.py:513-520, 557-568, 680-732, 804-820,
//            878-918, 980-1034; MesClass.py:70-125, 220-237, 324-351, 679-703).
//   k_init, k_info, k_obs_multi, k_metrics: reset bookkeeping, lazy info dict, PettingZoo packing
//            (WindEnvMulti.py:79-103) and the episode-metric partial sums (recordEpisodeVals.py:31-64).
#include <hip/hip_runtime.h>

#include "wg_device.h"
#include "wg_obs.h"
#include "wg_glue_lean.h"

// copy one context's sensor rings into this wave's LDS region with coalesced loads (the window loops of
// build_obs would otherwise issue long chains of dependent global loads); returns the bases to read from.
// Only what the observation reads is staged (WgParams::stage_ch, decided on the host from the sensor config): a
// channel whose windows / TI are observed is copied whole, a channel observed through its `current` value only
// contributes its newest sample, an unobserved channel (Env1.yaml: wd and power, 30 of the 65 floats per turbine)
// is not touched.  The LDS copy keeps the global layout, so build_obs indexes it unchanged.
// (RL = the rings fit the wave's LDS region: a compile-time fact of the kernel instantiation, so that the window loops'
// reads are LDS instructions — with a run-time choice between the two bases they were flat loads)
template <bool RL>
__device__ inline void stage_rings(const WgParams& p, const WgPtrs& d, int ctx_id, int lane, float* lds,
                                   const float*& rbase, const float*& fbase, const int n_pushed) {
    const float* gr = d.ring + (size_t)ctx_id * p.ring_stride;
    const float* gf = d.fring + (size_t)ctx_id * p.fring_stride;
    if (!RL) { rbase = gr; fbase = gf; return; }
    const int N = p.N;
    // LDS-DMA (global_load_lds): lane i of a request reads one float at its own global address and the 64 results land
    // at consecutive LDS words from a wave-uniform base — the LDS copy keeps the global layout, so a run of 64 ring
    // floats is one request, with no result registers, no LDS store instructions and nothing waited for until every
    // request of the turbine AND the farm rings is in flight.  (The register-staged copy held 32 VGPRs, and the farm
    // rings' requests went out only after the turbine rings had landed: a second memory round trip where they are read.)
    typedef const __attribute__((address_space(1))) void* GPtr;
    typedef __attribute__((address_space(3))) void* LPtr;
#pragma unroll
    for (int ch = 0; ch < WG_N_CH; ++ch) {
        const int off = p.ring_off[ch];
        if (p.stage_ch[ch] == 2) {
            const int len = N * p.ring_cap[ch];
#pragma nounroll
            for (int s0 = 0; s0 < len; s0 += WG_WAVE)
                if (s0 + lane < len) __builtin_amdgcn_global_load_lds((GPtr)(gr + off + s0 + lane), (LPtr)(lds + off + s0), 4, 0, 0);
        } else if (p.stage_ch[ch] == 1 && n_pushed > 0) {
            const int row = off + ((n_pushed - 1) % p.ring_cap[ch]) * N;
#pragma nounroll
            for (int s0 = 0; s0 < N; s0 += WG_WAVE)
                if (s0 + lane < N) __builtin_amdgcn_global_load_lds((GPtr)(gr + row + s0 + lane), (LPtr)(lds + row + s0), 4, 0, 0);
        }
    }
    // (the farm rings are staged only if something reads them: farm-level observations, the per-agent blocks of the
    // PettingZoo facade — Env1.yaml observes turbines only)
    const int n_farm = (p.farm_obs > 0 || d.multi_out != nullptr) ? p.fring_stride : 0;
#pragma nounroll
    for (int s0 = 0; s0 < n_farm; s0 += WG_WAVE)
        if (s0 + lane < n_farm) __builtin_amdgcn_global_load_lds((GPtr)(gf + s0 + lane), (LPtr)(lds + p.ring_stride + s0), 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // single wave: program order suffices
    rbase = lds; fbase = lds + p.ring_stride;
}

__device__ inline double deque_mean(const float* dq, int n_total, int maxlen) {
    const int n = n_total < maxlen ? n_total : maxlen;
    double s = 0;
    for (int i = 0; i < n; ++i) s += (double)dq[i];
    return s / (double)n;
}
// plan how many flow sub-steps the background episode must advance during the next step() so that it is
// ready exactly when the running episode truncates
__device__ inline int plan_shadow(const WgParams& p, const WgPtrs& d, const int live, const int steps_done, int e) {
    const int sh = live ^ 1;
    int work = 0;
    for (int f = 0; f < p.F; ++f) {
        const WgSlot& s = d.slot[(e * 2 + sh) * p.F + f];
        int w = s.dev_remaining + p.K * s.fill_remaining;
        if (w > work) work = w;
    }
    if (work == 0) return 0;
    const int inc = 1 + (p.extra_inc ? 1 : 0);
    const int tm = d.ctx[e * 2 + live].time_max;
    const long total = (long)((tm + inc - 1) / inc) + 1;
    return wg_shadow_share(work, total - steps_done, steps_done, e);
}

// ===================================================================================================
// k_glue: one wave per env.  phase 0 = after a flow step (step()); phase 1 = end of reset().
// ===================================================================================================
// (4 waves per SIMD = 16 per CU: at 4096 envs per GPU every env's wave is resident at once)
// MULTI: the per-agent observation buffer of the PettingZoo facade is written too (wg_set_obs_multi_buffer); the
// single-agent instantiation carries none of that code (it set the register pressure of all three build_obs copies).
// PRE: the handle's flow kernel builds a background episode's first observation when its development completes
// (WgPtrs::next_obs, single-wave steady variant): the swap copies it.  A template parameter so that the other
// handles' instantiations are the code they were.
template <bool MULTI, bool RL, int L, bool PRE>
__global__ void __launch_bounds__(WG_BLOCK, 4)
k_glue(const WgParams p, const WgPtrs d, const int phase, const uint8_t* __restrict__ mask,
       float* __restrict__ obs_out, float* __restrict__ reward_out, uint8_t* __restrict__ trunc_out,
       float* __restrict__ final_obs_out, const int lds_floats_per_wave, const int ring_floats) {
    extern __shared__ __attribute__((aligned(16))) float glue_lds[];
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * WG_NWAVES + (threadIdx.x >> 6);
    if (e >= p.B) return;
    WgEnv& env = d.env[e];
    const int N = p.N;
    float* obs = obs_out ? obs_out + (size_t)e * p.obs_dim : nullptr;
    // this wave's LDS: [ring_floats: staged sensor rings, when they fit][MULTI: farm block scratch, p.farm_obs floats]
    float* const wave_lds = glue_lds + (size_t)(threadIdx.x >> 6) * lds_floats_per_wave;
    float* my_lds = wave_lds;      // (staging region; used when RL)
    float* const mscr = MULTI ? wave_lds + ring_floats : nullptr;
    const float *rbase, *fbase;

    if (phase == 1) {
        if (mask && !mask[e]) return;
        // the freshly developed episode goes live: flush its deferred power-deque pushes (:766, :796)
        const int ctx_id = e * 2 + env.live;
        WgCtx& cx = d.ctx[ctx_id];
        if (lane < p.F) {     // wg_reset develops until every slot is ready; anything else is a bug
            const WgSlot& sl = d.slot[ctx_id * p.F + lane];
            if (sl.dev_remaining != 0 || sl.fill_remaining != 0) atomicOr(d.status, WG_STATUS_BIT_STATE);
        }
        if (lane == 0) {
            const int nf = cx.pend_farm_n < p.power_avg ? cx.pend_farm_n : p.power_avg;
            for (int q = 0; q < nf; ++q) {
                float v = deque_at(d.pend_farm + (size_t)ctx_id * p.power_avg, cx.pend_farm_n, p.power_avg, q);
                d.farm_pow[(size_t)e * p.power_avg + env.farm_pow_n % p.power_avg] = v;
                env.farm_pow_n++;
            }
            const int nb = cx.pend_base_n < p.power_avg ? cx.pend_base_n : p.power_avg;
            for (int q = 0; q < nb; ++q) {
                float v = deque_at(d.pend_base + (size_t)ctx_id * p.power_avg, cx.pend_base_n, p.power_avg, q);
                d.base_pow[(size_t)e * p.power_avg + env.base_pow_n % p.power_avg] = v;
                env.base_pow_n++;
            }
            cx.pend_farm_n = 0; cx.pend_base_n = 0;
            env.timestep = 0; env.done = 0; env.steps_done = 0;
            env.ep_return = 0.f; env.ep_power_sum = 0.f; env.ep_len = 0;
            env.shadow_iters = p.autoreset ? plan_shadow(p, d, env.live, env.steps_done, e) : 0;
        }
        if (obs) {
            const int np1 = cx.n_pushed;
            stage_rings<RL>(p, d, ctx_id, lane, my_lds, rbase, fbase, np1);
            build_obs<L>(p, d, ctx_id, lane, obs, nullptr, rbase, fbase, false,
                      MULTI ? d.multi_out + (size_t)e * N * p.obs_dim_multi : nullptr, np1, mscr);
        }
        return;
    }

    if (env.done) {
        if (lane == 0) atomicOr(d.status, WG_STATUS_BIT_STATE);
        return;
    }
    // Work on a register copy of the env header: through the reference every field access is a dependent
    // global round trip (the kernel was a ~25-deep latency chain).  All lanes apply the same scalar updates to
    // their copies; lane 0 alone consumes random numbers and writes the header back.
#ifndef WG_GLUE_ABLATE
#define WG_GLUE_ABLATE 0
#endif
    // debug build (-DWG_GLUE_TL, tools/glue_timeline.py): shader-clock stamps at the phase boundaries of this wave, left in
    // the env's final_obs row (which the build therefore destroys) — how the truncating waves' share of the kernel
    // was measured
#ifdef WG_GLUE_TL
    unsigned tl[8];
    const unsigned tl_w0 = (unsigned)wall_clock64();       // (100 MHz, one counter for the whole device)
#define WG_GSTAMP(k) tl[k] = (unsigned)clock64()
#else
#define WG_GSTAMP(k) do { } while (0)
#endif
    WG_GSTAMP(0);
    if (WG_GLUE_ABLATE == 1) return;
    EnvHot ev = env_load(env);
    const int live = ev.live;
    const int ctx_id = e * 2 + live;
    WgCtx& cx = d.ctx[ctx_id];
    const int time_max = cx.time_max;
    const float rated_power = cx.rated_power;
    const int n_pushed_live = cx.n_pushed;
    const size_t tb_a = (size_t)(ctx_id * p.F) * N;
    const float fp = d.step_farm_pow[e];
    const float bp = p.F == 2 ? d.step_base_pow[e] : 0.f;
    float* fq = d.farm_pow + (size_t)e * p.power_avg;
    float* bq = d.base_pow + (size_t)e * p.power_avg;

    // ---- issue every remaining global load now: they all depend only on (e, live), so they share ONE memory
    // round trip with the ring staging below instead of forming a chain of dependent round trips ----
    float l_yaw = 0.f, l_old = 0.f, l_pow = 0.f, l_powb = 0.f;
    if (lane < N) {
        l_yaw = d.yaw[tb_a + lane]; l_old = d.old_yaw[(size_t)e * N + lane]; l_pow = d.power[tb_a + lane];
        if (p.F == 2) l_powb = d.power[tb_a + N + lane];
    }
    float* met = d.metrics + (size_t)e * WG_N_METRICS;
    float l_met = lane < WG_N_METRICS ? met[lane] : 0.f;
    // (first 64 elements of the power deques: requested with the loads above, consumed by the deque loop below)
    const float pf_f = lane < p.power_avg ? fq[lane] : 0.f;
    const float pf_b = (p.F == 2 && lane < p.power_avg) ? bq[lane] : 0.f;
    // a background context flagged at the previous truncation has been initialised by the k_flow launch of this step
    if (p.autoreset && lane == 0) {
        WgCtx& bcx = d.ctx[e * 2 + (live ^ 1)];
        if (bcx.init_pending) bcx.init_pending = 0;
    }
    int l_work = 0;                       // background-episode work of farm `lane` (plan_shadow)
    if (p.autoreset && lane < p.F) {
        const WgSlot& sl = d.slot[(e * 2 + (live ^ 1)) * p.F + lane];
        l_work = sl.dev_remaining + p.K * sl.fill_remaining;
    }

    // power deques (:975-981): one lane per deque element, the new value patched in registers
    const int fslot = ev.farm_pow_n % p.power_avg, bslot = ev.base_pow_n % p.power_avg;   // (counts run over all episodes)
    ev.farm_pow_n++;
    if (p.F == 2) ev.base_pow_n++;
    double fsum = 0.0, bsum = 0.0, fl = 0.0, fo = 0.0;
    int nl = 0, no = 0;
    {
        const int nf = ev.farm_pow_n < p.power_avg ? ev.farm_pow_n : p.power_avg;
        const int nb = ev.base_pow_n < p.power_avg ? ev.base_pow_n : p.power_avg;
        const int wsz = p.power_avg / 10;
        for (int i = lane; i < p.power_avg; i += WG_WAVE) {
            const float fv = i == fslot ? fp : (i == lane ? pf_f : fq[i]);
            if (i < nf) fsum += (double)fv;
            if (p.F == 2) {
                const float bv = i == bslot ? bp : (i == lane ? pf_b : bq[i]);
                if (i < nb) bsum += (double)bv;
            }
            if (p.reward_mode == WG_REW_POWER_DIFF && i < nf) {
                // logical index (0 = oldest) of physical slot i in the ring
                int q = i - (ev.farm_pow_n - nf) % p.power_avg; if (q < 0) q += p.power_avg;
                if (q >= p.power_avg - wsz && q < p.power_avg) { fl += (double)fv; nl++; }
                if (q < wsz) { fo += (double)fv; no++; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { fsum += __shfl_xor(fsum, o, 64); bsum += __shfl_xor(bsum, o, 64); }
        if (p.reward_mode == WG_REW_POWER_DIFF) {       // (uniform: the other modes skip 36 cross-lane moves)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                fl += __shfl_xor(fl, o, 64); fo += __shfl_xor(fo, o, 64);
                nl += __shfl_xor(nl, o, 64); no += __shfl_xor(no, o, 64);
            }
        }
        fsum /= (double)nf; bsum /= (double)nb;
    }
    if (lane == 0) {
        fq[fslot] = fp;
        if (p.F == 2) bq[bslot] = bp;
        if (fp != fp) atomicOr(d.status, WG_STATUS_BIT_NAN_POWER);
    }
    // observation (:983)
    float* fin = final_obs_out ? final_obs_out + (size_t)e * p.obs_dim : nullptr;
    if (WG_GLUE_ABLATE == 2) return;
    // An env that truncates with same-step autoreset returns the NEXT episode's first observation (built below,
    // after the swap); the finished episode's last observation is only wanted as final_obs.  (Building it into `obs`
    // first and overwriting it made the truncating waves — about 9 per launch of 4096 — the kernel's tail.)
    WG_GSTAMP(1);
    const int truncated = ev.timestep >= time_max;                                    // :1003
    const bool swap_obs = truncated && p.autoreset && obs != nullptr;
    // (A truncating wave builds two observations — final_obs of the finished episode, the first one of the next — and
    // swaps the contexts: it ends ~5 us after the others and IS the end of the kernel, tools/glue_timeline.py.  Measured
    // and not kept: requesting the next context's header, deferred deque entries and rings with the first staging
    // (swap block 12.7 k -> 7.7 k cycles, but the common path slowed by as much as the tail gained); s_setprio(3) for
    // the truncating wave: no effect.)
    if (!swap_obs || fin) {
        stage_rings<RL>(p, d, ctx_id, lane, my_lds, rbase, fbase, n_pushed_live);
        if (WG_GLUE_ABLATE == 3) return;
        if (swap_obs) build_obs<L>(p, d, ctx_id, lane, fin, nullptr, rbase, fbase, false, nullptr, n_pushed_live);
        else build_obs<L>(p, d, ctx_id, lane, obs, fin, rbase, fbase, false,
                       MULTI ? d.multi_out + (size_t)e * N * p.obs_dim_multi : nullptr, n_pushed_live, mscr);
    }
    if (WG_GLUE_ABLATE == 4) return;
    WG_GSTAMP(2);

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
        if (p.F == 2) pbase += d.power[tb_a + N + t];
    }
    pen_s = wg_wave_sum(pen_s); pnow = wg_wave_sum(pnow); pbase = wg_wave_sum(pbase);

    double pr = 0.0;
    switch (p.reward_mode) {
    case WG_REW_BASELINE: pr = fsum / bsum - 1.0; break;                              // :882-891
    case WG_REW_POWER_AVG: pr = fsum / N / (double)rated_power; break;                // :896-897
    case WG_REW_NONE: pr = 0.0; break;
    case WG_REW_POWER_DIFF: pr = (fl / (double)nl - fo / (double)no) / N; break;      // :904-918
    }
    double pen = 0.0;
    if (p.action_penalty >= 0.001) {
        pen = p.penalty_type == WG_PEN_CHANGE ? p.action_penalty * ((double)pen_s / N)
                                              : p.action_penalty * ((double)pen_s / N / p.yaw_max_d);
    }
    const float reward = (float)(pr * p.power_scaling + 0.0 - pen);                   // :989-996
    ev.timestep += 1 + (p.extra_inc ? 1 : 0);                                         // :1027
    ev.steps_done += 1;
    // episode metrics (recordEpisodeVals.py:31-64; longer_steps_example.py:39-124)
    ev.ep_return += reward; ev.ep_power_sum += pnow; ev.ep_len += 1;
    if (lane == 0) {
        if (reward_out) reward_out[e] = reward;
        if (trunc_out) trunc_out[e] = (uint8_t)truncated;
        d.last_pow_agent[e] = pnow; d.last_pow_base[e] = pbase;
    }
    if (WG_GLUE_ABLATE == 5) return;
    if (lane < WG_N_METRICS) {            // lane m owns metric m (select chain: a switch became a scratch table)
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
    if (WG_GLUE_ABLATE == 6) return;
    WG_GSTAMP(3);
    if (truncated) {
        ev.ep_return = 0.f; ev.ep_power_sum = 0.f; ev.ep_len = 0;
        ev.episode += 1;
        if (!p.autoreset) {
            ev.done = 1;
            env_writeback(env, ev, lane);
            return;
        }
        // same-step autoreset: the next episode was developed in the background; make it live
        const int nxt = live ^ 1;
        const int nctx = e * 2 + nxt;
        WgCtx& ncx = d.ctx[nctx];
        const int pfn = ncx.pend_farm_n, pbn = ncx.pend_base_n, nnp = ncx.n_pushed;
        const int nf = pfn < p.power_avg ? pfn : p.power_avg;
        const int nb = pbn < p.power_avg ? pbn : p.power_avg;
        // readiness of the developed episode (lane f checks farm f) and the deferred power-deque pushes of its window
        // fill (:766, :796; lane q moves entry q): independent loads, one round trip
        if (lane < p.F) {
            const WgSlot& sl = d.slot[nctx * p.F + lane];
            if (sl.dev_remaining != 0 || sl.fill_remaining != 0) atomicOr(d.status, WG_STATUS_BIT_STATE);
        }
        for (int q = lane; q < nf; q += WG_WAVE)
            fq[(ev.farm_pow_n + q) % p.power_avg] = deque_at(d.pend_farm + (size_t)nctx * p.power_avg, pfn, p.power_avg, q);
        for (int q = lane; q < nb; q += WG_WAVE)
            bq[(ev.base_pow_n + q) % p.power_avg] = deque_at(d.pend_base + (size_t)nctx * p.power_avg, pbn, p.power_avg, q);
        if (lane == 0) { ncx.pend_farm_n = 0; ncx.pend_base_n = 0; }
        ev.farm_pow_n += nf; ev.base_pow_n += nb;
        ev.live = nxt; ev.timestep = 0; ev.steps_done = 0;
        // The retired context will hold the episode after the next one.  Its initialisation (sampling from the env's
        // PCG64, layout rotation, slot set-up) is NOT done here, on this kernel's one-wave-per-env latency chain: the
        // context is flagged and the generator snapshotted; the two farm workgroups of that context run wg_ctx_init
        // at the head of the next k_flow launch (WgCtx::init_pending).
        if (lane == 0) {
            WgCtx& rcx = d.ctx[e * 2 + live];
            rcx.init_pending = 1;
            if (PRE) d.next_obs_ok[e * 2 + live] = 0;       // the retired context will hold another episode
            rcx.episode_tag = ev.episode + 1;
            rcx.snap_state = env.rng_state; rcx.snap_inc = env.rng_inc;
            rcx.snap_has32 = env.rng_has32; rcx.snap_u32 = env.rng_u32;
        }
        if (obs) {
            // the next episode's first observation: built by the k_flow workgroup that completed its development
            // (wg_first_obs) — copied; otherwise (per-agent buffer, restored state) built here
            const bool pre = PRE && d.next_obs_ok[nctx] != 0;
            if (pre) {
                const float* no = d.next_obs + (size_t)nctx * p.obs_dim;
                for (int i = lane; i < p.obs_dim; i += WG_WAVE) obs[i] = no[i];
            } else {
                stage_rings<RL>(p, d, nctx, lane, my_lds, rbase, fbase, nnp);
                build_obs<L>(p, d, nctx, lane, obs, nullptr, rbase, fbase, false,
                          MULTI ? d.multi_out + (size_t)e * N * p.obs_dim_multi : nullptr, nnp, mscr);
            }
        }
    }
    if (WG_GLUE_ABLATE == 7) return;
    WG_GSTAMP(4);
    if (!p.autoreset) {
        ev.shadow_iters = 0;
    } else if (truncated) {
        ev.shadow_iters = 0;      // the initialising workgroups of the next k_flow launch plan their own first share
    } else {
        int work = l_work;                                        // max over the farms of the background ctx
        for (int o = 1; o < 4; o <<= 1) work = max(work, __shfl_xor(work, o, 64));
        work = __shfl(work, 0, 64);
        if (work == 0) ev.shadow_iters = 0;
        else {
            const int inc = 1 + (p.extra_inc ? 1 : 0);
            const long total = (long)((time_max + inc - 1) / inc) + 1;
            ev.shadow_iters = wg_shadow_share(work, total - ev.steps_done, ev.steps_done, e);
        }
    }
    if (WG_GLUE_ABLATE == 8) return;
    env_writeback(env, ev, lane);
#ifdef WG_GLUE_TL
    WG_GSTAMP(5);
    if (lane == 0 && fin) {
        for (int k = 0; k < 6; ++k) fin[k] = __uint_as_float(tl[k]);
        fin[6] = __uint_as_float((unsigned)truncated);
        fin[7] = __uint_as_float(tl_w0); fin[8] = __uint_as_float((unsigned)wall_clock64());
    }
#endif
}


// ===================================================================================================
// k_glue_lean: the glue of sums-mode handles (wg_create: every observed rolling mean has history_N = 1).  One wave per env
// like k_glue, written for the two things that bound that kernel — 4096 resident waves share 1024 SIMDs, so its ~1000
// VALU instructions per wave WERE its 14 us, and its loads formed three dependent round trips:
//   * the env header carries what the wave needs of the live context (time_max, rated power, push count) and the power
//     deques' running sums: it is wave-uniform, fetched through the scalar cache and processed on the scalar unit;
//   * the observation comes from running window sums (WgPtrs::wsum): per turbine the newest and the leaving sample of each
//     window are read instead of the whole rings, S += newest - leaving in exact double arithmetic, mean = S / W;
//   * the deque means of the reward come from running sums too.
// A truncating env (a handful per launch) swaps contexts and sums the new episode's windows afresh from its rings.
// Reference: Wind_Farm_Env.py:975-1034 (deques, reward, penalty, truncation), :804-820, :866-918; MesClass.py:70-125.
// phase 0 = after a flow step (step()); phase 1 = end of reset().
// ===================================================================================================
// (rare paths out of line / in their own kernel: inlined they cost the step path ~130 spilled VGPRs and 1 KB of scratch
// per lane — k_glue_lean ran 89 us.  They read the parameter blocks from their device-resident copies.)

// the first observation of an episode that becomes live (end of reset, swap): its window sums are summed afresh from its
// rings (n_pushed samples) into WgPtrs::wsum, slot by slot — rolled loops, nothing held in arrays: the callee's register
// need counts against every wave of the calling kernel — and the observation is then built from the stored sums
template <bool MULTI>
__device__ __attribute__((noinline)) void lean_fresh_obs(const WgParams* gp, const WgPtrs* gd, const int e, const int ctx_id,
                                                         const int lane, const int n_pushed, float* obs, float* om, float* mscr) {
    const WgParams& p = *gp;
    const WgPtrs& d = *gd;
    const int N = p.N, NS = N + 1;
    auto fresh_store = [&](const int ent) {
        const SumsEnt q = wg_sums_ent(p, d, ctx_id, ent);
        double* ws_ = d.wsum + (size_t)ctx_id * WG_N_SUMS * NS + ent;
#pragma nounroll
        for (int s = 0; s < WG_N_SUMS; ++s) {
            if (!((q.sm >> s) & 1u)) continue;
            const int ch = s < WG_N_CH ? s : WG_CH_WS;
            const int off = q.farm ? p.fring_off[ch] : p.ring_off[ch];
            const int cap = p.ring_cap[ch];
            const int cnt = p.sum_w[s] < n_pushed ? p.sum_w[s] : n_pushed;
            int row = (n_pushed - cnt) % cap;
            double acc = 0.0;
#pragma nounroll
            for (int k = 0; k < cnt; ++k) {
                const double v = (double)q.rb[off + row * q.stride];
                acc += s == WG_SUM_TI2 ? v * v : v;
                if (++row == cap) row = 0;
            }
            ws_[(size_t)s * NS] = acc;
        }
    };
    for (int t = lane; t < N; t += WG_WAVE) fresh_store(t);
    if (lane == 0 && p.sum_mask_f) fresh_store(N);
    if (!obs) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have landed before it reads the sums back
    auto ld = [&](const int ent) {
        ObsIn o;
        const SumsEnt q = wg_sums_ent(p, d, ctx_id, ent);
        const double* ws_ = d.wsum + (size_t)ctx_id * WG_N_SUMS * NS + ent;
#pragma unroll
        for (int s = 0; s < WG_N_SUMS; ++s) o.S[s] = ((q.sm >> s) & 1u) ? __builtin_nontemporal_load(ws_ + (size_t)s * NS) : 0.0;
#pragma unroll
        for (int ch = 0; ch < WG_N_CH; ++ch) {
            const int off = q.farm ? p.fring_off[ch] : p.ring_off[ch];
            o.cur[ch] = (n_pushed > 0 && ((q.cm >> ch) & 1u)) ? q.rb[off + ((n_pushed - 1) % p.ring_cap[ch]) * q.stride] : 0.f;
        }
        return o;
    };
    build_obs_sums<MULTI, true>(p, lane, obs, nullptr, om, n_pushed, mscr, ld(lane < N ? lane : 0), ld);
}

// end of reset() for sums-mode handles (phase 1 of k_glue): the freshly developed episode goes live
template <bool MULTI>
__global__ void __launch_bounds__(WG_BLOCK)
k_glue_lean_reset(const WgParams p, const WgPtrs d, const WgParams* gp, const WgPtrs* gd, const uint8_t* __restrict__ mask,
                  float* __restrict__ obs_out, const int lds_floats_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float glue_lds[];
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * WG_NWAVES + (threadIdx.x >> 6);
    if (e >= p.B) return;
    if (mask && !mask[e]) return;
    const int N = p.N, F = p.F, PA = p.power_avg;
    float* const mscr = MULTI ? glue_lds + (size_t)(threadIdx.x >> 6) * lds_floats_per_wave : nullptr;
    WgEnv& env = d.env[e];
    float* obs = obs_out ? obs_out + (size_t)e * p.obs_dim : nullptr;
    float* om = MULTI ? d.multi_out + (size_t)e * N * p.obs_dim_multi : nullptr;
    float* fq = d.farm_pow + (size_t)e * PA;
    float* bq = d.base_pow + (size_t)e * PA;
    // flush the episode's deferred power-deque pushes (:766, :796)
    const int ctx_id = e * 2 + env.live;
    WgCtx& cx = d.ctx[ctx_id];
    const int np1 = cx.n_pushed;
    if (lane < F) {     // wg_reset develops until every slot is ready; anything else is a bug
        const WgSlot& sl = d.slot[ctx_id * F + lane];
        if (sl.dev_remaining != 0 || sl.fill_remaining != 0) atomicOr(d.status, WG_STATUS_BIT_STATE);
    }
    if (lane == 0) {
        const int nf = cx.pend_farm_n < PA ? cx.pend_farm_n : PA;
        for (int q = 0; q < nf; ++q) {
            fq[env.farm_pow_n % PA] = deque_at(d.pend_farm + (size_t)ctx_id * PA, cx.pend_farm_n, PA, q);
            env.farm_pow_n++;
        }
        const int nb = cx.pend_base_n < PA ? cx.pend_base_n : PA;
        for (int q = 0; q < nb; ++q) {
            bq[env.base_pow_n % PA] = deque_at(d.pend_base + (size_t)ctx_id * PA, cx.pend_base_n, PA, q);
            env.base_pow_n++;
        }
        cx.pend_farm_n = 0; cx.pend_base_n = 0;
        // what the step path needs of the live context, and the power deques' running sums
        env.time_max_live = cx.time_max; env.rated_live = cx.rated_power; env.n_pushed_live = np1;
        {
            const int nfq = env.farm_pow_n < PA ? env.farm_pow_n : PA, nbq = env.base_pow_n < PA ? env.base_pow_n : PA;
            double sf = 0.0, sb = 0.0;
            for (int q = 0; q < nfq; ++q) sf += (double)fq[q];
            for (int q = 0; q < nbq; ++q) sb += (double)bq[q];
            env.fsum_run = sf; env.bsum_run = sb;
        }
        env.timestep = 0; env.done = 0; env.steps_done = 0;
        env.ep_return = 0.f; env.ep_power_sum = 0.f; env.ep_len = 0;
        env.shadow_iters = p.autoreset ? plan_shadow(p, d, env.live, env.steps_done, e) : 0;
    }
    lean_fresh_obs<MULTI>(gp, gd, e, ctx_id, lane, np1, obs, om, mscr);
}

// (the body of k_glue_lean: lean_step, wg_glue_lean.h)
template <bool MULTI, bool GEN>
__global__ void __launch_bounds__(WG_BLOCK, 4)
k_glue_lean(const WgParams p, const WgPtrs d, const WgParams* gp, const WgPtrs* gd, float* __restrict__ obs_out,
            float* __restrict__ reward_out, uint8_t* __restrict__ trunc_out, float* __restrict__ final_obs_out,
            const int lds_floats_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float glue_lds[];
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * WG_NWAVES + (threadIdx.x >> 6);
    if (e >= p.B) return;
    float* const mscr = MULTI ? glue_lds + (size_t)(threadIdx.x >> 6) * lds_floats_per_wave : nullptr;
    lean_step<MULTI, GEN, false>(p, d, gp, gd, e, lane, obs_out, reward_out, trunc_out, final_obs_out, mscr, LeanFused{});
}

// ===================================================================================================
// k_init: reset() bookkeeping for the masked envs: (re)seed, sample, initialise the context(s)
// ===================================================================================================
__global__ void __launch_bounds__(WG_BLOCK)
k_init(const WgParams p, const WgPtrs d, const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seeds) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * WG_NWAVES + (threadIdx.x >> 6);
    if (e >= p.B) return;
    if (mask && !mask[e]) return;
    WgEnv& env = d.env[e];
    if (lane == 0) {
        if (seeds && seeds[e] != 0xFFFFFFFFFFFFFFFFull) {
            wg_pcg_seed(env, seeds[e]);
            env.noise_key = seeds[e];
        } else if (!env.done && env.timestep > 0) {
            // an explicit reset that abandons a running episode starts a new episode index: the sensor-noise stream is
            // keyed by (seed, episode index, push index) and must not replay the abandoned episode's sequence
            env.episode += 1;
        }
        env.done = 0;
    }
    __threadfence_block();
    const int live = env.live;
    wg_ctx_init(p, d, env, e, live, lane, env.episode, 0, p.F);
    if (p.autoreset) wg_ctx_init(p, d, env, e, live ^ 1, lane, env.episode + 1, 0, p.F);
    if (lane < 2) d.ctx[e * 2 + lane].init_pending = 0;
    if (lane < 2 && d.next_obs_ok) d.next_obs_ok[e * 2 + lane] = 0;
}

// wg_reset: number of live farm slots of the masked envs whose episode is not fully developed yet
__global__ void k_unready(const WgParams p, const WgPtrs d, const uint8_t* __restrict__ mask, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.B * p.F) return;
    const int e = i / p.F, f = i - e * p.F;
    if (mask && !mask[e]) return;
    const WgSlot& sl = d.slot[(e * 2 + d.env[e].live) * p.F + f];
    if (sl.dev_remaining != 0 || sl.fill_remaining != 0) atomicAdd(out, 1);
}
extern "C" void wg_launch_unready(const WgParams* p, const WgPtrs* d, const uint8_t* mask, int* out, hipStream_t st) {
    const int n = p->B * p->F;
    hipLaunchKernelGGL(k_unready, dim3((n + 255) / 256), dim3(256), 0, st, *p, *d, mask, out);
}

// fresh-handle initialisation: generator state of np.random.default_rng(b)
__global__ void k_create(const WgParams p, const WgPtrs d) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.B) return;
    WgEnv& env = d.env[e];
    wg_pcg_seed(env, (uint64_t)e);
    env.noise_key = (uint64_t)e;
    env.live = 0; env.timestep = 0; env.episode = 0; env.done = 1; env.shadow_iters = 0;
    env.farm_pow_n = 0; env.base_pow_n = 0; env.steps_done = 0;
    env.ep_return = 0.f; env.ep_power_sum = 0.f; env.ep_len = 0;
}

// ===================================================================================================
// k_obs_multi: WindFarmEnvMulti._get_obs_multi (WindEnvMulti.py:79-103): [B, N, obs_dim_multi]
// ===================================================================================================
__global__ void k_obs_multi(const WgParams p, const WgPtrs d, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.B * p.N) return;
    const int e = i / p.N, t = i - e * p.N;
    const int ctx_id = e * 2 + d.env[e].live;
    const int n_pushed = d.ctx[ctx_id].n_pushed;
    float* o = out + (size_t)i * p.obs_dim_multi;
    float buf[64];
    // blocks can be long (history_N); write through a bounded staging buffer in pieces
    int n = 0;
    if (p.turb_obs <= 64) {
        int m = wg_turb_block(p, d, ctx_id, n_pushed, t, false, buf);
        for (int k = 0; k < m; ++k) o[n++] = wg_clip1(buf[k]);
    } else {
        // long blocks: reuse the single-agent observation of this turbine is not available here; recompute
        // directly into the output (values are scaled; clip in place afterwards)
        int m = wg_turb_block(p, d, ctx_id, n_pushed, t, false, o);
        for (int k = 0; k < m; ++k) o[k] = wg_clip1(o[k]);
        n = m;
    }
    if (p.farm_obs <= 64) {
        int m = wg_turb_block(p, d, ctx_id, n_pushed, 0, true, buf);
        for (int k = 0; k < m; ++k) o[n++] = wg_clip1(buf[k]);
    } else {
        int m = wg_turb_block(p, d, ctx_id, n_pushed, 0, true, o + n);
        for (int k = 0; k < m; ++k) o[n + k] = wg_clip1(o[n + k]);
    }
}

// ===================================================================================================
// k_info: lazy info dict (Wind_Farm_Env.py:522-555)
// ===================================================================================================
__global__ void k_info(const WgParams p, const WgPtrs d, const int field, void* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = p.N;
    const int per = (field == WG_INFO_ROTOR_UVW_AGENT || field == WG_INFO_ROTOR_UVW_BASE) ? N
                    : (field == WG_INFO_YAW_AGENT || field == WG_INFO_YAW_BASE || field == WG_INFO_WS_TURB ||
                       field == WG_INFO_WD_TURB || field == WG_INFO_POWER_TURB_AGENT ||
                       field == WG_INFO_POWER_TURB_BASE || field == WG_INFO_WS_TURB_BASE ||
                       field == WG_INFO_TURB_X || field == WG_INFO_TURB_Y) ? N : 1;
    if (i >= p.B * per) return;
    const int e = i / per, t = i - e * per;
    const WgEnv& env = d.env[e];
    const int ctx_id = e * 2 + env.live;
    const WgCtx& cx = d.ctx[ctx_id];
    const size_t ta = (size_t)(ctx_id * p.F) * N + t;
    const size_t tbs = (size_t)(ctx_id * p.F + (p.F - 1)) * N + t;
    float* fo = (float*)out;
    int* io = (int*)out;
    switch (field) {
    case WG_INFO_YAW_AGENT: fo[i] = d.yaw[ta]; break;
    case WG_INFO_YAW_BASE: fo[i] = d.yaw[tbs]; break;
    case WG_INFO_WS_GLOBAL: fo[i] = (float)cx.ws; break;
    case WG_INFO_WD_GLOBAL: fo[i] = (float)cx.wd; break;
    case WG_INFO_TI_GLOBAL: fo[i] = (float)cx.ti; break;
    case WG_INFO_WS_TURB: fo[i] = d.cur_ws[(size_t)ctx_id * N + t]; break;
    case WG_INFO_WD_TURB: fo[i] = d.cur_wd[(size_t)ctx_id * N + t]; break;
    case WG_INFO_POWER_TURB_AGENT: fo[i] = d.power[ta]; break;
    case WG_INFO_POWER_TURB_BASE: fo[i] = d.power[tbs]; break;
    case WG_INFO_POWER_AGENT: { float s = 0.f; for (int k = 0; k < N; ++k) s += d.power[ta + k]; fo[i] = s; break; }
    case WG_INFO_POWER_BASE: { float s = 0.f; for (int k = 0; k < N; ++k) s += d.power[tbs + k]; fo[i] = s; break; }
    case WG_INFO_WS_TURB_BASE: fo[i] = d.u[tbs]; break;
    case WG_INFO_TURB_X: fo[i] = (float)d.xr[(size_t)ctx_id * N + t]; break;
    case WG_INFO_TURB_Y: fo[i] = (float)d.yr[(size_t)ctx_id * N + t]; break;
    case WG_INFO_TIMESTEP: io[i] = env.timestep; break;
    case WG_INFO_TIME_MAX: io[i] = cx.time_max; break;
    case WG_INFO_FS_TIME: fo[i] = (float)d.slot[ctx_id * p.F].time; break;
    case WG_INFO_EPISODE: io[i] = env.episode; break;
    case WG_INFO_ROTOR_UVW_AGENT: fo[i * 3] = d.u[ta]; fo[i * 3 + 1] = d.v[ta]; fo[i * 3 + 2] = d.w[ta]; break;
    case WG_INFO_ROTOR_UVW_BASE: fo[i * 3] = d.u[tbs]; fo[i * 3 + 1] = d.v[tbs]; fo[i * 3 + 2] = d.w[tbs]; break;
    case WG_INFO_RATED_POWER: fo[i] = cx.rated_power; break;
    case WG_INFO_BOX_ID: io[i] = cx.box_id; break;
    case WG_INFO_STEP_POWER_AGENT: fo[i] = d.last_pow_agent[e]; break;
    case WG_INFO_STEP_POWER_BASE: fo[i] = d.last_pow_base[e]; break;
    case WG_INFO_WIND_F64: { double* dd = (double*)out; dd[i * 3] = cx.ws; dd[i * 3 + 1] = cx.wd; dd[i * 3 + 2] = cx.ti; break; }
    default: break;
    }
}

// ===================================================================================================
// k_metrics: deterministic reduction of the per-env running sums -> f32[WG_N_METRICS]
// ===================================================================================================
__global__ void __launch_bounds__(WG_BLOCK) k_metrics(const WgParams p, const WgPtrs d, float* __restrict__ out,
                                                      const int reset_after) {
    __shared__ float red[WG_BLOCK];
    for (int m = 0; m < WG_N_METRICS; ++m) {
        float s = 0.f;
        for (int e = threadIdx.x; e < p.B; e += WG_BLOCK) s += d.metrics[(size_t)e * WG_N_METRICS + m];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = WG_BLOCK / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[m] = red[0];
        __syncthreads();
    }
    if (reset_after)
        for (int i = threadIdx.x; i < p.B * WG_N_METRICS; i += WG_BLOCK) d.metrics[i] = 0.f;
}

// unscaled sensor values of the running episode, same layout as the observation: f32[B, obs_dim]
__global__ void __launch_bounds__(WG_BLOCK) k_measurements(const WgParams p, const WgPtrs d, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * WG_NWAVES + (threadIdx.x >> 6);
    if (e >= p.B) return;
    const int ctx_id = e * 2 + d.env[e].live;
    build_obs<1>(p, d, ctx_id, lane, out + (size_t)e * p.obs_dim, nullptr, d.ring + (size_t)ctx_id * p.ring_stride,
              d.fring + (size_t)ctx_id * p.fring_stride, true);
}
extern "C" void wg_launch_measurements(const WgParams* p, const WgPtrs* d, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_measurements, dim3((p->B + WG_NWAVES - 1) / WG_NWAVES), dim3(WG_BLOCK), 0, st, *p, *d, out);
}

// where cell (i, j, k) of an interleaved box lives: 4 x 4 x 4 bricks of 1 KB when every dimension is a multiple of 4, plain
// [Nx][Ny][Nz] otherwise (the rule box_lookup_dims in wg_flow.hip applies when it reads)
__device__ __forceinline__ size_t wg_box_cell(const int i, const int j, const int k, const int nx, const int ny, const int nz) {
    if (((nx | ny | nz) & 3) != 0) return ((size_t)i * ny + j) * nz + k;
    const size_t nbz = (size_t)(nz >> 2), nbyz = (size_t)(ny >> 2) * nbz;
    return ((size_t)(i >> 2) * nbyz + (size_t)(j >> 2) * nbz + (size_t)(k >> 2)) * 64 + (size_t)((i & 3) * 16 + (j & 3) * 4 + (k & 3));
}
// planar [3][Nx][Ny][Nz] -> interleaved cells of float4 (u, v, w, 0), brick-ordered (wg_box_cell)
__global__ void k_box_repack(const float* __restrict__ planar, float4* __restrict__ out, const int nx, const int ny, const int nz) {
    const size_t n_cells = (size_t)nx * ny * nz;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    const int k = (int)(i % nz), j = (int)((i / nz) % ny), ii = (int)(i / ((size_t)nz * ny));
    out[wg_box_cell(ii, j, k, nx, ny, nz)] = make_float4(planar[i], planar[n_cells + i], planar[2 * n_cells + i], 0.f);
}
extern "C" void wg_launch_box_repack(const float* planar, void* out, int nx, int ny, int nz, hipStream_t st) {
    const size_t n_cells = (size_t)nx * ny * nz;
    hipLaunchKernelGGL(k_box_repack, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, st, planar, (float4*)out, nx, ny, nz);
}

// 4x4x4 block average of the interleaved box (summation order x, y, z innermost, float — as the oracle does);
// output cell (i, j, k) = (v_k, w_k, v_k+1, w_k+1), k+1 periodic: see cbox_lookup_vw
__device__ __forceinline__ void coarse_cell(const float4* __restrict__ fine, int nx, int ny, int nz, int ii, int j, int k,
                                            float& v, float& w) {
    float ay = 0.f, az = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b)
            for (int c = 0; c < 4; ++c) {
                const float4 q = fine[wg_box_cell(ii * 4 + a, j * 4 + b, k * 4 + c, nx, ny, nz)];
                ay += q.y; az += q.z;
            }
    v = ay * (1.0f / 64.0f); w = az * (1.0f / 64.0f);
}
__global__ void k_box_coarsen(const float4* __restrict__ fine, float4* __restrict__ out, int nx, int ny, int nz) {
    const int cnx = nx / 4, cny = ny / 4, cnz = nz / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)cnx * cny * cnz) return;
    // output layout [cny][cnz][cnx]: x fastest (see cbox_lookup_vw)
    const int ii = (int)(i % cnx), k = (int)((i / cnx) % cnz), j = (int)(i / ((size_t)cnx * cnz));
    float4 o;
    coarse_cell(fine, nx, ny, nz, ii, j, k, o.x, o.y);
    coarse_cell(fine, nx, ny, nz, ii, j, k + 1 == cnz ? 0 : k + 1, o.z, o.w);
    out[i] = o;
}
extern "C" void wg_launch_box_coarsen(const void* fine, void* out, int nx, int ny, int nz, hipStream_t st) {
    const size_t n = (size_t)(nx / 4) * (ny / 4) * (nz / 4);
    hipLaunchKernelGGL(k_box_coarsen, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float4*)fine, (float4*)out, nx, ny, nz);
}

// stencil records (FlowPtrs::box8): record (i ny + j) nz + k = the 8 corners of cell origin (i, j, k), periodic wrap included
__global__ void k_box_stencil(const float4* __restrict__ fine, float4* __restrict__ out, const int nx, const int ny, const int nz) {
    const size_t n_cells = (size_t)nx * ny * nz;
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_cells) return;
    const int k = (int)(r % nz), j = (int)((r / nz) % ny), i = (int)(r / ((size_t)nz * ny));
    const int i1 = i + 1 == nx ? 0 : i + 1, j1 = j + 1 == ny ? 0 : j + 1, k1 = k + 1 == nz ? 0 : k + 1;
    float4* o = out + r * 8;
    o[0] = fine[wg_box_cell(i, j, k, nx, ny, nz)];   o[1] = fine[wg_box_cell(i1, j, k, nx, ny, nz)];
    o[2] = fine[wg_box_cell(i, j1, k, nx, ny, nz)];  o[3] = fine[wg_box_cell(i1, j1, k, nx, ny, nz)];
    o[4] = fine[wg_box_cell(i, j, k1, nx, ny, nz)];  o[5] = fine[wg_box_cell(i1, j, k1, nx, ny, nz)];
    o[6] = fine[wg_box_cell(i, j1, k1, nx, ny, nz)]; o[7] = fine[wg_box_cell(i1, j1, k1, nx, ny, nz)];
}
extern "C" void wg_launch_box_stencil(const void* fine, void* out, int nx, int ny, int nz, hipStream_t st) {
    const size_t n_cells = (size_t)nx * ny * nz;
    hipLaunchKernelGGL(k_box_stencil, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, st, (const float4*)fine, (float4*)out, nx, ny, nz);
}

// host-visible launch helpers (defined here so that the <<<>>> syntax stays in one translation unit)
extern "C" void wg_launch_glue(const WgParams* p, const WgPtrs* d, int phase, const uint8_t* mask, float* obs,
                               float* reward, uint8_t* trunc, float* final_obs, hipStream_t st, const WgParams* gp,
                               const WgPtrs* gd) {
    const int grid = (p->B + WG_NWAVES - 1) / WG_NWAVES;
    // rings of one env staged in LDS when they fit (<= 16 KiB per wave); otherwise read from global memory
    // (the per-agent buffer's farm-block scratch shares the wave's region: the fit is decided on the sum, so that the
    // workgroup's request never exceeds the 64 KB no kernel opts in beyond — ADVICE r3)
    int ring_floats = p->ring_stride + p->fring_stride;
    const int scratch = d->multi_out ? p->farm_obs : 0;
    if ((size_t)(ring_floats + scratch) * 4 > 16384) ring_floats = 0;
    const int per_wave = ring_floats + scratch;
    const size_t lds = (size_t)per_wave * 4 * WG_NWAVES;
    // lanes per turbine in the observation's window sums: the largest power of two with N * L <= 64 (16 turbines: 4,
    // 9: 4, 80: 1)
    int L = 1;
    while (L < 8 && p->N * (L * 2) <= WG_WAVE) L *= 2;
#define WG_GLUE_LAUNCH(M, R, LL, P) hipLaunchKernelGGL((k_glue<M, R, LL, P>), dim3(grid), dim3(WG_BLOCK), lds, st, *p, *d, phase, \
                                                      mask, obs, reward, trunc, final_obs, per_wave, ring_floats)
#define WG_GLUE_L(M, R, P) do { if (L == 1) WG_GLUE_LAUNCH(M, R, 1, P); else if (L == 2) WG_GLUE_LAUNCH(M, R, 2, P); \
                                else if (L == 4) WG_GLUE_LAUNCH(M, R, 4, P); else WG_GLUE_LAUNCH(M, R, 8, P); } while (0)
    const bool pre = d->next_obs_ok != nullptr && !d->multi_out;
    if (p->sums_mode) {
        // (gp / gd: device-resident copies of the parameter blocks, read by the rare paths that live out of line)
        const size_t lds2 = (size_t)scratch * 4 * WG_NWAVES;
        if (phase == 1) {
            if (d->multi_out) hipLaunchKernelGGL((k_glue_lean_reset<true>), dim3(grid), dim3(WG_BLOCK), lds2, st, *p, *d, gp, gd, mask, obs, scratch);
            else hipLaunchKernelGGL((k_glue_lean_reset<false>), dim3(grid), dim3(WG_BLOCK), lds2, st, *p, *d, gp, gd, mask, obs, scratch);
        } else {
            // generic instantiation: TI entries or anything farm-level (single-agent farm block, the per-agent blocks' farm part)
            const bool gen = p->turb_ti || p->farm_ti || p->farm_obs > 0 || (p->sum_mask_f | p->cur_mask_f) != 0;
#define WG_LEAN(M, G) hipLaunchKernelGGL((k_glue_lean<M, G>), dim3(grid), dim3(WG_BLOCK), lds2, st, *p, *d, gp, gd, obs, reward, trunc, final_obs, scratch)
            if (d->multi_out) { if (gen) WG_LEAN(true, true); else WG_LEAN(true, false); }
            else { if (gen) WG_LEAN(false, true); else WG_LEAN(false, false); }
#undef WG_LEAN
        }
    } else if (d->multi_out) { if (ring_floats > 0) WG_GLUE_L(true, true, false); else WG_GLUE_L(true, false, false); }
    else if (pre) { if (ring_floats > 0) WG_GLUE_L(false, true, true); else WG_GLUE_L(false, false, true); }
    else { if (ring_floats > 0) WG_GLUE_L(false, true, false); else WG_GLUE_L(false, false, false); }
#undef WG_GLUE_L
#undef WG_GLUE_LAUNCH
}
extern "C" void wg_launch_init(const WgParams* p, const WgPtrs* d, const uint8_t* mask, const uint64_t* seeds,
                               hipStream_t st) {
    const int grid = (p->B + WG_NWAVES - 1) / WG_NWAVES;
    hipLaunchKernelGGL(k_init, dim3(grid), dim3(WG_BLOCK), 0, st, *p, *d, mask, seeds);
}
extern "C" void wg_launch_create(const WgParams* p, const WgPtrs* d, hipStream_t st) {
    hipLaunchKernelGGL(k_create, dim3((p->B + 255) / 256), dim3(256), 0, st, *p, *d);
}
extern "C" void wg_launch_obs_multi(const WgParams* p, const WgPtrs* d, float* out, hipStream_t st) {
    const int n = p->B * p->N;
    hipLaunchKernelGGL(k_obs_multi, dim3((n + 255) / 256), dim3(256), 0, st, *p, *d, out);
}
extern "C" void wg_launch_info(const WgParams* p, const WgPtrs* d, int field, void* out, hipStream_t st) {
    const int n = p->B * p->N;
    hipLaunchKernelGGL(k_info, dim3((n + 255) / 256), dim3(256), 0, st, *p, *d, field, out);
}
extern "C" void wg_launch_metrics(const WgParams* p, const WgPtrs* d, float* out, int reset_after, hipStream_t st) {
    hipLaunchKernelGGL(k_metrics, dim3(1), dim3(WG_BLOCK), 0, st, *p, *d, out, reset_after);
}
