// wg_env_common.h — what the one-wave-per-env step kernels share (wg_env.hip: k_flow_env, steady inflow; wg_envb.hip: k_flow_envb,
// frozen-box inflow): the kernel-argument view, the rare paths (episode set-up, first observation of a completed background
// episode) and small wave helpers.  Everything is inline per translation unit.
#pragma once
#include <hip/hip_runtime.h>

#include "wg_flow_dev.h"
#include "wg_glue_lean.h"

#ifndef WG_ENV_DPP_SCAN
#define WG_ENV_DPP_SCAN 1   // list offsets from DPP prefix sums (0: __shfl_up ladders, for A/B builds: 67.0 / 52.2 against 67.8 / 53.5)
#endif
#ifndef WG_ENV_DEFER_INIT
#define WG_ENV_DEFER_INIT 1       // one wave per env: a retired context's next episode is set up after the wave's step, not before it
#endif
#ifndef WG_ENV_SPLIT_PLAN
#define WG_ENV_SPLIT_PLAN 1       // the background episode's farms are planned one by one, half a period apart (see the prologue)
#endif
#ifndef WG_ENV_FIRST_OBS_LATER
#define WG_ENV_FIRST_OBS_LATER 1  // a completed background episode's first observation is built in the launch after the completing one
#endif
#ifndef WG_ENV_PRIO
#define WG_ENV_PRIO 0       // 1: waves on a rare, long path raise their issue priority — measured slightly SLOWER (cfg2 65.2 vs 66.5, cfg4
                            // 54.3 vs 54.9 M env-steps/s): their extra time is their own latency chain, not contention
#endif
#ifndef WG_ENV_WAVES
#define WG_ENV_WAVES 4      // 128 VGPRs: 4096 envs = the chip's 4096 wave slots at 4 waves per SIMD, one dispatch round
#endif

// rare path: the episode a retired context will hold (WgCtx::init_pending), both farms at once.  KARG: the launch carries the
// glue's parameter blocks in its own arguments (k_flow_env<., GLUE != 0>) — the set-up is inlined and reads them from there
// (scalar loads that hit the constant cache); otherwise out of line, from the handle's copies in memory.
static __device__ __attribute__((noinline)) void env_init_episode_mem(const WgParams* gp, const WgPtrs* gd, WgEnv* env_rw,
                                                                      const int e, const int c, const int lane) {
    const WgCtx& cx = gd->ctx[e * 2 + c];
    WgRng rng{cx.snap_state, cx.snap_inc, cx.snap_has32, cx.snap_u32};
    wg_ctx_init(*gp, *gd, rng, e, c, lane, cx.episode_tag, 0, gp->F);
    if (lane == 0) {
        env_rw->rng_state = rng.rng_state; env_rw->rng_inc = rng.rng_inc;
        env_rw->rng_has32 = rng.rng_has32; env_rw->rng_u32 = rng.rng_u32;
    }
}
struct EnvKArgs;
template <bool KARG>
__device__ __forceinline__ void env_init_episode(const int e, const int c, const int lane);

__device__ __forceinline__ int env_scan(const int v, const int tid) {
#if WG_ENV_DPP_SCAN
    return wg_wave_scan_i(v);
#else
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int w = __shfl_up(inc, o, 64); if (tid >= o) inc += w; }
    return inc;
#endif
}

// uniform-grid table lookup (linear interpolation, 0 outside): tab_lookup with its parameters as scalars
__device__ __forceinline__ float env_tab(const float* __restrict__ ys, const float x0, const float inv_dx, const int n_tab, const float x) {
    const float fx = (x - x0) * inv_dx;
    if (!(fx >= 0.0f) || fx > (float)(n_tab - 1)) return 0.0f;
    int i = (int)fx;
    if (i > n_tab - 2) i = n_tab - 2;
    const float f = fx - (float)i;
    return ys[i] + f * (ys[i + 1] - ys[i]);
}

// Register discipline (the "register cliff" of k_flow): the by-value parameter blocks are NEVER read directly.
// Every phase fetches the parameters it needs through a fresh opaque pointer to the kernarg segment (wg_cold_args: scalar
// loads that hit the constant cache) — nothing is held in SGPRs across phases, so nothing overflows into VGPR lanes
// (v_readlane / v_writelane are VALU instructions: they were 15 % of what k_flow executed).
__device__ __forceinline__ bool role_dev_any(const int autoreset, const int dev_rem, const int fill_rem) {
    return autoreset != 0 && (dev_rem > 0 || fill_rem > 0);
}
// the kernel's arguments as they lie in the kernarg segment
struct EnvKArgs {
    FlowP p; FlowPtrs d; int mode; const float* actions; const uint8_t* mask; int chunk; WgParams gp; WgPtrs gd;
    float* obs; float* reward; uint8_t* trunc; float* final_obs;
};
typedef const __attribute__((address_space(4))) EnvKArgs* EnvKArgsPtr;
template <bool KARG>
__device__ __forceinline__ void env_init_episode(const int e, const int c, const int lane) {
    const EnvKArgsPtr kg = (EnvKArgsPtr)wg_cold_args();
    if (!KARG) { env_init_episode_mem(kg->d.gp, kg->d.gd, kg->d.env_rw + e, e, c, lane); return; }
    const WgCtx& cx = kg->gd.ctx[e * 2 + c];
    WgRng rng{cx.snap_state, cx.snap_inc, cx.snap_has32, cx.snap_u32};
    wg_ctx_init(kg->gp, kg->gd, rng, e, c, lane, cx.episode_tag, 0, kg->gp.F);
    if (lane == 0) {
        WgEnv* const env_rw = kg->d.env_rw + e;
        env_rw->rng_state = rng.rng_state; env_rw->rng_inc = rng.rng_inc;
        env_rw->rng_has32 = rng.rng_has32; env_rw->rng_u32 = rng.rng_u32;
    }
}
// First observation + window sums of a completed background episode (wg_first_obs, wg_flow_dev.h) for the configurations the
// fused step kernel serves — sums mode, no TI / farm-level entries, N <= 32 — inlined, with the parameter blocks read from the
// kernel's own arguments (scalar loads), one uniform base + 32-bit offsets, and every load of the four windows in flight
// together: the out-of-line generic routine waits for memory ~150 times (its loads sit behind spill reloads) and made the
// wave that called it one of its launch's last three (59 us against 40, cfg2 x 4096).  Same partial sums in the same order:
// bit-identical results.
template <bool KARG>
__device__ __forceinline__ void env_first_obs(const int ctx_id, const int n_pushed, const int lane) {
    const EnvKArgsPtr kg = (EnvKArgsPtr)wg_cold_args();
    bool lean = false;
    if (KARG) {
        const auto& p = kg->gp;
        lean = p.sums_mode && !(p.turb_ti || p.farm_ti || p.farm_obs > 0 || (p.sum_mask_f | p.cur_mask_f) != 0u) && p.N <= 32 &&
               kg->gd.next_obs != nullptr;
    }
    if (!lean) { wg_first_obs(kg->d.gp, kg->d.gd, ctx_id, n_pushed, lane); return; }
    const auto& p = kg->gp;
    const auto& d = kg->gd;
    const int N = p.N, NS = N + 1;
    int Lg = 1, lgs = 0;
    while (Lg < 8 && N * (Lg * 2) <= WG_WAVE) { Lg *= 2; ++lgs; }
    const int sub = lane & (Lg - 1), ent = lane >> lgs;      // (N Lg <= 64: every entity in one pass)
    const bool have = ent < N;
    const unsigned entc = have ? (unsigned)ent : 0u;
    const float* const rb = d.ring + (size_t)ctx_id * p.ring_stride;
    double* const ws_ = d.wsum + (size_t)ctx_id * WG_N_SUMS * NS + entc;
    float* const o = d.next_obs + (size_t)ctx_id * p.obs_dim + (size_t)entc * p.turb_obs;
    const unsigned sm = p.sum_mask_t, cmask = p.oc.cur_mask, rmask = p.oc.rol_mask;
    float v0[WG_N_CH][8], cv[WG_N_CH];
#pragma unroll
    for (int sl = 0; sl < WG_N_CH; ++sl) {
        const int off = p.ring_off[sl], cap = p.ring_cap[sl];
        const int cnt = p.sum_w[sl] < n_pushed ? p.sum_w[sl] : n_pushed;
        const int r0 = (n_pushed - cnt) % cap;
        const int last = n_pushed > 0 ? (n_pushed - 1) % cap : 0;
        cv[sl] = rb[(unsigned)(off + last * N) + entc];
#pragma unroll
        for (int u = 0; u < 8; ++u) {      // (clamped rows: always valid addresses, masked where they are added)
            int row = r0 + max(min(sub + u * Lg, cnt - 1), 0); if (row >= cap) row -= cap;
            v0[sl][u] = rb[(unsigned)(off + row * N) + entc];
        }
    }
    int n = 0;
#pragma unroll
    for (int sl = 0; sl < WG_N_CH; ++sl) {
        const bool on = have && ((sm >> sl) & 1u);
        const int off = p.ring_off[sl], cap = p.ring_cap[sl];
        const int cnt = p.sum_w[sl] < n_pushed ? p.sum_w[sl] : n_pushed;
        const int r0 = (n_pushed - cnt) % cap;
        double acc = 0.0;
        if (on) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (sub + u * Lg < cnt) acc += (double)v0[sl][u];
#pragma nounroll
            for (int k = sub + 8 * Lg; k < cnt; k += 8 * Lg) {      // (windows beyond 8 Lg samples)
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int row = r0 + min(k + u * Lg, cnt - 1); if (row >= cap) row -= cap;
                    v[u] = rb[(unsigned)(off + row * N) + entc];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k + u * Lg < cnt) acc += (double)v[u];
            }
        }
        for (int s2 = 1; s2 < Lg; s2 <<= 1) acc += __shfl_xor(acc, s2, 64);
        if (on && sub == 0) ws_[(size_t)sl * NS] = acc;
        if (have && sub == 0 && n_pushed > 0) {
            if ((cmask >> sl) & 1u) o[n++] = wg_scale_r(cv[sl], p.oc.mn[sl], p.oc.inv_rng[sl]);
            if ((rmask >> sl) & 1u) o[n++] = wg_scale_r(wg_sums_mean(p.oc, acc, sl, n_pushed), p.oc.mn[sl], p.oc.inv_rng[sl]);
        }
    }
    if (lane == 0) d.next_obs_ok[ctx_id] = 1;
}

// LDS flag between two waves of a workgroup (k_flow_env's pass waves).  The wait is bounded: a protocol error must not hang the
// GPU — the caller latches WG_STATUS_BIT_STATE on a time-out.
__device__ __forceinline__ void env_flag_set(int* const f) {
    __atomic_store_n(f, 1, __ATOMIC_RELAXED);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ bool env_flag_wait(int* const f) {
    for (int n = 0; n < (1 << 22); ++n) {
        if (__atomic_load_n(f, __ATOMIC_RELAXED) != 0) { asm volatile("" ::: "memory"); return true; }
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

struct EnvFlowOut {          // what the flow part hands to the glue tail of k_step_env
    int env_live, bg_init_pending;
    int truncates;            // the env truncates in this step (known from its header): the only case in which the glue needs the
                              // background context's wave to have finished (WPE 2)
    int steps_done, time_max_live;   // the env header as the prologue read it (before any wave of this launch can have written it): what
                              // the background context's wave plans its next share from (WPE 2)
    int rounds, first_obs;    // (WG_TIMELINE builds: flow rounds taken, first observation of a completed background episode built)
    // (k_flow_env with a pass wave, the running episode's main wave) per-lane values of the step for the glue: the samples the lane's
    // agent-farm turbine pushed into its rings, its yaw before / after, its power and its baseline twin's
    float g_nw[WG_N_CH], g_yaw, g_old, g_pw, g_pwb;
};
