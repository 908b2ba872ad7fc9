// wg_flow_dev.h — device helpers shared by the flow kernels (wg_flow.hip: k_flow, wg_env.hip: k_flow_env).
// Everything here is `static` / inline per translation unit (the library is built without relocatable device code).
#pragma once
#include <hip/hip_runtime.h>

#include "wg_device.h"
#include "wg_obs.h"
#include "wg_flow.h"

__device__ __forceinline__ float m0_cfrac(float ct, float sp) {
    float m = __builtin_amdgcn_rcpf(8.0f * sp * sp);
    m = fminf(m, 1.0f);
    const float a = fmaxf(1.0f - ct * m, 0.0f);
    return 1.0f - __builtin_amdgcn_sqrtf(a);
}

// Frozen emission record of a wake particle, packed into two 32-bit words (16-bit fixed point):
//   rec_a = ct (unorm16 over [0,1])              | k  (unorm16 over [0,0.25]) << 16
//   rec_b = u_e / u_max (unorm16 over [0,1])     | hv (snorm16 over [-16,16] m/s) << 16
// (round 6) u_e — the rotor wind speed at emission, the scale of the particle's deficit — lives IN the record: the brackets of
// a (target, source) pair are then read from the very lines the advection pass streams (record + py), and the separate
// 16-byte gather copy of rounds 2-5 (rec4: one more 128-byte line per candidate pair, 29 % of what a cfg2 step fetched, +14 %
// env-steps/s without it) is gone.  Its place was the initial wake width eps, which is a function of ct alone (model M0:
// eps = eps0 sqrt(beta), beta = (1 + sqrt(1 - ct)) / (2 sqrt(1 - ct))) and is recomputed where it is needed (m0_eps).
// u_e is stored relative to u_max = FlowP::ue_scale x the episode's free-stream speed: 1 x under steady inflow (the deficits
// only lower u: u_e <= U exactly), 2 x with turbulence — a step of U / 65535 ~ 2e-4 m/s, i.e. <= 5e-5 m/s of deficit per wake.
// Quantisation steps (1.5e-5, 3.8e-6, ~2e-4 m/s, 4.9e-4 m/s) are an order of magnitude below the fp32 parity tolerances
// (DESIGN.md §6).  (Tried first: u_e in 18 bits over [0, 32) m/s and hv in 14 — hv's 2e-3 m/s step moves a particle 5 cm in
// 50 s, which the steepest flank of a wake turns into 1e-3 m/s at a rotor: 24 oracle tests 30 % over their bar.)
// The streaming pass reads 8 record bytes per particle, and the "does this particle move" test needs only rec_b.
#define WG_K_MAX 0.25f
#define WG_HV_MAX 16.0f
__device__ __forceinline__ unsigned pack_a(float ct, float k) {
    const unsigned qc = (unsigned)(fminf(fmaxf(ct, 0.f), 1.f) * 65535.0f + 0.5f);
    const unsigned qk = (unsigned)(fminf(fmaxf(k, 0.f), WG_K_MAX) * (65535.0f / WG_K_MAX) + 0.5f);
    return qc | (qk << 16);
}
// (uf = u_e / u_max)
__device__ __forceinline__ unsigned pack_b(float uf, float hv) {
    const unsigned qu = (unsigned)(fminf(fmaxf(uf, 0.f), 1.f) * 65535.0f + 0.5f);
    const int qh = (int)rintf(fminf(fmaxf(hv, -WG_HV_MAX), WG_HV_MAX) * (32767.0f / WG_HV_MAX));
    return qu | ((unsigned)(qh & 0xffff) << 16);
}
__device__ __forceinline__ float rec_ct(unsigned a) { return (float)(a & 0xffffu) * (1.0f / 65535.0f); }
__device__ __forceinline__ float rec_k(unsigned a) { return (float)(a >> 16) * (WG_K_MAX / 65535.0f); }
__device__ __forceinline__ float rec_uf(unsigned b) { return (float)(b & 0xffffu) * (1.0f / 65535.0f); }      // u_e / u_max
__device__ __forceinline__ float rec_hv(unsigned b) { return (float)((int)b >> 16) * (WG_HV_MAX / 32767.0f); }
__device__ __forceinline__ bool rec_moves(unsigned b) { return (b >> 16) != 0u; }
// initial wake width of a particle whose (clamped) thrust coefficient is ct: eps0 sqrt(beta(ct)) — the value the emission computes,
// recomputed from the record's ct (quantised to 1.5e-5: d eps <= 5e-6)
__device__ __forceinline__ float m0_eps(float ct, float eps0) {
    // beta = (1 + rq) / (2 rq) = 0.5 + 0.5 / rq, rq = sqrt(1 - ct): two transcendental instructions (v_rsq, v_sqrt) on the advection
    // pass's per-particle chain
    const float beta = __builtin_fmaf(0.5f, __builtin_amdgcn_rsqf(1.0f - ct), 0.5f);
    return eps0 * __builtin_amdgcn_sqrtf(beta);
}
__device__ __forceinline__ float rec_eps(unsigned a, float eps0) { return m0_eps(rec_ct(a), eps0); }

// lateral position of a wake particle of age j (pre-step clock s_off) after one step of Hill-vortex deflection.  ONE
// definition with explicit fused operations: the advection pass stores this value and the deficit phase of the steady
// compact variant recomputes it for the particles it brackets (see flow_step) — both must produce the same bits
// whatever the surrounding code lets the compiler contract.
__device__ __forceinline__ float m0_advect(float py, unsigned ra, unsigned rb, int j, float s_off_f, float dpart_f,
                                           float inv_D, float dt, float eps0) {
#if defined(WG_ENV_ABLATE) && (WG_ENV_ABLATE & 4)      // (profiling builds, wrong results: what the advection arithmetic costs)
    return py + dt * rec_hv(rb) * 0.5f;
#endif
    const float xrel = __builtin_fmaf((float)j, dpart_f, s_off_f);
    const float ct = rec_ct(ra);
    const float sp = __builtin_fmaf(rec_k(ra), xrel * inv_D, m0_eps(ct, eps0));
    return __builtin_fmaf(rec_hv(rb) * m0_cfrac(ct, sp), dt, py);
}

// x^y for x >= 0 via v_log_f32 / v_exp_f32 (HIP's __powf expands to the full-precision routine)
__device__ __forceinline__ float fast_pow(float x, float y) {
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
}

// n % H for 0 <= n < 2^24, H >= 1, without the ~25-instruction integer division: quotient estimate from the float
// reciprocal (off by at most one), then one correction step each way
__device__ __forceinline__ int fast_mod(int n, int H, float invH) {
    int r = n - (int)((float)n * invH) * H;
    if (r < 0) r += H;
    if (r >= H) r -= H;
    return r;
}

// n % H for a wave-uniform 0 <= n with n * H < 2^32 and magic = floor(2^32 / H) + 1 (host): integer-only, so the
// compiler keeps it on the scalar unit (fast_mod's float reciprocal forces the vector ALU).  H = 1: magic wraps to 1,
// the quotient estimate is 0 and the correction loop is not an option — handled explicitly.
__device__ __forceinline__ int umod_small(int n, int H, unsigned magic) {
    if (H == 1) return 0;
    int r = n - (int)__umulhi((unsigned)n, magic) * H;
    if (r >= H) r -= H;
    if (r < 0) r += H;
    return r;
}

// uniform-grid table lookup (linear interpolation, 0 outside)
__device__ __forceinline__ float tab_lookup(const float* __restrict__ ys, const FlowP& p, float x) {
    const float fx = (x - p.tab_x0) * p.tab_inv_dx;
    if (!(fx >= 0.0f) || fx > (float)(p.n_tab - 1)) return 0.0f;
    int i = (int)fx;
    if (i > p.n_tab - 2) i = p.n_tab - 2;
    const float f = fx - (float)i;
    return ys[i] + f * (ys[i + 1] - ys[i]);
}


// Cold parameters.  k_flow's by-value parameter blocks are ~190 dwords; everything the hot loops do not touch used to
// stay in SGPRs across them anyway (the compiler hoists every kernarg load to the top) and the overflow — ~90 values at
// the main loop's head — went to VGPR lanes: v_writelane / v_readlane are VALU instructions, ~20 % of what a farm step
// executed.  Code that runs once per step (measurement tail, epilogue) therefore re-reads its parameters from the
// kernarg segment through a pointer the optimiser cannot see through: scalar loads that hit the constant cache, no
// register held across the loops.
struct KArgs { FlowP p; FlowPtrs d; };      // layout of the first two kernel arguments in the kernarg segment
typedef const __attribute__((address_space(4))) KArgs* KArgsPtr;
__device__ __forceinline__ KArgsPtr wg_cold_args() {
    KArgsPtr k = (KArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
}


// barrier that orders LDS traffic only: does NOT wait for outstanding global stores (a plain
// __syncthreads() drains vmcnt and exposes the full store latency at every phase boundary)
template <int NT>
__device__ __forceinline__ void lds_barrier() {
    if (NT > WG_WAVE) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // single-wave workgroup: program order suffices
}
// barrier that also makes this workgroup's global stores visible to its own later loads
template <int NT>
__device__ __forceinline__ void full_barrier() {
    if (NT > WG_WAVE) __syncthreads();
    else {
        // (the builtin, not inline asm: the compiler's wait-count pass sees that nothing is outstanding afterwards — with an
        // opaque wait it keeps "possibly pending" loads of skipped branches alive around the flow-step loop and orders
        // later register writes behind them with waits of its own, see wg_wait_vmem)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0)
        asm volatile("" ::: "memory");
    }
}

// s_waitcnt vmcnt(0) that the compiler's own wait-count pass sees (expcnt / lgkmcnt fields at their maxima = no wait), and
// that no memory access is moved across
__device__ __forceinline__ void wg_wait_vmem() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
}

// Rare path at the head of k_flow (WgCtx::init_pending): one wave sets up the episode a retired context will hold.
// Kept out of line so that its register needs (128-bit PCG64 arithmetic, double-precision sin / cos) stay out of the
// hot path's allocation.
static __device__ __attribute__((noinline)) void flow_init_episode(const WgParams* gp, const WgPtrs* gd, WgEnv* env_rw,
                                                            const int e, const int c, const int farm, const int lane) {
    const WgCtx& cx = gd->ctx[e * 2 + c];
    WgRng rng{cx.snap_state, cx.snap_inc, cx.snap_has32, cx.snap_u32};
    wg_ctx_init(*gp, *gd, rng, e, c, lane, cx.episode_tag, farm, farm + 1);
    if (farm == 0 && lane == 0) {
        env_rw->rng_state = rng.rng_state; env_rw->rng_inc = rng.rng_inc;
        env_rw->rng_has32 = rng.rng_has32; env_rw->rng_u32 = rng.rng_u32;
    }
}

// First observation of a background episode, built by the workgroup of its agent farm when the episode's development
// completes (its last window-fill push): the glue wave of the env's truncation then copies obs_dim floats instead of
// staging the rings and building a second observation — the truncating waves are the tail of k_glue.
// Out of line and rare (once per episode and env); reads the rings this workgroup just completed from global memory.
static __device__ __attribute__((noinline)) void wg_first_obs(const WgParams* gp, const WgPtrs* gd, const int ctx_id,
                                                       const int n_pushed, const int lane) {
    const WgParams& p = *gp;
    const WgPtrs& d = *gd;
    if (d.next_obs == nullptr) return;
    float* nobs = d.next_obs + (size_t)ctx_id * p.obs_dim;
    // sums mode without TI / farm-level entries: the observation is written below from the window sums, with the arithmetic
    // of the swap's own fallback (lean_swap) — a state restored from a blob (no prepared observation) then reproduces the
    // uninterrupted run bit for bit.  Otherwise: the ring-based builder.
    const bool from_sums = p.sums_mode && !(p.turb_ti || p.farm_ti || p.farm_obs > 0 || (p.sum_mask_f | p.cur_mask_f) != 0u);
    if (!from_sums)
        build_obs<1>(p, d, ctx_id, lane, nobs, nullptr, d.ring + (size_t)ctx_id * p.ring_stride,
                     d.fring + (size_t)ctx_id * p.fring_stride, false, nullptr, n_pushed);
    if (p.sums_mode) {
        // sums mode (k_glue_lean): the episode's window sums, summed afresh from its rings into WgPtrs::wsum — the swap then
        // finds them ready.  Lg = 2^k lanes share an entity (turbine t, or N = the farm-level deques); lane `sub` of a group
        // takes the window's samples sub, sub + Lg, ... eight at a time; double sums of floats are exact, so the partial sums
        // combine to THE sum in any order.
        const int N = p.N, NS = N + 1;
        int Lg = 1;
        while (Lg < 8 && N * (Lg * 2) <= WG_WAVE) Lg *= 2;
        const int sub = lane & (Lg - 1), per_pass = WG_WAVE / Lg;
        const bool farm_ent = p.sum_mask_f != 0u;
        for (int t0 = 0; t0 < N + (farm_ent ? 1 : 0); t0 += per_pass) {
            const int ent = t0 + lane / Lg;
            const bool have = ent < N || (ent == N && farm_ent);
            const SumsEnt q = wg_sums_ent(p, d, ctx_id, have ? ent : 0);
            double* ws_ = d.wsum + (size_t)ctx_id * WG_N_SUMS * NS + (have ? ent : 0);
            float* o = (from_sums && have) ? nobs + (size_t)ent * p.turb_obs : nullptr;
            int n = 0;
#pragma nounroll
            for (int sl = 0; sl < WG_N_SUMS; ++sl) {
                const bool on = have && ((q.sm >> sl) & 1u);
                const int ch = sl < WG_N_CH ? sl : WG_CH_WS;
                const int off = q.farm ? p.fring_off[ch] : p.ring_off[ch];
                const int cap = p.ring_cap[ch];
                const int cnt = p.sum_w[sl] < n_pushed ? p.sum_w[sl] : n_pushed;
                const int r0 = (n_pushed - cnt) % cap;
                const bool cur_on = o && sl < WG_N_CH && ((p.oc.cur_mask >> ch) & 1u) && n_pushed > 0;
                const float cv = cur_on ? q.rb[off + ((n_pushed - 1) % cap) * q.stride] : 0.f;
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
                // (wg_obs_turbine<false>: a channel's `current` entry, then its rolling mean — as lean_swap writes them)
                if (o && sl < WG_N_CH && sub == 0 && n_pushed > 0) {
                    if (cur_on) o[n++] = wg_scale_r(cv, p.oc.mn[ch], p.oc.inv_rng[ch]);
                    if ((p.oc.rol_mask >> ch) & 1u) o[n++] = wg_scale_r(wg_sums_mean(p.oc, acc, ch, n_pushed), p.oc.mn[ch], p.oc.inv_rng[ch]);
                }
            }
        }
    }
    if (lane == 0) d.next_obs_ok[ctx_id] = 1;
}

