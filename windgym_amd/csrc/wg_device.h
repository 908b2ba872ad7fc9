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
template <class R>
__device__ inline void wg_pcg_seed(R& e, uint64_t seed) {
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
template <class R>
__device__ inline uint64_t wg_pcg_next64(R& e) {
    e.rng_state = e.rng_state * WG_PCG_MULT + e.rng_inc;
    uint64_t hi = (uint64_t)(e.rng_state >> 64), lo = (uint64_t)e.rng_state;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(e.rng_state >> 122);
    return (x >> rot) | (x << ((-rot) & 63));
}
template <class R>
__device__ inline uint32_t wg_pcg_next32(R& e) {
    if (e.rng_has32) {
        e.rng_has32 = 0;
        return e.rng_u32;
    }
    uint64_t n = wg_pcg_next64(e);
    e.rng_has32 = 1;
    e.rng_u32 = (uint32_t)(n >> 32);
    return (uint32_t)(n & 0xffffffffu);
}
template <class R>
__device__ inline double wg_pcg_uniform(R& e, double low, double high) {
    // numpy rounds the product and the sum separately (random_uniform: off + rng * next_double): no fma contraction
    // here — it would differ by 1 ulp on some draws (HIP's __dmul_rn / __dadd_rn are plain operators and contract too)
#pragma clang fp contract(off)
    const double u = (double)(wg_pcg_next64(e) >> 11) * (1.0 / 9007199254740992.0);
    const double scaled = (high - low) * u;
    return low + scaled;
}
template <class R>
__device__ inline uint32_t wg_pcg_integers(R& e, uint32_t high) {
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

// the same interpolation done by a whole wave: every lane fetches one table node per pass and a ballot counts the nodes
// <= x — ONE memory round trip instead of the binary search's chain of dependent loads (the episode set-up at the head
// of k_flow sits on the launch's critical path).  Result valid in every lane; bit-identical to wg_tab_interp.
template <class T>
__device__ inline T wg_tab_interp_wave(const T* xs, const T* ys, int n, T x, int lane) {
    int cnt = 0;                         // nodes with xs[i] <= x among i < n - 1
    for (int i0 = 0; i0 < n - 1; i0 += 64) {
        const int i = i0 + lane;
        const bool le = (i < n - 1) && (xs[i] <= x);
        cnt += __popcll(__ballot(le));
    }
    if (!(x >= xs[0]) || x > xs[n - 1]) return (T)0;
    const int lo = cnt - 1, hi = lo + 1;         // xs ascending: the nodes <= x are a prefix; x >= xs[0] -> cnt >= 1
    const T f = (x - xs[lo]) / (xs[hi] - xs[lo]);
    return ys[lo] + f * (ys[hi] - ys[lo]);
}

// ---------------------------------------------------------------------------------------------------
// wave64 reductions
// ---------------------------------------------------------------------------------------------------
// Wave-wide sums with a FIXED tree (deterministic, the same in every kernel): four DPP steps inside each row of 16 lanes
// (quad swaps, half-row mirror, row mirror: one VALU instruction each, no LDS traffic), then the rows are combined with
// two lane permutes.  Result valid in every lane.
template <int CTRL>
__device__ __forceinline__ float wg_dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
#define WG_ROW_REDUCE(v, OP)                                                 \
    v = OP(v, wg_dpp_f<0xB1>(v));  /* quad_perm [1,0,3,2] */               \
    v = OP(v, wg_dpp_f<0x4E>(v));  /* quad_perm [2,3,0,1] */               \
    v = OP(v, wg_dpp_f<0x141>(v)); /* row_half_mirror */                   \
    v = OP(v, wg_dpp_f<0x140>(v)); /* row_mirror */
__device__ __forceinline__ float wg_addf(float a, float b) { return a + b; }
// sum over the 16 lanes of each row; valid in every lane of the row
__device__ inline float wg_row_sum(float v) {
    WG_ROW_REDUCE(v, wg_addf)
    return v;
}
__device__ inline float wg_wave_sum(float v) {
    WG_ROW_REDUCE(v, wg_addf)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ inline int wg_wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive prefix sum over the 64 lanes, in DPP steps only (four Kogge-Stone shifts inside each row of 16, then the row totals
// carried across with row_bcast:15 / row_bcast:31): six VALU instructions and no LDS crossbar round trips — a __shfl_up ladder
// is six dependent ds_bpermute_b32 (~100 cycles each on a wave's critical path)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int wg_dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ int wg_wave_scan_i(int v) {
    v += wg_dpp_i<0x111, 0xf>(v);      // row_shr:1
    v += wg_dpp_i<0x112, 0xf>(v);      // row_shr:2
    v += wg_dpp_i<0x114, 0xf>(v);      // row_shr:4
    v += wg_dpp_i<0x118, 0xf>(v);      // row_shr:8
    v += wg_dpp_i<0x142, 0xa>(v);      // row_bcast:15 -> rows 1, 3
    v += wg_dpp_i<0x143, 0xc>(v);      // row_bcast:31 -> rows 2, 3
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
// episode context initialisation (wave-cooperative): WindFarmEnv.reset up to fs.run (Wind_Farm_Env.py:689-732)
//   rng        the env's PCG64 (any struct with rng_state / rng_inc / rng_has32 / rng_u32): lane 0 draws from it
//   f_lo..f_hi farms whose slot + turbine arrays this call initialises (k_init: all; k_flow: the workgroup's own farm —
//              the two farm workgroups of a context run this redundantly from the same generator snapshot and write
//              IDENTICAL context-level values, see WgCtx::init_pending)
// ---------------------------------------------------------------------------------------------------
struct WgRng {
    wg_u128 rng_state, rng_inc;
    uint32_t rng_has32, rng_u32;
};

// Share of the background episode's remaining work (flow sub-steps) to run during the next step(), `left` steps
// before the running episode truncates.  work/left per step on average, spread EVENLY: floor(work/left + phi) with a
// low-discrepancy dither phi (golden-ratio sequence over the step index, de-phased per env).  The ratio is recomputed
// from the remaining quantities every step, so the schedule is self-correcting, and at left == 1 it returns all that
// remains: the episode is ready exactly at truncation.  (ceil(work/left) — the first version — front-loads: a
// background episode needing 280 steps during a 600-step episode ran on each of the first 280 launches, so after a
// synchronised start every launch carried twice the flow work of the steady state.)
#ifdef WG_TIMELINE
// debug build (-DWG_TIMELINE, env WG_TIMELINE_OUT=file): thread 0 of every workgroup records shader-clock stamps
// at the phase boundaries of its step; wg_destroy dumps them.  This is how the per-workgroup latency budget in
// DESIGN.md §4.1 was measured.
__shared__ long long wg_stamps[20];
__shared__ int wg_counts[8];
#define WG_STAMP(k) do { if (threadIdx.x == 0) wg_stamps[k] = clock64(); } while (0)
#define WG_STAMP2(k) WG_STAMP(k)
// WG_ICLK(i): cycles since the previous WG_ICLK of thread 0 into wg_counts[i] (episode set-up: wg_ctx_init)
#define WG_ICLK0() long long wg_iclk_ = clock64()
#define WG_ICLK(i) do { if (threadIdx.x == 0) { const long long c_ = clock64(); wg_counts[i] = (int)(c_ - wg_iclk_); wg_iclk_ = c_; } } while (0)
#else
#define WG_STAMP(k) do { } while (0)
#define WG_ICLK0() do { } while (0)
#define WG_ICLK(i) do { } while (0)
#endif

#ifndef WG_SHADOW_MARGIN
#define WG_SHADOW_MARGIN 2
#endif
__device__ inline int wg_shadow_share(const int work, long left, const int steps_done, const int e, const uint32_t phase = 0u) {
    if (work <= 0) return 0;
    // (finished WG_SHADOW_MARGIN steps EARLY: the launch of the truncating step then carries no background work, first
    // observation included — in the one-wave-per-env kernels the truncating wave is the launch's last one anyway)
    left -= WG_SHADOW_MARGIN;
    if (left < 1) left = 1;
    const uint32_t phi24 = ((uint32_t)steps_done * 2654435769u + (uint32_t)e * 0x9E3779B1u + phase) >> 8;   // [0, 2^24)
    // float arithmetic (a 64-bit integer division costs ~150 instructions on this kernel's latency chain): rcp(1) is
    // exact, so left == 1 still returns exactly `work`; elsewhere an off-by-one in the floor is absorbed by the
    // next step's recomputed ratio
    const float share = (float)work * __builtin_amdgcn_rcpf((float)left) + (float)phi24 * (1.0f / 16777216.0f);
    int it = (int)share;
    if (left == 1) it = work;
    return it > work ? work : it;
}

// wg_tab_interp_wave for a table of at most 64 nodes that every lane has already loaded its node of (xs_l, ys_l: node `lane`,
// anything beyond the table's end): the bracketing nodes come from the lanes that hold them, not from memory a second time.
// Same arithmetic on the same values.
__device__ inline double wg_tab_interp_lanes(const double xs_l, const double ys_l, const int n, const double x, const int lane) {
    const int cnt = __popcll(__ballot((lane < n - 1) && (xs_l <= x)));
    const double x_first = __shfl(xs_l, 0, 64), x_last = __shfl(xs_l, n - 1, 64);
    const int lo = cnt > 0 ? cnt - 1 : 0, hi = lo + 1;
    const double xl = __shfl(xs_l, lo, 64), xh = __shfl(xs_l, hi, 64), yl = __shfl(ys_l, lo, 64), yh = __shfl(ys_l, hi, 64);
    if (!(x >= x_first) || x > x_last) return 0.0;
    const double f = (x - xl) / (xh - xl);
    return yl + f * (yh - yl);
}

// Episode set-up of context c of env e: wind conditions and initial yaws drawn from the env's generator, the layout rotated
// into the flow frame, chain lengths / compact ring layout, slot clocks and turbine state of farms f_lo .. f_hi - 1.
// Once per episode and env — but in the one-wave-per-env step kernel the wave that runs it is the launch's last (its set-up
// took ~30 000 cycles on top of a 39 us step: ~12 dependent trips to memory and a serial chain of N draws in lane 0), so:
//   * P / D are the parameter blocks' types WITH their address space — the fused kernel passes its own by-value arguments
//     (scalar loads from the kernarg segment) — and __restrict__: no kernel writes the blocks, and without the promise every
//     store here (float / double / int arrays that may overlap them, as far as the compiler knows) made the next p.x / d.x a
//     reload;
//   * everything the set-up reads from memory is requested up front, in one trip: layout, power table nodes, jump constants;
//   * the draws are one per lane: the generator's state k draws ahead is A_k s + G_k inc (WgPtrs::pcg_jump), so turbine t's yaw
//     is a function of the state after lane 0's draws — not the end of a chain of t + 1 steps (~100 instructions each).
template <class P, class D, class R>
__device__ inline void wg_ctx_init(const P& __restrict__ p, const D& __restrict__ d, R& rng, const int e, const int c, const int lane,
                                   const int episode_tag, const int f_lo, const int f_hi) {
    const int N = p.N, F = p.F;
    const int ctx_id = e * 2 + c;
    WgCtx& cx = d.ctx[ctx_id];
    WG_ICLK0();
    // ---- loads ----
    const int tl = lane < N ? lane : 0;
    const double xp_l = d.x_pos[tl], yp_l = d.y_pos[tl];
    const int n_tab = p.n_tab;
    const int il = lane < n_tab ? lane : n_tab - 1;
    const double tws_l = d.tab_ws_d[il], tpw_l = d.tab_power_d[il];
    const uint64_t* const jmp = d.pcg_jump;
    const int kmax = N > 3 ? N : 3;                       // (the table has max(N, 3) + 1 rows)
    uint64_t jl[4], jn[4], j3[4];
    {
        const int kl = lane <= kmax ? lane : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { jl[i] = jmp[4 * (size_t)kl + i]; jn[i] = jmp[4 * (size_t)N + i]; j3[i] = jmp[4 * (size_t)3 + i]; }
    }
    const bool yaw_rnd = p.yaw_init == WG_YAWINIT_RANDOM, yaw_def = p.yaw_init == WG_YAWINIT_DEFINED && p.has_yaw_defined;
    const double ydef_l = yaw_def ? d.yaw_defined[tl] : 0.0;
    double ov[3] = {0.0, 0.0, 0.0};
    const bool has_ov = d.wind_override != nullptr;
    if (has_ov) { ov[0] = d.wind_override[(size_t)e * 3]; ov[1] = d.wind_override[(size_t)e * 3 + 1]; ov[2] = d.wind_override[(size_t)e * 3 + 2]; }

    // ---- draws ----
    const wg_u128 inc = rng.rng_inc;                      // (the stream: set at seeding)
    auto ahead = [&](const uint64_t* j, const wg_u128 s) {
        const wg_u128 A = ((wg_u128)j[1] << 64) | j[0], G = ((wg_u128)j[3] << 64) | j[2];
        return A * s + G * inc;
    };
    auto draw = [&](const wg_u128 s, const double lo, const double hi) {      // the uniform draw the generator in state s makes next
        struct { wg_u128 rng_state, rng_inc; } one{s, inc};
        return wg_pcg_uniform(one, lo, hi);
    };
    // _set_windconditions (:564-568): ws, ti, wd are draws 1, 2, 3 of the episode — lanes 0, 1, 2
    double ws, ti, wd;
    {
        // (lane 0's copy of the state is THE state — R may live in memory, written by lane 0 alone)
        const wg_u128 s_l = rng.rng_state;
        uint32_t iw[4] = {(uint32_t)s_l, (uint32_t)(s_l >> 32), (uint32_t)(s_l >> 64), (uint32_t)(s_l >> 96)};
#pragma unroll
        for (int i = 0; i < 4; ++i) iw[i] = (uint32_t)__shfl((int)iw[i], 0, 64);
        const wg_u128 s_in = ((wg_u128)iw[3] << 96) | ((wg_u128)iw[2] << 64) | ((wg_u128)iw[1] << 32) | (wg_u128)iw[0];
        const double lo = lane == 0 ? p.ws_min : lane == 1 ? p.ti_min : p.wd_min;
        const double hi = lane == 0 ? p.ws_max : lane == 1 ? p.ti_max : p.wd_max;
        const double u = draw(ahead(jl, s_in), lo, hi);
        ws = __shfl(u, 0, 64); ti = __shfl(u, 1, 64); wd = __shfl(u, 2, 64);
        if (lane == 0) rng.rng_state = ahead(j3, s_in);
    }
    if (has_ov) {                                         // FarmEval.set_wind_vals, per env; NaN = keep the sampled value
        if (ov[0] == ov[0]) ws = ov[0];
        if (ov[1] == ov[1]) wd = ov[1];
        if (ov[2] == ov[2]) ti = ov[2];
    }
    wg_u128 s_head = 0;           // lane 0: the generator's state after its own draws
    if (lane == 0) {
        uint32_t tseed = 0;
        if (p.turb_mode == WG_TURB_RANDOM || p.turb_mode == WG_TURB_BOX_SHIFT)
            tseed = wg_pcg_integers(rng, 100000);                                  // _def_site (:623, :642)
        cx.box_ox = 0.0; cx.box_oy = 0.0; cx.box_id = 0;
        if (p.turb_mode == WG_TURB_BOX_POOL) {     // tf_file = self.np_random.choice(self.TF_files) (:614)
            // FarmEval.update_tf(path) makes TF_files a list of ONE: the choice then consumes no random number
            const int fixed = d.box_override ? d.box_override[e] : -1;
            if (fixed >= 0) cx.box_id = fixed < p.n_boxes ? fixed : 0;
            else cx.box_id = (int)wg_pcg_integers(rng, (uint32_t)(p.n_boxes > 0 ? p.n_boxes : 1));
        }
        if (p.turb_mode == WG_TURB_BOX_SHIFT && p.bnx > 0) {
            // "MannGenerate" draws a fresh box per episode from its seed (:623-637).  Here the seed picks one of the
            // n_boxes generated realisations (wg_set_turbulence_boxes: a pool generated on the device) and a horizontal
            // offset into it: n_boxes x (Nx x Ny offsets) distinct inflows, no random number beyond the reference's draw
            if (p.n_boxes > 1) cx.box_id = (int)(tseed % (uint32_t)p.n_boxes);
            uint32_t a, b;
            wg_philox_turb(tseed, 0u, 0u, 0u, 0x4fu, a, b);
            cx.box_ox = (double)a * (1.0 / 4294967296.0) * p.bnx * p.bdx;
            cx.box_oy = (double)b * (1.0 / 4294967296.0) * p.bny * p.bdy;
        }
        s_head = rng.rng_state;
        cx.ws = ws; cx.ti = ti; cx.wd = wd; cx.turb_seed = tseed;
        // the context-level counters have ONE writer each: n_pushed / pend_farm_n belong to farm 0's workgroup (it also
        // advances them in its epilogue), pend_base_n to the last farm's — a late initialising workgroup of the other
        // farm must not zero what the first one has already advanced (pipelined set-up at the head of k_flow)
        if (f_lo == 0) { cx.n_pushed = 0; cx.pend_farm_n = 0; }
        if (f_hi == F) cx.pend_base_n = 0;
        cx.episode_tag = episode_tag;
    }
    {   // yaw init (:715-720), one turbine per lane; the baseline farm starts from the agent's yaws (:781).  Random: turbine
        // t's is draw t + 1 after lane 0's
        uint32_t sw[4] = {(uint32_t)s_head, (uint32_t)(s_head >> 32), (uint32_t)(s_head >> 64), (uint32_t)(s_head >> 96)};
#pragma unroll
        for (int i = 0; i < 4; ++i) sw[i] = (uint32_t)__shfl((int)sw[i], 0, 64);
        const wg_u128 s0 = ((wg_u128)sw[3] << 96) | ((wg_u128)sw[2] << 64) | ((wg_u128)sw[1] << 32) | (wg_u128)sw[0];
        const double ys = p.yaw_start;
        float* const yaw_c = d.yaw + (size_t)ctx_id * F * N;
        for (int t = lane; t < N; t += WG_WAVE) {
            float y0 = 0.f;
            if (yaw_rnd) {
                uint64_t jt[4];
                if (t < WG_WAVE) { jt[0] = jl[0]; jt[1] = jl[1]; jt[2] = jl[2]; jt[3] = jl[3]; }
                else { for (int i = 0; i < 4; ++i) jt[i] = jmp[4 * (size_t)t + i]; }
                y0 = (float)draw(ahead(jt, s0), -ys, ys);
            } else if (yaw_def) y0 = (float)(t < WG_WAVE ? ydef_l : d.yaw_defined[t]);
            for (int f = f_lo; f < f_hi; ++f) yaw_c[(size_t)f * N + t] = y0;
        }
        if (yaw_rnd && lane == 0) rng.rng_state = ahead(jn, s0);
    }
    WG_ICLK(5);
    {
        const double rp = n_tab <= WG_WAVE ? wg_tab_interp_lanes(tws_l, tpw_l, n_tab, ws, lane)
                                            : wg_tab_interp_wave<double>(d.tab_ws_d, d.tab_power_d, n_tab, ws, lane);   // :700
        if (lane == 0) cx.rated_power = (float)rp;
    }
    // flow frame: rotate the layout by theta = 270 - wd about the farm centre
    const double th = (270.0 - wd) * (WG_PI_D / 180.0);
    const double cth = cos(th), sth = sin(th);
    const double cx0 = p.cx0, cy0 = p.cy0;
    double* const xr_c = d.xr + (size_t)ctx_id * N;
    double* const yr_c = d.yr + (size_t)ctx_id * N;
    double xmin = 1e300, xmax = -1e300;
    for (int t = lane; t < N; t += WG_WAVE) {
        const double dx = (t < WG_WAVE ? xp_l : d.x_pos[t]) - cx0, dy = (t < WG_WAVE ? yp_l : d.y_pos[t]) - cy0;
        const double xr = cx0 + dx * cth + dy * sth;
        const double yr = cy0 - dx * sth + dy * cth;
        xr_c[t] = xr;
        yr_c[t] = yr;
        xmin = fmin(xmin, xr); xmax = fmax(xmax, xr);
    }
    xmin = wg_wave_min_d(xmin); xmax = wg_wave_max_d(xmax);
    // chain pruning: a particle of turbine t that is older than jneed[t] has passed the most downstream turbine of
    // the farm (the bracket of the farthest target uses ages floor(dx / dpart) and + 1) and can never reach a rotor
    // again -> the advection pass stops streaming it.  Every output of step() is unchanged.
    // compact rings: turbine t keeps ages 0 .. jneed[t] (R_t = jneed[t] + 1 rounded up to a multiple of 4, at most P);
    // exclusive prefix sum over the turbines (wave scan per 64 turbines, running base), then the owner of every quad of
    // ring slots
    const int P_ = p.P, full_chains = p.full_chains, compact = p.compact;
    const double dpart = p.dpart;
    int* const jneed_c = d.jneed + (size_t)ctx_id * N;
    int* const ro = d.roff + (size_t)ctx_id * (N + 1);
    uint8_t* const own = d.qown + (size_t)ctx_id * (p.NP >> 2);
    int base = 0;
    for (int t0 = 0; t0 < N; t0 += WG_WAVE) {
        const int t = t0 + lane;
        int len = 0;
        if (t < N) {
            const double dx = (t < WG_WAVE ? xp_l : d.x_pos[t]) - cx0, dy = (t < WG_WAVE ? yp_l : d.y_pos[t]) - cy0;
            const double dxm = xmax - (cx0 + dx * cth + dy * sth);
            const int jn_ = full_chains ? P_ : (int)(dxm / dpart) + 2;
            jneed_c[t] = jn_;
            len = (jn_ + 1 + 3) & ~3;
            if (len > P_) len = P_;
        }
        if (compact) {
            int incl = len;
#pragma unroll
            for (int o = 1; o < WG_WAVE; o <<= 1) {
                const int v = __shfl_up(incl, o, 64);
                if (lane >= o) incl += v;
            }
            if (t < N) {
                ro[t] = base + incl - len;
                // owner of every quad of ring slots: turbine t stamps its own quads (fire-and-forget byte stores; the
                // earlier binary search per quad was a chain of dependent loads on the launch's critical path)
                const int q0 = (base + incl - len) >> 2, q1 = (base + incl) >> 2;
                for (int q = q0; q < q1; ++q) own[q] = (uint8_t)t;
            }
            base += __shfl(incl, WG_WAVE - 1, 64);
        }
    }
    if (compact && lane == 0) ro[N] = base;
    WG_ICLK(6);
    const bool scripted = d.script_uvw != nullptr;
    if (lane == 0) {
        cx.dist = xmax - xmin;                                                     // :723-724
        const double t_inflow = (xmax - xmin) / ws;                                // :727
        cx.t_inflow = t_inflow;
        const int t_developed = (int)(t_inflow * 2);                               // :729
        cx.t_developed = t_developed;
        cx.time_max = p.never_truncate ? 9999999 : (int)(t_inflow * p.n_passthrough);   // :732
        int n_dev = (int)ceil((double)t_developed / p.dt_d - 1e-9);
        if (scripted) n_dev = 0;
        for (int f = f_lo; f < f_hi; ++f) {
            WgSlot& s = d.slot[ctx_id * F + f];
            s.head = P_ - 1; s.n_valid = 0; s.s_off = 0.0; s.time = 0.0; s.istep = 0; s.n_emitted = 0;
            s.dev_remaining = n_dev;
            s.fill_remaining = f == 0 ? p.fill_a : p.fill_b;
        }
    }
    for (int f = f_lo; f < f_hi; ++f) {
        const size_t tb = (size_t)(ctx_id * F + f) * N;
        const int cursor = scripted ? d.slot[ctx_id * F + f].cursor : 0;
        float* const u_ = d.u + tb; float* const v_ = d.v + tb; float* const w_ = d.w + tb;
        float* const ti_ = d.ti_loc + tb; float* const pw_ = d.power + tb; float* const ct_ = d.ct + tb;
        float4* const bnd_ = reinterpret_cast<float4*>(d.bnd) + tb;
        for (int t = lane; t < N; t += WG_WAVE) {
            float u = (float)ws, v = 0.f, w = 0.f, pw = 0.f;
            if (scripted) {
                int row = cursor < p.script_rows ? cursor : p.script_rows - 1;
                size_t sbase = (((size_t)f * p.script_rows + row) * p.B + e) * N;
                u = d.script_uvw[(sbase + t) * 3]; v = d.script_uvw[(sbase + t) * 3 + 1];
                w = d.script_uvw[(sbase + t) * 3 + 2]; pw = d.script_power[sbase + t];
            }
            u_[t] = u; v_[t] = v; w_[t] = w;
            ti_[t] = (float)ti; pw_[t] = pw; ct_[t] = 0.f;
            bnd_[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    WG_ICLK(7);
}

// ---------------------------------------------------------------------------------------------------
// MesClass (WindGym/MesClass.py)
// ---------------------------------------------------------------------------------------------------
// A sensor deque.  The turbine-level rings of a channel are stored TIME-major, [H][N]: one push of all N turbines is one
// contiguous store of N floats (turbine-major [N][H] made it N scattered 4-byte stores, i.e. N partial lines, per channel
// and step), and turbine t's samples are `stride` = N floats apart.  Farm-level rings: stride 1.
struct WgRing {
    const float* data;
    int n_pushed, hlen;
    int n_avail, start;      // start = physical slot of the oldest sample (one modulo per ring, not per element)
    int stride, cap;         // cap = physical slots of the ring (WgParams::ring_cap: the deque length, + 1 in sums mode)
    __device__ WgRing() {}
    __device__ WgRing(const float* d_, int n_pushed_, int hlen_, int stride_, int cap_)
        : data(d_), n_pushed(n_pushed_), hlen(hlen_), stride(stride_), cap(cap_) {
        n_avail = n_pushed < hlen ? n_pushed : hlen;
        start = (n_pushed - n_avail) % cap;
    }
    __device__ int avail() const { return n_avail; }
    __device__ float at(int q) const {   // q = 0 oldest
        int phys = start + q;
        if (phys >= cap) phys -= cap;
        return data[phys * stride];
    }
};

// sum of ring samples q = lo .. hi-1 in ascending order (bit-identical to the plain loop): the samples of a chunk of 8
// are requested together, then added one after the other — a plain `s += r.at(q)` loop is a chain of dependent
// LDS round trips (k_glue's observation phase: 25 + 10 of them per turbine for Env1.yaml).  Every index read is inside
// [lo, hi): surplus entries of the last chunk re-read sample hi-1 and are ignored.
__device__ inline float wg_ring_sum(const WgRing& r, const int lo, const int hi) {
    float s = 0.f;
    for (int q = lo; q < hi; q += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = r.at(min(q + k, hi - 1));
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (q + k < hi) s += v[k];
    }
    return s;
}

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
            const float s = wg_ring_sum(r, lo, hi);
            out[n++] = wg_scale(s / (float)(hi - lo), mn, rng);
        }
    }
    return n;
}

// turb_mes.calc_TI (MesClass.py:220-237), unscaled
__device__ inline float wg_calc_ti(const WgRing& r) {
    const int avail = r.avail();
    float U = wg_ring_sum(r, 0, avail);
    U /= (float)avail;
    float m2 = 0.f;
    for (int q = 0; q < avail; q += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = r.at(min(q + k, avail - 1));
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (q + k < avail) { const float dv = v[k] - U; m2 += dv * dv; }
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
                WgRing r = farm_level ? WgRing(fbase + p.fring_off[WG_CH_WS], n_pushed, H, 1, p.ring_cap[WG_CH_WS])
                                      : WgRing(rbase + p.ring_off[WG_CH_WS] + t, n_pushed, H, p.N, p.ring_cap[WG_CH_WS]);
                out[n++] = wg_scale(wg_calc_ti(r), p.ti_min_f, p.ti_rng_f);
            }
        }
        const int H = p.ch[ch].history_len;
        bool on = farm_level ? (p.farm_on[ch] != 0) : (p.turb_on[ch] != 0);
        WgRing r;
        if (farm_level) {
            // the farm object's yaw deque is never filled: the flags would allow it, the empty deque yields []
            r = WgRing(fbase + p.fring_off[ch], (ch == WG_CH_YAW) ? 0 : n_pushed, H, 1, p.ring_cap[ch]);
            if (ch == WG_CH_YAW) on = true;
        } else {
            r = WgRing(rbase + p.ring_off[ch] + t, n_pushed, H, p.N, p.ring_cap[ch]);
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
