// wg_flow.hip — k_flow, the dominant kernel of libwindgym_hip.so (gfx950, wave64).
//
// One workgroup per farm slot (env x ctx x farm); 64 / 128 / 256 threads chosen on the host from N x P (k_flow<NT,...>:
// each size has its own tuning — quads in flight per lane, occupancy target, record layout for the deficit gathers,
// chain pruning).  Which slot a workgroup serves is a hashed / reordered function of blockIdx (see k_flow) so that
// live farms and background episodes are spread over all XCDs, shader engines and CUs.  Per flow step it
//   (1) computes the emission record of every turbine (N threads),
//   (2) streams the slot's wake-particle SoA through HBM with 16-byte coalesced accesses: advection by the
//       Hill-vortex deflection speed + release of the new particles                        [HBM-bound part]
//   (3) phase A: every (target, source) turbine pair in parallel -> bracketing particles gathered (L2 hits:
//       the workgroup just wrote them), interpolated wake parameters staged in LDS,
//   (4) phase B: one thread per (target, rotor sample) sums the Gaussian deficits of all staged sources,
//       16-lane shuffles reduce the rotor average,
//   (5) power / Ct table lookup, measurement, sensor-ring push.
// Replaces DWMFlowSimulation.step() + rotor_avg_windspeed + power() + BasicControllers + _take_measurements
// + farm_mes.add_measurements (Wind_Farm_Env.py:480-495, 822-864, 943-979; BasicControllers.py:10-73;
// MesClass.py:568-591).  Model: DESIGN.md §2.
//
// Register discipline: the kernel takes a slim parameter block (FlowP) and keeps the per-turbine state in
// ONE LDS array of structs, so that the hot loops need few scalar registers (an earlier version with the
// full WgParams/WgPtrs by value spilled ~700 SGPRs to VGPR lanes and was VALU-bound).
#include <hip/hip_runtime.h>

#include "wg_flow_dev.h"
#include "wg_box_dev.h"
#include <type_traits>

struct __attribute__((aligned(8))) TurbLds {
    double xr, yr;
    float yaw, u, v, w, ti, pow, ct, cg, sg;
    float rct, rk, reps, rhv, rue;
    float sws, swd, syaw, sp;
    // conservative bounds over every particle this turbine has emitted in the episode (running maxima):
    // lateral (and vertical) excursion from the turbine's (y, z_hub), wake-growth rate k, initial width eps.
    // They let phase A discard a (target, source) pair BEFORE touching particle memory.
    float bd, bk, be;
    // compact rings (small-farm variant): offset / length of this turbine's ring inside the farm's particle arrays,
    // ring slot of its newest particle before (head) and after (head_n) the emissions of the current flow step
    int roff, rlen, head, head_n;
    // emission count (SlotRegs::n_emitted) right after this turbine's last emission of a MOVING record (hv != 0), 0 =
    // never: the chain holds a moving particle while fewer than rlen particles followed that one
    unsigned mvl;
};
static_assert(sizeof(TurbLds) == WG_TURB_LDS_BYTES, "keep WG_TURB_LDS_BYTES in sync");

struct SlotRegs {
    double s_off, time;
    int head, n_valid;
    unsigned istep;
    unsigned n_emitted;
};

// A farm's particle state in the compact-ring layout (small farms): SoA over the concatenated per-turbine rings of
// the farm slot, in global memory
struct PartLds {
    float* py; unsigned* ra; unsigned* rb;      // (interleaved record, FlowP::rec_il: rb == ra + 1 and particle i's words are ra[2 i], ra[2 i + 1])
    float *pz, *vl, *wl;        // turbulent inflow only
    const uint8_t* own;         // [L/4] owner turbine of a quad
    int L;                      // ring slots of the farm = roff[N]
    int il;                     // 1: interleaved record
    // the packed emission record (rec_a, rec_b) of ring slot i, whichever way it is stored
    __device__ __forceinline__ uint2 rec(const int i) const {
        return il ? *reinterpret_cast<const uint2*>(ra + 2 * i) : make_uint2(ra[i], rb[i]);
    }
};

// per-episode inflow context (uniform over the workgroup)
struct TurbCtx {
    uint32_t seed;
    double ox, oy;       // offset of the episode into the shared box (m)
    float sig;           // TI * U: standard deviation the unit-variance inflow is scaled to
    float alpha;         // low-pass coefficient of the meandering filter
    double ws;
    const float4* box4;  // the episode's box of the pool (fine / block-averaged copy)
    const float4* box4c;
};

// streamed particle state of the turbulent pass: touched once per launch -> non-temporal, so that it does not evict
// the meandering box from L2
#ifdef WG_NO_NT
#define WG_LDS(p) (*(p))
#define WG_STS(p, v) (*(p) = (v))
#else
#define WG_LDS(p) __builtin_nontemporal_load(p)
#define WG_STS(p, v) __builtin_nontemporal_store((v), (p))
#endif
#ifndef WG_NT_PY
#define WG_NT_PY 0
#endif
#ifndef WG_ABLATE
#define WG_ABLATE 0   // profiling only: 1 = no advection pass, 2 = no deficit phases
#endif

// One DWMFlowSimulation.step() of model M0 up to the new rotor inflow (T[t].u/v/w/ti); power and the
// measurement are done by the caller's per-turbine tail.
template <int NT, int TURB, bool RES, int SGM>
// (SGM = deficit model compiled in: 0 Gaussian, 1 super-Gaussian, 2 tabulated eddy-viscosity (Ainslie) deficit — a
// run-time switch in the pair loop cost the Gaussian default 5 us on cfg2)
// (the LDS arrays that lanes exchange data through — T, pair, tiap, tmask, jnl — are deliberately NOT __restrict__: for a
// noalias pointer the compiler may carry a value this lane loaded earlier across lds_barrier()'s memory clobber and miss
// what another lane stored in between; the read-only tables may be)
__device__ __forceinline__ void flow_step(const FlowP& p, const FlowPtrs& d, TurbLds* T,
                                          const float* __restrict__ tabct,
                                          const float* __restrict__ rdy, const float* __restrict__ rdz,
                                          float4* pair, float* tiap,
                                          unsigned* tmask, int* jnl,
                                          const size_t pbase, const double ws,
                                          const float ti_f, const float ti_pow, const TurbCtx& tc, SlotRegs& sr,
                                          const PartLds& pl, const bool first_step, int& add_acc) {
    const int tid = threadIdx.x;
    const int N = p.N, P = p.P;
    if (RES) __builtin_assume(N <= NT);
    const int TC = p.target_chunk;

    // travel of the chains over this step: particles released when the newest one is d_particle away
    const int head = sr.head, n_valid = sr.n_valid;
    double s_new = sr.s_off + ws * p.dt_d;
    int n_emit = 0;
    while (s_new >= p.dpart) { s_new -= p.dpart; ++n_emit; }
    if (n_emit > P) n_emit = P;
    int new_head = head + n_emit; if (new_head >= P) new_head -= P;
    int new_valid = n_valid + n_emit; if (new_valid > P) new_valid = P;
    const float s_off_f = (float)sr.s_off;

    // staging of the compact variants' deficit phases (the `pair` region, per chunk of TC targets):
    // cl[TC*N] u16 candidate list | def[TC*N] rotor-mean deficit | tiav[TC*N] added TI
    // (LF, the large-farm steady variant: staged per candidate, see lf_pair_phase)
    constexpr bool LF = NT == 256 && RES && TURB == WG_TURB_NONE && (WG_PAIR_FIRST != 0) && (WG_LF_PAIR != 0);
    unsigned short* cl = LF ? reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(pair) + WG_LF_OFF_CL(N)) : reinterpret_cast<unsigned short*>(pair);
    float* def = LF ? reinterpret_cast<float*>(reinterpret_cast<char*>(pair) + WG_LF_OFF_DEF(N)) : reinterpret_cast<float*>(cl + ((TC * N + 7) & ~7));
    float* tiav = LF ? def + p.lf_cap : def + TC * N;
    int* ncand = jnl + N + 1;
    // wake-added turbulence (wg_config.added_turbulence; turbulent inflow only): gadd[(t, sample)][3] =
    // the isotropic field at every rotor point, addv[3][TC * N] = a candidate pair's rotor-summed contribution
    const bool ADDED = TURB != WG_TURB_NONE && p.added != 0;
    float* addv = tiav + TC * N;
    float* gadd = reinterpret_cast<float*>(jnl + 4 * N + 4);

    // ---- single-wave steady compact variant (GL): the deficit phase's memory round trip is taken off the workgroup's
    // latency chain.  Candidate pairs and the ring slots that bracket them depend on the layout, the chain clocks and the
    // chains' running bounds only — all known before this step's records.  So: candidate pass -> the 8 gathers per
    // candidate are issued as LDS-DMA requests (global_load_lds: no result registers, the wave does not wait) -> records
    // and quad list (LDS / ALU work, ~5 k cycles = one loaded round trip) -> the deficit evaluation reads the landed
    // words from LDS.  (The bounds are those BEFORE this step's records: they cover every particle already in the
    // rings; a pair close enough to be bracketed by a particle released in this step is a candidate unconditionally.)
    constexpr bool PRE = RES && TURB == WG_TURB_NONE && (WG_PAIR_FIRST != 0);
    constexpr bool GL = PRE && NT == WG_WAVE && (WG_GLDS != 0);
    // interleaved record array (FlowP::rec_il; the host sets it for exactly these variants): a bracket pair is 16 contiguous
    // bytes of the array the advection pass streams
    constexpr bool IL = GL || LF;
    const float ws_f = (float)ws;
    const float ue_max = p.ue_scale * ws_f, ue_inv = __builtin_amdgcn_rcpf(ue_max);      // (scale of the record's u_e field)
    float* gat = reinterpret_cast<float*>(reinterpret_cast<char*>(pair) + p.lds_off_gat);
    typedef const __attribute__((address_space(1))) void* GPtr;
    typedef __attribute__((address_space(3))) void* LPtr;
    int gl_nc = 0;
    unsigned gl_mask = 0u;        // thread t < N: candidate sources of target t
    // bracket of candidate c (list entry cl[c]): target, source, distance, age j of the older... of the two particles AFTER the
    // step, interpolation weight; false if the chain has not reached the target yet
    auto gl_bracket = [&](const int c, int& i, int& tl, int& s2, double& dx, int& j, float& wgt) __attribute__((always_inline)) -> bool {
        i = cl[c];
        tl = (int)(((float)i + 0.5f) * p.inv_N);
        s2 = i - tl * N;
        dx = T[tl].xr - T[s2].xr;
        const double xi = (dx - s_new) * p.inv_dpart;
        const double jf = floor(xi);
        wgt = (float)(xi - jf);
        j = (int)jf;
        if (j < 0) { j = 0; wgt = 0.f; }
        return dx > 0.0 && j + 1 <= new_valid - 1;
    };
    // no particle in the chain's ring was released by a yawed turbine (pre-records state: this step's release is not in the
    // ring yet): every one of them still sits at the turbine's lateral position
    auto gl_resting = [&](const TurbLds& src) __attribute__((always_inline)) -> bool {
        return !(src.mvl != 0u && (int)(sr.n_emitted - src.mvl) < src.rlen);
    };
    // LDS-DMA gathers of candidate c: the record copies of the particles of pre-step age jp0 = j - n_emit and jp0 + 1 land at
    // gat4[l] and gat4[64 + l] (16 bytes per lane), their py at gat[512 + l] and gat[576 + l].  Lanes without a candidate / particles released in this
    // step (negative pre-step age: the turbine's record, not in memory yet) request a valid dummy address.
    // (the bracket — pair index, target, source, distance, age, weight, valid — stays in registers for the evaluation)
    int b_i = 0, b_tl = 0, b_s2 = 0, b_j = 0;
    double b_dx = 0.0;
    float b_wgt = 0.f;
    bool b_ok = false;
    auto gl_issue = [&](const int c, const int nc) __attribute__((always_inline)) -> bool {
        int i0 = 0, i1 = 0;
        bool rest = false;
        b_ok = c < nc && gl_bracket(c, b_i, b_tl, b_s2, b_dx, b_j, b_wgt);
        if (b_ok) {
            const int s2 = b_s2, j = b_j;
            const TurbLds& src = T[s2];
            const int Rs = src.rlen;
            rest = gl_resting(src);
            const int jp0 = j - n_emit, jp1 = jp0 + 1;
            int r0 = src.head - jp0; if (r0 < 0) r0 += Rs;
            int r1 = src.head - jp1; if (r1 < 0) r1 += Rs;
            if (jp0 >= 0) i0 = src.roff + r0;
            if (jp1 >= 0) i1 = src.roff + r1;
        }
        // (a resting chain's particles sit where they were released — at the turbine: their py is not fetched; the request
        // points into the rec_a line the lane fetches anyway and the consumer substitutes y_t)
        // (the records come from the interleaved (rec_a, rec_b) array the advection pass streams — the same lines; round 6: u_e
        // is in rec_b, the 16-byte gather copy of rounds 2-5 is gone.  4-byte requests: one word per lane and request.)
        const unsigned* q0p = pl.ra + 2 * i0;
        const unsigned* q1p = pl.ra + 2 * i1;
        __builtin_amdgcn_global_load_lds((GPtr)q0p, (LPtr)(gat + 0 * 64), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((GPtr)(q0p + 1), (LPtr)(gat + 1 * 64), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((GPtr)q1p, (LPtr)(gat + 2 * 64), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((GPtr)(q1p + 1), (LPtr)(gat + 3 * 64), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((GPtr)(rest ? (const float*)q0p : pl.py + i0), (LPtr)(gat + 4 * 64), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((GPtr)(rest ? (const float*)q1p : pl.py + i1), (LPtr)(gat + 5 * 64), 4, 0, 0);
        return rest;
    };
    bool gl_rest = false;
    if (GL) {
        // (a further flow step of the same launch gathers what the previous step's advection pass stored: the caller's loop
        // waits for the stores on its back edge)
        const float move_max = fabsf(p.hill) * ws_f * p.dt;
        // a target closer than this is bracketed by a particle released in this step
        const float near_f = (float)(sr.s_off + ws * p.dt_d + p.dpart) + 0.01f;
        const float* xf = reinterpret_cast<const float*>(jnl + 2 * N + 4);
        const float* yf = xf + N;
        WG_STAMP(4);
        const int npairs = N * N;
        const unsigned maskN = N >= 32 ? 0xffffffffu : (1u << N) - 1u;
        // four pairs per lane and trip: their tests are independent, so the LDS round trips of the four overlap (one pair
        // per trip was a chain of 3 dependent LDS reads per 64 pairs: 4.9 k cycles for cfg2's 256 pairs)
        for (int i0 = 0; i0 < ((WG_ABLATE & 2) ? 0 : npairs); i0 += 4 * NT) {
            bool cand[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * NT + tid;
                cand[u] = false;
                if (i < npairs) {
                    const int tg = (int)(((float)i + 0.5f) * p.inv_N);
                    const int s2 = i - tg * N;
                    // (float copies of the positions: their rounding, ~1e-3 m, is far inside the test's margin; the exact
                    // sign of dx is checked in double where the bracket is computed)
                    const float dxf = xf[tg] - xf[s2];
                    bool cd = (s2 != tg) && (dxf >= 0.f);
                    if (cd && !(dxf < near_f)) {
                        const TurbLds& src = T[s2];
                        const float sig_max = (src.bk * (dxf * p.inv_D) + src.be) * p.D;
                        const float bd = src.bd + (src.mvl != 0u ? move_max : 0.f);
                        const float gap = fabsf(yf[tg] - yf[s2]) - (p.R_rot + 5.0f * sig_max + bd);
                        cd = gap <= 1.0e-3f * p.D;
                    }
                    cand[u] = cd;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ib = i0 + u * NT, i = ib + tid;
                // (single wave: the list position is a running count in a register — and an LDS atomic would be ordered behind
                // every outstanding memory operation by the compiler once LDS-DMA requests exist in the kernel)
                const unsigned long long bal = __ballot(cand[u]);
                if (cand[u]) {
                    cl[gl_nc + __popcll(bal & ((1ull << (tid & 63)) - 1ull))] = (unsigned short)i;
                    def[i] = 0.f; tiav[i] = 0.f;      // a candidate the exact evaluation drops contributes zero
                }
                gl_nc += __popcll(bal);
                // thread t keeps the candidate sources of target t: its N bits of the ballots, no LDS masks
                const int lo = tid * N - ib;
                if (tid < N && lo < 64 && lo + N > 0)
                    gl_mask |= (unsigned)(lo >= 0 ? (bal >> lo) : (bal << -lo)) & maskN;
            }
        }
        lds_barrier<NT>();
        WG_STAMP(5);
        if (gl_nc > 0) gl_rest = gl_issue(tid, gl_nc);
        WG_STAMP(11);
    }

    // (1) emission records of this step, sin/cos of the yaw; clear the source masks
    for (int t = tid; t < N; t += NT) {
        TurbLds& q = T[t];
        const float g = q.yaw * WG_DEG2RAD_F;
        const float sg = __sinf(g), cg = __cosf(g);
        const float wsn = fmaxf(q.u * cg + q.v * sg, 0.0f);
        const float ctx = fminf(fmaxf(tab_lookup(tabct, p, wsn) * cg * cg, 0.0f), 0.96f);
        const float rq = __builtin_amdgcn_sqrtf(1.0f - ctx);
        const float beta = 0.5f * (1.0f + rq) * __builtin_amdgcn_rcpf(rq);
        q.rct = ctx;
        q.rk = p.ka * q.ti + p.kb;
        q.reps = p.eps0 * __builtin_amdgcn_sqrtf(beta);
        q.rhv = -p.hill * sg * q.u;
        // the packed record saturates outside [0, WG_K_MAX] x [-WG_HV_MAX, WG_HV_MAX]: never silently (wg_check reports it)
        q.rue = q.u;
        if (q.rk > WG_K_MAX || fabsf(q.rhv) > WG_HV_MAX || q.rue > ue_max) atomicOr(wg_cold_args()->d.status, WG_STATUS_BIT_RANGE);
        q.cg = cg;
        q.sg = sg;
        q.bk = fmaxf(q.bk, q.rk + WG_K_MAX / 65535.0f);
        q.be = fmaxf(q.be, q.reps + 1.0f / 65535.0f);
        if (RES) {
            {   // (q.head + n_emit) mod rlen without the integer division (n_emit is 0 or 1 in practice)
                int hn = q.head + n_emit;
                if (n_emit >= q.rlen) hn %= q.rlen; else if (hn >= q.rlen) hn -= q.rlen;
                q.head_n = hn;
            }
            if (n_emit > 0 && rec_moves(pack_b(q.rue * ue_inv, q.rhv))) q.mvl = sr.n_emitted + (unsigned)n_emit;
        }
    }
    if (!GL) for (int i = tid; i < TC * WG_MASK_WORDS; i += NT) tmask[i] = 0u;
    lds_barrier<NT>();
    WG_STAMP(2);

    // exact evaluation of one (target t, source s2) candidate from its two bracketing particles: interpolation, lateral
    // cut-off, deficit at the S rotor points -> def[i], tiav[i] (, addv) and the source's bit in the target's mask
    auto eval_pair = [&](const int i, const int tl, const int s2, const int t, const double dx, const float wgt,
                         const float py0, const float py1, const float pz0, const float pz1, const unsigned a0, const unsigned a1,
                         const unsigned b0_, const unsigned b1_) __attribute__((always_inline)) {
            const float u0 = rec_uf(b0_) * ue_max, u1 = rec_uf(b1_) * ue_max;      // (rotor wind speed at emission: in the record)
            const float w0 = 1.0f - wgt, w1 = wgt;
            const float yc = w0 * py0 + w1 * py1;
            float zc = p.hub;
            if (TURB != WG_TURB_NONE) zc = w0 * pz0 + w1 * pz1;
            const float kv = w0 * rec_k(a0) + w1 * rec_k(a1);
            const float epv = w0 * rec_eps(a0, p.eps0) + w1 * rec_eps(a1, p.eps0);
            const float xd = (float)dx * p.inv_D;
            const float sp = kv * xd + epv;
            const float sig = sp * p.D;
            const float yt = (float)T[t].yr;
            const float rc2 = (yt - yc) * (yt - yc) + (p.hub - zc) * (p.hub - zc);
            const float rcut = p.R_rot + 5.0f * sig;
            if (rc2 > rcut * rcut) return;
            const float ctv = w0 * rec_ct(a0) + w1 * rec_ct(a1);
            const float uev = w0 * u0 + w1 * u1;
            const float inv2s2 = __builtin_amdgcn_rcpf(2.0f * sig * sig);
            // super-Gaussian option (Blondel & Cathelain 2020): order n(x), centre-line deficit from
            // mass + momentum conservation; n = 2 is the Gaussian wake
            constexpr bool SG = SGM == 1;
            if constexpr (SGM == 2) {
                // Eddy-viscosity deficit (windgym_amd/ainslie.py): the fraction 1 - U / U0 tabulated over
                // (Ct, TI_amb, x / D, r / R), sampled 4-linearly — (Ct, TI, x) once per wake, r per rotor point.  TI is what the
                // particle's frozen wake-growth rate encodes: k = ka TI + kb.
                const float* __restrict__ tab = d.dtab;
                const float tiw = fmaxf((kv - p.kb) * p.an_inv_ka, 1e-6f);
                float fc = fminf(fmaxf((ctv - p.an_ct0) * p.an_inv_dct, 0.f), (float)(p.an_ct - 1));
                float ft = fminf(fmaxf((__logf(tiw) - p.an_lti0) * p.an_inv_dlti, 0.f), (float)(p.an_ti - 1));
                float fx = fminf(fmaxf(xd * p.an_inv_dx, 0.f), (float)(p.an_x - 1));
                const int ic = min((int)fc, p.an_ct - 2), it = min((int)ft, p.an_ti - 2), ix = min((int)fx, p.an_x - 2);
                const float wc = fc - (float)ic, wt = ft - (float)it, wx = fx - (float)ix;
                const int sx = p.an_r, st = p.an_x * sx, sc = p.an_ti * st;
                const int base = ic * sc + it * st + ix * sx;
                const float cgt_ = T[t].cg;
                const float ind_ = 0.5f * (1.0f - __builtin_amdgcn_sqrtf(1.0f - ctv));
                tiav[i] = p.no_ti_fold ? 0.f
                                       : p.tia * fast_pow(ind_, p.tib) * ti_pow * fast_pow(fmaxf(xd, 1.0f), p.tid) * __expf(-rc2 * inv2s2);
                float acc_ = 0.f, a0_ = 0.f, a1_ = 0.f, a2_ = 0.f;
                const float* gt_ = ADDED ? gadd + (t << p.S_shift) * 3 : nullptr;
                const float inv_R = 2.0f * p.inv_D;
                for (int sI = 0; sI < p.S; ++sI) {
                    const float dy = yt + rdy[sI] * cgt_ - yc, dz = p.hub + rdz[sI] - zc;
                    const float fr = __builtin_amdgcn_sqrtf(dy * dy + dz * dz) * inv_R * p.an_inv_dr;
                    if (!(fr < (float)p.an_r - 1.5f)) continue;        // beyond the table: no deficit
                    // the profile at the r nodes m - 1, m, m + 1 around fr (3-linear in Ct, TI, x; node -1 mirrors node 1)
                    const int m = (int)(fr + 0.5f), ml = m > 0 ? m - 1 : 1;
                    float fa = 0.f, fb = 0.f, fc_ = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int o = base + (q & 1 ? sc : 0) + (q & 2 ? st : 0) + (q & 4 ? sx : 0);
                        const float wq = (q & 1 ? wc : 1.0f - wc) * (q & 2 ? wt : 1.0f - wt) * (q & 4 ? wx : 1.0f - wx);
                        fa += wq * tab[o + ml]; fb += wq * tab[o + m];
                        if (ADDED) fc_ += wq * tab[o + m + 1]; else if (!(fr < (float)m)) fc_ += wq * tab[o + m + 1];
                    }
                    const float e = fr - (float)m;                       // in [-0.5, 0.5)
                    const float frac = e < 0.f ? fb + e * (fb - fa) : fb + e * (fc_ - fb);
                    const float du = uev * frac;
                    acc_ += du;
                    if (ADDED) {
                        // U k_mt = km1 dU + km2 R |d dU / dr|, the slope as the centred difference of the piecewise-linear
                        // profile over one node spacing (continuous in r): [f(fr + 1/2) - f(fr - 1/2)] / dr
                        const float hi = fb + (e + 0.5f) * (fc_ - fb), lo = fa + (e + 0.5f) * (fb - fa);
                        const float wk = uev * (p.km1 * frac + 0.5f * p.km2r * fabsf(hi - lo) * p.an_inv_dr * inv_R);
                        a0_ += wk * gt_[sI * 3]; a1_ += wk * gt_[sI * 3 + 1]; a2_ += wk * gt_[sI * 3 + 2];
                    }
                }
                if (ADDED) { addv[i] = a0_ * p.inv_S; addv[TC * N + i] = a1_ * p.inv_S; addv[2 * TC * N + i] = a2_ * p.inv_S; }
                def[i] = acc_ * p.inv_S;
                if (!GL && !LF) atomicOr(&tmask[tl * WG_MASK_WORDS + (s2 >> 5)], 1u << (s2 & 31));
                return;
            }
            float nsg = 2.0f, cf;
            if (SG) {
                nsg = p.sg_af * __expf(p.sg_bf * xd) + p.sg_cf;
                const float in2 = 2.0f * __builtin_amdgcn_rcpf(nsg);
                const float rad = exp2f(2.0f * in2 - 2.0f) - nsg * ctv * __builtin_amdgcn_rcpf(16.0f * tgammaf(in2) * fast_pow(sp, 2.0f * in2));
                cf = exp2f(in2 - 1.0f) - __builtin_amdgcn_sqrtf(fmaxf(rad, 0.0f));
            } else {
                cf = m0_cfrac(ctv, sp);
            }
            // wake-added turbulence (Crespo-Hernandez), Gaussian-weighted at the hub
            const float ind = 0.5f * (1.0f - __builtin_amdgcn_sqrtf(1.0f - ctv));
            tiav[i] = p.no_ti_fold ? 0.f
                                   : p.tia * fast_pow(ind, p.tib) * ti_pow * fast_pow(fmaxf(xd, 1.0f), p.tid) * __expf(-rc2 * inv2s2);
            // Gaussian deficit at the S rotor points of the target (lateral offsets scaled by cos(yaw_t))
            const float cgt = T[t].cg, amp = uev * cf;
            float acc = 0.f;
            if (ADDED) {
                // + this wake's share of the added turbulence: U k_mt g = dU (km1 + km2 R r / sigma^2) g at every point
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
                const float kg = p.km2r * inv2s2;
                const float* gt = gadd + (t << p.S_shift) * 3;
                const float inv2sp2 = __builtin_amdgcn_rcpf(2.0f * sp * sp), iD2 = p.inv_D * p.inv_D;
                for (int sI = 0; sI < p.S; ++sI) {
                    const float dy = yt + rdy[sI] * cgt - yc, dz = p.hub + rdz[sI] - zc;
                    const float r2 = dy * dy + dz * dz;
                    float du, grad2r;         // grad2r = km2 2R |d dU / dr| / dU
                    if (SG) {
                        const float rn = r2 > 0.f ? fast_pow(r2 * iD2, 0.5f * nsg) : 0.f;
                        du = amp * __expf(-rn * inv2sp2);
                        grad2r = r2 > 0.f ? 0.5f * p.km2r * nsg * rn * inv2sp2 * __builtin_amdgcn_rsqf(r2) : 0.f;
                    } else {
                        du = amp * __expf(-r2 * inv2s2);
                        grad2r = kg * __builtin_amdgcn_sqrtf(r2);
                    }
                    acc += du;
                    const float wk = du * (p.km1 + grad2r);
                    a0 += wk * gt[sI * 3]; a1 += wk * gt[sI * 3 + 1]; a2 += wk * gt[sI * 3 + 2];
                }
                addv[i] = a0 * p.inv_S; addv[TC * N + i] = a1 * p.inv_S; addv[2 * TC * N + i] = a2 * p.inv_S;
            } else if (SG) {
                const float inv2sp2 = __builtin_amdgcn_rcpf(2.0f * sp * sp), iD2 = p.inv_D * p.inv_D;
                for (int sI = 0; sI < p.S; ++sI) {
                    const float dy = yt + rdy[sI] * cgt - yc, dz = p.hub + rdz[sI] - zc;
                    const float r2 = dy * dy + dz * dz;
                    const float rn = r2 > 0.f ? fast_pow(r2 * iD2, 0.5f * nsg) : 0.f;
                    acc += amp * __expf(-rn * inv2sp2);
                }
            } else if (LF || (WG_S_UNROLL_ALL != 0)) {
                // (unrolled by 4: the LDS reads of the rotor-point offsets are in flight together — the rolled loop waits
                // for two dependent LDS reads per point; same additions in the same order)
#pragma unroll 4
                for (int sI = 0; sI < p.S; ++sI) {
                    const float dy = yt + rdy[sI] * cgt - yc, dz = p.hub + rdz[sI] - zc;
                    acc += amp * __expf(-(dy * dy + dz * dz) * inv2s2);
                }
            } else {
                for (int sI = 0; sI < p.S; ++sI) {
                    const float dy = yt + rdy[sI] * cgt - yc, dz = p.hub + rdz[sI] - zc;
                    acc += amp * __expf(-(dy * dy + dz * dz) * inv2s2);
                }
            }
            def[i] = acc * p.inv_S;
            if (!GL && !LF) atomicOr(&tmask[tl * WG_MASK_WORDS + (s2 >> 5)], 1u << (s2 & 31));
    };
    // (3)+(4) rotor-averaged inflow of the compact variants, as a closure: the steady variant runs it BEFORE the advection
    // pass (PRE), the turbulent ones after it.
    // PRE (steady inflow): the bracketing particles are gathered in their PRE-step state and advanced by the same
    // m0_advect() the advection pass applies (a particle's step is a function of its own record and age only; a particle
    // emitted in this step is the turbine's record in LDS).  The phase then depends on nothing the advection pass writes:
    // no wait for the pass's stores, the gathers are in flight before the streaming loads of the same lines (which then
    // hit in L2 instead of the other way round, after L2 has lost them), and a live workgroup's latency chain is three
    // memory round trips (state, gathers, stream) instead of five.
    auto res_pair_phase = [&](auto pre_tag) __attribute__((always_inline)) {
        constexpr bool PREV = decltype(pre_tag)::value;
        const float move_max = fabsf(p.hill) * ws_f * p.dt;
        const float* xf_ = reinterpret_cast<const float*>(jnl + 2 * N + 4);
        const float* yf_ = xf_ + N;
        // small-farm variant: PAIR-major.  In a small farm only a few (target, source) pairs interact (the same
        // column of a grid, its diagonal neighbours), so the pairs that pass a cheap conservative test are compacted
        // into a list and only those get the exact evaluation — one thread per candidate: bracketing particles
        // (L2 hits: this workgroup just streamed them), interpolation, lateral cut-off, then the Gaussian deficit at all S rotor points summed in the thread.
        // Thread t finally subtracts its sources' contributions in ascending source order (deterministic).
        // Staging (the `pair` region, per chunk of TC targets): cl[TC*N] u16 candidate list | def[TC*N] rotor-mean deficit | tiav[TC*N] added TI.
        // ambient inflow at the rotors (no wakes): thread t / the (t, sample) threads
        if (TURB == WG_TURB_BOX) {
            const int nitems = N << p.S_shift;
            for (int it = tid; it < ((nitems + NT - 1) & ~(NT - 1)); it += NT) {
                const int t = it >> p.S_shift, s = it & (p.S_pad - 1);
                const bool live = (it < nitems) && (s < p.S);
                float amb[3] = {0.f, 0.f, 0.f};
                if (live) {
                    const double bx = T[t].xr - tc.ws * sr.time + tc.ox;
                    const double by = T[t].yr + (double)(rdy[s] * T[t].cg) + tc.oy, bz = p.hub_d + (double)rdz[s];
                    if (p.box_pow2) box_lookup<true>(tc.box4, p, bx, by, bz, amb);
                    else box_lookup<false>(tc.box4, p, bx, by, bz, amb);
                }
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
                    for (int o = p.S_pad >> 1; o > 0; o >>= 1) amb[cc] += __shfl_xor(amb[cc], o, 64);
                if (live && s == 0) {
                    T[t].u = ws_f + tc.sig * amb[0] * p.inv_S; T[t].v = tc.sig * amb[1] * p.inv_S; T[t].w = tc.sig * amb[2] * p.inv_S;
                }
            }
        } else {

            for (int t = tid; t < N; t += NT) {
                float au = 0.f, av = 0.f, aw = 0.f;
                if (TURB == WG_TURB_RANDOM) {
                    // i.i.d. gusts at the S rotor points: their mean is one normal of variance sigma^2 / S
                    const float sc = tc.sig * p.inv_sqrt_S;
                    au = sc * wg_turb_normal(tc.seed, sr.istep, (uint32_t)t, 0u, 0x52u);
                    av = sc * wg_turb_normal(tc.seed, sr.istep, (uint32_t)t, 1u, 0x52u);
                    aw = sc * wg_turb_normal(tc.seed, sr.istep, (uint32_t)t, 2u, 0x52u);
                }
                T[t].u = ws_f + au; T[t].v = av; T[t].w = aw;
            }
        }
        // targets in chunks of TC (small farms: one chunk = all N x N pairs)
        for (int t0 = 0; t0 < N; t0 += TC) {
        const int nt = (N - t0) < TC ? (N - t0) : TC;
        const int npairs = nt * N;
        if (t0 > 0) lds_barrier<NT>();          // the previous chunk's sums are done with the staging arrays
        if (tid == 0) *ncand = 0;
        for (int i = tid; i < nt * WG_MASK_WORDS; i += NT) tmask[i] = 0u;
        lds_barrier<NT>();
        // pass 1: conservative test on every pair -> candidate list
        for (int i0 = 0; i0 < ((WG_ABLATE & 2) ? 0 : npairs); i0 += NT) {
            const int i = i0 + tid;
            bool cand = false;
            if (i < npairs) {
                const int tl = (int)(((float)i + 0.5f) * p.inv_N);
                const int s2 = i - tl * N;
                const int tg = t0 + tl;
                // (float copies of the rotated positions: their rounding, ~1e-3 m, is far inside the test's margin; the
                // exact sign of dx is decided in double by the evaluation below — cfg3 walks 6400 pairs per flow step here)
                const float dxf = xf_[tg] - xf_[s2];
                cand = (s2 != tg) && (dxf >= 0.f);
                if (cand) {
                    const TurbLds& src = T[s2];
                    const float sig_max = (src.bk * (dxf * p.inv_D) + src.be) * p.D;
                    // PRE: src.bd is the excursion bound BEFORE this step's advection; a particle moves by
                    // |hv C| dt <= |hill| u_e dt <= |hill| U dt in one step (steady inflow: u_e <= U, C <= 1)
                    const float bd = PREV ? src.bd + (src.mvl != 0u ? move_max : 0.f) : src.bd;
                    const float gap = fabsf(yf_[tg] - yf_[s2]) - (p.R_rot + 5.0f * sig_max + bd);
                    cand = gap <= 1.0e-3f * p.D;      // small margin for fp32 rounding of the bound itself
                }
            }
            const unsigned long long bal = __ballot(cand);
            if (bal) {
                const int lane = tid & 63;
                int base = 0;
                if (lane == 0) base = atomicAdd(ncand, __popcll(bal));
                base = __shfl(base, 0, 64);
                if (cand) cl[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)i;
            }
        }
        lds_barrier<NT>();
        const int nc = *ncand;
        if (ADDED && nc > 0) {
            // the isotropic field at the rotor points of the chunk's targets that have a candidate source (a free-stream
            // rotor needs none): "Synchronized" = the ambient field's clock and offset.  Flags: tmask word 3 is free in
            // this phase for N <= 96; kept simple: one pass over the candidate list marks gflag[t]
            int* gflag = reinterpret_cast<int*>(gadd + 3 * (N << p.S_shift));
            for (int tl = tid; tl < nt; tl += NT) gflag[tl] = 0;
            lds_barrier<NT>();
            for (int c = tid; c < nc; c += NT) gflag[(int)(((float)cl[c] + 0.5f) * p.inv_N)] = 1;
            lds_barrier<NT>();
            const int nitems = nt << p.S_shift;
            for (int it = tid; it < nitems; it += NT) {
                const int tl = it >> p.S_shift, s = it & (p.S_pad - 1);
                if (s >= p.S || !gflag[tl]) continue;
                const int t = t0 + tl;
#ifndef WG_NO_ADD_COUNT
                ++add_acc;                    // (roofline accounting: one 8-corner lookup of the isotropic box)
#endif
                float g3[3];
                abox_lookup(p, d, T[t].xr - tc.ws * sr.time + tc.ox, T[t].yr + (double)(rdy[s] * T[t].cg) + tc.oy,
                            p.hub_d + (double)rdz[s], g3);
                float* gp = gadd + ((t << p.S_shift) + s) * 3;
                gp[0] = g3[0]; gp[1] = g3[1]; gp[2] = g3[2];
            }
            lds_barrier<NT>();
        }
        // pass 2: exact evaluation of the candidates
        for (int c = tid; c < nc; c += NT) {
            const int i = cl[c];
            const int tl = (int)(((float)i + 0.5f) * p.inv_N);
            const int s2 = i - tl * N;
            const int t = t0 + tl;
            const TurbLds& src = T[s2];
            const double dx = T[t].xr - src.xr;
            if (!(dx > 0.0)) continue;                    // (the candidate test ran on float positions)
            const double xi = (dx - s_new) * p.inv_dpart;
            const double jf = floor(xi);
            float wgt = (float)(xi - jf);
            int j = (int)jf;
            if (j < 0) { j = 0; wgt = 0.f; }
            if (j + 1 > new_valid - 1) continue;          // the chain has not reached the target yet
            const int Rs = src.rlen;
            float py0, py1, pz0 = 0.f, pz1 = 0.f;
            unsigned a0, a1, b0_, b1_;
            if (PREV) {
                // ages j, j + 1 after the step are ages jp, jp + 1 = j - n_emit, ... before it (negative: released in this
                // step — the turbine's record, at the turbine)
                const int jp0 = j - n_emit, jp1 = jp0 + 1;
                int r0 = src.head - jp0; if (r0 < 0) r0 += Rs;
                int r1 = src.head - jp1; if (r1 < 0) r1 += Rs;
                if (jp0 < 0) r0 = 0;      // (released in this step: nothing to fetch — any slot of the ring will do)
                if (jp1 < 0) r1 = 0;
                const int i0 = src.roff + r0, i1 = src.roff + r1;
                const bool g0 = jp0 >= 0, g1 = jp1 >= 0;
                py0 = py1 = (float)src.yr;
                a0 = a1 = pack_a(src.rct, src.rk); b0_ = b1_ = pack_b(src.rue * ue_inv, src.rhv);
                {
                    // (both brackets requested together, whether needed or not — i0 / i1 are valid ring slots either way:
                    // loads under `if (g0)` / `if (g1)` came out as two round trips, one behind the other)
                    const uint2 q0 = pl.rec(i0), q1 = pl.rec(i1);
                    const float y0l = pl.py[i0], y1l = pl.py[i1];
                    if (g0) { a0 = q0.x; b0_ = q0.y; py0 = y0l; }
                    if (g1) { a1 = q1.x; b1_ = q1.y; py1 = y1l; }
                }
                if (g0 && jp0 < n_valid) py0 = m0_advect(py0, a0, b0_, jp0, s_off_f, p.dpart_f, p.inv_D, p.dt, p.eps0);
                if (g1 && jp1 < n_valid) py1 = m0_advect(py1, a1, b1_, jp1, s_off_f, p.dpart_f, p.inv_D, p.dt, p.eps0);
            } else {
                int r0 = src.head_n - j; if (r0 < 0) r0 += Rs;
                int r1 = r0 - 1; if (r1 < 0) r1 += Rs;
                const int i0 = src.roff + r0, i1 = src.roff + r1;
                py0 = pl.py[i0]; py1 = pl.py[i1];
                if (TURB != WG_TURB_NONE) { pz0 = pl.pz[i0]; pz1 = pl.pz[i1]; }
                {
                    const uint2 g0 = pl.rec(i0), g1 = pl.rec(i1);
                    a0 = g0.x; b0_ = g0.y; a1 = g1.x; b1_ = g1.y;
                }
            }
            eval_pair(i, tl, s2, t, dx, wgt, py0, py1, pz0, pz1, a0, a1, b0_, b1_);
        }
        lds_barrier<NT>();
        // thread t: superposition in ascending source order
        for (int tl = tid; tl < nt; tl += NT) {
            float dsum = 0.f, tia_max = 0.f, au = 0.f, av = 0.f, aw = 0.f;
            for (int wd = 0; wd * 32 < N; ++wd) {
                unsigned m = tmask[tl * WG_MASK_WORDS + wd];
                while (m) {
                    const int s2 = wd * 32 + __builtin_ctz(m);
                    m &= m - 1;
                    dsum += def[tl * N + s2];
                    tia_max = fmaxf(tia_max, tiav[tl * N + s2]);
                    if (ADDED) { au += addv[tl * N + s2]; av += addv[TC * N + tl * N + s2]; aw += addv[2 * TC * N + tl * N + s2]; }
                }
            }
            if (ADDED) { T[t0 + tl].u += au; T[t0 + tl].v += av; T[t0 + tl].w += aw; }
            T[t0 + tl].u -= dsum;
            T[t0 + tl].ti = __builtin_amdgcn_sqrtf(ti_f * ti_f + tia_max * tia_max);
        }
        }   // target chunks
        lds_barrier<NT>();
    };

    // Large-farm steady variant (LF; cfg3: 80 turbines, 6400 pairs, ~520 of them interacting): the whole farm in one go.
    //  (1) per-target masks of candidate sources: a lane holds one SOURCE (position, 5 bk, lateral reach) in registers and
    //      walks the targets — the ballot of the conservative test is 64 bits of the target's mask; no LDS atomics, and
    //      no LDS traffic beyond the targets' positions (per-pair reads of packed sources ran into the LDS bandwidth of a CU
    //      shared by 16 waves);
    //  (2) thread t: popcount -> exclusive prefix over the targets (wave scans + 4 wave totals) -> t writes its candidates
    //      into the list at its offset: the list is in ascending (target, source) order by construction;
    //  (3) one thread per candidate: brackets, gathers, exact evaluation -> def[c], tiav[c] (staged per candidate, lf_cap
    //      at a time: the dense per-chunk staging of res_pair_phase needed 3 target chunks on cfg3, each with its own
    //      gather round trip, five barriers and same-address LDS atomics);
    //  (4) thread t sums its slice of the list in order — the same additions in the same order as the chunked phase.
    // (Tried and dropped: half of the workgroups running the phase AFTER the advection pass, on post-step brackets, so that
    // the streaming pass of one workgroup overlaps the ALU / LDS work of another on the same CU — 3.85 M env-steps/s in
    // either order, 3.77 - 3.93 M mixed by workgroup parity / by blocks of 256: within run-to-run spread.)
    auto lf_pair_phase = [&]() __attribute__((always_inline)) {
        const float move_max = fabsf(p.hill) * ws_f * p.dt;
        const float* xf_ = reinterpret_cast<const float*>(jnl + 2 * N + 4);
        const float* yf_ = xf_ + N;
        unsigned* tm = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(pair) + WG_LF_OFF_TM(N));
        int* base = reinterpret_cast<int*>(reinterpret_cast<char*>(pair) + WG_LF_OFF_BASE(N));
        int* wtot = jnl;                          // (jnl[0..3]: unused by the compact variants)
        const int W = (N + 31) >> 5;
        for (int t = tid; t < N; t += NT) { T[t].u = ws_f; T[t].v = 0.f; T[t].w = 0.f; }
        {   // (1) lane = source (its packed bound in registers), loop over targets: the ballot of the test IS two words of
            // the target's mask.  64 sources per wave; the waves that share a source block take every other target.
            const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
            const int nsb = (N + 63) >> 6, per = (NT / 64) / nsb;      // source blocks (1 or 2), waves per block
            const int sb = wv % nsb, s2 = 64 * sb + lane;
            const bool sv = s2 < N;
            const TurbLds& src = T[sv ? s2 : 0];
            // PRE: src.bd is the excursion bound BEFORE this step's advection (see res_pair_phase)
            const float bd = src.bd + (src.mvl != 0u ? move_max : 0.f);
            const float sx = xf_[sv ? s2 : 0], sy = yf_[sv ? s2 : 0], sz = 5.0f * src.bk;
            const float sw = p.R_rot + 5.0f * src.be * p.D + bd + 1.0e-3f * p.D;
            // the targets' positions in registers too (lane l: targets l and 64 + l), read back with v_readlane — the loop
            // touches no LDS; lane k keeps the ballot of the wave's k-th target and stores it afterwards
            const int ta = __float_as_int(xf_[min(lane, N - 1)]), tb = __float_as_int(yf_[min(lane, N - 1)]);
            const int tc_ = __float_as_int(xf_[min(64 + lane, N - 1)]), td = __float_as_int(yf_[min(64 + lane, N - 1)]);
            const int t_first = wv / nsb;
            unsigned long long mybal = 0ull;
            int k = 0;
            // |dy| <= R + 5 sigma_max + bd + margin, sigma_max = bk dx + be D  (float positions: the exact sign of dx is decided
            // in double by the evaluation).  Branch-free: `&` on the predicates, one loop per half of the target registers.
            auto test = [&](const int t, const int xi_, const int yi_) __attribute__((always_inline)) {
                const float xt = __int_as_float(xi_), yt = __int_as_float(yi_);
                const float dxf = xt - sx;
                const bool ok = sv & (dxf >= 0.f) & (fabsf(yt - sy) - sz * dxf <= sw) & (s2 != t);
                const unsigned long long bal = (WG_ABLATE & 2) ? 0ull : __ballot(ok);      // (profiling builds: no candidates)
                mybal = lane == k ? bal : mybal;
            };
            int t = t_first;
            const int n_lo = min(N, 64);
            for (; t < n_lo; t += per, ++k) test(t, __builtin_amdgcn_readlane(ta, t), __builtin_amdgcn_readlane(tb, t));
            for (; t < N; t += per, ++k) test(t, __builtin_amdgcn_readlane(tc_, t - 64), __builtin_amdgcn_readlane(td, t - 64));
            const int tk = t_first + lane * per;
            if (tk < N) { tm[tk * 4 + 2 * sb] = (unsigned)mybal; tm[tk * 4 + 2 * sb + 1] = (unsigned)(mybal >> 32); }
        }
        lds_barrier<NT>();
        WG_STAMP(14);
        {   // (2)
            int cnt = 0;
            if (tid < N) for (int w = 0; w < W; ++w) cnt += __popc(tm[tid * 4 + w]);
            int inc = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if ((tid & 63) >= o) inc += v; }
            if ((tid & 63) == 63) wtot[tid >> 6] = inc;
            lds_barrier<NT>();
            WG_STAMP(15);
            int off = 0;
            for (int w = 0; w < (tid >> 6); ++w) off += wtot[w];
            if (tid < N) {
                int o = off + inc - cnt;
                base[tid] = o;
                if (tid == N - 1) base[N] = off + inc;
                for (int w = 0; w < W; ++w) {
                    unsigned m = tm[tid * 4 + w];
                    while (m) { cl[o++] = (unsigned short)((tid << 7) | (32 * w + __builtin_ctz(m))); m &= m - 1; }
                }
            }
        }
        lds_barrier<NT>();
        const int nc = base[N];
        WG_STAMP(5);
        float dsum = 0.f, tia_max = 0.f;              // thread t's running sums over the rounds
        for (int c0 = 0; c0 < nc; c0 += p.lf_cap) {
            const int c1 = min(nc, c0 + p.lf_cap);
            // (3) software-pipelined: a thread's NEXT candidate is decoded and its gathers requested before the current one
            // is evaluated (cfg3: ~520 candidates on 256 threads — two or three trips, whose gather round trips would
            // otherwise add up)
            struct Cand { int t, s2, jp0; float wgt; double dx; uint2 q0, q1; float y0, y1; bool ok, rest; };
            auto lf_issue = [&](Cand& k, const int c) __attribute__((always_inline)) {
                k.ok = false;
                if (c >= c1) return;
                const int slot = c - c0;
                def[slot] = 0.f; tiav[slot] = 0.f;
                const unsigned ent = cl[c];
                k.t = (int)(ent >> 7); k.s2 = (int)(ent & 127u);
                const TurbLds& src = T[k.s2];
                k.dx = T[k.t].xr - src.xr;
                if (!(k.dx > 0.0)) return;                    // (the candidate test ran on float positions)
                const double xi = (k.dx - s_new) * p.inv_dpart;
                const double jf = floor(xi);
                k.wgt = (float)(xi - jf);
                int j = (int)jf;
                if (j < 0) { j = 0; k.wgt = 0.f; }
                if (j + 1 > new_valid - 1) return;            // the chain has not reached the target yet
                const int Rs = src.rlen;
                // ages j, j + 1 after the step are ages jp, jp + 1 = j - n_emit, ... before it (negative: released in this
                // step — the turbine's record, at the turbine)
                k.jp0 = j - n_emit;
                int r0 = src.head - k.jp0; if (r0 < 0) r0 += Rs;
                int r1 = src.head - k.jp0 - 1; if (r1 < 0) r1 += Rs;
                if (k.jp0 < 0) r0 = 0;          // (released in this step: nothing to fetch — any slot of the ring will do)
                if (k.jp0 + 1 < 0) r1 = 0;
                const int i0 = src.roff + r0, i1 = src.roff + r1;
                k.q0 = pl.rec(i0); k.q1 = pl.rec(i1);      // (the interleaved record the advection pass streams: the pair is 16 contiguous bytes)
                // (a resting chain's particles sit where they were released — at the turbine — and do not move in this step either:
                // their py is not fetched, as in the single-wave variants; the baseline farms' workgroups gather one line per
                // candidate instead of two)
                k.rest = gl_resting(src);
                k.y0 = 0.f; k.y1 = 0.f;
                if (!k.rest) { k.y0 = pl.py[i0]; k.y1 = pl.py[i1]; }
                k.ok = true;
            };
            Cand nxt;
            lf_issue(nxt, c0 + tid);
            for (int c = c0 + tid; c < c1; c += NT) {
                const Cand k = nxt;
                lf_issue(nxt, c + NT);
                if (!k.ok) continue;
                const TurbLds& src = T[k.s2];
                const int jp0 = k.jp0, jp1 = jp0 + 1;
                float py0 = k.rest ? (float)src.yr : k.y0, py1 = k.rest ? (float)src.yr : k.y1;
                unsigned a0 = k.q0.x, b0_ = k.q0.y, a1 = k.q1.x, b1_ = k.q1.y;
                if (jp0 < 0) { py0 = (float)src.yr; a0 = pack_a(src.rct, src.rk); b0_ = pack_b(src.rue * ue_inv, src.rhv); }
                if (jp1 < 0) { py1 = (float)src.yr; a1 = pack_a(src.rct, src.rk); b1_ = pack_b(src.rue * ue_inv, src.rhv); }
                if (!k.rest && jp0 >= 0 && jp0 < n_valid) py0 = m0_advect(py0, a0, b0_, jp0, s_off_f, p.dpart_f, p.inv_D, p.dt, p.eps0);
                if (!k.rest && jp1 >= 0 && jp1 < n_valid) py1 = m0_advect(py1, a1, b1_, jp1, s_off_f, p.dpart_f, p.inv_D, p.dt, p.eps0);
                if (c == 0) WG_STAMP(12);
                eval_pair(c - c0, 0, k.s2, k.t, k.dx, k.wgt, py0, py1, 0.f, 0.f, a0, a1, b0_, b1_);
                if (c == 0) WG_STAMP(13);
            }
            lds_barrier<NT>();
            if (c0 == 0) WG_STAMP(11);
            if (tid < N) {                                    // (4)
                const int b0 = max(base[tid], c0), b1 = min(base[tid + 1], c1);
                for (int c = b0; c < b1; ++c) {
                    dsum += def[c - c0];
                    tia_max = fmaxf(tia_max, tiav[c - c0]);
                }
            }
            if (c1 < nc) lds_barrier<NT>();                   // (the next round lands in the same words)
        }
        if (tid < N) {
            T[tid].u -= dsum;
            T[tid].ti = __builtin_amdgcn_sqrtf(ti_f * ti_f + tia_max * tia_max);
        }
        lds_barrier<NT>();
    };
    if (PRE && !GL) {
        // (a further flow step of the same launch — background development, reset — gathers what the previous step's
        // advection pass stored)
        if (!first_step) full_barrier<NT>();
        if constexpr (LF) lf_pair_phase(); else res_pair_phase(std::true_type{});
        WG_STAMP(9);
    }

    // (2) pass over the particle SoA: advect over dt, release the new particles

    float* __restrict__ gpy = d.py + pbase;
    unsigned* __restrict__ gra = d.rec_a + pbase;
    unsigned* __restrict__ grb = d.rec_b + pbase;
    // large farms (256-thread workgroups): the frozen record is gathered from a 16-byte AoS copy, see phase A
    // chain pruning is compiled into the large-farm variant only: there the advection pass is traffic-bound and a
    // skipped quad saves real time; in the small-farm variants the extra predicate cost registers (spills at the 96
    // VGPR budget: +26 % on cfg2) and was measured slower even with two thirds of the particles skipped
    constexpr bool PRUNE = (NT == 256);
    if (RES && TURB == WG_TURB_NONE) {
        // compact rings, steady inflow: the unit of work is a quad of 4 consecutive ring slots (one owner turbine per
        // quad: ring lengths are multiples of 4).  Only chains that hold a particle of a yawed turbine move (hv != 0),
        // and whether a chain does is known WITHOUT reading it: TurbLds::mvl remembers the emission count at the
        // chain's last moving emission, and that particle is still in the ring while fewer than R_t particles
        // followed it.  Pass A (LDS only): turbine t lists its quads — all of them if the chain may move, otherwise
        // just the quads that receive this step's new particles.  Pass B: one listed quad per lane, its three words
        // (py, rec_a, rec_b) requested together — one memory round trip per 64 quads, none for a resting chain.
        // (16-bit list entries: turbine << ql_shift | quad index inside the turbine's ring; the host selects this variant
        // only where both fit)
        unsigned short* ql = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(pair) + (GL ? p.lds_off_ql : 0));
        const int qsh = p.ql_shift;
        // (not GL: the same word serves as the candidate counter of the deficit phase, the list aliases its staging)
        int* nq = jnl + N + 1;
        // (2^lsh adjacent lanes share a turbine: the listing loop of a moving chain — up to P / 4 entries — is split
        // between them; the host chose lsh with N << lsh <= NT)
        const int lsh = p.ql_lpt_shift, lpt = 1 << lsh;
        const int t = tid >> lsh, kl = tid & (lpt - 1);
        int gl_nlist = 0;
        if (GL) {
            // single wave: list positions from a wave-level prefix sum, no LDS atomics — the compiler orders an LDS
            // atomic behind every pending LDS-DMA request (s_waitcnt vmcnt(0)), which would put the gathers' round trip
            // back on the chain right here
            int cnt = 0, nqd = 0;
            bool full = false;
            unsigned tag = 0u;
            if (t < ((WG_ABLATE & 1) ? 0 : N)) {
                const TurbLds& tq = T[t];
                const int R = tq.rlen;
                nqd = R >> 2;
                const bool moving = tq.mvl != 0u && (int)(sr.n_emitted - tq.mvl) < R;
                tag = (unsigned)t << qsh;
                full = moving || n_emit >= 4 || n_emit >= R;
                if (full) cnt = nqd;
                else if (kl == 0) {
                    // A resting chain only receives this step's new particles (at most 3 here).  They are written straight to
                    // their ring slots — py and the interleaved record: two small stores per particle, no load —
                    // instead of listing their quads for the advection pass, which read-modify-writes whole quads (two or three
                    // lines fetched to change 28 bytes).  The slots hold the chain's oldest particles (pre-step ages >= R -
                    // n_emit), which no bracket of this step can touch, so the stores need not wait for the gathers in flight.
                    const float y0 = (float)tq.yr;
                    const unsigned na = pack_a(tq.rct, tq.rk), nb = pack_b(tq.rue * ue_inv, tq.rhv);
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        if (e < n_emit) {
                            int r = tq.head + 1 + e; if (r >= R) r -= R;
                            const int ix = tq.roff + r;
                            pl.py[ix] = y0;
                            reinterpret_cast<uint2*>(pl.ra)[ix] = make_uint2(na, nb);
                        }
                    }
                }
            }
            const int mine = kl == 0 ? cnt : 0;
            int inc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if ((tid & 63) >= o) inc += v; }
            gl_nlist = __shfl(inc, 63, 64);
            const int base = __shfl(inc - mine, (tid & 63) & ~(lpt - 1), 64);
            if (full) {
                for (int i = kl; i < nqd; i += lpt) ql[base + i] = (unsigned short)(tag | (unsigned)i);
            }
        } else if (LF) {
            // large farms: list offsets from a prefix sum over the turbines (wave scans + the wave totals in jnl[0..3], which the
            // compact variants do not use otherwise) — up to 40 lanes of a wave hitting ONE LDS counter with atomicAdd are
            // serialised one by one
            int cnt = 0, nqd = 0;
            bool full = false;
            unsigned tag = 0u;
            if (t < ((WG_ABLATE & 1) ? 0 : N)) {
                const TurbLds& tq = T[t];
                const int R = tq.rlen;
                nqd = R >> 2;
                const bool moving = tq.mvl != 0u && (int)(sr.n_emitted - tq.mvl) < R;
                tag = (unsigned)t << qsh;
                full = moving || n_emit >= 4 || n_emit >= R;
                if (full) cnt = nqd;
                else if (kl == 0) {
                    // (a resting chain only receives this step's new particles: stored straight to their ring slots, as in the
                    // single-wave variant above — no load, no trip through the pass; cfg3: +4 %, the baseline farms' workgroups
                    // are the tail of the launch and their pass was one exposed round trip)
                    const float y0 = (float)tq.yr;
                    const unsigned na = pack_a(tq.rct, tq.rk), nb = pack_b(tq.rue * ue_inv, tq.rhv);
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        if (e < n_emit) {
                            int r = tq.head + 1 + e; if (r >= R) r -= R;
                            const int ix = tq.roff + r;
                            pl.py[ix] = y0;
                            reinterpret_cast<uint2*>(pl.ra)[ix] = make_uint2(na, nb);
                        }
                    }
                }
            }
            const int mine = kl == 0 ? cnt : 0;
            int inc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if ((tid & 63) >= o) inc += v; }
            if ((tid & 63) == 63) jnl[tid >> 6] = inc;
            lds_barrier<NT>();
            int woff = 0;
            for (int w = 0; w < (tid >> 6); ++w) woff += jnl[w];
            if (tid == NT - 1) *nq = woff + inc;
            const int base = woff + __shfl(inc - mine, (tid & 63) & ~(lpt - 1), 64);
            if (full) {
                for (int i = kl; i < nqd; i += lpt) ql[base + i] = (unsigned short)(tag | (unsigned)i);
            }
        } else {
        if (tid == 0) *nq = 0;
        lds_barrier<NT>();
        if (t < ((WG_ABLATE & 1) ? 0 : N)) {
            const TurbLds& tq = T[t];
            const int R = tq.rlen, nqd = R >> 2;
            const bool moving = tq.mvl != 0u && (int)(sr.n_emitted - tq.mvl) < R;
            const unsigned tag = (unsigned)t << qsh;
            if (moving || n_emit >= 4 || n_emit >= R) {
                int base = 0;
                if (kl == 0) base = atomicAdd(nq, nqd);
                base = __shfl(base, (tid & 63) & ~(lpt - 1), 64);
                for (int i = kl; i < nqd; i += lpt) ql[base + i] = (unsigned short)(tag | (unsigned)i);
            } else if (n_emit > 0 && kl == 0) {
                int prev = -1;
                for (int e = 0; e < n_emit; ++e) {
                    int r = tq.head + 1 + e; if (r >= R) r -= R;
                    const int qd = r >> 2;
                    if (qd != prev) ql[atomicAdd(nq, 1)] = (unsigned short)(tag | (unsigned)qd);
                    prev = qd;
                }
            }
        }
        }
        lds_barrier<NT>();
        WG_STAMP(10);
        // pipelined advection (GLP): the loads of a lane's first listed quad are requested before the deficit evaluation
        // (its ~4 k cycles of ALU work hide their round trip), and inside the pass every lane requests its next quad
        // before it computes the current one.  Costs 12 + 12 registers: this variant is built at 4 waves per SIMD.
        constexpr bool LFP = !GL && NT == 256 && (WG_ADV_PIPE_LF != 0);     // (large farms: see WG_ADV_PIPE_LF)
        constexpr bool GLP = (GL && (WG_ADV_PIPE != 0)) || LFP;
        const int nlist_pre = GL ? gl_nlist : 0;
        // (two quads ahead when WG_ADV_PIPE = 2: `n_*` is the lane's next quad, `m_*` the one after it)
        struct QuadReq { float4 py; uint4 ra, rb; int t, kq, q; };
        QuadReq nq_, mq_;
        nq_.py = mq_.py = make_float4(0.f, 0.f, 0.f, 0.f);
        nq_.ra = nq_.rb = mq_.ra = mq_.rb = make_uint4(0u, 0u, 0u, 0u);
        nq_.t = nq_.kq = nq_.q = mq_.t = mq_.kq = mq_.q = 0;
        auto adv_request_to = [&](QuadReq& r, const int c, const bool valid) __attribute__((always_inline)) {
            const unsigned ent = valid ? ql[c] : 0u;
            r.t = (int)(ent >> qsh); r.kq = (int)(ent & ((1u << qsh) - 1u));
            r.q = valid ? (T[r.t].roff >> 2) + r.kq : 0;
            r.py = reinterpret_cast<const float4*>(pl.py)[r.q];
            r.ra = reinterpret_cast<const uint4*>(pl.ra)[IL ? 2 * r.q : r.q];
            r.rb = IL ? reinterpret_cast<const uint4*>(pl.ra)[2 * r.q + 1] : reinterpret_cast<const uint4*>(pl.rb)[r.q];
        };
        constexpr bool GLP2 = (GL && (WG_ADV_PIPE >= 2)) || (LFP && (WG_ADV_PIPE_LF >= 2));
        if (GL) {
            // deficit phase, part 2: the gathers issued before the records have landed (or do so now)
            for (int t = tid; t < N; t += NT) { T[t].u = ws_f; T[t].v = 0.f; T[t].w = 0.f; }
            // (the wait is the s_waitcnt BUILTIN, not inline asm: the compiler's wait-count bookkeeping sees it and stops
            // treating the LDS-DMA requests as pending — otherwise every later wait for an ordinary load degrades to
            // vmcnt(0), which drains the advection pass's stores once per 64 quads, and every LDS atomic gets one too.  It
            // does NOT order plain LDS reads of the landing zone behind the requests by itself; the empty asm keeps the
            // reads below the wait)
            const int l = tid & 63;
            wg_wait_vmem();
            const unsigned* gatu = reinterpret_cast<const unsigned*>(gat);
            uint2 g_q0 = make_uint2(gatu[l], gatu[64 + l]), g_q1 = make_uint2(gatu[128 + l], gatu[192 + l]);
            float g_py0 = gat[256 + l], g_py1 = gat[320 + l];
            // (GLP: the first quad of the advection pass is requested now — its round trip runs under the deficit
            // evaluation below)
            if (GLP) adv_request_to(nq_, tid, tid < nlist_pre);
            if (GLP2) adv_request_to(mq_, tid + NT, tid + NT < nlist_pre);
            for (int c0 = 0; c0 < gl_nc; c0 += NT) {
                const int c = c0 + tid;
                if (c0 > 0) {      // (more than 64 candidates: not the common case) the next batch lands in the same words
                    lds_barrier<NT>();
                    gl_rest = gl_issue(c, gl_nc);
                    wg_wait_vmem();
                    g_q0 = make_uint2(gatu[l], gatu[64 + l]); g_q1 = make_uint2(gatu[128 + l], gatu[192 + l]);
                    g_py0 = gat[256 + l]; g_py1 = gat[320 + l];
                }
                if (b_ok) {
                    const int i = b_i, tl = b_tl, s2 = b_s2, j = b_j;
                    const double dx = b_dx;
                    const float wgt = b_wgt;
                    const TurbLds& src = T[s2];
                    const int jp0 = j - n_emit, jp1 = jp0 + 1;
                    float py0 = g_py0, py1 = g_py1;
                    if (gl_rest) { py0 = (float)src.yr; py1 = py0; }      // (not fetched: see gl_issue)
                    unsigned a0 = g_q0.x, b0_ = g_q0.y, a1 = g_q1.x, b1_ = g_q1.y;
                    // released in this step: the turbine's record, at the turbine
                    if (jp0 < 0) { py0 = (float)src.yr; a0 = pack_a(src.rct, src.rk); b0_ = pack_b(src.rue * ue_inv, src.rhv); }
                    if (jp1 < 0) { py1 = (float)src.yr; a1 = pack_a(src.rct, src.rk); b1_ = pack_b(src.rue * ue_inv, src.rhv); }
                    if (jp0 >= 0 && jp0 < n_valid) py0 = m0_advect(py0, a0, b0_, jp0, s_off_f, p.dpart_f, p.inv_D, p.dt, p.eps0);
                    if (jp1 >= 0 && jp1 < n_valid) py1 = m0_advect(py1, a1, b1_, jp1, s_off_f, p.dpart_f, p.inv_D, p.dt, p.eps0);
                    eval_pair(i, tl, s2, tl, dx, wgt, py0, py1, 0.f, 0.f, a0, a1, b0_, b1_);
                }
            }
            lds_barrier<NT>();
            // thread t: superposition in ascending source order over its candidate mask
            if (tid < N) {
                float dsum = 0.f, tia_max = 0.f;
                unsigned m = gl_mask;
                while (m) {
                    const int s2 = __builtin_ctz(m);
                    m &= m - 1;
                    dsum += def[tid * N + s2];
                    tia_max = fmaxf(tia_max, tiav[tid * N + s2]);
                }
                T[tid].u -= dsum;
                T[tid].ti = __builtin_amdgcn_sqrtf(ti_f * ti_f + tia_max * tia_max);
            }
            lds_barrier<NT>();
            WG_STAMP(9);
        }
        const int nlist = GL ? gl_nlist : *nq;
        if (LFP) {
            adv_request_to(nq_, tid, tid < nlist);
            if (GLP2) adv_request_to(mq_, tid + NT, tid + NT < nlist);
        }
        for (int c = tid; c < nlist; c += NT) {
            int t, kq, q;
            float4 py; uint4 ra, rb;
            if (GLP) {
                t = nq_.t; kq = nq_.kq; q = nq_.q; py = nq_.py; ra = nq_.ra; rb = nq_.rb;
                if (GLP2) {
                    nq_ = mq_;
                    if (c + 2 * NT < nlist) adv_request_to(mq_, c + 2 * NT, true);
                } else if (c + NT < nlist) adv_request_to(nq_, c + NT, true);
            } else {
                const unsigned ent = ql[c];
                t = (int)(ent >> qsh); kq = (int)(ent & ((1u << qsh) - 1u));
                q = (T[t].roff >> 2) + kq;
                py = reinterpret_cast<const float4*>(pl.py)[q];
                ra = reinterpret_cast<const uint4*>(pl.ra)[IL ? 2 * q : q];
                rb = IL ? reinterpret_cast<const uint4*>(pl.ra)[2 * q + 1] : reinterpret_cast<const uint4*>(pl.rb)[q];
            }
            TurbLds& tq = T[t];
            const int R = tq.rlen, hd = tq.head;
            const int r0 = 4 * kq;
            int j0 = hd - r0; if (j0 < 0) j0 += R;             // age of ring slot r0 (slot r0+i: j0-i)
            int e0 = r0 - hd - 1; if (e0 < 0) e0 += R;         // emission index of slot r0 (r0+i: e0+i)
            const bool emits = (e0 < n_emit) || (n_emit > 0 && e0 + 3 >= R);   // wraps past R-1 -> 0
            float pyv[4] = {py.x, py.y, py.z, py.w};
            // (IL: `ra` / `rb` hold the quad's interleaved records (a0 b0 a1 b1) (a2 b2 a3 b3))
            unsigned rav[4] = {ra.x, IL ? ra.z : ra.y, IL ? rb.x : ra.z, IL ? rb.z : ra.w};
            unsigned rbv[4] = {IL ? ra.y : rb.x, IL ? ra.w : rb.y, IL ? rb.y : rb.z, rb.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int j = j0 - i; if (j < 0) j += R;
                if (j < n_valid) pyv[i] = m0_advect(pyv[i], rav[i], rbv[i], j, s_off_f, p.dpart_f, p.inv_D, p.dt, p.eps0);
            }
            const float y0 = (float)tq.yr;
            if (emits) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int ei = e0 + i; if (ei >= R) ei -= R;
                    if (ei < n_emit) {
                        pyv[i] = y0; rav[i] = pack_a(tq.rct, tq.rk); rbv[i] = pack_b(tq.rue * ue_inv, tq.rhv);
                    }
                }
                if (IL) {
                    reinterpret_cast<uint4*>(pl.ra)[2 * q] = make_uint4(rav[0], rbv[0], rav[1], rbv[1]);
                    reinterpret_cast<uint4*>(pl.ra)[2 * q + 1] = make_uint4(rav[2], rbv[2], rav[3], rbv[3]);
                } else {
                    reinterpret_cast<uint4*>(pl.ra)[q] = make_uint4(rav[0], rav[1], rav[2], rav[3]);
                    reinterpret_cast<uint4*>(pl.rb)[q] = make_uint4(rbv[0], rbv[1], rbv[2], rbv[3]);
                }
            }
#if WG_NT_PY
            {   // (streamed once per launch and not read again before the next launch: non-temporal, so that the lines do not
                // sit dirty in L2 until the end-of-kernel write-back)
                typedef float f4v __attribute__((ext_vector_type(4)));
                f4v pv = {pyv[0], pyv[1], pyv[2], pyv[3]};
                __builtin_nontemporal_store(pv, reinterpret_cast<f4v*>(pl.py) + q);
            }
#else
            reinterpret_cast<float4*>(pl.py)[q] = make_float4(pyv[0], pyv[1], pyv[2], pyv[3]);
#endif
            // (excursion bound over the VALID particles of the quad only: a slot that holds no particle yet keeps whatever
            // an earlier episode left there, and one such value would loosen the chain's bound for the whole episode)
            float ex = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int j = j0 - i; if (j < 0) j += R;
                if (j < n_valid) ex = fmaxf(ex, fabsf(pyv[i] - y0));
            }
            if (ex > tq.bd) atomicMax(reinterpret_cast<int*>(&tq.bd), __float_as_int(ex));   // ex >= 0: int order == float order
        }
    } else if (RES) {
        // compact rings, turbulent inflow: every valid particle meanders.  One ring slot per lane and sub-iteration
        // (consecutive lanes = consecutive particles of a chain, 0.2 D apart along x: the box gathers of a wave share
        // cache lines); the lookups of U slots are issued together, then the filter / position updates.
        const double xshift = tc.ox - tc.ws * sr.time;
        // "Random" gusts are keyed by the particle's slot in the oracle's full P-slot ring: its newest particle sits at
        // (emission count - 1) mod P
        const int hfull = sr.n_emitted == 0u ? P - 1 : (int)((sr.n_emitted - 1u) % (unsigned)P);
        auto turbulent_res = [&](auto coarse_tag, auto pow2_tag) {
            constexpr bool COARSE = decltype(coarse_tag)::value;
            constexpr bool POW2 = decltype(pow2_tag)::value;
            constexpr int U = 4;
            for (int b0 = tid; b0 < pl.L; b0 += NT * U) {
                float pyv[U], pzv[U], fv[U], fw[U], vls[U], wls[U];
                unsigned ras[U], rbs[U];
                int jv[U], tv[U], ev[U];
                // (every streamed word of the U slots is requested before the first box lookup waits for py / pz: the
                // filter state and the record then arrive under the lookups' round trip instead of after it)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ix = min(b0 + u * NT, pl.L - 1);
                    pyv[u] = pl.py[ix]; pzv[u] = pl.pz[ix];
                    vls[u] = pl.vl[ix]; wls[u] = pl.wl[ix]; ras[u] = pl.ra[ix]; rbs[u] = pl.rb[ix];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ix = min(b0 + u * NT, pl.L - 1);
                    const int t = pl.own[ix >> 2];
                    const TurbLds& tq = T[t];
                    const int R = tq.rlen;
                    const int r = ix - tq.roff;
                    int j = tq.head - r; if (j < 0) j += R;
                    jv[u] = j; tv[u] = t; ev[u] = R - 1 - j;       // = (r - head - 1) mod R: emission index of this slot
                    if (TURB != WG_TURB_RANDOM) {
                        const float xrel = s_off_f + (float)j * p.dpart_f;
                        const double bx = tq.xr + (double)xrel + xshift, by = (double)pyv[u] + tc.oy, bz = (double)pzv[u];
                        if (WG_ABLATE & 4) { fv[u] = (float)bx * 1e-6f; fw[u] = (float)(by + bz) * 1e-6f; }
                        else if (COARSE) cbox_lookup_vw<POW2>(tc.box4c, p, bx, by, bz, fv[u], fw[u]);
                        else { float f3[3]; box_lookup<POW2>(tc.box4, p, bx, by, bz, f3); fv[u] = f3[1]; fw[u] = f3[2]; }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ix = b0 + u * NT;
                    if (ix >= pl.L) continue;
                    const int j = jv[u];
                    TurbLds& tq = T[tv[u]];
                    float vlv = vls[u], wlv = wls[u];
                    const unsigned rav = ras[u], rbv = rbs[u];
                    if (j < n_valid) {
                        const float xrel = s_off_f + (float)j * p.dpart_f;
                        const float sp = rec_k(rav) * (xrel * p.inv_D) + rec_eps(rav, p.eps0);
                        if (TURB == WG_TURB_RANDOM) {
                            int rfull = hfull - j; if (rfull < 0) rfull += P;
                            const uint32_t key = (uint32_t)(tv[u] * P + rfull);
                            fv[u] = wg_turb_normal(tc.seed, sr.istep, key, 1u, 0x50u);
                            fw[u] = wg_turb_normal(tc.seed, sr.istep, key, 2u, 0x50u);
                        }
                        vlv += tc.alpha * (tc.sig * fv[u] - vlv);
                        wlv += tc.alpha * (tc.sig * fw[u] - wlv);
                        pyv[u] += (rec_hv(rbv) * m0_cfrac(rec_ct(rav), sp) + vlv) * p.dt;
                        pzv[u] += wlv * p.dt;
                    }
                    const float y0 = (float)tq.yr;
                    if (ev[u] < n_emit) {
                        pyv[u] = y0; pzv[u] = p.hub; vlv = 0.f; wlv = 0.f;
                        const unsigned na = pack_a(tq.rct, tq.rk), nb = pack_b(tq.rue * ue_inv, tq.rhv);
                        pl.ra[ix] = na; pl.rb[ix] = nb;
                    }
                    const float ex = j < n_valid ? fabsf(pyv[u] - y0) + fabsf(pzv[u] - p.hub) : 0.f;   // (valid particles only)
                    if (ex > tq.bd) atomicMax(reinterpret_cast<int*>(&tq.bd), __float_as_int(ex));
                    pl.py[ix] = pyv[u]; pl.pz[ix] = pzv[u]; pl.vl[ix] = vlv; pl.wl[ix] = wlv;
                }
            }
        };
        if (TURB == WG_TURB_BOX && p.coarse) {
            if (p.cbox_pow2) turbulent_res(std::true_type{}, std::true_type{});
            else turbulent_res(std::true_type{}, std::false_type{});
        } else if (TURB == WG_TURB_BOX && p.box_pow2) turbulent_res(std::false_type{}, std::true_type{});
        else turbulent_res(std::false_type{}, std::false_type{});
    } else if (TURB != WG_TURB_NONE) {
        // turbulent inflow: every valid particle meanders -> all state arrays are streamed (py, pz, vlp, wlp r/w,
        // record read) and the transverse inflow at the particle is looked up (frozen box: 8 corners x 2
        // components, L2/MALL-resident shared box; "Random": counter-based normals)
        float* __restrict__ gpz = d.pz + pbase;
        float* __restrict__ gvl = d.vlp + pbase;
        float* __restrict__ gwl = d.wlp + pbase;
        const double xshift = tc.ox - tc.ws * sr.time;
        // lane -> ONE ring slot per sub-iteration, U slots NT apart per iteration: consecutive lanes hold consecutive
        // particles of a chain (0.2 D apart along x), so one gather instruction of the wave touches ~11 cache lines
        // of an x-fastest box row instead of 64 — the gathers are bound by L2->L1 line transfers, not by HBM.
        // Software pipeline: the streamed words of the NEXT iteration are requested before the gathers of the
        // current one are consumed, so an iteration exposes one memory round trip instead of two.
        // (the box variant and the wrap rule are uniform: hoisted out of the loop as compile-time tags, so the
        // lookups of an iteration stay in one basic block)
        auto turbulent_pass = [&](auto coarse_tag, auto pow2_tag) {
        constexpr bool COARSE = decltype(coarse_tag)::value;
        constexpr bool POW2 = decltype(pow2_tag)::value;
        constexpr int U = 4;
        float npy[U], npz[U], nvl[U], nwl[U]; unsigned nra[U], nrb[U];
        // chain pruning: a slot is streamed only if its particle can still reach a rotor (age <= jnl[t]) or is
        // being emitted in this step
        auto slot_live = [&](const int ix) -> bool {
            if (ix >= p.NP) return false;
            if (!PRUNE) return true;
            const int t = (int)(((float)ix + 0.5f) * p.inv_P);
            int j = head - (ix - t * P); if (j < 0) j += P;
            return (j <= jnl[t] && j < n_valid) || (P - 1 - j < n_emit);
        };
        bool nlive[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ix = tid + u * NT;
            nlive[u] = slot_live(ix);
            if (nlive[u]) {
                npy[u] = WG_LDS(gpy + ix); npz[u] = WG_LDS(gpz + ix); nvl[u] = WG_LDS(gvl + ix); nwl[u] = WG_LDS(gwl + ix);
                nra[u] = WG_LDS(gra + ix); nrb[u] = WG_LDS(grb + ix);
            }
        }
        for (int b0 = tid; b0 < p.NP; b0 += NT * U) {
            float pyv[U], pzv[U], vlv[U], wlv[U], fv[U], fw[U]; unsigned rav[U], rbv[U];
            int jv[U], tv[U];
            bool live[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                pyv[u] = npy[u]; pzv[u] = npz[u]; vlv[u] = nvl[u]; wlv[u] = nwl[u]; rav[u] = nra[u]; rbv[u] = nrb[u];
                live[u] = nlive[u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ix = b0 + (U + u) * NT;
                nlive[u] = slot_live(ix);
                if (nlive[u]) {
                    npy[u] = WG_LDS(gpy + ix); npz[u] = WG_LDS(gpz + ix); nvl[u] = WG_LDS(gvl + ix); nwl[u] = WG_LDS(gwl + ix);
                    nra[u] = WG_LDS(gra + ix); nrb[u] = WG_LDS(grb + ix);
                }
            }
            // the inflow is looked up unconditionally (slots that hold no particle yet wrap into the box like any
            // other position, their result is discarded): no divergent branch between the lookups, so all gathers
            // of the iteration are in flight together
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ix = b0 + u * NT;
                const int t = min((int)(((float)ix + 0.5f) * p.inv_P), N - 1);
                const int r = ix - t * P;
                int j = head - r; if (j < 0) j += P;
                jv[u] = j; tv[u] = t;
                if (TURB != WG_TURB_RANDOM && (!PRUNE || live[u])) {     // (unconditional in the small-farm variants)
                    const float xrel = s_off_f + (float)j * p.dpart_f;
                    const double bx = T[t].xr + (double)xrel + xshift, by = (double)pyv[u] + tc.oy, bz = (double)pzv[u];
                    if (WG_ABLATE & 4) { fv[u] = (float)bx * 1e-6f; fw[u] = (float)(by + bz) * 1e-6f; }
                    else if (COARSE) cbox_lookup_vw<POW2>(tc.box4c, p, bx, by, bz, fv[u], fw[u]);
                    else { float f3[3]; box_lookup<POW2>(tc.box4, p, bx, by, bz, f3); fv[u] = f3[1]; fw[u] = f3[2]; }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ix = b0 + u * NT;
                if (!live[u]) continue;
                const int j = jv[u], t = tv[u];
                TurbLds& tq = T[t];
                if (j < n_valid) {
                    const float xrel = s_off_f + (float)j * p.dpart_f;
                    const float sp = rec_k(rav[u]) * (xrel * p.inv_D) + rec_eps(rav[u], p.eps0);
                    if (TURB == WG_TURB_RANDOM) {
                        fv[u] = wg_turb_normal(tc.seed, sr.istep, (uint32_t)ix, 1u, 0x50u);
                        fw[u] = wg_turb_normal(tc.seed, sr.istep, (uint32_t)ix, 2u, 0x50u);
                    }
                    vlv[u] += tc.alpha * (tc.sig * fv[u] - vlv[u]);
                    wlv[u] += tc.alpha * (tc.sig * fw[u] - wlv[u]);
                    pyv[u] += (rec_hv(rbv[u]) * m0_cfrac(rec_ct(rav[u]), sp) + vlv[u]) * p.dt;
                    pzv[u] += wlv[u] * p.dt;
                }
                int e = P - 1 - j; // = (r - head - 1) mod P: emission index of this slot
                const float y0 = (float)tq.yr;
                if (e < n_emit) {
                    pyv[u] = y0; pzv[u] = p.hub; vlv[u] = 0.f; wlv[u] = 0.f;
                    const unsigned na = pack_a(tq.rct, tq.rk), nb = pack_b(tq.rue * ue_inv, tq.rhv);
                    gra[ix] = na; grb[ix] = nb;
                }
                const float ex = j < n_valid ? fabsf(pyv[u] - y0) + fabsf(pzv[u] - p.hub) : 0.f;   // (valid particles only)
                if (ex > tq.bd) atomicMax(reinterpret_cast<int*>(&tq.bd), __float_as_int(ex));
                WG_STS(gpy + ix, pyv[u]); WG_STS(gpz + ix, pzv[u]); WG_STS(gvl + ix, vlv[u]); WG_STS(gwl + ix, wlv[u]);
            }
        }
        };
        if (TURB == WG_TURB_BOX && p.coarse) {
            if (p.cbox_pow2) turbulent_pass(std::true_type{}, std::true_type{});
            else turbulent_pass(std::true_type{}, std::false_type{});
        } else if (TURB == WG_TURB_BOX && p.box_pow2) turbulent_pass(std::false_type{}, std::true_type{});
        else turbulent_pass(std::false_type{}, std::false_type{});
    } else if (!(WG_ABLATE & 1)) {
        // thread -> quads of 4 consecutive ring slots of one turbine (P % 4 == 0).  A thread owns QB quads per
        // batch, strided by NT*4 floats so that every load instruction of the wave is one contiguous 1 KiB.
        // Two stages keep many loads in flight per lane (the pass is latency-bound otherwise):
        //   stage 1: rec_b (eps|hv) of all QB quads -> which quads move (any hv != 0) or receive a new particle
        //   stage 2: py and rec_a (ct|k) of those quads -> all issued before the first use
        // A quad that neither moves nor emits costs only its rec_b read (4 B per particle).
        // quads in flight per lane and occupancy are tuned together per workgroup size (measured, cfg2 / cfg3 / cfg4):
        // 128 threads: QB = 1 at 6 waves/SIMD (80 VGPRs) beats QB = 2 at 5 waves (96 VGPRs) by 6 %; 256 threads (large
        // farms, long streaming loops): QB = 2 at 5 waves is 8 % better; 64 threads: no difference
#ifdef WG_QB
        constexpr int QB = WG_QB;
#else
        constexpr int QB = (NT == 128) ? 1 : 2;
#endif
        const int stride = NT * 4;
        for (int b0 = tid * 4; b0 < p.NP; b0 += stride * QB) {
            uint4 rb[QB];
            bool need[QB], emits[QB];
            int j0s[QB], e0s[QB], ts[QB];
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int i4 = b0 + q * stride;
                // owner of the quad (P is a multiple of 4: a quad has one owner); i4 / P without the ~20-instruction
                // integer division: exact for i4 + 0.5 < 2^23 (the host enforces N * P < 2^23)
                const int tq_ = (int)(((float)i4 + 0.5f) * p.inv_P);
                const int t = PRUNE ? min(tq_, N - 1) : tq_;
                const int r0 = i4 - t * P;
                int j0 = head - r0; if (j0 < 0) j0 += P;           // age of ring slot r0 (slot r0+i: j0-i)
                int e0 = r0 - head - 1; if (e0 < 0) e0 += P;       // emission index of slot r0 (r0+i: e0+i)
                emits[q] = (i4 < p.NP) && ((e0 < n_emit) || (n_emit > 0 && e0 + 3 >= P));   // wraps past P-1 -> 0
                // chain pruning (large farms): all four particles (ages j0-3 .. j0, no wrap) are older than anything
                // a rotor can still see, or none has been emitted yet -> the quad is not streamed at all
                const bool live = PRUNE ? (i4 < p.NP) && (emits[q] || j0 < 3 || (j0 - 3 <= jnl[t] && j0 - 3 < n_valid))
                                        : (i4 < p.NP);
                j0s[q] = j0; e0s[q] = e0; ts[q] = t;
                rb[q] = live ? *reinterpret_cast<const uint4*>(grb + i4) : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int q = 0; q < QB; ++q)
                need[q] = emits[q] || rec_moves(rb[q].x) | rec_moves(rb[q].y) | rec_moves(rb[q].z) | rec_moves(rb[q].w);
            float4 py[QB];
            uint4 ra[QB];
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int i4 = b0 + q * stride;
                if (need[q]) {
                    py[q] = *reinterpret_cast<const float4*>(gpy + i4);
                    ra[q] = *reinterpret_cast<const uint4*>(gra + i4);
                }
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                if (!need[q]) continue;
                const int i4 = b0 + q * stride;
                float pyv[4] = {py[q].x, py[q].y, py[q].z, py[q].w};
                unsigned rav[4] = {ra[q].x, ra[q].y, ra[q].z, ra[q].w};
                unsigned rbv[4] = {rb[q].x, rb[q].y, rb[q].z, rb[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int j = j0s[q] - i; if (j < 0) j += P;
                    if (j < n_valid) {
                        const float xrel = s_off_f + (float)j * p.dpart_f;
                        const float sp = rec_k(rav[i]) * (xrel * p.inv_D) + rec_eps(rav[i], p.eps0);
                        pyv[i] += rec_hv(rbv[i]) * m0_cfrac(rec_ct(rav[i]), sp) * p.dt;
                    }
                }
                if (emits[q]) {
                    const TurbLds& tq = T[ts[q]];
                    const float y0 = (float)tq.yr;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int ei = e0s[q] + i; if (ei >= P) ei -= P;
                        if (ei < n_emit) {
                            pyv[i] = y0; rav[i] = pack_a(tq.rct, tq.rk); rbv[i] = pack_b(tq.rue * ue_inv, tq.rhv);
                        }
                    }
                    *reinterpret_cast<uint4*>(gra + i4) = make_uint4(rav[0], rav[1], rav[2], rav[3]);
                    *reinterpret_cast<uint4*>(grb + i4) = make_uint4(rbv[0], rbv[1], rbv[2], rbv[3]);
                }
                *reinterpret_cast<float4*>(gpy + i4) = make_float4(pyv[0], pyv[1], pyv[2], pyv[3]);
                {
                    TurbLds& tq = T[ts[q]];
                    const float y0 = (float)tq.yr;
                    float ex = 0.f;          // (valid particles only: see the compact variant)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int j = j0s[q] - i; if (j < 0) j += P;
                        if (j < n_valid) ex = fmaxf(ex, fabsf(pyv[i] - y0));
                    }
                    if (ex > tq.bd) atomicMax(reinterpret_cast<int*>(&tq.bd), __float_as_int(ex));   // ex >= 0: int order == float order
                }
            }
        }
    }
    // (GLP: a wait the compiler's bookkeeping sees, right after the pipelined pass — the last trip's conditional request
    // otherwise stays "possibly pending" in its view around the flow-step loop and it orders the next step's register
    // writes behind it with vmcnt(0) waits that drain the LDS-DMA gathers)
    if (RES && TURB == WG_TURB_NONE && NT == WG_WAVE && (WG_GLDS != 0) && (WG_PAIR_FIRST != 0) && (WG_ADV_PIPE != 0) && (WG_GL_POSTPASS_WAIT != 0)) wg_wait_vmem();
    sr.head = new_head; sr.n_valid = new_valid; sr.s_off = s_new; sr.time += p.dt_d; sr.istep += 1u;
    sr.n_emitted += (unsigned)n_emit;
    WG_STAMP(3);
    if (!GL) { full_barrier<NT>(); WG_STAMP(4); }      // (GL: the caller's loop waits on its back edge only — a single live step ends without it)   // this workgroup's particle stores are visible to its own gathers below

    // (3)+(4) rotor-averaged inflow
    if (RES) {
        if (!PRE) res_pair_phase(std::false_type{});
        return;
    }
    for (int t0 = 0; t0 < ((WG_ABLATE & 2) ? 0 : N); t0 += TC) {
        const int nt = (N - t0) < TC ? (N - t0) : TC;
        const int npairs = nt * N;
        if (t0 > 0) {
            for (int i = tid; i < nt * WG_MASK_WORDS; i += NT) tmask[i] = 0u;
            lds_barrier<NT>();
        }
        // phase A: one thread per (target, source) pair; contributing sources are flagged in tmask[target]
        for (int i = tid; i < npairs; i += NT) {
            const int tl = (int)(((float)i + 0.5f) * p.inv_N);
            const int s2 = i - tl * N;
            const int t = t0 + tl;
            float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
            const double dx = T[t].xr - T[s2].xr;
            // conservative pre-check without touching particle memory: every particle of chain s2 is within
            // bd of (y_s2, z_hub) and has sigma <= (bk x/D + be) D, so a pair farther apart than R + 5 sigma_max + bd
            // cannot pass the exact cut-off below
            bool cand = (s2 != t) && (dx > 0.0);
            if (cand) {
                const TurbLds& src = T[s2];
                const float sig_max = (src.bk * ((float)dx * p.inv_D) + src.be) * p.D;
                const float gap = fabsf((float)(T[t].yr - src.yr)) - (p.R_rot + 5.0f * sig_max + src.bd);
                cand = gap <= 1.0e-3f * p.D;      // small margin for fp32 rounding of the bound itself
            }
            if (cand) {
                const double xi = (dx - s_new) * p.inv_dpart;
                const double jf = floor(xi);
                float wgt = (float)(xi - jf);
                int j = (int)jf;
                if (j < 0) { j = 0; wgt = 0.f; }
                if (j + 1 <= new_valid - 1) {
                    const int Rs = RES ? T[s2].rlen : P;
                    int r0 = (RES ? T[s2].head_n : new_head) - j; if (r0 < 0) r0 += Rs;
                    int r1 = r0 - 1; if (r1 < 0) r1 += Rs;
                    const unsigned sb = RES ? (unsigned)T[s2].roff : (unsigned)(s2 * P);
                    const unsigned i0 = sb + (unsigned)r0, i1 = sb + (unsigned)r1;
                    // one round of gathers: the lines were streamed by this workgroup a moment ago (py, the record words —
                    // round 6: the record carries u_e, there is no separate copy to fetch)
                    const float py0 = RES ? pl.py[i0] : gpy[i0], py1 = RES ? pl.py[i1] : gpy[i1];
                    unsigned a0, a1, b0_, b1_;
                    if (RES) {
                        a0 = pl.ra[i0]; a1 = pl.ra[i1]; b0_ = pl.rb[i0]; b1_ = pl.rb[i1];
                    } else {
                        a0 = gra[i0]; a1 = gra[i1]; b0_ = grb[i0]; b1_ = grb[i1];
                    }
                    const float u0 = rec_uf(b0_) * ue_max, u1 = rec_uf(b1_) * ue_max;
                    const float k0 = rec_k(a0), k1 = rec_k(a1), e0 = rec_eps(a0, p.eps0), e1 = rec_eps(a1, p.eps0);
                    const float c0 = rec_ct(a0), c1 = rec_ct(a1);
                    const float w0 = 1.0f - wgt, w1 = wgt;
                    const float yc = w0 * py0 + w1 * py1;
                    float zc = p.hub;
                    if (TURB != WG_TURB_NONE) zc = RES ? w0 * pl.pz[i0] + w1 * pl.pz[i1] : w0 * d.pz[pbase + i0] + w1 * d.pz[pbase + i1];
                    const float kv = w0 * k0 + w1 * k1;
                    const float epv = w0 * e0 + w1 * e1;
                    const float xd = (float)dx * p.inv_D;
                    const float sp = kv * xd + epv;
                    const float sig = sp * p.D;
                    const float yt = (float)T[t].yr;
                    const float rc2 = (yt - yc) * (yt - yc) + (p.hub - zc) * (p.hub - zc);
                    const float rcut = p.R_rot + 5.0f * sig;
                    if (rc2 <= rcut * rcut) {
                        const float ctv = w0 * c0 + w1 * c1;
                        const float uev = w0 * u0 + w1 * u1;
                        const float cf = m0_cfrac(ctv, sp);
                        const float inv2s2 = __builtin_amdgcn_rcpf(2.0f * sig * sig);
                        // wake-added turbulence (Crespo-Hernandez), Gaussian-weighted at the hub
                        const float ind = 0.5f * (1.0f - __builtin_amdgcn_sqrtf(1.0f - ctv));
                        const float tia = p.no_ti_fold ? 0.f : p.tia * fast_pow(ind, p.tib) * ti_pow * fast_pow(fmaxf(xd, 1.0f), p.tid) *
                                                               __expf(-rc2 * inv2s2);
                        pp = make_float4(yc, zc, inv2s2, uev * cf);
                        tiap[i] = tia;
                        atomicOr(&tmask[tl * WG_MASK_WORDS + (s2 >> 5)], 1u << (s2 & 31));
                    }
                }
            }
            pair[i] = pp;
        }
        lds_barrier<NT>();
        WG_STAMP(5);
        // phase B: one thread per (target, sample); S_pad = S rounded up to a power of two
        const int nitems = nt << p.S_shift;
        for (int it = tid; it < ((nitems + NT - 1) & ~(NT - 1)); it += NT) {
            const int tl = it >> p.S_shift;
            const int s = it & (p.S_pad - 1);
            const bool live = (it < nitems) && (s < p.S);
            float acc = 0.f, tia_max = 0.f;
            float amb[3] = {0.f, 0.f, 0.f};
            // wake-added turbulence (see the compact variant): this thread's rotor point, all source wakes
            const bool ADDED = TURB != WG_TURB_NONE && p.added != 0;
            float g3[3] = {0.f, 0.f, 0.f}, add3[3] = {0.f, 0.f, 0.f};
            if (live) {
                const int t = t0 + tl;
                const float4* __restrict__ pr = pair + tl * N;
                const float ys = (float)T[t].yr + rdy[s] * T[t].cg;
                const float zs = p.hub + rdz[s];
                if (TURB == WG_TURB_BOX || ADDED) {
                    // ambient fluctuation at this rotor point (8 corners x 3 components of the frozen box): requested
                    // first, the scattered HBM reads overlap the wake sum below
                    const double bx = T[t].xr - tc.ws * sr.time + tc.ox;
                    const double by = T[t].yr + (double)(rdy[s] * T[t].cg) + tc.oy, bz = p.hub_d + (double)rdz[s];
                    if (TURB == WG_TURB_BOX) {
                        if (p.box_pow2) box_lookup<true>(tc.box4, p, bx, by, bz, amb);
                        else box_lookup<false>(tc.box4, p, bx, by, bz, amb);
                    }
                    if (ADDED) { abox_lookup(p, d, bx, by, bz, g3); ++add_acc; }
                }
                for (int wd = 0; wd * 32 < N; ++wd) {
                    unsigned m = tmask[tl * WG_MASK_WORDS + wd];
                    while (m) {                       // ascending source order -> deterministic sum
                        const int s2 = wd * 32 + __builtin_ctz(m);
                        m &= m - 1;
                        const float4 pp = pr[s2];
                        tia_max = fmaxf(tia_max, tiap[tl * N + s2]);
                        const float dy = ys - pp.x, dz = zs - pp.y;
                        const float r2 = dy * dy + dz * dz;
                        const float du = pp.w * __expf(-r2 * pp.z);
                        acc += du;
                        if (ADDED) {
                            const float wk = du * (p.km1 + p.km2r * pp.z * __builtin_amdgcn_sqrtf(r2));
                            add3[0] += wk * g3[0]; add3[1] += wk * g3[1]; add3[2] += wk * g3[2];
                        }
                    }
                }
            }
            for (int o = p.S_pad >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (ADDED) {
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
                    for (int o = p.S_pad >> 1; o > 0; o >>= 1) add3[cc] += __shfl_xor(add3[cc], o, 64);
            }
            if (TURB == WG_TURB_BOX) {
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
                    for (int o = p.S_pad >> 1; o > 0; o >>= 1) amb[cc] += __shfl_xor(amb[cc], o, 64);
            }
            if (live && s == 0) {
                TurbLds& q = T[t0 + tl];
                float au = 0.f, av = 0.f, aw = 0.f;
                if (TURB == WG_TURB_BOX) {
                    au = tc.sig * amb[0] * p.inv_S; av = tc.sig * amb[1] * p.inv_S; aw = tc.sig * amb[2] * p.inv_S;
                } else if (TURB == WG_TURB_RANDOM) {
                    // i.i.d. gusts at the S rotor points: their mean is one normal of variance sigma^2 / S
                    const float sc = tc.sig * p.inv_sqrt_S;
                    const uint32_t tt = (uint32_t)(t0 + tl);
                    au = sc * wg_turb_normal(tc.seed, sr.istep, tt, 0u, 0x52u);
                    av = sc * wg_turb_normal(tc.seed, sr.istep, tt, 1u, 0x52u);
                    aw = sc * wg_turb_normal(tc.seed, sr.istep, tt, 2u, 0x52u);
                }
                q.u = ws_f + au - acc * p.inv_S + add3[0] * p.inv_S;
                q.v = av + add3[1] * p.inv_S;
                q.w = aw + add3[2] * p.inv_S;
                q.ti = __builtin_amdgcn_sqrtf(ti_f * ti_f + tia_max * tia_max);
            }
        }
        lds_barrier<NT>();
    }
}

// replay mode (test hook): consume one scripted row instead of the physics
template <int NT>
__device__ __forceinline__ void script_step(const FlowP& p, const FlowPtrs& d, TurbLds* T, int e, int farm,
                                            int& cursor, double& time) {
    ++cursor;
    time += p.dt_d;
    const int row = cursor < p.script_rows ? cursor : p.script_rows - 1;
    const size_t base = (((size_t)farm * p.script_rows + row) * p.B + e) * p.N;
    for (int t = threadIdx.x; t < p.N; t += NT) {
        T[t].u = d.script_uvw[(base + t) * 3 + 0];
        T[t].v = d.script_uvw[(base + t) * 3 + 1];
        T[t].w = d.script_uvw[(base + t) * 3 + 2];
        T[t].pow = d.script_power[base + t];
    }
    lds_barrier<NT>();
}

// sum of a per-turbine LDS field over all turbines, by wave 0 (fixed shuffle tree -> deterministic);
// result valid in every lane of wave 0
#define WG_TURB_SUM(field)                                                                  \
    ([&]() {                                                                                \
        float _s = 0.f;                                                                     \
        for (int _t = (int)(threadIdx.x & 63); _t < N; _t += WG_WAVE) _s += T[_t].field;    \
        return N <= 16 ? wg_row_sum(_s) : wg_wave_sum(_s);   /* (lanes >= N hold 0: row 0 alone is the sum) */ \
    }())

template <int NT, int TURB, bool REPLAY, bool NOISE, bool RES, int SGM = 0>
__global__ void __launch_bounds__(NT, TURB != WG_TURB_NONE ? WG_BOX_WAVES : (RES ? (NT == WG_WAVE ? WG_FLOW_WAVES_GL : (NT == 256 ? WG_FLOW_WAVES_LF : WG_FLOW_WAVES_CG)) : WG_FLOW_WAVES))
k_flow(const FlowP p, const FlowPtrs d, const int mode, const float* __restrict__ actions,
       const uint8_t* __restrict__ mask, const int chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int F = p.F, N = p.N;
    // (the host selects the compact variants only for farms with at most one turbine per thread: every
    // `for (t = tid; t < N; t += NT)` below is a single trip, and the compiler may know it)
    if (RES) __builtin_assume(N <= NT);
    const int bid = blockIdx.x;
    // Block order.  Inflow "None" / "Random": farm-major — all agent farms first; the lighter baseline farms (un-yawed
    // chains do not move) form the tail of the launch, where CUs drain (+1.6 % on cfg2).  Frozen box: env-major — the
    // two farms of an env gather the same region of the box and share its lines in L2 when they run side by side
    // (farm-major: -4 % on cfg5).
    int farm, ec;
    if (TURB == WG_TURB_BOX) {
        if (F == 2 && (WG_BOX_XCD_PAIRS != 0) && (gridDim.x & 15u) == 0u) {
            // (workgroup i runs on XCD i % 8, each XCD has its own L2: the two farms of an env are 8 apart in the block order,
            // so that they land on the SAME XCD one dispatch round apart — adjacent indices share nothing but the MALL)
            farm = (bid >> 3) & 1;
            ec = ((bid >> 4) << 3) | (bid & 7);
        } else {
            ec = F == 2 ? (bid >> 1) : bid / F;
            farm = F == 2 ? (bid & 1) : bid - ec * F;
        }
    } else {
        const int nec = p.B * 2;
        farm = bid < nec ? 0 : bid / nec;
        ec = bid - farm * nec;
        // (tried: an env's flow workgroups on the XCD that runs its glue wave, so that one kernel would find in its L2 what the
        // previous one wrote last — 60.2 vs 59.8 M env-steps/s on cfg2, nothing on cfg3: the L2s are written back and invalidated
        // at kernel boundaries; not kept)
    }
    const int e = ec >> 1;
    // Which of the env's two contexts this workgroup serves is a pseudo-random function of the env index.  The
    // dispatcher hands workgroups to XCDs / shader engines / CUs in fixed round-robin patterns of blockIdx; with the
    // context bit taken straight from the index, "all live farms" (after a synchronised start every env has
    // live == 0) and "background work only" land on disjoint halves of the chip: measured 4.3 resident workgroups
    // per CU instead of 9 and a launch that lasts as long as the loaded half.  Hashing the bit spreads both kinds of
    // work over every XCD, SE and CU whatever the pattern is.
    const int c = (ec & 1) ^ (int)((((uint32_t)e * 2654435761u) >> 13) & 1u);
    const int tid = threadIdx.x;
    const int ctx_id = e * 2 + c;
    const int slot_id = ctx_id * F + farm;
    const size_t tb = (size_t)slot_id * N;
    const size_t pbase = (size_t)slot_id * p.pstride;

    WG_STAMP(0);
#if defined(WG_TIMELINE) && defined(WG_TIMELINE_WALL)
    const long long wg_wall0 = wall_clock64();
    if (tid == 0 && d.dbg) d.dbg[(size_t)blockIdx.x * 16] = 0;
#endif
    if ((WG_ABLATE & 32) && mode == WG_MODE_STEP) return;      // profiling: cost of the dispatch alone
    // ---- prologue: issue every independent global load up front (ONE exposed memory round trip) -------
    // (cold parameters: the pointers below are read from the kernarg segment where they are used and not kept — see
    // wg_cold_args)
    // Per-workgroup timeline (tools/timeline2.sh, cfg2): a memory round trip costs ~5-7 k cycles under load and the
    // prologue was THREE in sequence (kernarg -> env header, waited for before the mask branch -> headers + state).  So:
    //  * the env / slot / context headers are wave-uniform and not written by this workgroup before it reads them: they
    //    are fetched through the constant address space = scalar loads (K$ is invalidated at every kernel start), which do
    //    not queue behind, nor hold up, the vector loads of the turbine state;
    //  * no control flow between the loads (the mask byte is read unconditionally from a valid address).
    const KArgsPtr k0 = wg_cold_args();
    typedef const __attribute__((address_space(4))) WgEnv* CEnvPtr;
    typedef const __attribute__((address_space(4))) WgSlot* CSlotPtr;
    typedef const __attribute__((address_space(4))) WgCtx* CCtxPtr;
    const WgEnv& env = d.env[e];
    const CEnvPtr envc = (CEnvPtr)(d.env + e);
    const int env_live = envc->live, env_done = envc->done, env_shadow_iters = envc->shadow_iters;
    const uint64_t noise_key = envc->noise_key;
    const int init_pending = ((CCtxPtr)(d.ctx + ctx_id))->init_pending;
    const bool use_mask = mode == WG_MODE_RESET && mask != nullptr;
    const uint8_t mask_byte = *(use_mask ? mask + e : reinterpret_cast<const uint8_t*>(d.env + e));
    const uint8_t masked_out = use_mask ? (uint8_t)(mask_byte == 0) : (uint8_t)0;
    // An idle background workgroup (its context has no share of work in this launch: ~28 % of the workgroups of a cfg2
    // launch) leaves on its header alone, before any state is requested (round 4: 0 ... +1.6 % same-box, 9 MB of reads less)
#ifndef WG_NO_IDLE_EARLY
    if (mode == WG_MODE_STEP && c != env_live && !init_pending && env_shadow_iters <= 0) return;
#endif
    const int t_own = tid < N ? tid : 0;
    int dev_rem, fill_rem, cursor, n_pushed, pend_farm_n, pend_base_n;
    unsigned part0, flow0;       // accounting counters: read with the headers, so that the epilogue only stores
    SlotRegs sr;
    double ws;
    float ti_f, wd_env;
    uint32_t episode_tag;
    TurbCtx tc;
    double l_xr = 0, l_yr = 0;
    float l_yaw = 0, l_u = 0, l_v = 0, l_w = 0, l_ti = 0, l_pow = 0, l_ct = 0, l_act = 0;
    float4 l_bnd = make_float4(0.f, 0.f, 0.f, 0.f);
    int l_jn = 0, l_roff = 0, l_rnext = 0, L_ring = 0;
    auto load_headers = [&](const auto& slot, const auto& cx) __attribute__((always_inline)) {
        dev_rem = slot.dev_remaining; fill_rem = slot.fill_remaining;
        sr = SlotRegs{slot.s_off, slot.time, slot.head, slot.n_valid, slot.istep, slot.n_emitted};
        cursor = slot.cursor;
        part0 = slot.part_count; flow0 = slot.flow_count;
        ws = cx.ws;
        ti_f = (float)cx.ti;
        wd_env = (float)cx.wd;
        n_pushed = cx.n_pushed;
        pend_farm_n = cx.pend_farm_n; pend_base_n = cx.pend_base_n;
        episode_tag = (uint32_t)cx.episode_tag;
        tc.seed = cx.turb_seed; tc.ox = cx.box_ox; tc.oy = cx.box_oy; tc.ws = ws;
        tc.box4 = TURB == WG_TURB_BOX ? d.box4 + (size_t)cx.box_id * p.box_cells : nullptr;
        tc.box4c = TURB == WG_TURB_BOX && d.box4c ? d.box4c + (size_t)cx.box_id * p.cbox_cells : nullptr;
        tc.sig = (float)(cx.ti * ws);
        tc.alpha = TURB == WG_TURB_NONE ? 0.f
                                        : (float)(1.0 - exp(-2.0 * WG_PI_D * (ws / (p.fc_scale * p.D_d)) * p.dt_d));
    };
    // `fresh`: after the in-kernel episode set-up the headers were just rewritten by this workgroup -> plain (vector)
    // loads, not the scalar cache
    auto load_state = [&](const bool fresh) __attribute__((always_inline)) {
        const KArgsPtr kl = wg_cold_args();
        if (fresh) load_headers(d.slot[slot_id], d.ctx[ctx_id]);
        else load_headers(*(CSlotPtr)(d.slot + slot_id), *(CCtxPtr)(d.ctx + ctx_id));
        if (RES) {
            const int* ro = kl->d.roff + (size_t)ctx_id * (N + 1);
            l_roff = ro[t_own]; l_rnext = ro[t_own + 1]; L_ring = ro[N];
        }
        l_xr = kl->d.xr[(size_t)ctx_id * N + t_own];
        l_yr = kl->d.yr[(size_t)ctx_id * N + t_own];
        l_yaw = kl->d.yaw[tb + t_own]; l_u = kl->d.u[tb + t_own]; l_v = kl->d.v[tb + t_own]; l_w = kl->d.w[tb + t_own];
        l_ti = kl->d.ti_loc[tb + t_own]; l_pow = kl->d.power[tb + t_own]; l_ct = kl->d.ct[tb + t_own];
        l_bnd = reinterpret_cast<const float4*>(kl->d.bnd)[tb + t_own];
        if (!RES) l_jn = kl->d.jneed[(size_t)ctx_id * N + t_own];      // chain pruning by predicate: streaming variant only
    };
    load_state(false);
    // (first elements of the read-only tables: requested with the state loads — copied to LDS after the decision below, a
    // separate copy loop there costs every workgroup a second memory round trip)
    // (clamped indices instead of predicated loads: no branches, every request above and below is in flight together)
    const int i_tab = min(tid, p.n_tab - 1), i_s = min(tid, p.S - 1);
    const float pf_tp = k0->d.tab_power[i_tab], pf_tc = k0->d.tab_ct[i_tab];
    const float pf_dy = k0->d.rotor_dy[i_s], pf_dz = k0->d.rotor_dz[i_s];
    {
        const bool has_act = mode == WG_MODE_STEP && farm == 0;
        const float a = *(has_act ? actions + (size_t)e * N + t_own : k0->d.tab_ct);
        l_act = has_act ? a : 0.f;
    }

    const bool is_live = (c == env_live);
    int budget = 0;
    const bool live_step = (mode == WG_MODE_STEP) && is_live;
    if (mode == WG_MODE_STEP) {
        if (is_live) {
            if (env_done) return;
        } else {
            if (!p.autoreset) return;
#ifdef WG_NO_INIT_PATH
            if (false) {
#else
            if (init_pending) {
#endif
                // Rare path (one context per truncation): the episode this context will hold has not been set up yet
                // — k_glue flagged it when it retired the finished episode.  Wave 0 of BOTH farm workgroups runs the
                // same initialisation from the generator snapshot k_glue left in the context: identical context-level
                // values are written by both (a benign same-value race), each initialises only its own farm's slot
                // and turbine arrays, farm 0 commits the advanced generator.  The flag is cleared by the next k_glue.
                const WgParams& gp = *d.gp;
                if (tid < WG_WAVE) flow_init_episode(d.gp, d.gd, d.env_rw + e, e, c, farm, tid);
                full_barrier<NT>();
                load_state(true);
                // first share of the background work, planned here because k_glue could not know it yet
                const int fill_max = F == 2 ? max(gp.fill_a, gp.fill_b) : gp.fill_a;
                const int inc = 1 + (gp.extra_inc ? 1 : 0);
                const int tm = d.ctx[e * 2 + env_live].time_max;
                const long total = (long)((tm + inc - 1) / inc) + 1;
                budget = wg_shadow_share(dev_rem + p.K * fill_max, total - env.steps_done, env.steps_done, e);
            } else {
                if (dev_rem == 0 && fill_rem == 0) return;
                budget = env_shadow_iters;
            }
            if (budget <= 0) return;
        }
    } else {
        if (!is_live || (dev_rem == 0 && fill_rem == 0) || masked_out) return;
        budget = chunk;
    }

    if ((WG_ABLATE & 8) && mode == WG_MODE_STEP) return;      // profiling: cost of launch + first round trip
    // LDS carve: pair[target_chunk * N] | T[N] | tabp[n_tab] | tabct[n_tab] | rdy[S] | rdz[S] | tmask
    float4* pair = reinterpret_cast<float4*>(smem);
    TurbLds* T = reinterpret_cast<TurbLds*>(smem + p.lds_off_turb);
    float* tabp = reinterpret_cast<float*>(smem + p.lds_off_tab);
    float* tabct = tabp + p.n_tab;
    float* rdy = tabct + p.n_tab;
    float* rdz = rdy + p.S;
    unsigned* tmask = reinterpret_cast<unsigned*>(rdz + p.S);
    float* tiap = reinterpret_cast<float*>(tmask + (p.gl ? 0 : p.target_chunk * WG_MASK_WORDS));
    int* jnl = reinterpret_cast<int*>(tiap + (RES ? 0 : p.target_chunk * N));   // [N] chain pruning ages (tiap: sample-major phases only)
    float* xyf = reinterpret_cast<float*>(jnl + 2 * N + 4);   // [2][N] float copies of the rotated positions (GL candidate pass)
    float* oyaw = reinterpret_cast<float*>(jnl + N + 4);      // [N] yaw before this step's action (a separate array: TurbLds at 120 bytes
                                                             // keeps its fields of consecutive turbines in different LDS banks; 128 does not)
    PartLds pl;
    if (RES) {
        pl.py = d.py + pbase;
        if (p.rec_il) { pl.ra = d.rec_a + 2 * pbase; pl.rb = pl.ra + 1; }      // interleaved record (GL handles)
        else { pl.ra = d.rec_a + pbase; pl.rb = d.rec_b + pbase; }
        pl.il = p.rec_il;
        pl.pz = TURB != WG_TURB_NONE ? d.pz + pbase : nullptr;
        pl.vl = TURB != WG_TURB_NONE ? d.vlp + pbase : nullptr;
        pl.wl = TURB != WG_TURB_NONE ? d.wlp + pbase : nullptr;
        pl.own = d.qown + (size_t)ctx_id * (p.NP >> 2);
        pl.L = L_ring;
    }
    for (int t = tid; t < N; t += NT) {
        TurbLds& q = T[t];
        if (t == tid && tid < N) {
            xyf[t] = (float)l_xr; xyf[N + t] = (float)l_yr;
            q.xr = l_xr; q.yr = l_yr; q.yaw = l_yaw; q.u = l_u; q.v = l_v; q.w = l_w; q.ti = l_ti; q.pow = l_pow; q.ct = l_ct;
            q.bd = l_bnd.x; q.bk = l_bnd.y; q.be = l_bnd.z; q.mvl = __float_as_uint(l_bnd.w);
            if (!RES) jnl[t] = l_jn;
            if (RES) {
                q.roff = l_roff; q.rlen = l_rnext - l_roff;
                // ((n_emitted - 1) mod rlen: emission counts stay far below 2^24, fast_mod's range)
                q.head = sr.n_emitted == 0u ? q.rlen - 1
                                            : fast_mod((int)(sr.n_emitted - 1u), q.rlen, __builtin_amdgcn_rcpf((float)q.rlen));
                q.head_n = q.head;
            }
        }
        if (t >= NT) {   // N > 256 (not the common case): remaining turbines loaded the slow way
            q.xr = d.xr[(size_t)ctx_id * N + t]; q.yr = d.yr[(size_t)ctx_id * N + t];
            q.yaw = d.yaw[tb + t]; q.u = d.u[tb + t]; q.v = d.v[tb + t]; q.w = d.w[tb + t];
            q.ti = d.ti_loc[tb + t]; q.pow = d.power[tb + t]; q.ct = d.ct[tb + t];
            { const float4 b4 = reinterpret_cast<const float4*>(d.bnd)[tb + t]; q.bd = b4.x; q.bk = b4.y; q.be = b4.z; q.mvl = __float_as_uint(b4.w); }
            if (!RES) jnl[t] = d.jneed[(size_t)ctx_id * N + t];
            if (RES) {
                const int* ro = d.roff + (size_t)ctx_id * (N + 1);
                q.roff = ro[t]; q.rlen = ro[t + 1] - ro[t];
                q.head = sr.n_emitted == 0u ? q.rlen - 1 : (int)((sr.n_emitted - 1u) % (unsigned)q.rlen);
                q.head_n = q.head;
            }
        }
        q.sws = 0.f; q.swd = 0.f; q.syaw = 0.f; q.sp = 0.f;
    }
    if (tid == 0) { jnl[N] = 0; jnl[N + 3] = 0; }
    if (tid < p.n_tab) { tabp[tid] = pf_tp; tabct[tid] = pf_tc; }
    if (tid < p.S) { rdy[tid] = pf_dy; rdz[tid] = pf_dz; }
    for (int i = tid + NT; i < p.n_tab; i += NT) { tabp[i] = d.tab_power[i]; tabct[i] = d.tab_ct[i]; }
    for (int i = tid + NT; i < p.S; i += NT) { rdy[i] = d.rotor_dy[i]; rdz[i] = d.rotor_dz[i]; }

    const float ti_pow = fast_pow(ti_f, p.tic);

    // (every prologue load has been consumed by now: a wait the compiler's bookkeeping sees costs nothing here and keeps it
    // from ordering later register writes behind "possibly pending" prologue loads — in the GL variant such a wait
    // would sit between the LDS-DMA gathers and their use and drain them)
    if (RES && TURB == WG_TURB_NONE && NT == WG_WAVE) wg_wait_vmem();
    // WindFarmEnv._adjust_yaws (Wind_Farm_Env.py:822-864), float32 like numpy evaluates it on a float32 action
    if (live_step && farm == 0) {
        for (int t = tid; t < N; t += NT) {
            float yaw = t < NT ? l_yaw : d.yaw[tb + t];
            oyaw[t] = yaw;            // :932 (written to d.old_yaw in the epilogue: a store here is the oldest outstanding memory
                                      // operation when the step's first phases run, and the compiler's waits order behind it)
            const float a = t < NT ? l_act : actions[(size_t)e * N + t];
            if (p.action_method == WG_ACT_YAW) {
                yaw = fminf(fmaxf(yaw + a * p.yaw_step, p.yaw_min), p.yaw_max);
            } else {
                float tf = a + 1.0f;
                tf = tf * 0.5f;
                tf = tf * (p.yaw_max - p.yaw_min);
                tf = tf + p.yaw_min;
                const float ny = fminf(fmaxf(tf, yaw - p.yaw_step), yaw + p.yaw_step);
                yaw = fminf(fmaxf(ny, p.yaw_min), p.yaw_max);
            }
            T[t].yaw = yaw;
        }
    }
    lds_barrier<NT>();   // publishes T, the tables and the rotor offsets
    WG_STAMP(1);
    if ((WG_ABLATE & 16) && mode == WG_MODE_STEP) return;     // profiling: + state set-up and stream-in

    // live:   one env step = K sub-steps with measurement (Wind_Farm_Env.py:932-979)
    // else:   background development of a not-yet-live episode: flow-development steps (fs.run), then
    //         window-fill env steps (Wind_Farm_Env.py:722-796)
    int sub = 0, n_flow = 0, part_acc = 0, add_acc = 0;
    float base_acc = 0.f;
    const float inv_k = 1.0f / (float)p.K;
    // (GL variant: a further flow step of the launch gathers what this step's advection pass stored — the wait sits on the
    // loop's back edge, where the compiler's wait-count pass sees it on every path around the loop; a live step with K = 1
    // leaves the loop without it)
    constexpr bool GLK = RES && TURB == WG_TURB_NONE && NT == WG_WAVE && (WG_GLDS != 0) && (WG_PAIR_FIRST != 0);
    auto back_edge = [&]() __attribute__((always_inline)) { if (GLK) full_barrier<NT>(); };
    for (;; back_edge()) {
        if (!live_step && sub == 0 && (budget <= 0 || (dev_rem == 0 && fill_rem == 0))) break;
        const bool is_dev = !live_step && dev_rem > 0;
        if (live_step && farm == 1) {
            // BasicControllers.local_yaw_controller / global_yaw_controller (BasicControllers.py:10-73)
            for (int t = tid; t < N; t += NT) {
                float yaw = T[t].yaw;
                if (p.base_controller == WG_CTRL_LOCAL) {
                    // (steady inflow: v == 0 exactly — the deficits only act on u — so atan(v / u) = +-0; the IEEE division + atanf
                    // are ~60 VALU instructions a baseline-farm step need not execute.  Replay mode scripts (u, v, w) freely.)
                    const float wdir = (TURB == WG_TURB_NONE && !REPLAY) ? 0.f : atanf(T[t].v / T[t].u) * WG_RAD2DEG_F;
                    const float off = wdir - yaw;
                    const float sgn = (float)((off > 0.f) - (off < 0.f));
                    yaw = yaw + sgn * fminf(fabsf(off), p.yaw_step);
                } else {
                    const float sgn = (float)((yaw > 0.f) - (yaw < 0.f));
                    yaw = yaw - sgn * fminf(fabsf(yaw), p.yaw_step);
                }
                T[t].yaw = yaw;
            }
        }
        if (REPLAY && !is_dev) {
            lds_barrier<NT>();
            // sin/cos are not needed in replay mode: power comes from the script
            script_step<NT>(p, d, T, e, farm, cursor, sr.time);
        } else {
            flow_step<NT, TURB, RES, SGM>(p, d, T, tabct, rdy, rdz, pair, tiap, tmask, jnl, pbase, ws, ti_f, ti_pow, tc, sr, pl, n_flow == 0, add_acc);
            // roofline accounting: particles that can still reach a rotor (per lane; summed once in the epilogue)
            for (int t = tid; t < N; t += NT) part_acc += min(sr.n_valid, RES ? T[t].rlen : jnl[t] + 1);
            ++n_flow;
        }
        --budget;
        WG_STAMP(6);
        const bool measuring = !is_dev;
        const bool unit_end = measuring && (sub + 1 == p.K);
        // per-turbine tail (thread t owns turbine t): power / thrust with the current yaw (model M0 step 5),
        // WindFarmEnv._take_measurements (Wind_Farm_Env.py:480-495) accumulated over the k sub-steps and, at
        // the end of the env step, farm_mes.add_measurements' ring push (MesClass.py:568-591)
        const KArgsPtr kc = wg_cold_args();
        float* __restrict__ rbase = kc->d.ring + (size_t)ctx_id * kc->p.ring_stride;
        for (int t = tid; t < N; t += NT) {
            TurbLds& q = T[t];
            if (RES) q.head = q.head_n;          // the step's emissions are in the ring now
            if (!(REPLAY && !is_dev)) {
                const float wsn = fmaxf(q.u * q.cg + q.v * q.sg, 0.0f);
                q.pow = tab_lookup(tabp, p, wsn);
                q.ct = tab_lookup(tabct, p, wsn) * q.cg * q.cg;
            }
            if (measuring && farm == 0) {
                const float wsm = __builtin_amdgcn_sqrtf(q.u * q.u + q.v * q.v + q.w * q.w);
                const float wdm = ((TURB == WG_TURB_NONE && !REPLAY) ? 0.f : atanf(q.v / q.u) * WG_RAD2DEG_F) + wd_env;
                float val[WG_N_CH] = {q.sws + wsm, q.swd + wdm, q.syaw + q.yaw, q.sp + q.pow};
                if (unit_end) {
                    kc->d.cur_ws[(size_t)ctx_id * N + t] = wsm;
                    kc->d.cur_wd[(size_t)ctx_id * N + t] = wdm;
                    if (p.K != 1) {
#pragma unroll
                        for (int ch = 0; ch < WG_N_CH; ++ch) val[ch] *= inv_k;
                    }
                    if (NOISE) {
#pragma unroll
                        for (int ch = 0; ch < WG_N_CH; ++ch)
                            if (kc->p.noise_sigma[ch] != 0.f)
                                val[ch] += kc->p.noise_sigma[ch] * wg_noise_normal(noise_key, (uint32_t)n_pushed, (uint32_t)t,
                                                                                   (uint32_t)ch, episode_tag);
                    }
#pragma unroll
                    for (int ch = 0; ch < WG_N_CH; ++ch) {
                        const int H = kc->p.hlen[ch];
                        rbase[kc->p.ring_off[ch] + umod_small(n_pushed, H, kc->p.hmagic[ch]) * N + t] = val[ch];      // (time-major: WgRing)
                    }
                    // stage the pushed values for the farm-level mean / mean / sum
                    q.sws = val[0]; q.swd = val[1]; q.sp = val[3];
                } else {
                    q.sws = val[0]; q.swd = val[1]; q.syaw = val[2]; q.sp = val[3];
                }
            }
        }
        if (is_dev) { --dev_rem; continue; }
        if (farm == 1 || unit_end) lds_barrier<NT>();
        if (farm == 1 && tid < WG_WAVE) base_acc += WG_TURB_SUM(pow);   // fs_baseline...power().sum() (:954)
        if (++sub < p.K) continue;
        sub = 0;
        if (farm == 0) {
            if (tid < WG_WAVE) {
                const float sws = WG_TURB_SUM(sws), swd = WG_TURB_SUM(swd), tot = WG_TURB_SUM(sp);
                if (tid == 0) {
                    const FlowP __attribute__((address_space(4)))& pc = kc->p;
                    float* fbase = kc->d.fring + (size_t)ctx_id * pc.fring_stride;
                    fbase[pc.fring_off[WG_CH_WS] + umod_small(n_pushed, pc.hlen[WG_CH_WS], pc.hmagic[WG_CH_WS])] = sws * pc.inv_N;
                    fbase[pc.fring_off[WG_CH_WD] + umod_small(n_pushed, pc.hlen[WG_CH_WD], pc.hmagic[WG_CH_WD])] = swd * pc.inv_N;
                    fbase[pc.fring_off[WG_CH_POWER] + umod_small(n_pushed, pc.hlen[WG_CH_POWER], pc.hmagic[WG_CH_POWER])] = tot;
                    if (live_step) kc->d.step_farm_pow[e] = tot;
                    else kc->d.pend_farm[(size_t)ctx_id * pc.power_avg + umod_small(pend_farm_n, pc.power_avg, pc.pavg_magic)] = tot;
                }
            }
            ++n_pushed;
            if (!live_step) ++pend_farm_n;
            lds_barrier<NT>();
            for (int t = tid; t < N; t += NT) { T[t].sws = 0.f; T[t].swd = 0.f; T[t].syaw = 0.f; T[t].sp = 0.f; }
        } else {
            if (tid == 0) {
                const float bp = p.K == 1 ? base_acc : base_acc * inv_k;
                if (live_step) kc->d.step_base_pow[e] = bp;
                else kc->d.pend_base[(size_t)ctx_id * kc->p.power_avg + umod_small(pend_base_n, kc->p.power_avg, kc->p.pavg_magic)] = bp;
            }
            if (!live_step) ++pend_base_n;
            base_acc = 0.f;
        }
        if (live_step) break;
        --fill_rem;
    }

    WG_STAMP(7);
    // epilogue: write the slot back (thread t owns turbine t; no barrier needed for its own T[t])
    const KArgsPtr ke = wg_cold_args();
    for (int t = tid; t < N; t += NT) {
        const TurbLds& q = T[t];
        ke->d.yaw[tb + t] = q.yaw; ke->d.u[tb + t] = q.u; ke->d.v[tb + t] = q.v; ke->d.w[tb + t] = q.w;
        ke->d.ti_loc[tb + t] = q.ti; ke->d.power[tb + t] = q.pow; ke->d.ct[tb + t] = q.ct;
        reinterpret_cast<float4*>(ke->d.bnd)[tb + t] = make_float4(q.bd, q.bk, q.be, __uint_as_float(q.mvl));
        if (live_step && farm == 0) ke->d.old_yaw[(size_t)e * N + t] = oyaw[t];
    }
    if (NT > WG_WAVE) {          // (multi-wave workgroups: per-wave partial sums meet in the LDS word)
        part_acc = wg_wave_sum_i(part_acc);
        if ((tid & 63) == 0) atomicAdd(&jnl[N], part_acc);
        if (TURB != WG_TURB_NONE) { add_acc = wg_wave_sum_i(add_acc); if ((tid & 63) == 0) atomicAdd(&jnl[N + 3], add_acc); }
        lds_barrier<NT>();
        part_acc = jnl[N];
        if (TURB != WG_TURB_NONE) add_acc = jnl[N + 3];
    } else {
        part_acc = wg_wave_sum_i(part_acc);
        if (TURB != WG_TURB_NONE) add_acc = wg_wave_sum_i(add_acc);
    }
    if (tid == 0) {
        WgSlot& slot = ke->d.slot[slot_id];
        WgCtx& cx = ke->d.ctx[ctx_id];
        slot.part_count = part0 + (unsigned)part_acc;
        if (TURB != WG_TURB_NONE && add_acc != 0) slot.add_count += (unsigned)add_acc;
        slot.head = sr.head; slot.n_valid = sr.n_valid; slot.s_off = sr.s_off; slot.time = sr.time;
        slot.istep = sr.istep; slot.n_emitted = sr.n_emitted;
        slot.cursor = cursor;
        slot.dev_remaining = dev_rem; slot.fill_remaining = fill_rem;
        if (farm == 0) { cx.n_pushed = n_pushed; cx.pend_farm_n = pend_farm_n; }
        else cx.pend_base_n = pend_base_n;
        slot.flow_count = flow0 + (unsigned)n_flow;   // NOT a global atomic: one hot word serialises the whole grid
#ifdef WG_TIMELINE
        if (d.dbg && n_flow == 1) {
            wg_stamps[8] = clock64();
            for (int k = 0; k < 16; ++k) d.dbg[(size_t)blockIdx.x * 16 + k] = wg_stamps[k];
        }
#endif
    }
    // a background episode's development ends with this launch: its first observation (wg_first_obs)
    // (compact variants: wg_create leaves next_obs null for every other handle; the builder is single-wave code — wave 0 of a
    // larger workgroup runs it)
    bool built_first_obs = false;
    if (RES) {
        if (mode == WG_MODE_STEP && !live_step && farm == 0 && dev_rem == 0 && fill_rem == 0) {
            full_barrier<NT>();                  // the ring pushes have left the workgroup
            if (NT == WG_WAVE || tid < WG_WAVE) wg_first_obs(ke->d.gp, ke->d.gd, ctx_id, n_pushed, tid);
            built_first_obs = true;
        }
    }
#if defined(WG_TIMELINE) && defined(WG_TIMELINE_WALL)
    // (-DWG_TIMELINE_WALL, tools/timeline_wall.sh: every workgroup that took a step leaves its wall-clock start / end and what it did)
    if (tid == 0 && d.dbg) {
        long long* row = d.dbg + (size_t)blockIdx.x * 16;
        row[15] = wall_clock64(); row[14] = wg_wall0; row[13] = n_flow; row[12] = (init_pending ? 1 : 0) | (built_first_obs ? 2 : 0) | (live_step ? 4 : 0);
        row[0] = 1;
    }
#else
    (void)built_first_obs;
#endif
}

template <int NT, bool RES>
static void launch_nt(const FlowP* p, const FlowPtrs* d, int mode, const float* actions, const uint8_t* mask,
                      int chunk, hipStream_t st) {
    const int grid = p->B * 2 * p->F;
    const size_t lds = p->lds_bytes;
    const bool replay = d->script_uvw != nullptr, noise = p->noise != 0;
#define WG_LAUNCH(TURB, REPLAY, NOISE) \
    hipLaunchKernelGGL((k_flow<NT, TURB, REPLAY, NOISE, RES>), dim3(grid), dim3(NT), lds, st, *p, *d, mode, actions, mask, chunk)
    const int turb = (p->turb_mode >= WG_TURB_BOX) ? WG_TURB_BOX : p->turb_mode;
    if constexpr (RES) {
        if (p->deficit_model != 0 && !replay) {      // super-Gaussian / eddy-viscosity instantiations (compact variants only)
#define WG_LAUNCH_DM(TURB, NOISE, DM) \
    hipLaunchKernelGGL((k_flow<NT, TURB, false, NOISE, true, DM>), dim3(grid), dim3(NT), lds, st, *p, *d, mode, actions, mask, chunk)
#define WG_LAUNCH_SG(TURB, NOISE) do { if (p->deficit_model == 1) WG_LAUNCH_DM(TURB, NOISE, 1); else WG_LAUNCH_DM(TURB, NOISE, 2); } while (0)
            if (turb == WG_TURB_NONE) { if (noise) WG_LAUNCH_SG(WG_TURB_NONE, true); else WG_LAUNCH_SG(WG_TURB_NONE, false); }
            else if (turb == WG_TURB_RANDOM) { if (noise) WG_LAUNCH_SG(WG_TURB_RANDOM, true); else WG_LAUNCH_SG(WG_TURB_RANDOM, false); }
            else { if (noise) WG_LAUNCH_SG(WG_TURB_BOX, true); else WG_LAUNCH_SG(WG_TURB_BOX, false); }
#undef WG_LAUNCH_SG
#undef WG_LAUNCH_DM
            return;
        }
    }
    if (replay) {                       // replay mode ignores the physics
        if (noise) WG_LAUNCH(WG_TURB_NONE, true, true); else WG_LAUNCH(WG_TURB_NONE, true, false);
    } else if (turb == WG_TURB_NONE) {
        if (noise) WG_LAUNCH(WG_TURB_NONE, false, true); else WG_LAUNCH(WG_TURB_NONE, false, false);
    } else if (turb == WG_TURB_RANDOM) {
        if (noise) WG_LAUNCH(WG_TURB_RANDOM, false, true); else WG_LAUNCH(WG_TURB_RANDOM, false, false);
    } else {
        if (noise) WG_LAUNCH(WG_TURB_BOX, false, true); else WG_LAUNCH(WG_TURB_BOX, false, false);
    }
#undef WG_LAUNCH
}

// One workgroup per farm slot.  Small farms (N <= 32): compact per-turbine rings + pair-major deficit phases, 64 or 128
// threads; large farms: uniform rings with predicate pruning + (target, sample)-major deficit phases, 256 threads.
// Chosen on the host (FlowP.res / FlowP.block).
extern "C" void wg_launch_flow_env(const FlowP*, const FlowPtrs*, int, const float*, const uint8_t*, int, hipStream_t);
extern "C" void wg_launch_flow_envb(const FlowP*, const FlowPtrs*, int, const float*, const uint8_t*, int, hipStream_t);
extern "C" void wg_launch_flow(const FlowP* p, const FlowPtrs* d, int mode, const float* actions,
                               const uint8_t* mask, int chunk, hipStream_t st) {
    if (p->envw && d->script_uvw == nullptr) {      // one / two waves per env: steady inflow (wg_env.hip) / frozen box (wg_envb.hip)
        if (p->turb_mode == WG_TURB_NONE) wg_launch_flow_env(p, d, mode, actions, mask, chunk, st);
        else wg_launch_flow_envb(p, d, mode, actions, mask, chunk, st);
        return;
    }
    if (p->res) {
        if (p->block == 64) launch_nt<64, true>(p, d, mode, actions, mask, chunk, st);
        else if (p->block == 128) launch_nt<128, true>(p, d, mode, actions, mask, chunk, st);
        else launch_nt<256, true>(p, d, mode, actions, mask, chunk, st);
    } else {
        launch_nt<256, false>(p, d, mode, actions, mask, chunk, st);
    }
}

// ===================================================================================================
// one sample of the tabulated eddy-viscosity deficit (deficit_model 2; the pair evaluation of flow_step inlines the same
// rule per wake): 1 - U / U0 at (Ct, TI_amb from the frozen growth rate, x / D, r / R)
__device__ float an_point(const FlowP& p, const float* __restrict__ tab, const float ctv, const float kv, const float xd, const float rR) {
    const float fr = rR * p.an_inv_dr;
    if (!(fr < (float)p.an_r - 1.5f)) return 0.f;
    const float tiw = fmaxf((kv - p.kb) * p.an_inv_ka, 1e-6f);
    const float fc = fminf(fmaxf((ctv - p.an_ct0) * p.an_inv_dct, 0.f), (float)(p.an_ct - 1));
    const float ft = fminf(fmaxf((__logf(tiw) - p.an_lti0) * p.an_inv_dlti, 0.f), (float)(p.an_ti - 1));
    const float fx = fminf(fmaxf(xd * p.an_inv_dx, 0.f), (float)(p.an_x - 1));
    const int ic = min((int)fc, p.an_ct - 2), it = min((int)ft, p.an_ti - 2), ix = min((int)fx, p.an_x - 2);
    const float wc = fc - (float)ic, wt = ft - (float)it, wx = fx - (float)ix;
    const int sx = p.an_r, st = p.an_x * sx, sc = p.an_ti * st;
    const int base = ic * sc + it * st + ix * sx;
    const int m = (int)(fr + 0.5f), ml = m > 0 ? m - 1 : 1;
    float fa = 0.f, fb = 0.f, fcn = 0.f;
    for (int q = 0; q < 8; ++q) {
        const int o = base + (q & 1 ? sc : 0) + (q & 2 ? st : 0) + (q & 4 ? sx : 0);
        const float wq = (q & 1 ? wc : 1.0f - wc) * (q & 2 ? wt : 1.0f - wt) * (q & 4 ? wx : 1.0f - wx);
        fa += wq * tab[o + ml]; fb += wq * tab[o + m]; fcn += wq * tab[o + m + 1];
    }
    const float e = fr - (float)m;
    return e < 0.f ? fb + e * (fb - fa) : fb + e * (fcn - fb);
}

// k_windspeed: flow-field view — (u, v, w) on an XY grid at height z of ONE farm of ONE env, in the flow frame:
// fs.get_windspeed(XYView(z, x, y), include_wakes) behind WindFarmEnv._render_frame / init_render
// (Wind_Farm_Env.py:1040-1083, :464-476).  One thread per grid point; the same bracketed-chain evaluation as phase A
// of k_flow, without the lateral cut-off.  "Random" inflow has no spatial field and adds nothing here.
// ===================================================================================================
__global__ void __launch_bounds__(256)
k_windspeed(const FlowP p, const FlowPtrs d, const int e, const int farm, const float* __restrict__ xs, const int nx,
            const float* __restrict__ ys, const int ny, const float z, const int include_wakes,
            float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nx * ny) return;
    const int ix = idx / ny, iy = idx - ix * ny;
    const double px = (double)xs[ix], pyd = (double)ys[iy];
    const int N = p.N, P = p.P;
    const int ctx_id = e * 2 + d.env[e].live;
    const int slot_id = ctx_id * p.F + farm;
    const size_t pbase = (size_t)slot_id * p.pstride;
    const WgSlot& slot = d.slot[slot_id];
    const WgCtx& cx = d.ctx[ctx_id];
    const double s_off = slot.s_off;
    const int head = slot.head, n_valid = slot.n_valid;
    float amb[3] = {0.f, 0.f, 0.f};
    const bool boxm = p.turb_mode >= WG_TURB_BOX && d.box4 != nullptr;
    if (boxm) {
        const double bx = px - cx.ws * slot.time + cx.box_ox, by = pyd + cx.box_oy;
        const float4* b4 = d.box4 + (size_t)cx.box_id * p.box_cells;
        if (p.box_pow2) box_lookup<true>(b4, p, bx, by, (double)z, amb);
        else box_lookup<false>(b4, p, bx, by, (double)z, amb);
        const float sig = (float)(cx.ti * cx.ws);
        amb[0] *= sig; amb[1] *= sig; amb[2] *= sig;
    }
    float dsum = 0.f;
    const float yp = (float)pyd;
    for (int s2 = 0; include_wakes && s2 < N; ++s2) {
        const double dx = px - d.xr[(size_t)ctx_id * N + s2];
        if (!(dx > 0.0)) continue;
        const double xi = (dx - s_off) * p.inv_dpart;
        const double jf = floor(xi);
        float wgt = (float)(xi - jf);
        long long j = (long long)jf;
        if (j < 0) { j = 0; wgt = 0.f; }
        if (j + 1 > n_valid - 1) continue;
        size_t i0, i1;
        if (p.res) {
            // compact rings: turbine s2 keeps its R youngest particles (those that can still reach a rotor); the field
            // behind the last turbine row needs wg_config.full_chains
            const int* ro = d.roff + (size_t)ctx_id * (N + 1);
            const int R = ro[s2 + 1] - ro[s2];
            if (j + 1 >= R) continue;
            const int hd = (int)((slot.n_emitted - 1u) % (unsigned)R);       // n_valid > 0 here: something was emitted
            int r0 = hd - (int)j; if (r0 < 0) r0 += R;
            int r1 = r0 - 1; if (r1 < 0) r1 += R;
            i0 = pbase + (size_t)ro[s2] + r0; i1 = pbase + (size_t)ro[s2] + r1;
        } else {
            int r0 = head - (int)j; if (r0 < 0) r0 += P;
            int r1 = r0 - 1; if (r1 < 0) r1 += P;
            i0 = pbase + (size_t)s2 * P + r0; i1 = pbase + (size_t)s2 * P + r1;
        }
        const size_t rs = p.rec_il ? 2 : 1;      // (interleaved record: rec_b = rec_a + 1)
        const unsigned a0 = d.rec_a[i0 * rs], a1 = d.rec_a[i1 * rs], b0 = d.rec_b[i0 * rs], b1 = d.rec_b[i1 * rs];
        const float ue_max_ = p.ue_scale * (float)cx.ws;
        const float ue0 = rec_uf(b0) * ue_max_, ue1 = rec_uf(b1) * ue_max_;
        const float w0 = 1.0f - wgt, w1 = wgt;
        const float yc = w0 * d.py[i0] + w1 * d.py[i1];
        float zc = p.hub;
        if (p.turb_mode != WG_TURB_NONE) zc = w0 * d.pz[i0] + w1 * d.pz[i1];
        const float ctv = w0 * rec_ct(a0) + w1 * rec_ct(a1);
        const float kv = w0 * rec_k(a0) + w1 * rec_k(a1);
        const float epv = w0 * rec_eps(a0, p.eps0) + w1 * rec_eps(a1, p.eps0);
        const float uev = w0 * ue0 + w1 * ue1;
        const float sp = kv * ((float)dx * p.inv_D) + epv;
        const float sig = sp * p.D;
        const float inv2s2 = 1.0f / (2.0f * sig * sig);
        const float r2 = (yp - yc) * (yp - yc) + (z - zc) * (z - zc);
        if (p.deficit_model == 2) {
            dsum += uev * an_point(p, d.dtab, ctv, kv, (float)dx * p.inv_D, sqrtf(r2) * 2.0f * p.inv_D);
        } else if (p.deficit_model == 1) {
            const float xd = (float)dx * p.inv_D;
            const float nsg = p.sg_af * __expf(p.sg_bf * xd) + p.sg_cf, in2 = 2.0f / nsg;
            const float rad = exp2f(2.0f * in2 - 2.0f) - nsg * ctv / (16.0f * tgammaf(in2) * powf(sp, 2.0f * in2));
            const float cf = exp2f(in2 - 1.0f) - sqrtf(fmaxf(rad, 0.0f));
            const float rn = r2 > 0.f ? powf(r2 * p.inv_D * p.inv_D, 0.5f * nsg) : 0.f;
            dsum += uev * cf * __expf(-rn / (2.0f * sp * sp));
        } else {
            dsum += uev * m0_cfrac(ctv, sp) * __expf(-r2 * inv2s2);
        }
    }
    const size_t plane = (size_t)nx * ny;
    out[idx] = (float)cx.ws + amb[0] - dsum;
    out[plane + idx] = amb[1];
    out[2 * plane + idx] = amb[2];
}

extern "C" void wg_launch_windspeed(const FlowP* p, const FlowPtrs* d, int e, int farm, const float* xs, int nx,
                                    const float* ys, int ny, float z, int include_wakes, float* out, hipStream_t st) {
    const int n = nx * ny;
    hipLaunchKernelGGL(k_windspeed, dim3((n + 255) / 256), dim3(256), 0, st, *p, *d, e, farm, xs, nx, ys, ny, z,
                       include_wakes, out);
}
