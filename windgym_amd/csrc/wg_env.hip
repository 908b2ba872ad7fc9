// wg_env.hip — k_flow_env: ONE single-wave workgroup per env (gfx950, wave64); steady inflow, small farms.
//
// Why.  With one workgroup per farm slot (k_flow, GL variant) a 16-turbine farm keeps 16 of 64 lanes busy in every
// turbine-indexed phase (state load, emission records, power / measurement tail, superposition, state store), a live env
// costs two such waves plus ~0.9 background ones per step, and each wave walks its own ~30 k-cycle chain of dependent
// phases: the launch was VALU-issue bound at 66 % with a quarter of the lanes doing the per-turbine work.
// Here lane g = slot * N + turbine serves turbine t of farm slot `slot` = ctx * F + farm of ONE env: the running episode's
// agent and baseline farms and the background episode's two farms fill the wave (4 x 16 = 64 lanes on cfg2, 4 x 9 = 36 on
// cfg4), every turbine-indexed phase runs once for all of them, the candidate pairs and ring quads of all stepping slots
// share one work list each, and the turbine state of an env is ONE contiguous run of lanes in every array (slot-major
// layout: slot_id * N + t = env * 2 F N + lane).  The wave owns the whole env, so the env's glue can follow in the same
// wave (k_step_env below: step() as one launch, no cross-workgroup dependency).
//
// Same state layout (interleaved packed record + 16-byte gather copy: FlowP::rec_il), same arithmetic per element and the
// same summation orders as k_flow<64, NONE, false, NOISE, true> (GL): the two kernels are interchangeable launch by
// launch on the same handle, which is how the tests compare them (bit for bit).
//
// Order of a flow step (GL's): clocks -> candidate pass over the stepping slots (float positions, running
// chain bounds, ballots -> list in ascending (target lane, source) order, per-target source masks in registers) -> bracket
// gathers of the first 64 candidates requested -> emission records -> exact evaluation, one candidate per lane, the next
// batch requested before the current one is evaluated -> per-target sums in ascending source order -> quad list of the
// chains that move -> software-pipelined advection pass -> per-turbine tail (power, measurement, ring push).
// Replaces DWMFlowSimulation.step() + rotor_avg_windspeed + power() + BasicControllers + _take_measurements +
// farm_mes.add_measurements (Wind_Farm_Env.py:480-495, 822-864, 943-979; BasicControllers.py:10-73; MesClass.py:568-591).
#include <hip/hip_runtime.h>

#include "wg_flow_dev.h"

#ifndef WG_ENV_S_UNROLL
#define WG_ENV_S_UNROLL 0   // 1: the rotor-point loop of the pair evaluation unrolled by 4
#endif
#ifndef WG_ENV_WAVES
#define WG_ENV_WAVES 4      // 128 VGPRs: 4096 envs = the chip's 4096 wave slots at 4 waves per SIMD, one dispatch round
#endif

// per-slot clock of the current flow step, published by the slot's lane t = 0 for the work items of other lanes
struct __attribute__((aligned(8))) EnvSlotLds {
    double s_new;
    int n_emit, n_valid, new_valid;
    unsigned n_emitted;
    float s_off_f, near_f, move_max, ti_pow;
};
static_assert(sizeof(EnvSlotLds) == WG_ENV_SLOT_LDS_BYTES, "keep WG_ENV_SLOT_LDS_BYTES in sync (wg_flow.h)");

// rare path at the head of the launch: the episode a retired context will hold (WgCtx::init_pending), both farms at once
static __device__ __attribute__((noinline)) void env_init_episode(const WgParams* gp, const WgPtrs* gd, WgEnv* env_rw,
                                                                  const int e, const int c, const int lane) {
    const WgCtx& cx = gd->ctx[e * 2 + c];
    WgRng rng{cx.snap_state, cx.snap_inc, cx.snap_has32, cx.snap_u32};
    wg_ctx_init(*gp, *gd, rng, e, c, lane, cx.episode_tag, 0, gp->F);
    if (lane == 0) {
        env_rw->rng_state = rng.rng_state; env_rw->rng_inc = rng.rng_inc;
        env_rw->rng_has32 = rng.rng_has32; env_rw->rng_u32 = rng.rng_u32;
    }
}

// sum of a per-lane value over the N lanes of slot k, in the association k_flow's one-farm wave uses (its lanes 0 .. N-1:
// row 0 of the DPP tree, the other lanes hold 0).  Result valid in every lane.
__device__ __forceinline__ float env_slot_sum(const float v, const int k, const int N, const int tid) {
    const float r = __shfl(v, (tid + k * N) & 63, 64);
    const float s = tid < N ? r : 0.f;
    return N <= 16 ? __shfl(wg_row_sum(s), 0, 64) : wg_wave_sum(s);
}

template <bool NOISE>
__global__ void __launch_bounds__(64, WG_ENV_WAVES)
k_flow_env(const FlowP p, const FlowPtrs d, const int mode, const float* __restrict__ actions,
           const uint8_t* __restrict__ mask, const int chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, e = blockIdx.x;
    const int N = p.N, F = p.F, P = p.P, NS = 2 * F, NL = NS * N;
    const bool valid = tid < NL;
    const int g = valid ? tid : 0;
    const int k = (int)(((float)g + 0.5f) * p.inv_N);      // slot of the env: ctx * F + farm
    const int t = g - k * N;
    const int c = F == 2 ? (k >> 1) : k, farm = F == 2 ? (k & 1) : 0;
    const int ctx_id = e * 2 + c, slot_id = e * NS + k;
    const size_t tb = (size_t)e * NL + g;                   // == slot_id * N + t
    const size_t pb_env = (size_t)e * NS * p.pstride;       // particle block of the env's slot 0 (slot k: + k * pstride)

    WG_STAMP(0);
    // ---- prologue: every independent global load up front (one exposed round trip) ----------------------------------
    const KArgsPtr k0 = wg_cold_args();
    typedef const __attribute__((address_space(4))) WgEnv* CEnvPtr;
    const CEnvPtr envc = (CEnvPtr)(d.env + e);
    const int env_live = envc->live, env_done = envc->done, env_shadow_iters = envc->shadow_iters, env_steps_done = envc->steps_done;
    const uint64_t noise_key = envc->noise_key;
    const bool use_mask = mode == WG_MODE_RESET && mask != nullptr;
    const uint8_t mask_byte = *(use_mask ? mask + e : reinterpret_cast<const uint8_t*>(d.env + e));
    const bool masked_out = use_mask && mask_byte == 0;

    int dev_rem, fill_rem, cursor, n_pushed, pend_farm_n, pend_base_n, init_pending, time_max_c;
    int s_head, n_valid;
    unsigned part0, flow0, istep, n_emitted;
    double s_off, s_time, ws, l_xr, l_yr;
    float ti_f, wd_env, l_yaw, l_u, l_v, l_w, l_ti, l_pow, l_ct, l_act;
    float4 l_bnd;
    uint32_t episode_tag;
    int l_roff, l_rnext;
    auto load_state = [&]() __attribute__((always_inline)) {
        const KArgsPtr kl = wg_cold_args();
        const WgSlot& slot = kl->d.slot[slot_id];
        const WgCtx& cx = kl->d.ctx[ctx_id];
        dev_rem = slot.dev_remaining; fill_rem = slot.fill_remaining; cursor = slot.cursor;
        s_off = slot.s_off; s_time = slot.time; s_head = slot.head; n_valid = slot.n_valid; istep = slot.istep;
        n_emitted = slot.n_emitted; part0 = slot.part_count; flow0 = slot.flow_count;
        ws = cx.ws; ti_f = (float)cx.ti; wd_env = (float)cx.wd;
        n_pushed = cx.n_pushed; pend_farm_n = cx.pend_farm_n; pend_base_n = cx.pend_base_n;
        episode_tag = (uint32_t)cx.episode_tag; init_pending = cx.init_pending; time_max_c = cx.time_max;
        const int* ro = kl->d.roff + (size_t)ctx_id * (N + 1);
        l_roff = ro[t]; l_rnext = ro[t + 1];
        l_xr = kl->d.xr[(size_t)ctx_id * N + t]; l_yr = kl->d.yr[(size_t)ctx_id * N + t];
        l_yaw = kl->d.yaw[tb]; l_u = kl->d.u[tb]; l_v = kl->d.v[tb]; l_w = kl->d.w[tb];
        l_ti = kl->d.ti_loc[tb]; l_pow = kl->d.power[tb]; l_ct = kl->d.ct[tb];
        l_bnd = reinterpret_cast<const float4*>(kl->d.bnd)[tb];
    };
    load_state();
    const int i_tab = min(tid, p.n_tab - 1), i_s = min(tid, p.S - 1);
    const float pf_tp = k0->d.tab_power[i_tab], pf_tc = k0->d.tab_ct[i_tab];
    const float pf_dy = k0->d.rotor_dy[i_s], pf_dz = k0->d.rotor_dz[i_s];
    {   // (every agent-farm lane reads its turbine's action; only the running episode's lanes use it)
        const bool has_act = mode == WG_MODE_STEP && farm == 0;
        const float a = *(has_act ? actions + (size_t)e * N + t : k0->d.tab_ct);
        l_act = has_act ? a : 0.f;
    }

    // ---- roles --------------------------------------------------------------------------------------------------------
    // role_live: this lane's slot belongs to the running episode and takes one env step (K flow sub-steps with measurement);
    // role_dev:  its slot develops a not-yet-live episode for `budget` flow steps — the background context in STEP mode
    //            (Wind_Farm_Env.py:722-796 hidden behind the running episode), the masked envs' live
    //            context in RESET mode
    const bool is_live_c = (c == env_live);
    bool role_live = false, role_dev = false;
    int budget = 0;
    if (mode == WG_MODE_STEP) {
        role_live = is_live_c && !env_done;
        role_dev = !is_live_c && p.autoreset != 0;
        // (wave-uniform: both contexts' lanes see the background context's flag through their own loads)
        const int bg_pending = __shfl(init_pending, (env_live ^ 1) * F * N, 64);
        if (p.autoreset && bg_pending) {
            // rare path (one context per truncation): set the retired context's next episode up, both farms (see k_flow)
            const WgParams& gp = *d.gp;
            env_init_episode(d.gp, d.gd, d.env_rw + e, e, env_live ^ 1, tid);
            full_barrier<64>();
            load_state();
            const int fill_max = F == 2 ? max(gp.fill_a, gp.fill_b) : gp.fill_a;
            const int inc = 1 + (gp.extra_inc ? 1 : 0);
            const int tm = __shfl(time_max_c, env_live * F * N, 64);
            const long total = (long)((tm + inc - 1) / inc) + 1;
            const int dev0 = __shfl(dev_rem, (env_live ^ 1) * F * N, 64);
            budget = wg_shadow_share(dev0 + p.K * fill_max, total - env_steps_done, env_steps_done, e);
        } else {
            budget = env_shadow_iters;
        }
    } else {
        role_dev = is_live_c && !masked_out;
        budget = chunk;
    }
    {   // nothing to do for the whole env (masked out in RESET mode, finished env without autoreset, idle background)
        const bool any_work = valid && (role_live || (role_dev && budget > 0 && (dev_rem > 0 || fill_rem > 0)));
        if (!__ballot(any_work)) return;
    }

    // ---- LDS carve (host mirror: wg_create, FlowP::env_*) ---------------------------------------------------------------
    char* stage = smem;                                   // candidate list | per-candidate deficit, added TI  ∪  quad list
    unsigned short* cl = reinterpret_cast<unsigned short*>(stage);
    float* def = reinterpret_cast<float*>(stage + p.env_off_def);
    float* tiav = def + p.env_cap;
    unsigned short* ql = reinterpret_cast<unsigned short*>(stage);
    double* Lxr = reinterpret_cast<double*>(smem + p.env_off_turb);
    double* Lyr = Lxr + 64;
    float* Lxf = reinterpret_cast<float*>(Lyr + 64);
    float* Lyf = Lxf + 64;
    float* Lbd = Lyf + 64;
    float* Lbk = Lbd + 64;
    float* Lbe = Lbk + 64;
    unsigned* Lmvl = reinterpret_cast<unsigned*>(Lbe + 64);
    unsigned* Lra = Lmvl + 64;                            // this step's emission record, packed (pack_a / pack_b)
    unsigned* Lrb = Lra + 64;
    float* Lrue = reinterpret_cast<float*>(Lrb + 64);
    float* Lcg = Lrue + 64;
    int* Lroff = reinterpret_cast<int*>(Lcg + 64);
    int* Lrlen = Lroff + 64;
    int* Lhead = Lrlen + 64;
    EnvSlotLds* SL = reinterpret_cast<EnvSlotLds*>(Lhead + 64);
    float* tabp = reinterpret_cast<float*>(SL + 4);
    float* tabct = tabp + p.n_tab;
    float* rdy = tabct + p.n_tab;
    float* rdz = rdy + p.S;

    const int rlen = l_rnext - l_roff;
    int head = n_emitted == 0u ? rlen - 1 : fast_mod((int)(n_emitted - 1u), rlen, __builtin_amdgcn_rcpf((float)rlen));
    float bd0 = l_bnd.x;        // (Lbd is the live copy: the advection pass raises it with LDS atomics)
    float bk = l_bnd.y, be = l_bnd.z;
    unsigned mvl = __float_as_uint(l_bnd.w);
    if (valid) {
        Lxr[g] = l_xr; Lyr[g] = l_yr; Lxf[g] = (float)l_xr; Lyf[g] = (float)l_yr;
        Lbd[g] = bd0; Lbk[g] = bk; Lbe[g] = be; Lmvl[g] = mvl;
        Lroff[g] = l_roff; Lrlen[g] = rlen; Lhead[g] = head;
        Lcg[g] = 1.f; Lrue[g] = 0.f; Lra[g] = 0u; Lrb[g] = 0u;
    }
    if (tid < p.n_tab) { tabp[tid] = pf_tp; tabct[tid] = pf_tc; }
    if (tid < p.S) { rdy[tid] = pf_dy; rdz[tid] = pf_dz; }
    for (int i = tid + 64; i < p.n_tab; i += 64) { tabp[i] = k0->d.tab_power[i]; tabct[i] = k0->d.tab_ct[i]; }
    for (int i = tid + 64; i < p.S; i += 64) { rdy[i] = k0->d.rotor_dy[i]; rdz[i] = k0->d.rotor_dz[i]; }

    const float ti_pow = fast_pow(ti_f, p.tic);
    const float ws_f = (float)ws;
    const float yt_f = (float)l_yr;
    // this lane's turbine (registers; the cross-lane fields live in the LDS arrays above)
    float yaw = l_yaw, tu = l_u, tv = l_v, tw = l_w, tti = l_ti, tpow = l_pow, tct = l_ct;
    float cg = 1.f, sg = 0.f;
    float sws = 0.f, swd = 0.f, syaw = 0.f, sp_ = 0.f;
    float oyaw = l_yaw;

    // WindFarmEnv._adjust_yaws (Wind_Farm_Env.py:822-864), float32 like numpy evaluates it on a float32 action
    if (role_live && farm == 0 && valid) {
        const float a = l_act;
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
    }
    lds_barrier<64>();
    WG_STAMP(1);

    int sub = 0, n_flow = 0, part_acc = 0;
    float base_acc = 0.f;
    bool dev_stepped = false;
    const float inv_k = 1.0f / (float)p.K;
    const int NN = N * N;
    const float inv_NN = p.inv_N * p.inv_N;
    const unsigned maskN = N >= 32 ? 0xffffffffu : (1u << N) - 1u;
    const unsigned long long lane_lt = (1ull << tid) - 1ull;

    for (int round = 0;; ++round) {
        const bool active = dev_rem > 0 || fill_rem > 0;
        const bool stepping = valid && (role_live ? (round < p.K) : (role_dev && active && (budget > 0 || sub != 0)));
        const unsigned long long smask = __ballot(stepping);
        if (!smask) break;
        // (a further flow step of the launch gathers what the previous one's advection pass stored)
        if (round > 0) full_barrier<64>();
        const bool is_dev = !role_live && dev_rem > 0;
        // the stepping slots, ascending, two bits each, and this lane's rank among them
        int na = 0, arank = 0;
        unsigned code = 0u;
        for (int kk = 0; kk < NS; ++kk) {
            if ((smask >> (kk * N)) & 1ull) {
                code |= (unsigned)kk << (2 * na);
                if (kk < k) ++arank;
                ++na;
            }
        }

        // BasicControllers.local_yaw_controller / global_yaw_controller (BasicControllers.py:10-73): the running episode's
        // baseline farm, every sim sub-step, not clipped
        if (role_live && farm == 1 && stepping) {
            if (p.base_controller == WG_CTRL_LOCAL) {
                // (steady inflow: v == 0 exactly — the deficits only act on u — so atan(v / u) = +-0)
                const float off = 0.f - yaw;
                const float sgn = (float)((off > 0.f) - (off < 0.f));
                yaw = yaw + sgn * fminf(fabsf(off), p.yaw_step);
            } else {
                const float sgn = (float)((yaw > 0.f) - (yaw < 0.f));
                yaw = yaw - sgn * fminf(fabsf(yaw), p.yaw_step);
            }
        }

        // (0) the slot's clock: travel of its chains over this step, particles released
        double s_new = s_off + ws * p.dt_d;
        int n_emit = 0;
        if (stepping) {
            while (s_new >= p.dpart) { s_new -= p.dpart; ++n_emit; }
            if (n_emit > P) n_emit = P;
        }
        int new_head = s_head + n_emit; if (new_head >= P) new_head -= P;
        int new_valid = n_valid + n_emit; if (new_valid > P) new_valid = P;
        const float s_off_f = (float)s_off;
        if (stepping && t == 0) {
            EnvSlotLds& q = SL[k];
            q.s_new = s_new; q.n_emit = n_emit; q.n_valid = n_valid; q.new_valid = new_valid; q.n_emitted = n_emitted;
            q.s_off_f = s_off_f;
            // a target closer than this is bracketed by a particle released in this step
            q.near_f = (float)(s_off + ws * p.dt_d + p.dpart) + 0.01f;
            q.move_max = fabsf(p.hill) * ws_f * p.dt;
            q.ti_pow = ti_pow;
        }
        lds_barrier<64>();
        WG_STAMP(4);

        // (2) candidate pass: every (target, source) pair of the stepping slots against the chains' running bounds (those
        // BEFORE this step's records: they cover every particle already in the rings).  A pair close enough to be bracketed
        // by a particle released in this step is tested against the widest record the packed format can hold.
        int nc = 0;
        unsigned cmask = 0u;          // lane = target: its candidate sources
        const int npairs = na * NN;
        const int my_first = stepping ? (arank * N + t) * N : -(1 << 20);
        for (int i0 = 0; i0 < npairs; i0 += 4 * 64) {
            bool cand[4];
            int ent[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 64 + tid;
                cand[u] = false; ent[u] = 0;
                if (i < npairs) {
                    const int a = (int)(((float)i + 0.5f) * inv_NN);
                    const int r = i - a * NN;
                    const int tl = (int)(((float)r + 0.5f) * p.inv_N);
                    const int s2 = r - tl * N;
                    const int kk = (int)((code >> (2 * a)) & 3u);
                    const int gt = kk * N + tl, gs = kk * N + s2;
                    const float dxf = Lxf[gt] - Lxf[gs];
                    bool cd = (s2 != tl) && (dxf >= 0.f);
                    if (cd) {
                        const EnvSlotLds& q = SL[kk];
                        const bool nearp = dxf < q.near_f;
                        const float kb_ = nearp ? fmaxf(WG_K_MAX, Lbk[gs]) : Lbk[gs], eb_ = nearp ? fmaxf(p.env_eps_max, Lbe[gs]) : Lbe[gs];
                        const float sig_max = (kb_ * (dxf * p.inv_D) + eb_) * p.D;
                        const float bdv = Lbd[gs] + (Lmvl[gs] != 0u || nearp ? q.move_max : 0.f);
                        const float gap = fabsf(Lyf[gt] - Lyf[gs]) - (p.R_rot + 5.0f * sig_max + bdv);
                        cd = gap <= 1.0e-3f * p.D;
                    }
                    cand[u] = cd;
                    ent[u] = (gt << 5) | s2;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ib = i0 + u * 64;
                const unsigned long long bal = __ballot(cand[u]);
                if (cand[u]) cl[nc + __popcll(bal & lane_lt)] = (unsigned short)ent[u];
                nc += __popcll(bal);
                const int lo = my_first - ib;
                if (lo < 64 && lo + N > 0) cmask |= (unsigned)(lo >= 0 ? (bal >> lo) : (bal << -lo)) & maskN;
            }
        }
        // this target's range of the list: the candidates are in ascending (target lane, source) order
        int cbeg;
        {
            const int cnt = __popc(cmask);
            int inc = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (tid >= o) inc += v; }
            cbeg = inc - cnt;
        }
        lds_barrier<64>();
        WG_STAMP(5);

        // bracket of candidate c and the gathers of its two bracketing particles in their PRE-step state (advanced by the same
        // m0_advect() the advection pass applies: the phase depends on nothing the pass writes).  Ages j, j + 1 after the
        // step are ages jp0 = j - n_emit, jp0 + 1 before it; negative: released in this step — the turbine's record, at the
        // turbine.  A resting chain's particles sit where they were released: their py is not fetched.
        struct Cand { uint4 q0, q1; float y0, y1; double dx; float wgt; int gt, gs, jp0, pos; bool ok, rest; };
        auto issue = [&](Cand& cd, const int cidx, const int c0, const int c1) __attribute__((always_inline)) {
            cd.ok = false; cd.rest = false; cd.pos = cidx - c0;
            if (cidx >= c1) return;
            def[cd.pos] = 0.f; tiav[cd.pos] = 0.f;          // a candidate the exact evaluation drops contributes zero
            const unsigned en = cl[cidx];
            cd.gt = (int)(en >> 5);
            const int kk = (int)(((float)cd.gt + 0.5f) * p.inv_N);
            cd.gs = kk * N + (int)(en & 31u);
            const EnvSlotLds& q = SL[kk];
            cd.dx = Lxr[cd.gt] - Lxr[cd.gs];
            if (!(cd.dx > 0.0)) return;                       // (the candidate test ran on float positions)
            const double xi = (cd.dx - q.s_new) * p.inv_dpart;
            const double jf = floor(xi);
            cd.wgt = (float)(xi - jf);
            int j = (int)jf;
            if (j < 0) { j = 0; cd.wgt = 0.f; }
            if (j + 1 > q.new_valid - 1) return;              // the chain has not reached the target yet
            const int Rs = Lrlen[cd.gs], hd = Lhead[cd.gs];
            const unsigned mv = Lmvl[cd.gs];
            cd.rest = !(mv != 0u && (int)(q.n_emitted - mv) < Rs);
            cd.jp0 = j - q.n_emit;
            int r0 = hd - cd.jp0; if (r0 < 0) r0 += Rs;
            int r1 = hd - cd.jp0 - 1; if (r1 < 0) r1 += Rs;
            if (cd.jp0 < 0) r0 = 0;           // (released in this step: nothing to fetch — any slot of the ring will do)
            if (cd.jp0 + 1 < 0) r1 = 0;
            const size_t sb = pb_env + (size_t)kk * p.pstride + (size_t)Lroff[cd.gs];
            const uint4* r4 = d.rec4 + sb;
            cd.q0 = r4[r0]; cd.q1 = r4[r1];
            cd.y0 = 0.f; cd.y1 = 0.f;
            if (!cd.rest) { const float* py = d.py + sb; cd.y0 = py[r0]; cd.y1 = py[r1]; }
            cd.ok = true;
        };
        const int cap = p.env_cap;
        Cand nxt;
        issue(nxt, tid, 0, min(nc, cap));
        WG_STAMP(11);

        // (1) emission records of this step, sin / cos of the yaw
        if (stepping) {
            const float gy = yaw * WG_DEG2RAD_F;
            sg = __sinf(gy); cg = __cosf(gy);
            const float wsn = fmaxf(tu * cg + tv * sg, 0.0f);
            const float ctx = fminf(fmaxf(tab_lookup(tabct, p, wsn) * cg * cg, 0.0f), 0.96f);
            const float rq = __builtin_amdgcn_sqrtf(1.0f - ctx);
            const float beta = 0.5f * (1.0f + rq) * __builtin_amdgcn_rcpf(rq);
            const float rk = p.ka * tti + p.kb;
            const float reps = p.eps0 * __builtin_amdgcn_sqrtf(beta);
            const float rhv = -p.hill * sg * tu;
            // the packed record saturates outside [0, WG_K_MAX] x [-WG_HV_MAX, WG_HV_MAX]: never silently (wg_check reports it)
            if (rk > WG_K_MAX || fabsf(rhv) > WG_HV_MAX) atomicOr(wg_cold_args()->d.status, WG_STATUS_BIT_RANGE);
            const unsigned na_ = pack_a(ctx, rk), nb_ = pack_b(reps, rhv);
            bk = fmaxf(bk, rk + WG_K_MAX / 65535.0f);
            be = fmaxf(be, reps + 1.0f / 65535.0f);
            if (n_emit > 0 && rec_moves(nb_)) mvl = n_emitted + (unsigned)n_emit;
            Lra[g] = na_; Lrb[g] = nb_; Lrue[g] = tu; Lcg[g] = cg;
            Lbk[g] = bk; Lbe[g] = be; Lmvl[g] = mvl;
        }
        lds_barrier<64>();
        WG_STAMP(2);

        // exact evaluation, one candidate per lane and batch; results staged per candidate, `cap` at a time
        float dsum = 0.f, tia_max = 0.f;
        for (int c0 = 0; c0 < nc; c0 += cap) {
            const int c1 = min(nc, c0 + cap);
            if (c0 > 0) { lds_barrier<64>(); issue(nxt, c0 + tid, c0, c1); }
            for (int cb = c0; cb < c1; cb += 64) {
                const Cand cd = nxt;
                issue(nxt, cb + 64 + tid, c0, c1);
                if (!cd.ok) continue;
                const int gt = cd.gt, gs = cd.gs;
                const int kk = (int)(((float)gt + 0.5f) * p.inv_N);
                const EnvSlotLds& q = SL[kk];
                const int jp0 = cd.jp0, jp1 = jp0 + 1;
                const float ysrc = Lyf[gs];
                float py0 = cd.rest ? ysrc : cd.y0, py1 = cd.rest ? ysrc : cd.y1;
                unsigned a0 = cd.q0.x, b0_ = cd.q0.y, a1 = cd.q1.x, b1_ = cd.q1.y;
                float u0 = __uint_as_float(cd.q0.z), u1 = __uint_as_float(cd.q1.z);
                if (jp0 < 0) { py0 = ysrc; u0 = Lrue[gs]; a0 = Lra[gs]; b0_ = Lrb[gs]; }
                if (jp1 < 0) { py1 = ysrc; u1 = Lrue[gs]; a1 = Lra[gs]; b1_ = Lrb[gs]; }
                if (jp0 >= 0 && jp0 < q.n_valid) py0 = m0_advect(py0, a0, b0_, jp0, q.s_off_f, p.dpart_f, p.inv_D, p.dt);
                if (jp1 >= 0 && jp1 < q.n_valid) py1 = m0_advect(py1, a1, b1_, jp1, q.s_off_f, p.dpart_f, p.inv_D, p.dt);
                // interpolation, lateral cut-off, Gaussian deficit at the S rotor points (k_flow: eval_pair)
                const float wgt = cd.wgt;
                const float w0 = 1.0f - wgt, w1 = wgt;
                const float yc = w0 * py0 + w1 * py1;
                const float zc = p.hub;
                const float kv = w0 * rec_k(a0) + w1 * rec_k(a1);
                const float epv = w0 * rec_eps(b0_) + w1 * rec_eps(b1_);
                const float xd = (float)cd.dx * p.inv_D;
                const float sp = kv * xd + epv;
                const float sig = sp * p.D;
                const float yt = Lyf[gt];
                const float rc2 = (yt - yc) * (yt - yc) + (p.hub - zc) * (p.hub - zc);
                const float rcut = p.R_rot + 5.0f * sig;
                if (rc2 > rcut * rcut) continue;
                const float ctv = w0 * rec_ct(a0) + w1 * rec_ct(a1);
                const float uev = w0 * u0 + w1 * u1;
                const float inv2s2 = __builtin_amdgcn_rcpf(2.0f * sig * sig);
                const float cf = m0_cfrac(ctv, sp);
                // wake-added turbulence (Crespo-Hernandez), Gaussian-weighted at the hub
                const float ind = 0.5f * (1.0f - __builtin_amdgcn_sqrtf(1.0f - ctv));
                tiav[cd.pos] = p.no_ti_fold ? 0.f
                                            : p.tia * fast_pow(ind, p.tib) * q.ti_pow * fast_pow(fmaxf(xd, 1.0f), p.tid) * __expf(-rc2 * inv2s2);
                const float cgt = Lcg[gt], amp = uev * cf;
                float acc = 0.f;
#if WG_ENV_S_UNROLL
#pragma unroll 4
#endif
                for (int sI = 0; sI < p.S; ++sI) {
                    const float dy = yt + rdy[sI] * cgt - yc, dz = p.hub + rdz[sI] - zc;
                    acc += amp * __expf(-(dy * dy + dz * dz) * inv2s2);
                }
                def[cd.pos] = acc * p.inv_S;
            }
            lds_barrier<64>();
            // this target's slice of the round, in list order = ascending source order
            if (stepping) {
                const int b0 = max(cbeg, c0), b1 = min(cbeg + __popc(cmask), c1);
                for (int cc = b0; cc < b1; ++cc) {      // (x + 0.0f == x: rejected pairs do not change the sum)
                    dsum += def[cc - c0];
                    tia_max = fmaxf(tia_max, tiav[cc - c0]);
                }
            }
        }
        if (stepping) {
            tu = ws_f; tv = 0.f; tw = 0.f;
            tu -= dsum;
            tti = __builtin_amdgcn_sqrtf(ti_f * ti_f + tia_max * tia_max);
        }
        lds_barrier<64>();          // (the quad list below aliases the staging arrays)
        WG_STAMP(9);

        // (2) quad list of the chains that move (TurbLds::mvl, k_flow): a chain is listed whole while it may hold a particle
        // of a yawed turbine; a resting chain only receives this step's new particles, stored straight to their ring slots
        float* const spy = d.py + pb_env + (size_t)k * p.pstride;
        unsigned* const sra = d.rec_a + 2 * (pb_env + (size_t)k * p.pstride);      // interleaved (ct|k, eps|hv) record
        uint4* const sr4 = d.rec4 + pb_env + (size_t)k * p.pstride;
        int nlist;
        {
            int cnt = 0;
            const int nqd = rlen >> 2;
            bool full = false;
            if (stepping) {
                const bool moving = mvl != 0u && (int)(n_emitted - mvl) < rlen;
                full = moving || n_emit >= 4 || n_emit >= rlen;
                if (full) cnt = nqd;
                else {
                    const unsigned na_ = Lra[g], nb_ = Lrb[g];
#pragma unroll
                    for (int em = 0; em < 3; ++em) {
                        if (em < n_emit) {
                            int r = head + 1 + em; if (r >= rlen) r -= rlen;
                            const int ix = l_roff + r;
                            spy[ix] = yt_f;
                            reinterpret_cast<uint2*>(sra)[ix] = make_uint2(na_, nb_);
                            sr4[ix] = make_uint4(na_, nb_, __float_as_uint(Lrue[g]), 0u);
                        }
                    }
                }
            }
            int inc = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (tid >= o) inc += v; }
            nlist = __shfl(inc, 63, 64);
            if (full) {
                const int base = inc - cnt;
                const unsigned tag = (unsigned)g << 10;
                for (int i = 0; i < nqd; ++i) ql[base + i] = (unsigned short)(tag | (unsigned)i);
            }
        }
        lds_barrier<64>();
        WG_STAMP(10);

        // advection pass, software-pipelined: a lane requests its next listed quad before it computes the current one
        // (vmcnt counts loads and stores in ONE in-order queue: a plain load-compute-store loop waits for the previous
        // trip's stores whenever it waits for its loads)
        {
            struct QuadReq { float4 py; uint4 ra, rb; int g, kq; size_t q; };
            auto request = [&](QuadReq& r, const int cidx) __attribute__((always_inline)) {
                const bool v = cidx < nlist;
                const unsigned en = v ? ql[cidx] : 0u;
                r.g = (int)(en >> 10); r.kq = (int)(en & 1023u);
                const int kk = (int)(((float)r.g + 0.5f) * p.inv_N);
                r.q = ((pb_env + (size_t)kk * p.pstride + (size_t)Lroff[r.g]) >> 2) + (size_t)r.kq;
                if (v) {
                    r.py = reinterpret_cast<const float4*>(d.py)[r.q];
                    r.ra = reinterpret_cast<const uint4*>(d.rec_a)[2 * r.q];
                    r.rb = reinterpret_cast<const uint4*>(d.rec_a)[2 * r.q + 1];
                }
            };
            QuadReq nq;
            nq.py = make_float4(0.f, 0.f, 0.f, 0.f); nq.ra = nq.rb = make_uint4(0u, 0u, 0u, 0u);
            request(nq, tid);
            for (int cidx = tid; cidx < nlist; cidx += 64) {
                const QuadReq cur = nq;
                request(nq, cidx + 64);
                const int gq = cur.g, kq = cur.kq;
                const size_t q = cur.q;
                const int kk = (int)(((float)gq + 0.5f) * p.inv_N);
                const EnvSlotLds& sl = SL[kk];
                const int n_emit_q = sl.n_emit, n_valid_q = sl.n_valid;
                const float sof = sl.s_off_f;
                const int R = Lrlen[gq], hd = Lhead[gq];
                const int r0 = 4 * kq;
                int j0 = hd - r0; if (j0 < 0) j0 += R;             // age of ring slot r0 (slot r0+i: j0-i)
                int e0 = r0 - hd - 1; if (e0 < 0) e0 += R;         // emission index of slot r0 (r0+i: e0+i)
                const bool emits = (e0 < n_emit_q) || (n_emit_q > 0 && e0 + 3 >= R);   // wraps past R-1 -> 0
                float pyv[4] = {cur.py.x, cur.py.y, cur.py.z, cur.py.w};
                // (`ra` / `rb` hold the quad's interleaved records (a0 b0 a1 b1) (a2 b2 a3 b3))
                unsigned rav[4] = {cur.ra.x, cur.ra.z, cur.rb.x, cur.rb.z};
                unsigned rbv[4] = {cur.ra.y, cur.ra.w, cur.rb.y, cur.rb.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int j = j0 - i; if (j < 0) j += R;
                    if (j < n_valid_q) pyv[i] = m0_advect(pyv[i], rav[i], rbv[i], j, sof, p.dpart_f, p.inv_D, p.dt);
                }
                const float y0 = Lyf[gq];
                if (emits) {
                    const unsigned na_ = Lra[gq], nb_ = Lrb[gq];
                    const unsigned ue_ = __float_as_uint(Lrue[gq]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int ei = e0 + i; if (ei >= R) ei -= R;
                        if (ei < n_emit_q) {
                            pyv[i] = y0; rav[i] = na_; rbv[i] = nb_;
                            d.rec4[4 * q + i] = make_uint4(na_, nb_, ue_, 0u);
                        }
                    }
                    reinterpret_cast<uint4*>(d.rec_a)[2 * q] = make_uint4(rav[0], rbv[0], rav[1], rbv[1]);
                    reinterpret_cast<uint4*>(d.rec_a)[2 * q + 1] = make_uint4(rav[2], rbv[2], rav[3], rbv[3]);
                }
                reinterpret_cast<float4*>(d.py)[q] = make_float4(pyv[0], pyv[1], pyv[2], pyv[3]);
                // (excursion bound over the VALID particles of the quad only, see k_flow)
                float ex = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int j = j0 - i; if (j < 0) j += R;
                    if (j < n_valid_q) ex = fmaxf(ex, fabsf(pyv[i] - y0));
                }
                if (ex > Lbd[gq]) atomicMax(reinterpret_cast<int*>(&Lbd[gq]), __float_as_int(ex));   // ex >= 0: int order == float order
            }
        }
        WG_STAMP(3);

        // the slot's clock advances
        if (stepping) {
            s_head = new_head; n_valid = new_valid; s_off = s_new; s_time += p.dt_d; istep += 1u;
            n_emitted += (unsigned)n_emit;
            // the step's emissions are in the ring now
            head += n_emit;
            if (n_emit >= rlen) head %= rlen; else if (head >= rlen) head -= rlen;
            Lhead[g] = head;
            part_acc += min(n_valid, rlen);          // roofline accounting: particles that can still reach a rotor
            ++n_flow;
            --budget;
            if (!role_live) dev_stepped = true;
        }
        WG_STAMP(6);

        // per-turbine tail: power / thrust with the current yaw (model M0 step 5), WindFarmEnv._take_measurements
        // (Wind_Farm_Env.py:480-495) accumulated over the K sub-steps and, at the end of the env step,
        // farm_mes.add_measurements' ring push (MesClass.py:568-591)
        const bool measuring = stepping && !is_dev;
        const bool unit_end = measuring && (sub + 1 == p.K);
        const KArgsPtr kc = wg_cold_args();
        if (stepping) {
            const float wsn = fmaxf(tu * cg + tv * sg, 0.0f);
            tpow = tab_lookup(tabp, p, wsn);
            tct = tab_lookup(tabct, p, wsn) * cg * cg;
            if (measuring && farm == 0) {
                const float wsm = __builtin_amdgcn_sqrtf(tu * tu + tv * tv + tw * tw);
                const float wdm = 0.f + wd_env;      // (steady inflow: atan(v / u) = +-0, see k_flow)
                float val[WG_N_CH] = {sws + wsm, swd + wdm, syaw + yaw, sp_ + tpow};
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
                    float* __restrict__ rbase = kc->d.ring + (size_t)ctx_id * kc->p.ring_stride;
#pragma unroll
                    for (int ch = 0; ch < WG_N_CH; ++ch) {
                        const int H = kc->p.hlen[ch];
                        rbase[kc->p.ring_off[ch] + fast_mod(n_pushed, H, kc->p.inv_hlen[ch]) * N + t] = val[ch];      // (time-major: WgRing)
                    }
                    // stage the pushed values for the farm-level mean / mean / sum
                    sws = val[0]; swd = val[1]; sp_ = val[3];
                } else {
                    sws = val[0]; swd = val[1]; syaw = val[2]; sp_ = val[3];
                }
            }
        }
        // farm-level values, slot by slot (wave-uniform loop over the stepping slots that measure)
        const unsigned long long mmask = __ballot(measuring);
        for (int kk = 0; kk < NS; ++kk) {
            if (!((mmask >> (kk * N)) & 1ull)) continue;
            const bool fk = F == 2 ? (kk & 1) : 0;
            const bool mine = valid && k == kk;
            const bool k_unit_end = __shfl((int)unit_end, kk * N, 64) != 0;
            const bool k_live = (F == 2 ? (kk >> 1) : kk) == env_live && mode == WG_MODE_STEP;
            if (fk) {
                // fs_baseline...power().sum() (:954)
                const float s = env_slot_sum(mine ? tpow : 0.f, kk, N, tid);
                if (mine) base_acc += s;
                if (k_unit_end && mine && t == 0) {
                    const float bp = p.K == 1 ? base_acc : base_acc * inv_k;
                    if (k_live) kc->d.step_base_pow[e] = bp;
                    else kc->d.pend_base[(size_t)ctx_id * kc->p.power_avg + umod_small(pend_base_n, kc->p.power_avg, kc->p.pavg_magic)] = bp;
                }
            } else if (k_unit_end) {
                const float a_ws = env_slot_sum(mine ? sws : 0.f, kk, N, tid), a_wd = env_slot_sum(mine ? swd : 0.f, kk, N, tid);
                const float tot = env_slot_sum(mine ? sp_ : 0.f, kk, N, tid);
                if (mine && t == 0) {
                    const FlowP __attribute__((address_space(4)))& pc = kc->p;
                    float* fbase = kc->d.fring + (size_t)ctx_id * pc.fring_stride;
                    fbase[pc.fring_off[WG_CH_WS] + umod_small(n_pushed, pc.hlen[WG_CH_WS], pc.hmagic[WG_CH_WS])] = a_ws * pc.inv_N;
                    fbase[pc.fring_off[WG_CH_WD] + umod_small(n_pushed, pc.hlen[WG_CH_WD], pc.hmagic[WG_CH_WD])] = a_wd * pc.inv_N;
                    fbase[pc.fring_off[WG_CH_POWER] + umod_small(n_pushed, pc.hlen[WG_CH_POWER], pc.hmagic[WG_CH_POWER])] = tot;
                    if (k_live) kc->d.step_farm_pow[e] = tot;
                    else kc->d.pend_farm[(size_t)ctx_id * pc.power_avg + umod_small(pend_farm_n, pc.power_avg, pc.pavg_magic)] = tot;
                }
            }
        }
        // the slot's schedule
        if (stepping) {
            if (is_dev) --dev_rem;
            else if (++sub >= p.K) {
                sub = 0;
                if (farm == 0) { sws = 0.f; swd = 0.f; syaw = 0.f; sp_ = 0.f; }
                else base_acc = 0.f;
                if (!role_live) --fill_rem;
            }
        }
        // (context-level counters, kept by every lane of the context: both farms' pushes advance them)
        {
            const int ca = c * F * N;                       // the context's agent-farm lane 0
            const bool ctx_pushed = __shfl((int)unit_end, ca, 64) != 0;
            const bool ctx_live = c == env_live && mode == WG_MODE_STEP;
            if (ctx_pushed) { ++n_pushed; if (!ctx_live) ++pend_farm_n; }
            if (F == 2) {
                const bool base_pushed = __shfl((int)unit_end, ca + N, 64) != 0;
                if (base_pushed && !ctx_live) ++pend_base_n;
            }
        }
    }

    WG_STAMP(7);
    // ---- epilogue: the env's slots back to memory -----------------------------------------------------------------------
    const KArgsPtr ke = wg_cold_args();
    if (valid) {
        ke->d.yaw[tb] = yaw; ke->d.u[tb] = tu; ke->d.v[tb] = tv; ke->d.w[tb] = tw;
        ke->d.ti_loc[tb] = tti; ke->d.power[tb] = tpow; ke->d.ct[tb] = tct;
        reinterpret_cast<float4*>(ke->d.bnd)[tb] = make_float4(Lbd[g], bk, be, __uint_as_float(mvl));
        if (role_live && farm == 0) ke->d.old_yaw[(size_t)e * N + t] = oyaw;
    }
    for (int kk = 0; kk < NS; ++kk) {            // roofline accounting, per slot
        const int s = wg_wave_sum_i((valid && k == kk) ? part_acc : 0);
        if (k == kk) part_acc = s;
    }
    if (valid && t == 0) {
        WgSlot& slot = ke->d.slot[slot_id];
        slot.part_count = part0 + (unsigned)part_acc;
        slot.head = s_head; slot.n_valid = n_valid; slot.s_off = s_off; slot.time = s_time;
        slot.istep = istep; slot.n_emitted = n_emitted;
        slot.dev_remaining = dev_rem; slot.fill_remaining = fill_rem;
        slot.flow_count = flow0 + (unsigned)n_flow;
        WgCtx& cx = ke->d.ctx[ctx_id];
        if (farm == 0) { cx.n_pushed = n_pushed; cx.pend_farm_n = pend_farm_n; }
        if (farm == F - 1) cx.pend_base_n = pend_base_n;
    }
#ifdef WG_TIMELINE
    if (tid == 0 && d.dbg) {
        wg_stamps[8] = clock64();
        for (int kq = 0; kq < 16; ++kq) d.dbg[(size_t)blockIdx.x * 16 + kq] = wg_stamps[kq];
    }
#endif
    // a background episode whose agent farm completed its development in this launch: its window sums and first
    // observation are prepared for the swap (wg_first_obs, see k_flow)
    if (mode == WG_MODE_STEP) {
        const int bl = (env_live ^ 1) * F * N;            // the background context's agent-farm lane 0
        const bool done_now = dev_stepped && dev_rem == 0 && fill_rem == 0;
        if (__shfl((int)done_now, bl, 64)) {
            full_barrier<64>();                            // the ring pushes have left the wave
            wg_first_obs(ke->d.gp, ke->d.gd, e * 2 + (env_live ^ 1), __shfl(n_pushed, bl, 64), tid);
        }
    }
}

extern "C" void wg_launch_flow_env(const FlowP* p, const FlowPtrs* d, int mode, const float* actions, const uint8_t* mask,
                                   int chunk, hipStream_t st) {
    const int grid = p->B;
    const size_t lds = p->env_lds;
    if (p->noise) hipLaunchKernelGGL((k_flow_env<true>), dim3(grid), dim3(64), lds, st, *p, *d, mode, actions, mask, chunk);
    else hipLaunchKernelGGL((k_flow_env<false>), dim3(grid), dim3(64), lds, st, *p, *d, mode, actions, mask, chunk);
}
