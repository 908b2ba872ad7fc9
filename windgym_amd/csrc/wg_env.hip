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

#ifndef WG_ENV_ABLATE
#define WG_ENV_ABLATE 0     // profiling builds only: 1 = no bracket gathers, 2 = no advection pass, 4 = trivial advection arithmetic, 8 = no rotor-point loop
#endif
#include "wg_env_common.h"

#ifndef WG_ENV_S_UNROLL
#define WG_ENV_S_UNROLL 0   // 1: the rotor-point loop of the pair evaluation unrolled by 4
#endif
#ifndef WG_ENV_PP
#define WG_ENV_PP 0         // bit 0: two alternating request buffers in the advection pass, bit 1: in the evaluation (0: one buffer + a
                            // register copy per trip).  Measured, cfg2 x 4096 / cfg4 x 2048, same box: 0: 67.8 / 53.5, 1: 66.0, 2: 68.0, 3: 63.0 / 50.9 M
                            // env-steps/s — the second copy of the inlined phase costs more than the moves it saves
#endif
// Per-slot record in LDS.  Everything here is uniform over the N lanes of a slot: kept in LDS (written by the slot's lane
// t = 0, read by broadcast) instead of one VGPR per word in every lane — the kernel lives at 128 VGPRs (4 waves per SIMD:
// 4096 envs = one dispatch round) and would need ~180 with the slot state in registers.
struct __attribute__((aligned(16))) EnvSlotLds {
    // schedule (one 16-byte read decides whether the slot takes the next flow step)
    int dev_rem, fill_rem, sub, budget;
    // clock of the current flow step
    double s_new;
    int n_emit, n_valid, new_valid;
    unsigned n_emitted;
    float s_off_f, near_f, move_max, ti_pow;
    // the slot's persistent clock and its context's wind
    double s_off, ws;
    float ws_f, ti_f, wd_env, base_acc;
    // counters this slot owns (agent farm: n_pushed, pend_farm_n; last farm: pend_base_n) and accounting
    int n_pushed, pend_farm_n, pend_base_n, n_flow;
    // cold words, parked here between prologue and epilogue
    double c_time;
    int c_head;
    unsigned c_part, c_flow, c_istep, c_tag, n_emitted0;
    float out_pw;        // the env step's farm power: agent farm total / baseline farm (what k_glue reads as step_farm_pow / step_base_pow)
    unsigned c_add;      // WgSlot::add_count, parked like the other cold words
    int bg_init;         // (slot 0 of a background context's wave, WPE 2) its set-up flag was pending at the head of this launch
    int pad_[3];
};
static_assert(sizeof(EnvSlotLds) == WG_ENV_SLOT_LDS_BYTES, "keep WG_ENV_SLOT_LDS_BYTES in sync (wg_flow.h)");
// fixed LDS layout (compile-time offsets from the dynamic LDS base: no address registers): cross-lane turbine fields, indexed
// by lane g, then the slot records, then the staging region (per-candidate deficit | added TI | candidate list, aliased by the
// quad list), then the tables (FlowP::env_off_tab)
//   Lsrc4 (float x, float y, bk, be)   positions as floats + running bounds of the chain (candidate pass)
//   Lrec4 (ra, rb, -, cos yaw)         this step's emission record, packed (pack_a / pack_b: u_e is in rb), cos(yaw)
//   Lring (roff, rlen, head, -)        the chain's compact ring
//   Lxr                                downwind position in double (brackets are decided in double)
//   Lsrc2 (bd, mvl)                    excursion bound (raised by the advection pass with LDS atomics), last moving emission
#define WG_ENV_OFF_SRC4 0
#define WG_ENV_OFF_REC4 1024
#define WG_ENV_OFF_RING 2048
#define WG_ENV_OFF_XR 3072
#define WG_ENV_OFF_SRC2 3584
#define WG_ENV_OFF_SL 4096
#define WG_ENV_OFF_HDR (WG_ENV_OFF_SL + 4 * WG_ENV_SLOT_LDS_BYTES)      // int[32]  the env header as the prologue loaded it (the glue tail's copy)
#define WG_ENV_OFF_STAGE (WG_ENV_OFF_HDR + 128)
static_assert(WG_ENV_OFF_STAGE == WG_ENV_FIXED_LDS_BYTES, "keep WG_ENV_FIXED_LDS_BYTES in sync (wg_flow.h)");

// Sum of a per-lane value over the lanes of the lane's OWN slot, for all slots of the env at once, in the association
// k_flow's one-farm wave uses (values in lanes 0 .. N-1 of row 0 / rows 0-1, zeros elsewhere: wg_row_sum / wg_wave_sum).
// N = 16: the slots ARE the DPP rows.  Otherwise slot r is first gathered into row r (N < 16) or half r (N <= 32, F = 1).
// Valid in every lane of a slot.
__device__ __forceinline__ float env_slot_sums(const float v, const int N, const int NS, const int k, const int tid) {
    if (N == 16) return wg_row_sum(v);
    if (N < 16) {
        const int r = tid >> 4, i = tid & 15;
        const float x = __shfl(v, (r * N + i) & 63, 64);
        const float s = wg_row_sum((i < N && r < NS) ? x : 0.f);
        return __shfl(s, k << 4, 64);
    }
    const int h = tid >> 5, i = tid & 31;
    const float x = __shfl(v, (h * N + i) & 63, 64);
    float s = (i < N && h < NS) ? x : 0.f;
    WG_ROW_REDUCE(s, wg_addf)
    s += __shfl_xor(s, 16, 64);
    return __shfl(s, k << 5, 64);
}

// WPE = waves per env.  1: one wave serves all 2 F slots of the env (lane = slot * N + turbine).  2: a workgroup of two waves
// per env, wave c serves the F slots of context c (lane = farm * N + turbine) in its own LDS region — the running episode's
// step and the background episode's development then run side by side instead of one after the other, and the rare long
// paths of the background context (episode set-up, a second flow step, the first observation) are off the live wave's chain;
// the waves meet at ONE workgroup barrier before the live context's wave runs the glue (k_flow_env).  `smem` is the wave's
// own region, `wv` its context.
// SPLIT (small batches, WPE 2; 1: three waves per env, 2: four — a pass wave for the background context too, whose step has no glue
// behind it but the longer pass: both its farms' chains move): the running episode's context has a SECOND wave (role 1, the "pass wave")
// that runs the quad list and the advection pass of its step — a fifth of the main wave's instructions, and at these sizes a
// wave's life is its instruction chain (one dependent issue per ~7 cycles, EXPERIMENTS.md); the background context's wave has
// no glue behind its step and is not the env's last wave.  The pass wave repeats the prologue and the emission records from the
// same inputs (deterministic: the same records), lists the moving chains, requests its first trip and then waits for an LDS flag
// the main wave raises when the evaluation's gathers — which read the particles in their PRE-step state — have landed; only then
// does it store.  It writes the particles and the chains' excursion bounds (WgBnd.x), the main wave everything else; neither
// reads what the other writes in the same launch.  `smem_pass`: the pass wave's LDS region, whose header slot holds the flag.
// `zone`: LDS of the glue inputs the pass wave fetches ahead (LEAN_PRE_*, wg_glue_lean.h).
template <bool NOISE, int WPE, int GLUE, int SPLIT>
__device__ __forceinline__ void env_flow(char* const smem, char* const smem_pass, float* const zone, const int wv, const int role, const int mode,
                                         const float* __restrict__ actions, const uint8_t* __restrict__ mask, const int chunk, EnvFlowOut& out) {
    const int tid = threadIdx.x & 63, e = blockIdx.x;
    int N, F, NS, NL;
    float inv_N;
    {
        const KArgsPtr ka = wg_cold_args();
        N = ka->p.N; F = ka->p.F; inv_N = ka->p.inv_N;
        NS = WPE == 2 ? F : 2 * F; NL = NS * N;              // slots / lanes served by THIS wave
    }
    const int kbase = WPE == 2 ? wv * F : 0;                // the wave's first slot of the env
    const bool valid = tid < NL;
    const int g = valid ? tid : 0;
    const int k = (int)(((float)g + 0.5f) * inv_N);         // slot of the wave (WPE 1: of the env = ctx * F + farm)
    const int t = g - k * N;
    const int c = WPE == 2 ? wv : (F == 2 ? (k >> 1) : k), farm = F == 2 ? (k & 1) : 0;

    float4* const Lsrc4 = reinterpret_cast<float4*>(smem + WG_ENV_OFF_SRC4);
    uint4* const Lrec4 = reinterpret_cast<uint4*>(smem + WG_ENV_OFF_REC4);
    int4* const Lring = reinterpret_cast<int4*>(smem + WG_ENV_OFF_RING);
    double* const Lxr = reinterpret_cast<double*>(smem + WG_ENV_OFF_XR);
    float2* const Lsrc2 = reinterpret_cast<float2*>(smem + WG_ENV_OFF_SRC2);
    EnvSlotLds* const SL = reinterpret_cast<EnvSlotLds*>(smem + WG_ENV_OFF_SL);
    float* const def = reinterpret_cast<float*>(smem + WG_ENV_OFF_STAGE);
    float* const tiav = def + WG_ENV_CAP;
    unsigned short* const cl = reinterpret_cast<unsigned short*>(tiav + WG_ENV_CAP);
    unsigned short* const ql = reinterpret_cast<unsigned short*>(smem + WG_ENV_OFF_STAGE);
    EnvSlotLds& my = SL[k];

    WG_STAMP(0);
    // ---- prologue: every independent global load up front (one exposed round trip) ----------------------------------
    typedef const __attribute__((address_space(4))) WgEnv* CEnvPtr;
    int env_live;
    bool role_live = false, role_dev = false, defer_init = false, split_on = false;
    float yaw, tu, tti, oyaw;
    {
        const KArgsPtr k0 = wg_cold_args();
        // The env's 128-byte header as ONE coalesced load from the GLOBAL address space (lane i: word i), its fields broadcast
        // with v_readlane below, where the roles are decided — NOT scalar loads through the constant address space: the glue
        // tail of this very launch rewrites the header (such loads may be re-executed after the store, and
        // the scalar cache does not see this launch's vector stores).
        static_assert(sizeof(WgEnv) == 128, "the env header is loaded as 32 words");
        const int hw = reinterpret_cast<const int*>(k0->d.env + e)[tid & 31];
        if (GLUE != 0 && tid < 32) reinterpret_cast<int*>(smem + WG_ENV_OFF_HDR)[tid] = hw;      // (handed to the glue tail: no second load)
        out.bg_init_pending = 0; out.rounds = 0; out.first_obs = 0;
        const bool use_mask = mode == WG_MODE_RESET && mask != nullptr;
        const uint8_t mask_byte = *(use_mask ? mask + e : reinterpret_cast<const uint8_t*>(k0->d.env + e));
        const bool masked_out = use_mask && mask_byte == 0;
        const unsigned ctx_id = (unsigned)(e * 2 + c), slot_id = (unsigned)(e * 2 * F + kbase + k);
        const unsigned tb = (unsigned)((e * 2 * F + kbase) * N + g);      // == slot_id * N + t
        const unsigned tcx = ctx_id * (unsigned)N + (unsigned)t;

        int dev_rem, fill_rem, n_pushed, pend_farm_n, pend_base_n, init_pending, time_max_c, n_valid, c_head;
        unsigned n_emitted, c_part, c_flow, c_istep, c_tag, c_add;
        double s_off, ws, l_xr, l_yr, c_time;
        float ti_f, wd_env, l_yaw, l_u, l_ti, l_act;
        float4 l_bnd;
        int l_roff, l_rnext;
        auto load_state = [&]() __attribute__((always_inline)) {
            const KArgsPtr kl = wg_cold_args();
            const WgSlot& slot = kl->d.slot[slot_id];
            const WgCtx& cx = kl->d.ctx[ctx_id];
            dev_rem = slot.dev_remaining; fill_rem = slot.fill_remaining;
            s_off = slot.s_off; c_time = slot.time; c_head = slot.head; n_valid = slot.n_valid; c_istep = slot.istep;
            n_emitted = slot.n_emitted; c_part = slot.part_count; c_flow = slot.flow_count; c_add = slot.add_count;
            ws = cx.ws; ti_f = (float)cx.ti; wd_env = (float)cx.wd;
            n_pushed = cx.n_pushed; pend_farm_n = cx.pend_farm_n; pend_base_n = cx.pend_base_n;
            c_tag = (unsigned)cx.episode_tag; init_pending = cx.init_pending; time_max_c = cx.time_max;
            const int* ro = kl->d.roff + ctx_id * (unsigned)(N + 1);
            l_roff = ro[(unsigned)t]; l_rnext = ro[(unsigned)t + 1u];
            l_xr = kl->d.xr[tcx]; l_yr = kl->d.yr[tcx];
            l_yaw = kl->d.yaw[tb]; l_u = kl->d.u[tb]; l_ti = kl->d.ti_loc[tb];
            l_bnd = reinterpret_cast<const float4*>(kl->d.bnd)[tb];
        };
        load_state();
        const int n_tab = k0->p.n_tab, S = k0->p.S;
        const unsigned i_tab = (unsigned)min(tid, n_tab - 1), i_s = (unsigned)min(tid, S - 1);
        const float pf_tp = k0->d.tab_power[i_tab], pf_tc = k0->d.tab_ct[i_tab];
        const float pf_dy = k0->d.rotor_dy[i_s], pf_dz = k0->d.rotor_dz[i_s];
        {   // (every agent-farm lane reads its turbine's action; only the running episode's lanes use it)
            const bool has_act = mode == WG_MODE_STEP && farm == 0;
            const float a = *(has_act ? actions + (unsigned)(e * N + t) : k0->d.tab_ct);
            l_act = has_act ? a : 0.f;
        }

#define WG_HDR_I(f) __builtin_amdgcn_readlane(hw, (int)(offsetof(WgEnv, f) / 4))
        env_live = WG_HDR_I(live);
        int env_done = WG_HDR_I(done), env_shadow_iters = WG_HDR_I(shadow_iters), env_steps_done = WG_HDR_I(steps_done);
        int env_timestep = WG_HDR_I(timestep), env_time_max_live = WG_HDR_I(time_max_live);
#undef WG_HDR_I
        if (SPLIT && role == 1) {
            // (the pass wave: its flag down, and every word it has requested in its registers, before the barrier lets the main
            // waves go on — they rewrite some of those words at their ends)
            if (tid == 0) { *reinterpret_cast<int*>(smem + WG_ENV_OFF_HDR) = 0; if (GLUE != 0 && (SPLIT == 1 || wv == 0)) reinterpret_cast<int*>(zone)[LEAN_PRE_FLAG] = 0; }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        if (WPE == 2) {
            // Two waves per env read the same header, and the live context's wave rewrites it in its glue tail.  Both waves hold
            // their copies BEFORE either passes this barrier, so neither can see a word the other has already advanced, whatever
            // their relative timing (the decisions both waves derive from the header — live, truncates, the
            // background plan — must agree).  Every prologue load has been requested by now: the barrier adds no round trip.
            asm volatile("" : "+s"(env_live), "+s"(env_done), "+s"(env_shadow_iters), "+s"(env_steps_done), "+s"(env_timestep), "+s"(env_time_max_live) : : "memory");
            asm volatile("s_barrier" ::: "memory");
        }
        if (SPLIT && role == 1 && GLUE != 0 && c == env_live) {
            // The glue's step-independent inputs, fetched now — the main wave's glue then has no load on its common path: the old
            // window sums and leaving samples of the glue lane's turbine, the deque entries this step's powers replace, the
            // metrics.  Nothing in this launch writes them before the glue does (the rings' leaving slots are not the slot the
            // step pushes into: a ring holds one sample more than its deque).
            const EnvKArgsPtr kz = (EnvKArgsPtr)wg_cold_args();
            const WgParams& zp = *(const WgParams*)&kz->gp;
            const WgPtrs& zd = *(const WgPtrs*)&kz->gd;
#define WG_HDR_Z(f) __builtin_amdgcn_readlane(hw, (int)(offsetof(WgEnv, f) / 4))
            const int z_np = WG_HDR_Z(n_pushed_live), z_fn = WG_HDR_Z(farm_pow_n), z_bn = WG_HDR_Z(base_pow_n);
#undef WG_HDR_Z
            const int PA = zp.power_avg;
            const SumsRaw zr = wg_sums_load<false, false>(zp, zd, e, e * 2 + env_live, tid < N ? tid : 0, z_np);
            const float z_fo = zd.farm_pow[(size_t)e * PA + z_fn % PA];
            const float z_bo = F == 2 ? zd.base_pow[(size_t)e * PA + z_bn % PA] : 0.f;
            const float z_met = tid < WG_N_METRICS ? zd.metrics[(size_t)e * WG_N_METRICS + tid] : 0.f;
            double* const zs = reinterpret_cast<double*>(zone);
#pragma unroll
            for (int sI = 0; sI < WG_N_CH; ++sI) { zs[LEAN_PRE_S(sI, tid)] = zr.S[sI]; zone[LEAN_PRE_LV(sI, tid)] = zr.lv[sI]; }
            zone[LEAN_PRE_MET(tid)] = z_met;
            if (tid == 0) { zone[LEAN_PRE_FOLD] = z_fo; zone[LEAN_PRE_BOLD] = z_bo; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (tid == 0) env_flag_set(reinterpret_cast<int*>(zone) + LEAN_PRE_FLAG);
        }
        out.env_live = env_live;
        out.truncates = env_timestep >= env_time_max_live;
        out.steps_done = env_steps_done; out.time_max_live = env_time_max_live;

        // ---- roles ----------------------------------------------------------------------------------------------------
        // role_live: this lane's slot belongs to the running episode and takes one env step (K flow sub-steps with
        //            measurement);
        // role_dev:  its slot develops a not-yet-live episode for `budget` flow steps — the background context in STEP mode
        //            (Wind_Farm_Env.py:722-796 hidden behind the running episode), the masked envs' live
        //            context in RESET mode
        const bool is_live_c = (c == env_live);
        const int autoreset = k0->p.autoreset;
        int budget = 0;
        if (mode == WG_MODE_STEP) {
            role_live = is_live_c && !env_done;
            role_dev = !is_live_c && autoreset != 0;
            // (wave-uniform: both contexts' lanes see the background context's flag through their own loads; WPE 2: the
            // background context's wave sees its own, the live wave gets it through LDS after the barrier)
            const int bg_pending = WPE == 2 ? (is_live_c ? 0 : __shfl(init_pending, 0, 64)) : __shfl(init_pending, (env_live ^ 1) * F * N, 64);
            out.bg_init_pending = autoreset && bg_pending;
            if (SPLIT == 2 && role == 1 && out.bg_init_pending) return;      // (an episode set-up in this launch: the context runs unsplit)
            if (autoreset && bg_pending && WPE == 1 && (WG_ENV_DEFER_INIT != 0) && !out.truncates) {
                // rare path (one context per truncation): the retired context's next episode is set up AFTER this wave's step
                // (below) — its background lanes rest in this launch, the set-up needs no reload of the wave's state, and the
                // wave is no longer the launch's straggler (at the head of the launch it lived 55 us against 40 for the rest).
                // NOT when this very step truncates (one-step episodes: time_max <= 0): the glue tail then swaps the context in
                // in this launch and needs it developed — the head-of-launch path below gives it the whole budget
                defer_init = true;
                budget = 0;
            } else if (autoreset && bg_pending) {
                // rare path (one context per truncation): set the retired context's next episode up, both farms (see k_flow)
                const KArgsPtr ki = wg_cold_args();
                const WgParams& gp = *ki->d.gp;
                env_init_episode<GLUE != 0>(e, env_live ^ 1, tid);
                full_barrier<64>();
                load_state();
                const int fill_max = F == 2 ? max(gp.fill_a, gp.fill_b) : gp.fill_a;
                const int inc = 1 + (gp.extra_inc ? 1 : 0);
                const int tm = WPE == 2 ? ki->d.ctx[e * 2 + env_live].time_max : __shfl(time_max_c, env_live * F * N, 64);
                const long total = (long)((tm + inc - 1) / inc) + 1;
                const int dev0 = __shfl(dev_rem, WPE == 2 ? 0 : (env_live ^ 1) * F * N, 64);
                budget = wg_shadow_share(dev0 + gp.K * fill_max, total - env_steps_done, env_steps_done, e);
                if (WG_ENV_SPLIT_PLAN != 0)
                    budget = wg_shadow_share(dev_rem + gp.K * fill_rem, total - env_steps_done, env_steps_done, e, farm ? 0x80000000u : 0u);
            } else if (WG_ENV_SPLIT_PLAN != 0) {
                // The background episode's share of this step, planned per FARM: the flow steps its farm still needs over the env
                // steps left (wg_shadow_share: dithered, so the expected share is exact), the baseline farm's dither half a period
                // after the agent farm's — below one flow step per two env steps the two never step in the same launch, and a wave
                // carries 2 F + 1 farm steps where the per-context plan (WgEnv::shadow_iters, ignored here) gave it 2 F or 3 F:
                // the launch lasts as long as its heaviest wave.
                const int inc = k0->p.env_inc;
                const long total = (long)((env_time_max_live + inc - 1) / inc) + 1;
                budget = role_dev ? wg_shadow_share(dev_rem + k0->p.K * fill_rem, total - env_steps_done, env_steps_done, e, farm ? 0x80000000u : 0u) : 0;
            } else {
                budget = env_shadow_iters;
            }
        } else {
            role_dev = is_live_c && !masked_out;
            budget = chunk;
        }
        if (SPLIT) {
            // exactly one flow step for every slot of the wave that steps at all, no episode set-up in this launch
            const bool two = valid && role_dev && budget >= 2;
            split_on = mode == WG_MODE_STEP && k0->p.K == 1 && (is_live_c || (SPLIT == 2 && !out.bg_init_pending)) && !defer_init && !__ballot(two);
            if (role == 1 && !split_on) return;
        }
        if (WPE == 2 && valid && t == 0) {      // (what the other wave's glue reads of this one, also if it has nothing to do)
            my.dev_rem = dev_rem; my.fill_rem = fill_rem; my.bg_init = out.bg_init_pending; my.n_flow = 0; my.out_pw = 0.f;
        }
        {   // nothing to do for the whole env (masked out in RESET mode, finished env without autoreset, idle background)
            const bool any_work = valid && (role_live || (role_dev && budget > 0 && (dev_rem > 0 || fill_rem > 0)));
            if (!__ballot(any_work)) {
                // (two waves per env: the resting background wave of an episode completed in an EARLIER launch prepares its
                // first observation now — see the end of this function)
                if (WPE == 2 && (WG_ENV_FIRST_OBS_LATER != 0) && mode == WG_MODE_STEP && !is_live_c && autoreset && role == 0) {
                    const int d0 = __shfl(dev_rem, 0, 64), f0 = __shfl(fill_rem, 0, 64), np0 = __shfl(n_pushed, 0, 64);
                    const KArgsPtr kf = wg_cold_args();
                    if (d0 == 0 && f0 == 0 && kf->d.gd->next_obs_ok != nullptr && kf->d.gd->next_obs_ok[ctx_id] == 0) {
                        out.first_obs = 1;
                        env_first_obs<GLUE != 0>((int)ctx_id, np0, tid);
                    }
                }
                return;
            }
        }
#if WG_ENV_PRIO
        static_assert(WPE == 1, "WG_ENV_PRIO: one wave per env only");
        // The launch lasts as long as its slowest wave, and with ONE wave per env and one dispatch round the slowest waves are
        // the handful per launch on a rare path — episode set-up at the head of the launch, a background context that takes
        // two or more flow steps, the first observation of a completed episode, the swap at truncation (cfg2 x 4096: their
        // waves lived 51-58 us against 40 for the rest and WERE the kernel's last 7 us; cfg4: 28-33 against 21).  They are
        // known here, before the work starts: such a wave raises its issue priority over the three it shares its SIMD with.
        if (mode == WG_MODE_STEP) {
            const int bl = (env_live ^ 1) * F * N;              // the background context's agent-farm lane 0
            const int b_dev = __shfl(dev_rem, bl, 64), b_fill = __shfl(fill_rem, bl, 64);
            const bool bg_active = role_dev_any(autoreset, b_dev, b_fill);
            const bool multi_round = bg_active && budget >= 2 && b_dev + b_fill >= 2;
            const bool completes = bg_active && b_dev + k0->p.K * b_fill <= budget;
            const bool truncates = out.truncates != 0;
            if (out.bg_init_pending || multi_round || completes || truncates) __builtin_amdgcn_s_setprio(3);
        }
#endif

        // ---- state into LDS ---------------------------------------------------------------------------------------------
        const KArgsPtr k1 = wg_cold_args();
        const int rlen = l_rnext - l_roff;
        const int head = n_emitted == 0u ? rlen - 1 : fast_mod((int)(n_emitted - 1u), rlen, __builtin_amdgcn_rcpf((float)rlen));
        const float ws_f = (float)ws;
        if (valid) {
            Lxr[g] = l_xr;
            Lsrc4[g] = make_float4((float)l_xr, (float)l_yr, l_bnd.y, l_bnd.z);
            Lsrc2[g] = make_float2(l_bnd.x, l_bnd.w);
            Lring[g] = make_int4(l_roff, rlen, head, k);
            Lrec4[g] = make_uint4(0u, 0u, 0u, __float_as_uint(1.f));
            if (t == 0) {
                my.dev_rem = dev_rem; my.fill_rem = fill_rem; my.sub = 0; my.budget = budget;
                my.n_valid = n_valid; my.n_emitted = n_emitted; my.n_emitted0 = n_emitted;
                my.move_max = fabsf(k1->p.hill) * ws_f * k1->p.dt; my.ti_pow = fast_pow(ti_f, k1->p.tic);
                my.s_off = s_off; my.ws = ws; my.ws_f = ws_f; my.ti_f = ti_f; my.wd_env = wd_env; my.base_acc = 0.f;
                my.n_pushed = n_pushed; my.pend_farm_n = pend_farm_n; my.pend_base_n = pend_base_n; my.n_flow = 0;
                my.bg_init = out.bg_init_pending;
                my.c_time = c_time; my.c_head = c_head; my.c_part = c_part; my.c_flow = c_flow; my.c_istep = c_istep; my.c_tag = c_tag; my.c_add = c_add;
            }
        }
        // tables: power | ct | rotor points as (lateral offset, squared vertical offset from the wake centre height)
        float* const tabp = reinterpret_cast<float*>(smem + k1->p.env_off_tab);
        float* const tabct = tabp + n_tab;
        float2* const rpt = reinterpret_cast<float2*>(tabct + n_tab);
        const float hub = k1->p.hub;
        auto rp = [&](const float dy_, const float dz_) __attribute__((always_inline)) {
            const float dz = hub + dz_ - hub;              // (k_flow: p.hub + rdz - zc, zc = hub under steady inflow)
            return make_float2(dy_, dz * dz);
        };
        if (tid < n_tab) { tabp[tid] = pf_tp; tabct[tid] = pf_tc; }
        if (tid < S) rpt[tid] = rp(pf_dy, pf_dz);
        for (int i = tid + 64; i < n_tab; i += 64) { tabp[i] = k1->d.tab_power[i]; tabct[i] = k1->d.tab_ct[i]; }
        for (int i = tid + 64; i < S; i += 64) rpt[i] = rp(k1->d.rotor_dy[i], k1->d.rotor_dz[i]);

        // this lane's turbine (registers; the cross-lane fields live in the LDS arrays).  Steady inflow: v = w = 0 exactly
        // (the deficits only act on u), so they are constants here and stored as such.
        yaw = l_yaw; tu = l_u; tti = l_ti; oyaw = l_yaw;
        // WindFarmEnv._adjust_yaws (Wind_Farm_Env.py:822-864), float32 like numpy evaluates it on a float32 action
        if (role_live && farm == 0 && valid) {
            const float a = l_act, ymin = k1->p.yaw_min, ymax = k1->p.yaw_max, ystep = k1->p.yaw_step;
            if (k1->p.action_method == WG_ACT_YAW) {
                yaw = fminf(fmaxf(yaw + a * ystep, ymin), ymax);
            } else {
                float tf = a + 1.0f;
                tf = tf * 0.5f;
                tf = tf * (ymax - ymin);
                tf = tf + ymin;
                const float ny = fminf(fmaxf(tf, yaw - ystep), yaw + ystep);
                yaw = fminf(fmaxf(ny, ymin), ymax);
            }
        }
    }
    float tpow = 0.f, tct = 0.f, cg = 1.f;
    float sws = 0.f, swd = 0.f, syaw = 0.f, sp_ = 0.f;
    int part_acc = 0, stream_acc = 0;
    bool stepped = false;
    lds_barrier<64>();
    WG_STAMP(1);

    for (int round = 0;; ++round) {
        bool stepping, is_dev;
        int sub, K;
        {
            const KArgsPtr kr = wg_cold_args();
            K = kr->p.K;
            const int4 sch = *reinterpret_cast<const int4*>(&my.dev_rem);      // (dev_rem, fill_rem, sub, budget)
            const bool active = sch.x > 0 || sch.y > 0;
            stepping = valid && (role_live ? (round < K) : (role_dev && active && (sch.w > 0 || sch.z != 0)));
            is_dev = !role_live && sch.x > 0;
            sub = sch.z;
            if (!__ballot(stepping)) break;
        out.rounds = round + 1;
            // (a further flow step of the launch gathers what the previous one's advection pass stored)
            if (round > 0) full_barrier<64>();

            // BasicControllers.local_yaw_controller / global_yaw_controller (BasicControllers.py:10-73): the running episode's
            // baseline farm, every sim sub-step, not clipped
            if (role_live && farm == 1 && stepping) {
                const float ystep = kr->p.yaw_step;
                if (kr->p.base_controller == WG_CTRL_LOCAL) {
                    // (steady inflow: v == 0 exactly — the deficits only act on u — so atan(v / u) = +-0)
                    const float off = 0.f - yaw;
                    const float sgn = (float)((off > 0.f) - (off < 0.f));
                    yaw = yaw + sgn * fminf(fabsf(off), ystep);
                } else {
                    const float sgn = (float)((yaw > 0.f) - (yaw < 0.f));
                    yaw = yaw - sgn * fminf(fabsf(yaw), ystep);
                }
            }

            // (0) the slot's clock: travel of its chains over this step, particles released (lane t = 0 publishes)
            if (stepping && t == 0) {
                const double dpart = kr->p.dpart;
                const int P = kr->p.P;
                const double s_off = my.s_off, adv = my.ws * kr->p.dt_d;
                double s_new = s_off + adv;
                int n_emit = 0;
                while (s_new >= dpart) { s_new -= dpart; ++n_emit; }
                if (n_emit > P) n_emit = P;
                int new_valid = my.n_valid + n_emit; if (new_valid > P) new_valid = P;
                my.s_new = s_new; my.n_emit = n_emit; my.new_valid = new_valid;
                my.s_off_f = (float)s_off;
                // a target closer than this is bracketed by a particle released in this step
                my.near_f = (float)(s_off + adv + dpart) + 0.01f;
            }
        }
        lds_barrier<64>();
        WG_STAMP(4);

        // (2) candidate pass, lane = target: its sources are the N turbines of its own slot (every lane of a slot reads the same
        // LDS words: a broadcast), tested against the chains' running bounds (those BEFORE this step's records: they cover every
        // particle already in the rings).  A pair close enough to be bracketed by a particle released in this step is tested
        // against the widest record the packed format can hold.  Bit s of cmask = source s is a candidate.
        unsigned cmask = 0u;
        {
            const KArgsPtr kp = wg_cold_args();
            const float D = kp->p.D, inv_D = kp->p.inv_D, eps_max = kp->p.env_eps_max;
            const int gs0 = k * N;
            const float lim0 = kp->p.R_rot + 1.0e-3f * D;
            const float near_f = my.near_f, move_max = my.move_max;
            const float xt_f = Lsrc4[g].x, yt_f = Lsrc4[g].y;
#pragma unroll 4
            for (int s2 = 0; s2 < N; ++s2) {
                const float4 a = Lsrc4[gs0 + s2];
                const float2 b = Lsrc2[gs0 + s2];
                const float dxf = xt_f - a.x;
                const bool nearp = dxf < near_f;
                const float kb_ = nearp ? fmaxf(WG_K_MAX, a.z) : a.z, eb_ = nearp ? fmaxf(eps_max, a.w) : a.w;
                const float sig_max = (kb_ * (dxf * inv_D) + eb_) * D;
                const float bdv = b.x + (((__float_as_uint(b.y) != 0u) | nearp) ? move_max : 0.f);
                const float gap = fabsf(yt_f - a.y) - (5.0f * sig_max + bdv);
                const bool cd = (s2 != t) & (dxf >= 0.f) & (gap <= lim0);
                cmask |= cd ? (1u << s2) : 0u;
            }
            if (!stepping || (SPLIT && role == 1)) cmask = 0u;
        }
        // this target's range of the list: ascending (target lane, source) order by construction
        int cbeg, nc;
        {
            const int cnt = __popc(cmask);
            const int inc = env_scan(cnt, tid);
            cbeg = inc - cnt;
            nc = __builtin_amdgcn_readlane(inc, 63);
            unsigned m = cmask;
            int o = cbeg;
            while (m) { cl[o++] = (unsigned short)((g << 5) | __builtin_ctz(m)); m &= m - 1u; }
        }
        lds_barrier<64>();
        WG_STAMP(5);

        // bracket of candidate c and the gathers of its two bracketing particles in their PRE-step state (advanced by the same
        // m0_advect() the advection pass applies: the phase depends on nothing the pass writes).  Ages j, j + 1 after the
        // step are ages jp0 = j - n_emit, jp0 + 1 before it; negative: released in this step — the turbine's record, at the
        // turbine.  A resting chain's particles sit where they were released: their py is not fetched.
        // (particle addresses: the env's block as a uniform base + 32-bit offsets — an env's 2 F slots span < 4 GB)
        struct Cand { uint2 q0, q1; float y0, y1, xd, wgt; int en, jp0; bool ok, rest; };      // (en: list entry = target lane << 5 | source turbine)
        float dsum = 0.f, tia_max = 0.f;
        {
            const KArgsPtr kp = wg_cold_args();
            const unsigned pstride = (unsigned)kp->p.pstride;
            const size_t pb_env = (size_t)(e * 2 * F + kbase) * pstride;      // particle block of the wave's slot 0 (slot k: + k * pstride)
            // (the brackets' records come from the interleaved record array the advection pass streams: the same lines)
            const uint2* const rc_env = reinterpret_cast<const uint2*>(kp->d.rec_a + 2 * pb_env);
            const float* const py_env = kp->d.py + pb_env;
            const double inv_dpart = kp->p.inv_dpart;
            const float inv_D = kp->p.inv_D;
            auto issue = [&](Cand& cd, const int cidx, const int c0, const int c1) __attribute__((always_inline)) {
                cd.ok = false; cd.rest = false;
                if (cidx >= c1) return;
                def[cidx - c0] = 0.f; tiav[cidx - c0] = 0.f;      // a candidate the exact evaluation drops contributes zero
                const unsigned en = cl[cidx];
                cd.en = (int)en;
                const int gt = (int)(en >> 5);
                const int4 rt = Lring[gt];                        // (.w = the target's slot)
                const int gs = rt.w * N + (int)(en & 31u);
                const EnvSlotLds& q = SL[rt.w];
                const double dx = Lxr[gt] - Lxr[gs];
                if (!(dx > 0.0)) return;                          // (the candidate test ran on float positions)
                cd.xd = (float)dx * inv_D;
                const double xi = (dx - q.s_new) * inv_dpart;
                const double jf = floor(xi);
                cd.wgt = (float)(xi - jf);
                int j = (int)jf;
                if (j < 0) { j = 0; cd.wgt = 0.f; }
                if (j + 1 > q.new_valid - 1) return;              // the chain has not reached the target yet
                const int4 rg = Lring[gs];                        // (roff, rlen, head, slot)
                const int Rs = rg.y, hd = rg.z;
                const unsigned mv = __float_as_uint(Lsrc2[gs].y);
                cd.rest = !(mv != 0u && (int)(q.n_emitted - mv) < Rs);
                cd.jp0 = j - q.n_emit;
                int r0 = hd - cd.jp0; if (r0 < 0) r0 += Rs;
                int r1 = hd - cd.jp0 - 1; if (r1 < 0) r1 += Rs;
                if (cd.jp0 < 0) r0 = 0;           // (released in this step: nothing to fetch — any slot of the ring will do)
                if (cd.jp0 + 1 < 0) r1 = 0;
                const unsigned sb = (unsigned)rg.w * pstride + (unsigned)rg.x;
                cd.q0 = rc_env[sb + (unsigned)r0]; cd.q1 = rc_env[sb + (unsigned)r1];
                cd.y0 = 0.f; cd.y1 = 0.f;
                if (!cd.rest) { cd.y0 = py_env[sb + (unsigned)r0]; cd.y1 = py_env[sb + (unsigned)r1]; }
#if WG_ENV_ABLATE & 1      // (profiling builds, wrong results: no bracket gathers)
                cd.q0 = make_uint2(Lrec4[gs].x, Lrec4[gs].y); cd.q1 = cd.q0; cd.y0 = cd.y1 = Lsrc4[gs].y;
#endif
                cd.ok = true;
            };
            Cand ca, cb_;
            issue(ca, tid, 0, min(nc, WG_ENV_CAP));
            WG_STAMP(11);

            // (1) emission records of this step, sin / cos of the yaw
            if (stepping) {
                const KArgsPtr kq = wg_cold_args();
                const int n_tab = kq->p.n_tab;
                const float* const tabct = reinterpret_cast<const float*>(smem + kq->p.env_off_tab) + n_tab;
                const float gy = yaw * WG_DEG2RAD_F;
                const float sg = __sinf(gy);
                cg = __cosf(gy);
                const float wsn = fmaxf(tu * cg, 0.0f);
                const float ctx = fminf(fmaxf(env_tab(tabct, kq->p.tab_x0, kq->p.tab_inv_dx, n_tab, wsn) * cg * cg, 0.0f), 0.96f);
                const float rq = __builtin_amdgcn_sqrtf(1.0f - ctx);
                const float beta = 0.5f * (1.0f + rq) * __builtin_amdgcn_rcpf(rq);
                const float rk = kq->p.ka * tti + kq->p.kb;
                const float reps = kq->p.eps0 * __builtin_amdgcn_sqrtf(beta);
                const float rhv = -kq->p.hill * sg * tu;
                // the packed record saturates outside [0, WG_K_MAX] x [-WG_HV_MAX, WG_HV_MAX]: never silently (wg_check reports it)
                if (rk > WG_K_MAX || fabsf(rhv) > WG_HV_MAX) atomicOr(kq->d.status, WG_STATUS_BIT_RANGE);
                // (steady inflow: u_e <= U — stored as their ratio)
                const unsigned na_ = pack_a(ctx, rk), nb_ = pack_b(tu * __builtin_amdgcn_rcpf(kq->p.ue_scale * my.ws_f), rhv);
                float4 s4 = Lsrc4[g];
                s4.z = fmaxf(s4.z, rk + WG_K_MAX / 65535.0f);
                s4.w = fmaxf(s4.w, reps + 1.0f / 65535.0f);
                Lsrc4[g] = s4;
                const int n_emit = my.n_emit;
                if (n_emit > 0 && rec_moves(nb_)) Lsrc2[g].y = __uint_as_float(my.n_emitted + (unsigned)n_emit);
                Lrec4[g] = make_uint4(na_, nb_, __float_as_uint(tu), __float_as_uint(cg));
            }
            lds_barrier<64>();
            WG_STAMP(2);

            // exact evaluation, one candidate per lane and batch; results staged per candidate, WG_ENV_CAP at a time
            const KArgsPtr ke2 = wg_cold_args();
            const float dpart_f = ke2->p.dpart_f, D = ke2->p.D, dt = ke2->p.dt, R_rot = ke2->p.R_rot, eps0 = ke2->p.eps0, ue_scale = ke2->p.ue_scale;
            const float tia = ke2->p.no_ti_fold ? 0.f : ke2->p.tia, tib = ke2->p.tib, tid_ = ke2->p.tid, inv_S = ke2->p.inv_S;
            const int S = ke2->p.S;
            const float2* const rpt = reinterpret_cast<const float2*>(smem + ke2->p.env_off_tab + 8 * ke2->p.n_tab);
            auto eval_cand = [&](const Cand& cd, const int pos) __attribute__((always_inline)) {
                    if (!cd.ok) return;
#ifdef WG_TIMELINE
                    atomicAdd(&wg_counts[1], 1);
#endif
                    const int gt = cd.en >> 5;
                    const int kt = Lring[gt].w;
                    const int gs = kt * N + (cd.en & 31);
                    const float4 st = Lsrc4[gt];
                    const float ysrc = Lsrc4[gs].y;
                    const EnvSlotLds& q = SL[kt];
                    const int jp0 = cd.jp0, jp1 = jp0 + 1;
                    float py0 = cd.rest ? ysrc : cd.y0, py1 = cd.rest ? ysrc : cd.y1;
                    unsigned a0 = cd.q0.x, b0_ = cd.q0.y, a1 = cd.q1.x, b1_ = cd.q1.y;
                    if (jp0 < 0) {      // (jp1 < 0 implies jp0 < 0)
                        const uint4 rn = Lrec4[gs];
                        py0 = ysrc; a0 = rn.x; b0_ = rn.y;
                        if (jp1 < 0) { py1 = ysrc; a1 = rn.x; b1_ = rn.y; }
                    }
                    const float ue_max = ue_scale * q.ws_f;
                    const float u0 = rec_uf(b0_) * ue_max, u1 = rec_uf(b1_) * ue_max;
                    {
                        const int nv = q.n_valid;
                        const float sof = q.s_off_f;
                        // (a resting chain's particles do not move in this step either: hv = 0 in every record)
                        if (!cd.rest && jp0 >= 0 && jp0 < nv) py0 = m0_advect(py0, a0, b0_, jp0, sof, dpart_f, inv_D, dt, eps0);
                        if (!cd.rest && jp1 >= 0 && jp1 < nv) py1 = m0_advect(py1, a1, b1_, jp1, sof, dpart_f, inv_D, dt, eps0);
                    }
                    // interpolation, lateral cut-off, Gaussian deficit at the S rotor points (k_flow: eval_pair)
                    const float wgt = cd.wgt;
                    const float w0 = 1.0f - wgt, w1 = wgt;
                    const float yc = w0 * py0 + w1 * py1;
                    const float kv = w0 * rec_k(a0) + w1 * rec_k(a1);
                    const float epv = w0 * rec_eps(a0, eps0) + w1 * rec_eps(a1, eps0);
                    const float xd = cd.xd;
                    const float sp = kv * xd + epv;
                    const float sig = sp * D;
                    const float yt = st.y;
                    const float rc2 = (yt - yc) * (yt - yc);      // (+ (hub - zc)^2 = 0: steady inflow keeps the wake centre at hub height)
                    const float rcut = R_rot + 5.0f * sig;
                    if (rc2 > rcut * rcut) return;
#ifdef WG_TIMELINE
                    atomicAdd(&wg_counts[2], 1);
                    if (rc2 <= (R_rot + 3.0f * sig) * (R_rot + 3.0f * sig)) atomicAdd(&wg_counts[3], 1);
#endif
                    const float ctv = w0 * rec_ct(a0) + w1 * rec_ct(a1);
                    const float uev = w0 * u0 + w1 * u1;
                    const float inv2s2 = __builtin_amdgcn_rcpf(2.0f * sig * sig);
                    const float cf = m0_cfrac(ctv, sp);
                    // wake-added turbulence (Crespo-Hernandez), Gaussian-weighted at the hub
                    const float ind = 0.5f * (1.0f - __builtin_amdgcn_sqrtf(1.0f - ctv));
                    tiav[pos] = tia * fast_pow(ind, tib) * q.ti_pow * fast_pow(fmaxf(xd, 1.0f), tid_) * __expf(-rc2 * inv2s2);
                    const float cgt = __uint_as_float(Lrec4[gt].w), amp = uev * cf;
                    const float ninv = -inv2s2;
                    float acc = 0.f;
#pragma unroll 4
                    for (int sI = 0; sI < ((WG_ENV_ABLATE & 8) ? 1 : S); ++sI) {
                        const float2 rp = rpt[sI];
                        const float dy = yt + rp.x * cgt - yc;
                        acc += amp * __expf((dy * dy + rp.y) * ninv);
                    }
                    def[pos] = acc * inv_S;
            };
#ifdef WG_TIMELINE
            if (tid == 0) atomicAdd(&wg_counts[0], nc);
#endif
            for (int c0 = 0; c0 < nc; c0 += WG_ENV_CAP) {
                const int c1 = min(nc, c0 + WG_ENV_CAP);
                if (c0 > 0) { lds_barrier<64>(); issue(ca, c0 + tid, c0, c1); }
                // (two candidate buffers, alternating: the next batch's gathers are in flight while this one is evaluated)
#if WG_ENV_PP & 2
                for (int cb = c0; cb < c1; cb += 128) {
                    issue(cb_, cb + 64 + tid, c0, c1);
                    eval_cand(ca, cb - c0 + tid);
                    if (cb + 64 >= c1) break;
                    issue(ca, cb + 128 + tid, c0, c1);
                    eval_cand(cb_, cb + 64 - c0 + tid);
                }
#else
                for (int cb = c0; cb < c1; cb += 64) {
                    cb_ = ca;
                    issue(ca, cb + 64 + tid, c0, c1);
                    eval_cand(cb_, cb - c0 + tid);
                }
#endif
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
        }
        if (SPLIT && split_on && role == 0) {
            // every gather of this step has landed (the evaluation consumed them): the pass wave may store
            __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0)
            if (tid == 0) env_flag_set(reinterpret_cast<int*>(smem_pass + WG_ENV_OFF_HDR));
            // (roofline accounting of the pass this wave does not run: particles the step reads or writes)
            if (stepping) {
                const int4 rg = Lring[g];
                const unsigned mvl = __float_as_uint(Lsrc2[g].y);
                const int n_emit = my.n_emit;
                const bool full = (mvl != 0u && (int)(my.n_emitted - mvl) < rg.y) || n_emit >= 4 || n_emit >= rg.y;
                stream_acc += full ? rg.y : n_emit;
            }
        }
        if (stepping) {
            const float ti_f = my.ti_f;
            tu = my.ws_f;
            tu -= dsum;
            tti = __builtin_amdgcn_sqrtf(ti_f * ti_f + tia_max * tia_max);
        }
        lds_barrier<64>();          // (the quad list below aliases the staging arrays)
        WG_STAMP(9);

        if (!(SPLIT && split_on && role == 0)) {
            const KArgsPtr kp = wg_cold_args();
            const unsigned pstride = (unsigned)kp->p.pstride;
            const size_t pb_env = (size_t)(e * 2 * F + kbase) * pstride;
            float* const py_env = kp->d.py + pb_env;
            unsigned* const ra_env = kp->d.rec_a + 2 * pb_env;        // interleaved (ct|k, eps|hv) record
            // (2) quad list of the chains that move (TurbLds::mvl, k_flow): a chain is listed whole while it may hold a particle
            // of a yawed turbine; a resting chain only receives this step's new particles, stored straight to their ring slots
            int nlist;
            {
                int cnt = 0, nqd = 0;
                bool full = false;
                if (stepping) {
                    const int4 rg = Lring[g];                     // (roff, rlen, head, slot)
                    const int rlen = rg.y, n_emit = my.n_emit;
                    const unsigned mvl = __float_as_uint(Lsrc2[g].y);
                    nqd = rlen >> 2;
                    const bool moving = mvl != 0u && (int)(my.n_emitted - mvl) < rlen;
                    full = moving || n_emit >= 4 || n_emit >= rlen;
                    stream_acc += full ? rlen : n_emit;        // roofline accounting: particles this step reads or writes
                    if (full) cnt = nqd;
                    else if (n_emit > 0) {
                        const unsigned sb = (unsigned)k * pstride + (unsigned)rg.x;
                        const uint4 rn = Lrec4[g];
                        const float y0 = Lsrc4[g].y;
#pragma unroll
                        for (int em = 0; em < 3; ++em) {
                            if (em < n_emit) {
                                int r = rg.z + 1 + em; if (r >= rlen) r -= rlen;
                                const unsigned ix = sb + (unsigned)r;
                                py_env[ix] = y0;
                                reinterpret_cast<uint2*>(ra_env)[ix] = make_uint2(rn.x, rn.y);
                            }
                        }
                    }
                }
                const int inc = env_scan(cnt, tid);
                nlist = (WG_ENV_ABLATE & 2) ? 0 : __builtin_amdgcn_readlane(inc, 63);
                if (full) {
                    const int base = inc - cnt;
                    const unsigned tag = (unsigned)g << 10;
                    for (int i = 0; i < nqd; ++i) ql[base + i] = (unsigned short)(tag | (unsigned)i);
                }
            }
            lds_barrier<64>();
            WG_STAMP(10);
#ifdef WG_TIMELINE
            if (tid == 0) atomicAdd(&wg_counts[4], nlist);
#endif

            // advection pass, software-pipelined: a lane requests its next listed quad before it computes the current one
            // (vmcnt counts loads and stores in ONE in-order queue: a plain load-compute-store loop waits for the previous
            // trip's stores whenever it waits for its loads)
            const float dpart_f = kp->p.dpart_f, inv_D = kp->p.inv_D, dt = kp->p.dt, eps0 = kp->p.eps0;
            struct QuadReq { float4 py; uint4 ra, rb; int g, kq; unsigned q; };
            auto request = [&](QuadReq& r, const int cidx) __attribute__((always_inline)) {
                const bool v = cidx < nlist;
                const unsigned en = v ? ql[cidx] : 0u;
                r.g = (int)(en >> 10); r.kq = (int)(en & 1023u);
                const int4 rg = Lring[r.g];
                r.q = (((unsigned)rg.w * pstride + (unsigned)rg.x) >> 2) + (unsigned)r.kq;      // quad index inside the env's block
                if (v) {
                    r.py = reinterpret_cast<const float4*>(py_env)[r.q];
                    r.ra = reinterpret_cast<const uint4*>(ra_env)[2u * r.q];
                    r.rb = reinterpret_cast<const uint4*>(ra_env)[2u * r.q + 1u];
                }
            };
            auto advect_quad = [&](const QuadReq& cur, const int cidx) __attribute__((always_inline)) {
                if (cidx >= nlist) return;
                const int gq = cur.g, kq = cur.kq;
                const unsigned q = cur.q;
                const int4 rg = Lring[gq];
                const EnvSlotLds& sl = SL[rg.w];
                const int n_emit_q = sl.n_emit, n_valid_q = sl.n_valid;
                const float sof = sl.s_off_f;
                const int R = rg.y, hd = rg.z;
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
                    const float adv = m0_advect(pyv[i], rav[i], rbv[i], j, sof, dpart_f, inv_D, dt, eps0);
                    pyv[i] = j < n_valid_q ? adv : pyv[i];
                }
                const float y0 = Lsrc4[gq].y;
                if (emits) {
                    const uint4 rn = Lrec4[gq];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int ei = e0 + i; if (ei >= R) ei -= R;
                        if (ei < n_emit_q) {
                            pyv[i] = y0; rav[i] = rn.x; rbv[i] = rn.y;
                        }
                    }
                    reinterpret_cast<uint4*>(ra_env)[2u * q] = make_uint4(rav[0], rbv[0], rav[1], rbv[1]);
                    reinterpret_cast<uint4*>(ra_env)[2u * q + 1u] = make_uint4(rav[2], rbv[2], rav[3], rbv[3]);
                }
                reinterpret_cast<float4*>(py_env)[q] = make_float4(pyv[0], pyv[1], pyv[2], pyv[3]);
                // (excursion bound over the VALID particles of the quad only, see k_flow)
                float ex = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int j = j0 - i; if (j < 0) j += R;
                    if (j < n_valid_q) ex = fmaxf(ex, fabsf(pyv[i] - y0));
                }
                if (ex > Lsrc2[gq].x) atomicMax(reinterpret_cast<int*>(&Lsrc2[gq].x), __float_as_int(ex));   // ex >= 0: int order == float order
            };
            // (two request buffers, alternating: no register copies between the trips)
            QuadReq qa, qb;
            qa.py = qb.py = make_float4(0.f, 0.f, 0.f, 0.f); qa.ra = qa.rb = qb.ra = qb.rb = make_uint4(0u, 0u, 0u, 0u);
            request(qa, tid);
            if (SPLIT && role == 1) {      // (its first trip is in flight; no store before the main wave's gathers have landed)
                if (!env_flag_wait(reinterpret_cast<int*>(smem + WG_ENV_OFF_HDR))) atomicOr(kp->d.status, WG_STATUS_BIT_STATE);
            }
#if WG_ENV_PP & 1
            for (int base = 0; base < nlist; base += 128) {
                request(qb, base + 64 + tid);
                advect_quad(qa, base + tid);
                if (base + 64 >= nlist) break;
                request(qa, base + 128 + tid);
                advect_quad(qb, base + 64 + tid);
            }
#else
            for (int base = 0; base < nlist; base += 64) {
                qb = qa;
                request(qa, base + 64 + tid);
                advect_quad(qb, base + tid);
            }
#endif
        }
        lds_barrier<64>();
        WG_STAMP(3);
        if (SPLIT && role == 1) {      // the pass wave: the chains' excursion bounds, nothing else
            if (stepping) reinterpret_cast<float4*>(wg_cold_args()->d.bnd)[(unsigned)((e * 2 * F + kbase) * N + g)].x = Lsrc2[g].x;
            return;
        }

        // per-turbine tail: the step's emissions are in the ring now; power / thrust with the current yaw (model M0 step 5),
        // WindFarmEnv._take_measurements (Wind_Farm_Env.py:480-495) accumulated over the K sub-steps and, at the end of the env
        // step, farm_mes.add_measurements' ring push (MesClass.py:568-591)
        const bool measuring = stepping && !is_dev;
        const bool unit_end = measuring && (sub + 1 == K);
        const KArgsPtr kc = wg_cold_args();
        const float inv_k = 1.0f / (float)K;
        const unsigned ctx_id = (unsigned)(e * 2 + c);
        if (stepping) {
            {
                int4 rg = Lring[g];
                const int n_emit = my.n_emit;
                rg.z += n_emit;
                if (n_emit >= rg.y) rg.z %= rg.y; else if (rg.z >= rg.y) rg.z -= rg.y;
                Lring[g].z = rg.z;
                part_acc += min(my.new_valid, rg.y);          // roofline accounting: particles that can still reach a rotor
            }
            stepped = true;
            const int n_tab = kc->p.n_tab;
            const float tab_x0 = kc->p.tab_x0, tab_inv_dx = kc->p.tab_inv_dx;
            const float* const tabp = reinterpret_cast<const float*>(smem + kc->p.env_off_tab);
            const float wsn = fmaxf(tu * cg, 0.0f);
            tpow = env_tab(tabp, tab_x0, tab_inv_dx, n_tab, wsn);
            tct = env_tab(tabp + n_tab, tab_x0, tab_inv_dx, n_tab, wsn) * cg * cg;
            if (measuring && farm == 0) {
                const float wsm = __builtin_amdgcn_sqrtf(tu * tu);
                const float wdm = 0.f + my.wd_env;      // (steady inflow: atan(v / u) = +-0, see k_flow)
                float val[WG_N_CH] = {sws + wsm, swd + wdm, syaw + yaw, sp_ + tpow};
                if (unit_end) {
                    const int n_pushed = my.n_pushed;
                    const unsigned tcx = ctx_id * (unsigned)N + (unsigned)t;
                    kc->d.cur_ws[tcx] = wsm;
                    kc->d.cur_wd[tcx] = wdm;
                    if (K != 1) {
#pragma unroll
                        for (int ch = 0; ch < WG_N_CH; ++ch) val[ch] *= inv_k;
                    }
                    if (NOISE) {
                        const uint64_t noise_key = ((CEnvPtr)(kc->d.env + e))->noise_key;      // (set by k_init only: read-only in every step kernel)
                        const uint32_t episode_tag = my.c_tag;
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
                        rbase[(unsigned)(kc->p.ring_off[ch] + fast_mod(n_pushed, H, kc->p.inv_hlen[ch]) * N + t)] = val[ch];      // (time-major: WgRing)
                    }
                    if (SPLIT) {      // (what the glue would read back from the rings)
#pragma unroll
                        for (int ch = 0; ch < WG_N_CH; ++ch) out.g_nw[ch] = val[ch];
                    }
                    // stage the pushed values for the farm-level mean / mean / sum
                    sws = val[0]; swd = val[1]; sp_ = val[3];
                } else {
                    sws = val[0]; swd = val[1]; syaw = val[2]; sp_ = val[3];
                }
            }
        }
        // farm-level values of every measuring slot at once: agent farms push mean ws / mean wd / total power into the farm
        // rings at the end of an env step, baseline farms add up their power every sub-step (fs_baseline...power().sum(), :954)
        float a_ws = 0.f, a_wd = 0.f, a_pw = 0.f;
        if (__ballot(measuring)) {
            a_ws = env_slot_sums((measuring && farm == 0) ? sws : 0.f, N, NS, k, tid);
            a_wd = env_slot_sums((measuring && farm == 0) ? swd : 0.f, N, NS, k, tid);
            a_pw = env_slot_sums(measuring ? (farm == 0 ? sp_ : tpow) : 0.f, N, NS, k, tid);
        }
        // the slot's lane t = 0: farm-level pushes, clock advance, schedule
        if (stepping && t == 0) {
            const int n_emit = my.n_emit;
            my.n_valid = my.new_valid; my.s_off = my.s_new; my.n_emitted += (unsigned)n_emit;
            my.n_flow += 1; my.budget -= 1;
            float base_acc = my.base_acc;
            if (measuring && farm == 1) base_acc += a_pw;
            if (unit_end) {
                const FlowP __attribute__((address_space(4)))& pc = kc->p;
                if (farm == 0) {
                    const int n_pushed = my.n_pushed, pend_farm_n = my.pend_farm_n;
                    float* fbase = kc->d.fring + (size_t)ctx_id * pc.fring_stride;
                    fbase[pc.fring_off[WG_CH_WS] + umod_small(n_pushed, pc.hlen[WG_CH_WS], pc.hmagic[WG_CH_WS])] = a_ws * pc.inv_N;
                    fbase[pc.fring_off[WG_CH_WD] + umod_small(n_pushed, pc.hlen[WG_CH_WD], pc.hmagic[WG_CH_WD])] = a_wd * pc.inv_N;
                    fbase[pc.fring_off[WG_CH_POWER] + umod_small(n_pushed, pc.hlen[WG_CH_POWER], pc.hmagic[WG_CH_POWER])] = a_pw;
                    my.out_pw = a_pw;
                    if (role_live) kc->d.step_farm_pow[e] = a_pw;
                    else {
                        kc->d.pend_farm[(size_t)ctx_id * pc.power_avg + umod_small(pend_farm_n, pc.power_avg, pc.pavg_magic)] = a_pw;
                        my.pend_farm_n = pend_farm_n + 1;
                    }
                    my.n_pushed = n_pushed + 1;
                } else {
                    const float bp = K == 1 ? base_acc : base_acc * inv_k;
                    my.out_pw = bp;
                    if (role_live) kc->d.step_base_pow[e] = bp;
                    else {
                        const int pend_base_n = my.pend_base_n;
                        kc->d.pend_base[(size_t)ctx_id * pc.power_avg + umod_small(pend_base_n, pc.power_avg, pc.pavg_magic)] = bp;
                        my.pend_base_n = pend_base_n + 1;
                    }
                    base_acc = 0.f;
                }
            }
            my.base_acc = base_acc;
            if (is_dev) my.dev_rem -= 1;
            else if (sub + 1 >= K) { my.sub = 0; if (!role_live) my.fill_rem -= 1; }
            else my.sub = sub + 1;
        }
        if (unit_end && farm == 0) { sws = 0.f; swd = 0.f; syaw = 0.f; sp_ = 0.f; }
        lds_barrier<64>();
        WG_STAMP(6);
    }

    WG_STAMP(7);
    // ---- epilogue: the env's slots back to memory (the slots that took a step) ---------------------------------------------
    const KArgsPtr ke = wg_cold_args();
    const float part_slot = env_slot_sums((float)part_acc, N, NS, k, tid);      // (exact: far below 2^24)
    const float stream_slot = env_slot_sums((float)stream_acc, N, NS, k, tid);
    if (valid && stepped) {
        const unsigned tb = (unsigned)((e * 2 * F + kbase) * N + g);
        const float4 s4 = Lsrc4[g];
        const float2 s2 = Lsrc2[g];
        ke->d.yaw[tb] = yaw; ke->d.u[tb] = tu; ke->d.v[tb] = 0.f; ke->d.w[tb] = 0.f;
        ke->d.ti_loc[tb] = tti; ke->d.power[tb] = tpow; ke->d.ct[tb] = tct;
        if (SPLIT && split_on) {      // (the excursion bound is the pass wave's)
            float4* const bp = reinterpret_cast<float4*>(ke->d.bnd) + tb;
            bp->y = s4.z; bp->z = s4.w; bp->w = s2.y;
        } else reinterpret_cast<float4*>(ke->d.bnd)[tb] = make_float4(s2.x, s4.z, s4.w, s2.y);
        if (role_live && farm == 0) ke->d.old_yaw[(unsigned)(e * N + t)] = oyaw;
        if (t == 0) {
            const int n_flow = my.n_flow, P = ke->p.P;
            const double dt_d = ke->p.dt_d;
            double tm = my.c_time;
            for (int i = 0; i < n_flow; ++i) tm += dt_d;
            const unsigned n_emitted = my.n_emitted;
            int hd = my.c_head + (int)((n_emitted - my.n_emitted0) % (unsigned)P); if (hd >= P) hd -= P;
            WgSlot& slot = ke->d.slot[(unsigned)(e * 2 * F + kbase + k)];
            slot.part_count = my.c_part + (unsigned)part_slot;
            slot.add_count = my.c_add + (unsigned)stream_slot;      // (steady inflow: no wake-added turbulence lookups — the word counts the
                                                          // particles the advection passes actually touched, wg_added_lookups)
            slot.head = hd; slot.n_valid = my.n_valid; slot.s_off = my.s_off; slot.time = tm;
            slot.istep = my.c_istep + (unsigned)n_flow; slot.n_emitted = n_emitted;
            slot.dev_remaining = my.dev_rem; slot.fill_remaining = my.fill_rem;
            slot.flow_count = my.c_flow + (unsigned)n_flow;
            WgCtx& cx = ke->d.ctx[(unsigned)(e * 2 + c)];
            if (farm == 0) { cx.n_pushed = my.n_pushed; cx.pend_farm_n = my.pend_farm_n; }
            if (farm == F - 1) cx.pend_base_n = my.pend_base_n;
        }
    }
    if (SPLIT) {      // (the running episode's main wave: lane t of the agent farm is the glue's lane t)
        out.g_yaw = yaw; out.g_old = oyaw; out.g_pw = tpow;
        out.g_pwb = F == 2 ? __shfl(tpow, min(tid + N, 63), 64) : 0.f;
    }
    WG_STAMP(8);
    // A background episode whose development is complete: its window sums and first observation are prepared for the swap
    // (wg_first_obs, see k_flow) — in the launch AFTER the one that completed it (WG_ENV_FIRST_OBS_LATER; development ends
    // WG_SHADOW_MARGIN steps early, so there is one): its slots rest then, and the wave that builds the observation is not also
    // the one that took the episode's last flow steps.  An episode that completes in the very launch of the truncating step is
    // prepared at once (one wave per env) or summed by the swap itself (lean_swap's fallback).
    if (mode == WG_MODE_STEP && (WPE == 1 || c != env_live) && !defer_init) {
        const EnvSlotLds& bs = SL[WPE == 2 ? 0 : (env_live ^ 1) * F];      // the background context's agent farm
        const int bctx = e * 2 + (env_live ^ 1);
        bool build = false;
        if (ke->p.autoreset && bs.dev_rem == 0 && bs.fill_rem == 0) {
            if (WG_ENV_FIRST_OBS_LATER != 0) build = bs.n_flow == 0 ? (ke->d.gd->next_obs_ok != nullptr && ke->d.gd->next_obs_ok[bctx] == 0) : out.truncates != 0;
            else build = bs.n_flow > 0;
        }
        if (build) {
            const int np = bs.n_pushed;
            out.first_obs = 1;
            full_barrier<64>();                            // the ring pushes have left the wave
            env_first_obs<GLUE != 0>(bctx, np, tid);
        }
    }
    if (defer_init) {
        // (see the prologue) the retired context's next episode, set up after the wave's own step; the slots' remaining work goes
        // to the LDS records the glue plans the first share from
        env_init_episode<GLUE != 0>(e, env_live ^ 1, tid);
        full_barrier<64>();
        if (valid && t == 0 && c != env_live) {
            const WgSlot& slot = ke->d.slot[(unsigned)(e * 2 * F + kbase + k)];
            my.dev_rem = slot.dev_remaining; my.fill_rem = slot.fill_remaining;
        }
        lds_barrier<64>();
    }
}


// k_flow_env<NOISE, 0>: the flow step alone (RESET-mode development; handles whose glue is a separate launch).
// k_flow_env<NOISE, 1 / 2>: step() as ONE launch — the env's wave runs its glue (lean_step: sums-mode handles without TI /
// farm-level entries; 2 = with the per-agent observation buffer of the PettingZoo facade) as the tail of its flow step.  No
// cross-workgroup dependency: the wave owns both contexts of its env.  (Wind_Farm_Env.py:920-1034 in one kernel.)
template <bool NOISE, int GLUE, int WPE, int SPLIT = 0>
__global__ void __launch_bounds__(64 * (WPE + SPLIT), WG_ENV_WAVES)
k_flow_env(const FlowP p_, const FlowPtrs d_, const int mode, const float* __restrict__ actions,
           const uint8_t* __restrict__ mask, const int chunk, const WgParams gp_, const WgPtrs gd_,
           float* __restrict__ obs_out, float* __restrict__ reward_out, uint8_t* __restrict__ trunc_out,
           float* __restrict__ final_obs_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef WG_TIMELINE
    if (threadIdx.x == 0) { for (int i = 0; i < 20; ++i) wg_stamps[i] = 0; wg_stamps[14] = wall_clock64(); for (int i = 0; i < 8; ++i) wg_counts[i] = 0; }
    __syncthreads();
#endif
    // (WPE 2: wave c of the workgroup serves context c of the env, in its own LDS region)
    const int wv4 = WPE == 2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    // (SPLIT: wave 2 = the pass wave of the running episode's context — which one, it reads from the env's header first: the main
    // waves cannot rewrite it before the prologue's workgroup barrier, where this wave arrives with its loads done)
    // (SPLIT 2: waves 2, 3 = the pass waves of contexts 0, 1)
    int wv = wv4, role = 0;
    if (SPLIT == 1 && wv4 == 2) {
        role = 1;
        wv = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(wg_cold_args()->d.env + blockIdx.x)[offsetof(WgEnv, live) / 4]) & 1;
    }
    if (SPLIT == 2) { wv = wv4 & 1; role = wv4 >> 1; }
    const int lds_wave = WPE == 2 ? wg_cold_args()->p.env_lds : 0;
    char* const sm = smem + wv4 * lds_wave;
    char* const sm_pass = smem + (SPLIT == 2 ? 2 + wv : 2) * lds_wave;
    float* const zone = SPLIT ? reinterpret_cast<float*>(smem + (2 + SPLIT) * lds_wave) : nullptr;
    EnvFlowOut fo;
    env_flow<NOISE, WPE, GLUE, SPLIT>(sm, sm_pass, zone, wv, role, mode, actions, mask, chunk, fo);
    if (SPLIT && role == 1) {      // (a pass wave: the workgroup barrier of a truncating step, nothing else)
        if (GLUE != 0 && fo.truncates) { __builtin_amdgcn_s_waitcnt(0x0070); __syncthreads(); }
        return;
    }
    if (GLUE != 0) {
        // (every store of the flow part — rings, turbine state, headers, a prepared first observation — has left the wave
        // before anything reads it back; LDS still holds the slots' records)
        full_barrier<64>();
        if (WPE == 2) {
            // Two waves per env.  The glue needs nothing of the background context's wave unless the env truncates in this step
            // (then it swaps that context in): the waves meet at a workgroup barrier ONLY then — both know from the env's header
            // — and otherwise run to their ends side by side: the background wave plans its own next share and clears its
            // set-up flag, which the glue does for one wave per env.
            if (wv != fo.env_live) {
                const EnvKArgsPtr kb = (EnvKArgsPtr)wg_cold_args();
                const int e = (int)blockIdx.x;
                if (kb->p.autoreset && !fo.truncates && (threadIdx.x & 63) == 0) {
                    const int F = kb->p.F, K = kb->p.K;
                    const EnvSlotLds* const SLb = reinterpret_cast<const EnvSlotLds*>(sm + WG_ENV_OFF_SL);
                    int work = 0;
                    for (int f = 0; f < F; ++f) work = max(work, SLb[f].dev_rem + K * SLb[f].fill_rem);
                    const int steps_done = fo.steps_done + 1, time_max = fo.time_max_live;      // (as the glue sees them: from the prologue's
                                                                                              // copy — the live wave may have rewritten the header by now)
                    const int inc = 1 + (kb->gp.extra_inc ? 1 : 0);
                    const long total = (long)((time_max + inc - 1) / inc) + 1;
                    kb->d.env_rw[e].shadow_iters = work == 0 ? 0 : wg_shadow_share(work, total - steps_done, steps_done, e);
                }
                if (fo.bg_init_pending && (threadIdx.x & 63) == 0) kb->d.ctx[e * 2 + wv].init_pending = 0;
                if (fo.truncates) __builtin_amdgcn_s_waitcnt(0x0070);       // (its last stores, before the barrier releases the glue)
            }
            if (fo.truncates) __syncthreads();
        }
        if (WPE == 1 || wv == fo.env_live) {
            // (the glue's parameter blocks are read where they are used, through the opaque kernarg pointer: by value they were
            // all fetched at the kernel's entry and 130 of them parked in VGPR lanes across the flow step)
            const EnvKArgsPtr kg = (EnvKArgsPtr)wg_cold_args();
            const int F = kg->p.F, K = kg->p.K;
            const EnvSlotLds* const SLa = reinterpret_cast<const EnvSlotLds*>(sm + WG_ENV_OFF_SL);      // the live context's slots
            const int la = WPE == 2 ? 0 : fo.env_live * F, lb = (fo.env_live ^ 1) * F;
            LeanFused fz;
            fz.fp = SLa[la].out_pw;
            fz.bp = F == 2 ? SLa[la + 1].out_pw : 0.f;
            int work = 0;
            if (WPE == 1) for (int f = 0; f < F; ++f) work = max(work, SLa[lb + f].dev_rem + K * SLa[lb + f].fill_rem);
            fz.work = work;
            fz.bg_init_pending = WPE == 2 ? 0 : fo.bg_init_pending;
            fz.plan_elsewhere = WPE == 2;
            fz.hw = reinterpret_cast<const int*>(sm + WG_ENV_OFF_HDR)[threadIdx.x & 31];
            fz.pre = nullptr;
            if (SPLIT) {
                // the glue's inputs: the step's own from this wave's registers, the rest from the pass wave's fetch (long done)
                if (!env_flag_wait(reinterpret_cast<int*>(zone) + LEAN_PRE_FLAG)) atomicOr(kg->d.status, WG_STATUS_BIT_STATE);
                const int gl = (int)(threadIdx.x & 63), own = gl < kg->p.N ? gl : 0;
                fz.pre = zone;
#pragma unroll
                for (int ch = 0; ch < WG_N_CH; ++ch) fz.nw[ch] = __shfl(fo.g_nw[ch], own, 64);
                fz.yaw = fo.g_yaw; fz.old_yaw = fo.g_old; fz.pw = fo.g_pw; fz.pwb = fo.g_pwb;
            }
            WG_STAMP(12);
            lean_step<GLUE == 2, false, true>(*(const WgParams*)&kg->gp, *(const WgPtrs*)&kg->gd, kg->d.gp, kg->d.gd, (int)blockIdx.x,
                                              (int)(threadIdx.x & 63), kg->obs, kg->reward, kg->trunc, kg->final_obs, nullptr, fz);
            WG_STAMP(13);
        }
    }
#ifdef WG_TIMELINE
    {
        const KArgsPtr kt = wg_cold_args();
        if (threadIdx.x == 0 && kt->d.dbg) {
            wg_stamps[15] = wall_clock64();
            long long* row = kt->d.dbg + (size_t)blockIdx.x * 32;      // (the buffer holds 16 words per farm slot: 64 per env)
            for (int kq = 0; kq < 16; ++kq) row[kq] = wg_stamps[kq];
            row[16] = fo.bg_init_pending; row[17] = fo.rounds; row[18] = fo.first_obs;
            row[19] = GLUE != 0 ? (long long)kt->d.env[blockIdx.x].timestep : 0;      // (0 right after a swap)
            for (int kq = 0; kq < 8; ++kq) row[20 + kq] = wg_counts[kq];
            for (int kq = 0; kq < 4; ++kq) row[28 + kq] = wg_stamps[16 + kq];      // glue: header in / loads in / reward + metrics / observation      // candidates listed / fetched / inside 5 sigma / inside 3 sigma, quads listed
        }
    }
#endif
}

extern "C" void wg_launch_flow_env(const FlowP* p, const FlowPtrs* d, int mode, const float* actions, const uint8_t* mask,
                                   int chunk, hipStream_t st) {
    const int grid = p->B, wpe = p->env_wpe == 2 ? 2 : 1;
    const size_t lds = (size_t)p->env_lds * wpe;
    static const WgParams gp0{};
    static const WgPtrs gd0{};
#define WG_FLOW_ENV(NZ, W) hipLaunchKernelGGL((k_flow_env<NZ, 0, W>), dim3(grid), dim3(64 * W), lds, st, *p, *d, mode, actions, mask, chunk, gp0, gd0, \
                                              (float*)nullptr, (float*)nullptr, (uint8_t*)nullptr, (float*)nullptr)
    if (wpe == 2) { if (p->noise) WG_FLOW_ENV(true, 2); else WG_FLOW_ENV(false, 2); }
    else { if (p->noise) WG_FLOW_ENV(true, 1); else WG_FLOW_ENV(false, 1); }
#undef WG_FLOW_ENV
}

// step() as one launch (wg_api.hip: launch_step, handles with FlowP::env_fused)
extern "C" void wg_launch_step_env(const FlowP* p, const FlowPtrs* d, const WgParams* gp, const WgPtrs* gd, const float* actions,
                                   float* obs, float* reward, uint8_t* trunc, float* final_obs, hipStream_t st) {
    const int grid = p->B, wpe = p->env_wpe == 2 ? 2 : 1;
    const size_t lds = (size_t)p->env_lds * wpe;
#define WG_STEP_ENV(NZ, G, W) hipLaunchKernelGGL((k_flow_env<NZ, G, W>), dim3(grid), dim3(64 * W), lds, st, *p, *d, (int)WG_MODE_STEP, actions, \
                                                 (const uint8_t*)nullptr, 0, *gp, *gd, obs, reward, trunc, final_obs)
#define WG_STEP_ENV_S(NZ, G, SP) hipLaunchKernelGGL((k_flow_env<NZ, G, 2, SP>), dim3(grid), dim3(64 * (2 + SP)), (size_t)p->env_lds * (2 + SP) + LEAN_PRE_BYTES, st, \
                                                    *p, *d, (int)WG_MODE_STEP, actions, (const uint8_t*)nullptr, 0, *gp, *gd, obs, reward, trunc, final_obs)
#define WG_STEP_ENV_W(NZ, G) do { if (wpe == 2 && p->env_split == 2) WG_STEP_ENV_S(NZ, G, 2); else if (wpe == 2 && p->env_split == 1) WG_STEP_ENV_S(NZ, G, 1); \
                                  else if (wpe == 2) WG_STEP_ENV(NZ, G, 2); else WG_STEP_ENV(NZ, G, 1); } while (0)
    if (gd->multi_out) { if (p->noise) WG_STEP_ENV_W(true, 2); else WG_STEP_ENV_W(false, 2); }
    else { if (p->noise) WG_STEP_ENV_W(true, 1); else WG_STEP_ENV_W(false, 1); }
#undef WG_STEP_ENV_W
#undef WG_STEP_ENV_S
#undef WG_STEP_ENV
}
