// wg_envb.hip — k_flow_envb: the one-launch step kernel for FROZEN-BOX turbulent inflow (BASELINE cfg5; gfx950, wave64).
//
// The k_flow_env pattern (wg_env.hip) for turbtype Mann*: the wave owns the env — or one of its two contexts (WPE 2) — so the
// env's glue runs as the tail of its flow step (step() is ONE launch), the slot-uniform state lives in LDS records, the
// parameter blocks are read through the kernarg pointer where they are used, and — what the per-slot kernel
// (k_flow<64, BOX>: 253 VGPRs, 2 waves per SIMD, 210 spilled SGPRs) could not do — the agent farm and the baseline farm of an
// episode are served by the SAME wave instructions wherever they read the turbulence boxes:
//   * lane g = turbine * NS + slot (turbine-major): a wave's farm slots are interleaved lane by lane, so the rows of the
//     rotor-point phase come out as (turbine 0: agent, baseline), (turbine 1: agent, baseline), ... — both farms look the
//     ambient and the wake-added field up at the same rotor (only cos(yaw) moves the lateral offsets), in the same gather
//     instruction, whose lanes then share their cache lines (the fine box is 2.9 lines of 128 B per point: 96 useful bytes);
//   * the particle pass gives each of the two farms half of the wave: lanes 0-31 = 32 consecutive ring slots of the agent
//     farm, lanes 32-63 = the same ring slots of the baseline farm — the two chains meander through the same cells of the
//     block-averaged box, and consecutive lanes are consecutive particles of a chain (0.2 D apart along x: they share lines too).
// Deficit phase: sample-major, like the round-1 large-farm variant (flow_step, RES = false) — candidate pairs are compacted
// into a list (target-major = lane order), ONE lane per candidate gathers its two bracketing particles and stages the
// interpolated wake (yc, zc, 1 / 2 sigma^2, amplitude | added TI); then one lane per (target, rotor point) looks up the
// ambient and the wake-added field at ITS point and sums the staged wakes of its target in ascending source order.  No
// per-point field staging in LDS, one pass over the rotor points instead of two.
//
// State layout: exactly k_flow<64, BOX, RES>'s (SoA py / pz / vlp / wlp / rec_a / rec_b over compact rings, slot-major
// turbine arrays) — the two kernels are interchangeable launch by launch on one handle (tests/test_gpu_variant_ab.py); values
// agree to float rounding (different summation order over the rotor points), both are held to the oracle's bars.
// Order of a flow step (k_flow's for turbulent inflow): clocks -> emission records -> particle pass (meandering through the
// low-pass filtered box inflow, release) -> [stores visible] -> candidates -> staged wakes -> rotor points -> tail.
// Replaces DWMFlowSimulation.step() + rotor_avg_windspeed + power() + BasicControllers + _take_measurements +
// farm_mes.add_measurements for Mann-box inflow (Wind_Farm_Env.py:480-495, 611-659, 822-864, 943-979;
// BasicControllers.py:10-73; MesClass.py:568-591), and — with the glue tail — the whole of step() (:920-1034).
#include <hip/hip_runtime.h>

#include "wg_env_common.h"
#include "wg_box_dev.h"

#ifndef WG_ENVB_U
#define WG_ENVB_U 2          // ring slots per lane and trip of the particle pass (their 6 streamed words and 4 box cells in flight together)
#endif
#ifndef WG_ENVB_ABLATE
#define WG_ENVB_ABLATE 0     // profiling builds only (results are wrong): 1 = no box cells for the particles, 2 = no ambient lookups at the
                             // rotor points, 4 = no wake-added lookups, 8 = no bracket gathers of the candidates, 16 = no particle pass
#endif
#ifndef WG_ENVB_WAVES
#define WG_ENVB_WAVES 4      // <= 128 VGPRs
#endif

struct __attribute__((aligned(16))) EnvbSlotLds {
    int dev_rem, fill_rem, sub, budget;          // schedule (one 16-byte read)
    double s_new;                                // clock of the current flow step
    int n_emit, n_valid, new_valid;
    unsigned n_emitted;
    float s_off_f, ti_pow, sig, alpha;           // sig = TI * U (what the unit-variance box is scaled to), low-pass coefficient
    double s_off, ws;                            // the slot's persistent clock and its context's wind
    float ws_f, ti_f, wd_env, base_acc;
    int n_pushed, pend_farm_n, pend_base_n, n_flow;
    double time, ox, oy, xshift;                 // fs.time (advanced every flow step), the episode's offset into the box, ox - ws * time
                                                 // of the PRE-step clock (what the particles are looked up with)
    long long box_cell0, cbox_cell0;             // first cell of the episode's box of the pool (fine / block-averaged copy)
    int c_head;                                  // cold words, parked here between prologue and epilogue
    unsigned c_part, c_flow, c_istep, c_tag, n_emitted0;
    float out_pw;                                // the env step's farm power (what the glue reads as step_farm_pow / step_base_pow)
    unsigned c_add;
    int bg_init, add_acc, stepping, L;           // L: ring slots of the farm (roff[N] of its context)
};
static_assert(sizeof(EnvbSlotLds) == 208, "EnvbSlotLds layout (EnvbLds below)");

// LDS layout of a wave's region: compile-time offsets, the per-lane arrays sized for NLP lanes (64; 16 with four waves per env:
// a wave then serves ONE farm slot of at most 16 turbines), indexed by lane g = t * NS + k
//   XR / YR  double[NLP]   position        SRC4 float4[NLP] (x, y as floats, bk, be): candidate pass
//   REC4     uint4[NLP]    this step's emission record: (rec_a, rec_b, -, bits of cos yaw)
//   RING     int4[NLP]     (roff, rlen, head before, head after this step's release)
//   UVW      float4[NLP]   rotor inflow of the step (u, v, w, ti): written by the rotor-point phase
//   BD       float[NLP]    excursion bound of the chain (raised by the particle pass with LDS atomics)
//   CR       int2[NLP]     (first entry, entries) of the target's slice of the candidate list
//   ROW      u8[NLP]       lane of the i-th stepping (slot, turbine) row
//   SL       EnvbSlotLds[4] | PARK float4[3][NLP] the lane's turbine registers, parked across the particle / rotor phases
//   HDR      int[32]       the env header as the prologue loaded it (the glue tail's copy)
//   STAGE    staged wakes: float4[cap] | added TI float[cap] | candidate list u16[..] | tables (FlowP::env_off_tab)
template <int NLP> struct EnvbLds {
    static constexpr int XR = 0, YR = 8 * NLP, SRC4 = 16 * NLP, REC4 = 32 * NLP, RING = 48 * NLP, UVW = 64 * NLP, BD = 80 * NLP,
                         CR = 84 * NLP, ROW = 92 * NLP, SL = 93 * NLP, PARK = SL + 4 * 208, HDR = PARK + 48 * NLP, STAGE = HDR + 128;
    static_assert(STAGE == WG_ENVB_FIXED_LDS_BYTES(NLP), "keep WG_ENVB_FIXED_LDS_BYTES in sync (wg_flow.h)");
};
// cross-wave flags of a workgroup of four waves (behind the waves' regions): [w] = wave w's flow part is complete and its stores
// have left the wave; [4 + c] = context c's episode set-up (run by its agent-farm wave) is complete
#define WG_ENVB_FLAG_BYTES 64
__device__ __forceinline__ void envb_flag_set(int* const f) {
    __atomic_store_n(f, 1, __ATOMIC_RELAXED);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// (bounded: a protocol error must not hang the GPU — the caller latches WG_STATUS_BIT_STATE on a time-out)
__device__ __forceinline__ bool envb_flag_wait(int* const f) {
    for (int n = 0; n < (1 << 22); ++n) {
        if (__atomic_load_n(f, __ATOMIC_RELAXED) != 0) { asm volatile("" ::: "memory"); return true; }
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}


__device__ __forceinline__ long long envb_uni64(const long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// The 8 corners of a point in a box stored like box_lookup_dims reads it (wg_box_dev.h: 4 x 4 x 4 bricks when every dimension is
// a multiple of 4) as 32-bit CELL offsets from the box's first cell + the trilinear weights — same cells, same weights.  The
// kernel adds them to a base pointer that is wave-uniform wherever the wave serves one episode: the loads then carry a scalar
// base and ONE address register each (8 x 64-bit addresses per lookup were 16 VGPRs, 32 with the wake-added box in flight too).
// A box of the pool has fewer than 2^28 cells (wg_set_turbulence_boxes falls back to k_flow otherwise).
template <bool POW2>
__device__ __forceinline__ void envb_box_corners(const int bnx, const int bny, const int bnz, const double inv_dx, const double inv_dy,
                                                 const double inv_dz, const double x, const double y, const double z, unsigned (&o)[8],
                                                 float& tx, float& ty, float& tz) {
    const double fx = x * inv_dx, fy = y * inv_dy, fz = z * inv_dz;
    const double ix = floor(fx), iy = floor(fy), iz = floor(fz);
    tx = (float)(fx - ix); ty = (float)(fy - iy); tz = (float)(fz - iz);
    int i0, j0, k0, i1, j1, k1;
    if (POW2) {
        i0 = (int)ix & (bnx - 1); j0 = (int)iy & (bny - 1); k0 = (int)iz & (bnz - 1);
        i1 = (i0 + 1) & (bnx - 1); j1 = (j0 + 1) & (bny - 1); k1 = (k0 + 1) & (bnz - 1);
    } else {
        i0 = (int)ix % bnx; if (i0 < 0) i0 += bnx;
        j0 = (int)iy % bny; if (j0 < 0) j0 += bny;
        k0 = (int)iz % bnz; if (k0 < 0) k0 += bnz;
        i1 = i0 + 1 == bnx ? 0 : i0 + 1; j1 = j0 + 1 == bny ? 0 : j0 + 1; k1 = k0 + 1 == bnz ? 0 : k0 + 1;
    }
    const bool brick = ((bnx | bny | bnz) & 3) == 0;
    const unsigned nbz = (unsigned)(bnz >> 2), nbyz = (unsigned)(bny >> 2) * nbz;
    const unsigned X0 = brick ? (unsigned)(i0 >> 2) * nbyz * 64u + (unsigned)(i0 & 3) * 16u : (unsigned)i0 * (unsigned)bny * (unsigned)bnz;
    const unsigned X1 = brick ? (unsigned)(i1 >> 2) * nbyz * 64u + (unsigned)(i1 & 3) * 16u : (unsigned)i1 * (unsigned)bny * (unsigned)bnz;
    const unsigned Y0 = brick ? (unsigned)(j0 >> 2) * nbz * 64u + (unsigned)(j0 & 3) * 4u : (unsigned)j0 * (unsigned)bnz;
    const unsigned Y1 = brick ? (unsigned)(j1 >> 2) * nbz * 64u + (unsigned)(j1 & 3) * 4u : (unsigned)j1 * (unsigned)bnz;
    const unsigned Z0 = brick ? (unsigned)(k0 >> 2) * 64u + (unsigned)(k0 & 3) : (unsigned)k0;
    const unsigned Z1 = brick ? (unsigned)(k1 >> 2) * 64u + (unsigned)(k1 & 3) : (unsigned)k1;
    o[0] = X0 + Y0 + Z0; o[1] = X1 + Y0 + Z0; o[2] = X0 + Y1 + Z0; o[3] = X1 + Y1 + Z0;
    o[4] = X0 + Y0 + Z1; o[5] = X1 + Y0 + Z1; o[6] = X0 + Y1 + Z1; o[7] = X1 + Y1 + Z1;
}
// Stencil records (FlowPtrs::box8): the record of a point = its cell origin, wrapped periodically — in double, exact for any
// dimension: r = i - floor(i / n) n with one correction step each way (i / n is taken as i * (1 / n), which can be off by one
// at multiples of n).  Same cells and weights as envb_box_corners; the 8 corners are the record's 8 consecutive float4.
__device__ __forceinline__ unsigned envb_wrap(const double i, const int n, const double inv_n) {
    double r = i - floor(i * inv_n) * (double)n;
    if (r < 0.0) r += (double)n;
    if (r >= (double)n) r -= (double)n;
    return (unsigned)(int)r;
}
__device__ __forceinline__ unsigned envb_stencil_record(const int bnx, const int bny, const int bnz, const double inv_dx, const double inv_dy,
                                                        const double inv_dz, const double inx, const double iny, const double inz,
                                                        const double x, const double y, const double z, float& tx, float& ty, float& tz) {
    const double fx = x * inv_dx, fy = y * inv_dy, fz = z * inv_dz;
    const double ix = floor(fx), iy = floor(fy), iz = floor(fz);
    tx = (float)(fx - ix); ty = (float)(fy - iy); tz = (float)(fz - iz);
    const unsigned i0 = envb_wrap(ix, bnx, inx), j0 = envb_wrap(iy, bny, iny), k0 = envb_wrap(iz, bnz, inz);
    return (i0 * (unsigned)bny + j0) * (unsigned)bnz + k0;
}
// (v[0..7] = v000 v100 v010 v110 v001 v101 v011 v111; the association of box_lookup_dims / the oracle: x, then y, then z)
__device__ __forceinline__ void envb_tri3(const float4 (&v)[8], const float tx, const float ty, const float tz, float* __restrict__ out) {
#define WG_TRI(f)                                                         \
    ([&]() {                                                              \
        const float c00 = v[0].f + tx * (v[1].f - v[0].f);                \
        const float c10 = v[2].f + tx * (v[3].f - v[2].f);                \
        const float c01 = v[4].f + tx * (v[5].f - v[4].f);                \
        const float c11 = v[6].f + tx * (v[7].f - v[6].f);                \
        const float d0 = c00 + ty * (c10 - c00);                          \
        const float d1 = c01 + ty * (c11 - c01);                          \
        return d0 + tz * (d1 - d0);                                       \
    }())
    out[0] = WG_TRI(x); out[1] = WG_TRI(y); out[2] = WG_TRI(z);
#undef WG_TRI
}
// the 4 cells (z pairs packed) of a point in the block-averaged meandering box (cbox_lookup_vw_dims, wg_box_dev.h): offsets + weights
template <bool POW2>
__device__ __forceinline__ void envb_cbox_corners(const int bnx, const int bny, const int bnz, const double inv_bdx, const double inv_bdy,
                                                  const double inv_bdz, const double x, const double y, const double z, unsigned (&o)[4],
                                                  float& tx, float& ty, float& tz) {
    const double fx = (x * inv_bdx - 1.5) * 0.25, fy = (y * inv_bdy - 1.5) * 0.25, fz = (z * inv_bdz - 1.5) * 0.25;
    const double ix = floor(fx), iy = floor(fy), iz = floor(fz);
    tx = (float)(fx - ix); ty = (float)(fy - iy); tz = (float)(fz - iz);
    int i0, j0, k0, i1, j1;
    if (POW2) {
        i0 = (int)ix & (bnx - 1); j0 = (int)iy & (bny - 1); k0 = (int)iz & (bnz - 1);
        i1 = (i0 + 1) & (bnx - 1); j1 = (j0 + 1) & (bny - 1);
    } else {
        i0 = (int)ix % bnx; if (i0 < 0) i0 += bnx;
        j0 = (int)iy % bny; if (j0 < 0) j0 += bny;
        k0 = (int)iz % bnz; if (k0 < 0) k0 += bnz;
        i1 = i0 + 1 == bnx ? 0 : i0 + 1; j1 = j0 + 1 == bny ? 0 : j0 + 1;
    }
    const unsigned r0 = ((unsigned)j0 * (unsigned)bnz + (unsigned)k0) * (unsigned)bnx, r1 = ((unsigned)j1 * (unsigned)bnz + (unsigned)k0) * (unsigned)bnx;
    o[0] = r0 + (unsigned)i0; o[1] = r0 + (unsigned)i1; o[2] = r1 + (unsigned)i0; o[3] = r1 + (unsigned)i1;
}
__device__ __forceinline__ void envb_tri2(const float4 (&c)[4], const float tx, const float ty, const float tz, float& fv, float& fw) {
#define WG_TRI2(lo, hi)                                                   \
    ([&]() {                                                              \
        const float e00 = c[0].lo + tx * (c[1].lo - c[0].lo);             \
        const float e10 = c[2].lo + tx * (c[3].lo - c[2].lo);             \
        const float e01 = c[0].hi + tx * (c[1].hi - c[0].hi);             \
        const float e11 = c[2].hi + tx * (c[3].hi - c[2].hi);             \
        const float d0 = e00 + ty * (e10 - e00);                          \
        const float d1 = e01 + ty * (e11 - e01);                          \
        return d0 + tz * (d1 - d0);                                       \
    }())
    fv = WG_TRI2(x, z); fw = WG_TRI2(y, w);
#undef WG_TRI2
}

// sum of a per-lane value over the lanes of the lane's OWN slot (lanes g = t * NS + k with the same k), all slots at once;
// lanes beyond the wave's turbines must pass 0.  Valid in every lane.
__device__ __forceinline__ float envb_slot_sums(float v, const int NS) {
    for (int o = NS; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool NOISE, int WPE, int GLUE>
__device__ __forceinline__ void envb_flow(char* const smem, char* const wg_shared, const int wv, const int mode, const float* __restrict__ actions,
                                          const uint8_t* __restrict__ mask, const int chunk, EnvFlowOut& out) {
    const int tid = threadIdx.x & 63, e = blockIdx.x;
    int N, F, NS, NL, nsh;
    {
        const KArgsPtr ka = wg_cold_args();
        N = ka->p.N; F = ka->p.F;
        NS = WPE == 4 ? 1 : (WPE == 2 ? F : 2 * F); NL = NS * N;      // slots / lanes served by THIS wave
        nsh = NS == 4 ? 2 : (NS == 2 ? 1 : 0);
    }
    const int kbase = WPE == 4 ? wv : (WPE == 2 ? wv * F : 0);      // the wave's first slot of the env
    const bool valid = tid < NL;
    const int g = valid ? tid : 0;
    const int t = g >> nsh, k = g & (NS - 1);               // turbine, slot of the wave (WPE 1: of the env = ctx * F + farm)
    // (four waves per env: wave w = farm slot w = context w >> 1, farm w & 1 — the host selects it for F = 2 only)
    const int c = WPE == 4 ? (wv >> 1) : (WPE == 2 ? wv : (F == 2 ? (k >> 1) : k)), farm = WPE == 4 ? (wv & 1) : (F == 2 ? (k & 1) : 0);
    constexpr int NLP = WPE == 4 ? 16 : 64;
    typedef EnvbLds<NLP> LO;

    double* const Lxr = reinterpret_cast<double*>(smem + LO::XR);
    double* const Lyr = reinterpret_cast<double*>(smem + LO::YR);
    float4* const Lsrc4 = reinterpret_cast<float4*>(smem + LO::SRC4);
    uint4* const Lrec4 = reinterpret_cast<uint4*>(smem + LO::REC4);
    int4* const Lring = reinterpret_cast<int4*>(smem + LO::RING);
    float4* const Luvw = reinterpret_cast<float4*>(smem + LO::UVW);
    float* const Lbd = reinterpret_cast<float*>(smem + LO::BD);
    int2* const Lcr = reinterpret_cast<int2*>(smem + LO::CR);
    unsigned char* const Lrow = reinterpret_cast<unsigned char*>(smem + LO::ROW);
    EnvbSlotLds* const SL = reinterpret_cast<EnvbSlotLds*>(smem + LO::SL);
    float4* const Lpark = reinterpret_cast<float4*>(smem + LO::PARK);
    EnvbSlotLds& my = SL[k];

    // ---- prologue: every independent global load up front (one exposed round trip) ----------------------------------
    typedef const __attribute__((address_space(4))) WgEnv* CEnvPtr;
    int env_live;
    bool role_live = false, role_dev = false, defer_init = false;
    float yaw, tu, tv, tw, tti, oyaw, mvl_bits;
    {
        const KArgsPtr k0 = wg_cold_args();
        // (the env header as one coalesced load from the global address space: see env_flow, wg_env.hip)
        static_assert(sizeof(WgEnv) == 128, "the env header is loaded as 32 words");
        const int hw = reinterpret_cast<const int*>(k0->d.env + e)[tid & 31];
        if (GLUE != 0 && tid < 32) reinterpret_cast<int*>(smem + LO::HDR)[tid] = hw;     // (handed to the glue tail: no second load)
        out.bg_init_pending = 0; out.rounds = 0; out.first_obs = 0;
        const bool use_mask = mode == WG_MODE_RESET && mask != nullptr;
        const uint8_t mask_byte = *(use_mask ? mask + e : reinterpret_cast<const uint8_t*>(k0->d.env + e));
        const bool masked_out = use_mask && mask_byte == 0;
        const unsigned ctx_id = (unsigned)(e * 2 + c), slot_id = (unsigned)(e * 2 * F + kbase + k);
        const unsigned tb = slot_id * (unsigned)N + (unsigned)t;
        const unsigned tcx = ctx_id * (unsigned)N + (unsigned)t;

        int dev_rem, fill_rem, n_pushed, pend_farm_n, pend_base_n, init_pending, time_max_c, n_valid, c_head, box_id;
        unsigned n_emitted, c_part, c_flow, c_istep, c_tag, c_add;
        double s_off, ws, ti_d, l_xr, l_yr, c_time, box_ox, box_oy;
        float wd_env, l_yaw, l_u, l_v, l_w, l_ti, l_act;
        float4 l_bnd;
        int l_roff, l_rnext, l_rtot;
        auto load_state = [&]() __attribute__((always_inline)) {
            const KArgsPtr kl = wg_cold_args();
            const WgSlot& slot = kl->d.slot[slot_id];
            const WgCtx& cx = kl->d.ctx[ctx_id];
            dev_rem = slot.dev_remaining; fill_rem = slot.fill_remaining;
            s_off = slot.s_off; c_time = slot.time; c_head = slot.head; n_valid = slot.n_valid; c_istep = slot.istep;
            n_emitted = slot.n_emitted; c_part = slot.part_count; c_flow = slot.flow_count; c_add = slot.add_count;
            ws = cx.ws; ti_d = cx.ti; wd_env = (float)cx.wd;
            n_pushed = cx.n_pushed; pend_farm_n = cx.pend_farm_n; pend_base_n = cx.pend_base_n;
            c_tag = (unsigned)cx.episode_tag; init_pending = cx.init_pending; time_max_c = cx.time_max;
            box_ox = cx.box_ox; box_oy = cx.box_oy; box_id = cx.box_id;
            const int* ro = kl->d.roff + ctx_id * (unsigned)(N + 1);
            l_roff = ro[(unsigned)t]; l_rnext = ro[(unsigned)t + 1u]; l_rtot = ro[(unsigned)N];
            l_xr = kl->d.xr[tcx]; l_yr = kl->d.yr[tcx];
            l_yaw = kl->d.yaw[tb]; l_u = kl->d.u[tb]; l_v = kl->d.v[tb]; l_w = kl->d.w[tb]; l_ti = kl->d.ti_loc[tb];
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
        if (WPE >= 2) {      // (every wave holds its copy of the header before any can rewrite it: see env_flow)
            asm volatile("" : "+s"(env_live), "+s"(env_done), "+s"(env_shadow_iters), "+s"(env_steps_done), "+s"(env_timestep), "+s"(env_time_max_live) : : "memory");
            // (lgkmcnt: with four waves, wave 0's zeroing of the workgroup's flags has landed before anyone can set one)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        out.env_live = env_live;
        out.truncates = env_timestep >= env_time_max_live;
        out.steps_done = env_steps_done; out.time_max_live = env_time_max_live;

        // ---- roles (as env_flow: running episode = one env step, background episode = its share of development) -------
        const bool is_live_c = (c == env_live);
        const int autoreset = k0->p.autoreset;
        const int bg_lane = WPE >= 2 ? 0 : (env_live ^ 1) * F;          // lane of the background context's agent farm, turbine 0
        int budget = 0;
        if (mode == WG_MODE_STEP) {
            role_live = is_live_c && !env_done;
            role_dev = !is_live_c && autoreset != 0;
            const int bg_pending = WPE >= 2 ? (is_live_c ? 0 : __shfl(init_pending, 0, 64)) : __shfl(init_pending, bg_lane, 64);
            out.bg_init_pending = autoreset && bg_pending;
            if (autoreset && bg_pending && WPE == 1 && (WG_ENV_DEFER_INIT != 0) && !out.truncates) {
                defer_init = true;      // rare path: the retired context's next episode is set up AFTER this wave's step (env_flow)
                budget = 0;
            } else if (autoreset && bg_pending) {
                const KArgsPtr ki = wg_cold_args();
                const WgParams& gp = *ki->d.gp;
                if (WPE == 4) {
                    // the context's agent-farm wave sets BOTH farms up; the baseline farm's wave waits for it (same CU: the
                    // stores need no cache maintenance to be seen, only to have left the wave)
                    int* const fl = reinterpret_cast<int*>(wg_shared) + 4 + c;
                    if (farm == 0) { env_init_episode<GLUE != 0>(e, env_live ^ 1, tid); full_barrier<64>(); if (tid == 0) envb_flag_set(fl); }
                    else if (!envb_flag_wait(fl)) atomicOr(ki->d.status, WG_STATUS_BIT_STATE);
                } else {
                    env_init_episode<GLUE != 0>(e, env_live ^ 1, tid);
                    full_barrier<64>();
                }
                load_state();
                const int tm = WPE >= 2 ? ki->d.ctx[e * 2 + env_live].time_max : __shfl(time_max_c, env_live * F, 64);
                const int inc = 1 + (gp.extra_inc ? 1 : 0);
                const long total = (long)((tm + inc - 1) / inc) + 1;
                budget = wg_shadow_share(dev_rem + gp.K * fill_rem, total - env_steps_done, env_steps_done, e, farm ? 0x80000000u : 0u);
            } else {
                // the background episode's share of this step, planned per farm half a dither period apart (env_flow)
                const int inc = k0->p.env_inc;
                const long total = (long)((env_time_max_live + inc - 1) / inc) + 1;
                budget = role_dev ? wg_shadow_share(dev_rem + k0->p.K * fill_rem, total - env_steps_done, env_steps_done, e, farm ? 0x80000000u : 0u) : 0;
            }
        } else {
            role_dev = is_live_c && !masked_out;
            budget = chunk;
        }
        (void)env_shadow_iters;
        if (WPE >= 2 && valid && t == 0) {      // (what the other waves' glue / plan reads of this one, also if it has nothing to do)
            my.dev_rem = dev_rem; my.fill_rem = fill_rem; my.bg_init = out.bg_init_pending; my.n_flow = 0; my.out_pw = 0.f;
        }
        {   // nothing to do for the whole wave (masked out in RESET mode, finished env without autoreset, idle background)
            const bool any_work = valid && (role_live || (role_dev && budget > 0 && (dev_rem > 0 || fill_rem > 0)));
            if (!__ballot(any_work)) {
                if (WPE >= 2 && (WPE == 2 || farm == 0) && (WG_ENV_FIRST_OBS_LATER != 0) && mode == WG_MODE_STEP && !is_live_c && autoreset) {
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

        // ---- state into LDS ---------------------------------------------------------------------------------------------
        const KArgsPtr k1 = wg_cold_args();
        const int rlen = l_rnext - l_roff;
        const int head = n_emitted == 0u ? rlen - 1 : fast_mod((int)(n_emitted - 1u), rlen, __builtin_amdgcn_rcpf((float)rlen));
        if (valid) {
            Lxr[g] = l_xr; Lyr[g] = l_yr;
            Lsrc4[g] = make_float4((float)l_xr, (float)l_yr, l_bnd.y, l_bnd.z);
            Lbd[g] = l_bnd.x;
            Lring[g] = make_int4(l_roff, rlen, head, head);
            Lrec4[g] = make_uint4(0u, 0u, 0u, __float_as_uint(1.f));
            Luvw[g] = make_float4(l_u, l_v, l_w, l_ti);
            if (t == 0) {
                const float ti_f = (float)ti_d;
                my.dev_rem = dev_rem; my.fill_rem = fill_rem; my.sub = 0; my.budget = budget;
                my.n_valid = n_valid; my.n_emitted = n_emitted; my.n_emitted0 = n_emitted;
                my.ti_pow = fast_pow(ti_f, k1->p.tic);
                my.sig = (float)(ti_d * ws);
                my.alpha = (float)(1.0 - exp(-2.0 * WG_PI_D * (ws / (k1->p.fc_scale * k1->p.D_d)) * k1->p.dt_d));
                my.s_off = s_off; my.ws = ws; my.ws_f = (float)ws; my.ti_f = ti_f; my.wd_env = wd_env; my.base_acc = 0.f;
                my.n_pushed = n_pushed; my.pend_farm_n = pend_farm_n; my.pend_base_n = pend_base_n; my.n_flow = 0;
                my.time = c_time; my.ox = box_ox; my.oy = box_oy;
                my.box_cell0 = (long long)box_id * k1->p.box_cells; my.cbox_cell0 = (long long)box_id * k1->p.cbox_cells;
                my.bg_init = out.bg_init_pending; my.add_acc = 0; my.stepping = 0; my.L = l_rtot;
                my.c_head = c_head; my.c_part = c_part; my.c_flow = c_flow; my.c_istep = c_istep; my.c_tag = c_tag; my.c_add = c_add;
            }
        }
        // tables: power | ct | rotor point offsets (dy | dz)
        float* const tabp = reinterpret_cast<float*>(smem + k1->p.env_off_tab);
        float* const tabct = tabp + n_tab;
        float* const rdy = tabct + n_tab;
        float* const rdz = rdy + S;
        if (tid < n_tab) { tabp[tid] = pf_tp; tabct[tid] = pf_tc; }
        if (tid < S) { rdy[tid] = pf_dy; rdz[tid] = pf_dz; }
        for (int i = tid + 64; i < n_tab; i += 64) { tabp[i] = k1->d.tab_power[i]; tabct[i] = k1->d.tab_ct[i]; }
        for (int i = tid + 64; i < S; i += 64) { rdy[i] = k1->d.rotor_dy[i]; rdz[i] = k1->d.rotor_dz[i]; }

        yaw = l_yaw; tu = l_u; tv = l_v; tw = l_w; tti = l_ti; oyaw = l_yaw; mvl_bits = l_bnd.w;
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
    float tpow = 0.f, tct = 0.f, cg = 1.f, sg = 0.f;
    float sws = 0.f, swd = 0.f, syaw = 0.f, sp_ = 0.f;
    int part_acc = 0;
    bool stepped = false;
    lds_barrier<64>();

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
            // (a further flow step of the launch reads what the previous one's particle pass stored)
            if (round > 0) full_barrier<64>();

            // BasicControllers.local_yaw_controller / global_yaw_controller (BasicControllers.py:10-73): the running episode's
            // baseline farm, every sim sub-step, not clipped
            if (role_live && farm == 1 && stepping) {
                const float ystep = kr->p.yaw_step;
                if (kr->p.base_controller == WG_CTRL_LOCAL) {
                    const float wdir = atanf(tv / tu) * WG_RAD2DEG_F;
                    const float off = wdir - yaw;
                    const float sgn = (float)((off > 0.f) - (off < 0.f));
                    yaw = yaw + sgn * fminf(fabsf(off), ystep);
                } else {
                    const float sgn = (float)((yaw > 0.f) - (yaw < 0.f));
                    yaw = yaw - sgn * fminf(fabsf(yaw), ystep);
                }
            }
            // (0) the slot's clock: travel of its chains over this step, particles released (lane t = 0 publishes)
            if (valid && t == 0) {
                my.stepping = stepping ? 1 : 0;
                if (stepping) {
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
                    my.xshift = my.ox - my.ws * my.time;
                }
            }
        }
        lds_barrier<64>();

        // (1) emission records of this step, sin / cos of the yaw
        if (stepping) {
            const KArgsPtr kq = wg_cold_args();
            const int n_tab = kq->p.n_tab;
            const float* const tabct = reinterpret_cast<const float*>(smem + kq->p.env_off_tab) + n_tab;
            const float gy = yaw * WG_DEG2RAD_F;
            sg = __sinf(gy);
            cg = __cosf(gy);
            const float wsn = fmaxf(tu * cg + tv * sg, 0.0f);
            const float ctx = fminf(fmaxf(env_tab(tabct, kq->p.tab_x0, kq->p.tab_inv_dx, n_tab, wsn) * cg * cg, 0.0f), 0.96f);
            const float rq = __builtin_amdgcn_sqrtf(1.0f - ctx);
            const float beta = 0.5f * (1.0f + rq) * __builtin_amdgcn_rcpf(rq);
            const float rk = kq->p.ka * tti + kq->p.kb;
            const float reps = kq->p.eps0 * __builtin_amdgcn_sqrtf(beta);
            const float rhv = -kq->p.hill * sg * tu;
            // the packed record saturates outside [0, WG_K_MAX] x [-WG_HV_MAX, WG_HV_MAX]: never silently (wg_check reports it)
            if (rk > WG_K_MAX || fabsf(rhv) > WG_HV_MAX) atomicOr(kq->d.status, WG_STATUS_BIT_RANGE);
            const float ue_max = kq->p.ue_scale * my.ws_f;        // (turbulent inflow: 2 U)
            if (tu > ue_max) atomicOr(kq->d.status, WG_STATUS_BIT_RANGE);
            const unsigned na_ = pack_a(ctx, rk), nb_ = pack_b(tu * __builtin_amdgcn_rcpf(ue_max), rhv);
            float4 s4 = Lsrc4[g];
            s4.z = fmaxf(s4.z, rk + WG_K_MAX / 65535.0f);
            s4.w = fmaxf(s4.w, reps + 1.0f / 65535.0f);
            Lsrc4[g] = s4;
            const int n_emit = my.n_emit;
            if (n_emit > 0 && rec_moves(nb_)) mvl_bits = __uint_as_float(my.n_emitted + (unsigned)n_emit);
            Lrec4[g] = make_uint4(na_, nb_, __float_as_uint(tu), __float_as_uint(cg));
            int4 rg = Lring[g];
            int hn = rg.z + n_emit;
            if (n_emit >= rg.y) hn %= rg.y; else if (hn >= rg.y) hn -= rg.y;
            Lring[g].w = hn;
            part_acc += min(my.new_valid, rg.y);          // roofline accounting: particles that can still reach a rotor
        }
        // Register discipline: nothing of the lane's turbine is needed until the tail — parked in LDS across the two phases
        // that want the registers for loads in flight (particle pass: streamed words + box cells of two trips; rotor points:
        // 16 box cells per lane).  The rotor inflow (u, v, w, ti) lives in Luvw anyway.
        if (valid) {
            Lpark[g] = make_float4(yaw, oyaw, mvl_bits, sg);
            Lpark[NLP + g] = make_float4(sws, swd, syaw, sp_);
            Lpark[2 * NLP + g] = make_float4(tpow, tct, __int_as_float(part_acc), cg);
        }
        lds_barrier<64>();

        // (2) particle pass: every valid particle of the stepping farms meanders with the low-pass filtered transverse inflow at
        // its position (the block-averaged box; the fine one when the box has no such copy), the new particles are released.
        // A context whose two farms both step gives each HALF of the wave, the same 32 ring slots at a time: the two farms'
        // gathers go out in one instruction and share their lines.
        {
            const KArgsPtr kp = wg_cold_args();
            const unsigned pstride = (unsigned)kp->p.pstride;
            const float dpart_f = kp->p.dpart_f, inv_D = kp->p.inv_D, dt = kp->p.dt, hub = kp->p.hub, eps0 = kp->p.eps0;
            const int n_ctx_w = WPE >= 2 ? 1 : 2;
            for (int cc = 0; cc < ((WG_ENVB_ABLATE & 16) ? 0 : n_ctx_w); ++cc) {
                const int k0s = WPE >= 2 ? 0 : cc * F;                         // the context's first slot of the wave
                const int st0 = SL[k0s].stepping, st1 = (F == 2 && WPE != 4) ? SL[k0s + 1].stepping : 0;
                if (!(st0 | st1)) continue;
                const bool both = st0 && st1;
                const int lpf = both ? 32 : 64;
                const int kf = both ? k0s + (tid >> 5) : (st0 ? k0s : k0s + 1);      // this lane's slot
                const int l = both ? (tid & 31) : tid;
                const EnvbSlotLds& q = SL[kf];
                const int L = q.L, n_emit = q.n_emit, n_valid = q.n_valid;
                const float sof = q.s_off_f, sig = q.sig, alpha = q.alpha;
                const double xshift = q.xshift, oy = q.oy;
                const unsigned ctxw = (unsigned)(e * 2 + (WPE >= 2 ? c : cc));
                // (particle addresses: the env's block as a wave-uniform base + a 32-bit offset per lane — an env's 2 F slots span
                // less than 4 GB — so every access carries a scalar base and one address register)
                const size_t pbe = (size_t)(e * 2 * F) * pstride;
                const unsigned po = (unsigned)(kbase + kf) * pstride;                  // this lane's farm slot inside the env's block
                float* const py_ = kp->d.py + pbe; float* const pz_ = kp->d.pz + pbe;
                float* const vl_ = kp->d.vlp + pbe; float* const wl_ = kp->d.wlp + pbe;
                unsigned* const ra_ = kp->d.rec_a + pbe; unsigned* const rb_ = kp->d.rec_b + pbe;
                const uint8_t* const own_ = kp->d.qown + (size_t)ctxw * (unsigned)(kp->p.NP >> 2);
                // (the two farms of a context read the same box of the pool: a wave-uniform base pointer)
                const float4* const cbox = kp->d.box4c ? kp->d.box4c + envb_uni64(SL[k0s].cbox_cell0) : nullptr;
                const float4* const fbox = kp->d.box4 + envb_uni64(SL[k0s].box_cell0);
                auto pass = [&](auto coarse_tag, auto pow2_tag) __attribute__((always_inline)) {
                    constexpr bool COARSE = decltype(coarse_tag)::value;
                    constexpr bool POW2 = decltype(pow2_tag)::value;
                    constexpr int U = COARSE ? WG_ENVB_U : 1;      // (boxes without a block-averaged copy: 8 cells per particle, one at a time)
                    constexpr int NCELL = COARSE ? 4 : 8;
                    const KArgsPtr kb = wg_cold_args();
                    const int cnx = kb->p.cnx, cny = kb->p.cny, cnz = kb->p.cnz, bnx = kb->p.bnx, bny = kb->p.bny, bnz = kb->p.bnz;
                    const double ibx = kb->p.inv_bdx, iby = kb->p.inv_bdy, ibz = kb->p.inv_bdz;
                    // software pipeline: the streamed words of the NEXT trip are requested before this trip's box cells are —
                    // a trip exposes one memory round trip (its gathers), not two
                    struct PReq { float py, pz, vl, wl; unsigned ra, rb; int own; };
                    PReq nx[U];
                    auto request = [&](PReq (&r)[U], const int b0) __attribute__((always_inline)) {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int ix = min(b0 + u * lpf + l, L - 1);
                            const unsigned ia = po + (unsigned)ix;
                            r[u].py = py_[ia]; r[u].pz = pz_[ia]; r[u].vl = vl_[ia]; r[u].wl = wl_[ia];
                            r[u].ra = ra_[ia]; r[u].rb = rb_[ia]; r[u].own = own_[(unsigned)ix >> 2];
                        }
                    };
                    request(nx, 0);
                    for (int b0 = 0; b0 < L; b0 += lpf * U) {
                        PReq cu[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) cu[u] = nx[u];
                        request(nx, b0 + lpf * U);
                        float4 cell[U][NCELL];
                        float wx[U], wy[U], wz[U];
                        int jv[U], gv[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int ix = min(b0 + u * lpf + l, L - 1);
                            const int gq = (cu[u].own << nsh) + kf;                   // lane of (turbine, slot)
                            gv[u] = gq;
                            const int4 rg = Lring[gq];
                            int j = rg.z - (ix - rg.x); if (j < 0) j += rg.y;
                            jv[u] = j;
                            const float xrel = sof + (float)j * dpart_f;
                            const double bx = Lxr[gq] + (double)xrel + xshift, by = (double)cu[u].py + oy, bz = (double)cu[u].pz;
                            unsigned oc[NCELL];
                            if constexpr (COARSE) envb_cbox_corners<POW2>(cnx, cny, cnz, ibx, iby, ibz, bx, by, bz, oc, wx[u], wy[u], wz[u]);
                            else envb_box_corners<POW2>(bnx, bny, bnz, ibx, iby, ibz, bx, by, bz, oc, wx[u], wy[u], wz[u]);
                            const float4* const bb = COARSE ? cbox : fbox;
#pragma unroll
                            for (int i = 0; i < NCELL; ++i) cell[u][i] = (WG_ENVB_ABLATE & 1) ? make_float4(wx[u], wy[u], wz[u], (float)oc[i]) : bb[oc[i]];
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int ix = b0 + u * lpf + l;
                            if (ix >= L) continue;
                            const unsigned ia = po + (unsigned)ix;
                            const int j = jv[u], gq = gv[u];
                            const int R = Lring[gq].y;
                            float fv, fw;
                            if constexpr (COARSE) {
                                envb_tri2(cell[u], wx[u], wy[u], wz[u], fv, fw);
                            } else {
                                float f3[3];
                                envb_tri3(cell[u], wx[u], wy[u], wz[u], f3);
                                fv = f3[1]; fw = f3[2];
                            }
                            float pyv = cu[u].py, pzv = cu[u].pz, vlv = cu[u].vl, wlv = cu[u].wl;
                            const unsigned rav = cu[u].ra, rbv = cu[u].rb;
                            if (j < n_valid) {
                                const float xrel = sof + (float)j * dpart_f;
                                const float sp = rec_k(rav) * (xrel * inv_D) + rec_eps(rav, eps0);
                                vlv += alpha * (sig * fv - vlv);
                                wlv += alpha * (sig * fw - wlv);
                                pyv += (rec_hv(rbv) * m0_cfrac(rec_ct(rav), sp) + vlv) * dt;
                                pzv += wlv * dt;
                            }
                            const float y0 = Lsrc4[gq].y;
                            if (R - 1 - j < n_emit) {                                 // (= (r - head - 1) mod R: emission index of this slot)
                                const uint4 rn = Lrec4[gq];
                                pyv = y0; pzv = hub; vlv = 0.f; wlv = 0.f;
                                ra_[ia] = rn.x; rb_[ia] = rn.y;
                            }
                            const float ex = j < n_valid ? fabsf(pyv - y0) + fabsf(pzv - hub) : 0.f;   // (valid particles only)
                            if (ex > Lbd[gq]) atomicMax(reinterpret_cast<int*>(&Lbd[gq]), __float_as_int(ex));   // ex >= 0: int order == float order
                            py_[ia] = pyv; pz_[ia] = pzv; vl_[ia] = vlv; wl_[ia] = wlv;
                        }
                    }
                };
                if (kp->p.coarse) { if (kp->p.cbox_pow2) pass(std::true_type{}, std::true_type{}); else pass(std::true_type{}, std::false_type{}); }
                else if (kp->p.box_pow2) pass(std::false_type{}, std::true_type{});
                else pass(std::false_type{}, std::false_type{});
            }
        }
        // the slot's clock after the release: what the brackets and the rotor lookups of this step see
        if (stepping && t == 0) { my.time += wg_cold_args()->p.dt_d; }
        full_barrier<64>();                  // this wave's particle stores are visible to its own gathers below

        // (3) candidate pass, lane = target: its sources are the N turbines of its own slot, tested against the chains' running
        // bounds (after this step's records and excursions).  Bit s of cmask = source s is a candidate.
        unsigned cmask = 0u;
        {
            const KArgsPtr kp = wg_cold_args();
            const float D = kp->p.D, inv_D = kp->p.inv_D;
            const float lim0 = kp->p.R_rot + 1.0e-3f * D;
            const float xt_f = Lsrc4[g].x, yt_f = Lsrc4[g].y;
#pragma unroll 4
            for (int s2 = 0; s2 < N; ++s2) {
                const int gs = (s2 << nsh) + k;
                const float4 a = Lsrc4[gs];
                const float bdv = Lbd[gs];
                const float dxf = xt_f - a.x;
                const float sig_max = (a.z * (dxf * inv_D) + a.w) * D;
                const float gap = fabsf(yt_f - a.y) - (5.0f * sig_max + bdv);
                const bool cd = (s2 != t) & (dxf >= 0.f) & (gap <= lim0);
                cmask |= cd ? (1u << s2) : 0u;
            }
            if (!stepping) cmask = 0u;
        }
        int cbeg, nc, inc;
        unsigned short* const cl = reinterpret_cast<unsigned short*>(smem + wg_cold_args()->p.envb_off_cl);
        {
            const int cnt = __popc(cmask);
            inc = env_scan(cnt, tid);
            cbeg = inc - cnt;
            nc = __builtin_amdgcn_readlane(inc, 63);
            unsigned m = cmask;
            int o = cbeg;
            while (m) { cl[o++] = (unsigned short)((g << 5) | __builtin_ctz(m)); m &= m - 1u; }
            if (valid) Lcr[g] = make_int2(cbeg, cnt);
            // rows of the rotor-point phase = the stepping (turbine, slot) lanes in lane order
            const unsigned long long rm = __ballot(stepping);
            if (stepping) Lrow[__popcll(rm & ((1ull << tid) - 1ull))] = (unsigned char)g;
        }
        const unsigned long long row_mask = __ballot(stepping);
        const int n_rows = __popcll(row_mask);
        lds_barrier<64>();

        // (4) + (5), in chunks of targets whose candidates fit the staging region (one chunk unless a farm is very dense)
        {
            const KArgsPtr kp = wg_cold_args();
            const int cap = kp->p.env_cap;
            float4* const pp = reinterpret_cast<float4*>(smem + LO::STAGE);
            float* const tiap = reinterpret_cast<float*>(pp + cap);
            const int S = kp->p.S, S_pad = kp->p.S_pad, sshift = kp->p.S_shift;
            const bool ADDED = kp->p.added != 0;
            int g_lo = 0, base = 0;
            while (g_lo < 64) {
                // the largest run of targets from g_lo whose candidates fit (inc is monotone: "fits" holds on a prefix)
                const int g_hi = min(64, (int)__popcll(__ballot(inc - base <= cap)));
                const int top = g_hi >= 64 ? nc : __builtin_amdgcn_readlane(inc, max(g_hi - 1, 0));
                // (4) one lane per candidate: bracket, gathers of the two bracketing particles (post-step state), interpolated wake
                {
                    const unsigned pstride = (unsigned)kp->p.pstride;
                    const double inv_dpart = kp->p.inv_dpart;
                    const float inv_D = kp->p.inv_D, D = kp->p.D, hub = kp->p.hub, R_rot = kp->p.R_rot;
                    const float tia = kp->p.no_ti_fold ? 0.f : kp->p.tia, tib = kp->p.tib, tid_ = kp->p.tid;
                    const size_t pbe = (size_t)(e * 2 * F) * pstride;
                    const float* const py_e = kp->d.py + pbe; const float* const pz_e = kp->d.pz + pbe;
                    const float eps0 = kp->p.eps0;
                    const unsigned* const ra_e = kp->d.rec_a + pbe; const unsigned* const rb_e = kp->d.rec_b + pbe;
                    for (int cb = base; cb < top; cb += 64) {
                        const int cidx = cb + tid;
                        if (cidx < top) {
                            float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
                            float tv_ = 0.f;
                            const unsigned en = cl[cidx];
                            const int gt = (int)(en >> 5);
                            const int kt = gt & (NS - 1);
                            const int gs = ((int)(en & 31u) << nsh) + kt;
                            const EnvbSlotLds& q = SL[kt];
                            const double dx = Lxr[gt] - Lxr[gs];
                            bool ok = dx > 0.0;                       // (the candidate test ran on float positions)
                            const double xi = (dx - q.s_new) * inv_dpart;
                            const double jf = floor(xi);
                            float wgt = (float)(xi - jf);
                            int j = (int)jf;
                            if (j < 0) { j = 0; wgt = 0.f; }
                            ok = ok && (j + 1 <= q.new_valid - 1);    // else: the chain has not reached the target yet
                            if (ok) {
                                const int4 rg = Lring[gs];
                                const int Rs = rg.y;
                                int r0 = rg.w - j; if (r0 < 0) r0 += Rs;
                                int r1 = r0 - 1; if (r1 < 0) r1 += Rs;
                                const unsigned sb = (unsigned)(kbase + kt) * pstride + (unsigned)rg.x;      // (inside the env's block)
                                const unsigned i0 = sb + (unsigned)r0, i1 = sb + (unsigned)r1;
#if WG_ENVB_ABLATE & 8
                                const uint4 rr = Lrec4[gs];
                                const float py0 = Lsrc4[gs].y + 1e-9f * (float)i0, py1 = py0 + 1e-9f * (float)i1, pz0 = hub, pz1 = hub;
                                const unsigned a0 = rr.x, a1 = rr.x, b0_ = rr.y, b1_ = rr.y;
#else
                                const float py0 = py_e[i0], py1 = py_e[i1], pz0 = pz_e[i0], pz1 = pz_e[i1];
                                const unsigned a0 = ra_e[i0], a1 = ra_e[i1], b0_ = rb_e[i0], b1_ = rb_e[i1];
#endif
                                const float ue_max = kp->p.ue_scale * q.ws_f;
                                const float u0 = rec_uf(b0_) * ue_max, u1 = rec_uf(b1_) * ue_max;
                                const float w0 = 1.0f - wgt, w1 = wgt;
                                const float yc = w0 * py0 + w1 * py1;
                                const float zc = w0 * pz0 + w1 * pz1;
                                const float kv = w0 * rec_k(a0) + w1 * rec_k(a1);
                                const float epv = w0 * rec_eps(a0, eps0) + w1 * rec_eps(a1, eps0);
                                const float xd = (float)dx * inv_D;
                                const float sp = kv * xd + epv;
                                const float sig = sp * D;
                                const float yt = Lsrc4[gt].y;
                                const float rc2 = (yt - yc) * (yt - yc) + (hub - zc) * (hub - zc);
                                const float rcut = R_rot + 5.0f * sig;
                                if (rc2 <= rcut * rcut) {
                                    const float ctv = w0 * rec_ct(a0) + w1 * rec_ct(a1);
                                    const float uev = w0 * u0 + w1 * u1;
                                    const float cf = m0_cfrac(ctv, sp);
                                    const float inv2s2 = __builtin_amdgcn_rcpf(2.0f * sig * sig);
                                    // wake-added turbulence (Crespo-Hernandez), Gaussian-weighted at the hub
                                    const float ind = 0.5f * (1.0f - __builtin_amdgcn_sqrtf(1.0f - ctv));
                                    tv_ = tia * fast_pow(ind, tib) * q.ti_pow * fast_pow(fmaxf(xd, 1.0f), tid_) * __expf(-rc2 * inv2s2);
                                    pv = make_float4(yc, zc, inv2s2, uev * cf);
                                }
                            }
                            pp[cidx - base] = pv; tiap[cidx - base] = tv_;
                        }
                    }
                }
                lds_barrier<64>();
                // (5) one lane per (row, rotor point): ambient + wake-added field at the point, the staged wakes of the row's target
                {
                    const int rho_lo = g_lo >= 64 ? n_rows : (int)__popcll(row_mask & ((1ull << g_lo) - 1ull));
                    const int rho_hi = g_hi >= 64 ? n_rows : (int)__popcll(row_mask & ((1ull << g_hi) - 1ull));
                    const int rpp = 64 >> sshift;                 // rows per pass
                    const int s = tid & (S_pad - 1);
                    const float* const rdy = reinterpret_cast<const float*>(smem + kp->p.env_off_tab) + 2 * kp->p.n_tab;
                    const float* const rdz = rdy + S;
                    const float rdy_s = rdy[min(s, S - 1)], rdz_s = rdz[min(s, S - 1)];
                    const float hub = kp->p.hub, inv_S = kp->p.inv_S, km1 = kp->p.km1, km2r = kp->p.km2r;
                    const double hub_d = kp->p.hub_d;
                    // (two waves per env: the wave serves ONE episode — a wave-uniform base pointer of its box)
                    const long long ucell0 = WPE >= 2 ? envb_uni64(SL[0].box_cell0) : 0;
                    auto rows = [&](auto pow2_tag, auto apow2_tag) __attribute__((always_inline)) {
                        constexpr bool POW2 = decltype(pow2_tag)::value;
                        constexpr bool APOW2 = decltype(apow2_tag)::value;
                        for (int r0 = rho_lo; r0 < rho_hi; r0 += rpp) {
                            // (the box geometry — 30 scalars — is fetched through a fresh kernarg pointer every trip: held across the
                            // loop it is what spills)
                            const KArgsPtr kb = wg_cold_args();
                            const float4* const abox = kb->d.abox4;
                            // stencil records: one aligned 128-byte record per point instead of 8 cells of the brick-ordered box
                            const float4* const box8 = kb->d.box8;
                            const float4* const abox8 = kb->d.abox8;
                            const int rho = r0 + (tid >> sshift);
                            const bool live = rho < rho_hi && s < S;
                            const int gt = Lrow[min(rho, max(rho_hi - 1, 0))];
                            const int kt = gt & (NS - 1);
                            const EnvbSlotLds& q = SL[kt];
                            float amb[3] = {0.f, 0.f, 0.f}, g3[3] = {0.f, 0.f, 0.f};
                            float acc = 0.f, accw = 0.f, tia_max = 0.f;
                            bool addl = false;
                            if (live) {
                                const float cgt = __uint_as_float(Lrec4[gt].w);
                                const double xr = Lxr[gt], yr = Lyr[gt];
                                const int2 cr = Lcr[gt];
                                addl = ADDED && cr.y > 0;
                                const double bx = xr - q.ws * q.time + q.ox;
                                const double by = yr + (double)(rdy_s * cgt) + q.oy, bz = hub_d + (double)rdz_s;
                                // the 8 + 8 cells of the point are requested first; the wakes of the row's target are summed while they
                                // are in flight (the sums need nothing of the lookups: the wake-added share is (sum of weights) x field)
                                const long long cell0 = WPE >= 2 ? ucell0 : q.box_cell0;
                                float4 va[8], vb[8];
                                float ax, ay, az, gx = 0.f, gy_ = 0.f, gz = 0.f;
                                if (box8) {
                                    const unsigned ra_ = envb_stencil_record(kb->p.bnx, kb->p.bny, kb->p.bnz, kb->p.inv_bdx, kb->p.inv_bdy, kb->p.inv_bdz,
                                                                             kb->p.inv_bn[0], kb->p.inv_bn[1], kb->p.inv_bn[2], bx, by, bz, ax, ay, az);
                                    const float4* const rec = box8 + cell0 * 8;
#pragma unroll
                                    for (int i = 0; i < 8; ++i) va[i] = (WG_ENVB_ABLATE & 2) ? make_float4(ax, ay, az, (float)ra_) : rec[ra_ * 8u + (unsigned)i];
                                } else {
                                    const float4* const fb = kb->d.box4 + cell0;
                                    unsigned oa[8];
                                    envb_box_corners<POW2>(kb->p.bnx, kb->p.bny, kb->p.bnz, kb->p.inv_bdx, kb->p.inv_bdy, kb->p.inv_bdz, bx, by, bz, oa, ax, ay, az);
#pragma unroll
                                    for (int i = 0; i < 8; ++i) va[i] = (WG_ENVB_ABLATE & 2) ? make_float4(ax, ay, az, (float)oa[i]) : fb[oa[i]];
                                }
                                if (addl) {
                                    if (abox8) {
                                        const unsigned rb_ = envb_stencil_record(kb->p.anx, kb->p.any, kb->p.anz, kb->p.inv_adx, kb->p.inv_ady, kb->p.inv_adz,
                                                                                 kb->p.inv_an[0], kb->p.inv_an[1], kb->p.inv_an[2], bx, by, bz, gx, gy_, gz);
#pragma unroll
                                        for (int i = 0; i < 8; ++i) vb[i] = (WG_ENVB_ABLATE & 4) ? make_float4(gx, gy_, gz, (float)rb_) : abox8[rb_ * 8u + (unsigned)i];
                                    } else {
                                        unsigned ob[8];
                                        envb_box_corners<APOW2>(kb->p.anx, kb->p.any, kb->p.anz, kb->p.inv_adx, kb->p.inv_ady, kb->p.inv_adz, bx, by, bz, ob, gx, gy_, gz);
#pragma unroll
                                        for (int i = 0; i < 8; ++i) vb[i] = (WG_ENVB_ABLATE & 4) ? make_float4(gx, gy_, gz, (float)ob[i]) : abox[ob[i]];
                                    }
                                }
                                const float ys = (float)yr + rdy_s * cgt, zs = hub + rdz_s;
                                for (int cq = cr.x; cq < cr.x + cr.y; ++cq) {       // ascending source order
                                    const float4 pw = pp[cq - base];
                                    tia_max = fmaxf(tia_max, tiap[cq - base]);
                                    const float dy = ys - pw.x, dz = zs - pw.y;
                                    const float r2 = dy * dy + dz * dz;
                                    const float du = pw.w * __expf(-r2 * pw.z);
                                    acc += du;
                                    if (ADDED) accw += du * (km1 + km2r * pw.z * __builtin_amdgcn_sqrtf(r2));
                                }
                                envb_tri3(va, ax, ay, az, amb);
                                if (addl) envb_tri3(vb, gx, gy_, gz, g3);
                            }
                            float a0 = accw * g3[0], a1 = accw * g3[1], a2 = accw * g3[2];
                            if (S_pad == 16) {
                                acc = wg_row_sum(acc); amb[0] = wg_row_sum(amb[0]); amb[1] = wg_row_sum(amb[1]); amb[2] = wg_row_sum(amb[2]);
                                if (ADDED) { a0 = wg_row_sum(a0); a1 = wg_row_sum(a1); a2 = wg_row_sum(a2); }
                            } else {
                                for (int o = S_pad >> 1; o > 0; o >>= 1) {
                                    acc += __shfl_xor(acc, o, 64);
                                    amb[0] += __shfl_xor(amb[0], o, 64); amb[1] += __shfl_xor(amb[1], o, 64); amb[2] += __shfl_xor(amb[2], o, 64);
                                    if (ADDED) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
                                }
                            }
                            if (live && s == 0) {
                                const float sig = q.sig, ti_f = q.ti_f;
                                Luvw[gt] = make_float4(q.ws_f + sig * amb[0] * inv_S - acc * inv_S + a0 * inv_S,
                                                       sig * amb[1] * inv_S + a1 * inv_S, sig * amb[2] * inv_S + a2 * inv_S,
                                                       __builtin_amdgcn_sqrtf(ti_f * ti_f + tia_max * tia_max));
                                if (addl) atomicAdd(&SL[kt].add_acc, S);        // (roofline accounting: 8-corner lookups of the isotropic box)
                            }
                        }
                    };
                    if (kp->p.box_pow2) { if (kp->p.abox_pow2) rows(std::true_type{}, std::true_type{}); else rows(std::true_type{}, std::false_type{}); }
                    else { if (kp->p.abox_pow2) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{}); }
                }
                lds_barrier<64>();           // (the next chunk lands in the same staging words; the tail reads Luvw)
                g_lo = g_hi; base = top;
                if (base >= nc && g_hi >= 64) break;
                if (g_hi >= 64) break;
            }
        }

        // per-turbine tail: power / thrust with the current yaw, WindFarmEnv._take_measurements (Wind_Farm_Env.py:480-495)
        // accumulated over the K sub-steps and, at the end of the env step, farm_mes.add_measurements' ring push (MesClass.py:568-591)
        const bool measuring = stepping && !is_dev;
        const bool unit_end = measuring && (sub + 1 == K);
        const KArgsPtr kc = wg_cold_args();
        const float inv_k = 1.0f / (float)K;
        const unsigned ctx_id = (unsigned)(e * 2 + c);
        {   // the lane's turbine back from LDS; its rotor inflow as the rotor-point phase left it (a resting slot's: unchanged)
            const float4 pa = Lpark[g], pb = Lpark[NLP + g], pc = Lpark[2 * NLP + g], r = Luvw[g];
            yaw = pa.x; oyaw = pa.y; mvl_bits = pa.z; sg = pa.w;
            sws = pb.x; swd = pb.y; syaw = pb.z; sp_ = pb.w;
            tpow = pc.x; tct = pc.y; part_acc = __float_as_int(pc.z); cg = pc.w;
            tu = r.x; tv = r.y; tw = r.z; tti = r.w;
        }
        if (stepping) {
            Lring[g].z = Lring[g].w;              // the step's emissions are in the ring now
            stepped = true;
            const int n_tab = kc->p.n_tab;
            const float tab_x0 = kc->p.tab_x0, tab_inv_dx = kc->p.tab_inv_dx;
            const float* const tabp = reinterpret_cast<const float*>(smem + kc->p.env_off_tab);
            const float wsn = fmaxf(tu * cg + tv * sg, 0.0f);
            tpow = env_tab(tabp, tab_x0, tab_inv_dx, n_tab, wsn);
            tct = env_tab(tabp + n_tab, tab_x0, tab_inv_dx, n_tab, wsn) * cg * cg;
            if (measuring && farm == 0) {
                const float wsm = __builtin_amdgcn_sqrtf(tu * tu + tv * tv + tw * tw);
                const float wdm = atanf(tv / tu) * WG_RAD2DEG_F + my.wd_env;
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
                    sws = val[0]; swd = val[1]; sp_ = val[3];      // staged for the farm-level mean / mean / sum
                } else {
                    sws = val[0]; swd = val[1]; syaw = val[2]; sp_ = val[3];
                }
            }
        }
        // farm-level values of every measuring slot at once
        float a_ws = 0.f, a_wd = 0.f, a_pw = 0.f;
        if (__ballot(measuring)) {
            a_ws = envb_slot_sums((measuring && farm == 0) ? sws : 0.f, NS);
            a_wd = envb_slot_sums((measuring && farm == 0) ? swd : 0.f, NS);
            a_pw = envb_slot_sums(measuring ? (farm == 0 ? sp_ : tpow) : 0.f, NS);
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
    }

    // ---- epilogue: the wave's slots back to memory (the slots that took a step) ---------------------------------------------
    const KArgsPtr ke = wg_cold_args();
    const float part_slot = envb_slot_sums(valid ? (float)part_acc : 0.f, NS);      // (exact: far below 2^24)
    if (valid && stepped) {
        const unsigned tb = (unsigned)(e * 2 * F + kbase + k) * (unsigned)N + (unsigned)t;
        const float4 s4 = Lsrc4[g];
        ke->d.yaw[tb] = yaw; ke->d.u[tb] = tu; ke->d.v[tb] = tv; ke->d.w[tb] = tw;
        ke->d.ti_loc[tb] = tti; ke->d.power[tb] = tpow; ke->d.ct[tb] = tct;
        reinterpret_cast<float4*>(ke->d.bnd)[tb] = make_float4(Lbd[g], s4.z, s4.w, mvl_bits);
        if (role_live && farm == 0) ke->d.old_yaw[(unsigned)(e * N + t)] = oyaw;
        if (t == 0) {
            const int n_flow = my.n_flow, P = ke->p.P;
            const unsigned n_emitted = my.n_emitted;
            int hd = my.c_head + (int)((n_emitted - my.n_emitted0) % (unsigned)P); if (hd >= P) hd -= P;
            WgSlot& slot = ke->d.slot[(unsigned)(e * 2 * F + kbase + k)];
            slot.part_count = my.c_part + (unsigned)part_slot;
            slot.add_count = my.c_add + (unsigned)my.add_acc;
            slot.head = hd; slot.n_valid = my.n_valid; slot.s_off = my.s_off; slot.time = my.time;
            slot.istep = my.c_istep + (unsigned)n_flow; slot.n_emitted = n_emitted;
            slot.dev_remaining = my.dev_rem; slot.fill_remaining = my.fill_rem;
            slot.flow_count = my.c_flow + (unsigned)n_flow;
            WgCtx& cx = ke->d.ctx[(unsigned)(e * 2 + c)];
            if (farm == 0) { cx.n_pushed = my.n_pushed; cx.pend_farm_n = my.pend_farm_n; }
            if (farm == F - 1) cx.pend_base_n = my.pend_base_n;
        }
    }
    // a background episode whose development is complete: window sums + first observation for the swap (see env_flow)
    if (mode == WG_MODE_STEP && (WPE == 1 || (c != env_live && (WPE == 2 || farm == 0))) && !defer_init) {
        const EnvbSlotLds& bs = SL[WPE >= 2 ? 0 : (env_live ^ 1) * F];      // the background context's agent farm
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
        env_init_episode<GLUE != 0>(e, env_live ^ 1, tid);
        full_barrier<64>();
        if (valid && t == 0 && c != env_live) {
            const WgSlot& slot = ke->d.slot[(unsigned)(e * 2 * F + kbase + k)];
            my.dev_rem = slot.dev_remaining; my.fill_rem = slot.fill_remaining;
        }
        lds_barrier<64>();
    }
}

// k_flow_envb<NOISE, 0, WPE>: the flow step alone (RESET-mode development; handles whose glue is a separate launch).
// k_flow_envb<NOISE, 1 / 2, WPE>: step() as ONE launch — the glue (lean_step; 2 = with the per-agent observation buffer) as the
// tail of the flow step, exactly as k_flow_env runs it.
// WPE = waves per env: 1 (one wave serves the env's 2 F slots), 2 (one wave per context), 4 (one wave per FARM SLOT, F = 2, at most
// 16 turbines, up to 1024 envs = the chip's 4096 wave slots): the loops of a farm step — particle trips, rotor-point passes — are
// then spread over four waves instead of two, and the wave that runs the glue (the running episode's agent farm) meets its
// baseline farm's wave through a flag in LDS; the background context's waves meet the same way to plan their next share.
template <bool NOISE, int GLUE, int WPE>
__global__ void __launch_bounds__(64 * WPE, WG_ENVB_WAVES)
k_flow_envb(const FlowP p_, const FlowPtrs d_, const int mode, const float* __restrict__ actions,
            const uint8_t* __restrict__ mask, const int chunk, const WgParams gp_, const WgPtrs gd_,
            float* __restrict__ obs_out, float* __restrict__ reward_out, uint8_t* __restrict__ trunc_out,
            float* __restrict__ final_obs_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef EnvbLds<(WPE == 4 ? 16 : 64)> LO;
    const int wv = WPE >= 2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int lds_wave = WPE >= 2 ? wg_cold_args()->p.env_lds : 0;
    char* const sm = smem + wv * lds_wave;
    char* const shared = smem + WPE * lds_wave;             // (WPE 4: the workgroup's flags)
    int* const flags = reinterpret_cast<int*>(shared);
    if (WPE == 4 && threadIdx.x < 8) flags[threadIdx.x] = 0;      // (before the prologue's workgroup barrier: nobody sets one earlier)
    EnvFlowOut fo;
    envb_flow<NOISE, WPE, GLUE>(sm, shared, wv, mode, actions, mask, chunk, fo);
    if (GLUE != 0) {
        full_barrier<64>();
        const EnvKArgsPtr kb = (EnvKArgsPtr)wg_cold_args();
        const int e = (int)blockIdx.x, F = kb->p.F, K = kb->p.K;
        const int c_w = WPE == 4 ? (wv >> 1) : wv, farm_w = WPE == 4 ? (wv & 1) : 0;      // this wave's context / farm (WPE >= 2)
        const bool lane0 = (threadIdx.x & 63) == 0;
        if (WPE == 4 && lane0) envb_flag_set(flags + wv);      // this wave's flow part is complete, its stores have left the wave
        if (WPE >= 2) {
            // The glue needs nothing of the background context's waves unless the env truncates in this step (then it swaps that
            // context in): the waves meet at a workgroup barrier ONLY then — all know from the env's header — and otherwise run to
            // their ends side by side: the background context plans its own next share and clears its set-up flag.
            if (c_w != fo.env_live && farm_w == 0) {
                if (kb->p.autoreset && !fo.truncates && lane0) {
                    int work = 0;
                    if (WPE == 4) {
                        if (!envb_flag_wait(flags + wv + 1)) atomicOr(kb->d.status, WG_STATUS_BIT_STATE);
                        for (int f = 0; f < F; ++f) {
                            const EnvbSlotLds* const s_ = reinterpret_cast<const EnvbSlotLds*>(smem + (wv + f) * lds_wave + LO::SL);
                            work = max(work, s_->dev_rem + K * s_->fill_rem);
                        }
                    } else {
                        const EnvbSlotLds* const SLb = reinterpret_cast<const EnvbSlotLds*>(sm + LO::SL);
                        for (int f = 0; f < F; ++f) work = max(work, SLb[f].dev_rem + K * SLb[f].fill_rem);
                    }
                    const int steps_done = fo.steps_done + 1, time_max = fo.time_max_live;      // (as the glue sees them)
                    const int inc = 1 + (kb->gp.extra_inc ? 1 : 0);
                    const long total = (long)((time_max + inc - 1) / inc) + 1;
                    kb->d.env_rw[e].shadow_iters = work == 0 ? 0 : wg_shadow_share(work, total - steps_done, steps_done, e);
                }
                if (fo.bg_init_pending && lane0) kb->d.ctx[e * 2 + c_w].init_pending = 0;
            }
            if (c_w != fo.env_live && fo.truncates) __builtin_amdgcn_s_waitcnt(0x0070);       // (its last stores, before the barrier releases the glue)
            if (fo.truncates) __syncthreads();
        }
        if (WPE == 1 || (c_w == fo.env_live && farm_w == 0)) {
            const EnvKArgsPtr kg = (EnvKArgsPtr)wg_cold_args();
            LeanFused fz;
            fz.pre = nullptr;
            if (WPE == 4) {
                // the running episode's agent-farm wave: its baseline farm's wave is the next one
                if (F == 2 && !envb_flag_wait(flags + wv + 1)) { if (lane0) atomicOr(kg->d.status, WG_STATUS_BIT_STATE); }
                fz.fp = reinterpret_cast<const EnvbSlotLds*>(sm + LO::SL)->out_pw;
                fz.bp = F == 2 ? reinterpret_cast<const EnvbSlotLds*>(sm + lds_wave + LO::SL)->out_pw : 0.f;
                fz.work = 0;
            } else {
                const EnvbSlotLds* const SLa = reinterpret_cast<const EnvbSlotLds*>(sm + LO::SL);      // the live context's slots
                const int la = WPE == 2 ? 0 : fo.env_live * F, lb = (fo.env_live ^ 1) * F;
                fz.fp = SLa[la].out_pw;
                fz.bp = F == 2 ? SLa[la + 1].out_pw : 0.f;
                int work = 0;
                if (WPE == 1) for (int f = 0; f < F; ++f) work = max(work, SLa[lb + f].dev_rem + K * SLa[lb + f].fill_rem);
                fz.work = work;
            }
            fz.bg_init_pending = WPE >= 2 ? 0 : fo.bg_init_pending;
            fz.plan_elsewhere = WPE >= 2;
            fz.hw = reinterpret_cast<const int*>(sm + LO::HDR)[threadIdx.x & 31];
            lean_step<GLUE == 2, false, true>(*(const WgParams*)&kg->gp, *(const WgPtrs*)&kg->gd, kg->d.gp, kg->d.gd, (int)blockIdx.x,
                                              (int)(threadIdx.x & 63), kg->obs, kg->reward, kg->trunc, kg->final_obs, nullptr, fz);
        }
    }
}

extern "C" void wg_launch_flow_envb(const FlowP* p, const FlowPtrs* d, int mode, const float* actions, const uint8_t* mask,
                                    int chunk, hipStream_t st) {
    const int grid = p->B, wpe = p->env_wpe == 4 ? 4 : (p->env_wpe == 2 ? 2 : 1);
    const size_t lds = (size_t)p->env_lds * wpe + (wpe == 4 ? WG_ENVB_FLAG_BYTES : 0);
    static const WgParams gp0{};
    static const WgPtrs gd0{};
#define WG_FLOW_ENVB(NZ, W) hipLaunchKernelGGL((k_flow_envb<NZ, 0, W>), dim3(grid), dim3(64 * W), lds, st, *p, *d, mode, actions, mask, chunk, gp0, gd0, \
                                               (float*)nullptr, (float*)nullptr, (uint8_t*)nullptr, (float*)nullptr)
    if (wpe == 4) { if (p->noise) WG_FLOW_ENVB(true, 4); else WG_FLOW_ENVB(false, 4); }
    else if (wpe == 2) { if (p->noise) WG_FLOW_ENVB(true, 2); else WG_FLOW_ENVB(false, 2); }
    else { if (p->noise) WG_FLOW_ENVB(true, 1); else WG_FLOW_ENVB(false, 1); }
#undef WG_FLOW_ENVB
}

// step() as one launch (wg_api.hip: launch_step, handles with FlowP::env_fused and frozen-box inflow)
extern "C" void wg_launch_step_envb(const FlowP* p, const FlowPtrs* d, const WgParams* gp, const WgPtrs* gd, const float* actions,
                                    float* obs, float* reward, uint8_t* trunc, float* final_obs, hipStream_t st) {
    const int grid = p->B, wpe = p->env_wpe == 4 ? 4 : (p->env_wpe == 2 ? 2 : 1);
    const size_t lds = (size_t)p->env_lds * wpe + (wpe == 4 ? WG_ENVB_FLAG_BYTES : 0);
#define WG_STEP_ENVB(NZ, G, W) hipLaunchKernelGGL((k_flow_envb<NZ, G, W>), dim3(grid), dim3(64 * W), lds, st, *p, *d, (int)WG_MODE_STEP, actions, \
                                                  (const uint8_t*)nullptr, 0, *gp, *gd, obs, reward, trunc, final_obs)
#define WG_STEP_ENVB_W(NZ, G) do { if (wpe == 4) WG_STEP_ENVB(NZ, G, 4); else if (wpe == 2) WG_STEP_ENVB(NZ, G, 2); else WG_STEP_ENVB(NZ, G, 1); } while (0)
    if (gd->multi_out) { if (p->noise) WG_STEP_ENVB_W(true, 2); else WG_STEP_ENVB_W(false, 2); }
    else { if (p->noise) WG_STEP_ENVB_W(true, 1); else WG_STEP_ENVB_W(false, 1); }
#undef WG_STEP_ENVB_W
#undef WG_STEP_ENVB
}
