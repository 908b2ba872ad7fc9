// wg_flow.h — slim parameter block of k_flow (see wg_flow.hip for why it is separate from WgParams).
#pragma once
#include "wg_state.h"

#ifndef WG_BOX_WAVES
#define WG_BOX_WAVES 2    // turbulent variants: measured best with the full register budget (no spills, deeper gather ILP)
#endif
#ifndef WG_FLOW_WAVES_128
#define WG_FLOW_WAVES_128 6   // 128-thread variants stream one quad per lane at a time (QB = 1): 80 VGPRs, 6 waves/SIMD
#endif
#ifndef WG_FLOW_WAVES_CG
#define WG_FLOW_WAVES_CG 5    // small-farm variants (compact rings, 64 / 128 threads): 96 VGPRs; measured 4 / 5 / 6 waves: 104 / 93.5 / 96 us on cfg2
#endif
#ifndef WG_FLOW_WAVES_GL
#define WG_FLOW_WAVES_GL 4    // single-wave steady variant (GL): 128 VGPRs, no spills — room for the pipelined advection pass's second quad;
                              // 4 vs 5 waves per SIMD measured equal without the pipeline (the launch is not occupancy-bound), -4.5 % with it
#endif
#ifndef WG_FLOW_WAVES
#define WG_FLOW_WAVES 5   // min waves/SIMD the register allocator must leave room for (5 -> <= 96 VGPRs; measured best)
#endif

#ifndef WG_GLDS
#define WG_GLDS 1         // single-wave steady compact variant: deficit-phase gathers as early LDS-DMA requests (0 = register gathers, for A/B builds)
#endif
#ifndef WG_ADV_PIPE
#define WG_ADV_PIPE 1      // GL variant: software-pipelined advection pass (first quad requested before the deficit evaluation, the next
                           // quad before the current one is computed): cfg2 k_flow 61.5 -> 58.8 us same box (0 = plain loop, for A/B builds)
#endif
#ifndef WG_ADV_PIPE_LF
#define WG_ADV_PIPE_LF 1   // 256-thread compact steady variant (large farms): quads requested ahead in its advection pass
                           // (cfg3 same-box: 0 / 1 / 2 quads ahead = 3.80 / 4.04 / 3.93 M env-steps/s).  vmcnt counts
                           // loads and stores in ONE in-order queue: a plain load-compute-store loop waits for the previous trip's
                           // stores whenever it waits for its loads — two round trips per trip (cfg3: 20 trips of ~2 us).  Requested
                           // before the stores, the next quads' loads no longer queue behind them.
#endif
#ifndef WG_FLOW_WAVES_LF
#define WG_FLOW_WAVES_LF 3  // ... built at 3 waves per SIMD (168 VGPRs, nothing spilled): bandwidth-bound, so three resident workgroups per
                            // CU do (cfg3 same-box, 4 / 3 waves: 3.88 / 4.00 M env-steps/s; its LDS carve allows 4: 34 KB each)
#endif
#ifndef WG_LF_PAIR
#define WG_LF_PAIR 1      // 256-thread compact steady variant: pair phase over the whole farm at once — per-target source masks built
                          // without ballots or LDS atomics, candidate list from a prefix sum (ascending (target, source) order),
                          // results staged per CANDIDATE (0 = the chunked, densely staged phase shared with the small-farm variants)
#endif
// LDS layout of that phase inside the staging region (bytes; n = turbines): packed sources | masks | list offsets + wave totals |
// candidate list (worst case n (n - 1) / 2 entries) | per-candidate deficit, added TI (lf_cap each)
#define WG_LF_OFF_TM(n) (16 * (size_t)(n))
#define WG_LF_OFF_BASE(n) (32 * (size_t)(n))
#define WG_LF_OFF_CL(n) ((36 * (size_t)(n) + 4 * 8 + 15) & ~(size_t)15)
#define WG_LF_OFF_DEF(n) ((WG_LF_OFF_CL(n) + (size_t)(n) * ((n) - 1) + 15) & ~(size_t)15)
#ifndef WG_S_UNROLL_ALL
#define WG_S_UNROLL_ALL 0   // 1: the rotor-point loop of the Gaussian pair evaluation unrolled by 4 in every variant (A/B builds)
#endif
#ifndef WG_GL_POSTPASS_WAIT
#define WG_GL_POSTPASS_WAIT 1   // GL variant: the compiler-visible vmcnt(0) right after the pipelined advection pass (0: A/B builds)
#endif
#ifndef WG_BOX_XCD_PAIRS
#define WG_BOX_XCD_PAIRS 1   // frozen-box variants: the two farms of an env on the same XCD (0 = adjacent block indices, for A/B builds)
#endif
#ifndef WG_PAIR_FIRST
#define WG_PAIR_FIRST 1   // steady compact variant: deficit phase BEFORE the advection pass (0 = round-2 order, for A/B builds)
#endif

struct FlowP {
    int B, N, F, K, P, S, S_pad, S_shift, NP, n_tab;
    int autoreset, action_method, base_controller, power_avg, script_rows, noise;
    int block;                    // threads per workgroup: 64, 128 or 256
    int res;                      // 1: compact per-turbine rings + pair-major deficit phases (small farms), 0: legacy streaming variant
    int pstride;                  // floats between the particle blocks of consecutive farm slots (>= NP, see wg_create)
    int target_chunk;             // targets whose pair parameters are staged in LDS at once
    int lf_cap;                   // large-farm steady variant (WG_LF_PAIR): candidates whose results fit the staging region at once
    int lds_off_turb, lds_off_tab, lds_bytes;
    int rec_il;                   // the packed emission record is ONE interleaved array (rec_a[2 i] = ct|k, rec_a[2 i + 1] = u_e|hv; rec_b = rec_a + 1):
                                  // steady compact handles (GL / k_flow_env, LF) — a bracket pair is 16 contiguous bytes of the array
                                  // the advection pass streams, an emission writes one sector
    int gl;                       // the launch runs the GL variant of k_flow (LDS-DMA gathers; no per-target source masks in LDS)
    int lds_off_ql, lds_off_gat;  // single-wave steady compact variant: quad list of its own (0 = aliases the pair staging) and the
                                  // landing zone of the deficit phase's LDS-DMA gathers (WG_GAT_BYTES)
    int ql_shift;                 // compact steady advection: quad-list entry = turbine << ql_shift | quad index in its ring (16 bits)
    int ql_lpt_shift;             // ... and 2^ql_lpt_shift lanes share the listing of one turbine's quads (block / N, at most 8)
    // k_flow_env (wg_env.hip: ONE wave per env, lane = slot * N + turbine): enabled; bytes of dynamic LDS; offset of the tables
    // (the rest of the carve is fixed: WG_ENV_*)
    int envw, env_lds, env_off_tab;
    int env_cap, envb_off_cl;         // k_flow_envb (wg_envb.hip, frozen-box inflow): staged wakes per chunk of targets; offset of the candidate list
    int env_wpe;                      // waves per env: 1, or 2 (a workgroup of two waves, one per context, each with its own LDS region of env_lds bytes)
    int env_split;                    // k_flow_env's small-batch instantiation: a third wave per env runs the advection pass of the running episode's context
    int env_fused;                    // step() as ONE launch: the env's wave runs its glue (lean_step) as the tail of its flow step
    int env_inc;                      // env time steps per step(): 1 + extra_timestep_inc (the background plan's steps left)
    float env_eps_max;                // widest initial wake width a record can hold: min(1, eps0 sqrt(beta(ct = 0.96)))
    float dt, D, inv_D, hub, dpart_f, R_rot, inv_N, inv_S, inv_P;
    double dt_d, dpart, inv_dpart;
    float yaw_min, yaw_max, yaw_step;
    float ka, kb, eps0, hill, tia, tib, tic, tid;
    float ue_scale;               // a particle's u_e is stored as a 16-bit fraction of ue_scale x the episode's free-stream speed (wg_flow_dev.h)
    float tab_x0, tab_inv_dx;     // uniform-grid turbine table
    int hlen[WG_N_CH], ring_off[WG_N_CH], fring_off[WG_N_CH];
    float inv_hlen[WG_N_CH], inv_power_avg;   // reciprocals for the division-free ring positions (fast_mod)
    unsigned hmagic[WG_N_CH], pavg_magic;     // floor(2^32 / H) + 1: n mod H in three integer (scalar) operations, n * H < 2^32
    int ring_stride, fring_stride;
    float noise_sigma[WG_N_CH];
    // turbulent inflow (Random / frozen Mann box)
    int turb_mode, bnx, bny, bnz, box_pow2;
    int coarse, cnx, cny, cnz, cbox_pow2;   // block-averaged (4^3) copy of the box for the particle lookups
    long long box_cells, cbox_cells;        // cells of one box of the pool (fine / block-averaged): box k starts at k * cells
    double inv_bdx, inv_bdy, inv_bdz, fc_scale, D_d, hub_d;
    float inv_sqrt_S;
    // wake-added turbulence (wg_config.added_turbulence): isotropic box, k_mt constants (km2r = 2 km2 R_rot)
    int added, anx, any, anz, abox_pow2;
    int no_ti_fold, deficit_model;
    float km1, km2r;
    float sg_af, sg_bf, sg_cf;    // super-Gaussian order n(x) = af exp(bf x/D) + cf (deficit_model = 1)
    double inv_adx, inv_ady, inv_adz;
    double inv_bn[3], inv_an[3];  // 1 / (cells per dimension) of the fine / wake-added box: index wrap in double (stencil records)
    // deficit_model 2: tabulated eddy-viscosity (Ainslie / DWM) deficit, FlowPtrs::dtab [an_ct][an_ti][an_x][an_r] (wg_set_deficit_table):
    // Ct uniform from an_ct0, TI log-uniform from exp(an_lti0), x / D and r / R uniform from 0
    int an_ct, an_ti, an_x, an_r;
    float an_ct0, an_inv_dct, an_lti0, an_inv_dlti, an_inv_dx, an_inv_dr, an_inv_ka;
};

struct FlowPtrs {
    float *py, *pz, *vlp, *wlp;
    unsigned *rec_a, *rec_b;      // packed emission record incl. u_e (wg_flow_dev.h: pack_a / pack_b)
    const float4* box4;          // interleaved copy of the turbulence box: [Nx][Ny][Nz] x (u, v, w, 0)
    const float4* box4c;         // the same block-averaged over 4x4x4 cells
    const float4* abox4;         // isotropic box of the wake-added turbulence, interleaved like box4
    // STENCIL RECORDS (k_flow_envb's rotor-point lookups): for every cell origin (i, j, k) the 8 corners of its trilinear stencil,
    // periodic wrap included, as ONE aligned 128-byte record — float4[8] = v000 v100 v010 v110 v001 v101 v011 v111, record
    // (i ny + j) nz + k.  8 x the float4 box in HBM (8.6 GB for the reference's 2048 x 512 x 64 box: the card has 288), and a rotor
    // point costs ONE 128-byte line instead of the ~3.3 lines its 2 x 2 x 2 stencil straddles in the brick-ordered box — the
    // lookups were 62 % of cfg5's fetched bytes.  Null when the pool is too large for it (wg_set_turbulence_boxes).
    const float4* box8;
    const float4* abox8;
    const float* dtab;           // deficit_model 2: the deficit table (see FlowP::an_*)
    float *yaw, *u, *v, *w, *ti_loc, *power, *ct;
    float* bnd;                   // [n_slots][N][4] conservative chain bounds (excursion, k, eps) + last moving emission (uint bits)
    WgSlot* slot;
    WgCtx* ctx;
    const WgEnv* env;
    const double *xr, *yr;
    const int* jneed;             // [B*2][N] chain pruning: ages above this are not advected (see ctx_init)
    const int* roff;              // [B*2][N+1] compact ring offsets (res)
    const uint8_t* qown;          // [B*2][NP/4] owner turbine of every quad of ring slots (res)
    int* status;                  // sticky error word
    float *ring, *fring, *cur_ws, *cur_wd, *pend_farm, *pend_base, *old_yaw, *step_farm_pow, *step_base_pow;
    const float *rotor_dy, *rotor_dz, *tab_power, *tab_ct;   // tab_*: resampled on the uniform grid
    const float *script_uvw, *script_power;
    long long* dbg;               // WG_TIMELINE debug builds: [n_blocks][16] phase stamps
    // device-resident copies of the full parameter / pointer blocks: read (scalar loads) only on the rare episode
    // initialisation path at the head of k_flow, so that the hot path keeps its slim kernel arguments
    const WgParams* gp;
    const WgPtrs* gd;
    WgEnv* env_rw;                // == env; the initialising workgroup of farm 0 commits the advanced generator
};

// k_flow_env's LDS carve (wg_env.hip): fixed part (cross-lane turbine fields: three 16-byte and two 8-byte arrays of 64 entries,
// then four slot records) | staging: per-candidate deficit, added TI (WG_ENV_CAP floats each), candidate list (u16), aliased by
// the quad list | tables (FlowP::env_off_tab)
#define WG_ENV_SLOT_LDS_BYTES 160
#define WG_ENV_FIXED_LDS_BYTES (3 * 64 * 16 + 2 * 64 * 8 + 4 * WG_ENV_SLOT_LDS_BYTES + 128)      // (+ the env header's 32 words)
#define WG_ENV_CAP 256
// k_flow_envb's LDS carve (wg_envb.hip): fixed part (per-lane turbine fields + row table + four slot records of 208 bytes + the lanes' parked registers) |
// staged wakes float4[env_cap] | added TI float[env_cap] | candidate list (u16) | tables (FlowP::env_off_tab)
#define WG_ENVB_FIXED_LDS_BYTES(nlp) ((93 + 48) * (nlp) + 4 * 208 + 128)      // nlp = lanes the per-lane arrays are sized for: 64, or 16 (four waves per env)
// sizeof(TurbLds) in wg_flow.hip; kept here so the host can size the dynamic LDS
#define WG_TURB_LDS_BYTES 120
// landing zone of the LDS-DMA gathers, per candidate lane: the 16-byte record copy (rec_a, rec_b, u_e, 0) of the two
// bracketing particles + their py
#define WG_GAT_BYTES (2 * 64 * 16 + 2 * 64 * 4)
// per-target bit mask of contributing sources: 32-bit words per target (N <= 32 * WG_MASK_WORDS)
#define WG_MASK_WORDS 4
