// wg_state.h — device-resident state of libwindgym_hip.so (internal; the public ABI is include/windgym_hip.h).
//
// Data layout in HBM.  B envs; every env owns TWO episode contexts ("ctx"): the live episode
// and the next one, which is developed in the background so that autoreset never stalls the batch
// (reference: WindFarmEnv.reset runs hundreds of hidden flow steps, Wind_Farm_Env.py:722-796).
// Every ctx owns F farm "slots" (agent farm, baseline farm).  slot id = (env*2 + ctx)*F + farm.
//
//   particle SoA, fp32, [n_slots][N*P] each, ring index fastest -> a workgroup streams contiguous memory:
//       py                      transverse position of the wake particle (r/w every step)
//       rec_a, rec_b            frozen emission record ct|k, u_e|hv as fixed point (read by the advection pass AND by the deficit
//                               phase's bracket gathers: round 6 — rounds 2-5 kept u_e in a separate array / a 16-byte gather copy)
//       pz, vlp, wlp            vertical position + low-pass filtered transverse turbulence (box inflow only)
//   turbine SoA, fp32, [n_slots][N]: yaw, u, v, w, ti_loc, power, ct
//   sensors, fp32, per ctx: ring[ch][N][H_ch], farm ring[ch][H_ch]   (MesClass deques)
#pragma once
#include <stdint.h>

#include "../../include/windgym_hip.h"

#define WG_BLOCK 256
#define WG_WAVE 64
#define WG_NWAVES (WG_BLOCK / WG_WAVE)

typedef unsigned __int128 wg_u128;
// numpy's PCG64 multiplier (pcg64.h: PCG_DEFAULT_MULTIPLIER_128)
#define WG_PCG_MULT ((((wg_u128)2549297995355413924ULL) << 64) | (wg_u128)4865540595714422341ULL)

enum { WG_MODE_STEP = 0, WG_MODE_RESET = 1 };

// Running window sums of the live episode's sensor deques (WgPtrs::wsum), maintained by the lean glue kernel: per step it
// reads the newest and the leaving sample of each window instead of staging the rings ("sums mode", wg_create).  Slot s < WG_N_CH: sum
// of the newest min(window_len, history_len, pushed) samples of channel s — the one window a rolling mean with history_N
// = 1 reads (MesClass.py:85-91); slots 4 / 5: sum of v and of v^2 over the whole ws deque (calc_TI, MesClass.py:220-237).
// Double accumulators: adding and later subtracting the same float is exact for the linear slots (the sums do not drift); the
// sum of squares (WG_SUM_TI2) rounds — see wg_sums_apply (wg_obs.h).
#define WG_N_SUMS 6
enum { WG_SUM_TI1 = 4, WG_SUM_TI2 = 5 };

// Sticky device status word (WgPtrs::status): one bit per condition, latched with atomicOr, so that a saturated emission
// record cannot hide a NaN power or a step on a finished env; wg_check reports the most serious one set.
enum { WG_STATUS_BIT_NAN_POWER = 1, WG_STATUS_BIT_STATE = 2, WG_STATUS_BIT_RANGE = 4 };

// What one turbine's block of the observation is made of (sums mode, wg_obs_turbine).  Scaling with the
// reciprocal of the range: v -> 2 (v - mn) inv_rng - 1, clipped to [-1, 1] (turb_mes._scale_val, MesClass.py:324-326, to
// one float rounding).
struct WgObsCfg {
    unsigned cur_mask, rol_mask;       // channels observed through `current` / through the rolling mean (turbine level)
    int turb_ti, turb_obs, obs_dim, obs_dim_multi;
    int hlen[WG_N_CH], wlen[WG_N_CH];
    float mn[WG_N_CH], inv_rng[WG_N_CH], ti_mn, inv_ti_rng;
    double inv_w[WG_N_CH];             // 1 / min(window, history): the divisor once the window is full
};

struct WgParams {
    int B, N, F, K, P, S, NP;
    float dt, D, inv_D, hub;
    double dt_d, dpart, D_d, hub_d;
    float yaw_min, yaw_max, yaw_step;
    double yaw_min_d, yaw_max_d, yaw_step_d, yaw_start;
    int action_method, base_controller, yaw_init, has_yaw_defined;
    double ws_min, ws_max, ti_min, ti_max, wd_min, wd_max, n_passthrough;
    int never_truncate;
    int full_chains;     // 1: no chain pruning (wg_config.full_chains)
    wg_channel ch[WG_N_CH];
    int turb_on[WG_N_CH];   // level flags per channel for the turbine block (yaw always on)
    int farm_on[WG_N_CH];   // level flags for the farm block (yaw never)
    int turb_ti, farm_ti;
    float sc_min[WG_N_CH], sc_rng[WG_N_CH];  // (float)min, (float)(max-min) of _scale_val per channel (turbine)
    float sc_rng_farm_power;                 // (float)(power_max*N - 0)
    float ti_min_f, ti_rng_f;
    int noise;
    float noise_sigma[WG_N_CH];
    int reward_mode, power_avg;
    double power_scaling, action_penalty;
    int penalty_type;
    int fill_a, fill_b, autoreset, extra_inc, turb_mode;
    float ka, kb, eps0, hill, tia, tib, tic, tid;
    double fc_scale;
    int n_tab;
    int obs_dim, obs_dim_multi, hist_max, turb_obs, farm_obs;
    int ring_off[WG_N_CH];   // float offset of channel ch inside a ctx's turbine-ring block
    int ring_stride;         // floats per ctx of turbine rings
    int fring_off[WG_N_CH];
    int fring_stride;
    // frozen turbulence box(es)
    int n_boxes;
    int bnx, bny, bnz;
    double bdx, bdy, bdz;
    // replay mode
    int script_rows;
    int compact;         // 1: per-turbine ring lengths (small-farm k_flow variant), see WgPtrs::roff
    int stage_ch[WG_N_CH];   // k_glue ring staging per channel: 0 = not observed, 1 = newest sample only, 2 = whole ring
    double cx0, cy0;     // farm centre (mean of the layout), the pivot of the flow-frame rotation
    // sums mode (every observed rolling mean has history_N = 1): the lean glue kernel keeps running window sums
    // (WgPtrs::wsum): which slots are maintained at turbine level (_t) and for the farm-level deques (_f), which channels are
    // observed through their newest sample; sum_w[s] = min(window, history) of slot s
    int sums_mode;
    unsigned ring_magic[WG_N_CH];   // floor(2^32 / ring_cap) + 1 (wg_umod)
    int ring_cap[WG_N_CH];   // physical length of the sensor rings: history_len, + 1 in sums mode (the sample leaving a window
                             // as long as the deque must survive the push that displaces it)
    WgObsCfg oc;
    unsigned sum_mask_t, sum_mask_f, cur_mask_t, cur_mask_f;
    int sum_w[WG_N_SUMS];
};

// per farm slot
struct WgSlot {
    double s_off;       // distance travelled by the newest particle since its emission
    double time;        // fs.time
    int head;           // ring slot of the newest particle (all turbines of a farm emit together)
    int n_valid;        // particles emitted so far (saturates at P)
    int dev_remaining;  // flow-development steps still to run (fs.run(t_developed))
    int fill_remaining; // window-fill env steps still to run
    int cursor;         // replay mode row
    unsigned flow_count; // flow_step() executions of this slot (roofline accounting; summed on the host)
    unsigned istep;      // flow steps since the farm was built (counter of the inflow random stream)
    unsigned part_count; // particles the advection passes streamed (roofline accounting, like flow_count)
    unsigned n_emitted;  // compact rings: particles each chain has emitted so far (ring head of turbine t = (n_emitted - 1) mod R_t)
    unsigned add_count;  // rotor points at which the wake-added turbulence box was looked up (roofline accounting: 96 B each)
};

// per episode context
struct WgCtx {
    double ws, wd, ti;
    double dist, t_inflow;
    float rated_power;
    int t_developed, time_max;
    int n_pushed;        // add_measurements calls so far (all MesClass deques advance together)
    uint32_t turb_seed;
    double box_ox, box_oy;   // horizontal offset of this episode into the shared turbulence box (m)
    int episode_tag;     // episode index this ctx belongs to (noise counter)
    // fill pushes into the env-level power deques are deferred until the ctx goes live
    int pend_farm_n, pend_base_n;
    // Pipelined autoreset: at truncation k_glue only flags the retired context (init_pending = 1) and snapshots the
    // env's generator; the sampling + layout rotation + slot initialisation of the episode after next then run at the
    // head of the NEXT k_flow launch, in the two farm workgroups of that context (throughput-hidden among thousands of
    // workgroups instead of sitting on k_glue's one-wave-per-env latency chain).  The following k_glue clears the flag.
    int init_pending;
    int box_id;          // which box of the pool this episode uses (turb_mode BOX_POOL)
    uint32_t snap_has32, snap_u32;
    wg_u128 snap_state, snap_inc;
};

// per env
struct WgEnv {
    wg_u128 rng_state, rng_inc;   // PCG64 (numpy Generator behind gymnasium's np_random)
    uint32_t rng_has32, rng_u32;
    uint64_t noise_key;
    int live;            // which ctx is the running episode
    int timestep;
    int episode;         // episodes finished
    int done;            // truncated and not reset (autoreset off)  -> step() is an error
    int shadow_iters;    // flow sub-steps the background ctx advances during the next step()
    int farm_pow_n, base_pow_n;   // total pushes into farm_pow_deq / base_pow_deq
    int steps_done;      // step() calls in the running episode
    float ep_return, ep_power_sum;
    int ep_len;
    // lean glue kernel (sums mode): what it needs of the LIVE context, kept in the env header so that its loads depend on
    // the env index only — set wherever `live` changes (end of reset, swap); n_pushed_live = sensor pushes of the live
    // episode the window sums account for (the flow kernel has pushed one more when the glue runs)
    int time_max_live;
    float rated_live;
    int n_pushed_live;
    // running sums of the power deques (farm_pow / base_pow): S += new - overwritten, exact in double like the window
    // sums; recomputed from the deques at the end of reset and at every swap
    double fsum_run, bsum_run;
};

struct WgPtrs {
    // particles
    float *py, *pz, *vlp, *wlp;
    unsigned *rec_a, *rec_b;   // packed emission record: ct|k (16 + 16 bits) and u_e|hv (18 + 14 bits) as fixed point
    // turbines [n_slots][N]
    float *yaw, *u, *v, *w, *ti_loc, *power, *ct;
    float* bnd;               // [n_slots][N][4]: running maxima over a chain: excursion, k, eps (pruning bounds); [3] = bits of the emission count at the chain's last moving emission (TurbLds::mvl)
    WgSlot* slot;
    WgCtx* ctx;
    WgEnv* env;
    double *xr, *yr;          // [B*2][N] flow-frame positions
    float* multi_out;         // optional [B][N][obs_dim_multi]: per-agent observations written by the glue every step
    int* jneed;               // [B*2][N] oldest particle age of a chain that can still reach a rotor (chain pruning)
    // Compact rings (small farms): turbine t of a context keeps only the R_t = roff[t+1] - roff[t]
    // youngest particles of its chain — the ones that can still reach a rotor — at [slot base + roff[t], + R_t); R_t
    // is a multiple of 4 set per episode from the rotated layout.  qown[q] = owner turbine of quad q (4 ring slots).
    int* roff;                // [B*2][N+1]
    uint8_t* qown;            // [B*2][NP/4]
    float *ring, *fring;      // [B*2][ring_stride], [B*2][fring_stride]
    float *cur_ws, *cur_wd;   // [B*2][N] last sub-step measurement (info dict)
    float *pend_farm, *pend_base;   // [B*2][power_avg]
    float *farm_pow, *base_pow;     // [B][power_avg]
    float *old_yaw;           // [B][N]
    float *step_farm_pow, *step_base_pow;   // [B] produced by the flow kernel, consumed by the glue kernel
    float *last_pow_agent, *last_pow_base;  // [B] farm power of the step just taken (before an autoreset swap)
    float* metrics;           // [B][WG_N_METRICS] running per-env sums
    // First observation of a background episode, built by the k_flow workgroup that completes its development
    // (wg_first_obs) so that the truncating glue wave copies it instead of staging and building a second observation;
    // next_obs_ok[ctx] = 1 while next_obs[ctx] holds the context's current episode (cleared when the context
    // is retired — k_glue — or reset — k_init; not part of the state blob: without it the glue builds the observation
    // itself).  Null unless the handle runs the single-wave steady flow kernel (wg_create).
    float* next_obs;          // [B*2][obs_dim]
    int* next_obs_ok;         // [B*2]
    // sums mode: wsum[B*2][WG_N_SUMS][N + 1] running window sums of a context's deques (entity N = the farm-level deques):
    // advanced by k_glue_lean while the episode is live, summed afresh when it goes live (end of reset; swap: by the flow
    // kernel's wg_first_obs where it prepares the episode, else by lean_swap); null otherwise
    double* wsum;
    int* status;              // sticky error word
    const double* wind_override;   // [B][3] (ws, wd, ti) or null; NaN = keep the sampled value
    const int* box_override;       // [B] box of the pool env e uses (FarmEval.update_tf: TF_files = [path]) or null; < 0 = draw
    // config tables
    const double *x_pos, *y_pos, *yaw_defined;
    const uint64_t* pcg_jump;      // [max(N, 3) + 1][4]: (A_k lo, hi, G_k lo, hi), k draws ahead: state_k = A_k state_0 + G_k inc (wg_ctx_init)
    const float *rotor_dy, *rotor_dz;
    const float *tab_ws, *tab_power, *tab_ct;
    const double *tab_ws_d, *tab_power_d;
    // inflow
    const float* box;
    // replay
    const float *script_uvw, *script_power;
};
