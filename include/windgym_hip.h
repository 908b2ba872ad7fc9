/*
 * windgym_hip.h — C ABI of libwindgym_hip.so: the batched, MI355X-resident WindGym step() transition.
 *
 * What this boundary replaces in the reference (DTUWindEnergy/WindGym @ 2025-04-18):
 *   - the duck-typed flow-simulation object `fs` / `fs_baseline` (external DYNAMIKS DWMFlowSimulation;
 *     every call site is listed in SURVEY.md Appendix A: WindGym/Wind_Farm_Env.py:702-711, 734, 745,
 *     770-782, 793, 945, 953; rotor_avg_windspeed :485-490; power() :495, :539-540),
 *   - the sensor model `farm_mes` (WindGym/MesClass.py:354-703),
 *   - the baseline yaw controllers (WindGym/BasicControllers/BasicControllers.py:10-73),
 *   - and the body of WindFarmEnv.reset()/step() that strings them together
 *     (WindGym/Wind_Farm_Env.py:680-802, 920-1034),
 * for a batch of `n_envs` independent farms at once.  One handle == one GPU == one shard of the env axis.
 *
 * Conventions
 *   - every entry point returns 0 on success or a negative wg_status; wg_last_error() gives the message
 *     of the last failure on the calling thread.  No C++ exception crosses the boundary.
 *   - all `*_dev` pointers are DEVICE pointers owned by the caller (PyTorch tensors' data_ptr());
 *     the library owns only the state it allocates in wg_create(); nothing is allocated on the step path.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls on one handle must be
 *     stream-ordered by the caller; a handle is not thread-safe; different handles are independent.
 *   - floating point state on the device is fp32; configuration scalars are passed as double.
 *
 * The physics is "model M0" (DESIGN.md §2): DYNAMIKS itself is not available to this build
 * (SURVEY.md §0.2-0.4), so parity for the flow values is against oracle/ (a CPU restatement of M0),
 * while the glue semantics are pinned by golden vectors recorded from the reference's own code
 * (tests/golden/).
 */
#ifndef WINDGYM_HIP_H
#define WINDGYM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WG_ABI_VERSION 4

typedef enum wg_status {
    WG_OK = 0,
    WG_ERR_INVALID = -1,      /* bad argument / inconsistent config (reference: ValueError)          */
    WG_ERR_UNSUPPORTED = -2,  /* reference: NotImplementedError (ActionMethod "absolute", Track_power)*/
    WG_ERR_HIP = -3,          /* a HIP runtime call failed                                            */
    WG_ERR_NAN_POWER = -4,    /* reference: Exception("NaN Power"), Wind_Farm_Env.py:980-981           */
    WG_ERR_STATE = -5,        /* step() on a truncated env without autoreset (reference tears fs down)*/
    WG_ERR_NOMEM = -6,
    WG_ERR_RANGE = -7         /* a wake particle's emission record left the range of its 16-bit fixed-point storage
                               * (wake-growth rate k > 0.25, i.e. local TI > 0.65, or deflection speed |hv| > 16 m/s,
                               * i.e. rotor wind speed > 40 m/s, or — turbulent inflow only — a rotor wind speed above twice
                               * the episode's free-stream speed, the scale of the record's u_e field): the value was
                               * saturated; reported by wg_check                                                      */
} wg_status;

/* sensor channels, order fixed by MesClass.turb_mes.get_measurements (MesClass.py:328-351) */
enum { WG_CH_WS = 0, WG_CH_WD = 1, WG_CH_YAW = 2, WG_CH_POWER = 3, WG_N_CH = 4 };

/* one `Mes` (MesClass.py:23-125) */
typedef struct wg_channel {
    int32_t current;       /* *_current        */
    int32_t rolling_mean;  /* *_rolling_mean   */
    int32_t history_n;     /* *_history_N      */
    int32_t history_len;   /* *_history_length (deque maxlen) */
    int32_t window_len;    /* *_window_length  */
} wg_channel;

enum { WG_ACT_YAW = 0, WG_ACT_WIND = 1 };                 /* ActionMethod, Wind_Farm_Env.py:822-864 */
enum { WG_CTRL_LOCAL = 0, WG_CTRL_GLOBAL = 1 };           /* BaseController, BasicControllers.py    */
enum { WG_YAWINIT_ZEROS = 0, WG_YAWINIT_RANDOM = 1, WG_YAWINIT_DEFINED = 2 }; /* :146-162, WindEnv.py */
enum { WG_REW_BASELINE = 0, WG_REW_POWER_AVG = 1, WG_REW_NONE = 2, WG_REW_POWER_DIFF = 3 }; /* :172-194 */
enum { WG_PEN_CHANGE = 0, WG_PEN_TOTAL = 1 };             /* _action_penalty, :804-820              */
enum { WG_NOISE_NONE = 0, WG_NOISE_NORMAL = 1 };          /* farm_mes noise, MesClass.py:436-444    */
/* turbtype (:598-668): "None" -> NONE; "Random" -> RANDOM (draws a seed, :642); "MannFixed" -> BOX (the same
 * frozen box every episode, no draw, :646-657); "MannGenerate"/"MannLoad" -> BOX_SHIFT (draws a seed like :623
 * and uses it as a random horizontal offset into the one shared box instead of generating 0.8 GB per env);
 * "MannLoad" -> BOX_POOL: a pool of K boxes (the TF_* files) resident on the GPU, one drawn per episode exactly like
 * tf_file = self.np_random.choice(self.TF_files) (:614 — the same PCG64 draw as integers(0, K)). */
enum { WG_TURB_NONE = 0, WG_TURB_RANDOM = 1, WG_TURB_BOX = 2, WG_TURB_BOX_SHIFT = 3, WG_TURB_BOX_POOL = 4 };

typedef struct wg_config {
    int32_t abi_version;       /* must be WG_ABI_VERSION */
    /* ---- batch / discretisation ------------------------------------------------------------ */
    int32_t n_envs;            /* B: farms simulated by this handle                                   */
    int32_t n_turb;            /* N = nx*ny (Wind_Farm_Env.py:139)                                    */
    int32_t n_farms;           /* F: 1, or 2 when Baseline_comp (agent farm + baseline farm, :217-220)*/
    int32_t k_sub;             /* sim_steps_per_env_step = int(dt_env/dt_sim) (:105)                  */
    int32_t n_particles;       /* P: wake-particle ring slots per turbine                             */
    int32_t n_rotor_pts;       /* S: rotor quadrature points                                          */
    double dt_sim;             /* s (:102)                                                            */
    double rotor_diameter;     /* D, turbine.diameter() (:244)                                        */
    double hub_height;         /* turbine.hub_height()                                                */
    double d_particle;         /* particle spacing in D; reference hard-codes 0.2 (:116)              */
    const double* x_pos;       /* [N] layout frame (east),  Wind_Farm_Env.py:246-252                  */
    const double* y_pos;       /* [N] layout frame (north)                                            */
    const double* rotor_dy;    /* [S] quadrature offsets in the rotor plane, metres                   */
    const double* rotor_dz;    /* [S]                                                                 */
    /* ---- turbine power / Ct table (py_wake tabular turbine; linear interpolation) ------------ */
    int32_t n_tab;
    const double* tab_ws;      /* [n_tab] ascending                                                   */
    const double* tab_power;   /* [n_tab] W                                                           */
    const double* tab_ct;      /* [n_tab]                                                             */
    /* ---- yaw actuation ----------------------------------------------------------------------- */
    double yaw_min, yaw_max;   /* deg (farm.yaw_min / yaw_max)                                        */
    double yaw_step;           /* deg per sim step (:114)                                             */
    double yaw_start;          /* 15 deg: range of yaw_init "Random" (:110, :715-720)                 */
    int32_t action_method;     /* WG_ACT_*                                                            */
    int32_t base_controller;   /* WG_CTRL_*                                                           */
    int32_t yaw_init;          /* WG_YAWINIT_*                                                        */
    const double* yaw_defined; /* [N] or NULL; used with WG_YAWINIT_DEFINED (FarmEval.set_yaw_vals)   */
    /* ---- wind-condition sampling at reset (_set_windconditions, :557-568) --------------------- */
    double ws_min, ws_max, ti_min, ti_max, wd_min, wd_max;
    double n_passthrough;      /* time_max = int(t_inflow * n_passthrough) (:732)                     */
    int32_t never_truncate;    /* FarmEval.reset: time_max = 9999999 (FarmEval.py:54-61)              */
    /* ---- sensors (farm_mes ctor, :409-451) ---------------------------------------------------- */
    wg_channel ch[WG_N_CH];
    int32_t turb_ws, turb_wd, turb_ti, turb_power, farm_ws, farm_wd, farm_ti, farm_power; /* mes_level */
    double ws_scale_min, ws_scale_max;     /* 2, 25 (:440-441)                                         */
    double wd_scale_min, wd_scale_max;     /* wd_min-5, wd_max+5 (:443-444)                            */
    double ti_scale_min, ti_scale_max;     /* TI_min_mes, TI_max_mes                                   */
    double power_max;                      /* maxturbpower (:112)                                      */
    int32_t noise;                         /* WG_NOISE_*                                               */
    double noise_sigma[WG_N_CH];           /* 0, 2, 0, 0 (MesClass.py:436-439)                         */
    /* ---- reward (:866-918, :989-996) ----------------------------------------------------------- */
    int32_t reward_mode;       /* WG_REW_*                                                            */
    int32_t power_avg;         /* deque maxlen (:142-143)                                             */
    double power_scaling;
    double action_penalty;
    int32_t penalty_type;      /* WG_PEN_*                                                            */
    /* ---- reset (:722-796) ----------------------------------------------------------------------- */
    int32_t fill_steps_agent;  /* steps_on_reset (:229-240)                                           */
    int32_t fill_steps_base;   /* hist_max (:784)                                                     */
    int32_t autoreset;         /* 0: reference semantics (step after truncation is an error);         */
                               /* 1: same-step autoreset from a pre-developed next episode            */
    int32_t extra_timestep_inc;/* WindFarmEnvMulti.step increments timestep twice (WindEnvMulti.py:219)*/
    /* ---- inflow --------------------------------------------------------------------------------- */
    int32_t turb_mode;         /* WG_TURB_*                                                           */
    /* ---- model M0 constants; 0 selects the documented default -------------------- */
    double m0_ka, m0_kb;       /* k* = ka*TI + kb              (0.38, 0.004)                          */
    double m0_eps;             /* sigma0/D = eps*sqrt(beta)    (0.2)                                  */
    double m0_hill;            /* Hill-vortex deflection speed factor (0.4)                           */
    double m0_ti_a, m0_ti_b, m0_ti_c, m0_ti_d; /* Crespo-Hernandez added TI: a*ind^b*TI^c*(x/D)^d     */
    double m0_fc_scale;        /* meandering low-pass cut-off f_c = U/(fc_scale*D)   (2.0)            */
    /* ---- chain pruning -------------------------------------------------------------------------- */
    int32_t full_chains;       /* 0: a wake particle that has passed the most downstream turbine of the farm is no
                                * longer advected — it cannot reach a rotor any more, every output of step() is
                                * unchanged; 1: advect all P slots of every chain (needed only if wg_get_windspeed
                                * must be exact behind the last turbine row)                               */
    /* ---- model options (ABI 2) ------------------------------------------------------------------- */
    int32_t added_turbulence;  /* wake-added small-scale turbulence (reference: addedTurbulenceModel =
                                * [Synchronized]AutoScalingIsotropicMannTurbulence(), Wind_Farm_Env.py:618, :638, :644,
                                * :659).  0: none; 1: an isotropic unit-variance box (wg_set_added_turbulence_box) sampled
                                * at the rotor points and scaled per source wake by k_mt = km1 |dU| + km2 |d dU / d(r/R)|
                                * (Madsen et al. 2010), same advection offset as the ambient box ("Synchronized").
                                * Ignored with turb_mode NONE (the reference has no model there, :664).              */
    int32_t no_ti_fold;        /* 1: the Crespo-Hernandez added TI is NOT folded into the k of emitted particles     */
    int32_t deficit_model;     /* 0: Gaussian (north_star); 1: super-Gaussian of Blondel & Cathelain (2020);
                                * 2: tabulated eddy-viscosity deficit (Ainslie / DWM; wg_set_deficit_table)          */
    int32_t reserved0_;
    double m0_km1, m0_km2;     /* DWM added-turbulence scaling constants (0.6, 0.35); 0 selects the default          */
    double m0_sg_af, m0_sg_bf, m0_sg_cf; /* super-Gaussian order n(x) = af exp(bf x/D) + cf (3.11, -0.68, 2.41)      */
} wg_config;

typedef struct wg_env_s* wg_handle;

/* which per-env quantity wg_get_info() copies out; names = keys of WindFarmEnv._get_info (:522-555) */
typedef enum wg_info_field {
    WG_INFO_YAW_AGENT = 0,        /* "yaw angles agent"            f32[B,N] */
    WG_INFO_YAW_BASE = 1,         /* "yaw angles base"             f32[B,N] */
    WG_INFO_WS_GLOBAL = 2,        /* "Wind speed Global"           f32[B]   */
    WG_INFO_WD_GLOBAL = 3,        /* "Wind direction Global"       f32[B]   */
    WG_INFO_TI_GLOBAL = 4,        /* "Turbulence intensity"        f32[B]   */
    WG_INFO_WS_TURB = 5,          /* "Wind speed at turbines"      f32[B,N] */
    WG_INFO_WD_TURB = 6,          /* "Wind direction at turbines"  f32[B,N] */
    WG_INFO_POWER_TURB_AGENT = 7, /* "Power pr turbine agent"      f32[B,N] */
    WG_INFO_POWER_TURB_BASE = 8,  /* "Power pr turbine baseline"   f32[B,N] */
    WG_INFO_POWER_AGENT = 9,      /* "Power agent"                 f32[B]   */
    WG_INFO_POWER_BASE = 10,      /* "Power baseline"              f32[B]   */
    WG_INFO_WS_TURB_BASE = 11,    /* "Wind speed at turbines baseline" (u component) f32[B,N] */
    WG_INFO_TURB_X = 12,          /* "Turbine x positions" (flow frame)  f32[B,N] */
    WG_INFO_TURB_Y = 13,          /* "Turbine y positions"         f32[B,N] */
    WG_INFO_TIMESTEP = 14,        /* env.timestep                  i32[B]   */
    WG_INFO_TIME_MAX = 15,        /* env.time_max                  i32[B]   */
    WG_INFO_FS_TIME = 16,         /* fs.time                       f32[B]   */
    WG_INFO_EPISODE = 17,         /* episodes completed so far     i32[B]   */
    WG_INFO_ROTOR_UVW_AGENT = 18, /* fs.windTurbines.rotor_avg_windspeed  f32[B,N,3] */
    WG_INFO_ROTOR_UVW_BASE = 19,  /* fs_baseline ...                      f32[B,N,3] */
    WG_INFO_RATED_POWER = 20,     /* turbine.power(ws) (:700)      f32[B]   */
    WG_INFO_WIND_F64 = 21,        /* (ws, wd, ti) as sampled, in double precision  f64[B,3] */
    /* farm power of the step just taken, BEFORE a same-step autoreset swapped the next episode in: what
     * infos["Power agent"] of a vector env must carry so that RecordEpisodeVals (wrappers/recordEpisodeVals.py:43-46)
     * adds the terminal step's power to the episode that ended.  Equal to POWER_AGENT / POWER_BASE for envs that did
     * not truncate.                                                                                            */
    WG_INFO_STEP_POWER_AGENT = 22, /* f32[B] */
    WG_INFO_STEP_POWER_BASE = 23,  /* f32[B] */
    WG_INFO_BOX_ID = 24            /* index of the turbulence box the running episode drew from the pool  i32[B] */
} wg_info_field;

/* number of floats of the episode-metric vector produced by wg_metrics (the all-reduce payload;
 * distributed form of wrappers/recordEpisodeVals.py:31-64 and of the WindFarmMonitor callback) */
#define WG_N_METRICS 8
enum {
    WG_MET_EP_RETURN_SUM = 0, WG_MET_EP_LENGTH_SUM = 1, WG_MET_EP_MEAN_POWER_SUM = 2, WG_MET_N_EPISODES = 3,
    WG_MET_STEP_REWARD_SUM = 4, WG_MET_FARM_POWER_SUM = 5, WG_MET_BASE_POWER_SUM = 6, WG_MET_N_STEPS = 7
};

const char* wg_last_error(void);
int wg_abi_version(void);

/* Build a batch of farms on HIP device `device`.  Host arrays referenced by cfg are copied.            */
int wg_create(const wg_config* cfg, int device, wg_handle* out);
int wg_destroy(wg_handle h);

/* Sizes derived from the config: observation length O (farm_mes.observed_variables, MesClass.py:610-618)
 * and the per-agent observation length of the PettingZoo facade as actually produced by
 * WindFarmEnvMulti._get_obs_multi (WindEnvMulti.py:79-103).                                          */
int wg_obs_dim(wg_handle h, int* obs_dim, int* obs_dim_multi);
int wg_hist_max(wg_handle h, int* hist_max);

/* Shared frozen-turbulence box for turb_mode BOX / BOX_SHIFT: 3 (u,v,w) planes of nx*ny*nz fp32 (z fastest),
 * unit variance of u; spacing in metres.  The library makes its own interleaved copy (the caller's buffer may be
 * released afterwards).  Must be called before wg_reset in the box modes.                               */
int wg_set_turbulence_box(wg_handle h, const float* box_dev, int nx, int ny, int nz,
                          double dx, double dy, double dz);

/* Evaluation over several turbulence boxes (AgentEval.eval_multiple's `turbbox` loop, AgentEval.py:579-617, through
 * FarmEval.update_tf(path): TF_files = [path], FarmEval.py:86-90): env e uses box ids_host[e] of the pool at every
 * following reset instead of drawing one (a list of one consumes no random number); < 0 = draw; NULL removes the table. */
int wg_set_box_ids(wg_handle h, const int32_t* ids_host);

/* Isotropic box of the wake-added turbulence (wg_config.added_turbulence = 1): 3 planes like above, unit variance;
 * the reference's default is hipersim's L = 5 m, Gamma = 0, 128^3 cells of 3 m
 * (examples/longer_steps_example.py:153).  The library keeps an interleaved copy.  Must be set before wg_reset.   */
int wg_set_added_turbulence_box(wg_handle h, const float* box_dev, int nx, int ny, int nz,
                                double dx, double dy, double dz);

/* Pool of n_boxes frozen boxes of equal shape for turb_mode BOX_POOL (turbtype "MannLoad": one of the TF_* files per
 * reset, Wind_Farm_Env.py:611-618).  boxes_dev: HOST array of n_boxes DEVICE pointers, each 3 planes like above.  The
 * library keeps its own interleaved copies (n_boxes x 1.07 GB for the reference's 2048 x 512 x 64 boxes — sized for
 * 288 GB of HBM).  wg_set_turbulence_box is the pool of one.                                                */
int wg_set_turbulence_boxes(wg_handle h, const float* const* boxes_dev, int n_boxes, int nx, int ny, int nz,
                            double dx, double dy, double dz);

/* Evaluation sweeps (FarmEval.set_wind_vals for a whole batch, FarmEval.py:63-78): fix the wind conditions of
 * env b to wind_host[b] = (ws, wd, ti) for every following episode; NaN entries keep the sampled value.  The env's
 * generator still consumes its draws, so seeds stay aligned.  NULL removes the override.                  */
int wg_set_wind(wg_handle h, const double* wind_host /*[B,3]*/);

/* The same with a caller-owned DEVICE buffer f64[B,3] that is read (stream-ordered) whenever an episode of env b is
 * initialised — the caller may rewrite it between steps.  This is how site-based sampling (`sample_site`,
 * WindFarmEnv._set_windconditions / _sample_site, Wind_Farm_Env.py:569-594: wd from the sector frequencies, ws from
 * the sector's Weibull, both clipped to the env's ranges; TI stays uniform = NaN here) is fed to a running batch
 * without host synchronisation.  NULL removes the override.                                                 */
int wg_set_wind_device(wg_handle h, const double* wind_dev /*[B,3]*/);

/* Test hook ("replay mode"): replace the flow physics of both farms by scripted tables so that the glue can
 * be checked against golden vectors recorded from the reference.  uvw_dev: f32[F,T,B,N,3], power_dev:
 * f32[F,T,B,N]; every flow sub-step of farm f in env b consumes row cursor[f,b]++ .  NULL disables.     */
int wg_set_flow_script(wg_handle h, const float* uvw_dev, const float* power_dev, int n_rows);

/* reset(): WindFarmEnv.reset (:680-802) for the envs whose mask byte is non-zero (NULL = all).
 * seeds: host array [B] of uint64 or NULL.  seeds[b] == UINT64_MAX keeps env b's generator running
 * (gymnasium reset(seed=None)); otherwise the env's PCG64 generator is re-seeded exactly like
 * gymnasium's np_random (np.random.default_rng(seed)).  Outputs may be NULL.
 * With autoreset on, a reset initialises BOTH episode contexts of a masked env — the episode that starts now and the
 * look-ahead episode that is developed in the background — so it consumes two episodes' worth of draws from the env's
 * generator: episode k of an env always uses draws k of its stream (one draw set per episode, in order), but an explicit
 * reset(seed=None) in the middle of an autoreset run discards the look-ahead episode that was already drawn, i.e. the
 * stream then runs one episode ahead of a reference env that was reset at the same moments.  A wind override set after
 * a reset is seen from the episode after the look-ahead one.  The sensor-noise stream is keyed by (env seed, episode
 * index, push index): an explicit reset starts a new episode index, so noise sequences are not replayed.   */
int wg_reset(wg_handle h, const uint8_t* env_mask_host, const uint64_t* seeds_host,
             float* obs_dev /*[B,O]*/, void* stream);

/* step(): WindFarmEnv.step (:920-1034) for all envs.  actions_dev f32[B,N] in [-1,1].
 * obs_dev f32[B,O] (after autoreset: first observation of the next episode), reward_dev f32[B],
 * truncated_dev u8[B]; final_obs_dev f32[B,O] or NULL receives the last observation of the episode that
 * ended (== obs for envs that did not truncate).  Asynchronous on `stream`.                            */
int wg_step(wg_handle h, const float* actions_dev, float* obs_dev, float* reward_dev,
            uint8_t* truncated_dev, float* final_obs_dev, void* stream);

/* step() as ONE graph launch: with enable != 0 the kernels of a step are captured once per
 * distinct set of I/O pointers into a HIP graph (at most 32 sets are cached, least recently used evicted) and every
 * following wg_step is a single hipGraphLaunch on the caller's stream.  Results are identical to the direct launches.
 * Setters that change kernel arguments (turbulence box, wind override, flow script, obs-multi buffer) drop the cached
 * graphs.  Default: off (two direct launches; measured faster on the host for a two-kernel step — DESIGN.md §4.4);
 * the environment variable WG_STEP_GRAPH=1 switches it on at wg_create.                                   */
int wg_set_step_graph(wg_handle h, int enable);

/* step() is asynchronous, so the errors the reference raises inside step() (Exception("NaN Power"), stepping
 * a torn-down env) are latched in a sticky device word.  wg_check synchronises `stream` and returns it
 * (0, WG_ERR_NAN_POWER, WG_ERR_STATE or WG_ERR_RANGE — the conditions are latched independently and the most serious one
 * is reported, in that order); a wg_reset of the whole batch clears it.                                                     */
int wg_check(wg_handle h, void* stream);

/* Per-agent observations of the PettingZoo facade for the current state: f32[B,N,obs_dim_multi].      */
int wg_obs_multi(wg_handle h, float* obs_dev, void* stream);

/* The same, fused into the step: once a caller-owned buffer f32[B,N,obs_dim_multi] is registered, every following
 * wg_step / wg_reset also writes the per-agent observations there (the values wg_obs_multi would return right
 * after the call) — WindFarmEnvMulti.step calls _get_obs_multi every step (WindEnvMulti.py:188-227).  NULL stops. */
int wg_set_obs_multi_buffer(wg_handle h, float* obs_multi_dev);

/* Unscaled, unclipped sensor values of the running episodes in the layout of the observation: f32[B,O]
 * (farm_measurements.get_*_turb() / get_*_farm(), the "... measured" entries of _get_info :529-537).   */
int wg_get_measurements(wg_handle h, float* out_dev, void* stream);

/* Flow-field view of ONE farm (0 = agent, 1 = baseline) of ONE env: (u, v, w) at the nx x ny points
 * (x_dev[i], y_dev[j], z) of the flow frame -> uvw_dev f32[3, nx, ny].  Replaces
 * fs.get_windspeed(XYView(z=hub_height, x=a, y=b), include_wakes=True) behind WindFarmEnv._render_frame /
 * init_render (Wind_Farm_Env.py:1040-1083, :464-476; AgentEval.py:220-228).                               */
int wg_get_windspeed(wg_handle h, int env, int farm, const float* x_dev, int nx, const float* y_dev, int ny,
                     float z, int include_wakes, float* uvw_dev, void* stream);

/* Lazy info dict: copy one field to out_dev (dtype/shape per wg_info_field).                           */
int wg_get_info(wg_handle h, wg_info_field field, void* out_dev, void* stream);

/* Episode-metric partial sums of this shard since the last call with reset_after != 0: f32[WG_N_METRICS]
 * in device memory (ready for one RCCL all-reduce(sum)).                                              */
int wg_metrics(wg_handle h, float* out_dev, int reset_after, void* stream);

/* Checkpoint / golden replay: serialise the full device state.  Call with blob_host == NULL to get size.  The blob
 * carries a header (magic, ABI version, batch geometry); wg_set_state rejects a blob taken from a differently
 * configured handle.  The wind override of wg_set_wind is configuration, not state (re-apply it after a restore). */
int wg_get_state(wg_handle h, void* blob_host, size_t* size);
int wg_set_state(wg_handle h, const void* blob_host, size_t size);

/* HIP-event timing of the step kernels on the stream they were launched on: average milliseconds per
 * launch of the dominant flow kernel and of the glue kernel since the last call, and the average number of
 * farm flow-steps one flow launch executed (live farms + background episode development) — the unit
 * count behind bench.py's roofline; particles_per_launch = wake particles the advection passes actually streamed
 * (chain pruning: particles behind the last turbine are not touched).  enable = 0 stops; enable = n >= 1 records events around every n-th
 * wg_step (an event pair per launch costs a few percent of a ~200 us step).                            */
int wg_kernel_timing(wg_handle h, int enable, double* flow_ms_avg, double* glue_ms_avg, int* n_launches,
                     double* flow_steps_per_launch, double* particles_per_launch);

/* deficit_model 2: the wake-deficit table the flow kernels sample instead of the Gaussian — what the reference's
 * particleDeficitGenerator=jDWMAinslieGenerator() (Wind_Farm_Env.py:706, :774) solves inside DYNAMIKS, solved on the host
 * (windgym_amd/ainslie.py restates the published DWM eddy-viscosity model).  table_dev: f32[n_ct][n_ti][n_x][n_r], the deficit
 * fraction 1 - U / U0 at Ct uniform in [ct0, ct1], ambient TI log-uniform in [ti0, ti1], x / D uniform in [0, x_max_D],
 * r / R uniform in [0, r_max_R] (linear between nodes; 0 from half a node inside r_max_R on).  Borrowed: the buffer must outlive the handle.  Before wg_reset.          */
int wg_set_deficit_table(wg_handle h, const float* table_dev, int n_ct, double ct0, double ct1, int n_ti, double ti0, double ti1,
                         int n_x, double x_max_D, int n_r, double r_max_R);

/* Mann spectral-tensor turbulence box generated on the device — MannTurbulenceField.generate(alphaepsilon, L, Gamma, Nxyz,
 * dxyz, seed) of hipersim / dynamiks behind turbtype "MannFixed" / "MannGenerate" (Wind_Farm_Env.py:624-637, :649-658;
 * tests/test_basics.py:38-45).  box_dev: f32[3][nx][ny][nz] (z fastest: what wg_set_turbulence_box takes), normalised to
 * unit standard deviation of u (the reference rescales with scale_TI(TI, U); the kernels multiply by TI * U per env).
 * noise_dev: optional complex white noise f32[3][nx][ny][nz][2] (E|n|^2 = 1) to use instead of the built-in Philox stream
 * keyed by `seed` — lets a test feed the identical noise to a CPU restatement.  Sheared von Karman tensor per wave-number
 * cell in a HIP kernel, three in-place inverse C2C transforms in hipFFT, synchronous on `stream`.  No handle needed.   */
int wg_generate_mann_box(int device, float* box_dev, int nx, int ny, int nz, double dx, double dy, double dz,
                         double alphaepsilon, double L, double Gamma, uint64_t seed, const float* noise_dev, void* stream);

/* Host only: the eddy-lifetime factor beta(kL) = Gamma (kL)^(-2/3) / sqrt(2F1(1/3, 17/6; 4/3; -(kL)^-2)) (Mann 1998) on n
 * log-spaced points kL = 10^[log10_lo, log10_hi] — the table wg_generate_mann_box interpolates (4096 points over
 * [1e-6, 1e6]); exported so that tests pin it against an independent 2F1.                                               */
int wg_mann_beta_table(double Gamma, int n, double log10_lo, double log10_hi, double* beta_out);

/* Steady-state farm power for a batch of cases — the inner loop of PyWakeAgent.yaw_optimizer_srf_vect
 * (WindGym/Agents/PyWakeAgent.py:144-288: every Serial-Refine step evaluates the farm power of yaw_n candidate yaw vectors
 * per wind condition): power_dev f32[n_cases][n_turb] (W) for ws / wd / ti f32[n_cases] and yaw_dev f32[n_cases][n_turb]
 * (degrees, flow frame).  Layout, turbine table, rotor points and model constants are the handle's.
 * model 0: the steady state of the env's own flow model (what wg_step converges to under constant yaws) — the Gaussian
 *          deficit with wake-TI folding; WG_ERR_UNSUPPORTED on a handle created with another deficit_model / no_ti_fold;
 * model 1: the reference agent's py_wake model restated from the publications (Blondel & Cathelain 2020 super-Gaussian
 *          at the rotor centre, linear superposition, Jimenez deflection, Ct cos^2(yaw)).  One kernel launch (k_steady).   */
int wg_steady_power(wg_handle h, int model, int n_cases, const float* ws_dev, const float* wd_dev, const float* ti_dev,
                    const float* yaw_dev, float* power_dev, void* stream);

/* Rotor points at which one flow launch looked the wake-added turbulence box up (8 corners x (u, v, w) = 96 bytes each:
 * only the rotors of targets with a candidate source wake do), averaged over the window the LAST wg_kernel_timing call
 * closed — the a7 term of bench.py's algorithmic bytes.  0 without wg_config.added_turbulence.
 * Steady inflow on the one-wave-per-env kernel (wg_flow_variant: 2; no such lookups exist there): the same word counts the
 * wake particles the advection passes actually touched per launch — moving chains whole, resting chains their new particles —
 * the numerator of bench.py's roofline.frac_touched.                                                                          */
int wg_added_lookups(wg_handle h, double* rotor_points_per_launch);

/* Algorithmic HBM bytes one wg_step() moves (the figure bench.py's roofline uses).       */
int wg_algorithmic_bytes(wg_handle h, double* bytes_per_step);

/* Which flow kernel the handle launches (diagnostics / tests; chosen in wg_create from the farm geometry and the inflow; the
 * hooks WG_FLOW_BLOCK / WG_FLOW_RES / WG_FLOW_ENV / WG_ENV_WPE / WG_STEP_FUSED override it, honoured only
 * with WG_DEBUG_HOOKS=1): threads per wave-group, 1 = compact per-turbine rings with pair-major deficit phases, and the farm
 * slots one wave serves: 0 = one (k_flow), 2 = every slot of an env / of one of its
 * contexts (k_flow_env: one or two waves per env; with the lean glue, wg_step is then ONE kernel launch).                     */
int wg_flow_variant(wg_handle h, int* block, int* compact, int* duo);

#ifdef __cplusplus
}
#endif
#endif /* WINDGYM_HIP_H */
