// wg_api.hip — host side of libwindgym_hip.so: the C ABI declared in include/windgym_hip.h.
// Owns the device state (allocated once in wg_create; nothing is allocated on the step path) and launches
// the kernels of wg_kernels.hip on the caller's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "wg_state.h"
#include "wg_flow.h"
#include "wg_steady.h"

extern "C" {
void wg_launch_flow(const FlowP*, const FlowPtrs*, int, const float*, const uint8_t*, int, hipStream_t);
void wg_launch_flow_env(const FlowP*, const FlowPtrs*, int, const float*, const uint8_t*, int, hipStream_t);
void wg_launch_step_env(const FlowP*, const FlowPtrs*, const WgParams*, const WgPtrs*, const float*, float*, float*, uint8_t*, float*, hipStream_t);
void wg_launch_step_envb(const FlowP*, const FlowPtrs*, const WgParams*, const WgPtrs*, const float*, float*, float*, uint8_t*, float*, hipStream_t);
void wg_launch_glue(const WgParams*, const WgPtrs*, int, const uint8_t*, float*, float*, uint8_t*, float*, hipStream_t, const WgParams*, const WgPtrs*);
void wg_launch_init(const WgParams*, const WgPtrs*, const uint8_t*, const uint64_t*, hipStream_t);
void wg_launch_create(const WgParams*, const WgPtrs*, hipStream_t);
void wg_launch_obs_multi(const WgParams*, const WgPtrs*, float*, hipStream_t);
void wg_launch_info(const WgParams*, const WgPtrs*, int, void*, hipStream_t);
void wg_launch_metrics(const WgParams*, const WgPtrs*, float*, int, hipStream_t);
void wg_launch_box_repack(const float*, void*, int, int, int, hipStream_t);
void wg_launch_measurements(const WgParams*, const WgPtrs*, float*, hipStream_t);
void wg_launch_box_coarsen(const void*, void*, int, int, int, hipStream_t);
void wg_launch_box_stencil(const void*, void*, int, int, int, hipStream_t);
void wg_launch_windspeed(const FlowP*, const FlowPtrs*, int, int, const float*, int, const float*, int, float, int, float*, hipStream_t);
void wg_launch_unready(const WgParams*, const WgPtrs*, const uint8_t*, int*, hipStream_t);
void wg_launch_steady(const void*, const float*, const float*, const float*, const float*, float*, hipStream_t);
}

// Measurement / test hooks (environment variables that select kernel variants): honoured ONLY with WG_DEBUG_HOOKS=1 — a
// stray WG_FLOW_BLOCK in a user's shell must not silently change the kernel — and announced once on stderr when used.
// (binding.py guards its own hooks, WG_LIB / WG_NOCHECK, the same way; tests/conftest.py sets the switch)
static const char* wg_hook(const char* name) {
    static const bool on = [] { const char* v = getenv("WG_DEBUG_HOOKS"); return v && v[0] == '1' && v[1] == 0; }();
    if (!on) return nullptr;
    const char* v = getenv(name);
    if (v) {
        static std::string seen;
        static std::mutex seen_mu;      // (wg_create may run on several threads at once: one handle per device)
        std::lock_guard<std::mutex> lock(seen_mu);
        const std::string tag = std::string(name) + "=" + v + ";";
        if (seen.find(tag) == std::string::npos) {
            seen += tag;
            fprintf(stderr, "[windgym] WG_DEBUG_HOOKS: %s=%s\n", name, v);
        }
    }
    return v;
}

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
// (other translation units of the library report through the same thread-local message: wg_mann.hip)
extern "C" int wg_set_last_error_(int code, const char* msg) { return fail(code, msg ? msg : ""); }
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t _e = (x);                                                                        \
        if (_e != hipSuccess)                                                                       \
            return fail(WG_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(_e));                \
    } while (0)

struct Alloc {
    void* ptr;
    size_t bytes;
};

struct wg_env_s {
    WgParams p;
    WgPtrs d;
    FlowP fp;
    FlowPtrs fd;
    unsigned long long flow_steps_mark = 0, particles_mark = 0, added_mark = 0;
    double added_per_launch = 0.0;  // window closed by the last wg_kernel_timing
    long n_step_launches = 0;
    void* box4 = nullptr;            // interleaved copy of the caller's turbulence box (owned)
    void* box4c = nullptr;           // block-averaged copy for the particle lookups (owned)
    int* box_ids_dev = nullptr;      // wg_set_box_ids
    void* abox4 = nullptr;           // interleaved isotropic box of the wake-added turbulence (owned)
    void* box8 = nullptr;            // stencil records of the box pool / of the wake-added box (owned; FlowPtrs::box8, k_flow_envb)
    void* abox8 = nullptr;
    int added = 0, no_ti_fold = 0, deficit_model = 0;   // wg_config model options
    double km1 = 0.6, km2 = 0.35, sg_af = 3.11, sg_bf = -0.68, sg_cf = 2.41;
    double* wind_dev = nullptr;      // per-env wind override (wg_set_wind)
    int device;
    std::vector<Alloc> allocs;      // everything owned by the handle (state blob = allocs flagged `state`)
    std::vector<size_t> state_idx;  // indices into allocs that make up the serialisable state
    uint8_t* mask_dev = nullptr;
    uint64_t* seeds_dev = nullptr;
    int reset_launches = 0;         // upper bound of RESET-mode flow launches needed to develop any episode
    int reset_chunk = 32;
    // timing
    bool timing = false;
    int timing_period = 1, timing_phase = 0;   // record events around every `period`-th step() only
    std::vector<hipEvent_t> ev;     // pairs (start, stop) around flow launches, then glue launches
    std::vector<int> ev_kind;
    size_t ev_used = 0;
    double alg_bytes = 0;
    // step() as ONE graph launch: the launches of a step are captured once per distinct set of I/O pointers on an
    // internal stream and replayed on the caller's stream (wg_set_step_graph)
    struct StepGraph {
        const void* key[5];
        hipGraph_t graph;
        hipGraphExec_t exec;
        unsigned long long last_use;
    };
    std::vector<StepGraph> graphs;
    hipStream_t cap_stream = nullptr;
    int graph_mode = 0;
    unsigned long long graph_clock = 0;
    int* unready_dev = nullptr;     // wg_reset: live slots of the masked envs that are not developed yet
    WgParams* p_dev = nullptr;      // device copies of p / d, read by the episode-initialisation path at the head of k_flow
    WgPtrs* d_dev = nullptr;
};

static const size_t WG_MAX_STEP_GRAPHS = 32;   // distinct (actions, obs, reward, truncated, final_obs) pointer sets cached
static const size_t WG_MAX_TIMING_EVENTS = 4096;

static void drop_step_graphs(wg_env_s* h) {
    // a cached exec may still be in flight on the caller's stream: destroying it under a running launch is undefined
    if (!h->graphs.empty()) { hipSetDevice(h->device); hipDeviceSynchronize(); }
    for (auto& g : h->graphs) {
        hipGraphExecDestroy(g.exec);
        hipGraphDestroy(g.graph);
    }
    h->graphs.clear();
}

// the device copies of the parameter blocks follow every host-side change (setters are rare, synchronous calls)
static int sync_dev_params(wg_env_s* h) {
    hipError_t e = hipMemcpy(h->p_dev, &h->p, sizeof(WgParams), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_dev, &h->d, sizeof(WgPtrs), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(WG_ERR_HIP, std::string("hipMemcpy(params): ") + hipGetErrorString(e));
    return 0;
}

static int use_device(wg_env_s* h) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != h->device) {
        hipError_t e = hipSetDevice(h->device);
        if (e != hipSuccess) return fail(WG_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    }
    return 0;
}

template <typename T>
static int dev_alloc(wg_env_s* h, T** out, size_t n, bool state, bool zero = true) {
    size_t bytes = sizeof(T) * (n ? n : 1);
    void* ptr = nullptr;
    hipError_t e = hipMalloc(&ptr, bytes);
    if (e != hipSuccess) return fail(WG_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    if (zero) {
        e = hipMemset(ptr, 0, bytes);
        if (e != hipSuccess) return fail(WG_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(e));
    }
    h->allocs.push_back({ptr, bytes});
    if (state) h->state_idx.push_back(h->allocs.size() - 1);
    *out = (T*)ptr;
    return 0;
}
template <typename T, typename S>
static int dev_upload(wg_env_s* h, const T** out, const S* src, size_t n) {
    std::vector<T> tmp(n ? n : 1);
    for (size_t i = 0; i < n; ++i) tmp[i] = (T)src[i];
    T* p = nullptr;
    int rc = dev_alloc(h, &p, n, false, false);
    if (rc) return rc;
    HIPCHK(hipMemcpy(p, tmp.data(), sizeof(T) * (n ? n : 1), hipMemcpyHostToDevice));
    *out = p;
    return 0;
}

static double defd(double v, double dflt) { return v == 0.0 ? dflt : v; }

static int ch_count(const wg_channel& c, int on) { return (c.current && on) + (c.rolling_mean && on) * c.history_n; }

extern "C" const char* wg_last_error(void) { return g_err.c_str(); }
extern "C" int wg_abi_version(void) { return WG_ABI_VERSION; }

extern "C" int wg_create(const wg_config* c, int device, wg_handle* out) {
    if (!c || !out) return fail(WG_ERR_INVALID, "null argument");
    if (c->abi_version != WG_ABI_VERSION) return fail(WG_ERR_INVALID, "abi_version mismatch");
    if (c->n_envs < 1 || c->n_turb < 1) return fail(WG_ERR_INVALID, "n_envs and n_turb must be >= 1");
    if (c->n_farms != 1 && c->n_farms != 2) return fail(WG_ERR_INVALID, "n_farms must be 1 or 2");
    if (c->k_sub < 1) return fail(WG_ERR_INVALID, "dt_env must be a multiple of dt_sim");
    if (c->n_particles < 8 || c->n_particles % 4) return fail(WG_ERR_INVALID, "n_particles must be a multiple of 4 (>= 8)");
    if (c->n_rotor_pts < 1 || c->n_rotor_pts > 64) return fail(WG_ERR_INVALID, "n_rotor_pts must be in [1, 64]");
    if (c->n_turb > 32 * WG_MASK_WORDS) return fail(WG_ERR_UNSUPPORTED, "n_turb > 128 is not supported by this build");
    if ((long long)c->n_turb * c->n_particles >= (1ll << 22) || c->n_particles > 16384)
        return fail(WG_ERR_UNSUPPORTED, "n_turb * n_particles must stay below 2^22 (and n_particles <= 16384)");
    if (!c->x_pos || !c->y_pos || !c->rotor_dy || !c->rotor_dz || !c->tab_ws || !c->tab_power || !c->tab_ct || c->n_tab < 2)
        return fail(WG_ERR_INVALID, "layout / rotor points / turbine table missing");
    if (c->action_method != WG_ACT_YAW && c->action_method != WG_ACT_WIND)
        return fail(WG_ERR_UNSUPPORTED, "The ActionMethod must be yaw or wind (absolute is not implemented)");
    if (c->reward_mode < 0 || c->reward_mode > 3)
        return fail(WG_ERR_INVALID, "The Power_reward must be either Baseline, Power_avg, None or Power_diff");
    if (c->reward_mode == WG_REW_POWER_DIFF && c->power_avg < 40)
        return fail(WG_ERR_INVALID, "The Power_avg must be larger then 40 for the Power_diff reward");
    if (c->reward_mode == WG_REW_BASELINE && c->n_farms != 2)
        return fail(WG_ERR_INVALID, "Baseline reward needs the baseline farm (n_farms = 2)");
    if (c->power_avg < 1) return fail(WG_ERR_INVALID, "Power_avg must be >= 1");
    if (c->turb_mode < WG_TURB_NONE || c->turb_mode > WG_TURB_BOX_POOL) return fail(WG_ERR_INVALID, "Invalid turbulence type specified");
    for (int i = 0; i < WG_N_CH; ++i)
        if (c->ch[i].history_len < 1 || c->ch[i].window_len < 1 || c->ch[i].history_n < 1)
            return fail(WG_ERR_INVALID, "sensor history/window lengths must be >= 1");
    {
        // the 16-bit emission record (wg_flow.hip): k = ka TI_loc + kb <= 0.25 and |hv| <= |hill| u <= 16 m/s must hold
        // over the sampling ranges (TI_loc^2 <= ti_max^2 + 0.34^2, the Crespo-Hernandez bound at Ct = 0.96); the
        // kernel also flags any violation at run time (wind overrides): WG_ERR_RANGE
        const double ka = defd(c->m0_ka, 0.38), kb = defd(c->m0_kb, 0.004), hill = defd(c->m0_hill, 0.4);
        const double tia = c->no_ti_fold ? 0.0 : 0.466 * defd(c->m0_ti_a, 0.73);
        const double k_hi = ka * std::sqrt(c->ti_max * c->ti_max + tia * tia) + kb;
        if (k_hi > 0.25 || std::fabs(hill) * c->ws_max > 16.0)
            return fail(WG_ERR_INVALID, "wind ranges exceed the 16-bit emission record: need ka*sqrt(TI_max^2 + 0.34^2) + kb <= 0.25 "
                                            "(TI_max <= 0.55 with the default constants) and |hill| * ws_max <= 16 m/s");
    }
    if (c->deficit_model < 0 || c->deficit_model > 2)
        return fail(WG_ERR_INVALID, "deficit_model must be 0 (Gaussian), 1 (super-Gaussian) or 2 (tabulated eddy-viscosity deficit)");
    HIPCHK(hipSetDevice(device));
    wg_env_s* h = new wg_env_s();
    h->device = device;
    h->added = (c->added_turbulence != 0 && c->turb_mode != WG_TURB_NONE) ? 1 : 0;
    h->no_ti_fold = c->no_ti_fold != 0; h->deficit_model = c->deficit_model;
    h->km1 = defd(c->m0_km1, 0.6); h->km2 = defd(c->m0_km2, 0.35);
    h->sg_af = defd(c->m0_sg_af, 3.11); h->sg_bf = defd(c->m0_sg_bf, -0.68); h->sg_cf = defd(c->m0_sg_cf, 2.41);
    WgParams& p = h->p;
    memset(&p, 0, sizeof(p));
    p.B = c->n_envs; p.N = c->n_turb; p.F = c->n_farms; p.K = c->k_sub; p.P = c->n_particles; p.S = c->n_rotor_pts;
    p.NP = p.N * p.P;
    p.dt = (float)c->dt_sim; p.dt_d = c->dt_sim;
    p.D = (float)c->rotor_diameter; p.D_d = c->rotor_diameter; p.inv_D = 1.0f / (float)c->rotor_diameter;
    p.hub = (float)c->hub_height; p.hub_d = c->hub_height;
    p.dpart = c->d_particle * c->rotor_diameter;
    p.yaw_min = (float)c->yaw_min; p.yaw_max = (float)c->yaw_max; p.yaw_step = (float)c->yaw_step;
    p.yaw_min_d = c->yaw_min; p.yaw_max_d = c->yaw_max; p.yaw_step_d = c->yaw_step; p.yaw_start = c->yaw_start;
    p.action_method = c->action_method; p.base_controller = c->base_controller; p.yaw_init = c->yaw_init;
    p.has_yaw_defined = c->yaw_defined != nullptr;
    p.ws_min = c->ws_min; p.ws_max = c->ws_max; p.ti_min = c->ti_min; p.ti_max = c->ti_max;
    p.wd_min = c->wd_min; p.wd_max = c->wd_max; p.n_passthrough = c->n_passthrough;
    p.never_truncate = c->never_truncate;
    p.full_chains = c->full_chains;
    for (int i = 0; i < WG_N_CH; ++i) p.ch[i] = c->ch[i];
    p.turb_on[WG_CH_WS] = c->turb_ws; p.turb_on[WG_CH_WD] = c->turb_wd; p.turb_on[WG_CH_YAW] = 1;
    p.turb_on[WG_CH_POWER] = c->turb_power;
    p.farm_on[WG_CH_WS] = c->farm_ws; p.farm_on[WG_CH_WD] = c->farm_wd; p.farm_on[WG_CH_YAW] = 0;
    p.farm_on[WG_CH_POWER] = c->farm_power;
    p.turb_ti = c->turb_ti; p.farm_ti = c->farm_ti;
    const double mn[WG_N_CH] = {c->ws_scale_min, c->wd_scale_min, c->yaw_min, 0.0};
    const double mx[WG_N_CH] = {c->ws_scale_max, c->wd_scale_max, c->yaw_max, c->power_max};
    for (int i = 0; i < WG_N_CH; ++i) { p.sc_min[i] = (float)mn[i]; p.sc_rng[i] = (float)(mx[i] - mn[i]); }
    p.sc_rng_farm_power = (float)(c->power_max * c->n_turb);
    p.ti_min_f = (float)c->ti_scale_min; p.ti_rng_f = (float)(c->ti_scale_max - c->ti_scale_min);
    p.noise = c->noise;
    for (int i = 0; i < WG_N_CH; ++i) p.noise_sigma[i] = (float)c->noise_sigma[i];
    p.reward_mode = c->reward_mode; p.power_avg = c->power_avg; p.power_scaling = c->power_scaling;
    p.action_penalty = c->action_penalty; p.penalty_type = c->penalty_type;
    p.fill_a = c->fill_steps_agent; p.fill_b = c->fill_steps_base; p.autoreset = c->autoreset;
    p.extra_inc = c->extra_timestep_inc; p.turb_mode = c->turb_mode;
    p.ka = (float)defd(c->m0_ka, 0.38); p.kb = (float)defd(c->m0_kb, 0.004); p.eps0 = (float)defd(c->m0_eps, 0.2);
    p.hill = (float)defd(c->m0_hill, 0.4);
    p.tia = (float)defd(c->m0_ti_a, 0.73); p.tib = (float)defd(c->m0_ti_b, 0.8325);
    p.tic = (float)defd(c->m0_ti_c, 0.0325); p.tid = (float)defd(c->m0_ti_d, -0.32);
    p.fc_scale = defd(c->m0_fc_scale, 2.0);
    p.n_tab = c->n_tab;
    p.turb_obs = ch_count(c->ch[WG_CH_WS], c->turb_ws) + ch_count(c->ch[WG_CH_WD], c->turb_wd) +
                 ch_count(c->ch[WG_CH_YAW], 1) + (c->turb_ti ? 1 : 0) + ch_count(c->ch[WG_CH_POWER], c->turb_power);
    p.farm_obs = ch_count(c->ch[WG_CH_WS], c->farm_ws) + ch_count(c->ch[WG_CH_WD], c->farm_wd) +
                 (c->farm_ti ? 1 : 0) + ch_count(c->ch[WG_CH_POWER], c->farm_power);
    p.obs_dim = p.turb_obs * p.N + p.farm_obs;
    p.obs_dim_multi = p.turb_obs + p.farm_obs;
    p.hist_max = std::max(c->ch[WG_CH_WS].history_len, std::max(c->ch[WG_CH_WD].history_len, c->ch[WG_CH_YAW].history_len));
    // what k_glue stages of the turbine rings (the farm rings are tiny and always staged): a channel read through
    // windows (rolling means) or through the TI of its whole deque needs the whole ring, a channel read through its
    // `current` value only needs the newest sample, anything else is not part of the observation.  The per-agent
    // observation (PettingZoo facade) reads the same turbine-level channels.
    for (int i = 0; i < WG_N_CH; ++i) {
        const bool on = p.turb_on[i] != 0;
        int st = 0;
        if (on && c->ch[i].current) st = 1;
        if (on && c->ch[i].rolling_mean) st = 2;
        if (i == WG_CH_WS && (c->turb_ti || c->farm_ti)) st = 2;
        p.stage_ch[i] = st;
    }

    // Sums mode: every rolling mean the observation reads has history_N = 1, i.e. it is the mean of the newest
    // min(window, history, pushed) samples (Mes.get_measurements' first window, MesClass.py:85-91) — a window that
    // slides by exactly one sample per push.  The lean glue kernel (k_glue_lean) then keeps the window sums of the live
    // episode (and the ws deque's sum / sum of squares for calc_TI) as running sums, S += newest - leaving in exact double
    // arithmetic, instead of staging and re-summing the rings every step (Env1.yaml: 4 ring samples per turbine instead of
    // 35).  The flow kernels are untouched: they push into rings that hold ONE sample more than the deque (ring_cap = H +
    // 1), so that the sample leaving a window as long as the deque is still there when the glue runs.  Configurations with
    // several windows per channel (2turb.yaml: history_N = 100) keep the ring-staging kernel.  WG_SUMS=0 forces that path
    // (tests compare the two).
    {
        bool ok = true;
        for (int i = 0; i < WG_N_CH; ++i) {
            const bool used = p.turb_on[i] || (i != WG_CH_YAW && p.farm_on[i]);
            if (used && c->ch[i].rolling_mean && c->ch[i].history_n != 1) ok = false;
        }
        if (const char* ev = wg_hook("WG_SUMS")) if (atoi(ev) == 0) ok = false;
        p.sums_mode = ok ? 1 : 0;
        for (int i = 0; i < WG_N_CH; ++i) {
            p.ring_cap[i] = c->ch[i].history_len + (ok ? 1 : 0);
            p.ring_magic[i] = (unsigned)((1ull << 32) / (unsigned long long)p.ring_cap[i]) + 1u;
        }
        if (ok) {
            for (int i = 0; i < WG_N_CH; ++i) {
                p.sum_w[i] = std::min(c->ch[i].window_len, c->ch[i].history_len);
                if (p.turb_on[i] && c->ch[i].rolling_mean) p.sum_mask_t |= 1u << i;
                if (p.turb_on[i] && c->ch[i].current) p.cur_mask_t |= 1u << i;
                if (i != WG_CH_YAW && p.farm_on[i] && c->ch[i].rolling_mean) p.sum_mask_f |= 1u << i;
                if (i != WG_CH_YAW && p.farm_on[i] && c->ch[i].current) p.cur_mask_f |= 1u << i;
            }
            p.sum_w[WG_SUM_TI1] = p.sum_w[WG_SUM_TI2] = c->ch[WG_CH_WS].history_len;
            if (c->turb_ti || c->farm_ti) p.sum_mask_t |= (1u << WG_SUM_TI1) | (1u << WG_SUM_TI2);
            if (c->farm_ti) p.sum_mask_f |= (1u << WG_SUM_TI1) | (1u << WG_SUM_TI2);
        }
        // one turbine's block of the observation (wg_obs_turbine)
        WgObsCfg& oc = p.oc;
        oc.cur_mask = p.cur_mask_t; oc.rol_mask = p.sum_mask_t & 0xfu;
        oc.turb_ti = c->turb_ti; oc.turb_obs = p.turb_obs; oc.obs_dim = p.obs_dim; oc.obs_dim_multi = p.obs_dim_multi;
        for (int i = 0; i < WG_N_CH; ++i) {
            oc.hlen[i] = c->ch[i].history_len; oc.wlen[i] = c->ch[i].window_len;
            oc.mn[i] = p.sc_min[i]; oc.inv_rng[i] = 1.0f / p.sc_rng[i];
            oc.inv_w[i] = 1.0 / (double)std::min(c->ch[i].window_len, c->ch[i].history_len);
        }
        oc.ti_mn = p.ti_min_f; oc.inv_ti_rng = 1.0f / p.ti_rng_f;
    }
    {
        int off = 0, foff = 0;
        for (int i = 0; i < WG_N_CH; ++i) {
            p.ring_off[i] = off; off += p.N * p.ring_cap[i];
            p.fring_off[i] = foff; foff += p.ring_cap[i];
        }
        p.ring_stride = off; p.fring_stride = foff;
    }

    WgPtrs& d = h->d;
    memset(&d, 0, sizeof(d));
    const size_t n_slots = (size_t)p.B * 2 * p.F, n_ctx = (size_t)p.B * 2;
    int rc = 0;
#define A(field, n, st) if (!rc) rc = dev_alloc(h, &d.field, (n), (st))
    // Particle blocks of consecutive farm slots are `pstride` floats apart: N x P rounded up to an ODD multiple of
    // 256 bytes.  The compact rings of a farm use only the front of its block; an odd multiple of the 256-byte granule
    // keeps the touched parts from aliasing onto the same residues of the HBM channel interleave (measured: no
    // difference on cfg2 either way; kept as the safe choice).
    size_t pstride = ((size_t)p.NP + 63) / 64;      // in 256-byte granules
    if (pstride % 2 == 0) pstride += 1;
    pstride *= 64;
    if (const char* ev = wg_hook("WG_PSTRIDE_PAD")) pstride = (size_t)p.NP + (size_t)atoi(ev);
    h->fp.pstride = (int)pstride;
    A(py, n_slots * pstride, true);     // (rec_a / rec_b: allocated with the kernel variant below — one interleaved array for GL handles)

    if (p.turb_mode != WG_TURB_NONE) {
        A(pz, n_slots * pstride, true); A(vlp, n_slots * pstride, true); A(wlp, n_slots * pstride, true);
    }
    A(yaw, n_slots * p.N, true); A(u, n_slots * p.N, true); A(v, n_slots * p.N, true); A(w, n_slots * p.N, true);
    A(ti_loc, n_slots * p.N, true); A(power, n_slots * p.N, true); A(ct, n_slots * p.N, true);
    A(bnd, n_slots * p.N * 4, true);
    A(slot, n_slots, true); A(ctx, n_ctx, true); A(env, (size_t)p.B, true);
    A(xr, n_ctx * p.N, true); A(yr, n_ctx * p.N, true); A(jneed, n_ctx * p.N, true);
    A(roff, n_ctx * (p.N + 1), true); A(qown, n_ctx * (size_t)(p.NP / 4), true);
    A(ring, n_ctx * p.ring_stride, true); A(fring, n_ctx * p.fring_stride, true);
    A(cur_ws, n_ctx * p.N, true); A(cur_wd, n_ctx * p.N, true);
    A(pend_farm, n_ctx * p.power_avg, true); A(pend_base, n_ctx * p.power_avg, true);
    A(farm_pow, (size_t)p.B * p.power_avg, true); A(base_pow, (size_t)p.B * p.power_avg, true);
    A(old_yaw, (size_t)p.B * p.N, true);
    A(step_farm_pow, (size_t)p.B, true); A(step_base_pow, (size_t)p.B, true);
    A(last_pow_agent, (size_t)p.B, true); A(last_pow_base, (size_t)p.B, true);
    A(metrics, (size_t)p.B * WG_N_METRICS, true);
    A(next_obs, n_ctx * (size_t)p.obs_dim, false); A(next_obs_ok, n_ctx, false);
    if (p.sums_mode) A(wsum, n_ctx * WG_N_SUMS * (size_t)(p.N + 1), true);
    A(status, 1, true);
#undef A
    if (!rc) rc = dev_alloc(h, &h->mask_dev, (size_t)p.B, false);
    if (!rc) rc = dev_alloc(h, &h->seeds_dev, (size_t)p.B, false);
    if (!rc) rc = dev_alloc(h, &h->unready_dev, 1, false);
    if (!rc) rc = dev_alloc(h, &h->p_dev, 1, false);
    if (!rc) rc = dev_alloc(h, &h->d_dev, 1, false);
    {
        double sx = 0, sy = 0;          // farm centre, summed in layout order like the oracle does
        for (int t = 0; t < p.N; ++t) { sx += c->x_pos[t]; sy += c->y_pos[t]; }
        p.cx0 = sx / p.N; p.cy0 = sy / p.N;
    }
    if (!rc) rc = dev_upload<double>(h, &d.x_pos, c->x_pos, p.N);
    if (!rc) rc = dev_upload<double>(h, &d.y_pos, c->y_pos, p.N);
    if (!rc && c->yaw_defined) rc = dev_upload<double>(h, &d.yaw_defined, c->yaw_defined, p.N);
    if (!rc) {
        // PCG64 k draws ahead, k = 0 .. max(N, 3): the LCG step s -> M s + inc composed k times is s -> A_k s + G_k inc with A_k = M^k,
        // G_k = 1 + M + ... + M^(k-1) (mod 2^128) — the N yaw draws of an episode set-up are then one draw per lane
        const int kmax = std::max(p.N, 3);             // (k = 3: the state after the three wind draws)
        std::vector<uint64_t> jump((size_t)(kmax + 1) * 4);
        wg_u128 A = 1, G = 0;
        for (int k = 0; k <= kmax; ++k) {
            jump[4 * (size_t)k] = (uint64_t)A; jump[4 * (size_t)k + 1] = (uint64_t)(A >> 64);
            jump[4 * (size_t)k + 2] = (uint64_t)G; jump[4 * (size_t)k + 3] = (uint64_t)(G >> 64);
            A = A * WG_PCG_MULT; G = G * WG_PCG_MULT + 1;
        }
        rc = dev_upload<uint64_t>(h, &d.pcg_jump, jump.data(), jump.size());
    }
    if (!rc) rc = dev_upload<float>(h, &d.rotor_dy, c->rotor_dy, p.S);
    if (!rc) rc = dev_upload<float>(h, &d.rotor_dz, c->rotor_dz, p.S);
    if (!rc) rc = dev_upload<float>(h, &d.tab_ws, c->tab_ws, p.n_tab);
    if (!rc) rc = dev_upload<float>(h, &d.tab_power, c->tab_power, p.n_tab);
    if (!rc) rc = dev_upload<float>(h, &d.tab_ct, c->tab_ct, p.n_tab);
    if (!rc) rc = dev_upload<double>(h, &d.tab_ws_d, c->tab_ws, p.n_tab);
    if (!rc) rc = dev_upload<double>(h, &d.tab_power_d, c->tab_power, p.n_tab);
    if (rc) {
        wg_destroy(h);
        return rc;
    }
    // ---- k_flow parameter block -------------------------------------------------------------------
    {
        // turbine table on a uniform grid (O(1) lookup in the kernel); tables that are already uniform
        // (V80: 1 m/s; as_tabular: 0.5 m/s) are used as they are, others are resampled on 1024 points
        bool uniform = true;
        const double dx0 = c->tab_ws[1] - c->tab_ws[0];
        for (int i = 2; i < c->n_tab; ++i)
            if (std::fabs((c->tab_ws[i] - c->tab_ws[i - 1]) - dx0) > 1e-9 * std::fabs(dx0)) uniform = false;
        std::vector<float> pu, cu;
        int nu = c->n_tab;
        double x0 = c->tab_ws[0], dxu = dx0;
        if (uniform) {
            for (int i = 0; i < nu; ++i) { pu.push_back((float)c->tab_power[i]); cu.push_back((float)c->tab_ct[i]); }
        } else {
            nu = 1024;
            dxu = (c->tab_ws[c->n_tab - 1] - x0) / (nu - 1);
            int k = 0;
            for (int i = 0; i < nu; ++i) {
                const double x = std::min(x0 + i * dxu, c->tab_ws[c->n_tab - 1]);
                while (k + 2 < c->n_tab && c->tab_ws[k + 1] <= x) ++k;
                const double f = (x - c->tab_ws[k]) / (c->tab_ws[k + 1] - c->tab_ws[k]);
                pu.push_back((float)(c->tab_power[k] + f * (c->tab_power[k + 1] - c->tab_power[k])));
                cu.push_back((float)(c->tab_ct[k] + f * (c->tab_ct[k + 1] - c->tab_ct[k])));
            }
        }
        const float* tpu = nullptr;
        const float* tcu = nullptr;
        rc = dev_upload<float>(h, &tpu, pu.data(), (size_t)nu);
        if (!rc) rc = dev_upload<float>(h, &tcu, cu.data(), (size_t)nu);
        if (rc) { wg_destroy(h); return rc; }
        FlowP& f = h->fp;
        const size_t pstride_keep = (size_t)f.pstride;
        memset(&f, 0, sizeof(f));
        f.pstride = (int)pstride_keep;
        f.B = p.B; f.N = p.N; f.F = p.F; f.K = p.K; f.P = p.P; f.S = p.S; f.NP = p.NP; f.n_tab = nu;
        f.S_pad = 1; f.S_shift = 0;
        while (f.S_pad < p.S) { f.S_pad <<= 1; f.S_shift++; }
        f.autoreset = p.autoreset; f.action_method = p.action_method; f.base_controller = p.base_controller;
        f.power_avg = p.power_avg; f.script_rows = 0; f.noise = (p.noise == WG_NOISE_NORMAL);
        int tc = 1024 / p.N; if (tc < 1) tc = 1; if (tc > p.N) tc = p.N;
        // phase B maps one thread per (target, sample): keep a chunk's items a multiple of the sample group
        f.target_chunk = tc;
        // Kernel variant.  Small farms (N <= 32: all N x N pairs are staged at once): compact per-turbine rings +
        // pair-major deficit phases, single-wave workgroups (128 threads selectable for tests).  Large farms: uniform P-slot rings with
        // predicate pruning, (target, sample)-major deficit phases, 256 threads.
        const bool small = p.N <= 32 && tc == p.N;
        f.res = small ? 1 : 0;
        f.block = small ? 64 : 256;        // small farms: single-wave workgroups (no s_barrier at all): cfg2 80 vs 91 us at 128 threads
        // large farms with steady inflow: the compact / pair-major variant at 256 threads, pairs staged per chunk of
        // targets (cfg3: 206 vs 224 us once the candidate pre-check is tight — the compacted candidate list then holds
        // ~520 of the 6400 pairs, which the sample-major phases walk in full); turbulent large farms keep the legacy one
        if (!small && p.N <= 255 && p.turb_mode == WG_TURB_NONE) f.res = 1;
        // the compact steady advection lists (turbine, quad-in-ring) in 16 bits: both must fit
        int ql_shift = 0;
        while ((1 << ql_shift) < p.P / 4) ++ql_shift;
        const bool ql_fits = ((long)p.N << ql_shift) <= 65536;
        if (!ql_fits) { f.res = 0; f.block = 256; }
        if (const char* ev = wg_hook("WG_FLOW_BLOCK")) {       // tests: force an instantiation
            const int b = atoi(ev);
            if (b == 256) { f.res = 0; f.block = 256; }
            else if ((b == 64 || b == 128) && small) { f.res = 1; f.block = b; }
        }
        if (const char* ev = wg_hook("WG_FLOW_RES")) {          // experiments: compact / pair-major variant for any farm size
            if (atoi(ev) != 0 && p.N <= 255) { f.res = 1; if (!small) f.block = 256; } else if (atoi(ev) == 0) { f.res = 0; f.block = 256; }
        }
        if (f.res && !ql_fits) { f.res = 0; f.block = 256; }       // (after the env overrides too)
        f.ql_shift = ql_shift;
        f.ql_lpt_shift = 0;
        while (f.ql_lpt_shift < 3 && (p.N << (f.ql_lpt_shift + 1)) <= f.block) ++f.ql_lpt_shift;
        p.compact = f.res;
        // LDS carve.  The staging region of the deficit phases doubles as the quad list of the compact steady advection
        // pass (one 16-bit entry per quad of the farm's rings: at most NP / 2 bytes)
        // Compact variant: the staging of a chunk of targets is 10 bytes per (target, source) pair (16-bit candidate
        // list, deficit, added TI) and the per-pair added-TI array of the sample-major phases is not needed — large farms
        // spend that room on more targets per chunk (cfg3: 27 instead of 12 -> 3 chunks instead of 7, same LDS).
        const int tc0 = tc;
        auto carve = [&]() -> size_t {
            tc = tc0;
            f.target_chunk = tc;
            f.lds_off_ql = 0; f.lds_off_gat = 0; f.gl = 0;      // (re-decided below: the fallback carve runs with res = 0)
            size_t off = sizeof(float) * 4 * (size_t)tc * p.N;
            if (f.res) {
                const size_t ql = ((size_t)p.NP / 2 + 15) & ~(size_t)15;
                if (!small) {
                    const size_t budget = std::max(ql, off) + sizeof(float) * (size_t)tc * p.N;
                    const int fit = (int)((budget - 16) / ((h->added ? 22 : 10) * (size_t)p.N));
                    tc = std::max(tc, std::min(p.N, fit));
                    f.target_chunk = tc;
                }
                // (+ 12 bytes per pair for the added-turbulence contributions when that model is on)
                off = std::max(ql, ((size_t)(h->added ? 22 : 10) * tc * p.N + 16 + 15) & ~(size_t)15);
                f.lf_cap = 0;
                if ((WG_LF_PAIR != 0) && (WG_PAIR_FIRST != 0) && f.block == 256 && p.turb_mode == WG_TURB_NONE) {
                    // large-farm steady variant: results staged per candidate (layout: wg_flow.h, WG_LF_OFF_*); at least 512
                    // candidates at once, whatever the chunked carve would have taken otherwise
                    const size_t fixed = WG_LF_OFF_DEF(p.N);
                    off = std::max(off, (fixed + 8 * 512 + 15) & ~(size_t)15);
                    f.lf_cap = (int)((off - fixed) / 8);
                    // (tests: a small staging capacity, so that the candidates of one flow step take several rounds)
                    if (const char* ev = wg_hook("WG_LF_CAP")) f.lf_cap = std::max(64, std::min(f.lf_cap, atoi(ev)));
                }
                // single-wave steady variant: the deficit phase's gathers are LDS-DMA requests issued before the record /
                // quad-list phases, so the candidate list and the quad list are alive together (no aliasing) and the
                // gathers need a landing zone
                if ((WG_GLDS != 0) && (WG_PAIR_FIRST != 0) && f.block == 64 && p.turb_mode == WG_TURB_NONE) {
                    f.gl = 1;
                    off = ((size_t)10 * tc * p.N + 16 + 15) & ~(size_t)15;
                    f.lds_off_ql = (int)off; off += ql;
                    f.lds_off_gat = (int)off; off += WG_GAT_BYTES;
                }
            }
            f.lds_off_turb = (int)off; off += (size_t)WG_TURB_LDS_BYTES * p.N;
            off = (off + 15) & ~(size_t)15;
            f.lds_off_tab = (int)off; off += sizeof(float) * (2 * (size_t)nu + 2 * (size_t)p.S) + (f.gl ? 0 : sizeof(unsigned) * WG_MASK_WORDS * (size_t)tc) + (f.res ? 0 : sizeof(float) * (size_t)tc * p.N);
            off += sizeof(int) * (4 * (size_t)p.N + 4);      // chain-pruning ages + the particle counter + the candidate counter + the quad counter (+ pad) + the yaws before the action + float positions
            if (h->added && f.res) off += sizeof(float) * 3 * ((size_t)p.N << f.S_shift) + sizeof(int) * (size_t)tc;   // gadd: isotropic field at the rotor points + per-target flags
            return (off + 15) & ~(size_t)15;
        };
        // The compact variant's quad list grows with N * P (NP / 2 bytes): a configuration whose carve exceeds what a
        // workgroup may allocate falls back to the uniform-ring variant (LDS independent of P), and one that fits
        // neither is refused here — not at the first launch, where the failure would be silent.
        int lds_limit = 65536;
        {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0)
                lds_limit = std::min(v, 65536);      // (no kernel opts in to more than the default 64 KB of dynamic LDS)
        }
        size_t off = carve();
        if (f.res && off > (size_t)lds_limit) {
            f.res = 0; f.block = 256;
            p.compact = 0;
            off = carve();
        }
        if (off > (size_t)lds_limit) {
            wg_destroy(h);
            return fail(WG_ERR_UNSUPPORTED, "wg_create: k_flow needs " + std::to_string(off) + " bytes of LDS per workgroup (limit " +
                                            std::to_string(lds_limit) + "): too many turbines for one workgroup");
        }
        // k_flow_env (wg_env.hip): ONE wave per env, lane = slot * N + turbine over the env's 2 F farm slots — the default for
        // steady inflow wherever the slots fit a wave (cfg2: 4 x 16 lanes, cfg4: 4 x 9).  It runs on the GL variant's state
        // layout (interleaved record + 16-byte gather copy), so the two are interchangeable launch by launch.  A hook that
        // asks for a specific older variant (WG_FLOW_BLOCK / WG_FLOW_RES) switches it off; WG_FLOW_ENV=0 / 1
        // forces it off / on where eligible (tests run all of them).
        {
            const int NSl = 2 * p.F, NL = NSl * p.N;
            bool env_ok = f.res && small && f.block == 64 && f.gl && p.turb_mode == WG_TURB_NONE && NL <= 64 && p.N <= 32 &&
                          p.P <= 4096 && !h->added && h->deficit_model == 0;
            // upper bound of a farm's ring quads whatever the wind direction: turbine t keeps roundup4(floor(dx_max / d) + 3)
            // particles, dx_max <= its distance to the farthest turbine (wg_ctx_init)
            size_t qf = 0;
            for (int t = 0; t < p.N; ++t) {
                double dm = 0;
                for (int j = 0; j < p.N; ++j) dm = std::max(dm, std::hypot(c->x_pos[j] - c->x_pos[t], c->y_pos[j] - c->y_pos[t]));
                // (+ 4, not + 3: the device takes floor((xmax - xr_t) / d) after a cos / sin rotation, which rounding can make one
                // particle longer than this host bound — one more quad after rounding up; the list has no other slack)
                long len = (((long)(dm / p.dpart) + 4) + 3) & ~3L;
                if (len > p.P || p.full_chains) len = p.P;
                qf += (size_t)len / 4;
            }
            const size_t clb = ((size_t)2 * NL * (p.N > 1 ? p.N - 1 : 1) + 15) & ~(size_t)15;      // candidate list: every ordered pair of a slot
            const size_t stage = std::max((size_t)8 * WG_ENV_CAP + clb, ((size_t)2 * NSl * qf + 15) & ~(size_t)15);
            f.env_off_tab = (int)((WG_ENV_FIXED_LDS_BYTES + stage + 15) & ~(size_t)15);
            size_t o = (size_t)f.env_off_tab + sizeof(float) * (2 * (size_t)nu + 2 * (size_t)p.S);      // power | ct | rotor points (float2)
            f.env_lds = (int)((o + 15) & ~(size_t)15);
            if (f.env_lds > 32768 || f.env_lds > lds_limit) env_ok = false;
            // k_flow_envb (wg_envb.hip): the same pattern for frozen-box inflow (cfg5) on the turbulent small-farm layout (SoA
            // record, what k_flow<64, BOX> runs on: the two are interchangeable).  Gaussian deficit; rotor points in rows
            // of S_pad <= 16 lanes; the wake-added field and TI folding are run-time options.
            bool envb_ok = f.res && small && f.block == 64 && !f.gl && p.turb_mode >= WG_TURB_BOX && NL <= 64 && p.N <= 32 &&
                           p.P <= 4096 && f.S_pad <= 16 && h->deficit_model == 0;
            // waves per env of k_flow_envb: four (one per farm slot; two farms of at most 16 turbines) while every env's four waves are
            // resident at once (1024 envs = the chip's 4096 wave slots), else two (one per context) up to 2048 envs, else one
            // (four waves per env — one per farm slot, WG_ENV_WPE=4 — is built and tested but NOT the default: measured 104.3 vs 101.7 us
            // per step on cfg5 x 1024, the launch is bound by the bytes it moves, not by a wave's chain of trips)
            int envb_wpe = (p.B <= 2048) ? 2 : 1;
            if (const char* ev = wg_hook("WG_ENV_WPE")) {
                const int w = atoi(ev);
                envb_wpe = (w == 4 && p.F == 2 && p.N <= 16) ? 4 : (w == 2 ? 2 : (w == 4 ? 2 : 1));
            }
            if (envb_ok) {
                auto carve_b = [&](const int wpe) -> size_t {
                    const int nlp = wpe == 4 ? 16 : 64;
                    const int nl_w = (wpe == 4 ? 1 : (wpe == 2 ? p.F : 2 * p.F)) * p.N;      // lanes a wave serves
                    f.env_cap = wpe == 4 ? 128 : 256;
                    size_t ob = WG_ENVB_FIXED_LDS_BYTES(nlp) + (size_t)20 * f.env_cap;
                    f.envb_off_cl = (int)ob;
                    ob = (ob + (size_t)2 * nl_w * (p.N > 1 ? p.N - 1 : 1) + 15) & ~(size_t)15;      // candidate list: every ordered pair of a slot
                    f.env_off_tab = (int)ob;
                    ob += sizeof(float) * (2 * (size_t)nu + 2 * (size_t)p.S);
                    return (ob + 15) & ~(size_t)15;
                };
                size_t lb = carve_b(envb_wpe);
                if (envb_wpe > 1 && envb_wpe * lb + 64 > (size_t)lds_limit) { envb_wpe = envb_wpe == 4 ? 2 : 1; lb = carve_b(envb_wpe); }
                if (envb_wpe > 1 && envb_wpe * lb + 64 > (size_t)lds_limit) { envb_wpe = 1; lb = carve_b(1); }
                f.env_lds = (int)lb;
                if (f.env_lds > 32768 || f.env_lds > lds_limit) envb_ok = false;
            }
            env_ok = env_ok || envb_ok;
            f.env_inc = 1 + (p.extra_inc ? 1 : 0);
            f.env_eps_max = std::min(1.0f, (float)(p.eps0 * std::sqrt(3.0))) + 2.0f / 65535.0f;
            const bool asked_old = wg_hook("WG_FLOW_BLOCK") || wg_hook("WG_FLOW_RES");
            f.envw = (env_ok && !asked_old) ? 1 : 0;
            if (const char* ev = wg_hook("WG_FLOW_ENV")) f.envw = (env_ok && atoi(ev) != 0) ? 1 : 0;
            // waves per env: two (one per context, side by side, meeting only when the env truncates) while every env's pair of
            // waves is resident at once — 2048 envs fill the chip's 4096 wave slots; beyond that one wave per env (cfg2 ms per
            // step, two waves / one wave / per-slot kernels + k_glue_lean: 256 envs 0.0231 / 0.0294 / 0.0260, 1024: 0.0276 /
            // 0.0360 / 0.0322, 2048: 0.0340 / 0.0411 / 0.0435, 4096: 0.0643 / 0.0624 / 0.0682)
            f.env_wpe = (p.B <= 2048 && 2 * f.env_lds <= lds_limit) ? 2 : 1;
            if (const char* ev = wg_hook("WG_ENV_WPE")) f.env_wpe = (atoi(ev) == 2 && 2 * f.env_lds <= lds_limit) ? 2 : 1;
            // small batches: a pass wave for the running episode's context (wg_env.hip, SPLIT: three waves per env) while they sit at
            // most three to a SIMD (1024 envs) and a farm's pass is four trips or more (cfg2 x 256 / 512 / 1024: +19 / +18 / +10 %; x 2048:
            // -21 %, more than one dispatch round; cfg4's 3 x 3 farm: two trips, -3 to -6 %: the wave costs more than it takes over)
            f.env_split = (env_ok && !envb_ok && f.env_wpe == 2 && p.B <= 1024 && qf >= 256 && p.K == 1 && 4 * f.env_lds + 4096 <= lds_limit) ? (p.B <= 512 ? 2 : 1) : 0;
            if (const char* ev = wg_hook("WG_ENV_SPLIT"))
                f.env_split = (atoi(ev) != 0 && env_ok && !envb_ok && f.env_wpe == 2 && p.B <= 2048 && p.K == 1 && 4 * f.env_lds + 4096 <= lds_limit) ? (atoi(ev) == 2 ? 2 : 1) : 0;
            if (envb_ok) f.env_wpe = envb_wpe;
        }
        // packed emission record: two arrays, or one interleaved (ct|k, u_e|hv) array for the steady compact variants whose
        // deficit phase gathers bracket pairs from it (GL / k_flow_env: 64 threads; LF: 256 threads)
        f.rec_il = (f.gl || (f.res && f.block == 256 && p.turb_mode == WG_TURB_NONE && (WG_LF_PAIR != 0) && (WG_PAIR_FIRST != 0))) ? 1 : 0;
        if (f.rec_il) {
            if (!rc) rc = dev_alloc(h, &d.rec_a, 2 * n_slots * pstride_keep, true);
            d.rec_b = d.rec_a ? d.rec_a + 1 : nullptr;
        } else {
            if (!rc) rc = dev_alloc(h, &d.rec_a, n_slots * pstride_keep, true);
            if (!rc) rc = dev_alloc(h, &d.rec_b, n_slots * pstride_keep, true);
        }
        if (rc) { wg_destroy(h); return rc; }
        f.lds_bytes = (int)((off + 15) & ~(size_t)15);
        if (const char* ev = wg_hook("WG_LDS_PAD")) f.lds_bytes = std::min(lds_limit, f.lds_bytes + atoi(ev));      // (measurement: what a workgroup's LDS size alone costs)
        f.dt = p.dt; f.D = p.D; f.inv_D = p.inv_D; f.hub = p.hub; f.dpart_f = (float)p.dpart; f.R_rot = 0.5f * p.D;
        f.inv_N = 1.0f / (float)p.N; f.inv_S = 1.0f / (float)p.S; f.inv_P = 1.0f / (float)p.P;
        f.inv_power_avg = 1.0f / (float)(p.power_avg > 0 ? p.power_avg : 1);
        f.pavg_magic = (unsigned)((1ull << 32) / (unsigned long long)(p.power_avg > 0 ? p.power_avg : 1)) + 1u;
        f.dt_d = p.dt_d; f.dpart = p.dpart; f.inv_dpart = 1.0 / p.dpart;
        f.yaw_min = p.yaw_min; f.yaw_max = p.yaw_max; f.yaw_step = p.yaw_step;
        f.ka = p.ka; f.kb = p.kb; f.eps0 = p.eps0; f.hill = p.hill; f.tia = p.tia; f.tib = p.tib; f.tic = p.tic; f.tid = p.tid;
        f.tab_x0 = (float)x0; f.tab_inv_dx = (float)(1.0 / dxu);
        f.ue_scale = p.turb_mode == WG_TURB_NONE ? 1.0f : 2.0f;
        for (int i = 0; i < WG_N_CH; ++i) {
            // (the flow kernels only push: what they call the history length is the rings' physical capacity)
            f.hlen[i] = p.ring_cap[i]; f.ring_off[i] = p.ring_off[i]; f.fring_off[i] = p.fring_off[i];
            f.inv_hlen[i] = 1.0f / (float)p.ring_cap[i];
            f.hmagic[i] = (unsigned)((1ull << 32) / (unsigned long long)p.ring_cap[i]) + 1u;
            f.noise_sigma[i] = p.noise_sigma[i];
        }
        f.ring_stride = p.ring_stride; f.fring_stride = p.fring_stride;
        f.turb_mode = p.turb_mode; f.fc_scale = p.fc_scale; f.D_d = p.D_d; f.hub_d = p.hub_d;
        f.inv_sqrt_S = 1.0f / std::sqrt((float)p.S);
        f.added = h->added; f.no_ti_fold = h->no_ti_fold; f.deficit_model = h->deficit_model;
        f.km1 = (float)h->km1; f.km2r = (float)(2.0 * h->km2 * 0.5 * p.D_d);
        f.sg_af = (float)h->sg_af; f.sg_bf = (float)h->sg_bf; f.sg_cf = (float)h->sg_cf;
        f.an_inv_ka = (float)(1.0 / defd(c->m0_ka, 0.38));
        if (h->deficit_model != 0 && !f.res) {
            wg_destroy(h);
            return fail(WG_ERR_UNSUPPORTED, "deficit_model 1 / 2 (super-Gaussian, eddy-viscosity table) are built into the compact k_flow variants only "
                                            "(farms of up to 32 turbines, larger ones with steady inflow)");
        }
        FlowPtrs& g = h->fd;
        memset(&g, 0, sizeof(g));
        g.py = d.py; g.rec_a = d.rec_a; g.rec_b = d.rec_b;
        g.pz = d.pz; g.vlp = d.vlp; g.wlp = d.wlp;
        g.bnd = d.bnd;
        g.dbg = nullptr;
        if (wg_hook("WG_TIMELINE_OUT")) {
            long long* dbgp = nullptr;
            if (!dev_alloc(h, &dbgp, (size_t)p.B * 2 * p.F * 16, false)) g.dbg = dbgp;
        }
        g.yaw = d.yaw; g.u = d.u; g.v = d.v; g.w = d.w; g.ti_loc = d.ti_loc; g.power = d.power; g.ct = d.ct;
        g.slot = d.slot; g.ctx = d.ctx; g.env = d.env; g.xr = d.xr; g.yr = d.yr; g.jneed = d.jneed;
        g.ring = d.ring; g.fring = d.fring; g.cur_ws = d.cur_ws; g.cur_wd = d.cur_wd;
        g.pend_farm = d.pend_farm; g.pend_base = d.pend_base; g.old_yaw = d.old_yaw;
        g.step_farm_pow = d.step_farm_pow; g.step_base_pow = d.step_base_pow;
        g.rotor_dy = d.rotor_dy; g.rotor_dz = d.rotor_dz; g.tab_power = tpu; g.tab_ct = tcu;
        g.gp = h->p_dev; g.gd = h->d_dev; g.env_rw = d.env;
        g.roff = d.roff; g.qown = d.qown; g.status = d.status;
    }

    // how many RESET-mode launches develop the slowest possible episode: the chain needs
    // int(2 * dist / ws) steps (dist <= layout diagonal, ws >= ws_min) plus the window fill
    double ext_x = 0, ext_y = 0, xmn = 1e300, xmx = -1e300, ymn = 1e300, ymx = -1e300;
    for (int t = 0; t < p.N; ++t) {
        xmn = std::min(xmn, c->x_pos[t]); xmx = std::max(xmx, c->x_pos[t]);
        ymn = std::min(ymn, c->y_pos[t]); ymx = std::max(ymx, c->y_pos[t]);
    }
    ext_x = xmx - xmn; ext_y = ymx - ymn;
    const double diag = std::sqrt(ext_x * ext_x + ext_y * ext_y);
    const double ws_lo = std::max(1e-3, std::min(c->ws_min, c->ws_max));
    const long max_dev = (long)std::ceil(2.0 * diag / ws_lo / c->dt_sim) + 2;
    const long max_work = max_dev + (long)p.K * (std::max(p.fill_a, p.fill_b) + 1);
    h->reset_chunk = 32;
    h->reset_launches = (int)((max_work + h->reset_chunk - 1) / h->reset_chunk) + 1;

    // algorithmic bytes of one step(): advection streams py r/w + 4 record floats per particle;
    // per turbine 7 floats r/w (+ positions); glue: rings, obs, actions, reward/flag
    // inflow None: py r/w + 4 record floats; turbulent: + pz, vlp, wlp r/w (24) and, for the box, 8 corners x
    // 2 components gathered per particle (64) and 8 x 3 per rotor point (96)
    const bool turb = p.turb_mode != WG_TURB_NONE, boxm = p.turb_mode >= WG_TURB_BOX;
    // record: 2 x 32-bit packed words per particle (8 B)
    const double per_farm_step = (double)p.NP * (4 + 4 + 8 + (turb ? 24 : 0) + (boxm ? 64 : 0)) +
                                 (double)p.N * (7 * 4 * 2 + 16) + (boxm ? (double)p.N * p.S * 96 : 0.0);
    h->alg_bytes = (double)p.B * ((double)p.K * p.F * per_farm_step + 20.0 * p.N + 12.0 * p.obs_dim + 20.0);

    // The first observation of a background episode is built inside k_flow only by the single-wave steady variant (small
    // farms: one wave reads 35 floats per turbine; measured on the multi-wave variants — cfg3, cfg5 — the building
    // workgroup became the flow kernel's tail: -5 % / -1.3 %).  Other handles keep the glue's own second build.
    // (sums mode: wg_first_obs prepares the episode's window sums as well, in WgPtrs::wsum)
    // (round 4, sums mode: the glue's own rebuild sums every window of the new episode — 14-15 us of k_glue_lean on cfg3 /
    // cfg5 against 10 with prepared sums — so every compact variant prepares them now; WG_FIRST_OBS_GL_ONLY=1 for A/B runs)
    {   // one launch per step where the env kernel runs and the lean glue's specialised instantiation applies
        const bool gen = p.turb_ti || p.farm_ti || p.farm_obs > 0 || (p.sum_mask_f | p.cur_mask_f) != 0;
        h->fp.env_fused = (h->fp.envw && p.sums_mode && !gen && p.power_avg <= 64) ? 1 : 0;
        if (const char* ev = wg_hook("WG_STEP_FUSED")) h->fp.env_fused = (h->fp.env_fused && atoi(ev) != 0) ? 1 : 0;
    }
    const bool prep_all = p.sums_mode && !wg_hook("WG_FIRST_OBS_GL_ONLY");
    if (!(h->fp.gl || (h->fp.res && prep_all))) { d.next_obs = nullptr; d.next_obs_ok = nullptr; }
    wg_launch_create(&p, &d, nullptr);
    if (sync_dev_params(h)) { wg_destroy(h); return WG_ERR_HIP; }
    {
        hipError_t e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            wg_destroy(h);
            return fail(WG_ERR_HIP, std::string("wg_create: ") + hipGetErrorString(e));
        }
    }
    if (const char* ev = wg_hook("WG_STEP_GRAPH")) h->graph_mode = atoi(ev) != 0;
    if (wg_hook("WG_DEBUG"))
        fprintf(stderr, "[windgym] k_flow variant: %s%s, %d threads, LDS %d B per workgroup, slot stride %d floats\n",
                h->fp.envw ? (h->fp.turb_mode == WG_TURB_NONE ? "compact rings / pair-major, one wave per env (k_flow_env)" : "compact rings / sample-major, one wave per env (k_flow_envb)")
                          : (h->fp.res ? "compact rings / pair-major" : "uniform rings / sample-major"),
                h->fp.envw && h->fp.env_split ? " + a pass wave for the running episode's context" : "",
                h->fp.block, h->fp.envw ? h->fp.env_lds : h->fp.lds_bytes, h->fp.pstride);
    *out = h;
    return 0;
}

extern "C" int wg_destroy(wg_handle h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    if (h->fd.dbg && wg_hook("WG_TIMELINE_OUT")) {      // -DWG_TIMELINE builds: dump the phase stamps of the last launch
        const size_t n = (size_t)h->p.B * 2 * h->p.F * 16;
        std::vector<long long> host(n);
        if (hipMemcpy(host.data(), h->fd.dbg, n * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE* f = fopen(wg_hook("WG_TIMELINE_OUT"), "wb")) { fwrite(host.data(), sizeof(long long), n, f); fclose(f); }
        }
    }
    drop_step_graphs(h);
    if (h->cap_stream) hipStreamDestroy(h->cap_stream);
    for (auto& e : h->ev) hipEventDestroy(e);
    for (auto& a : h->allocs) hipFree(a.ptr);
    if (h->box4) hipFree(h->box4);
    if (h->box4c) hipFree(h->box4c);
    if (h->abox4) hipFree(h->abox4);
    if (h->box8) hipFree(h->box8);
    if (h->abox8) hipFree(h->abox8);
    delete h;
    return 0;
}

extern "C" int wg_obs_dim(wg_handle h, int* obs_dim, int* obs_dim_multi) {
    if (!h) return fail(WG_ERR_INVALID, "null handle");
    if (obs_dim) *obs_dim = h->p.obs_dim;
    if (obs_dim_multi) *obs_dim_multi = h->p.obs_dim_multi;
    return 0;
}
extern "C" int wg_hist_max(wg_handle h, int* hist_max) {
    if (!h || !hist_max) return fail(WG_ERR_INVALID, "null argument");
    *hist_max = h->p.hist_max;
    return 0;
}

// Stencil records of a box pool (FlowPtrs::box8 / abox8) for handles that run k_flow_envb: 8 x the interleaved copy.  Built only
// while the whole pool's records stay below a quarter of the device memory and a box below 2^28 cells (32-bit record offsets);
// otherwise the kernel keeps reading the brick-ordered box (same cells, same weights: the results are identical either way).
static int build_stencil_records(wg_env_s* h, void** owned, const float4** kernel_ptr, const void* box4, int n_boxes, int nx, int ny, int nz) {
    *kernel_ptr = nullptr;
    if (!(h->fp.envw && h->fp.turb_mode != WG_TURB_NONE) || wg_hook("WG_NO_BOX8")) return 0;
    const size_t n_cells = (size_t)nx * ny * nz, bytes = n_cells * 128 * (size_t)n_boxes;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    if (n_cells >= ((size_t)1 << 28) || bytes > total_b / 4 || bytes > free_b / 2) return 0;
    if (hipMalloc(owned, bytes) != hipSuccess) { *owned = nullptr; (void)hipGetLastError(); return 0; }
    for (int k = 0; k < n_boxes; ++k)
        wg_launch_box_stencil((const char*)box4 + (size_t)k * n_cells * 16, (char*)*owned + (size_t)k * n_cells * 128, nx, ny, nz, nullptr);
    HIPCHK(hipDeviceSynchronize());
    *kernel_ptr = (const float4*)*owned;
    return 0;
}

extern "C" int wg_set_turbulence_boxes(wg_handle h, const float* const* boxes_dev, int n_boxes, int nx, int ny, int nz,
                                       double dx, double dy, double dz) {
    if (!h) return fail(WG_ERR_INVALID, "null handle");
    if (!boxes_dev || n_boxes < 1 || n_boxes > 64 || nx < 2 || ny < 2 || nz < 2 || !(dx > 0) || !(dy > 0) || !(dz > 0))
        return fail(WG_ERR_INVALID, "turbulence box: null pointer, bad pool size (1..64) or bad dimensions");
    for (int k = 0; k < n_boxes; ++k)
        if (!boxes_dev[k]) return fail(WG_ERR_INVALID, "turbulence box: null pointer in the pool");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    drop_step_graphs(h);           // the kernel arguments captured in them are about to change
    // the kernel-side pointers are cleared first: a failure below must not leave them dangling
    h->d.box = nullptr; h->fd.box4 = nullptr; h->fd.box4c = nullptr; h->fd.box8 = nullptr; h->fp.coarse = 0;
    if (h->box8) { void* q = h->box8; h->box8 = nullptr; HIPCHK(hipFree(q)); }
    if (h->box4) { void* q = h->box4; h->box4 = nullptr; HIPCHK(hipFree(q)); }
    if (h->box4c) { void* q = h->box4c; h->box4c = nullptr; HIPCHK(hipFree(q)); }
    const size_t n_cells = (size_t)nx * ny * nz;
    HIPCHK(hipMalloc(&h->box4, n_cells * 16 * (size_t)n_boxes));
    for (int k = 0; k < n_boxes; ++k)
        wg_launch_box_repack(boxes_dev[k], (char*)h->box4 + (size_t)k * n_cells * 16, nx, ny, nz, nullptr);
    HIPCHK(hipDeviceSynchronize());
    h->d.box = boxes_dev[0];
    h->p.n_boxes = n_boxes;
    h->p.bnx = nx; h->p.bny = ny; h->p.bnz = nz; h->p.bdx = dx; h->p.bdy = dy; h->p.bdz = dz;
    h->fd.box4 = (const float4*)h->box4;
    if (h->fp.envw && h->fp.turb_mode != WG_TURB_NONE && n_cells >= ((size_t)1 << 28)) {
        // (k_flow_envb addresses the cells of a box with 32-bit offsets: a larger box runs on the per-slot kernel + k_glue_lean,
        // which share its state layout)
        h->fp.envw = 0; h->fp.env_fused = 0;
    }
    h->fp.bnx = nx; h->fp.bny = ny; h->fp.bnz = nz; h->fp.box_cells = (long long)n_cells;
    h->fp.box_pow2 = ((nx & (nx - 1)) == 0) && ((ny & (ny - 1)) == 0) && ((nz & (nz - 1)) == 0);
    h->fp.coarse = (nx % 4 == 0 && ny % 4 == 0 && nz % 4 == 0 && nx >= 8 && ny >= 8 && nz >= 8);
    h->fd.box4c = nullptr;
    if (h->fp.coarse) {
        const int cx = nx / 4, cy = ny / 4, cz = nz / 4;
        const size_t c_cells = (size_t)cx * cy * cz;
        HIPCHK(hipMalloc(&h->box4c, c_cells * 16 * (size_t)n_boxes));
        for (int k = 0; k < n_boxes; ++k)
            wg_launch_box_coarsen((const char*)h->box4 + (size_t)k * n_cells * 16, (char*)h->box4c + (size_t)k * c_cells * 16,
                                  nx, ny, nz, nullptr);
        HIPCHK(hipDeviceSynchronize());
        h->fp.cnx = cx; h->fp.cny = cy; h->fp.cnz = cz; h->fp.cbox_cells = (long long)c_cells;
        h->fp.cbox_pow2 = ((cx & (cx - 1)) == 0) && ((cy & (cy - 1)) == 0) && ((cz & (cz - 1)) == 0);
        h->fd.box4c = (const float4*)h->box4c;
    }
    h->fp.inv_bdx = 1.0 / dx; h->fp.inv_bdy = 1.0 / dy; h->fp.inv_bdz = 1.0 / dz;
    h->fp.inv_bn[0] = 1.0 / nx; h->fp.inv_bn[1] = 1.0 / ny; h->fp.inv_bn[2] = 1.0 / nz;
    if (int rc = build_stencil_records(h, &h->box8, &h->fd.box8, h->box4, n_boxes, nx, ny, nz)) return rc;
    return sync_dev_params(h);
}

extern "C" int wg_set_added_turbulence_box(wg_handle h, const float* box_dev, int nx, int ny, int nz, double dx,
                                           double dy, double dz) {
    if (!h || !box_dev || nx < 2 || ny < 2 || nz < 2 || !(dx > 0) || !(dy > 0) || !(dz > 0))
        return fail(WG_ERR_INVALID, "added-turbulence box: null pointer or bad dimensions");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    drop_step_graphs(h);
    h->fd.abox4 = nullptr; h->fd.abox8 = nullptr;
    if (h->abox8) { void* q = h->abox8; h->abox8 = nullptr; HIPCHK(hipFree(q)); }
    if (h->abox4) { void* q = h->abox4; h->abox4 = nullptr; HIPCHK(hipFree(q)); }
    const size_t n_cells = (size_t)nx * ny * nz;
    HIPCHK(hipMalloc(&h->abox4, n_cells * 16));
    wg_launch_box_repack(box_dev, h->abox4, nx, ny, nz, nullptr);
    HIPCHK(hipDeviceSynchronize());
    h->fd.abox4 = (const float4*)h->abox4;
    if (h->fp.envw && h->fp.turb_mode != WG_TURB_NONE && n_cells >= ((size_t)1 << 28)) { h->fp.envw = 0; h->fp.env_fused = 0; }
    h->fp.anx = nx; h->fp.any = ny; h->fp.anz = nz;
    h->fp.abox_pow2 = ((nx & (nx - 1)) == 0) && ((ny & (ny - 1)) == 0) && ((nz & (nz - 1)) == 0);
    h->fp.inv_adx = 1.0 / dx; h->fp.inv_ady = 1.0 / dy; h->fp.inv_adz = 1.0 / dz;
    h->fp.inv_an[0] = 1.0 / nx; h->fp.inv_an[1] = 1.0 / ny; h->fp.inv_an[2] = 1.0 / nz;
    if (int rc = build_stencil_records(h, &h->abox8, &h->fd.abox8, h->abox4, 1, nx, ny, nz)) return rc;
    return 0;
}

extern "C" int wg_set_deficit_table(wg_handle h, const float* table_dev, int n_ct, double ct0, double ct1, int n_ti, double ti0,
                                    double ti1, int n_x, double x_max_D, int n_r, double r_max_R) {
    if (!h || !table_dev || n_ct < 2 || n_ti < 2 || n_x < 2 || n_r < 2 || !(ct1 > ct0) || !(ti0 > 0) || !(ti1 > ti0) || !(x_max_D > 0) ||
        !(r_max_R > 0))
        return fail(WG_ERR_INVALID, "wg_set_deficit_table: null pointer or bad grid");
    if ((long long)n_ct * n_ti * n_x * n_r >= (1ll << 30)) return fail(WG_ERR_INVALID, "wg_set_deficit_table: table too large");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    drop_step_graphs(h);
    FlowP& f = h->fp;
    h->fd.dtab = table_dev;            // borrowed: the caller keeps the buffer alive
    f.an_ct = n_ct; f.an_ti = n_ti; f.an_x = n_x; f.an_r = n_r;
    f.an_ct0 = (float)ct0; f.an_inv_dct = (float)((n_ct - 1) / (ct1 - ct0));
    f.an_lti0 = (float)std::log(ti0); f.an_inv_dlti = (float)((n_ti - 1) / (std::log(ti1) - std::log(ti0)));
    f.an_inv_dx = (float)((n_x - 1) / x_max_D); f.an_inv_dr = (float)((n_r - 1) / r_max_R);
    return 0;
}

extern "C" int wg_set_turbulence_box(wg_handle h, const float* box_dev, int nx, int ny, int nz, double dx,
                                     double dy, double dz) {
    const float* one[1] = {box_dev};
    if (!box_dev) return fail(WG_ERR_INVALID, "turbulence box: null pointer or bad dimensions");
    return wg_set_turbulence_boxes(h, one, 1, nx, ny, nz, dx, dy, dz);
}

extern "C" int wg_set_wind(wg_handle h, const double* wind_host) {
    if (!h) return fail(WG_ERR_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    drop_step_graphs(h);
    if (!wind_host) { h->d.wind_override = nullptr; return sync_dev_params(h); }
    if (!h->wind_dev) {
        double* w = nullptr;
        int rc = dev_alloc(h, &w, (size_t)h->p.B * 3, false);
        if (rc) return rc;
        h->wind_dev = w;
    }
    HIPCHK(hipMemcpy(h->wind_dev, wind_host, sizeof(double) * 3 * (size_t)h->p.B, hipMemcpyHostToDevice));
    h->d.wind_override = h->wind_dev;
    return sync_dev_params(h);
}

extern "C" int wg_set_box_ids(wg_handle h, const int32_t* ids_host) {
    if (!h) return fail(WG_ERR_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    drop_step_graphs(h);
    if (!ids_host) { h->d.box_override = nullptr; return sync_dev_params(h); }
    // (a caller error must not produce results labelled with the wrong box: wg_ctx_init would fold a bad id to box 0 and
    // ignores the table without a pool)
    if (h->p.turb_mode != WG_TURB_BOX_POOL || h->p.n_boxes < 1)
        return fail(WG_ERR_INVALID, "wg_set_box_ids: the handle has no box pool (turb_mode WG_TURB_BOX_POOL + wg_set_turbulence_boxes first)");
    for (int b = 0; b < h->p.B; ++b)
        if (ids_host[b] >= h->p.n_boxes)
            return fail(WG_ERR_INVALID, "wg_set_box_ids: env " + std::to_string(b) + " asks for box " + std::to_string(ids_host[b]) +
                                            " of a pool of " + std::to_string(h->p.n_boxes));
    if (!h->box_ids_dev) {
        int* w = nullptr;
        int rc = dev_alloc(h, &w, (size_t)h->p.B, false);
        if (rc) return rc;
        h->box_ids_dev = w;
    }
    HIPCHK(hipMemcpy(h->box_ids_dev, ids_host, sizeof(int) * (size_t)h->p.B, hipMemcpyHostToDevice));
    h->d.box_override = h->box_ids_dev;
    return sync_dev_params(h);
}

extern "C" int wg_set_wind_device(wg_handle h, const double* wind_dev) {
    if (!h) return fail(WG_ERR_INVALID, "null handle");
    if (h->d.wind_override != wind_dev) drop_step_graphs(h);
    h->d.wind_override = wind_dev;      // borrowed; read by wg_ctx_init (k_init / head of k_flow) in stream order
    if (int rc = use_device(h)) return rc;
    return sync_dev_params(h);
}

extern "C" int wg_set_flow_script(wg_handle h, const float* uvw_dev, const float* power_dev, int n_rows) {
    if (!h) return fail(WG_ERR_INVALID, "null handle");
    drop_step_graphs(h);
    h->d.script_uvw = uvw_dev;
    h->d.script_power = power_dev;
    h->p.script_rows = n_rows;
    h->fd.script_uvw = uvw_dev;
    h->fd.script_power = power_dev;
    h->fp.script_rows = n_rows;
    if (int rc = use_device(h)) return rc;
    return sync_dev_params(h);
}

// The event pool is created by wg_kernel_timing(enable), i.e. outside any timed region, and is bounded: when it is
// exhausted the remaining steps of the window simply go unsampled.
static bool time_begin(wg_env_s* h, int kind, hipStream_t st) {
    if (!h->timing || h->ev_used + 2 > h->ev.size()) return false;
    h->ev_kind[h->ev_used / 2] = kind;
    hipEventRecord(h->ev[h->ev_used], st);
    return true;
}
static void time_end(wg_env_s* h, hipStream_t st) {
    hipEventRecord(h->ev[h->ev_used + 1], st);
    h->ev_used += 2;
}

extern "C" int wg_reset(wg_handle h, const uint8_t* env_mask_host, const uint64_t* seeds_host, float* obs_dev,
                        void* stream) {
    if (!h) return fail(WG_ERR_INVALID, "null handle");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->device));
    if (h->p.turb_mode >= WG_TURB_BOX && !h->d.box)
        return fail(WG_ERR_INVALID, "turbtype Mann*: call wg_set_turbulence_box before wg_reset");
    if (h->added && !h->fd.abox4)
        return fail(WG_ERR_INVALID, "added_turbulence: call wg_set_added_turbulence_box before wg_reset");
    if (h->deficit_model == 2 && !h->fd.dtab)
        return fail(WG_ERR_INVALID, "deficit_model 2: call wg_set_deficit_table before wg_reset");
    const uint8_t* mask = nullptr;
    const uint64_t* seeds = nullptr;
    bool all = true;
    if (env_mask_host) {
        for (int b = 0; b < h->p.B; ++b) all = all && env_mask_host[b] != 0;
        HIPCHK(hipMemcpyAsync(h->mask_dev, env_mask_host, (size_t)h->p.B, hipMemcpyHostToDevice, st));
        mask = h->mask_dev;
    }
    if (seeds_host) {
        HIPCHK(hipMemcpyAsync(h->seeds_dev, seeds_host, sizeof(uint64_t) * (size_t)h->p.B, hipMemcpyHostToDevice, st));
        seeds = h->seeds_dev;
    }
    // the sticky error word is cleared only by a reset of the whole batch: a masked reset must not drop an error
    // another env latched
    if (all) HIPCHK(hipMemsetAsync(h->d.status, 0, sizeof(int), st));
    wg_launch_init(&h->p, &h->d, mask, seeds, st);
    // RESET-mode launches develop the masked envs' episodes (fs.run(t_developed) + window fill).  The planned count
    // covers every episode the config's wind range can produce; a wind override (wg_set_wind / wg_set_wind_device,
    // FarmEval.set_wind_vals) may ask for a slower wind than ws_min, so the device reports how many live slots are
    // still developing and the loop goes on until none is.
    const int n_plan = h->d.script_uvw ? ((h->p.K * (std::max(h->p.fill_a, h->p.fill_b) + 1)) / h->reset_chunk + 2)
                                       : h->reset_launches;
    long launched = 0;
    int batch = n_plan;
    for (;;) {
        for (int i = 0; i < batch; ++i) wg_launch_flow(&h->fp, &h->fd, WG_MODE_RESET, nullptr, mask, h->reset_chunk, st);
        {
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) return fail(WG_ERR_HIP, std::string("wg_reset: k_flow launch failed: ") + hipGetErrorString(le));
        }
        launched += batch;
        int unready = 0;
        HIPCHK(hipMemsetAsync(h->unready_dev, 0, sizeof(int), st));
        wg_launch_unready(&h->p, &h->d, mask, h->unready_dev, st);
        HIPCHK(hipMemcpyAsync(&unready, h->unready_dev, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (unready == 0) break;
        if (launched > 4000000L / h->reset_chunk)
            return fail(WG_ERR_STATE, "wg_reset: episode development does not terminate (wind speed override ~ 0?)");
        batch = std::max(8, n_plan / 4);
    }
    wg_launch_glue(&h->p, &h->d, 1, mask, obs_dev, nullptr, nullptr, nullptr, st, h->p_dev, h->d_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

// the launches of one step(); `sample`: bracket the two kernels with timing events
static void launch_step(wg_env_s* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* truncated_dev,
                        float* final_obs_dev, hipStream_t st, bool sample) {
    // step() as ONE launch (k_flow_env with the glue as its tail): sums-mode handles on the env kernel whose observation has
    // no TI / farm-level entries; a flow script (replay mode) falls back to the two launches.  (Null output pointers are
    // legal on either path: lean_step checks each one before it stores.)
    if (h->fp.env_fused && h->fd.script_uvw == nullptr) {
        if (sample) sample = time_begin(h, 0, st);
        if (h->fp.turb_mode == WG_TURB_NONE) wg_launch_step_env(&h->fp, &h->fd, &h->p, &h->d, actions_dev, obs_dev, reward_dev, truncated_dev, final_obs_dev, st);
        else wg_launch_step_envb(&h->fp, &h->fd, &h->p, &h->d, actions_dev, obs_dev, reward_dev, truncated_dev, final_obs_dev, st);
        if (sample) time_end(h, st);
        return;
    }
    if (sample) sample = time_begin(h, 0, st);
    wg_launch_flow(&h->fp, &h->fd, WG_MODE_STEP, actions_dev, nullptr, 0, st);
    if (sample) { time_end(h, st); sample = time_begin(h, 1, st); }
    wg_launch_glue(&h->p, &h->d, 0, nullptr, obs_dev, reward_dev, truncated_dev, final_obs_dev, st, h->p_dev, h->d_dev);
    if (sample) time_end(h, st);
}

extern "C" int wg_step(wg_handle h, const float* actions_dev, float* obs_dev, float* reward_dev,
                       uint8_t* truncated_dev, float* final_obs_dev, void* stream) {
    if (!h || !actions_dev || !obs_dev) return fail(WG_ERR_INVALID, "null argument");
    if (int rc = use_device(h)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool sample = h->timing && (h->timing_phase++ % h->timing_period == 0);
    h->n_step_launches++;
    if (!h->graph_mode || sample) {
        launch_step(h, actions_dev, obs_dev, reward_dev, truncated_dev, final_obs_dev, st, sample);
        // a rejected launch (bad configuration, out-of-resources) must not pass for a step that returned stale outputs
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) return fail(WG_ERR_HIP, std::string("wg_step: kernel launch failed: ") + hipGetErrorString(le));
        return 0;
    }
    // graph mode: one hipGraphLaunch per step.  A graph is captured once per distinct set of I/O pointers (a training
    // loop recycles a handful of action / observation buffers) on the handle's own stream — the caller's stream may be
    // the null stream, which cannot be captured — and replayed on the caller's stream.
    const void* key[5] = {actions_dev, obs_dev, reward_dev, truncated_dev, final_obs_dev};
    wg_env_s::StepGraph* g = nullptr;
    for (auto& c : h->graphs)
        if (!memcmp(c.key, key, sizeof(key))) { g = &c; break; }
    if (!g) {
        if (h->graphs.size() >= WG_MAX_STEP_GRAPHS) {      // evict the least recently used entry
            size_t lru = 0;
            for (size_t i = 1; i < h->graphs.size(); ++i)
                if (h->graphs[i].last_use < h->graphs[lru].last_use) lru = i;
            hipDeviceSynchronize();          // (the evicted exec may still be running)
            hipGraphExecDestroy(h->graphs[lru].exec);
            hipGraphDestroy(h->graphs[lru].graph);
            h->graphs.erase(h->graphs.begin() + lru);
        }
        wg_env_s::StepGraph ng;
        memcpy(ng.key, key, sizeof(key));
        // Whatever fails between Begin and End, the capture is always ended (a stream left in capture mode would break
        // every later graph-mode step) and the partial graph destroyed; the step then falls back to direct launches
        // and graph mode is switched off for the handle.
        hipError_t e = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal);
        bool ok = e == hipSuccess;
        if (ok) {
            launch_step(h, actions_dev, obs_dev, reward_dev, truncated_dev, final_obs_dev, h->cap_stream, false);
            const hipError_t le = hipGetLastError();
            ng.graph = nullptr;
            e = hipStreamEndCapture(h->cap_stream, &ng.graph);
            ok = le == hipSuccess && e == hipSuccess && ng.graph != nullptr;
            if (ok) {
                e = hipGraphInstantiate(&ng.exec, ng.graph, nullptr, nullptr, 0);
                ok = e == hipSuccess;
            }
            if (!ok && ng.graph) hipGraphDestroy(ng.graph);
        }
        if (!ok) {
            (void)hipGetLastError();
            h->graph_mode = false;
            if (wg_hook("WG_DEBUG")) fprintf(stderr, "[windgym] step-graph capture failed; using direct launches\n");
            launch_step(h, actions_dev, obs_dev, reward_dev, truncated_dev, final_obs_dev, st, false);
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) return fail(WG_ERR_HIP, std::string("wg_step: kernel launch failed: ") + hipGetErrorString(le));
            return 0;
        }
        h->graphs.push_back(ng);
        g = &h->graphs.back();
    }
    g->last_use = ++h->graph_clock;
    HIPCHK(hipGraphLaunch(g->exec, st));
    return 0;
}

extern "C" int wg_set_step_graph(wg_handle h, int enable) {
    if (!h) return fail(WG_ERR_INVALID, "null handle");
    h->graph_mode = enable != 0;
    if (!h->graph_mode) drop_step_graphs(h);
    return 0;
}

extern "C" int wg_check(wg_handle h, void* stream) {
    if (!h) return fail(WG_ERR_INVALID, "null handle");
    if (int rc = use_device(h)) return rc;
    int status = 0;
    HIPCHK(hipMemcpyAsync(&status, h->d.status, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(hipGetLastError());
    // (a bit mask: reported by priority — the reference treats the first two as fatal exceptions)
    if (status & WG_STATUS_BIT_NAN_POWER) return fail(WG_ERR_NAN_POWER, "NaN Power");
    if (status & WG_STATUS_BIT_STATE)
        return fail(WG_ERR_STATE, "step() on a truncated env without reset (or background episode not ready)");
    if (status & WG_STATUS_BIT_RANGE)
        return fail(WG_ERR_RANGE, "a wake particle's emission record left its 16-bit range (k > 0.25: local TI > 0.65, or |hv| > 16 m/s: "
                            "rotor wind speed > 40 m/s) and was saturated");
    return 0;
}

extern "C" int wg_set_obs_multi_buffer(wg_handle h, float* obs_multi_dev) {
    if (!h) return fail(WG_ERR_INVALID, "null handle");
    // k_glue keeps one farm block per wave in LDS for the per-agent packing (wg_launch_glue): it must fit a workgroup's
    // 64 KB by itself (the staged rings give way first)
    if (obs_multi_dev && (size_t)h->p.farm_obs * 4 * WG_NWAVES > 65536)
        return fail(WG_ERR_UNSUPPORTED, "wg_set_obs_multi_buffer: the farm-level observation block is too long for the fused "
                                        "per-agent packing (use wg_obs_multi)");
    if (h->d.multi_out != obs_multi_dev) drop_step_graphs(h);
    h->d.multi_out = obs_multi_dev;     // borrowed; written by k_glue in every following wg_step / wg_reset
    if (int rc = use_device(h)) return rc;
    // (the per-agent instantiation of k_glue neither consumes nor retires the prepared first observations of
    // WgPtrs::next_obs: switching between the two must not leave a flag of an episode that has since been replaced)
    if (h->d.next_obs_ok) {
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemset(h->d.next_obs_ok, 0, sizeof(int) * (size_t)h->p.B * 2));
    }
    return sync_dev_params(h);
}

extern "C" int wg_obs_multi(wg_handle h, float* obs_dev, void* stream) {
    if (!h || !obs_dev) return fail(WG_ERR_INVALID, "null argument");
    if (int rc = use_device(h)) return rc;
    wg_launch_obs_multi(&h->p, &h->d, obs_dev, (hipStream_t)stream);
    return 0;
}

extern "C" int wg_get_measurements(wg_handle h, float* out_dev, void* stream) {
    if (!h || !out_dev) return fail(WG_ERR_INVALID, "null argument");
    if (int rc = use_device(h)) return rc;
    wg_launch_measurements(&h->p, &h->d, out_dev, (hipStream_t)stream);
    return 0;
}

extern "C" int wg_get_windspeed(wg_handle h, int env, int farm, const float* x_dev, int nx, const float* y_dev, int ny,
                                float z, int include_wakes, float* uvw_dev, void* stream) {
    if (!h || !x_dev || !y_dev || !uvw_dev) return fail(WG_ERR_INVALID, "null argument");
    if (int rc = use_device(h)) return rc;
    if (env < 0 || env >= h->p.B || farm < 0 || farm >= h->p.F) return fail(WG_ERR_INVALID, "env / farm index out of range");
    if (nx <= 0 || ny <= 0 || (long long)nx * ny > (1ll << 30)) return fail(WG_ERR_INVALID, "bad grid size");
    if (h->fp.script_rows > 0) return fail(WG_ERR_UNSUPPORTED, "no flow field in flow-script replay mode");
    wg_launch_windspeed(&h->fp, &h->fd, env, farm, x_dev, nx, y_dev, ny, z, include_wakes, uvw_dev, (hipStream_t)stream);
    return 0;
}

extern "C" int wg_get_info(wg_handle h, wg_info_field field, void* out_dev, void* stream) {
    if (!h || !out_dev) return fail(WG_ERR_INVALID, "null argument");
    if (int rc = use_device(h)) return rc;
    if ((int)field < 0 || (int)field > WG_INFO_BOX_ID) return fail(WG_ERR_INVALID, "unknown info field");
    wg_launch_info(&h->p, &h->d, (int)field, out_dev, (hipStream_t)stream);
    return 0;
}

extern "C" int wg_metrics(wg_handle h, float* out_dev, int reset_after, void* stream) {
    if (!h || !out_dev) return fail(WG_ERR_INVALID, "null argument");
    if (int rc = use_device(h)) return rc;
    wg_launch_metrics(&h->p, &h->d, out_dev, reset_after, (hipStream_t)stream);
    return 0;
}

// The state blob starts with a header that identifies the handle geometry it was taken from; the payload is the
// concatenation of the handle's state allocations (a blob only makes sense for an identically configured handle).
// The host-owned wind override table (wg_set_wind) is configuration, not state: re-apply it after wg_set_state.
struct StateHeader {
    uint32_t magic;
    int32_t abi, B, N, F, P, S, K, turb_mode, block, ring_stride, fring_stride, power_avg, n_allocs;
    int32_t res, reserved0, pstride, n_boxes;      // particle-ring layout (uniform / compact rings); reserved0: always 0 (was: k_flow_duo)
    uint64_t payload;
};
static const uint32_t WG_STATE_MAGIC = 0x32474757u;   // "WGG2" (round 6: u_e lives in the packed record; blobs of the "WGGS" layout are refused)
static StateHeader state_header(const wg_env_s* h) {
    StateHeader sh;
    memset(&sh, 0, sizeof(sh));
    sh.magic = WG_STATE_MAGIC; sh.abi = WG_ABI_VERSION;
    sh.B = h->p.B; sh.N = h->p.N; sh.F = h->p.F; sh.P = h->p.P; sh.S = h->p.S; sh.K = h->p.K;
    sh.turb_mode = h->p.turb_mode; sh.block = h->fp.block; sh.ring_stride = h->p.ring_stride;
    sh.fring_stride = h->p.fring_stride; sh.power_avg = h->p.power_avg; sh.n_allocs = (int32_t)h->state_idx.size();
    sh.res = h->fp.res; sh.reserved0 = 0; sh.pstride = h->fp.pstride; sh.n_boxes = h->p.n_boxes;
    for (size_t i : h->state_idx) sh.payload += h->allocs[i].bytes;
    return sh;
}

extern "C" int wg_get_state(wg_handle h, void* blob_host, size_t* size) {
    if (!h || !size) return fail(WG_ERR_INVALID, "null argument");
    const StateHeader sh = state_header(h);
    const size_t total = sizeof(StateHeader) + sh.payload;
    if (!blob_host) {
        *size = total;
        return 0;
    }
    if (*size < total) return fail(WG_ERR_INVALID, "state buffer too small");
    if (int rc = use_device(h)) return rc;
    HIPCHK(hipDeviceSynchronize());
    char* o = (char*)blob_host;
    memcpy(o, &sh, sizeof(sh));
    o += sizeof(sh);
    for (size_t i : h->state_idx) {
        HIPCHK(hipMemcpy(o, h->allocs[i].ptr, h->allocs[i].bytes, hipMemcpyDeviceToHost));
        o += h->allocs[i].bytes;
    }
    *size = total;
    return 0;
}
extern "C" int wg_set_state(wg_handle h, const void* blob_host, size_t size) {
    if (!h || !blob_host) return fail(WG_ERR_INVALID, "null argument");
    const StateHeader want = state_header(h);
    if (size < sizeof(StateHeader)) return fail(WG_ERR_INVALID, "state blob too small");
    StateHeader got;
    memcpy(&got, blob_host, sizeof(got));
    if (got.magic != WG_STATE_MAGIC) return fail(WG_ERR_INVALID, "not a windgym state blob");
    if (memcmp(&got, &want, sizeof(got)) || size != sizeof(StateHeader) + want.payload)
        return fail(WG_ERR_INVALID, "state blob was taken from a differently configured handle (or another ABI version)");
    if (int rc = use_device(h)) return rc;
    HIPCHK(hipDeviceSynchronize());
    const char* o = (const char*)blob_host + sizeof(StateHeader);
    if (h->d.next_obs_ok) HIPCHK(hipMemset(h->d.next_obs_ok, 0, sizeof(int) * (size_t)h->p.B * 2));      // (derived data, not in the blob)
    for (size_t i : h->state_idx) {
        HIPCHK(hipMemcpy(h->allocs[i].ptr, o, h->allocs[i].bytes, hipMemcpyHostToDevice));
        o += h->allocs[i].bytes;
    }
    return 0;
}

extern "C" int wg_kernel_timing(wg_handle h, int enable, double* flow_ms_avg, double* glue_ms_avg, int* n_launches,
                                double* flow_steps_per_launch, double* particles_per_launch) {
    if (!h) return fail(WG_ERR_INVALID, "null handle");
    HIPCHK(hipDeviceSynchronize());
    unsigned long long fs_now = 0, pt_now = 0, ad_now = 0;
    {
        const size_t n_slots = (size_t)h->p.B * 2 * h->p.F;
        std::vector<WgSlot> slots(n_slots);
        HIPCHK(hipMemcpy(slots.data(), h->d.slot, sizeof(WgSlot) * n_slots, hipMemcpyDeviceToHost));
        for (const WgSlot& s : slots) { fs_now += s.flow_count; pt_now += s.part_count; ad_now += s.add_count; }
    }
    double fsum = 0, gsum = 0;
    int nf = 0, ng = 0;
    if (h->ev_used) {
        HIPCHK(hipDeviceSynchronize());
        for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]) != hipSuccess) continue;
            if (h->ev_kind[i / 2] == 0) { fsum += ms; nf++; } else { gsum += ms; ng++; }
        }
    }
    if (flow_ms_avg) *flow_ms_avg = nf ? fsum / nf : 0.0;
    if (glue_ms_avg) *glue_ms_avg = ng ? gsum / ng : 0.0;
    if (n_launches) *n_launches = nf;
    // farm flow-steps per STEP-mode launch since the last call (all launches, sampled or not)
    if (flow_steps_per_launch) *flow_steps_per_launch = h->n_step_launches ? (double)(fs_now - h->flow_steps_mark) / h->n_step_launches : 0.0;
    // particles streamed by the advection passes per launch (chain pruning makes this < farm steps x N x P);
    // the per-slot counters are 32-bit: differences stay exact as long as a slot streams < 2^32 particles per window
    // (counted on the device by the pruning-capable 256-thread variant; the others stream all N x P slots per farm step)
    if (particles_per_launch) {
        const double per_launch = h->n_step_launches ? 1.0 / h->n_step_launches : 0.0;
        *particles_per_launch = (double)(pt_now - h->particles_mark) * per_launch;
    }
    h->added_per_launch = h->n_step_launches ? (double)(ad_now - h->added_mark) / h->n_step_launches : 0.0;
    h->added_mark = ad_now;
    h->n_step_launches = 0;
    h->flow_steps_mark = fs_now;
    h->particles_mark = pt_now;
    h->ev_used = 0;
    h->timing = enable != 0;
    h->timing_period = enable > 1 ? enable : 1;
    h->timing_phase = 0;
    if (h->timing && h->ev.empty()) {          // the event pool, created here so that no timed region ever pays for it
        for (size_t i = 0; i < WG_MAX_TIMING_EVENTS; ++i) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            h->ev.push_back(e);
        }
        h->ev_kind.assign(WG_MAX_TIMING_EVENTS / 2, 0);
    }
    return 0;
}

extern "C" int wg_steady_power(wg_handle h, int model, int n_cases, const float* ws_dev, const float* wd_dev, const float* ti_dev,
                               const float* yaw_dev, float* power_dev, void* stream) {
    if (!h || !ws_dev || !wd_dev || !ti_dev || !yaw_dev || !power_dev) return fail(WG_ERR_INVALID, "null argument");
    if (model != 0 && model != 1) return fail(WG_ERR_INVALID, "wg_steady_power: model must be 0 (steady state of M0) or 1 (Blondel-Cathelain + Jimenez)");
    if (n_cases < 1) return fail(WG_ERR_INVALID, "wg_steady_power: n_cases must be >= 1");
    // model 0 is documented as the steady state of THIS handle's flow model; k_steady carries the Gaussian M0 with TI folding
    // only — a handle created with another deficit / without the folding must not get Gaussian powers silently
    if (model == 0 && (h->deficit_model != 0 || h->no_ti_fold))
        return fail(WG_ERR_UNSUPPORTED, "wg_steady_power: model 0 (steady state of the handle's flow model) is implemented for the Gaussian deficit "
                                        "with wake-TI folding only; this handle uses deficit_model " + std::to_string(h->deficit_model) +
                                        (h->no_ti_fold ? " without TI folding" : ""));
    if (int rc = use_device(h)) return rc;
    SteadyP sp;
    memset(&sp, 0, sizeof(sp));
    sp.n_cases = n_cases; sp.N = h->p.N; sp.S = h->p.S; sp.n_tab = h->p.n_tab; sp.model = model;
    sp.n_quad = model == 0 ? 48 : 20;      // quadrature of the deflection integral (steady.py's defaults)
    sp.D = h->p.D; sp.hub = h->p.hub; sp.R = 0.5f * h->p.D;
    sp.ka = h->p.ka; sp.kb = h->p.kb; sp.eps0 = h->p.eps0; sp.hill = h->p.hill;
    sp.tia = h->p.tia; sp.tib = h->p.tib; sp.tic = h->p.tic; sp.tid = h->p.tid;
    sp.cx0 = h->p.cx0; sp.cy0 = h->p.cy0;
    sp.x_pos = h->d.x_pos; sp.y_pos = h->d.y_pos; sp.rotor_dy = h->d.rotor_dy; sp.rotor_dz = h->d.rotor_dz;
    sp.tab_ws = h->d.tab_ws; sp.tab_power = h->d.tab_power; sp.tab_ct = h->d.tab_ct;
    wg_launch_steady(&sp, ws_dev, wd_dev, ti_dev, yaw_dev, power_dev, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int wg_added_lookups(wg_handle h, double* rotor_points_per_launch) {
    if (!h || !rotor_points_per_launch) return fail(WG_ERR_INVALID, "null argument");
    *rotor_points_per_launch = h->added_per_launch;
    return 0;
}

extern "C" int wg_flow_variant(wg_handle h, int* block, int* compact, int* duo) {
    if (!h) return fail(WG_ERR_INVALID, "null handle");
    if (block) *block = h->fp.block;
    if (compact) *compact = h->fp.res;
    if (duo) *duo = h->fp.envw ? 2 : 0;      // 0: one farm slot per workgroup, 2: k_flow_env (one or two waves per env)
    return 0;
}

extern "C" int wg_algorithmic_bytes(wg_handle h, double* bytes_per_step) {
    if (!h || !bytes_per_step) return fail(WG_ERR_INVALID, "null argument");
    *bytes_per_step = h->alg_bytes;
    return 0;
}
