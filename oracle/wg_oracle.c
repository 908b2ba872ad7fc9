/*
 * wg_oracle.c — ORACLE / TEST INFRASTRUCTURE.  NOT part of the product path.
 *
 * Plain-C, one-env-at-a-time restatement of the batched WindGym step() transition:
 *   - the glue (yaw actuation, baseline controllers, measurement means, MesClass rings / windows / TI /
 *     scaling, rewards, penalties, truncation, reset arithmetic) follows the reference line by line —
 *     each function cites the reference file:line it restates; pinned against golden vectors recorded
 *     from the reference's own code (tests/golden/glue_*.npz, mes_windows.json);
 *   - the flow physics is "model M0" (DESIGN.md §2).  The reference's physics lives in the third-party
 *     package dynamiks @ 77f4f875c6401fde150e351a60143a5b522cf950 (pyproject.toml:24; + hipersim>=0.1.7,
 *     py_wake), which is absent from /root/reference, not installed and not fetchable; the reference's
 *     tests hold no numeric fixture for flow values (tests/test_basics.py asserts shapes/signs only).
 *     => PHYSICS PARITY UNPINNED against DYNAMIKS; pinned only by analytic limits (tests/test_oracle_physics.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp; -DWGO_REAL_FLOAT builds the fp32 variant with the
 * wgof_ prefix so both precisions live in one shared object).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/windgym_hip.h"
#include "pcg64.h"
#include "philox.h"

#ifdef WGO_REAL_FLOAT
typedef float real;
#define WGO(name) wgof_##name
#define R_SQRT sqrtf
#define R_EXP expf
#define R_POW powf
#define R_COS cosf
#define R_SIN sinf
#define R_ATAN atanf
#define R_FLOOR floorf
#define R_FABS fabsf
#else
typedef double real;
#define WGO(name) wgo_##name
#define R_SQRT sqrt
#define R_EXP exp
#define R_POW pow
#define R_COS cos
#define R_SIN sin
#define R_ATAN atan
#define R_FLOOR floor
#define R_FABS fabs
#endif

#define PI_D 3.14159265358979323846

/* ------------------------------------------------------------------------------------------------ */
#define WG_COARSE 4   /* block size of the meandering (particle) box */
#define WGO_MAX_BOXES 64
typedef struct farm_t {
    /* wake-particle chains: ring of P slots per turbine, [N*P], slot index = t*P + r */
    real *py, *pz, *vlp, *wlp;           /* transverse position, low-pass filtered transverse velocity */
    real *ct_e, *k_e, *eps_e, *hv_e, *u_e; /* frozen emission record                                   */
    int head;                            /* ring slot of the newest particle (same for all turbines)    */
    int n_valid;                         /* particles emitted so far, saturating at P                   */
    double s_off;                        /* distance travelled by the newest particle since emission    */
    double time;                         /* fs.time                                                     */
    uint32_t istep;                      /* flow steps since the farm was built (inflow stream counter)  */
    /* turbines [N] */
    real *yaw, *u, *v, *w, *ti_loc, *power, *ct;
    long cursor;                         /* replay mode: row of the scripted table                      */
} farm_t;

typedef struct ctx_t {
    double ws, wd, ti;
    double *xr, *yr;                     /* [N] flow-frame positions                                   */
    double dist, t_inflow;
    int t_developed, time_max;
    double rated_power;
    uint32_t turb_seed;
    double box_ox, box_oy;               /* horizontal offset of this episode into the shared box (m)   */
    int box_id;                          /* which box of the pool this episode uses (MannLoad, :611-618) */
    farm_t farm[2];
    /* MesClass state: rings [4][N][H_c] + farm-level rings ws, wd, power */
    double* ring[WG_N_CH];
    double* fring[WG_N_CH];
    long n_pushed;
    /* "current" values of the last sub-step (info dict) */
    double *cur_ws, *cur_wd;
} ctx_t;

typedef struct env_t {
    ctx_t ctx;
    wgo_pcg64 rng;
    uint64_t noise_key;
    double *farm_pow, *base_pow;         /* deques (maxlen power_avg), persist across episodes (:142-143) */
    long farm_pow_n, base_pow_n;         /* total pushes                                                */
    int timestep;
    int episode;
    int done;                            /* truncated and not reset (autoreset off)                     */
    int nan_power;
    double* old_yaws;
    /* RecordEpisodeVals-style accumulators (wrappers/recordEpisodeVals.py:31-64) */
    double ep_return, ep_power_sum;
    long ep_len;
    int ep_finished;                     /* set by the step that truncated                              */
    double fin_return, fin_mean_power; long fin_len;
    double last_pnow, last_pbase;
} env_t;

typedef struct oracle_t {
    wg_config cfg;
    double *x_pos, *y_pos, *rotor_dy, *rotor_dz, *tab_ws, *tab_power, *tab_ct, *yaw_defined;
    int B, N, F, K, P, S;
    int obs_dim, obs_dim_multi, hist_max;
    env_t* env;
    /* replay mode */
    const double *script_uvw, *script_power;
    long script_rows;
    double metrics[WG_N_METRICS];
    /* optional frozen turbulence box(es) (unit variance), host memory; pool of n_boxes boxes of equal
     * shape for turbtype "MannLoad" (one TF_* file drawn per reset with np_random.choice, :611-618) */
    const float* box;
    const float* box_pool[WGO_MAX_BOXES];
    float* cbox_pool[WGO_MAX_BOXES];
    int n_boxes;
    int bnx, bny, bnz;
    double bdx, bdy, bdz;
    /* meandering box: the same field block-averaged over WG_COARSE^3 cells (model M0 §2.6: the wake particles
     * only feel scales the DWM low-pass filter would keep anyway; the coarse box stays cache-resident) */
    float* cbox;
    int cnx, cny, cnz, coarse;
    /* isotropic box of the wake-added turbulence (wg_config.added_turbulence), unit variance, host memory */
    const int* box_ids;                  /* [B] fixed box per env (FarmEval.update_tf) or NULL */
    const float* abox;
    int anx, any, anz;
    double adx, ady, adz;
    /* deficit_model 2: tabulated eddy-viscosity deficit [n_ct][n_ti][n_x][n_r] (borrowed; include/windgym_hip.h) */
    const float* dtab;
    int an_ct, an_ti, an_x, an_r;
    double an_ct0, an_dct, an_lti0, an_dlti, an_dx, an_dr;      /* node spacings (Ct, ln TI, x / D, r / R) */
} oracle_t;

/* ------------------------------------------------------------------------------------------------ */
static double* dup_d(const double* p, int n) {
    if (!p || n <= 0) return NULL;
    double* q = (double*)malloc(sizeof(double) * (size_t)n);
    memcpy(q, p, sizeof(double) * (size_t)n);
    return q;
}
static real* alloc_r(size_t n) { return (real*)calloc(n ? n : 1, sizeof(real)); }
static double* alloc_d(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

static double defd(double v, double d) { return v == 0.0 ? d : v; }

/* linear interpolation in the tabular turbine (py_wake PowerCtTabular, method='linear'); outside the
 * table: 0 (below cut-in / above cut-out) */
static double tab_interp(const double* xs, const double* ys, int n, double x) {
    if (!(x >= xs[0]) || x > xs[n - 1]) return 0.0;
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (xs[mid] <= x) lo = mid; else hi = mid;
    }
    double f = (x - xs[lo]) / (xs[hi] - xs[lo]);
    return ys[lo] + f * (ys[hi] - ys[lo]);
}

/* ---- observation length ------------------------------------------------------------------------- */
/* MesClass.turb_mes.observed_variables (MesClass.py:248-270) with the flags farm_mes passes (:455-489) */
static int ch_count(const wg_channel* c, int level_on) {
    int cur = c->current && level_on, rol = c->rolling_mean && level_on;
    return cur + rol * c->history_n;
}
static int turb_obs_count(const wg_config* c) {
    return ch_count(&c->ch[WG_CH_WS], c->turb_ws) + ch_count(&c->ch[WG_CH_WD], c->turb_wd) +
           ch_count(&c->ch[WG_CH_YAW], 1) + (c->turb_ti ? 1 : 0) + ch_count(&c->ch[WG_CH_POWER], c->turb_power);
}
/* farm_mes.farm_observed_variables (MesClass.py:447-452) */
static int farm_obs_count(const wg_config* c) {
    return ch_count(&c->ch[WG_CH_WS], c->farm_ws) + ch_count(&c->ch[WG_CH_WD], c->farm_wd) +
           (c->farm_ti ? 1 : 0) + ch_count(&c->ch[WG_CH_POWER], c->farm_power);
}

/* ---- allocation ----------------------------------------------------------------------------------- */
static void farm_alloc(farm_t* f, int N, int P) {
    size_t np = (size_t)N * P;
    f->py = alloc_r(np); f->pz = alloc_r(np); f->vlp = alloc_r(np); f->wlp = alloc_r(np);
    f->ct_e = alloc_r(np); f->k_e = alloc_r(np); f->eps_e = alloc_r(np); f->hv_e = alloc_r(np);
    f->u_e = alloc_r(np);
    f->yaw = alloc_r(N); f->u = alloc_r(N); f->v = alloc_r(N); f->w = alloc_r(N);
    f->ti_loc = alloc_r(N); f->power = alloc_r(N); f->ct = alloc_r(N);
}
static void farm_free(farm_t* f) {
    free(f->py); free(f->pz); free(f->vlp); free(f->wlp);
    free(f->ct_e); free(f->k_e); free(f->eps_e); free(f->hv_e); free(f->u_e);
    free(f->yaw); free(f->u); free(f->v); free(f->w); free(f->ti_loc); free(f->power); free(f->ct);
}

void* WGO(create)(const wg_config* cfg) {
    if (!cfg || cfg->abi_version != WG_ABI_VERSION) return NULL;
    oracle_t* o = (oracle_t*)calloc(1, sizeof(oracle_t));
    o->cfg = *cfg;
    wg_config* c = &o->cfg;
    o->B = c->n_envs; o->N = c->n_turb; o->F = c->n_farms; o->K = c->k_sub; o->P = c->n_particles;
    o->S = c->n_rotor_pts;
    o->x_pos = dup_d(cfg->x_pos, o->N); o->y_pos = dup_d(cfg->y_pos, o->N);
    o->rotor_dy = dup_d(cfg->rotor_dy, o->S); o->rotor_dz = dup_d(cfg->rotor_dz, o->S);
    o->tab_ws = dup_d(cfg->tab_ws, cfg->n_tab); o->tab_power = dup_d(cfg->tab_power, cfg->n_tab);
    o->tab_ct = dup_d(cfg->tab_ct, cfg->n_tab);
    o->yaw_defined = dup_d(cfg->yaw_defined, o->N);
    c->m0_ka = defd(c->m0_ka, 0.38); c->m0_kb = defd(c->m0_kb, 0.004); c->m0_eps = defd(c->m0_eps, 0.2);
    c->m0_hill = defd(c->m0_hill, 0.4);
    c->m0_ti_a = defd(c->m0_ti_a, 0.73); c->m0_ti_b = defd(c->m0_ti_b, 0.8325);
    c->m0_ti_c = defd(c->m0_ti_c, 0.0325); c->m0_ti_d = defd(c->m0_ti_d, -0.32);
    c->m0_fc_scale = defd(c->m0_fc_scale, 2.0);
    c->m0_km1 = defd(c->m0_km1, 0.6); c->m0_km2 = defd(c->m0_km2, 0.35);
    c->m0_sg_af = defd(c->m0_sg_af, 3.11); c->m0_sg_bf = defd(c->m0_sg_bf, -0.68); c->m0_sg_cf = defd(c->m0_sg_cf, 2.41);
    o->obs_dim = turb_obs_count(c) * o->N + farm_obs_count(c);
    /* per-agent vector of WindFarmEnvMulti._get_obs_multi (WindEnvMulti.py:79-103): own turbine block ++
     * farm_mes.farm_mes.get_measurements(); the farm object's yaw deque is never filled, so it contributes
     * nothing */
    o->obs_dim_multi = turb_obs_count(c) + farm_obs_count(c);
    /* turb_mes.max_hist ignores the power history (MesClass.py:239-246) */
    o->hist_max = c->ch[WG_CH_WS].history_len;
    if (c->ch[WG_CH_WD].history_len > o->hist_max) o->hist_max = c->ch[WG_CH_WD].history_len;
    if (c->ch[WG_CH_YAW].history_len > o->hist_max) o->hist_max = c->ch[WG_CH_YAW].history_len;
    o->env = (env_t*)calloc((size_t)o->B, sizeof(env_t));
    for (int b = 0; b < o->B; ++b) {
        env_t* e = &o->env[b];
        ctx_t* x = &e->ctx;
        x->xr = alloc_d(o->N); x->yr = alloc_d(o->N);
        for (int f = 0; f < o->F; ++f) farm_alloc(&x->farm[f], o->N, o->P);
        for (int ch = 0; ch < WG_N_CH; ++ch) {
            x->ring[ch] = alloc_d((size_t)o->N * c->ch[ch].history_len);
            x->fring[ch] = alloc_d((size_t)c->ch[ch].history_len);
        }
        x->cur_ws = alloc_d(o->N); x->cur_wd = alloc_d(o->N);
        e->farm_pow = alloc_d(c->power_avg); e->base_pow = alloc_d(c->power_avg);
        e->old_yaws = alloc_d(o->N);
        wgo_pcg64_seed(&e->rng, (uint64_t)b);
        e->noise_key = (uint64_t)b;
        e->done = 1; /* must be reset first */
    }
    return o;
}

void WGO(destroy)(void* h) {
    oracle_t* o = (oracle_t*)h;
    if (!o) return;
    for (int b = 0; b < o->B; ++b) {
        env_t* e = &o->env[b];
        ctx_t* x = &e->ctx;
        free(x->xr); free(x->yr);
        for (int f = 0; f < o->F; ++f) farm_free(&x->farm[f]);
        for (int ch = 0; ch < WG_N_CH; ++ch) { free(x->ring[ch]); free(x->fring[ch]); }
        free(x->cur_ws); free(x->cur_wd);
        free(e->farm_pow); free(e->base_pow); free(e->old_yaws);
    }
    free(o->env);
    for (int k = 0; k < WGO_MAX_BOXES; ++k) free(o->cbox_pool[k]);
    free(o->x_pos); free(o->y_pos); free(o->rotor_dy); free(o->rotor_dz);
    free(o->tab_ws); free(o->tab_power); free(o->tab_ct); free(o->yaw_defined);
    free(o);
}

int WGO(obs_dim)(void* h, int* obs_dim, int* obs_dim_multi) {
    oracle_t* o = (oracle_t*)h;
    if (obs_dim) *obs_dim = o->obs_dim;
    if (obs_dim_multi) *obs_dim_multi = o->obs_dim_multi;
    return 0;
}
int WGO(hist_max)(void* h) { return ((oracle_t*)h)->hist_max; }

void WGO(set_flow_script)(void* h, const double* uvw, const double* power, long n_rows) {
    oracle_t* o = (oracle_t*)h;
    o->script_uvw = uvw; o->script_power = power; o->script_rows = n_rows;
}
static float* coarsen_box(oracle_t* o, const float* box, int nx, int ny, int nz) {
    const size_t nc = (size_t)o->cnx * o->cny * o->cnz, nf = (size_t)nx * ny * nz;
    float* cb = (float*)malloc(sizeof(float) * 3 * nc);
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < o->cnx; ++i)
            for (int j = 0; j < o->cny; ++j)
                for (int k = 0; k < o->cnz; ++k) {
                    /* fixed summation order (x, y, z innermost) in float, like the device kernel */
                    float acc = 0.f;
                    for (int a = 0; a < WG_COARSE; ++a)
                        for (int b = 0; b < WG_COARSE; ++b)
                            for (int cc = 0; cc < WG_COARSE; ++cc)
                                acc += box[c * nf + ((size_t)(i * WG_COARSE + a) * ny + (j * WG_COARSE + b)) * nz + (k * WG_COARSE + cc)];
                    cb[c * nc + ((size_t)i * o->cny + j) * o->cnz + k] = acc * (1.0f / (WG_COARSE * WG_COARSE * WG_COARSE));
                }
    return cb;
}
/* pool of n_boxes boxes of equal shape (n_boxes = 1: the single shared box of MannFixed / MannGenerate) */
int WGO(set_turbulence_boxes)(void* h, const float* const* boxes, int n_boxes, int nx, int ny, int nz, double dx,
                              double dy, double dz) {
    oracle_t* o = (oracle_t*)h;
    if (n_boxes < 1 || n_boxes > WGO_MAX_BOXES) return -1;
    o->box = boxes[0]; o->bnx = nx; o->bny = ny; o->bnz = nz; o->bdx = dx; o->bdy = dy; o->bdz = dz;
    o->n_boxes = n_boxes;
    o->coarse = (nx % WG_COARSE == 0 && ny % WG_COARSE == 0 && nz % WG_COARSE == 0 &&
                 nx >= 2 * WG_COARSE && ny >= 2 * WG_COARSE && nz >= 2 * WG_COARSE);
    if (o->coarse) { o->cnx = nx / WG_COARSE; o->cny = ny / WG_COARSE; o->cnz = nz / WG_COARSE; }
    for (int k = 0; k < WGO_MAX_BOXES; ++k) { free(o->cbox_pool[k]); o->cbox_pool[k] = NULL; o->box_pool[k] = NULL; }
    for (int k = 0; k < n_boxes; ++k) {
        o->box_pool[k] = boxes[k];
        if (o->coarse) o->cbox_pool[k] = coarsen_box(o, boxes[k], nx, ny, nz);
    }
    return 0;
}
void WGO(set_box_ids)(void* h, const int* ids) { ((oracle_t*)h)->box_ids = ids; }   /* borrowed */
/* isotropic box of the wake-added turbulence: [3][nx][ny][nz] float, z fastest, unit variance (borrowed) */
void WGO(set_added_turbulence_box)(void* h, const float* box, int nx, int ny, int nz, double dx, double dy,
                                   double dz) {
    oracle_t* o = (oracle_t*)h;
    o->abox = box; o->anx = nx; o->any = ny; o->anz = nz; o->adx = dx; o->ady = dy; o->adz = dz;
}
void WGO(set_deficit_table)(void* h, const float* tab, int n_ct, double ct0, double ct1, int n_ti, double ti0, double ti1,
                            int n_x, double x_max_D, int n_r, double r_max_R) {
    oracle_t* o = (oracle_t*)h;
    o->dtab = tab; o->an_ct = n_ct; o->an_ti = n_ti; o->an_x = n_x; o->an_r = n_r;
    o->an_ct0 = ct0; o->an_dct = (ct1 - ct0) / (n_ct - 1);
    o->an_lti0 = log(ti0); o->an_dlti = (log(ti1) - log(ti0)) / (n_ti - 1);
    o->an_dx = x_max_D / (n_x - 1); o->an_dr = r_max_R / (n_r - 1);
}
void WGO(set_turbulence_box)(void* h, const float* box, int nx, int ny, int nz, double dx, double dy,
                             double dz) {
    const float* one[1] = {box};
    WGO(set_turbulence_boxes)(h, one, 1, nx, ny, nz, dx, dy, dz);
}

/* ================================================================================================== */
/* Model M0 — flow physics                                                             */
/* ================================================================================================== */

/* centreline deficit fraction C(x) of a Gaussian wake with linear expansion, capped in the near wake
 * at the momentum-theory value 1-sqrt(1-ct) (Bastankhah & Porte-Agel 2014; Niayifar & Porte-Agel 2016):
 *   sigma/D = k*x/D + eps ,  C = 1 - sqrt(1 - ct*min(1, 1/(8 (sigma/D)^2)))                           */
static inline real m0_sigma_over_d(real k, real eps, real x_over_d) { return k * x_over_d + eps; }
static inline real m0_cfrac(real ct, real sp) {
    real m = (real)1 / ((real)8 * sp * sp);
    if (m > (real)1) m = (real)1;
    real a = (real)1 - ct * m;
    if (a < (real)0) a = (real)0;
    return (real)1 - R_SQRT(a);
}

/* Super-Gaussian wake (wg_config.deficit_model = 1; Blondel & Cathelain 2020, "An alternative form of the
 * super-Gaussian wind turbine wake model", Wind Energ. Sci. 5, eqs. 5-7 — the model behind the reference's PyWakeAgent):
 *   dU / U = C exp(-(r/D)^n / (2 (sigma/D)^2)),  n = af exp(bf x/D) + cf,
 *   C = 2^(2/n - 1) - sqrt(2^(4/n - 2) - n ct / (16 Gamma(2/n) (sigma/D)^(4/n)))   (mass + momentum conservation);
 * n = 2 is the Gaussian wake.  sigma/D keeps M0's form k x/D + eps. */
static inline real m0_sg_cfrac(real ct, real sp, real n) {
    real a1 = R_POW((real)2, (real)2 / n - (real)1);
    real a2 = R_POW((real)2, (real)4 / n - (real)2);
    real g = (real)tgamma((double)((real)2 / n));
    real a = a2 - n * ct / ((real)16 * g * R_POW(sp, (real)4 / n));
    if (a < (real)0) a = (real)0;
    return a1 - R_SQRT(a);
}

/* Tabulated eddy-viscosity deficit (wg_config.deficit_model = 2; wg_set_deficit_table): the profile 1 - U / U0 at
 * (Ct, TI_amb, x / D) is 3-linear between the table's nodes — an_setup keeps the 8 corner weights / offsets of one wake;
 * TI_amb is what the particle's frozen growth rate encodes (k = ka TI + kb).  an_frac samples it at r / R: linear between
 * the r nodes, 0 from half a node inside the table's edge on; *slope = |d frac / d (r / R)| as the centred difference of that
 * piecewise-linear profile over one node spacing, [f(fr + 1/2) - f(fr - 1/2)] / dr (continuous in r; node -1 mirrors node 1). */
typedef struct { real w[8]; size_t o[8]; } an_wake;
static void an_setup(const oracle_t* o, real ctv, real kv, real xd, an_wake* aw) {
    real tiw = (kv - (real)o->cfg.m0_kb) / (real)o->cfg.m0_ka;
    if (tiw < (real)1e-6) tiw = (real)1e-6;
    real fq[3] = {(ctv - (real)o->an_ct0) / (real)o->an_dct, ((real)log((double)tiw) - (real)o->an_lti0) / (real)o->an_dlti,
                  xd / (real)o->an_dx};
    const int nq[3] = {o->an_ct, o->an_ti, o->an_x};
    const size_t sq[3] = {(size_t)o->an_ti * o->an_x * o->an_r, (size_t)o->an_x * o->an_r, (size_t)o->an_r};
    int iq[3]; real wq[3];
    for (int a = 0; a < 3; ++a) {
        if (fq[a] < (real)0) fq[a] = (real)0;
        if (fq[a] > (real)(nq[a] - 1)) fq[a] = (real)(nq[a] - 1);
        iq[a] = (int)fq[a]; if (iq[a] > nq[a] - 2) iq[a] = nq[a] - 2;
        wq[a] = fq[a] - (real)iq[a];
    }
    for (int q = 0; q < 8; ++q) {
        aw->w[q] = (q & 1 ? wq[0] : (real)1 - wq[0]) * (q & 2 ? wq[1] : (real)1 - wq[1]) * (q & 4 ? wq[2] : (real)1 - wq[2]);
        aw->o[q] = (iq[0] + (q & 1 ? 1 : 0)) * sq[0] + (iq[1] + (q & 2 ? 1 : 0)) * sq[1] + (iq[2] + (q & 4 ? 1 : 0)) * sq[2];
    }
}
static real an_frac(const oracle_t* o, const an_wake* aw, real r_over_R, real* slope) {
    real fr = r_over_R / (real)o->an_dr;
    *slope = 0;
    if (!(fr < (real)o->an_r - (real)1.5)) return 0;
    int m = (int)(fr + (real)0.5), ml = m > 0 ? m - 1 : 1;
    real fa = 0, fb = 0, fc = 0;
    for (int q = 0; q < 8; ++q) {
        fa += aw->w[q] * o->dtab[aw->o[q] + ml]; fb += aw->w[q] * o->dtab[aw->o[q] + m];
        fc += aw->w[q] * o->dtab[aw->o[q] + m + 1];
    }
    real e = fr - (real)m;
    real hi = fb + (e + (real)0.5) * (fc - fb), lo = fa + (e + (real)0.5) * (fb - fa);
    *slope = R_FABS(hi - lo) / (real)o->an_dr;
    return e < (real)0 ? fb + e * (fb - fa) : fb + e * (fc - fb);
}

/* trilinear, periodic lookup of one component of the frozen box at (x,y,z) metres; cell coordinates are
 * formed in double precision (x - U t reaches 1e5 m), the interpolation weights in `real` */
static inline real cbox_lookup(const oracle_t* o, int bid, int comp, double x, double y, double z) {
    /* coarse cell i averages fine cells [4i, 4i+3], i.e. it is centred at fine index 4i + 1.5 */
    const double h = 0.5 * (WG_COARSE - 1);
    double fx = (x / o->bdx - h) / WG_COARSE, fy = (y / o->bdy - h) / WG_COARSE, fz = (z / o->bdz - h) / WG_COARSE;
    double ix = floor(fx), iy = floor(fy), iz = floor(fz);
    real tx = (real)(fx - ix), ty = (real)(fy - iy), tz = (real)(fz - iz);
    long i0 = (long)fmod(ix, (double)o->cnx); if (i0 < 0) i0 += o->cnx;
    long j0 = (long)fmod(iy, (double)o->cny); if (j0 < 0) j0 += o->cny;
    long k0 = (long)fmod(iz, (double)o->cnz); if (k0 < 0) k0 += o->cnz;
    long i1 = (i0 + 1) % o->cnx, j1 = (j0 + 1) % o->cny, k1 = (k0 + 1) % o->cnz;
    const float* p = o->cbox_pool[bid] + (size_t)comp * o->cnx * o->cny * o->cnz;
#define BX(i, j, k) ((real)p[((size_t)(i) * o->cny + (j)) * o->cnz + (k)])
    real c00 = BX(i0, j0, k0) + tx * (BX(i1, j0, k0) - BX(i0, j0, k0));
    real c10 = BX(i0, j1, k0) + tx * (BX(i1, j1, k0) - BX(i0, j1, k0));
    real c01 = BX(i0, j0, k1) + tx * (BX(i1, j0, k1) - BX(i0, j0, k1));
    real c11 = BX(i0, j1, k1) + tx * (BX(i1, j1, k1) - BX(i0, j1, k1));
#undef BX
    real c0 = c00 + ty * (c10 - c00);
    real c1 = c01 + ty * (c11 - c01);
    return c0 + tz * (c1 - c0);
}

static inline real box_lookup(const oracle_t* o, int bid, int comp, double x, double y, double z) {
    double fx = x / o->bdx, fy = y / o->bdy, fz = z / o->bdz;
    double ix = floor(fx), iy = floor(fy), iz = floor(fz);
    real tx = (real)(fx - ix), ty = (real)(fy - iy), tz = (real)(fz - iz);
    long i0 = (long)fmod(ix, (double)o->bnx); if (i0 < 0) i0 += o->bnx;
    long j0 = (long)fmod(iy, (double)o->bny); if (j0 < 0) j0 += o->bny;
    long k0 = (long)fmod(iz, (double)o->bnz); if (k0 < 0) k0 += o->bnz;
    long i1 = (i0 + 1) % o->bnx, j1 = (j0 + 1) % o->bny, k1 = (k0 + 1) % o->bnz;
    const float* p = o->box_pool[bid] + (size_t)comp * o->bnx * o->bny * o->bnz;
#define BX(i, j, k) ((real)p[((size_t)(i) * o->bny + (j)) * o->bnz + (k)])
    real c00 = BX(i0, j0, k0) + tx * (BX(i1, j0, k0) - BX(i0, j0, k0));
    real c10 = BX(i0, j1, k0) + tx * (BX(i1, j1, k0) - BX(i0, j1, k0));
    real c01 = BX(i0, j0, k1) + tx * (BX(i1, j0, k1) - BX(i0, j0, k1));
    real c11 = BX(i0, j1, k1) + tx * (BX(i1, j1, k1) - BX(i0, j1, k1));
#undef BX
    real c0 = c00 + ty * (c10 - c00);
    real c1 = c01 + ty * (c11 - c01);
    return c0 + tz * (c1 - c0);
}

/* the same lookup in the added-turbulence box */
static inline real abox_lookup(const oracle_t* o, int comp, double x, double y, double z) {
    double fx = x / o->adx, fy = y / o->ady, fz = z / o->adz;
    double ix = floor(fx), iy = floor(fy), iz = floor(fz);
    real tx = (real)(fx - ix), ty = (real)(fy - iy), tz = (real)(fz - iz);
    long i0 = (long)fmod(ix, (double)o->anx); if (i0 < 0) i0 += o->anx;
    long j0 = (long)fmod(iy, (double)o->any); if (j0 < 0) j0 += o->any;
    long k0 = (long)fmod(iz, (double)o->anz); if (k0 < 0) k0 += o->anz;
    long i1 = (i0 + 1) % o->anx, j1 = (j0 + 1) % o->any, k1 = (k0 + 1) % o->anz;
    const float* p = o->abox + (size_t)comp * o->anx * o->any * o->anz;
#define BX(i, j, k) ((real)p[((size_t)(i) * o->any + (j)) * o->anz + (k)])
    real c00 = BX(i0, j0, k0) + tx * (BX(i1, j0, k0) - BX(i0, j0, k0));
    real c10 = BX(i0, j1, k0) + tx * (BX(i1, j1, k0) - BX(i0, j1, k0));
    real c01 = BX(i0, j0, k1) + tx * (BX(i1, j0, k1) - BX(i0, j0, k1));
    real c11 = BX(i0, j1, k1) + tx * (BX(i1, j1, k1) - BX(i0, j1, k1));
#undef BX
    real c0 = c00 + ty * (c10 - c00);
    real c1 = c01 + ty * (c11 - c01);
    return c0 + tz * (c1 - c0);
}

static inline int has_box(const oracle_t* o) {
    return (o->cfg.turb_mode == WG_TURB_BOX || o->cfg.turb_mode == WG_TURB_BOX_SHIFT || o->cfg.turb_mode == WG_TURB_BOX_POOL) && o->box;
}

/* frozen-box fluctuation (one component) at a point of the flow frame at time `time`: Taylor's hypothesis,
 * the unit-variance box is scaled to TI*U (MannTurbulenceField.scale_TI, Wind_Farm_Env.py:617, :637, :658) */
static inline real box_fluct(const oracle_t* o, const ctx_t* x, int comp, double time, double px, double py,
                             double pz) {
    return (real)(x->ti * x->ws) * box_lookup(o, x->box_id, comp, px - x->ws * time + x->box_ox, py + x->box_oy, pz);
}
/* the same for the wake particles: coarse (block-averaged) box when the box is divisible by WG_COARSE */
static inline real box_fluct_particle(const oracle_t* o, const ctx_t* x, int comp, double time, double px, double py,
                                      double pz) {
    const double bx = px - x->ws * time + x->box_ox, by = py + x->box_oy;
    return (real)(x->ti * x->ws) * (o->coarse ? cbox_lookup(o, x->box_id, comp, bx, by, pz) : box_lookup(o, x->box_id, comp, bx, by, pz));
}

/* what a particle emitted *now* by turbine t would carry (uses the turbine's last rotor wind and its
 * current yaw) */
static void m0_record_now(const oracle_t* o, const farm_t* f, int t, real* ct, real* k, real* eps, real* hv,
                          real* ue) {
    const wg_config* c = &o->cfg;
    real g = f->yaw[t] * (real)(PI_D / 180.0);
    real cg = R_COS(g), sg = R_SIN(g);
    real wsn = f->u[t] * cg + f->v[t] * sg;
    if (wsn < 0) wsn = 0;
    real ct0 = (real)tab_interp(o->tab_ws, o->tab_ct, c->n_tab, (double)wsn);
    real ctx_ = ct0 * cg * cg;
    if (ctx_ > (real)0.96) ctx_ = (real)0.96;
    if (ctx_ < 0) ctx_ = 0;
    real q = R_SQRT((real)1 - ctx_);
    real beta = (real)0.5 * ((real)1 + q) / q;
    *ct = ctx_;
    *k = (real)c->m0_ka * f->ti_loc[t] + (real)c->m0_kb;
    *eps = (real)c->m0_eps * R_SQRT(beta);
    *hv = -(real)c->m0_hill * sg * f->u[t];
    *ue = f->u[t];
}

/* one DWMFlowSimulation.step() of model M0 for one farm */
static void m0_flow_step(oracle_t* o, env_t* e, int fi) {
    const wg_config* c = &o->cfg;
    ctx_t* x = &e->ctx;
    farm_t* f = &x->farm[fi];
    const int N = o->N, P = o->P, S = o->S;
    const real D = (real)c->rotor_diameter, dt = (real)c->dt_sim;
    const real inv_D = (real)1 / D;
    const double dpart = c->d_particle * c->rotor_diameter;
    const double inv_dpart = 1.0 / dpart;
    const real R_rot = (real)0.5 * D;
    const real hub = (real)c->hub_height;

    /* (1) emission records of this step */
    real rct[N], rk[N], reps[N], rhv[N], rue[N];
    for (int t = 0; t < N; ++t) m0_record_now(o, f, t, &rct[t], &rk[t], &reps[t], &rhv[t], &rue[t]);

    /* (2) advect every particle over dt: Hill-vortex-style lateral self-induced speed hv*C(x) plus the
     * low-pass filtered ambient transverse velocity at the particle (meandering) */
    const int box = has_box(o);
    const int rnd = (c->turb_mode == WG_TURB_RANDOM);
    const real sig_amb = (real)(x->ti * x->ws);
    real alpha = 0;
    if (box || rnd) {
        /* meandering: first-order low pass of the transverse inflow at the particle, cut-off U/(2D) (DWM) */
        double fc = x->ws / (c->m0_fc_scale * c->rotor_diameter);
        alpha = (real)(1.0 - exp(-2.0 * PI_D * fc * c->dt_sim));
    }
    for (int t = 0; t < N; ++t) {
        for (int r = 0; r < P; ++r) {
            int j = f->head - r; if (j < 0) j += P;      /* age index of ring slot r */
            if (j >= f->n_valid) continue;
            size_t i = (size_t)t * P + r;
            real xrel = (real)f->s_off + (real)j * (real)dpart;
            real sp = m0_sigma_over_d(f->k_e[i], f->eps_e[i], xrel * inv_D);
            real cf = m0_cfrac(f->ct_e[i], sp);
            real vy = f->hv_e[i] * cf;
            real vz = 0;
            if (box || rnd) {
                real fv, fw;
                if (box) {
                    fv = box_fluct_particle(o, x, 1, f->time, x->xr[t] + (double)xrel, (double)f->py[i], (double)f->pz[i]);
                    fw = box_fluct_particle(o, x, 2, f->time, x->xr[t] + (double)xrel, (double)f->py[i], (double)f->pz[i]);
                } else {
                    fv = sig_amb * (real)wgo_turb_normal(x->turb_seed, f->istep, (uint32_t)i, 1, 0x50);
                    fw = sig_amb * (real)wgo_turb_normal(x->turb_seed, f->istep, (uint32_t)i, 2, 0x50);
                }
                f->vlp[i] += alpha * (fv - f->vlp[i]);
                f->wlp[i] += alpha * (fw - f->wlp[i]);
                vy += f->vlp[i];
                vz += f->wlp[i];
            }
            f->py[i] += vy * dt;
            f->pz[i] += vz * dt;
        }
    }
    /* (3) travel + emission: particles are released every d_particle*D of travel, at the exact distance,
     * so a chain stays equispaced: particle of age j sits at x_t + s_off + j*dpart */
    f->time += c->dt_sim;
    f->istep += 1;
    f->s_off += x->ws * c->dt_sim;
    while (f->s_off >= dpart) {
        f->s_off -= dpart;
        f->head = (f->head + 1) % P;
        if (f->n_valid < P) f->n_valid++;
        for (int t = 0; t < N; ++t) {
            size_t i = (size_t)t * P + f->head;
            f->py[i] = (real)x->yr[t]; f->pz[i] = hub; f->vlp[i] = 0; f->wlp[i] = 0;
            f->ct_e[i] = rct[t]; f->k_e[i] = rk[t]; f->eps_e[i] = reps[t]; f->hv_e[i] = rhv[t];
            f->u_e[i] = rue[t];
        }
    }
    /* (4) rotor-averaged inflow of every turbine: ambient minus linearly superposed Gaussian deficits of
     * all upstream chains, wake centre / record interpolated between the two bracketing particles */
    /* wake-added turbulence (reference: [Synchronized]AutoScalingIsotropicMannTurbulence, Wind_Farm_Env.py:618, :638,
     * :644, :659; restated from the DWM literature — Madsen, Larsen G.C., Larsen T.J., Troldborg & Mikkelsen 2010,
     * "Calibration and validation of the dynamic wake meandering model", J. Sol. Energy Eng. 132, eq. (11); IEC
     * 61400-1 ed. 4 Annex E): a unit-variance ISOTROPIC small-scale field g, frozen and advected with the ambient
     * field's clock and offset ("Synchronized"), is scaled inside every wake by
     *     k_mt(r) = km1 |dU(r)| / U + km2 |d (dU / U) / d (r / R)| ,
     * dU the wake's deficit at the point, r the distance from the wake centre, R the rotor radius.  At rotor point p of
     * target t:  uvw_added = U sum_s k_mt,s(p) g(x_t - U t + o_x, y_p + o_y, z_p)  (per source wake, superposed
     * linearly like the deficits); the rotor average enters (u, v, w)_t.  Gaussian wake: dU = A E, |d dU / dr| = A E r / sigma^2. */
    const int added = c->added_turbulence && (box || rnd) && o->abox;
    const real km1 = (real)c->m0_km1, km2 = (real)c->m0_km2;
    for (int t = 0; t < N; ++t) {
        real g = f->yaw[t] * (real)(PI_D / 180.0);
        real cg = R_COS(g);
        real dsum = 0, tiadd_max = 0;
        real gadd[3][S > 0 ? S : 1], addsum[3] = {0, 0, 0};
        if (added) {
            for (int s = 0; s < S; ++s)
                for (int cc = 0; cc < 3; ++cc)
                    gadd[cc][s] = abox_lookup(o, cc, x->xr[t] - x->ws * f->time + x->box_ox,
                                              x->yr[t] + (double)((real)o->rotor_dy[s] * cg) + x->box_oy,
                                              c->hub_height + o->rotor_dz[s]);
        }
        for (int s2 = 0; s2 < N; ++s2) {
            if (s2 == t) continue;
            double dx = x->xr[t] - x->xr[s2];
            if (!(dx > 0.0)) continue;
            double xi = (dx - f->s_off) * inv_dpart;
            double jf = floor(xi);
            real wgt = (real)(xi - jf);
            long j = (long)jf;
            if (j < 0) { j = 0; wgt = 0; }
            if (j + 1 > f->n_valid - 1) continue;     /* wake front has not arrived yet */
            int r0 = f->head - (int)j; r0 %= P; if (r0 < 0) r0 += P;
            int r1 = r0 - 1; if (r1 < 0) r1 += P;
            size_t i0 = (size_t)s2 * P + r0, i1 = (size_t)s2 * P + r1;
            real w0 = (real)1 - wgt, w1 = wgt;
            real yc = w0 * f->py[i0] + w1 * f->py[i1];
            real zc = w0 * f->pz[i0] + w1 * f->pz[i1];
            real ctv = w0 * f->ct_e[i0] + w1 * f->ct_e[i1];
            real kv = w0 * f->k_e[i0] + w1 * f->k_e[i1];
            real epv = w0 * f->eps_e[i0] + w1 * f->eps_e[i1];
            real uev = w0 * f->u_e[i0] + w1 * f->u_e[i1];
            real xd = (real)dx * inv_D;
            real sp = m0_sigma_over_d(kv, epv, xd);
            const int sg = c->deficit_model == 1;
            const real nsg = sg ? (real)c->m0_sg_af * R_EXP((real)c->m0_sg_bf * xd) + (real)c->m0_sg_cf : (real)2;
            real cf = sg ? m0_sg_cfrac(ctv, sp, nsg) : m0_cfrac(ctv, sp);
            real sig = sp * D;
            /* lateral cut-off: a wake whose centre is farther than R + 5 sigma from the rotor centre is
             * neglected (its Gaussian tail is < exp(-12.5) of the centreline deficit) */
            real rc2 = ((real)x->yr[t] - yc) * ((real)x->yr[t] - yc) + (hub - zc) * (hub - zc);
            real rcut = R_rot + (real)5 * sig;
            if (rc2 > rcut * rcut) continue;
            real inv2s2 = (real)1 / ((real)2 * sig * sig);
            real amp = uev * cf;
            const int an = c->deficit_model == 2 && o->dtab;
            an_wake aw;
            if (an) an_setup(o, ctv, kv, xd, &aw);
            for (int s = 0; s < S; ++s) {
                real ys = (real)x->yr[t] + (real)o->rotor_dy[s] * cg;
                real zs = hub + (real)o->rotor_dz[s];
                real r2 = (ys - yc) * (ys - yc) + (zs - zc) * (zs - zc);
                real du, grad;            /* grad = |d dU / dr| / dU */
                if (an) {
                    real slope;
                    real frac = an_frac(o, &aw, R_SQRT(r2) / R_rot, &slope);
                    du = uev * frac;
                    dsum += du;
                    if (added) {
                        /* U k_mt = km1 dU + km2 R |d dU / dr| (slope per R: an_frac) */
                        real wk = uev * (km1 * frac + km2 * slope);
                        for (int cc = 0; cc < 3; ++cc) addsum[cc] += wk * gadd[cc][s];
                    }
                    continue;
                }
                if (sg) {
                    /* (r/D)^n / (2 (sigma/D)^2); d/dr: n (r/D)^n / (r 2 (sigma/D)^2) */
                    real rn = r2 > (real)0 ? R_POW(r2 * inv_D * inv_D, (real)0.5 * nsg) : (real)0;
                    real inv2sp2 = (real)1 / ((real)2 * sp * sp);
                    du = amp * R_EXP(-rn * inv2sp2);
                    grad = r2 > (real)0 ? nsg * rn * inv2sp2 / R_SQRT(r2) : (real)0;
                } else {
                    du = amp * R_EXP(-r2 * inv2s2);
                    grad = R_SQRT(r2) * ((real)2 * inv2s2);
                }
                dsum += du;
                if (added) {
                    /* U k_mt = km1 dU + km2 R |d dU / dr| = dU (km1 + km2 R r / sigma^2) for the Gaussian */
                    real wk = du * (km1 + km2 * R_rot * grad);
                    for (int cc = 0; cc < 3; ++cc) addsum[cc] += wk * gadd[cc][s];
                }
            }
            /* wake-added turbulence (Crespo & Hernandez 1996), weighted by the Gaussian at the hub */
            real ind = (real)0.5 * ((real)1 - R_SQRT((real)1 - ctv));
            real xdc = xd < (real)1 ? (real)1 : xd;
            real tia = (real)c->m0_ti_a * R_POW(ind, (real)c->m0_ti_b) * R_POW((real)x->ti, (real)c->m0_ti_c) *
                       R_POW(xdc, (real)c->m0_ti_d) * R_EXP(-rc2 * inv2s2);
            if (c->no_ti_fold) tia = 0;
            if (tia > tiadd_max) tiadd_max = tia;
        }
        real amb[3] = {0, 0, 0};
        if (box) {
            for (int s = 0; s < S; ++s)
                for (int cc = 0; cc < 3; ++cc)
                    amb[cc] += box_fluct(o, x, cc, f->time, x->xr[t], x->yr[t] + (double)((real)o->rotor_dy[s] * cg),
                                         c->hub_height + o->rotor_dz[s]);
            amb[0] /= (real)S; amb[1] /= (real)S; amb[2] /= (real)S;
        } else if (rnd) {
            /* i.i.d. gusts at the S rotor points: their mean is one normal of variance sigma^2 / S */
            const real sc = sig_amb / R_SQRT((real)S);
            for (int cc = 0; cc < 3; ++cc)
                amb[cc] = sc * (real)wgo_turb_normal(x->turb_seed, f->istep, (uint32_t)t, (uint32_t)cc, 0x52);
        }
        f->u[t] = (real)x->ws + amb[0] - dsum / (real)S + addsum[0] / (real)S;
        f->v[t] = amb[1] + addsum[1] / (real)S;
        f->w[t] = amb[2] + addsum[2] / (real)S;
        f->ti_loc[t] = R_SQRT((real)(x->ti * x->ti) + tiadd_max * tiadd_max);
    }
    /* (5) turbine power / thrust at the new inflow with the current yaw */
    for (int t = 0; t < N; ++t) {
        real g = f->yaw[t] * (real)(PI_D / 180.0);
        real cg = R_COS(g), sg = R_SIN(g);
        real wsn = f->u[t] * cg + f->v[t] * sg;
        if (wsn < 0) wsn = 0;
        f->power[t] = (real)tab_interp(o->tab_ws, o->tab_power, c->n_tab, (double)wsn);
        f->ct[t] = (real)tab_interp(o->tab_ws, o->tab_ct, c->n_tab, (double)wsn) * cg * cg;
    }
}

/* replay mode (test hook): the flow double used to record the golden vectors */
static void script_load(oracle_t* o, env_t* e, int b, int fi) {
    farm_t* f = &e->ctx.farm[fi];
    long row = f->cursor;
    if (row >= o->script_rows) row = o->script_rows - 1;
    size_t base = (((size_t)fi * o->script_rows + row) * o->B + b) * o->N;
    for (int t = 0; t < o->N; ++t) {
        f->u[t] = (real)o->script_uvw[(base + t) * 3 + 0];
        f->v[t] = (real)o->script_uvw[(base + t) * 3 + 1];
        f->w[t] = (real)o->script_uvw[(base + t) * 3 + 2];
        f->power[t] = (real)o->script_power[base + t];
    }
}
static void flow_step(oracle_t* o, env_t* e, int b, int fi) {
    if (o->script_uvw) {
        farm_t* f = &e->ctx.farm[fi];
        f->cursor++;
        f->time += o->cfg.dt_sim;
        script_load(o, e, b, fi);
    } else {
        m0_flow_step(o, e, fi);
    }
}

/* ================================================================================================== */
/* Glue — restates the reference                                                                      */
/* ================================================================================================== */

/* Mes.get_measurements (WindGym/MesClass.py:70-125) on logical history m[0..avail) (0 = oldest).
 * Appends to out[], returns the number of values appended.  Values are rounded to float32 like the
 * reference's `np.array(return_vals, dtype=np.float32)` (:125). */
typedef double (*hist_get_fn)(const void* self, long q);
static int mes_get(const wg_channel* c, int cur_on, int rol_on, long avail, hist_get_fn get, const void* self,
                   float* out) {
    int n = 0;
    if (avail == 0) return 0;                                   /* :75-76 */
    if (cur_on) out[n++] = (float)get(self, avail - 1);         /* :78-80 */
    if (rol_on) {
        long W = c->window_len;
        for (int i = 0; i < c->history_n; ++i) {                /* :85 */
            long lo, hi;
            if (i == 0) {                                       /* :86-91 latest window */
                lo = avail - W; if (lo < 0) lo = 0; hi = avail;
            } else if (i == c->history_n - 1 && avail >= W) {   /* :93-97 oldest window */
                lo = 0; hi = W;
            } else if (avail < W) {                             /* :101-103 */
                lo = 0; hi = avail;
            } else {                                            /* :104-116 middle windows */
                long spacing = (avail - W) / (c->history_n - 1);
                if (spacing < 1) spacing = 1;
                long pos = (long)i * spacing;
                if (pos > avail - W) pos = avail - W;
                lo = pos; hi = pos + W;
            }
            double s = 0;
            for (long q = lo; q < hi; ++q) s += get(self, q);
            out[n++] = (float)(s / (double)(hi - lo));          /* :120 */
        }
    }
    return n;
}

typedef struct ring_view { const double* data; long n_pushed; int hlen; } ring_view;
static double ring_get(const void* self, long q) {
    const ring_view* r = (const ring_view*)self;
    long avail = r->n_pushed < r->hlen ? r->n_pushed : r->hlen;
    long phys = (r->n_pushed - avail + q) % r->hlen;
    return r->data[phys];
}
static long ring_avail(long n_pushed, int hlen) { return n_pushed < hlen ? n_pushed : hlen; }

/* turb_mes._scale_val (MesClass.py:324-326) evaluated like numpy does on a float32 array with Python
 * scalars: every intermediate is float32 */
static float scale_f32(float v, double mn, double mx) {
    float t = v - (float)mn;
    t = 2.0f * t;
    t = t / (float)(mx - mn);
    return t - 1.0f;
}

/* turb_mes.calc_TI (MesClass.py:220-237): population std of the whole ws deque over its mean */
static float calc_ti(const ring_view* r) {
    long avail = ring_avail(r->n_pushed, r->hlen);
    double U = 0;
    for (long q = 0; q < avail; ++q) U += ring_get(r, q);
    U /= (double)avail;
    double m2 = 0, m1 = 0;
    for (long q = 0; q < avail; ++q) m1 += ring_get(r, q) - U;
    m1 /= (double)avail;
    for (long q = 0; q < avail; ++q) {
        double d = ring_get(r, q) - U - m1;
        m2 += d * d;
    }
    return (float)(sqrt(m2 / (double)avail) / U);
}

/* one turb_mes.get_measurements(scaled=True) block (MesClass.py:328-340); `farm_level` selects the
 * farm_mes.farm_mes object (flags & farm_*, power_max*N, its own TI from the farm-level ws deque) */
static int turb_block(const oracle_t* o, const ctx_t* x, int t, int farm_level, int scaled, float* out) {
    const wg_config* c = &o->cfg;
    int n = 0;
    const int N = o->N;
    int on_ws = farm_level ? c->farm_ws : c->turb_ws, on_wd = farm_level ? c->farm_wd : c->turb_wd;
    int on_ti = farm_level ? c->farm_ti : c->turb_ti, on_p = farm_level ? c->farm_power : c->turb_power;
    double pmax = farm_level ? c->power_max * N : c->power_max;
    double mn[WG_N_CH] = {c->ws_scale_min, c->wd_scale_min, c->yaw_min, 0.0};
    double mx[WG_N_CH] = {c->ws_scale_max, c->wd_scale_max, c->yaw_max, pmax};
    int on[WG_N_CH] = {on_ws, on_wd, 1, on_p};
    for (int ch = 0; ch < WG_N_CH; ++ch) {
        if (ch == WG_CH_POWER) {
            /* TI sits between yaw and power (:337) */
            if (on_ti) {
                ring_view r = {farm_level ? x->fring[WG_CH_WS] : x->ring[WG_CH_WS] + (size_t)t * c->ch[WG_CH_WS].history_len,
                               x->n_pushed, c->ch[WG_CH_WS].history_len};
                float ti = calc_ti(&r);
                out[n++] = scaled ? scale_f32(ti, c->ti_scale_min, c->ti_scale_max) : ti;
            }
        }
        ring_view r;
        r.hlen = c->ch[ch].history_len;
        if (farm_level) {
            r.data = x->fring[ch];
            r.n_pushed = (ch == WG_CH_YAW) ? 0 : x->n_pushed;   /* farm yaw is never recorded */
        } else {
            r.data = x->ring[ch] + (size_t)t * r.hlen;
            r.n_pushed = x->n_pushed;
        }
        float tmp[512];
        int m = mes_get(&c->ch[ch], c->ch[ch].current && on[ch], c->ch[ch].rolling_mean && on[ch],
                        ring_avail(r.n_pushed, r.hlen), ring_get, &r, tmp);
        for (int i = 0; i < m; ++i) out[n++] = scaled ? scale_f32(tmp[i], mn[ch], mx[ch]) : tmp[i];
    }
    return n;
}

/* farm_mes.get_measurements(scaled=True) (MesClass.py:679-703) + WindFarmEnv._get_obs clip (:513-520) */
static void get_obs(const oracle_t* o, const env_t* e, double* obs) {
    const wg_config* c = &o->cfg;
    const ctx_t* x = &e->ctx;
    const int N = o->N;
    float buf[4096];
    int n = 0;
    for (int t = 0; t < N; ++t) n += turb_block(o, x, t, 0, 1, buf + n);
    /* farm block: ws, wd, TI (mean of the *scaled* turbine TIs, :670-673), power */
    float tmp[512];
    if (c->farm_ws) {
        ring_view r = {x->fring[WG_CH_WS], x->n_pushed, c->ch[WG_CH_WS].history_len};
        int m = mes_get(&c->ch[WG_CH_WS], c->ch[WG_CH_WS].current, c->ch[WG_CH_WS].rolling_mean,
                        ring_avail(r.n_pushed, r.hlen), ring_get, &r, tmp);
        for (int i = 0; i < m; ++i) buf[n++] = scale_f32(tmp[i], c->ws_scale_min, c->ws_scale_max);
    }
    if (c->farm_wd) {
        ring_view r = {x->fring[WG_CH_WD], x->n_pushed, c->ch[WG_CH_WD].history_len};
        int m = mes_get(&c->ch[WG_CH_WD], c->ch[WG_CH_WD].current, c->ch[WG_CH_WD].rolling_mean,
                        ring_avail(r.n_pushed, r.hlen), ring_get, &r, tmp);
        for (int i = 0; i < m; ++i) buf[n++] = scale_f32(tmp[i], c->wd_scale_min, c->wd_scale_max);
    }
    if (c->farm_ti) {
        /* numpy float32 mean: pairwise summation; for N < 8 a plain loop, else 8 accumulators.  A double
         * accumulator rounded once is within 1 ulp(f32) of it. */
        double s = 0;
        for (int t = 0; t < N; ++t) {
            ring_view r = {x->ring[WG_CH_WS] + (size_t)t * c->ch[WG_CH_WS].history_len, x->n_pushed,
                           c->ch[WG_CH_WS].history_len};
            s += (double)scale_f32(calc_ti(&r), c->ti_scale_min, c->ti_scale_max);
        }
        buf[n++] = (float)(s / N);
    }
    if (c->farm_power) {
        ring_view r = {x->fring[WG_CH_POWER], x->n_pushed, c->ch[WG_CH_POWER].history_len};
        int m = mes_get(&c->ch[WG_CH_POWER], c->ch[WG_CH_POWER].current, c->ch[WG_CH_POWER].rolling_mean,
                        ring_avail(r.n_pushed, r.hlen), ring_get, &r, tmp);
        for (int i = 0; i < m; ++i) buf[n++] = scale_f32(tmp[i], 0.0, c->power_max * N);
    }
    for (int i = 0; i < n; ++i) {
        float v = buf[i];
        if (v < -1.0f) v = -1.0f;
        if (v > 1.0f) v = 1.0f;
        obs[i] = (double)v;
    }
}

/* WindFarmEnvMulti._get_obs_multi (WindEnvMulti.py:79-103) */
static void get_obs_multi(const oracle_t* o, const env_t* e, double* obs) {
    const ctx_t* x = &e->ctx;
    float buf[2048];
    for (int t = 0; t < o->N; ++t) {
        int n = turb_block(o, x, t, 0, 1, buf);
        n += turb_block(o, x, 0, 1, 1, buf + n);
        for (int i = 0; i < n; ++i) {
            float v = buf[i];
            if (v < -1.0f) v = -1.0f;
            if (v > 1.0f) v = 1.0f;
            obs[(size_t)t * o->obs_dim_multi + i] = (double)v;
        }
    }
}

/* farm_mes.add_measurements (MesClass.py:568-591) */
static void add_measurements(const oracle_t* o, env_t* e, double* ws, double* wd, double* yaw, double* pw) {
    const wg_config* c = &o->cfg;
    ctx_t* x = &e->ctx;
    const int N = o->N;
    double* arr[WG_N_CH] = {ws, wd, yaw, pw};
    if (c->noise == WG_NOISE_NORMAL) {
        for (int ch = 0; ch < WG_N_CH; ++ch) {
            if (c->noise_sigma[ch] == 0.0) continue;
            for (int t = 0; t < N; ++t)
                arr[ch][t] += c->noise_sigma[ch] *
                              wgo_noise_normal(e->noise_key, (uint32_t)x->n_pushed, (uint32_t)t, (uint32_t)ch,
                                               (uint32_t)e->episode);
        }
    }
    for (int ch = 0; ch < WG_N_CH; ++ch) {
        int H = c->ch[ch].history_len;
        long slot = x->n_pushed % H;
        for (int t = 0; t < N; ++t) x->ring[ch][(size_t)t * H + slot] = arr[ch][t];
    }
    /* farm level: mean ws, mean wd, SUM power — always recorded (:589-591) */
    double sws = 0, swd = 0, sp = 0;
    for (int t = 0; t < N; ++t) { sws += ws[t]; swd += wd[t]; sp += pw[t]; }
    x->fring[WG_CH_WS][x->n_pushed % c->ch[WG_CH_WS].history_len] = sws / N;
    x->fring[WG_CH_WD][x->n_pushed % c->ch[WG_CH_WD].history_len] = swd / N;
    x->fring[WG_CH_POWER][x->n_pushed % c->ch[WG_CH_POWER].history_len] = sp;
    x->n_pushed++;
}

/* WindFarmEnv._take_measurements (Wind_Farm_Env.py:480-495) accumulated into sums */
static void take_measurements(const oracle_t* o, env_t* e, double* sws, double* swd, double* syaw, double* sp) {
    ctx_t* x = &e->ctx;
    farm_t* f = &x->farm[0];
    for (int t = 0; t < o->N; ++t) {
        double u = f->u[t], v = f->v[t], w = f->w[t];
        double ws = sqrt(u * u + v * v + w * w);            /* :485-487 */
        double wd = atan(v / u) * (180.0 / PI_D) + x->wd;   /* :489-492 (arctan, not arctan2) */
        x->cur_ws[t] = ws; x->cur_wd[t] = wd;
        sws[t] += ws; swd[t] += wd; syaw[t] += f->yaw[t]; sp[t] += f->power[t];
    }
}

/* BasicControllers.local_yaw_controller / global_yaw_controller (BasicControllers.py:10-73); not clipped */
static void base_controller(const oracle_t* o, farm_t* f) {
    const wg_config* c = &o->cfg;
    for (int t = 0; t < o->N; ++t) {
        double yaw = f->yaw[t];
        if (c->base_controller == WG_CTRL_LOCAL) {
            double wdir = atan((double)f->v[t] / (double)f->u[t]) * (180.0 / PI_D);
            double off = wdir - yaw;
            double sgn = (off > 0) - (off < 0);
            double sc = fabs(off); if (sc > c->yaw_step) sc = c->yaw_step;
            yaw = yaw + sgn * sc;
        } else {
            double sgn = (yaw > 0) - (yaw < 0);
            double sc = fabs(yaw); if (sc > c->yaw_step) sc = c->yaw_step;
            yaw = yaw - sgn * sc;
        }
        f->yaw[t] = (real)yaw;
    }
}

static double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* WindFarmEnv._adjust_yaws (Wind_Farm_Env.py:822-864).  The action arrives as a float32 array (the action
 * space dtype, :461-463), so numpy evaluates `action * yaw_step` and the "wind" target in float32 before
 * they meet the float64 yaw array; restated with the same rounding points. */
static void adjust_yaws(const oracle_t* o, farm_t* f, const double* action) {
    const wg_config* c = &o->cfg;
    for (int t = 0; t < o->N; ++t) {
        double yaw = f->yaw[t];
        float af = (float)action[t];
        if (c->action_method == WG_ACT_YAW) {
            float prod = af * (float)c->yaw_step;
            yaw = clipd(yaw + (double)prod, c->yaw_min, c->yaw_max);                      /* :832-836 */
        } else {
            float tf = af + 1.0f;
            tf = tf / 2.0f;
            tf = tf * (float)(c->yaw_max - c->yaw_min);
            tf = tf + (float)c->yaw_min;                                                   /* :841-843 */
            double ny = clipd((double)tf, yaw - c->yaw_step, yaw + c->yaw_step);          /* :849-855 */
            yaw = clipd(ny, c->yaw_min, c->yaw_max);
        }
        f->yaw[t] = (real)yaw;
    }
}

static double deque_mean(const double* d, long n_total, int maxlen) {
    long n = n_total < maxlen ? n_total : maxlen;
    double s = 0;
    for (long i = 0; i < n; ++i) s += d[i];
    return s / (double)n; /* n == 0 -> NaN like np.mean([]) */
}
/* logical element q (0 = oldest) of a deque stored as a ring */
static double deque_at(const double* d, long n_total, int maxlen, long q) {
    long n = n_total < maxlen ? n_total : maxlen;
    return d[(n_total - n + q) % maxlen];
}
static void deque_push(double* d, long* n_total, int maxlen, double v) {
    d[*n_total % maxlen] = v;
    (*n_total)++;
}

/* rewards (Wind_Farm_Env.py:866-918) and _action_penalty (:804-820) */
static double compute_reward(const oracle_t* o, env_t* e) {
    const wg_config* c = &o->cfg;
    farm_t* f = &e->ctx.farm[0];
    double pr = 0;
    switch (c->reward_mode) {
    case WG_REW_BASELINE:
        pr = deque_mean(e->farm_pow, e->farm_pow_n, c->power_avg) /
             deque_mean(e->base_pow, e->base_pow_n, c->power_avg) - 1.0;                 /* :882-891 */
        break;
    case WG_REW_POWER_AVG:
        pr = deque_mean(e->farm_pow, e->farm_pow_n, c->power_avg) / o->N / e->ctx.rated_power; /* :896-897 */
        break;
    case WG_REW_NONE: pr = 0.0; break;
    case WG_REW_POWER_DIFF: {                                                            /* :904-918 */
        int wsz = c->power_avg / 10;
        long n = e->farm_pow_n < c->power_avg ? e->farm_pow_n : c->power_avg;
        double sl = 0, so = 0; long nl = 0, no = 0;
        for (long q = c->power_avg - wsz; q < c->power_avg && q < n; ++q) { sl += deque_at(e->farm_pow, e->farm_pow_n, c->power_avg, q); nl++; }
        for (long q = 0; q < wsz && q < n; ++q) { so += deque_at(e->farm_pow, e->farm_pow_n, c->power_avg, q); no++; }
        pr = (sl / (double)nl - so / (double)no) / o->N;
        break;
    }
    }
    double pen = 0;
    if (c->action_penalty >= 0.001) {                                                    /* :808-811 */
        double s = 0;
        if (c->penalty_type == WG_PEN_CHANGE) {
            for (int t = 0; t < o->N; ++t) s += fabs(e->old_yaws[t] - (double)f->yaw[t]);
            pen = c->action_penalty * (s / o->N);                                        /* :815 */
        } else {
            for (int t = 0; t < o->N; ++t) s += fabs((double)f->yaw[t]);
            pen = c->action_penalty * (s / o->N / c->yaw_max);                           /* :818 */
        }
    }
    return pr * c->power_scaling + 0.0 - pen;                                            /* :989-996 */
}

static void farm_init(const oracle_t* o, ctx_t* x, farm_t* f, const real* yaw0) {
    f->head = o->P - 1; f->n_valid = 0; f->s_off = 0.0; f->time = 0.0; f->istep = 0;
    for (int t = 0; t < o->N; ++t) {
        f->yaw[t] = yaw0[t];
        f->u[t] = (real)x->ws; f->v[t] = 0; f->w[t] = 0;
        f->ti_loc[t] = (real)x->ti;
        f->power[t] = 0; f->ct[t] = 0;
    }
}

/* WindFarmEnv.reset (Wind_Farm_Env.py:680-802) for one env */
static void reset_env(oracle_t* o, int b, uint64_t seed, int reseed) {
    const wg_config* c = &o->cfg;
    env_t* e = &o->env[b];
    ctx_t* x = &e->ctx;
    const int N = o->N, K = o->K;
    if (reseed) { wgo_pcg64_seed(&e->rng, seed); e->noise_key = seed; }   /* :689 */
    e->timestep = 0; e->done = 0;                                          /* :690 */
    /* _set_windconditions (:557-568): ws, ti, wd ~ U[min,max] in this order */
    x->ws = wgo_pcg64_uniform(&e->rng, c->ws_min, c->ws_max);
    x->ti = wgo_pcg64_uniform(&e->rng, c->ti_min, c->ti_max);
    x->wd = wgo_pcg64_uniform(&e->rng, c->wd_min, c->wd_max);
    /* _def_site (:598-678): "Random" draws a turbulence seed; "None"/"MannFixed" draw nothing */
    x->turb_seed = 0; x->box_ox = 0; x->box_oy = 0; x->box_id = 0;
    if (c->turb_mode == WG_TURB_BOX_POOL)       /* tf_file = self.np_random.choice(self.TF_files) (:614) */
    {
        /* FarmEval.update_tf(path): TF_files = [path] -> np_random.choice of ONE file consumes no random number */
        const int fixed = o->box_ids ? o->box_ids[b] : -1;
        if (fixed >= 0) x->box_id = fixed < o->n_boxes ? fixed : 0;
        else x->box_id = (int)wgo_pcg64_integers(&e->rng, (uint32_t)(o->n_boxes > 0 ? o->n_boxes : 1));
    }
    if (c->turb_mode == WG_TURB_RANDOM || c->turb_mode == WG_TURB_BOX_SHIFT)
        x->turb_seed = wgo_pcg64_integers(&e->rng, 100000);
    if (c->turb_mode == WG_TURB_BOX_SHIFT && o->box) {
        /* the episode's seed picks one of the pool's generated realisations and an offset into it (:623-637) */
        if (o->n_boxes > 1) x->box_id = (int)(x->turb_seed % (uint32_t)o->n_boxes);
        double fx, fy;
        wgo_turb_offset(x->turb_seed, &fx, &fy);
        x->box_ox = fx * o->bnx * o->bdx;
        x->box_oy = fy * o->bny * o->bdy;
    }
    /* flow frame (model M0): rotate the layout by theta = 270 - wd about the farm centre */
    double th = (270.0 - x->wd) * (PI_D / 180.0), cx = 0, cy = 0;
    for (int t = 0; t < N; ++t) { cx += o->x_pos[t]; cy += o->y_pos[t]; }
    cx /= N; cy /= N;
    double xmin = 1e300, xmax = -1e300;
    for (int t = 0; t < N; ++t) {
        double dx = o->x_pos[t] - cx, dy = o->y_pos[t] - cy;
        x->xr[t] = cx + dx * cos(th) + dy * sin(th);
        x->yr[t] = cy - dx * sin(th) + dy * cos(th);
        if (x->xr[t] < xmin) xmin = x->xr[t];
        if (x->xr[t] > xmax) xmax = x->xr[t];
    }
    for (int ch = 0; ch < WG_N_CH; ++ch) {  /* _init_farm_mes (:697): fresh sensor state */
        memset(x->ring[ch], 0, sizeof(double) * (size_t)N * c->ch[ch].history_len);
        memset(x->fring[ch], 0, sizeof(double) * (size_t)c->ch[ch].history_len);
    }
    x->n_pushed = 0;
    x->rated_power = tab_interp(o->tab_ws, o->tab_power, c->n_tab, x->ws);   /* :700 */
    /* yaw init (:715-720; WindEnv.py:28-56) */
    real yaw0[N];
    for (int t = 0; t < N; ++t) {
        if (c->yaw_init == WG_YAWINIT_RANDOM) yaw0[t] = (real)wgo_pcg64_uniform(&e->rng, -c->yaw_start, c->yaw_start);
        else if (c->yaw_init == WG_YAWINIT_DEFINED && o->yaw_defined) yaw0[t] = (real)o->yaw_defined[t];
        else yaw0[t] = 0;
    }
    x->dist = xmax - xmin;                                  /* :723-724 */
    x->t_inflow = x->dist / x->ws;                          /* :727 */
    x->t_developed = (int)(x->t_inflow * 2);                /* :729 */
    x->time_max = c->never_truncate ? 9999999 : (int)(x->t_inflow * c->n_passthrough); /* :732 */
    int n_dev = (int)ceil((double)x->t_developed / c->dt_sim - 1e-9);
    if (o->script_uvw) n_dev = 0;                           /* the flow double's run() consumes no rows */
    double sws[N], swd[N], syaw[N], sp[N];
    for (int fi = 0; fi < o->F; ++fi) {
        farm_t* f = &x->farm[fi];
        long keep_cursor = f->cursor;
        farm_init(o, x, f, fi == 0 ? yaw0 : x->farm[0].yaw);   /* :781 baseline starts from the agent's yaws */
        f->cursor = keep_cursor;
        if (o->script_uvw) script_load(o, e, b, fi);
        for (int i = 0; i < n_dev; ++i) m0_flow_step(o, e, fi);   /* fs.run(t_developed) (:734, :782) */
        if (fi == 0) {
            for (int s = 0; s < c->fill_steps_agent; ++s) {       /* :737-766 */
                for (int t = 0; t < N; ++t) sws[t] = swd[t] = syaw[t] = sp[t] = 0;
                for (int k = 0; k < K; ++k) { flow_step(o, e, b, 0); take_measurements(o, e, sws, swd, syaw, sp); }
                for (int t = 0; t < N; ++t) { sws[t] /= K; swd[t] /= K; syaw[t] /= K; sp[t] /= K; }
                add_measurements(o, e, sws, swd, syaw, sp);
                double tot = 0; for (int t = 0; t < N; ++t) tot += sp[t];
                deque_push(e->farm_pow, &e->farm_pow_n, c->power_avg, tot);
            }
        } else {
            for (int s = 0; s < c->fill_steps_base; ++s) {        /* :784-796, no controller during the fill */
                double acc = 0;
                for (int k = 0; k < K; ++k) {
                    flow_step(o, e, b, 1);
                    double tot = 0; for (int t = 0; t < N; ++t) tot += (double)f->power[t];
                    acc += tot;
                }
                deque_push(e->base_pow, &e->base_pow_n, c->power_avg, acc / K);
            }
        }
    }
}

int WGO(reset)(void* h, const uint8_t* mask, const uint64_t* seeds, double* obs) {
    oracle_t* o = (oracle_t*)h;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < o->B; ++b) {
        if (mask && !mask[b]) continue;
        int reseed = seeds && seeds[b] != UINT64_MAX;
        /* an explicit reset that abandons a running episode starts a new episode index: the sensor-noise stream is
         * keyed by (seed, episode index, push index) and must not replay the abandoned episode's sequence */
        if (!reseed && !o->env[b].done && o->env[b].timestep > 0) o->env[b].episode++;
        reset_env(o, b, reseed ? seeds[b] : 0, reseed);
        o->env[b].ep_return = 0; o->env[b].ep_power_sum = 0; o->env[b].ep_len = 0;
    }
    if (obs)
        for (int b = 0; b < o->B; ++b)
            if (!mask || mask[b]) get_obs(o, &o->env[b], obs + (size_t)b * o->obs_dim);
    return 0;
}

/* WindFarmEnv.step (Wind_Farm_Env.py:920-1034) for one env */
static void step_env(oracle_t* o, int b, const double* action, double* obs, double* reward, uint8_t* trunc,
                     double* final_obs) {
    const wg_config* c = &o->cfg;
    env_t* e = &o->env[b];
    ctx_t* x = &e->ctx;
    const int N = o->N, K = o->K;
    farm_t* fa = &x->farm[0];
    for (int t = 0; t < N; ++t) e->old_yaws[t] = fa->yaw[t];            /* :932 */
    adjust_yaws(o, fa, action);                                          /* :934 */
    double sws[N], swd[N], syaw[N], sp[N], base_acc = 0;
    for (int t = 0; t < N; ++t) sws[t] = swd[t] = syaw[t] = sp[t] = 0;
    for (int k = 0; k < K; ++k) {                                        /* :943 */
        flow_step(o, e, b, 0);                                           /* :945 */
        if (o->F == 2) {                                                 /* :948-954 */
            farm_t* fb = &x->farm[1];
            base_controller(o, fb);
            flow_step(o, e, b, 1);
            double tot = 0; for (int t = 0; t < N; ++t) tot += (double)fb->power[t];
            base_acc += tot;
        }
        take_measurements(o, e, sws, swd, syaw, sp);                     /* :957-963 */
    }
    for (int t = 0; t < N; ++t) { sws[t] /= K; swd[t] /= K; syaw[t] /= K; sp[t] /= K; }  /* :965-969 */
    add_measurements(o, e, sws, swd, syaw, sp);                          /* :972 */
    double tot = 0; for (int t = 0; t < N; ++t) tot += sp[t];
    deque_push(e->farm_pow, &e->farm_pow_n, c->power_avg, tot);          /* :975 */
    if (o->F == 2) deque_push(e->base_pow, &e->base_pow_n, c->power_avg, base_acc / K);   /* :979 */
    if (isnan(tot)) e->nan_power = 1;                                    /* :980-981 */
    get_obs(o, e, obs);                                                  /* :983 */
    double r = compute_reward(o, e);                                     /* :989-996 */
    int truncated = e->timestep >= x->time_max;                          /* :1003 */
    e->timestep += 1 + (c->extra_timestep_inc ? 1 : 0);                  /* :1027 (+ WindEnvMulti.py:219) */
    *reward = r;
    *trunc = (uint8_t)truncated;
    /* RecordEpisodeVals / monitor accumulators; "Power agent" = current farm power (:539) */
    double pnow = 0; for (int t = 0; t < N; ++t) pnow += (double)fa->power[t];
    double pbase = 0;
    if (o->F == 2) for (int t = 0; t < N; ++t) pbase += (double)x->farm[1].power[t];
    e->last_pnow = pnow; e->last_pbase = pbase;
    e->ep_return += r; e->ep_power_sum += pnow; e->ep_len += 1;
    e->ep_finished = 0;
    if (final_obs) memcpy(final_obs, obs, sizeof(double) * (size_t)o->obs_dim);
    if (truncated) {
        e->ep_finished = 1;
        e->fin_return = e->ep_return; e->fin_len = e->ep_len;
        e->fin_mean_power = e->ep_power_sum / (double)e->ep_len;   /* recordEpisodeVals.py:51-56 */
        e->ep_return = 0; e->ep_power_sum = 0; e->ep_len = 0;
        e->episode++;
        if (c->autoreset) {
            reset_env(o, b, 0, 0);
            get_obs(o, e, obs);
        } else {
            e->done = 1;
        }
    }
}

int WGO(step)(void* h, const double* actions, double* obs, double* reward, uint8_t* trunc, double* final_obs) {
    oracle_t* o = (oracle_t*)h;
    for (int b = 0; b < o->B; ++b)
        if (o->env[b].done) return WG_ERR_STATE;
    double met[WG_N_METRICS] = {0};
#pragma omp parallel for schedule(static)
    for (int b = 0; b < o->B; ++b)
        step_env(o, b, actions + (size_t)b * o->N, obs + (size_t)b * o->obs_dim, reward + b, trunc + b,
                 final_obs ? final_obs + (size_t)b * o->obs_dim : NULL);
    /* metric partial sums (sequential: deterministic order) */
    for (int b = 0; b < o->B; ++b) {
        env_t* e = &o->env[b];
        met[WG_MET_STEP_REWARD_SUM] += reward[b];
        met[WG_MET_FARM_POWER_SUM] += e->last_pnow;
        met[WG_MET_BASE_POWER_SUM] += e->last_pbase;
        met[WG_MET_N_STEPS] += 1;
        if (e->ep_finished) {
            met[WG_MET_EP_RETURN_SUM] += e->fin_return;
            met[WG_MET_EP_LENGTH_SUM] += (double)e->fin_len;
            met[WG_MET_EP_MEAN_POWER_SUM] += e->fin_mean_power;
            met[WG_MET_N_EPISODES] += 1;
        }
    }
    for (int i = 0; i < WG_N_METRICS; ++i) o->metrics[i] += met[i];
    for (int b = 0; b < o->B; ++b)
        if (o->env[b].nan_power) return WG_ERR_NAN_POWER;
    return 0;
}

void WGO(metrics)(void* h, double* out, int reset_after) {
    oracle_t* o = (oracle_t*)h;
    for (int i = 0; i < WG_N_METRICS; ++i) out[i] = o->metrics[i];
    if (reset_after) memset(o->metrics, 0, sizeof(o->metrics));
}

void WGO(obs_multi)(void* h, double* obs) {
    oracle_t* o = (oracle_t*)h;
    for (int b = 0; b < o->B; ++b) get_obs_multi(o, &o->env[b], obs + (size_t)b * o->N * o->obs_dim_multi);
}

void WGO(get_obs)(void* h, double* obs) {
    oracle_t* o = (oracle_t*)h;
    for (int b = 0; b < o->B; ++b) get_obs(o, &o->env[b], obs + (size_t)b * o->obs_dim);
}

/* info dict fields as doubles (shape per wg_info_field) */
int WGO(get_info)(void* h, int field, double* out) {
    oracle_t* o = (oracle_t*)h;
    const int N = o->N;
    for (int b = 0; b < o->B; ++b) {
        env_t* e = &o->env[b];
        ctx_t* x = &e->ctx;
        farm_t* fa = &x->farm[0];
        farm_t* fb = &x->farm[o->F == 2 ? 1 : 0];
        switch (field) {
        case WG_INFO_YAW_AGENT: for (int t = 0; t < N; ++t) out[b * N + t] = fa->yaw[t]; break;
        case WG_INFO_YAW_BASE: for (int t = 0; t < N; ++t) out[b * N + t] = fb->yaw[t]; break;
        case WG_INFO_WS_GLOBAL: out[b] = x->ws; break;
        case WG_INFO_WD_GLOBAL: out[b] = x->wd; break;
        case WG_INFO_TI_GLOBAL: out[b] = x->ti; break;
        case WG_INFO_WS_TURB: for (int t = 0; t < N; ++t) out[b * N + t] = x->cur_ws[t]; break;
        case WG_INFO_WD_TURB: for (int t = 0; t < N; ++t) out[b * N + t] = x->cur_wd[t]; break;
        case WG_INFO_POWER_TURB_AGENT: for (int t = 0; t < N; ++t) out[b * N + t] = fa->power[t]; break;
        case WG_INFO_POWER_TURB_BASE: for (int t = 0; t < N; ++t) out[b * N + t] = fb->power[t]; break;
        case WG_INFO_POWER_AGENT: { double s = 0; for (int t = 0; t < N; ++t) s += fa->power[t]; out[b] = s; break; }
        case WG_INFO_POWER_BASE: { double s = 0; for (int t = 0; t < N; ++t) s += fb->power[t]; out[b] = s; break; }
        case WG_INFO_WS_TURB_BASE: for (int t = 0; t < N; ++t) out[b * N + t] = fb->u[t]; break;
        case WG_INFO_TURB_X: for (int t = 0; t < N; ++t) out[b * N + t] = x->xr[t]; break;
        case WG_INFO_TURB_Y: for (int t = 0; t < N; ++t) out[b * N + t] = x->yr[t]; break;
        case WG_INFO_TIMESTEP: out[b] = e->timestep; break;
        case WG_INFO_TIME_MAX: out[b] = x->time_max; break;
        case WG_INFO_FS_TIME: out[b] = fa->time; break;
        case WG_INFO_EPISODE: out[b] = e->episode; break;
        case WG_INFO_ROTOR_UVW_AGENT:
            for (int t = 0; t < N; ++t) { out[(b * N + t) * 3] = fa->u[t]; out[(b * N + t) * 3 + 1] = fa->v[t]; out[(b * N + t) * 3 + 2] = fa->w[t]; }
            break;
        case WG_INFO_ROTOR_UVW_BASE:
            for (int t = 0; t < N; ++t) { out[(b * N + t) * 3] = fb->u[t]; out[(b * N + t) * 3 + 1] = fb->v[t]; out[(b * N + t) * 3 + 2] = fb->w[t]; }
            break;
        case WG_INFO_RATED_POWER: out[b] = x->rated_power; break;
        case WG_INFO_BOX_ID: out[b] = x->box_id; break;
        case WG_INFO_STEP_POWER_AGENT: out[b] = e->last_pnow; break;
        case WG_INFO_STEP_POWER_BASE: out[b] = e->last_pbase; break;
        default: return WG_ERR_INVALID;
        }
    }
    return 0;
}

/* Flow-field view: (u, v, w) on an XY grid at height z in the flow frame — fs.get_windspeed(XYView(z, x, y),
 * include_wakes) behind WindFarmEnv._render_frame / init_render (Wind_Farm_Env.py:1040-1083, :464-476).  Model M0
 * at a point: ambient (+ frozen-box fluctuation; the "Random" inflow has no spatial field, so it adds nothing here)
 * minus the Gaussian deficits of all chains passing upstream of the point, wake centre / record interpolated between
 * the two bracketing particles exactly like the rotor inflow (no lateral cut-off).  out[3][nx][ny]. */
int WGO(get_windspeed)(void* h, int b, int fi, const double* xs, int nx, const double* ys, int ny, double z,
                       int include_wakes, double* out) {
    oracle_t* o = (oracle_t*)h;
    if (b < 0 || b >= o->B || fi < 0 || fi >= o->F) return WG_ERR_INVALID;
    const wg_config* c = &o->cfg;
    const ctx_t* x = &o->env[b].ctx;
    const farm_t* f = &x->farm[fi];
    const int N = o->N, P = o->P;
    const real D = (real)c->rotor_diameter, inv_D = (real)1 / D;
    const double dpart = c->d_particle * c->rotor_diameter, inv_dpart = 1.0 / dpart;
    const int box = has_box(o);
    const size_t plane = (size_t)nx * ny;
    for (int ix = 0; ix < nx; ++ix)
        for (int iy = 0; iy < ny; ++iy) {
            const double px = xs[ix], py = ys[iy];
            real amb[3] = {0, 0, 0};
            if (box)
                for (int cc = 0; cc < 3; ++cc) amb[cc] = box_fluct(o, x, cc, f->time, px, py, z);
            real dsum = 0;
            for (int s2 = 0; include_wakes && s2 < N; ++s2) {
                double dx = px - x->xr[s2];
                if (!(dx > 0.0)) continue;
                double xi = (dx - f->s_off) * inv_dpart;
                double jf = floor(xi);
                real wgt = (real)(xi - jf);
                long j = (long)jf;
                if (j < 0) { j = 0; wgt = 0; }
                if (j + 1 > f->n_valid - 1) continue;
                int r0 = f->head - (int)j; r0 %= P; if (r0 < 0) r0 += P;
                int r1 = r0 - 1; if (r1 < 0) r1 += P;
                size_t i0 = (size_t)s2 * P + r0, i1 = (size_t)s2 * P + r1;
                real w0 = (real)1 - wgt, w1 = wgt;
                real yc = w0 * f->py[i0] + w1 * f->py[i1];
                real zc = w0 * f->pz[i0] + w1 * f->pz[i1];
                real ctv = w0 * f->ct_e[i0] + w1 * f->ct_e[i1];
                real kv = w0 * f->k_e[i0] + w1 * f->k_e[i1];
                real epv = w0 * f->eps_e[i0] + w1 * f->eps_e[i1];
                real uev = w0 * f->u_e[i0] + w1 * f->u_e[i1];
                real sp = m0_sigma_over_d(kv, epv, (real)dx * inv_D);
                real sig = sp * D;
                real inv2s2 = (real)1 / ((real)2 * sig * sig);
                real r2 = ((real)py - yc) * ((real)py - yc) + ((real)z - zc) * ((real)z - zc);
                if (c->deficit_model == 2 && o->dtab) {
                    an_wake aw; real slope;
                    an_setup(o, ctv, kv, (real)dx * inv_D, &aw);
                    dsum += uev * an_frac(o, &aw, R_SQRT(r2) / ((real)0.5 * D), &slope);
                } else if (c->deficit_model == 1) {
                    real xd = (real)dx * inv_D;
                    real nsg = (real)c->m0_sg_af * R_EXP((real)c->m0_sg_bf * xd) + (real)c->m0_sg_cf;
                    real rn = r2 > (real)0 ? R_POW(r2 * inv_D * inv_D, (real)0.5 * nsg) : (real)0;
                    dsum += uev * m0_sg_cfrac(ctv, sp, nsg) * R_EXP(-rn / ((real)2 * sp * sp));
                } else {
                    dsum += uev * m0_cfrac(ctv, sp) * R_EXP(-r2 * inv2s2);
                }
            }
            out[0 * plane + (size_t)ix * ny + iy] = (real)x->ws + amb[0] - dsum;
            out[1 * plane + (size_t)ix * ny + iy] = amb[1];
            out[2 * plane + (size_t)ix * ny + iy] = amb[2];
        }
    return 0;
}

/* debug access to the particle chains of one farm: py, u_e, ct_e by age index (0 = newest) */
int WGO(get_chain)(void* h, int b, int fi, int t, double* py, double* ue, double* ct, double* s_off) {
    oracle_t* o = (oracle_t*)h;
    farm_t* f = &o->env[b].ctx.farm[fi];
    for (int j = 0; j < f->n_valid; ++j) {
        int r = f->head - j; r %= o->P; if (r < 0) r += o->P;
        size_t i = (size_t)t * o->P + r;
        py[j] = f->py[i]; ue[j] = f->u_e[i]; ct[j] = f->ct_e[i];
    }
    *s_off = f->s_off;
    return f->n_valid;
}

void WGO(set_threads)(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int WGO(max_threads)(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
