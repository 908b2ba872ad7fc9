// wg_kernels.hip — HIP kernels of the batched WindGym step() transition for MI355X (gfx950, wave64).
//
//   k_flow   one 256-thread workgroup per farm slot (env x ctx x farm).  Streams the slot's wake-particle
//            SoA through HBM with 16-byte coalesced accesses (advection + emission), then the 4 waves
//            evaluate the Gaussian deficit superposition at the rotor sample points with the per-pair wake
//            parameters staged in LDS, look up power/Ct, and push the sensor rings.
//            Replaces DWMFlowSimulation.step() + rotor_avg_windspeed + power() + the baseline controllers
//            + _take_measurements + farm_mes.add_measurements
//            (Wind_Farm_Env.py:480-495, 822-864, 943-979; BasicControllers.py:10-73; MesClass.py:568-591).
//   k_glue   one wave per env: power deques, observation (MesClass windows / TI / scaling / clip), reward,
//            penalty, truncation, same-step autoreset by swapping in the pre-developed next episode, and
//            sampling of the episode after that (Wind_Farm_Env.py:513-520, 557-568, 680-732, 804-820,
//            878-918, 980-1034; MesClass.py:70-125, 220-237, 324-351, 679-703).
//   k_init, k_info, k_obs_multi, k_metrics: reset bookkeeping, lazy info dict, PettingZoo packing
//            (WindEnvMulti.py:79-103) and the episode-metric partial sums (recordEpisodeVals.py:31-64).
#include <hip/hip_runtime.h>

#include "wg_device.h"

// ===================================================================================================
// model M0 pieces
// ===================================================================================================
__device__ __forceinline__ float m0_cfrac(float ct, float sp) {
    float m = __builtin_amdgcn_rcpf(8.0f * sp * sp);
    m = fminf(m, 1.0f);
    float a = fmaxf(1.0f - ct * m, 0.0f);
    return 1.0f - __builtin_amdgcn_sqrtf(a);
}

struct FlowLds {
    double *xr, *yr;
    float *yaw, *u, *v, *w, *ti, *pow, *ct, *cg;
    float *rct, *rk, *reps, *rhv, *rue;
    float *sws, *swd, *syaw, *sp;
    float *tabws, *tabp, *tabct;
    float *rdy, *rdz;
    float4* pair;
};

__host__ __device__ inline size_t flow_lds_bytes(int N, int S, int n_tab) {
    size_t b = 0;
    b += sizeof(double) * 2 * N;
    b += sizeof(float) * (8 + 5 + 4) * N;
    b += sizeof(float) * 3 * n_tab;
    b += sizeof(float) * 2 * S;
    b = (b + 15) & ~(size_t)15;
    b += sizeof(float4) * WG_NWAVES * N;
    return b;
}

__device__ inline FlowLds flow_lds_carve(char* smem, int N, int S, int n_tab) {
    FlowLds L;
    L.xr = (double*)smem;
    L.yr = L.xr + N;
    float* f = (float*)(L.yr + N);
    L.yaw = f; f += N; L.u = f; f += N; L.v = f; f += N; L.w = f; f += N;
    L.ti = f; f += N; L.pow = f; f += N; L.ct = f; f += N; L.cg = f; f += N;
    L.rct = f; f += N; L.rk = f; f += N; L.reps = f; f += N; L.rhv = f; f += N; L.rue = f; f += N;
    L.sws = f; f += N; L.swd = f; f += N; L.syaw = f; f += N; L.sp = f; f += N;
    L.tabws = f; f += n_tab; L.tabp = f; f += n_tab; L.tabct = f; f += n_tab;
    L.rdy = f; f += S; L.rdz = f; f += S;
    size_t off = ((char*)f - smem + 15) & ~(size_t)15;
    L.pair = (float4*)(smem + off);
    return L;
}

// One DWMFlowSimulation.step() of model M0 for the slot owned by this workgroup.  Slot scalars live in
// registers (uniform across the block); turbine state lives in LDS.
template <int TURB>
__device__ inline void flow_step(const WgParams& p, const WgPtrs& d, const FlowLds& L, size_t pbase, double ws,
                                 double ti_amb, int& head, int& n_valid, double& s_off, double& time) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N, P = p.P, S = p.S, NP = p.NP;

    // (1) emission records of this step + cos(yaw)
    for (int t = tid; t < N; t += WG_BLOCK) {
        float g = L.yaw[t] * WG_DEG2RAD_F;
        const float sg = sinf(g), cg = cosf(g);
        float wsn = fmaxf(L.u[t] * cg + L.v[t] * sg, 0.0f);
        float ct0 = wg_tab_interp<float>(L.tabws, L.tabct, p.n_tab, wsn);
        float ctx = fminf(fmaxf(ct0 * cg * cg, 0.0f), 0.96f);
        float q = sqrtf(1.0f - ctx);
        float beta = 0.5f * (1.0f + q) / q;
        L.rct[t] = ctx;
        L.rk[t] = p.ka * L.ti[t] + p.kb;
        L.reps[t] = p.eps0 * sqrtf(beta);
        L.rhv[t] = -p.hill * sg * L.u[t];
        L.rue[t] = L.u[t];
        L.cg[t] = cg;
    }
    __syncthreads();

    // (2)+(3) advect every particle over dt, release the new particles (streaming pass over the SoA)
    double s_new = s_off + ws * p.dt_d;
    int n_emit = 0;
    while (s_new >= p.dpart) { s_new -= p.dpart; ++n_emit; }
    if (n_emit > P) n_emit = P;
    int new_head = (head + n_emit) % P;
    int new_valid = n_valid + n_emit; if (new_valid > P) new_valid = P;

    float* __restrict__ gpy = d.py + pbase;
    float* __restrict__ gct = d.ct_e + pbase;
    float* __restrict__ gk = d.k_e + pbase;
    float* __restrict__ geps = d.eps_e + pbase;
    float* __restrict__ ghv = d.hv_e + pbase;
    float* __restrict__ gue = d.u_e + pbase;
    for (int i4 = tid * 4; i4 < NP; i4 += WG_BLOCK * 4) {
        const int t = i4 / P;
        const int r0 = i4 - t * P;
        float4 py4 = *reinterpret_cast<const float4*>(gpy + i4);
        float4 ct4 = *reinterpret_cast<const float4*>(gct + i4);
        float4 k4 = *reinterpret_cast<const float4*>(gk + i4);
        float4 ep4 = *reinterpret_cast<const float4*>(geps + i4);
        float4 hv4 = *reinterpret_cast<const float4*>(ghv + i4);
        float pyv[4] = {py4.x, py4.y, py4.z, py4.w};
        float ctv[4] = {ct4.x, ct4.y, ct4.z, ct4.w};
        float kv[4] = {k4.x, k4.y, k4.z, k4.w};
        float epv[4] = {ep4.x, ep4.y, ep4.z, ep4.w};
        float hvv[4] = {hv4.x, hv4.y, hv4.z, hv4.w};
        bool any_emit = false;
        bool em[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + q;
            int ei = r - head - 1; if (ei < 0) ei += P;
            em[q] = ei < n_emit;
            any_emit |= em[q];
            if (em[q]) {
                pyv[q] = (float)L.yr[t];
                ctv[q] = L.rct[t]; kv[q] = L.rk[t]; epv[q] = L.reps[t]; hvv[q] = L.rhv[t];
            } else {
                int j = head - r; if (j < 0) j += P;
                if (j < n_valid) {
                    float xrel = (float)(s_off + (double)j * p.dpart);
                    float sp = kv[q] * (xrel * p.inv_D) + epv[q];
                    float cf = m0_cfrac(ctv[q], sp);
                    pyv[q] += hvv[q] * cf * p.dt;
                }
            }
        }
        *reinterpret_cast<float4*>(gpy + i4) = make_float4(pyv[0], pyv[1], pyv[2], pyv[3]);
        if (any_emit) {
            *reinterpret_cast<float4*>(gct + i4) = make_float4(ctv[0], ctv[1], ctv[2], ctv[3]);
            *reinterpret_cast<float4*>(gk + i4) = make_float4(kv[0], kv[1], kv[2], kv[3]);
            *reinterpret_cast<float4*>(geps + i4) = make_float4(epv[0], epv[1], epv[2], epv[3]);
            *reinterpret_cast<float4*>(ghv + i4) = make_float4(hvv[0], hvv[1], hvv[2], hvv[3]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (em[q]) gue[i4 + q] = L.rue[t];
        }
    }
    head = new_head;
    n_valid = new_valid;
    s_off = s_new;
    time += p.dt_d;
    __syncthreads();   // particle stores of this workgroup are visible to its own gathers below

    // (4) rotor-averaged inflow: one wave per target turbine; wake parameters of every upstream chain are
    // staged in LDS (pair[]), then the lanes sweep the (source, sample) pairs.
    float4* pair = L.pair + wave * N;
    const float ws_f = (float)ws, ti_f = (float)ti_amb;
    const float inv_S = 1.0f / (float)S;
    for (int t = wave; t < N; t += WG_NWAVES) {
        const double xt = L.xr[t];
        const float yt = (float)L.yr[t];
        const float cgt = L.cg[t];
        float tia_max = 0.f;
        for (int s2 = lane; s2 < N; s2 += WG_WAVE) {
            float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
            const double dx = xt - L.xr[s2];
            if (s2 != t && dx > 0.0) {
                const double xi = (dx - s_off) / p.dpart;
                const double jf = floor(xi);
                float wgt = (float)(xi - jf);
                long j = (long)jf;
                if (j < 0) { j = 0; wgt = 0.f; }
                if (j + 1 <= (long)n_valid - 1) {
                    int r0 = (int)(((long)head - j) % P); if (r0 < 0) r0 += P;
                    int r1 = r0 - 1; if (r1 < 0) r1 += P;
                    const size_t i0 = (size_t)s2 * P + r0, i1 = (size_t)s2 * P + r1;
                    const float w0 = 1.0f - wgt, w1 = wgt;
                    const float yc = w0 * gpy[i0] + w1 * gpy[i1];
                    const float ctv = w0 * gct[i0] + w1 * gct[i1];
                    const float kv = w0 * gk[i0] + w1 * gk[i1];
                    const float epv = w0 * geps[i0] + w1 * geps[i1];
                    const float uev = w0 * gue[i0] + w1 * gue[i1];
                    const float xd = (float)dx * p.inv_D;
                    const float sp = kv * xd + epv;
                    const float cf = m0_cfrac(ctv, sp);
                    const float sig = sp * p.D;
                    const float inv2s2 = __builtin_amdgcn_rcpf(2.0f * sig * sig);
                    pp = make_float4(yc, p.hub, inv2s2, uev * cf);
                    // wake-added turbulence (Crespo-Hernandez), Gaussian-weighted at the hub
                    const float ind = 0.5f * (1.0f - __builtin_amdgcn_sqrtf(1.0f - ctv));
                    const float xdc = fmaxf(xd, 1.0f);
                    const float rc2 = (yt - yc) * (yt - yc);
                    const float tia = p.tia * __powf(ind, p.tib) * __powf(ti_f, p.tic) * __powf(xdc, p.tid) *
                                      __expf(-rc2 * inv2s2);
                    tia_max = fmaxf(tia_max, tia);
                }
            }
            pair[s2] = pp;
        }
        float acc = 0.f;
        {
            const int total = N * S;
            int s2 = lane / S, s = lane - s2 * S;
            const int d2 = WG_WAVE / S, ds = WG_WAVE - d2 * S;
            for (int idx = lane; idx < total; idx += WG_WAVE) {
                const float4 pp = pair[s2];
                if (pp.w != 0.f) {
                    const float ys = yt + L.rdy[s] * cgt;
                    const float zs = p.hub + L.rdz[s];
                    const float r2 = (ys - pp.x) * (ys - pp.x) + (zs - pp.y) * (zs - pp.y);
                    acc += pp.w * __expf(-r2 * pp.z);
                }
                s2 += d2; s += ds;
                if (s >= S) { s -= S; ++s2; }
            }
        }
        acc = wg_wave_sum(acc);
        tia_max = wg_wave_max(tia_max);
        if (lane == 0) {
            L.u[t] = ws_f - acc * inv_S;
            L.v[t] = 0.f;
            L.w[t] = 0.f;
            L.ti[t] = sqrtf(ti_f * ti_f + tia_max * tia_max);
        }
    }
    __syncthreads();

    // (5) power / thrust with the current yaw
    for (int t = tid; t < N; t += WG_BLOCK) {
        const float g = L.yaw[t] * WG_DEG2RAD_F;
        const float sg = sinf(g), cg = L.cg[t];
        const float wsn = fmaxf(L.u[t] * cg + L.v[t] * sg, 0.0f);
        L.pow[t] = wg_tab_interp<float>(L.tabws, L.tabp, p.n_tab, wsn);
        L.ct[t] = wg_tab_interp<float>(L.tabws, L.tabct, p.n_tab, wsn) * cg * cg;
    }
    __syncthreads();
}

// replay mode (test hook): consume one scripted row instead of the physics
__device__ inline void script_step(const WgParams& p, const WgPtrs& d, const FlowLds& L, int e, int farm,
                                   int& cursor, double& time, bool advance) {
    if (advance) { ++cursor; time += p.dt_d; }
    int row = cursor < p.script_rows ? cursor : p.script_rows - 1;
    size_t base = (((size_t)farm * p.script_rows + row) * p.B + e) * p.N;
    for (int t = threadIdx.x; t < p.N; t += WG_BLOCK) {
        L.u[t] = d.script_uvw[(base + t) * 3 + 0];
        L.v[t] = d.script_uvw[(base + t) * 3 + 1];
        L.w[t] = d.script_uvw[(base + t) * 3 + 2];
        L.pow[t] = d.script_power[base + t];
    }
    __syncthreads();
}

// WindFarmEnv._take_measurements (Wind_Farm_Env.py:480-495), accumulated over the k sub-steps
__device__ inline void take_measurements(const WgParams& p, const WgPtrs& d, const FlowLds& L, int ctx_id,
                                         float wd_env) {
    for (int t = threadIdx.x; t < p.N; t += WG_BLOCK) {
        const float u = L.u[t], v = L.v[t], w = L.w[t];
        const float ws = sqrtf(u * u + v * v + w * w);
        const float wd = atanf(v / u) * WG_RAD2DEG_F + wd_env;
        d.cur_ws[(size_t)ctx_id * p.N + t] = ws;
        d.cur_wd[(size_t)ctx_id * p.N + t] = wd;
        L.sws[t] += ws; L.swd[t] += wd; L.syaw[t] += L.yaw[t]; L.sp[t] += L.pow[t];
    }
}

// farm_mes.add_measurements (MesClass.py:568-591): noise, ring push, farm-level mean/mean/sum.
// Returns (to thread 0) the farm power pushed to farm_pow_deq (Wind_Farm_Env.py:766, :975).
__device__ inline float push_measurements(const WgParams& p, const WgPtrs& d, const FlowLds& L, int ctx_id,
                                          WgCtx& cx, uint64_t noise_key) {
    const int N = p.N;
    const int n_pushed = cx.n_pushed;
    const float inv_k = 1.0f / (float)p.K;
    float* rbase = d.ring + (size_t)ctx_id * p.ring_stride;
    for (int t = threadIdx.x; t < N; t += WG_BLOCK) {
        float val[WG_N_CH] = {L.sws[t] * inv_k, L.swd[t] * inv_k, L.syaw[t] * inv_k, L.sp[t] * inv_k};
        if (p.K == 1) { val[0] = L.sws[t]; val[1] = L.swd[t]; val[2] = L.syaw[t]; val[3] = L.sp[t]; }
        if (p.noise == WG_NOISE_NORMAL) {
#pragma unroll
            for (int ch = 0; ch < WG_N_CH; ++ch)
                if (p.noise_sigma[ch] != 0.f)
                    val[ch] += p.noise_sigma[ch] *
                               wg_noise_normal(noise_key, (uint32_t)n_pushed, (uint32_t)t, (uint32_t)ch,
                                               (uint32_t)cx.episode_tag);
        }
#pragma unroll
        for (int ch = 0; ch < WG_N_CH; ++ch) {
            const int H = p.ch[ch].history_len;
            rbase[p.ring_off[ch] + (size_t)t * H + (n_pushed % H)] = val[ch];
        }
        L.sws[t] = val[0]; L.swd[t] = val[1]; L.sp[t] = val[3];
    }
    __syncthreads();
    float tot = 0.f;
    if (threadIdx.x == 0) {
        float sws = 0.f, swd = 0.f;
        for (int t = 0; t < N; ++t) { sws += L.sws[t]; swd += L.swd[t]; tot += L.sp[t]; }
        float* fbase = d.fring + (size_t)ctx_id * p.fring_stride;
        fbase[p.fring_off[WG_CH_WS] + n_pushed % p.ch[WG_CH_WS].history_len] = sws / (float)N;
        fbase[p.fring_off[WG_CH_WD] + n_pushed % p.ch[WG_CH_WD].history_len] = swd / (float)N;
        fbase[p.fring_off[WG_CH_POWER] + n_pushed % p.ch[WG_CH_POWER].history_len] = tot;
        cx.n_pushed = n_pushed + 1;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < N; t += WG_BLOCK) { L.sws[t] = 0.f; L.swd[t] = 0.f; L.syaw[t] = 0.f; L.sp[t] = 0.f; }
    __syncthreads();
    return tot;
}

// BasicControllers.local_yaw_controller / global_yaw_controller (BasicControllers.py:10-73)
__device__ inline void base_controller(const WgParams& p, const FlowLds& L) {
    for (int t = threadIdx.x; t < p.N; t += WG_BLOCK) {
        float yaw = L.yaw[t];
        if (p.base_controller == WG_CTRL_LOCAL) {
            const float wdir = atanf(L.v[t] / L.u[t]) * WG_RAD2DEG_F;
            const float off = wdir - yaw;
            const float sgn = (float)((off > 0.f) - (off < 0.f));
            yaw = yaw + sgn * fminf(fabsf(off), p.yaw_step);
        } else {
            const float sgn = (float)((yaw > 0.f) - (yaw < 0.f));
            yaw = yaw - sgn * fminf(fabsf(yaw), p.yaw_step);
        }
        L.yaw[t] = yaw;
    }
    __syncthreads();
}

// WindFarmEnv._adjust_yaws (Wind_Farm_Env.py:822-864), float32 like numpy evaluates it on a float32 action
__device__ inline void adjust_yaws(const WgParams& p, const WgPtrs& d, const FlowLds& L, int e,
                                   const float* __restrict__ actions) {
    for (int t = threadIdx.x; t < p.N; t += WG_BLOCK) {
        float yaw = L.yaw[t];
        d.old_yaw[(size_t)e * p.N + t] = yaw;          // :932
        const float a = actions[(size_t)e * p.N + t];
        if (p.action_method == WG_ACT_YAW) {
            yaw = fminf(fmaxf(yaw + a * p.yaw_step, p.yaw_min), p.yaw_max);
        } else {
            float tf = a + 1.0f;
            tf = tf / 2.0f;
            tf = tf * (p.yaw_max - p.yaw_min);
            tf = tf + p.yaw_min;
            float ny = fminf(fmaxf(tf, yaw - p.yaw_step), yaw + p.yaw_step);
            yaw = fminf(fmaxf(ny, p.yaw_min), p.yaw_max);
        }
        L.yaw[t] = yaw;
    }
    __syncthreads();
}

template <int TURB>
__global__ void __launch_bounds__(WG_BLOCK)
k_flow(const WgParams p, const WgPtrs d, const int mode, const float* __restrict__ actions,
       const uint8_t* __restrict__ mask, const int chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int F = p.F, N = p.N;
    const int bid = blockIdx.x;
    const int farm = bid % F;
    const int ec = bid / F;
    const int c = ec & 1;
    const int e = ec >> 1;
    const int tid = threadIdx.x;
    const int ctx_id = e * 2 + c;
    const int slot_id = ctx_id * F + farm;

    const WgEnv& env = d.env[e];
    const bool is_live = (c == env.live);
    WgSlot& slot = d.slot[slot_id];
    const bool ready = (slot.dev_remaining == 0 && slot.fill_remaining == 0);
    int budget = 0;
    if (mode == WG_MODE_STEP) {
        if (is_live) {
            if (env.done) return;
        } else {
            if (!p.autoreset || ready) return;
            budget = env.shadow_iters;
            if (budget <= 0) return;
        }
    } else {
        if (!is_live || ready || (mask && !mask[e])) return;
        budget = chunk;
    }

    FlowLds L = flow_lds_carve(smem, N, p.S, p.n_tab);
    WgCtx& cx = d.ctx[ctx_id];
    const double ws = cx.ws, ti_amb = cx.ti;
    const float wd_env = (float)cx.wd;
    const size_t tb = (size_t)slot_id * N;
    const size_t pbase = (size_t)slot_id * p.NP;
    for (int t = tid; t < N; t += WG_BLOCK) {
        L.xr[t] = d.xr[(size_t)ctx_id * N + t];
        L.yr[t] = d.yr[(size_t)ctx_id * N + t];
        L.yaw[t] = d.yaw[tb + t]; L.u[t] = d.u[tb + t]; L.v[t] = d.v[tb + t]; L.w[t] = d.w[tb + t];
        L.ti[t] = d.ti_loc[tb + t]; L.pow[t] = d.power[tb + t]; L.ct[t] = d.ct[tb + t];
        L.sws[t] = 0.f; L.swd[t] = 0.f; L.syaw[t] = 0.f; L.sp[t] = 0.f;
    }
    for (int i = tid; i < p.n_tab; i += WG_BLOCK) {
        L.tabws[i] = d.tab_ws[i]; L.tabp[i] = d.tab_power[i]; L.tabct[i] = d.tab_ct[i];
    }
    for (int i = tid; i < p.S; i += WG_BLOCK) { L.rdy[i] = d.rotor_dy[i]; L.rdz[i] = d.rotor_dz[i]; }
    __syncthreads();

    int head = slot.head, n_valid = slot.n_valid, cursor = slot.cursor;
    double s_off = slot.s_off, time = slot.time;
    int dev_rem = slot.dev_remaining, fill_rem = slot.fill_remaining;
    const bool replay = d.script_uvw != nullptr;

    if (mode == WG_MODE_STEP && is_live) {
        // ---- one env step of the running episode (Wind_Farm_Env.py:932-979) ----
        if (farm == 0) adjust_yaws(p, d, L, e, actions);
        float base_acc = 0.f;
        for (int k = 0; k < p.K; ++k) {
            if (farm == 1) base_controller(p, L);
            if (replay) script_step(p, d, L, e, farm, cursor, time, true);
            else flow_step<TURB>(p, d, L, pbase, ws, ti_amb, head, n_valid, s_off, time);
            if (farm == 0) {
                take_measurements(p, d, L, ctx_id, wd_env);
                __syncthreads();
            } else if (tid == 0) {
                float tot = 0.f;
                for (int t = 0; t < N; ++t) tot += L.pow[t];
                base_acc += tot;
            }
        }
        if (farm == 0) {
            float tot = push_measurements(p, d, L, ctx_id, cx, env.noise_key);
            if (tid == 0) d.step_farm_pow[e] = tot;
        } else if (tid == 0) {
            d.step_base_pow[e] = p.K == 1 ? base_acc : base_acc / (float)p.K;
        }
    } else {
        // ---- background development of a not-yet-live episode (Wind_Farm_Env.py:722-796) ----
        while (budget > 0 && (dev_rem > 0 || fill_rem > 0)) {
            if (dev_rem > 0) {
                flow_step<TURB>(p, d, L, pbase, ws, ti_amb, head, n_valid, s_off, time);
                --dev_rem; --budget;
            } else {
                float base_acc = 0.f;
                for (int k = 0; k < p.K; ++k) {
                    if (replay) script_step(p, d, L, e, farm, cursor, time, true);
                    else flow_step<TURB>(p, d, L, pbase, ws, ti_amb, head, n_valid, s_off, time);
                    if (farm == 0) {
                        take_measurements(p, d, L, ctx_id, wd_env);
                        __syncthreads();
                    } else if (tid == 0) {
                        float tot = 0.f;
                        for (int t = 0; t < N; ++t) tot += L.pow[t];
                        base_acc += tot;
                    }
                }
                if (farm == 0) {
                    float tot = push_measurements(p, d, L, ctx_id, cx, env.noise_key);
                    if (tid == 0) {
                        d.pend_farm[(size_t)ctx_id * p.power_avg + cx.pend_farm_n % p.power_avg] = tot;
                        cx.pend_farm_n += 1;
                    }
                } else if (tid == 0) {
                    d.pend_base[(size_t)ctx_id * p.power_avg + cx.pend_base_n % p.power_avg] =
                        p.K == 1 ? base_acc : base_acc / (float)p.K;
                    cx.pend_base_n += 1;
                }
                --fill_rem; budget -= p.K;
            }
        }
    }

    // write the slot back
    for (int t = tid; t < N; t += WG_BLOCK) {
        d.yaw[tb + t] = L.yaw[t]; d.u[tb + t] = L.u[t]; d.v[tb + t] = L.v[t]; d.w[tb + t] = L.w[t];
        d.ti_loc[tb + t] = L.ti[t]; d.power[tb + t] = L.pow[t]; d.ct[tb + t] = L.ct[t];
    }
    if (tid == 0) {
        slot.head = head; slot.n_valid = n_valid; slot.s_off = s_off; slot.time = time; slot.cursor = cursor;
        slot.dev_remaining = dev_rem; slot.fill_remaining = fill_rem;
    }
}

template __global__ void k_flow<WG_TURB_NONE>(const WgParams, const WgPtrs, const int, const float*,
                                              const uint8_t*, const int);

// ===================================================================================================
// episode context initialisation (wave-cooperative): WindFarmEnv.reset up to fs.run (:689-732)
// ===================================================================================================
__device__ inline void ctx_init(const WgParams& p, const WgPtrs& d, int e, int c, int lane, int episode_tag) {
    const int N = p.N, F = p.F;
    const int ctx_id = e * 2 + c;
    WgEnv& env = d.env[e];
    WgCtx& cx = d.ctx[ctx_id];
    double ws = 0, ti = 0, wd = 0;
    if (lane == 0) {
        ws = wg_pcg_uniform(env, p.ws_min, p.ws_max);     // _set_windconditions (:564-568)
        ti = wg_pcg_uniform(env, p.ti_min, p.ti_max);
        wd = wg_pcg_uniform(env, p.wd_min, p.wd_max);
        uint32_t tseed = 0;
        if (p.turb_mode == WG_TURB_RANDOM) tseed = wg_pcg_integers(env, 100000);   // _def_site (:640-644)
        for (int t = 0; t < N; ++t) {                     // yaw init (:715-720)
            float y0 = 0.f;
            if (p.yaw_init == WG_YAWINIT_RANDOM) y0 = (float)wg_pcg_uniform(env, -p.yaw_start, p.yaw_start);
            else if (p.yaw_init == WG_YAWINIT_DEFINED && p.has_yaw_defined) y0 = (float)d.yaw_defined[t];
            d.yaw[(size_t)(ctx_id * F) * N + t] = y0;
        }
        cx.ws = ws; cx.ti = ti; cx.wd = wd; cx.turb_seed = tseed;
        cx.rated_power = (float)wg_tab_interp<double>(d.tab_ws_d, d.tab_power_d, p.n_tab, ws);   // :700
        cx.n_pushed = 0; cx.pend_farm_n = 0; cx.pend_base_n = 0; cx.episode_tag = episode_tag;
    }
    ws = __shfl(ws, 0, 64); ti = __shfl(ti, 0, 64); wd = __shfl(wd, 0, 64);
    // flow frame: rotate the layout by theta = 270 - wd about the farm centre
    const double th = (270.0 - wd) * (WG_PI_D / 180.0);
    const double cth = cos(th), sth = sin(th);
    double cx0 = 0, cy0 = 0;
    for (int t = 0; t < N; ++t) { cx0 += d.x_pos[t]; cy0 += d.y_pos[t]; }
    cx0 /= N; cy0 /= N;
    double xmin = 1e300, xmax = -1e300;
    for (int t = lane; t < N; t += WG_WAVE) {
        const double dx = d.x_pos[t] - cx0, dy = d.y_pos[t] - cy0;
        const double xr = cx0 + dx * cth + dy * sth;
        const double yr = cy0 - dx * sth + dy * cth;
        d.xr[(size_t)ctx_id * N + t] = xr;
        d.yr[(size_t)ctx_id * N + t] = yr;
        xmin = fmin(xmin, xr); xmax = fmax(xmax, xr);
    }
    xmin = wg_wave_min_d(xmin); xmax = wg_wave_max_d(xmax);
    int n_dev = 0;
    if (lane == 0) {
        cx.dist = xmax - xmin;                                                     // :723-724
        cx.t_inflow = cx.dist / ws;                                                // :727
        cx.t_developed = (int)(cx.t_inflow * 2);                                   // :729
        cx.time_max = p.never_truncate ? 9999999 : (int)(cx.t_inflow * p.n_passthrough);   // :732
        n_dev = (int)ceil((double)cx.t_developed / p.dt_d - 1e-9);
        if (d.script_uvw) n_dev = 0;
        for (int f = 0; f < F; ++f) {
            WgSlot& s = d.slot[ctx_id * F + f];
            s.head = p.P - 1; s.n_valid = 0; s.s_off = 0.0; s.time = 0.0;
            s.dev_remaining = n_dev;
            s.fill_remaining = f == 0 ? p.fill_a : p.fill_b;
        }
    }
    __threadfence_block();
    for (int f = 0; f < F; ++f) {
        const size_t tb = (size_t)(ctx_id * F + f) * N;
        const int cursor = d.slot[ctx_id * F + f].cursor;
        for (int t = lane; t < N; t += WG_WAVE) {
            float y0 = d.yaw[(size_t)(ctx_id * F) * N + t];   // :781 baseline starts from the agent's yaws
            d.yaw[tb + t] = y0;
            float u = (float)ws, v = 0.f, w = 0.f, pw = 0.f;
            if (d.script_uvw) {
                int row = cursor < p.script_rows ? cursor : p.script_rows - 1;
                size_t base = (((size_t)f * p.script_rows + row) * p.B + e) * N;
                u = d.script_uvw[(base + t) * 3]; v = d.script_uvw[(base + t) * 3 + 1];
                w = d.script_uvw[(base + t) * 3 + 2]; pw = d.script_power[base + t];
            }
            d.u[tb + t] = u; d.v[tb + t] = v; d.w[tb + t] = w;
            d.ti_loc[tb + t] = (float)ti; d.power[tb + t] = pw; d.ct[tb + t] = 0.f;
        }
    }
}

// ===================================================================================================
// observation (farm_mes.get_measurements(scaled=True) + clip, MesClass.py:679-703, Wind_Farm_Env.py:513-520)
// ===================================================================================================
__device__ inline void build_obs(const WgParams& p, const WgPtrs& d, int ctx_id, int lane, float* __restrict__ obs,
                                 float* __restrict__ obs2) {
    const int N = p.N;
    const int n_pushed = d.ctx[ctx_id].n_pushed;
    float ti_sum = 0.f;
    float buf[8];
    for (int t = lane; t < N; t += WG_WAVE) {
        // turbine block written straight to its place (block length is fixed = turb_obs)
        float* o = obs + (size_t)t * p.turb_obs;
        int n = 0;
        const float* rbase = d.ring + (size_t)ctx_id * p.ring_stride;
        for (int ch = 0; ch < WG_N_CH; ++ch) {
            const int H = p.ch[ch].history_len;
            if (ch == WG_CH_POWER && p.turb_ti) {
                WgRing r{rbase + p.ring_off[WG_CH_WS] + (size_t)t * p.ch[WG_CH_WS].history_len, n_pushed,
                         p.ch[WG_CH_WS].history_len};
                float v = wg_clip1(wg_scale(wg_calc_ti(r), p.ti_min_f, p.ti_rng_f));
                o[n] = v; if (obs2) obs2[(size_t)t * p.turb_obs + n] = v;
                ++n;
            }
            WgRing r{rbase + p.ring_off[ch] + (size_t)t * H, n_pushed, H};
            const bool on = p.turb_on[ch] != 0;
            const bool cur_on = p.ch[ch].current && on, rol_on = p.ch[ch].rolling_mean && on;
            // stream the values out one at a time (window count is unbounded: history_N up to 100s)
            const int avail = r.avail();
            if (avail == 0) continue;
            if (cur_on) {
                float v = wg_clip1(wg_scale(r.at(avail - 1), p.sc_min[ch], p.sc_rng[ch]));
                o[n] = v; if (obs2) obs2[(size_t)t * p.turb_obs + n] = v;
                ++n;
            }
            if (rol_on) {
                const int W = p.ch[ch].window_len, HN = p.ch[ch].history_n;
                for (int i = 0; i < HN; ++i) {
                    int lo, hi;
                    if (i == 0) { lo = avail - W; if (lo < 0) lo = 0; hi = avail; }
                    else if (i == HN - 1 && avail >= W) { lo = 0; hi = W; }
                    else if (avail < W) { lo = 0; hi = avail; }
                    else {
                        int spacing = (avail - W) / (HN - 1); if (spacing < 1) spacing = 1;
                        int pos = i * spacing; if (pos > avail - W) pos = avail - W;
                        lo = pos; hi = pos + W;
                    }
                    float s = 0.f;
                    for (int q = lo; q < hi; ++q) s += r.at(q);
                    float v = wg_clip1(wg_scale(s / (float)(hi - lo), p.sc_min[ch], p.sc_rng[ch]));
                    o[n] = v; if (obs2) obs2[(size_t)t * p.turb_obs + n] = v;
                    ++n;
                }
            }
        }
        if (p.farm_ti) {   // farm TI = mean of the *scaled* turbine TIs (MesClass.py:670-673)
            WgRing r{rbase + p.ring_off[WG_CH_WS] + (size_t)t * p.ch[WG_CH_WS].history_len, n_pushed,
                     p.ch[WG_CH_WS].history_len};
            ti_sum += wg_scale(wg_calc_ti(r), p.ti_min_f, p.ti_rng_f);
        }
    }
    if (p.farm_ti) ti_sum = wg_wave_sum(ti_sum);
    if (lane == 0 && p.farm_obs > 0) {
        float* o = obs + (size_t)N * p.turb_obs;
        float* o2 = obs2 ? obs2 + (size_t)N * p.turb_obs : nullptr;
        int n = 0;
        const float* fbase = d.fring + (size_t)ctx_id * p.fring_stride;
        const int chs[3] = {WG_CH_WS, WG_CH_WD, WG_CH_POWER};
        for (int ci = 0; ci < 3; ++ci) {
            const int ch = chs[ci];
            if (ch == WG_CH_POWER && p.farm_ti) {
                float v = wg_clip1(ti_sum / (float)N);
                o[n] = v; if (o2) o2[n] = v;
                ++n;
            }
            if (!p.farm_on[ch]) continue;
            WgRing r{fbase + p.fring_off[ch], n_pushed, p.ch[ch].history_len};
            const float rng = ch == WG_CH_POWER ? p.sc_rng_farm_power : p.sc_rng[ch];
            // farm windows are few: reuse the generic helper through a small buffer when it fits
            const int cnt = (p.ch[ch].current ? 1 : 0) + (p.ch[ch].rolling_mean ? p.ch[ch].history_n : 0);
            if (cnt <= 8) {
                int m = wg_mes_get(p.ch[ch], p.ch[ch].current, p.ch[ch].rolling_mean, r, p.sc_min[ch], rng, buf);
                for (int i = 0; i < m; ++i) { float v = wg_clip1(buf[i]); o[n] = v; if (o2) o2[n] = v; ++n; }
            } else {
                const int avail = r.avail();
                if (avail == 0) continue;
                if (p.ch[ch].current) { float v = wg_clip1(wg_scale(r.at(avail - 1), p.sc_min[ch], rng)); o[n] = v; if (o2) o2[n] = v; ++n; }
                if (p.ch[ch].rolling_mean) {
                    const int W = p.ch[ch].window_len, HN = p.ch[ch].history_n;
                    for (int i = 0; i < HN; ++i) {
                        int lo, hi;
                        if (i == 0) { lo = avail - W; if (lo < 0) lo = 0; hi = avail; }
                        else if (i == HN - 1 && avail >= W) { lo = 0; hi = W; }
                        else if (avail < W) { lo = 0; hi = avail; }
                        else {
                            int spacing = (avail - W) / (HN - 1); if (spacing < 1) spacing = 1;
                            int pos = i * spacing; if (pos > avail - W) pos = avail - W;
                            lo = pos; hi = pos + W;
                        }
                        float s = 0.f;
                        for (int q = lo; q < hi; ++q) s += r.at(q);
                        float v = wg_clip1(wg_scale(s / (float)(hi - lo), p.sc_min[ch], rng));
                        o[n] = v; if (o2) o2[n] = v; ++n;
                    }
                }
            }
        }
    }
}

__device__ inline double deque_mean(const float* dq, int n_total, int maxlen) {
    const int n = n_total < maxlen ? n_total : maxlen;
    double s = 0;
    for (int i = 0; i < n; ++i) s += (double)dq[i];
    return s / (double)n;
}
__device__ inline float deque_at(const float* dq, int n_total, int maxlen, int q) {
    const int n = n_total < maxlen ? n_total : maxlen;
    return dq[(n_total - n + q) % maxlen];
}

// plan how many flow sub-steps the background episode must advance during the next step() so that it is
// ready exactly when the running episode truncates
__device__ inline int plan_shadow(const WgParams& p, const WgPtrs& d, int e) {
    const WgEnv& env = d.env[e];
    const int live = env.live, sh = live ^ 1;
    int work = 0;
    for (int f = 0; f < p.F; ++f) {
        const WgSlot& s = d.slot[(e * 2 + sh) * p.F + f];
        int w = s.dev_remaining + p.K * s.fill_remaining;
        if (w > work) work = w;
    }
    if (work == 0) return 0;
    const int inc = 1 + (p.extra_inc ? 1 : 0);
    const int tm = d.ctx[e * 2 + live].time_max;
    const long total = (long)((tm + inc - 1) / inc) + 1;
    long left = total - env.steps_done;
    if (left < 1) left = 1;
    return (int)((work + left - 1) / left);
}

// ===================================================================================================
// k_glue: one wave per env.  phase 0 = after a flow step (step()); phase 1 = end of reset().
// ===================================================================================================
__global__ void __launch_bounds__(WG_BLOCK)
k_glue(const WgParams p, const WgPtrs d, const int phase, const uint8_t* __restrict__ mask,
       float* __restrict__ obs_out, float* __restrict__ reward_out, uint8_t* __restrict__ trunc_out,
       float* __restrict__ final_obs_out) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * WG_NWAVES + (threadIdx.x >> 6);
    if (e >= p.B) return;
    WgEnv& env = d.env[e];
    const int N = p.N;
    float* obs = obs_out ? obs_out + (size_t)e * p.obs_dim : nullptr;

    if (phase == 1) {
        if (mask && !mask[e]) return;
        // the freshly developed episode goes live: flush its deferred power-deque pushes (:766, :796)
        const int ctx_id = e * 2 + env.live;
        WgCtx& cx = d.ctx[ctx_id];
        if (lane == 0) {
            const int nf = cx.pend_farm_n < p.power_avg ? cx.pend_farm_n : p.power_avg;
            for (int q = 0; q < nf; ++q) {
                float v = deque_at(d.pend_farm + (size_t)ctx_id * p.power_avg, cx.pend_farm_n, p.power_avg, q);
                d.farm_pow[(size_t)e * p.power_avg + env.farm_pow_n % p.power_avg] = v;
                env.farm_pow_n++;
            }
            const int nb = cx.pend_base_n < p.power_avg ? cx.pend_base_n : p.power_avg;
            for (int q = 0; q < nb; ++q) {
                float v = deque_at(d.pend_base + (size_t)ctx_id * p.power_avg, cx.pend_base_n, p.power_avg, q);
                d.base_pow[(size_t)e * p.power_avg + env.base_pow_n % p.power_avg] = v;
                env.base_pow_n++;
            }
            cx.pend_farm_n = 0; cx.pend_base_n = 0;
            env.timestep = 0; env.done = 0; env.steps_done = 0;
            env.ep_return = 0.f; env.ep_power_sum = 0.f; env.ep_len = 0;
            env.shadow_iters = p.autoreset ? plan_shadow(p, d, e) : 0;
        }
        if (obs) build_obs(p, d, ctx_id, lane, obs, nullptr);
        return;
    }

    if (env.done) {
        if (lane == 0) atomicMin(d.status, (int)WG_ERR_STATE);
        return;
    }
    const int live = env.live;
    const int ctx_id = e * 2 + live;
    WgCtx& cx = d.ctx[ctx_id];
    const size_t tb_a = (size_t)(ctx_id * p.F) * N;

    // power deques (:975-981)
    if (lane == 0) {
        const float fp = d.step_farm_pow[e];
        d.farm_pow[(size_t)e * p.power_avg + env.farm_pow_n % p.power_avg] = fp;
        env.farm_pow_n++;
        if (p.F == 2) {
            d.base_pow[(size_t)e * p.power_avg + env.base_pow_n % p.power_avg] = d.step_base_pow[e];
            env.base_pow_n++;
        }
        if (fp != fp) atomicMin(d.status, (int)WG_ERR_NAN_POWER);
    }
    // observation (:983)
    float* fin = final_obs_out ? final_obs_out + (size_t)e * p.obs_dim : nullptr;
    build_obs(p, d, ctx_id, lane, obs, fin);

    // action penalty sums (:804-820) and current farm powers ("Power agent", :539)
    float pen_s = 0.f, pnow = 0.f, pbase = 0.f;
    for (int t = lane; t < N; t += WG_WAVE) {
        const float y = d.yaw[tb_a + t];
        pen_s += p.penalty_type == WG_PEN_CHANGE ? fabsf(d.old_yaw[(size_t)e * N + t] - y) : fabsf(y);
        pnow += d.power[tb_a + t];
        if (p.F == 2) pbase += d.power[tb_a + N + t];
    }
    pen_s = wg_wave_sum(pen_s); pnow = wg_wave_sum(pnow); pbase = wg_wave_sum(pbase);

    int truncated = 0;
    if (lane == 0) {
        double pr = 0.0;
        const float* fq = d.farm_pow + (size_t)e * p.power_avg;
        const float* bq = d.base_pow + (size_t)e * p.power_avg;
        switch (p.reward_mode) {
        case WG_REW_BASELINE:
            pr = deque_mean(fq, env.farm_pow_n, p.power_avg) / deque_mean(bq, env.base_pow_n, p.power_avg) - 1.0;
            break;
        case WG_REW_POWER_AVG:
            pr = deque_mean(fq, env.farm_pow_n, p.power_avg) / N / (double)cx.rated_power;
            break;
        case WG_REW_NONE: pr = 0.0; break;
        case WG_REW_POWER_DIFF: {
            const int wsz = p.power_avg / 10;
            const int n = env.farm_pow_n < p.power_avg ? env.farm_pow_n : p.power_avg;
            double sl = 0, so = 0; int nl = 0, no = 0;
            for (int q = p.power_avg - wsz; q < p.power_avg && q < n; ++q) { sl += deque_at(fq, env.farm_pow_n, p.power_avg, q); nl++; }
            for (int q = 0; q < wsz && q < n; ++q) { so += deque_at(fq, env.farm_pow_n, p.power_avg, q); no++; }
            pr = (sl / (double)nl - so / (double)no) / N;
            break;
        }
        }
        double pen = 0.0;
        if (p.action_penalty >= 0.001) {
            pen = p.penalty_type == WG_PEN_CHANGE ? p.action_penalty * ((double)pen_s / N)
                                                  : p.action_penalty * ((double)pen_s / N / p.yaw_max_d);
        }
        const float reward = (float)(pr * p.power_scaling + 0.0 - pen);             // :989-996
        truncated = env.timestep >= cx.time_max;                                    // :1003
        env.timestep += 1 + (p.extra_inc ? 1 : 0);                                  // :1027
        env.steps_done += 1;
        if (reward_out) reward_out[e] = reward;
        if (trunc_out) trunc_out[e] = (uint8_t)truncated;
        // episode metrics (recordEpisodeVals.py:31-64; longer_steps_example.py:39-124)
        float* met = d.metrics + (size_t)e * WG_N_METRICS;
        env.ep_return += reward; env.ep_power_sum += pnow; env.ep_len += 1;
        met[WG_MET_STEP_REWARD_SUM] += reward;
        met[WG_MET_FARM_POWER_SUM] += pnow;
        met[WG_MET_BASE_POWER_SUM] += pbase;
        met[WG_MET_N_STEPS] += 1.f;
        if (truncated) {
            met[WG_MET_EP_RETURN_SUM] += env.ep_return;
            met[WG_MET_EP_LENGTH_SUM] += (float)env.ep_len;
            met[WG_MET_EP_MEAN_POWER_SUM] += env.ep_power_sum / (float)env.ep_len;
            met[WG_MET_N_EPISODES] += 1.f;
            env.ep_return = 0.f; env.ep_power_sum = 0.f; env.ep_len = 0;
            env.episode += 1;
        }
    }
    truncated = __shfl(truncated, 0, 64);
    if (truncated) {
        if (!p.autoreset) {
            if (lane == 0) env.done = 1;
            return;
        }
        // same-step autoreset: the next episode was developed in the background; make it live
        const int nxt = live ^ 1;
        const int nctx = e * 2 + nxt;
        WgCtx& ncx = d.ctx[nctx];
        if (lane == 0) {
            bool ok = true;
            for (int f = 0; f < p.F; ++f) {
                const WgSlot& s = d.slot[nctx * p.F + f];
                ok = ok && s.dev_remaining == 0 && s.fill_remaining == 0;
            }
            if (!ok) atomicMin(d.status, (int)WG_ERR_STATE);
            const int nf = ncx.pend_farm_n < p.power_avg ? ncx.pend_farm_n : p.power_avg;
            for (int q = 0; q < nf; ++q) {
                float v = deque_at(d.pend_farm + (size_t)nctx * p.power_avg, ncx.pend_farm_n, p.power_avg, q);
                d.farm_pow[(size_t)e * p.power_avg + env.farm_pow_n % p.power_avg] = v;
                env.farm_pow_n++;
            }
            const int nb = ncx.pend_base_n < p.power_avg ? ncx.pend_base_n : p.power_avg;
            for (int q = 0; q < nb; ++q) {
                float v = deque_at(d.pend_base + (size_t)nctx * p.power_avg, ncx.pend_base_n, p.power_avg, q);
                d.base_pow[(size_t)e * p.power_avg + env.base_pow_n % p.power_avg] = v;
                env.base_pow_n++;
            }
            ncx.pend_farm_n = 0; ncx.pend_base_n = 0;
            env.live = nxt; env.timestep = 0; env.steps_done = 0;
        }
        __threadfence_block();
        if (obs) build_obs(p, d, nctx, lane, obs, nullptr);
        // the retired context starts developing the episode after the next one
        ctx_init(p, d, e, live, lane, env.episode + 1);
        __threadfence_block();
    }
    if (lane == 0) env.shadow_iters = p.autoreset ? plan_shadow(p, d, e) : 0;
}

// ===================================================================================================
// k_init: reset() bookkeeping for the masked envs: (re)seed, sample, initialise the context(s)
// ===================================================================================================
__global__ void __launch_bounds__(WG_BLOCK)
k_init(const WgParams p, const WgPtrs d, const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seeds) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * WG_NWAVES + (threadIdx.x >> 6);
    if (e >= p.B) return;
    if (mask && !mask[e]) return;
    WgEnv& env = d.env[e];
    if (lane == 0) {
        if (seeds && seeds[e] != 0xFFFFFFFFFFFFFFFFull) {
            wg_pcg_seed(env, seeds[e]);
            env.noise_key = seeds[e];
        }
        env.done = 0;
    }
    __threadfence_block();
    const int live = env.live;
    ctx_init(p, d, e, live, lane, env.episode);
    if (p.autoreset) ctx_init(p, d, e, live ^ 1, lane, env.episode + 1);
}

// fresh-handle initialisation: generator state of np.random.default_rng(b)
__global__ void k_create(const WgParams p, const WgPtrs d) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.B) return;
    WgEnv& env = d.env[e];
    wg_pcg_seed(env, (uint64_t)e);
    env.noise_key = (uint64_t)e;
    env.live = 0; env.timestep = 0; env.episode = 0; env.done = 1; env.shadow_iters = 0;
    env.farm_pow_n = 0; env.base_pow_n = 0; env.steps_done = 0;
    env.ep_return = 0.f; env.ep_power_sum = 0.f; env.ep_len = 0;
}

// ===================================================================================================
// k_obs_multi: WindFarmEnvMulti._get_obs_multi (WindEnvMulti.py:79-103): [B, N, obs_dim_multi]
// ===================================================================================================
__global__ void k_obs_multi(const WgParams p, const WgPtrs d, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.B * p.N) return;
    const int e = i / p.N, t = i - e * p.N;
    const int ctx_id = e * 2 + d.env[e].live;
    const int n_pushed = d.ctx[ctx_id].n_pushed;
    float* o = out + (size_t)i * p.obs_dim_multi;
    float buf[64];
    // blocks can be long (history_N); write through a bounded staging buffer in pieces
    int n = 0;
    if (p.turb_obs <= 64) {
        int m = wg_turb_block(p, d, ctx_id, n_pushed, t, false, buf);
        for (int k = 0; k < m; ++k) o[n++] = wg_clip1(buf[k]);
    } else {
        // long blocks: reuse the single-agent observation of this turbine is not available here; recompute
        // directly into the output (values are scaled; clip in place afterwards)
        int m = wg_turb_block(p, d, ctx_id, n_pushed, t, false, o);
        for (int k = 0; k < m; ++k) o[k] = wg_clip1(o[k]);
        n = m;
    }
    if (p.farm_obs <= 64) {
        int m = wg_turb_block(p, d, ctx_id, n_pushed, 0, true, buf);
        for (int k = 0; k < m; ++k) o[n++] = wg_clip1(buf[k]);
    } else {
        int m = wg_turb_block(p, d, ctx_id, n_pushed, 0, true, o + n);
        for (int k = 0; k < m; ++k) o[n + k] = wg_clip1(o[n + k]);
    }
}

// ===================================================================================================
// k_info: lazy info dict (Wind_Farm_Env.py:522-555)
// ===================================================================================================
__global__ void k_info(const WgParams p, const WgPtrs d, const int field, void* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = p.N;
    const int per = (field == WG_INFO_ROTOR_UVW_AGENT || field == WG_INFO_ROTOR_UVW_BASE) ? N
                    : (field == WG_INFO_YAW_AGENT || field == WG_INFO_YAW_BASE || field == WG_INFO_WS_TURB ||
                       field == WG_INFO_WD_TURB || field == WG_INFO_POWER_TURB_AGENT ||
                       field == WG_INFO_POWER_TURB_BASE || field == WG_INFO_WS_TURB_BASE ||
                       field == WG_INFO_TURB_X || field == WG_INFO_TURB_Y) ? N : 1;
    if (i >= p.B * per) return;
    const int e = i / per, t = i - e * per;
    const WgEnv& env = d.env[e];
    const int ctx_id = e * 2 + env.live;
    const WgCtx& cx = d.ctx[ctx_id];
    const size_t ta = (size_t)(ctx_id * p.F) * N + t;
    const size_t tbs = (size_t)(ctx_id * p.F + (p.F - 1)) * N + t;
    float* fo = (float*)out;
    int* io = (int*)out;
    switch (field) {
    case WG_INFO_YAW_AGENT: fo[i] = d.yaw[ta]; break;
    case WG_INFO_YAW_BASE: fo[i] = d.yaw[tbs]; break;
    case WG_INFO_WS_GLOBAL: fo[i] = (float)cx.ws; break;
    case WG_INFO_WD_GLOBAL: fo[i] = (float)cx.wd; break;
    case WG_INFO_TI_GLOBAL: fo[i] = (float)cx.ti; break;
    case WG_INFO_WS_TURB: fo[i] = d.cur_ws[(size_t)ctx_id * N + t]; break;
    case WG_INFO_WD_TURB: fo[i] = d.cur_wd[(size_t)ctx_id * N + t]; break;
    case WG_INFO_POWER_TURB_AGENT: fo[i] = d.power[ta]; break;
    case WG_INFO_POWER_TURB_BASE: fo[i] = d.power[tbs]; break;
    case WG_INFO_POWER_AGENT: { float s = 0.f; for (int k = 0; k < N; ++k) s += d.power[ta + k]; fo[i] = s; break; }
    case WG_INFO_POWER_BASE: { float s = 0.f; for (int k = 0; k < N; ++k) s += d.power[tbs + k]; fo[i] = s; break; }
    case WG_INFO_WS_TURB_BASE: fo[i] = d.u[tbs]; break;
    case WG_INFO_TURB_X: fo[i] = (float)d.xr[(size_t)ctx_id * N + t]; break;
    case WG_INFO_TURB_Y: fo[i] = (float)d.yr[(size_t)ctx_id * N + t]; break;
    case WG_INFO_TIMESTEP: io[i] = env.timestep; break;
    case WG_INFO_TIME_MAX: io[i] = cx.time_max; break;
    case WG_INFO_FS_TIME: fo[i] = (float)d.slot[ctx_id * p.F].time; break;
    case WG_INFO_EPISODE: io[i] = env.episode; break;
    case WG_INFO_ROTOR_UVW_AGENT: fo[i * 3] = d.u[ta]; fo[i * 3 + 1] = d.v[ta]; fo[i * 3 + 2] = d.w[ta]; break;
    case WG_INFO_ROTOR_UVW_BASE: fo[i * 3] = d.u[tbs]; fo[i * 3 + 1] = d.v[tbs]; fo[i * 3 + 2] = d.w[tbs]; break;
    case WG_INFO_RATED_POWER: fo[i] = cx.rated_power; break;
    default: break;
    }
}

// ===================================================================================================
// k_metrics: deterministic reduction of the per-env running sums -> f32[WG_N_METRICS]
// ===================================================================================================
__global__ void __launch_bounds__(WG_BLOCK) k_metrics(const WgParams p, const WgPtrs d, float* __restrict__ out,
                                                      const int reset_after) {
    __shared__ float red[WG_BLOCK];
    for (int m = 0; m < WG_N_METRICS; ++m) {
        float s = 0.f;
        for (int e = threadIdx.x; e < p.B; e += WG_BLOCK) s += d.metrics[(size_t)e * WG_N_METRICS + m];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = WG_BLOCK / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[m] = red[0];
        __syncthreads();
    }
    if (reset_after)
        for (int i = threadIdx.x; i < p.B * WG_N_METRICS; i += WG_BLOCK) d.metrics[i] = 0.f;
}

// host-visible launch helpers (defined here so that the <<<>>> syntax stays in one translation unit)
extern "C" void wg_launch_flow(const WgParams* p, const WgPtrs* d, int mode, const float* actions,
                               const uint8_t* mask, int chunk, hipStream_t st) {
    const int grid = p->B * 2 * p->F;
    const size_t lds = flow_lds_bytes(p->N, p->S, p->n_tab);
    hipLaunchKernelGGL(k_flow<WG_TURB_NONE>, dim3(grid), dim3(WG_BLOCK), lds, st, *p, *d, mode, actions, mask, chunk);
}
extern "C" void wg_launch_glue(const WgParams* p, const WgPtrs* d, int phase, const uint8_t* mask, float* obs,
                               float* reward, uint8_t* trunc, float* final_obs, hipStream_t st) {
    const int grid = (p->B + WG_NWAVES - 1) / WG_NWAVES;
    hipLaunchKernelGGL(k_glue, dim3(grid), dim3(WG_BLOCK), 0, st, *p, *d, phase, mask, obs, reward, trunc, final_obs);
}
extern "C" void wg_launch_init(const WgParams* p, const WgPtrs* d, const uint8_t* mask, const uint64_t* seeds,
                               hipStream_t st) {
    const int grid = (p->B + WG_NWAVES - 1) / WG_NWAVES;
    hipLaunchKernelGGL(k_init, dim3(grid), dim3(WG_BLOCK), 0, st, *p, *d, mask, seeds);
}
extern "C" void wg_launch_create(const WgParams* p, const WgPtrs* d, hipStream_t st) {
    hipLaunchKernelGGL(k_create, dim3((p->B + 255) / 256), dim3(256), 0, st, *p, *d);
}
extern "C" void wg_launch_obs_multi(const WgParams* p, const WgPtrs* d, float* out, hipStream_t st) {
    const int n = p->B * p->N;
    hipLaunchKernelGGL(k_obs_multi, dim3((n + 255) / 256), dim3(256), 0, st, *p, *d, out);
}
extern "C" void wg_launch_info(const WgParams* p, const WgPtrs* d, int field, void* out, hipStream_t st) {
    const int n = p->B * p->N;
    hipLaunchKernelGGL(k_info, dim3((n + 255) / 256), dim3(256), 0, st, *p, *d, field, out);
}
extern "C" void wg_launch_metrics(const WgParams* p, const WgPtrs* d, float* out, int reset_after, hipStream_t st) {
    hipLaunchKernelGGL(k_metrics, dim3(1), dim3(WG_BLOCK), 0, st, *p, *d, out, reset_after);
}
