// wg_flow.hip — k_flow, the dominant kernel of libwindgym_hip.so (gfx950, wave64).
//
// One 256-thread workgroup per farm slot (env x ctx x farm).  Per flow step it
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

#include "wg_device.h"
#include "wg_flow.h"

struct __attribute__((aligned(8))) TurbLds {
    double xr, yr;
    float yaw, u, v, w, ti, pow, ct, cg, sg;
    float rct, rk, reps, rhv, rue;
    float sws, swd, syaw, sp;
};   // 22 dwords: lanes t = 0..15 of a column access hit 16 distinct banks
static_assert(sizeof(TurbLds) == WG_TURB_LDS_BYTES, "keep WG_TURB_LDS_BYTES in sync");

__device__ __forceinline__ float m0_cfrac(float ct, float sp) {
    float m = __builtin_amdgcn_rcpf(8.0f * sp * sp);
    m = fminf(m, 1.0f);
    const float a = fmaxf(1.0f - ct * m, 0.0f);
    return 1.0f - __builtin_amdgcn_sqrtf(a);
}

// uniform-grid table lookup (linear interpolation, 0 outside)
__device__ __forceinline__ float tab_lookup(const float* __restrict__ ys, const FlowP& p, float x) {
    const float fx = (x - p.tab_x0) * p.tab_inv_dx;
    if (!(fx >= 0.0f) || fx > (float)(p.n_tab - 1)) return 0.0f;
    int i = (int)fx;
    if (i > p.n_tab - 2) i = p.n_tab - 2;
    const float f = fx - (float)i;
    return ys[i] + f * (ys[i + 1] - ys[i]);
}

struct SlotRegs {
    double s_off, time;
    int head, n_valid;
};

__device__ __forceinline__ void flow_step(const FlowP& p, const FlowPtrs& d, TurbLds* __restrict__ T,
                                          const float* __restrict__ tabp, const float* __restrict__ tabct,
                                          const float* __restrict__ rdy, const float* __restrict__ rdz,
                                          float4* __restrict__ pair, const size_t pbase, const double ws,
                                          const float ti_f, const float ti_pow, SlotRegs& sr) {
    const int tid = threadIdx.x;
    const int N = p.N, P = p.P;

    // (1) emission records of this step, sin/cos of the yaw
    if (tid < N) {
        for (int t = tid; t < N; t += WG_BLOCK) {
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
            q.rue = q.u;
            q.cg = cg;
            q.sg = sg;
        }
    }
    __syncthreads();

    // (2) streaming pass over the particle SoA: advect over dt, release the new particles
    const int head = sr.head, n_valid = sr.n_valid;
    double s_new = sr.s_off + ws * p.dt_d;
    int n_emit = 0;
    while (s_new >= p.dpart) { s_new -= p.dpart; ++n_emit; }
    if (n_emit > P) n_emit = P;
    int new_head = head + n_emit; if (new_head >= P) new_head -= P;
    int new_valid = n_valid + n_emit; if (new_valid > P) new_valid = P;
    const float s_off_f = (float)sr.s_off;

    float* __restrict__ gpy = d.py + pbase;
    float* __restrict__ gct = d.ct_e + pbase;
    float* __restrict__ gk = d.k_e + pbase;
    float* __restrict__ geps = d.eps_e + pbase;
    float* __restrict__ ghv = d.hv_e + pbase;
    float* __restrict__ gue = d.u_e + pbase;
    {
        // thread -> 4 consecutive ring slots of one turbine (P % 4 == 0); (t, r0) advance incrementally
        int t = (tid * 4) / P;
        int r0 = tid * 4 - t * P;
        const int dt_ = (WG_BLOCK * 4) / P, dr_ = (WG_BLOCK * 4) - dt_ * P;
        for (int i4 = tid * 4; i4 < p.NP; i4 += WG_BLOCK * 4) {
            float4 py4 = *reinterpret_cast<const float4*>(gpy + i4);
            const float4 hv4 = *reinterpret_cast<const float4*>(ghv + i4);
            int j0 = head - r0; if (j0 < 0) j0 += P;              // age of ring slot r0 (slot r0+q: j0-q)
            int e0 = r0 - head - 1; if (e0 < 0) e0 += P;          // emission index of slot r0 (r0+q: e0+q)
            const bool emits = (e0 < n_emit) || (n_emit > 0 && e0 + 3 >= P);   // slots r0..r0+3 wrap past e = P-1 -> 0
            const bool moves = (hv4.x != 0.f) | (hv4.y != 0.f) | (hv4.z != 0.f) | (hv4.w != 0.f);
            if (moves || emits) {
                float4 ct4 = *reinterpret_cast<const float4*>(gct + i4);
                float4 k4 = *reinterpret_cast<const float4*>(gk + i4);
                float4 ep4 = *reinterpret_cast<const float4*>(geps + i4);
                float pyv[4] = {py4.x, py4.y, py4.z, py4.w};
                const float hvv[4] = {hv4.x, hv4.y, hv4.z, hv4.w};
                const float ctv[4] = {ct4.x, ct4.y, ct4.z, ct4.w};
                const float kv[4] = {k4.x, k4.y, k4.z, k4.w};
                const float epv[4] = {ep4.x, ep4.y, ep4.z, ep4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int j = j0 - q; if (j < 0) j += P;
                    if (j < n_valid) {
                        const float xrel = s_off_f + (float)j * p.dpart_f;
                        const float sp = kv[q] * (xrel * p.inv_D) + epv[q];
                        pyv[q] += hvv[q] * m0_cfrac(ctv[q], sp) * p.dt;
                    }
                }
                if (emits) {
                    const TurbLds& tq = T[t];
                    const float y0 = (float)tq.yr;
                    float c4[4] = {ctv[0], ctv[1], ctv[2], ctv[3]}, kk[4] = {kv[0], kv[1], kv[2], kv[3]};
                    float e4[4] = {epv[0], epv[1], epv[2], epv[3]}, h4[4] = {hvv[0], hvv[1], hvv[2], hvv[3]};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        int ei = e0 + q; if (ei >= P) ei -= P;
                        if (ei < n_emit) {
                            pyv[q] = y0; c4[q] = tq.rct; kk[q] = tq.rk; e4[q] = tq.reps; h4[q] = tq.rhv;
                            gue[i4 + q] = tq.rue;
                        }
                    }
                    *reinterpret_cast<float4*>(gct + i4) = make_float4(c4[0], c4[1], c4[2], c4[3]);
                    *reinterpret_cast<float4*>(gk + i4) = make_float4(kk[0], kk[1], kk[2], kk[3]);
                    *reinterpret_cast<float4*>(geps + i4) = make_float4(e4[0], e4[1], e4[2], e4[3]);
                    *reinterpret_cast<float4*>(ghv + i4) = make_float4(h4[0], h4[1], h4[2], h4[3]);
                }
                *reinterpret_cast<float4*>(gpy + i4) = make_float4(pyv[0], pyv[1], pyv[2], pyv[3]);
            }
            t += dt_; r0 += dr_;
            if (r0 >= P) { r0 -= P; ++t; }
        }
    }
    sr.head = new_head; sr.n_valid = new_valid; sr.s_off = s_new; sr.time += p.dt_d;
    __syncthreads();   // this workgroup's particle stores are visible to its own gathers below (same CU)

    // (3)+(4) rotor-averaged inflow
    const int TC = p.target_chunk;
    const float ws_f = (float)ws;
    for (int t0 = 0; t0 < N; t0 += TC) {
        const int nt = (N - t0) < TC ? (N - t0) : TC;
        const int npairs = nt * N;
        // phase A: one thread per (target, source) pair
        for (int i = tid; i < npairs; i += WG_BLOCK) {
            const int tl = (int)(((float)i + 0.5f) * p.inv_N);
            const int s2 = i - tl * N;
            const int t = t0 + tl;
            float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
            const double dx = T[t].xr - T[s2].xr;
            if (s2 != t && dx > 0.0) {
                const double xi = (dx - s_new) * p.inv_dpart;
                const double jf = floor(xi);
                float wgt = (float)(xi - jf);
                int j = (int)jf;
                if (j < 0) { j = 0; wgt = 0.f; }
                if (j + 1 <= new_valid - 1) {
                    int r0 = new_head - j; if (r0 < 0) r0 += P;
                    int r1 = r0 - 1; if (r1 < 0) r1 += P;
                    const int i0 = s2 * P + r0, i1 = s2 * P + r1;
                    const float w0 = 1.0f - wgt, w1 = wgt;
                    const float yc = w0 * gpy[i0] + w1 * gpy[i1];
                    const float kv = w0 * gk[i0] + w1 * gk[i1];
                    const float epv = w0 * geps[i0] + w1 * geps[i1];
                    const float xd = (float)dx * p.inv_D;
                    const float sp = kv * xd + epv;
                    const float sig = sp * p.D;
                    const float yt = (float)T[t].yr;
                    const float rc2 = (yt - yc) * (yt - yc);
                    const float rcut = p.R_rot + 5.0f * sig;
                    if (rc2 <= rcut * rcut) {
                        const float ctv = w0 * gct[i0] + w1 * gct[i1];
                        const float uev = w0 * gue[i0] + w1 * gue[i1];
                        const float cf = m0_cfrac(ctv, sp);
                        const float inv2s2 = __builtin_amdgcn_rcpf(2.0f * sig * sig);
                        // wake-added turbulence (Crespo-Hernandez), Gaussian-weighted at the hub
                        const float ind = 0.5f * (1.0f - __builtin_amdgcn_sqrtf(1.0f - ctv));
                        const float tia = p.tia * __powf(ind, p.tib) * ti_pow * __powf(fmaxf(xd, 1.0f), p.tid) *
                                          __expf(-rc2 * inv2s2);
                        pp = make_float4(yc, inv2s2, uev * cf, tia);
                    }
                }
            }
            pair[i] = pp;
        }
        __syncthreads();
        // phase B: one thread per (target, sample); S_pad = S rounded up to a power of two
        const int nitems = nt << p.S_shift;
        for (int it = tid; it < ((nitems + WG_BLOCK - 1) & ~(WG_BLOCK - 1)); it += WG_BLOCK) {
            const int tl = it >> p.S_shift;
            const int s = it & (p.S_pad - 1);
            const bool live = (it < nitems) && (s < p.S);
            float acc = 0.f, tia_max = 0.f;
            if (live) {
                const int t = t0 + tl;
                const float4* __restrict__ pr = pair + tl * N;
                const float ys = (float)T[t].yr + rdy[s] * T[t].cg;
                const float dz = rdz[s];
                const float dz2 = dz * dz;
                for (int s2 = 0; s2 < N; ++s2) {
                    const float4 pp = pr[s2];
                    tia_max = fmaxf(tia_max, pp.w);
                    if (pp.z != 0.f) {
                        const float dy = ys - pp.x;
                        acc += pp.z * __expf(-(dy * dy + dz2) * pp.y);
                    }
                }
            }
            for (int o = p.S_pad >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (live && s == 0) {
                TurbLds& q = T[t0 + tl];
                q.u = ws_f - acc * p.inv_S;
                q.v = 0.f;
                q.w = 0.f;
                q.ti = __builtin_amdgcn_sqrtf(ti_f * ti_f + tia_max * tia_max);
            }
        }
        __syncthreads();
    }

    // (5) power / thrust with the current yaw
    if (tid < N) {
        for (int t = tid; t < N; t += WG_BLOCK) {
            TurbLds& q = T[t];
            const float wsn = fmaxf(q.u * q.cg + q.v * q.sg, 0.0f);
            q.pow = tab_lookup(tabp, p, wsn);
            q.ct = tab_lookup(tabct, p, wsn) * q.cg * q.cg;
        }
    }
    __syncthreads();
}

// replay mode (test hook): consume one scripted row instead of the physics
__device__ __forceinline__ void script_step(const FlowP& p, const FlowPtrs& d, TurbLds* T, int e, int farm,
                                            int& cursor, double& time) {
    ++cursor;
    time += p.dt_d;
    const int row = cursor < p.script_rows ? cursor : p.script_rows - 1;
    const size_t base = (((size_t)farm * p.script_rows + row) * p.B + e) * p.N;
    for (int t = threadIdx.x; t < p.N; t += WG_BLOCK) {
        T[t].u = d.script_uvw[(base + t) * 3 + 0];
        T[t].v = d.script_uvw[(base + t) * 3 + 1];
        T[t].w = d.script_uvw[(base + t) * 3 + 2];
        T[t].pow = d.script_power[base + t];
    }
    __syncthreads();
}

template <bool REPLAY, bool NOISE>
__global__ void __launch_bounds__(WG_BLOCK, WG_FLOW_WAVES)
k_flow(const FlowP p, const FlowPtrs d, const int mode, const float* __restrict__ actions,
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
    int dev_rem = slot.dev_remaining, fill_rem = slot.fill_remaining;
    const bool ready = (dev_rem == 0 && fill_rem == 0);
    int budget = 0;
    const bool live_step = (mode == WG_MODE_STEP) && is_live;
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

    // LDS carve: pair[target_chunk * N] | T[N] | tabp[n_tab] | tabct[n_tab] | rdy[S] | rdz[S]
    float4* pair = reinterpret_cast<float4*>(smem);
    TurbLds* T = reinterpret_cast<TurbLds*>(smem + p.lds_off_turb);
    float* tabp = reinterpret_cast<float*>(smem + p.lds_off_tab);
    float* tabct = tabp + p.n_tab;
    float* rdy = tabct + p.n_tab;
    float* rdz = rdy + p.S;

    WgCtx& cx = d.ctx[ctx_id];
    const double ws = cx.ws;
    const float ti_f = (float)cx.ti;
    const float wd_env = (float)cx.wd;
    const size_t tb = (size_t)slot_id * N;
    const size_t pbase = (size_t)slot_id * p.NP;
    for (int t = tid; t < N; t += WG_BLOCK) {
        TurbLds& q = T[t];
        q.xr = d.xr[(size_t)ctx_id * N + t];
        q.yr = d.yr[(size_t)ctx_id * N + t];
        q.yaw = d.yaw[tb + t]; q.u = d.u[tb + t]; q.v = d.v[tb + t]; q.w = d.w[tb + t];
        q.ti = d.ti_loc[tb + t]; q.pow = d.power[tb + t]; q.ct = d.ct[tb + t];
        q.sws = 0.f; q.swd = 0.f; q.syaw = 0.f; q.sp = 0.f;
    }
    for (int i = tid; i < p.n_tab; i += WG_BLOCK) { tabp[i] = d.tab_power[i]; tabct[i] = d.tab_ct[i]; }
    for (int i = tid; i < p.S; i += WG_BLOCK) { rdy[i] = d.rotor_dy[i]; rdz[i] = d.rotor_dz[i]; }
    __syncthreads();

    SlotRegs sr{slot.s_off, slot.time, slot.head, slot.n_valid};
    int cursor = slot.cursor;
    const float ti_pow = __powf(ti_f, p.tic);

    // WindFarmEnv._adjust_yaws (Wind_Farm_Env.py:822-864), float32 like numpy evaluates it on a float32 action
    if (live_step && farm == 0) {
        for (int t = tid; t < N; t += WG_BLOCK) {
            float yaw = T[t].yaw;
            d.old_yaw[(size_t)e * N + t] = yaw;          // :932
            const float a = actions[(size_t)e * N + t];
            if (p.action_method == WG_ACT_YAW) {
                yaw = fminf(fmaxf(yaw + a * p.yaw_step, p.yaw_min), p.yaw_max);
            } else {
                float tf = a + 1.0f;
                tf = tf / 2.0f;
                tf = tf * (p.yaw_max - p.yaw_min);
                tf = tf + p.yaw_min;
                const float ny = fminf(fmaxf(tf, yaw - p.yaw_step), yaw + p.yaw_step);
                yaw = fminf(fmaxf(ny, p.yaw_min), p.yaw_max);
            }
            T[t].yaw = yaw;
        }
        __syncthreads();
    }

    // live:   one env step = K sub-steps with measurement (Wind_Farm_Env.py:932-979)
    // else:   background development of a not-yet-live episode: flow-development steps (fs.run), then
    //         window-fill env steps (Wind_Farm_Env.py:722-796)
    int sub = 0, n_flow = 0;
    float base_acc = 0.f;
    for (;;) {
        if (!live_step && sub == 0 && (budget <= 0 || (dev_rem == 0 && fill_rem == 0))) break;
        const bool is_dev = !live_step && dev_rem > 0;
        if (live_step && farm == 1) {
            // BasicControllers.local_yaw_controller / global_yaw_controller (BasicControllers.py:10-73)
            for (int t = tid; t < N; t += WG_BLOCK) {
                float yaw = T[t].yaw;
                if (p.base_controller == WG_CTRL_LOCAL) {
                    const float wdir = atanf(T[t].v / T[t].u) * WG_RAD2DEG_F;
                    const float off = wdir - yaw;
                    const float sgn = (float)((off > 0.f) - (off < 0.f));
                    yaw = yaw + sgn * fminf(fabsf(off), p.yaw_step);
                } else {
                    const float sgn = (float)((yaw > 0.f) - (yaw < 0.f));
                    yaw = yaw - sgn * fminf(fabsf(yaw), p.yaw_step);
                }
                T[t].yaw = yaw;
            }
            __syncthreads();
        }
        if (REPLAY && !is_dev) script_step(p, d, T, e, farm, cursor, sr.time);
        else { flow_step(p, d, T, tabp, tabct, rdy, rdz, pair, pbase, ws, ti_f, ti_pow, sr); ++n_flow; }
        --budget;
        if (is_dev) { --dev_rem; continue; }
        if (farm == 0) {
            // WindFarmEnv._take_measurements (Wind_Farm_Env.py:480-495), accumulated over the k sub-steps
            for (int t = tid; t < N; t += WG_BLOCK) {
                TurbLds& q = T[t];
                const float wsm = __builtin_amdgcn_sqrtf(q.u * q.u + q.v * q.v + q.w * q.w);
                const float wdm = atanf(q.v / q.u) * WG_RAD2DEG_F + wd_env;
                d.cur_ws[(size_t)ctx_id * N + t] = wsm;
                d.cur_wd[(size_t)ctx_id * N + t] = wdm;
                q.sws += wsm; q.swd += wdm; q.syaw += q.yaw; q.sp += q.pow;
            }
            __syncthreads();
        } else if (tid == 0) {
            float tot = 0.f;
            for (int t = 0; t < N; ++t) tot += T[t].pow;
            base_acc += tot;
        }
        if (++sub < p.K) continue;
        sub = 0;
        if (farm == 0) {
            // farm_mes.add_measurements (MesClass.py:568-591): noise, ring push, farm-level mean/mean/sum
            const int n_pushed = cx.n_pushed;
            const float inv_k = 1.0f / (float)p.K;
            float* __restrict__ rbase = d.ring + (size_t)ctx_id * p.ring_stride;
            for (int t = tid; t < N; t += WG_BLOCK) {
                TurbLds& q = T[t];
                float val[WG_N_CH] = {q.sws, q.swd, q.syaw, q.sp};
                if (p.K != 1) {
#pragma unroll
                    for (int ch = 0; ch < WG_N_CH; ++ch) val[ch] *= inv_k;
                }
                if (NOISE) {
#pragma unroll
                    for (int ch = 0; ch < WG_N_CH; ++ch)
                        if (p.noise_sigma[ch] != 0.f)
                            val[ch] += p.noise_sigma[ch] * wg_noise_normal(env.noise_key, (uint32_t)n_pushed, (uint32_t)t,
                                                                           (uint32_t)ch, (uint32_t)cx.episode_tag);
                }
#pragma unroll
                for (int ch = 0; ch < WG_N_CH; ++ch) {
                    const int H = p.hlen[ch];
                    rbase[p.ring_off[ch] + t * H + (n_pushed % H)] = val[ch];
                }
                q.sws = val[0]; q.swd = val[1]; q.sp = val[3];
            }
            __syncthreads();
            if (tid == 0) {
                float sws = 0.f, swd = 0.f, tot = 0.f;
                for (int t = 0; t < N; ++t) { sws += T[t].sws; swd += T[t].swd; tot += T[t].sp; }
                float* fbase = d.fring + (size_t)ctx_id * p.fring_stride;
                fbase[p.fring_off[WG_CH_WS] + n_pushed % p.hlen[WG_CH_WS]] = sws * p.inv_N;
                fbase[p.fring_off[WG_CH_WD] + n_pushed % p.hlen[WG_CH_WD]] = swd * p.inv_N;
                fbase[p.fring_off[WG_CH_POWER] + n_pushed % p.hlen[WG_CH_POWER]] = tot;
                cx.n_pushed = n_pushed + 1;
                if (live_step) d.step_farm_pow[e] = tot;
                else {
                    d.pend_farm[(size_t)ctx_id * p.power_avg + cx.pend_farm_n % p.power_avg] = tot;
                    cx.pend_farm_n += 1;
                }
            }
            __syncthreads();
            for (int t = tid; t < N; t += WG_BLOCK) { T[t].sws = 0.f; T[t].swd = 0.f; T[t].syaw = 0.f; T[t].sp = 0.f; }
            __syncthreads();
        } else if (tid == 0) {
            const float bp = p.K == 1 ? base_acc : base_acc / (float)p.K;
            if (live_step) d.step_base_pow[e] = bp;
            else {
                d.pend_base[(size_t)ctx_id * p.power_avg + cx.pend_base_n % p.power_avg] = bp;
                cx.pend_base_n += 1;
            }
            base_acc = 0.f;
        }
        if (live_step) break;
        --fill_rem;
    }

    // write the slot back
    for (int t = tid; t < N; t += WG_BLOCK) {
        const TurbLds& q = T[t];
        d.yaw[tb + t] = q.yaw; d.u[tb + t] = q.u; d.v[tb + t] = q.v; d.w[tb + t] = q.w;
        d.ti_loc[tb + t] = q.ti; d.power[tb + t] = q.pow; d.ct[tb + t] = q.ct;
    }
    if (tid == 0) {
        slot.head = sr.head; slot.n_valid = sr.n_valid; slot.s_off = sr.s_off; slot.time = sr.time;
        slot.cursor = cursor;
        slot.dev_remaining = dev_rem; slot.fill_remaining = fill_rem;
        if (n_flow) atomicAdd(d.flow_steps, (unsigned long long)n_flow);
    }
}

extern "C" void wg_launch_flow(const FlowP* p, const FlowPtrs* d, int mode, const float* actions,
                               const uint8_t* mask, int chunk, hipStream_t st) {
    const int grid = p->B * 2 * p->F;
    const size_t lds = p->lds_bytes;
    const bool replay = d->script_uvw != nullptr, noise = p->noise != 0;
    if (replay) {
        if (noise) hipLaunchKernelGGL((k_flow<true, true>), dim3(grid), dim3(WG_BLOCK), lds, st, *p, *d, mode, actions, mask, chunk);
        else hipLaunchKernelGGL((k_flow<true, false>), dim3(grid), dim3(WG_BLOCK), lds, st, *p, *d, mode, actions, mask, chunk);
    } else {
        if (noise) hipLaunchKernelGGL((k_flow<false, true>), dim3(grid), dim3(WG_BLOCK), lds, st, *p, *d, mode, actions, mask, chunk);
        else hipLaunchKernelGGL((k_flow<false, false>), dim3(grid), dim3(WG_BLOCK), lds, st, *p, *d, mode, actions, mask, chunk);
    }
}
