// wg_steady.hip — k_steady: steady-state farm power for a batch of (wind condition, yaw vector) cases on the MI355X.
//
// The inner loop of the reference's PyWakeAgent (WindGym/Agents/PyWakeAgent.py:144-288, yaw_optimizer_srf_vect: every
// refine step evaluates the farm power of yaw_n candidate yaw vectors per wind condition) — SURVEY.md §8 row f4.  The
// Serial-Refine loop itself stays on the host (windgym_amd/steady.py); each of its steps is ONE launch of this kernel over
// [conditions x candidates] cases.
//   model 0: the steady state of the env's own flow model M0 — the deficit evaluation of k_flow /
//            k_windspeed with the chain replaced by its fixed point: every source's frozen record is the record it emits
//            now, the wake centre is the integral of the Hill-vortex deflection speed along the chain;
//   model 1: the reference agent's py_wake model restated from the publications — Blondel & Cathelain (2020)
//            super-Gaussian deficit at the rotor centre + Jimenez deflection (steady.blondel_jimenez_power).
// One single-wave workgroup per case: turbines are visited upstream -> downstream (rank by flow-frame x); lane s evaluates
// source s for the current target (its S rotor points, its deflection quadrature), wave reductions give the target's
// inflow.  fp32; the fp64 torch restatement of windgym_amd/steady.py is the oracle (tests/test_steady_kernel.py).
#include <hip/hip_runtime.h>

#include "wg_device.h"
#include "wg_steady.h"

__device__ __forceinline__ float st_interp(const float* __restrict__ xs, const float* __restrict__ ys, const int n, const float x) {
    if (!(x >= xs[0]) || x > xs[n - 1]) return 0.f;        // 0 outside the table (steady._interp)
    int i = 0;
    while (i < n - 2 && xs[i + 1] <= x) ++i;
    const float f = (x - xs[i]) / (xs[i + 1] - xs[i]);
    return ys[i] + f * (ys[i + 1] - ys[i]);
}
__device__ __forceinline__ float st_cfrac(const float ct, const float sp) {
    const float m = fminf(1.0f / (8.0f * sp * sp), 1.0f);
    return 1.0f - sqrtf(fmaxf(1.0f - ct * m, 0.0f));
}

__global__ void __launch_bounds__(64)
k_steady(const SteadyP p, const float* __restrict__ ws_in, const float* __restrict__ wd_in, const float* __restrict__ ti_in,
         const float* __restrict__ yaw_in, float* __restrict__ power_out) {
    extern __shared__ float st_lds[];
    const int c = blockIdx.x, lane = threadIdx.x, N = p.N;
    if (c >= p.n_cases) return;
    float* xr = st_lds;            // [N] each
    float* yr = xr + N;
    float* cg = yr + N;
    float* sg = cg + N;
    float* u = sg + N;
    float* til = u + N;
    float* ct = til + N;
    float* hv = ct + N;
    int* by_rank = reinterpret_cast<int*>(hv + N);
    const float ws = ws_in[c], ti = ti_in[c];
    const double th = (270.0 - (double)wd_in[c]) * (WG_PI_D / 180.0);
    const double cth = cos(th), sth = sin(th);
    for (int t = lane; t < N; t += 64) {
        const double dx = p.x_pos[t] - p.cx0, dy = p.y_pos[t] - p.cy0;
        xr[t] = (float)(p.cx0 + dx * cth + dy * sth - p.cx0);      // (relative to the farm centre: fp32 keeps the metres)
        yr[t] = (float)(p.cy0 - dx * sth + dy * cth - p.cy0);
        const float g = yaw_in[(size_t)c * N + t] * WG_DEG2RAD_F;
        cg[t] = cosf(g); sg[t] = sinf(g);
        u[t] = ws; til[t] = ti; ct[t] = 0.f; hv[t] = 0.f;
    }
    __syncthreads();
    // upstream -> downstream: rank of turbine t = number of turbines ahead of it (ties by index, like a stable argsort)
    for (int t = lane; t < N; t += 64) {
        int r = 0;
        for (int o = 0; o < N; ++o) r += (xr[o] < xr[t] || (xr[o] == xr[t] && o < t)) ? 1 : 0;
        by_rank[r] = t;
    }
    __syncthreads();
    const float inv_D = 1.0f / p.D;
    for (int pos = 0; pos < N; ++pos) {
        const int t = by_rank[pos];
        const float xt = xr[t], yt = yr[t], cgt = cg[t];
        float dsum = 0.f, tia_max = 0.f;
        for (int s = lane; s < N; s += 64) {
            const float dx = xt - xr[s];
            if (!(dx > 1e-9f)) continue;
            const float cts = ct[s];
            if (p.model == 0) {
                const float k = p.ka * til[s] + p.kb;
                const float q = sqrtf(1.0f - cts);
                const float eps = p.eps0 * sqrtf(0.5f * (1.0f + q) / q);
                const float xd = dx * inv_D;
                const float sp = k * xd + eps, sig = sp * p.D;
                // wake-centre deflection: (hv / U) * integral_0^dx C(x') dx' (trapezoid, n_quad points)
                const int Q = p.n_quad;
                const float h = dx / (float)(Q - 1);
                float integ = 0.f;
                for (int i = 0; i < Q; ++i) {
                    const float cq = st_cfrac(cts, k * ((float)i * h * inv_D) + eps);
                    integ += (i == 0 || i == Q - 1) ? 0.5f * cq : cq;
                }
                integ *= h;
                const float yc = yr[s] + hv[s] / ws * integ;
                const float rc2 = (yt - yc) * (yt - yc);
                const float rcut = p.R + 5.0f * sig;
                if (rc2 > rcut * rcut) continue;
                const float amp = u[s] * st_cfrac(cts, sp);
                const float inv2s2 = 1.0f / (2.0f * sig * sig);
                float acc = 0.f;
                for (int i = 0; i < p.S; ++i) {
                    const float dy = yt + p.rotor_dy[i] * cgt - yc, dz = p.rotor_dz[i];
                    acc += amp * expf(-(dy * dy + dz * dz) * inv2s2);
                }
                dsum += acc;
                const float ind = 0.5f * (1.0f - sqrtf(1.0f - cts));
                tia_max = fmaxf(tia_max, p.tia * powf(ind, p.tib) * powf(ti, p.tic) * powf(fmaxf(xd, 1.0f), p.tid) * expf(-rc2 * inv2s2));
            } else {
                // Blondel & Cathelain (2020) at the rotor centre, Jimenez deflection (py_wake defaults: steady.py)
                const float xd = fmaxf(dx, 1e-9f) * inv_D;
                const float q = sqrtf(1.0f - fminf(cts, 0.999f));
                const float beta = 0.5f * (1.0f + q) / q;
                const float sigma = (0.17f * ti + 0.005f) * xd + 0.2f * sqrtf(beta);
                const float n = 3.11f * expf(-0.68f * xd) + 2.41f;
                const float in2 = 2.0f / n;
                const float C = exp2f(in2 - 1.0f) - sqrtf(fmaxf(exp2f(2.0f * in2 - 2.0f) - n * cts / (16.0f * tgammaf(in2) * powf(sigma, 2.0f * in2)), 0.0f));
                const int Q = p.n_quad;
                const float a0 = cg[s] * cg[s] * sg[s] * cts * 0.5f;
                float defl = 0.f, x_prev = 0.f, f_prev = sinf(a0);
                for (int i = 1; i < Q; ++i) {      // points clustered towards the rotor: (10^(1.1 i / (Q - 1)) - 1) / (10^1.1 - 1)
                    const float s01 = (exp10f(1.1f * (float)i / (float)(Q - 1)) - 1.0f) * (1.0f / (12.589254117941675f - 1.0f));
                    const float xq = dx * s01;
                    const float den = 1.0f + 0.1f * xq * inv_D;
                    const float f = sinf(a0 / (den * den));
                    defl += 0.5f * (f + f_prev) * (xq - x_prev);
                    x_prev = xq; f_prev = f;
                }
                const float r = fabsf(yt - (yr[s] - defl)) * inv_D;
                dsum += ws * C * expf(-powf(r, n) / (2.0f * sigma * sigma));
            }
        }
        dsum = wg_wave_sum(dsum);
        tia_max = wg_wave_max(tia_max);
        if (lane == 0) {
            if (p.model == 0) {
                const float ut = ws - dsum / (float)p.S;
                u[t] = ut;
                til[t] = sqrtf(ti * ti + tia_max * tia_max);
                ct[t] = fminf(fmaxf(st_interp(p.tab_ws, p.tab_ct, p.n_tab, fmaxf(ut * cgt, 0.f)) * cgt * cgt, 0.f), 0.96f);
                hv[t] = -p.hill * sg[t] * ut;
            } else {
                const float ut = ws - dsum;
                u[t] = ut;
                ct[t] = fminf(fmaxf(st_interp(p.tab_ws, p.tab_ct, p.n_tab, fmaxf(ut * cgt, 0.f)) * cgt * cgt, 0.f), 0.999f);
            }
        }
        __syncthreads();
    }
    for (int t = lane; t < N; t += 64)
        power_out[(size_t)c * N + t] = st_interp(p.tab_ws, p.tab_power, p.n_tab, fmaxf(u[t] * cg[t], 0.f));
}

extern "C" void wg_launch_steady(const void* sp, const float* ws, const float* wd, const float* ti, const float* yaw, float* power,
                                 hipStream_t st) {
    const SteadyP* p = (const SteadyP*)sp;
    const size_t lds = (size_t)p->N * (8 * sizeof(float) + sizeof(int));
    hipLaunchKernelGGL(k_steady, dim3(p->n_cases), dim3(64), lds, st, *p, ws, wd, ti, yaw, power);
}
