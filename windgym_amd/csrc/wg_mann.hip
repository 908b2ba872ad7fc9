// wg_mann.hip — Mann (1994 / 1998) spectral-tensor turbulence boxes generated on the MI355X: wg_generate_mann_box.
//
// Replaces MannTurbulenceField.generate(alphaepsilon, L, Gamma, Nxyz, dxyz, seed) of hipersim / dynamiks for
// turbtype "MannFixed" / "MannGenerate" (Wind_Farm_Env.py:624-637, :649-658) — SURVEY.md §8 row f2.  The box is built
// where it is used: one kernel evaluates the sheared spectral tensor times complex white noise per wave-number cell,
// hipFFT transforms it back (three in-place C2C inverse transforms, one component at a time: one complex work array),
// a second kernel keeps the real part, and the box is normalised to unit standard deviation of u.  Nothing crosses PCIe.
//
// Algorithm (published; the restatement in windgym_amd/mann.py and the numpy pin of tests/test_mann_generator.py follow
// the same formulas): isotropic von Karman field with random complex Gaussian amplitudes dZ_iso = sqrt(E(k0) / 4 pi) / k0^2
// (k0 x n), distorted by rapid-distortion shear with the eddy lifetime beta(|k| L) = Gamma (kL)^(-2/3) /
// sqrt(2F1(1/3, 17/6; 4/3; -(kL)^-2)) (Mann 1998, eqs. 3.16-3.18).
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/windgym_hip.h"

extern "C" int wg_set_last_error_(int code, const char* msg);      // wg_api.hip
#define MCHK(x)                                                                                        \
    do {                                                                                               \
        hipError_t _e = (x);                                                                           \
        if (_e != hipSuccess) { rc = wg_set_last_error_(WG_ERR_HIP, (std::string(#x) + ": " + hipGetErrorString(_e)).c_str()); goto done; } \
    } while (0)

// ---- eddy lifetime ----------------------------------------------------------------------------------------------------
// 2F1(1/3, 17/6; 4/3; -x) by its Euler integral (the parameters are symmetric: with a = 17/6, b = 1/3, c = 4/3 the
// weight (1 - t)^(c - b - 1) is 1): (1/3) int_0^1 t^(-2/3) (1 + x t)^(-17/6) dt = int_0^1 (1 + x s^3)^(-17/6) ds with
// t = s^3, = G(U) / U with U = x^(1/3), G(U) = int_0^U (1 + u^3)^(-17/6) du — a smooth, rapidly decaying integrand
// (panelled 8-point Gauss-Legendre in double precision, accumulated along the table).
static double mann_g(double u) { return std::pow(1.0 + u * u * u, -17.0 / 6.0); }
// 8-point Gauss-Legendre on [a, b]
static double gl8(double a, double b) {
    static const double x[4] = {0.1834346424956498049, 0.5255324099163289858, 0.7966664774136267396, 0.9602898564975362317};
    static const double w[4] = {0.3626837833783619830, 0.3137066458778872873, 0.2223810344533744706, 0.1012285362903762591};
    const double c = 0.5 * (a + b), h = 0.5 * (b - a);
    double s = 0.0;
    for (int k = 0; k < 4; ++k) s += w[k] * (mann_g(c - h * x[k]) + mann_g(c + h * x[k]));
    return s * h;
}
// int_a^b g with panels no wider than 1/8 (g varies on the scale of 1 around u ~ 1 and is flat or negligible elsewhere)
static double mann_int(double a, double b) {
    const int np = (int)std::ceil((b - a) * 8.0) < 1 ? 1 : (int)std::ceil((b - a) * 8.0);
    double s = 0.0;
    for (int k = 0; k < np; ++k) s += gl8(a + (b - a) * k / np, a + (b - a) * (k + 1) / np);
    return s;
}

extern "C" int wg_mann_beta_table(double Gamma, int n, double log10_lo, double log10_hi, double* beta_out) {
    if (!beta_out || n < 2 || !(log10_hi > log10_lo)) return wg_set_last_error_(WG_ERR_INVALID, "wg_mann_beta_table: bad arguments");
    // G(U) is accumulated along the table: U = (kL)^(-2/3) grows as kL falls, so the table is walked from its last entry
    // (largest kL, smallest U) to its first; the integrand is negligible beyond u = 64 (g < 1e-15)
    double G = 0.0, U_prev = 0.0;
    for (int i = n - 1; i >= 0; --i) {
        const double kl = std::pow(10.0, log10_lo + (log10_hi - log10_lo) * i / (n - 1));
        const double U = std::pow(kl, -2.0 / 3.0);
        const double b = U < 64.0 ? U : 64.0;
        if (b > U_prev) { G += mann_int(U_prev, b); U_prev = b; }
        const double F = G / U;                               // 2F1(1/3, 17/6; 4/3; -(kL)^-2)
        beta_out[i] = Gamma == 0.0 ? 0.0 : Gamma * std::pow(kl, -2.0 / 3.0) / std::sqrt(F);
    }
    return 0;
}

// ---- kernels ------------------------------------------------------------------------------------------------------------
#define WG_MANN_TAB 4096
#define WG_MANN_LOG_LO (-6.0f)
#define WG_MANN_LOG_HI 6.0f

struct MannP {
    int nx, ny, nz;
    float dk1, dk2, dk3;          // 2 pi / (N d)
    float L, ae_L53;              // alphaepsilon L^(5/3)
    uint32_t seed_lo, seed_hi;
};

// complex standard normal (E|n|^2 = 1) of (component c, cell idx): Philox4x32-10 keyed by the seed, Box-Muller on both
// branches.  tests/test_mann_generator.py feeds the numpy pin the same numbers through wgo_mann_noise (oracle/rng_api.c).
__device__ __forceinline__ float2 mann_noise(uint32_t seed_lo, uint32_t seed_hi, uint64_t idx, uint32_t c) {
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = c, c3 = 0x4d414e4eu;      // "MANN"
    uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float u1 = ((float)(c0 >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(c1 >> 8) * (1.0f / 16777216.0f);
    // |n|^2 = -ln u1: E = 1 (the real and the imaginary part have variance 1/2 each)
    const float r = sqrtf(-logf(u1));
    float sn, cs;
    sincosf(6.2831853071795864f * u2, &sn, &cs);
    return make_float2(r * cs, r * sn);
}

__device__ __forceinline__ float fftfreq_k(int i, int n, float dk) { return (float)(i < (n + 1) / 2 ? i : i - n) * dk; }

// dZ_comp(k) for every wave-number cell; z fastest (hipFFT's row-major [nx][ny][nz])
__global__ void __launch_bounds__(256)
k_mann_dz(const MannP p, const int comp, const float* __restrict__ beta_tab, const float2* __restrict__ noise,
          float2* __restrict__ out) {
    const size_t cells = (size_t)p.nx * p.ny * p.nz;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cells) return;
    const int k = (int)(idx % p.nz), j = (int)((idx / p.nz) % p.ny), i = (int)(idx / ((size_t)p.nz * p.ny));
    const float k1 = fftfreq_k(i, p.nx, p.dk1), k2 = fftfreq_k(j, p.ny, p.dk2), k3 = fftfreq_k(k, p.nz, p.dk3);
    const float kk2 = k1 * k1 + k2 * k2 + k3 * k3;
    if (kk2 == 0.f) { out[idx] = make_float2(0.f, 0.f); return; }          // the mean is removed (dZ[0, 0, 0] = 0)
    const float kk = sqrtf(kk2);
    // beta(|k| L): log-spaced table, linear interpolation in log10(kL)
    float lk = log10f(fminf(fmaxf(kk * p.L, 1e-6f), 1e6f));
    float pos = (lk - WG_MANN_LOG_LO) * ((float)(WG_MANN_TAB - 1) / (WG_MANN_LOG_HI - WG_MANN_LOG_LO));
    int i0 = (int)floorf(pos);
    i0 = i0 < 0 ? 0 : (i0 > WG_MANN_TAB - 2 ? WG_MANN_TAB - 2 : i0);
    const float w = pos - (float)i0;
    const float beta = beta_tab[i0] * (1.0f - w) + beta_tab[i0 + 1] * w;
    const float k30 = k3 + beta * k1;
    const float k02 = fmaxf(k1 * k1 + k2 * k2 + k30 * k30, 1e-24f);
    const float k0 = sqrtf(k02);
    // von Karman energy spectrum at the undistorted wave number
    const float k0L = k0 * p.L;
    const float E0 = p.ae_L53 * (k0L * k0L) * (k0L * k0L) * powf(1.0f + k0L * k0L, -17.0f / 6.0f);
    const float amp = sqrtf(E0 * (1.0f / (4.0f * 3.14159265358979f))) / k02;
    // white noise of the three components
    float2 n0, n1, n2;
    if (noise) {
        n0 = noise[idx]; n1 = noise[cells + idx]; n2 = noise[2 * cells + idx];
    } else {
        n0 = mann_noise(p.seed_lo, p.seed_hi, idx, 0u); n1 = mann_noise(p.seed_lo, p.seed_hi, idx, 1u);
        n2 = mann_noise(p.seed_lo, p.seed_hi, idx, 2u);
    }
    // isotropic incompressible field amp (k0 x n); only a3 and the requested component are needed
    const float2 a3 = make_float2(amp * (k1 * n1.x - k2 * n0.x), amp * (k1 * n1.y - k2 * n0.y));
    float2 r;
    if (comp == 2) {
        const float s = k02 / kk2;
        r = make_float2(s * a3.x, s * a3.y);
    } else {
        // rapid-distortion coefficients (Mann 1998, eqs. 3.16-3.18)
        const float k12 = fmaxf(k1 * k1 + k2 * k2, 1e-24f);
        float zeta;
        if (k1 == 0.f) zeta = comp == 0 ? -beta : 0.f;
        else {
            const float C1 = beta * k1 * k1 * (k02 - 2.0f * k30 * k30 + beta * k1 * k30) / (kk2 * k12);
            const float C2 = k2 * k02 / (k12 * sqrtf(k12)) * atan2f(beta * k1 * sqrtf(k12), k02 - k30 * k1 * beta);
            zeta = comp == 0 ? C1 - k2 / k1 * C2 : k2 / k1 * C1 + C2;
        }
        const float2 a = comp == 0 ? make_float2(amp * (k2 * n2.x - k30 * n1.x), amp * (k2 * n2.y - k30 * n1.y))
                                   : make_float2(amp * (k30 * n0.x - k1 * n2.x), amp * (k30 * n0.y - k1 * n2.y));
        r = make_float2(a.x + zeta * a3.x, a.y + zeta * a3.y);
    }
    out[idx] = r;
}

// real part x scale -> one component of the box; per-block partial sums of v and v^2 (double) for the normalisation
__global__ void __launch_bounds__(256)
k_mann_real(const float2* __restrict__ z, float* __restrict__ out, const size_t cells, const float scale,
            double* __restrict__ part) {
    __shared__ double s1[256], s2[256];
    double a = 0.0, b = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (size_t)gridDim.x * blockDim.x) {
        const float v = z[i].x * scale;
        out[i] = v;
        a += (double)v; b += (double)v * (double)v;
    }
    s1[threadIdx.x] = a; s2[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (part && threadIdx.x == 0) { part[2 * blockIdx.x] = s1[0]; part[2 * blockIdx.x + 1] = s2[0]; }
}
__global__ void k_mann_scale(float* __restrict__ x, const size_t n, const float s) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= s;
}

// ---- entry point ------------------------------------------------------------------------------------------------------
extern "C" int wg_generate_mann_box(int device, float* box_dev, int nx, int ny, int nz, double dx, double dy, double dz,
                                    double alphaepsilon, double L, double Gamma, uint64_t seed, const float* noise_dev,
                                    void* stream) {
    if (!box_dev || nx < 2 || ny < 2 || nz < 2 || !(dx > 0) || !(dy > 0) || !(dz > 0) || !(L > 0) || !(alphaepsilon > 0))
        return wg_set_last_error_(WG_ERR_INVALID, "wg_generate_mann_box: null pointer or bad grid / spectrum parameters");
    int rc = 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t cells = (size_t)nx * ny * nz;
    float2* work = nullptr;
    float* tab_dev = nullptr;
    double* part_dev = nullptr;
    hipfftHandle plan = 0;
    bool have_plan = false;
    const int nblk = 1024;
    std::vector<double> tab64(WG_MANN_TAB), part(2 * nblk);
    std::vector<float> tab32(WG_MANN_TAB);
    MannP p;
    double sd = 0.0;
    {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return wg_set_last_error_(WG_ERR_HIP, (std::string("hipSetDevice: ") + hipGetErrorString(e)).c_str());
    }
    wg_mann_beta_table(Gamma, WG_MANN_TAB, WG_MANN_LOG_LO, WG_MANN_LOG_HI, tab64.data());
    for (int i = 0; i < WG_MANN_TAB; ++i) tab32[i] = (float)tab64[i];
    p.nx = nx; p.ny = ny; p.nz = nz;
    p.dk1 = (float)(2.0 * M_PI / (nx * dx)); p.dk2 = (float)(2.0 * M_PI / (ny * dy)); p.dk3 = (float)(2.0 * M_PI / (nz * dz));
    p.L = (float)L; p.ae_L53 = (float)(alphaepsilon * std::pow(L, 5.0 / 3.0));
    p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
    MCHK(hipMalloc(&work, cells * sizeof(float2)));
    MCHK(hipMalloc(&tab_dev, WG_MANN_TAB * sizeof(float)));
    MCHK(hipMalloc(&part_dev, 2 * nblk * sizeof(double)));
    MCHK(hipMemcpyAsync(tab_dev, tab32.data(), WG_MANN_TAB * sizeof(float), hipMemcpyHostToDevice, st));
    if (hipfftPlan3d(&plan, nx, ny, nz, HIPFFT_C2C) != HIPFFT_SUCCESS) { rc = wg_set_last_error_(WG_ERR_HIP, "hipfftPlan3d failed"); goto done; }
    have_plan = true;
    if (hipfftSetStream(plan, st) != HIPFFT_SUCCESS) { rc = wg_set_last_error_(WG_ERR_HIP, "hipfftSetStream failed"); goto done; }
    {
        // np.fft.ifftn(dZ) * N * sqrt(dV): hipFFT's inverse is unnormalised
        const double dV = std::pow(2.0 * M_PI, 3) / (nx * dx * ny * dy * nz * dz);
        const float scale = (float)std::sqrt(dV);
        for (int c = 0; c < 3; ++c) {
            hipLaunchKernelGGL(k_mann_dz, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, p, c, tab_dev,
                               (const float2*)noise_dev, work);
            if (hipfftExecC2C(plan, (hipfftComplex*)work, (hipfftComplex*)work, HIPFFT_BACKWARD) != HIPFFT_SUCCESS) {
                rc = wg_set_last_error_(WG_ERR_HIP, "hipfftExecC2C failed"); goto done;
            }
            hipLaunchKernelGGL(k_mann_real, dim3(nblk), dim3(256), 0, st, work, box_dev + (size_t)c * cells, cells, scale,
                               c == 0 ? part_dev : nullptr);
            if (c == 0) MCHK(hipMemcpyAsync(part.data(), part_dev, 2 * nblk * sizeof(double), hipMemcpyDeviceToHost, st));
        }
        MCHK(hipStreamSynchronize(st));
        MCHK(hipGetLastError());
        double s1 = 0.0, s2 = 0.0;
        for (int b = 0; b < nblk; ++b) { s1 += part[2 * b]; s2 += part[2 * b + 1]; }
        const double mean = s1 / (double)cells;
        sd = std::sqrt(std::fmax(s2 / (double)cells - mean * mean, 0.0));       // population std, like ndarray.std()
        if (!(sd > 0.0)) { rc = wg_set_last_error_(WG_ERR_INVALID, "wg_generate_mann_box: degenerate field (zero variance)"); goto done; }
        hipLaunchKernelGGL(k_mann_scale, dim3(2048), dim3(256), 0, st, box_dev, 3 * cells, (float)(1.0 / sd));
        MCHK(hipStreamSynchronize(st));
    }
done:
    if (have_plan) hipfftDestroy(plan);
    if (work) hipFree(work);
    if (tab_dev) hipFree(tab_dev);
    if (part_dev) hipFree(part_dev);
    return rc;
}
