// Dispatch-floor probe: how long does a launch of many tiny workgroups take on this GPU when each workgroup does (almost)
// nothing?  k_flow launches 16 384 single-wave workgroups per step (cfg2); its "-DWG_ABLATE=8" build (state loads, then
// return) lasts 21 us.  This probe separates what that floor is made of: workgroup count vs wave count, the size of the
// by-value parameter block (kernarg -> SGPR preload), dynamic LDS, registers per wave (launch_bounds), one dependent
// global load.
// Build: hipcc --offload-arch=gfx950 -O3 -o dispatch_probe dispatch_probe.hip ; run: ./dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Big { int v[190]; };          // ~ sizeof(FlowP) + sizeof(FlowPtrs) in dwords

__global__ void k_empty(int* out) { if (out == nullptr) asm volatile("s_nop 0"); }
__global__ void k_big(Big b, int* out) { if (b.v[189] == 12345) out[0] = b.v[7]; }
// one global load per lane, result decides an (untaken) store: the wave lives one memory round trip
__global__ void k_load1(const int* __restrict__ in, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (in[i] == 12345) out[0] = 1;
}
// two dependent loads
__global__ void k_load2(const int* __restrict__ in, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = in[i];
    if (in[(j & 1023) + (i & ~1023)] == 12345) out[0] = 1;
}
template <int MINW>
__global__ void __launch_bounds__(64, MINW) k_regs(const int* __restrict__ in, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (in[i] == 12345) out[0] = 1;
}
__global__ void k_lds(const int* __restrict__ in, int* out) {
    extern __shared__ int sm[];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    sm[threadIdx.x] = in[i];
    if (sm[threadIdx.x] == 12345) out[0] = 1;
}

// the same load, but the wave owns scratch memory (a dynamically indexed private array)
__global__ void k_scratch(const int* __restrict__ in, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    volatile int priv[48];
    const int v = in[i];
    if (v == 12345) {
        for (int k = 0; k < 48; ++k) priv[k] = v + k;
        out[0] = priv[v & 31];
    }
}
__device__ __attribute__((noinline)) int rare_path(const int* in, int i) {
    int acc = 0;
    for (int k = 0; k < 4; ++k) acc += in[(acc + i + k) & 1023];      // dependent loads
    return acc;
}
// no scratch-resident data, but an out-of-line call on a rare path (what k_flow's episode set-up is)
__global__ void k_call(const int* __restrict__ in, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = in[i];
    if (v == 12345) out[0] = rare_path(in, i);
}

template <class F>
static float time_us(F&& launch, int reps = 200) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms * 1e3f / reps;
}

int main() {
    int *in, *out;
    const int total = 16384 * 64 * 4;
    hipMalloc(&in, sizeof(int) * total);
    hipMalloc(&out, 64);
    hipMemset(in, 0, sizeof(int) * total);
    Big big;
    for (int i = 0; i < 190; ++i) big.v[i] = i;
    hipDeviceSynchronize();
    printf("# back-to-back launches on the null stream, us per launch (includes the inter-kernel gap)\n");
    for (int wgs : {1024, 4096, 16384, 65536}) {
        printf("empty      %6d x  64 : %7.2f us\n", wgs, time_us([&] { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(64), 0, 0, out); }));
    }
    for (int wgs : {1024, 4096, 16384}) {
        printf("empty      %6d x 256 : %7.2f us\n", wgs, time_us([&] { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, 0, out); }));
    }
    printf("empty      %6d x 512 : %7.2f us\n", 2048, time_us([&] { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(512), 0, 0, out); }));
    printf("empty      %6d x1024 : %7.2f us\n", 1024, time_us([&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(1024), 0, 0, out); }));
    printf("big args    16384 x  64 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_big, dim3(16384), dim3(64), 0, 0, big, out); }));
    printf("big args     4096 x 256 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_big, dim3(4096), dim3(256), 0, 0, big, out); }));
    printf("load1       16384 x  64 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_load1, dim3(16384), dim3(64), 0, 0, in, out); }));
    printf("load1        4096 x 256 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_load1, dim3(4096), dim3(256), 0, 0, in, out); }));
    printf("load1        5120 x  64 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_load1, dim3(5120), dim3(64), 0, 0, in, out); }));
    printf("load2       16384 x  64 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_load2, dim3(16384), dim3(64), 0, 0, in, out); }));
    printf("load2        4096 x 256 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_load2, dim3(4096), dim3(256), 0, 0, in, out); }));
    printf("load1 occ8  16384 x  64 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_regs<8>, dim3(16384), dim3(64), 0, 0, in, out); }));
    printf("load1 occ5  16384 x  64 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_regs<5>, dim3(16384), dim3(64), 0, 0, in, out); }));
    printf("load1 occ2  16384 x  64 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_regs<2>, dim3(16384), dim3(64), 0, 0, in, out); }));
    printf("scratch     16384 x  64 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_scratch, dim3(16384), dim3(64), 0, 0, in, out); }));
    printf("scratch      4096 x 256 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_scratch, dim3(4096), dim3(256), 0, 0, in, out); }));
    printf("rare call   16384 x  64 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_call, dim3(16384), dim3(64), 0, 0, in, out); }));
    // alternating with a scratch-free kernel (as k_flow / k_glue do)
    printf("scratch+empty alternating (sum of both) : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_scratch, dim3(16384), dim3(64), 0, 0, in, out); hipLaunchKernelGGL(k_load1, dim3(4096), dim3(64), 0, 0, in, out); }));
    printf("load1+load1   alternating (sum of both) : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_load1, dim3(16384), dim3(64), 0, 0, in, out); hipLaunchKernelGGL(k_load1, dim3(4096), dim3(64), 0, 0, in, out); }));
    for (int lds : {1024, 6144, 8192, 16384}) {
        printf("lds %5d   16384 x  64 : %7.2f us\n", lds, time_us([&] { hipLaunchKernelGGL(k_lds, dim3(16384), dim3(64), lds, 0, in, out); }));
    }
    printf("lds 24576    4096 x 256 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_lds, dim3(4096), dim3(256), 24576, 0, in, out); }));
    // the gap alone: 1 workgroup
    printf("empty           1 x  64 : %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, out); }));
    return 0;
}
