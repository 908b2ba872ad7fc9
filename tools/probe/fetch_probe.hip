// FETCH_SIZE / WRITE_SIZE unit check on this GPU: stream-read N bytes (coalesced 16 B per lane), scattered 4-byte
// reads one per 128-byte line, and stream-write N bytes.  Build: hipcc --offload-arch=gfx950 -O3 -o fetch_probe fetch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_stream_read(const float4* __restrict__ in, float* __restrict__ out, size_t n4) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345f) out[0] = acc;
}
__global__ void k_line_read(const float* __restrict__ in, float* __restrict__ out, size_t nlines) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < nlines; i += (size_t)gridDim.x * blockDim.x) acc += in[i * 32];     // one float per 128-byte line
    if (acc == 1.2345f) out[0] = acc;
}
__global__ void k_line_read16(const float4* __restrict__ in, float* __restrict__ out, size_t nlines) {   // 16 B per 128-byte line
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < nlines; i += (size_t)gridDim.x * blockDim.x) { const float4 v = in[i * 8]; acc += v.x + v.w; }
    if (acc == 1.2345f) out[0] = acc;
}
__global__ void k_half_read(const float4* __restrict__ in, float* __restrict__ out, size_t nlines) {     // the first 64 of every 128 bytes
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < nlines * 4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = in[(i >> 2) * 8 + (i & 3)]; acc += v.x + v.w; }
    if (acc == 1.2345f) out[0] = acc;
}
// 16 B of a pseudo-RANDOM line per lane (`span` lines of 128 B, a power of two): DRAM pages are not walked in order
__global__ void k_rand_read16(const float4* __restrict__ in, float* __restrict__ out, size_t nlines, size_t span) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < nlines; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const float4 v = in[(h & (span - 1)) * 8];
        acc += v.x + v.w;
    }
    if (acc == 1.2345f) out[0] = acc;
}
// runs of `run` consecutive lines starting at pseudo-random places (run = 2 .. 16: what a chain segment looks like)
__global__ void k_rand_runs(const float4* __restrict__ in, float* __restrict__ out, size_t nlines, size_t span, int run) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < nlines * 8; i += (size_t)gridDim.x * blockDim.x) {      // 16 B per lane, 8 lanes per line
        const size_t line = i >> 3, r = line / run;
        unsigned long long h = (unsigned long long)r * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const float4 v = in[(((h & (span - 1)) & ~(size_t)(run - 1)) + line % run) * 8 + (i & 7)];
        acc += v.x + v.w;
    }
    if (acc == 1.2345f) out[0] = acc;
}
__global__ void k_stream_write(float4* __restrict__ out, size_t n4) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_line_write4(float* __restrict__ out, size_t nlines) {          // 4 bytes per 128-byte line
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < nlines; i += (size_t)gridDim.x * blockDim.x) out[i * 32] = 1.f;
}
__global__ void k_line_write64(float4* __restrict__ out, size_t nlines) {        // 64 of every 128 bytes, 16 B per lane
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < nlines * 4; i += (size_t)gridDim.x * blockDim.x) out[(i >> 2) * 8 + (i & 3)] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main() {
    const size_t bytes = 1ull << 32;      // 4 GiB: 16 x the Infinity Cache
    float4 *a; float* o;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&o, 64);
    (void)hipMemset(a, 0, bytes);
    (void)hipDeviceSynchronize();
    k_stream_read<<<4096, 256>>>(a, o, bytes / 16);
    k_line_read<<<4096, 256>>>((const float*)a, o, bytes / 128);
    k_line_read16<<<4096, 256>>>(a, o, bytes / 128);
    k_half_read<<<4096, 256>>>(a, o, bytes / 128);
    // (durations in the kernel trace: does a narrow read of a line cost the bandwidth of all of its 128 bytes?)
    k_stream_read<<<4096, 256>>>(a, o, bytes / 16);
    k_line_read<<<4096, 256>>>((const float*)a, o, bytes / 128);
    k_rand_read16<<<4096, 256>>>(a, o, bytes / 128, bytes / 128);
    k_rand_read16<<<4096, 256>>>(a, o, bytes / 128, bytes / 128);
    k_rand_runs<<<4096, 256>>>(a, o, bytes / 128, bytes / 128, 2);
    k_rand_runs<<<4096, 256>>>(a, o, bytes / 128, bytes / 128, 4);
    k_rand_runs<<<4096, 256>>>(a, o, bytes / 128, bytes / 128, 16);
    k_stream_write<<<4096, 256>>>(a, bytes / 16);
    k_line_write4<<<4096, 256>>>((float*)a, bytes / 128);
    k_line_write64<<<4096, 256>>>(a, bytes / 128);
    (void)hipDeviceSynchronize();
    printf("bytes per kernel: %zu (= %zu KiB)\n", bytes, bytes >> 10);
    return 0;
}
