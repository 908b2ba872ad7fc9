// What does an agent-scope release (`__threadfence()` = buffer_wbl2 sc1 + wait on gfx950) cost when EVERY workgroup of a
// launch issues one after its stores?  (Question behind a fused glue tail for kernels with several workgroups per env: the
// last-arriving workgroup of an env would have to see the others' stores across XCDs.)
// Build: hipcc --offload-arch=gfx950 -O3 -o fence_probe fence_probe.hip ; run under rocprofv3 --kernel-trace --stats
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>      // 0: stores only, 1: + __threadfence(), 2: + fence + atomic counter + last-arriver acquire and read-back
__global__ void __launch_bounds__(256) k_probe(float4* __restrict__ buf, int* __restrict__ cnt, float* __restrict__ out, const int per_wg4, const int group) {
    float4* mine = buf + (size_t)blockIdx.x * per_wg4;
    float acc = 0.f;
    // some reading + arithmetic so that workgroups finish at different times, then the stores
    for (int i = threadIdx.x; i < per_wg4; i += 256) { const float4 v = mine[i]; acc += v.x * 1.0001f + v.y; }
    for (int i = threadIdx.x; i < per_wg4; i += 256) mine[i] = make_float4(acc, 1.f, 2.f, (float)blockIdx.x);
    if (MODE >= 1) __threadfence();
    if (MODE == 2) {
        __shared__ int last;
        __syncthreads();
        const int g = blockIdx.x / group;
        if (threadIdx.x == 0) last = (atomicAdd(&cnt[g], 1) == group - 1);
        __syncthreads();
        if (last) {
            __threadfence();
            if (threadIdx.x == 0) cnt[g] = 0;
            float s = 0.f;
            for (int w = 0; w < group; ++w) s += buf[(size_t)(g * group + w) * per_wg4 + threadIdx.x].w;      // the others' stores
            // every workgroup of the group wrote its own index into .w
            float expect = 0.f;
            for (int w = 0; w < group; ++w) expect += (float)(g * group + w);
            if (s != expect) atomicAdd(out, 1.0f);      // count of stale reads
        }
    }
}
int main() {
    const int n_wg = 4096, per_wg4 = 2048 /* 32 KB per workgroup */, group = 4;
    float4* buf; int* cnt; float* out;
    (void)hipMalloc(&buf, (size_t)n_wg * per_wg4 * 16); (void)hipMalloc(&cnt, n_wg * 4); (void)hipMalloc(&out, 4);
    (void)hipMemset(buf, 0, (size_t)n_wg * per_wg4 * 16); (void)hipMemset(cnt, 0, n_wg * 4); (void)hipMemset(out, 0, 4);
    for (int rep = 0; rep < 5; ++rep) {
        k_probe<0><<<n_wg, 256>>>(buf, cnt, out, per_wg4, group);
        k_probe<1><<<n_wg, 256>>>(buf, cnt, out, per_wg4, group);
        k_probe<2><<<n_wg, 256>>>(buf, cnt, out, per_wg4, group);
    }
    float bad = -1.f;
    (void)hipMemcpy(&bad, out, 4, hipMemcpyDeviceToHost);
    printf("stale read-backs by last arrivers: %.0f\n", bad);
    return 0;
}
