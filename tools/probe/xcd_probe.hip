// Can the LAST-arriving workgroup of a group read the other members' stores WITHOUT an agent-scope release (no buffer_wbl2),
// when all members run on the same XCD (workgroup i -> XCD i % 8) and share its L2?  Members: store, wait for the stores
// (vmcnt 0), barrier, relaxed agent-scope atomic; last arriver: agent-scope ACQUIRE only (buffer_inv sc1), read back.
// same_xcd = 1: the G members of group g are workgroups (g / 8 * G + m) * 8 + g % 8; 0: G consecutive workgroups (G XCDs).
// Prints stale read-backs for both placements; time the kernels with rocprofv3 --kernel-trace.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>      // 0: stores only; 1: hand-over, acquire only; 2: hand-over with release + acquire (__threadfence both sides)
__global__ void __launch_bounds__(256) k_probe(float4* __restrict__ buf, int* __restrict__ cnt, unsigned* __restrict__ xcc_seen, float* __restrict__ out,
                                               const int per_wg4, const int G, const int same_xcd, const float tag) {
    const int bid = blockIdx.x;
    int g, m;
    if (same_xcd) { const int x = bid & 7, r = bid >> 3; m = r % G; g = (r / G) * 8 + x; }
    else { g = bid / G; m = bid - g * G; }
    float4* mine = buf + ((size_t)g * G + m) * per_wg4;
    float acc = 0.f;
    for (int i = threadIdx.x; i < per_wg4; i += 256) { const float4 v = mine[i]; acc += v.x * 1.0001f + v.y; }
    for (int i = threadIdx.x; i < per_wg4; i += 256) mine[i] = make_float4(acc, 1.f, 2.f, tag + (float)m);
    if (MODE == 0) return;
    if (MODE == 2) __threadfence();
    __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0) expcnt(0) lgkmcnt(0): the stores have been acknowledged
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicOr(&xcc_seen[g], 1u << (xcc & 7u));
        last = (__hip_atomic_fetch_add(&cnt[g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1);
    }
    __syncthreads();
    if (!last) return;
    if (MODE == 2) __threadfence(); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x == 0) cnt[g] = 0;
    float s = 0.f, expect = 0.f;
    for (int w = 0; w < G; ++w) { s += buf[((size_t)g * G + w) * per_wg4 + threadIdx.x * 7 % per_wg4].w; expect += tag + (float)w; }
    if (s != expect) atomicAdd(out, 1.0f);
    if (threadIdx.x == 0 && __popc(xcc_seen[g]) != 1) atomicAdd(out + 1, 1.0f);      // groups that spanned XCDs
}
int main() {
    const int n_wg = 4096, per_wg4 = 2048, G = 4;
    float4* buf; int* cnt; unsigned* seen; float* out;
    (void)hipMalloc(&buf, (size_t)n_wg * per_wg4 * 16); (void)hipMalloc(&cnt, n_wg * 4); (void)hipMalloc(&seen, n_wg * 4); (void)hipMalloc(&out, 8);
    (void)hipMemset(buf, 0, (size_t)n_wg * per_wg4 * 16); (void)hipMemset(cnt, 0, n_wg * 4);
    float tag = 1.f;
    for (int same = 1; same >= 0; --same) {
        for (int mode = 1; mode <= 2; ++mode) {
            (void)hipMemset(out, 0, 8); (void)hipMemset(seen, 0, n_wg * 4);
            for (int rep = 0; rep < 20; ++rep) {
                tag += 8.f;
                k_probe<0><<<n_wg, 256>>>(buf, cnt, seen, out, per_wg4, G, same, tag - 4.f);      // other values in the lines first
                if (mode == 1) k_probe<1><<<n_wg, 256>>>(buf, cnt, seen, out, per_wg4, G, same, tag);
                else k_probe<2><<<n_wg, 256>>>(buf, cnt, seen, out, per_wg4, G, same, tag);
            }
            float res[2];
            (void)hipMemcpy(res, out, 8, hipMemcpyDeviceToHost);
            printf("same_xcd %d  %s: stale read-backs %.0f of %d hand-overs, groups spanning XCDs %.0f\n", same,
                   mode == 1 ? "acquire only (no L2 write-back)" : "release + acquire (__threadfence)", res[0], 20 * n_wg / G * 256, res[1]);
        }
    }
    return 0;
}
