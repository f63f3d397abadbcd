// atomic_latency.hip -- what a device-scope atomicAdd with return costs a wave on MI355X, alone and under contention (the question
// behind a run-time work queue for the early units).  One wave per workgroup, N dependent atomics each, timed with
// s_memtime; grid = 1 (uncontended), 32 (one XCD's worth when spread: every 8th workgroup index), 256 (all CUs), on one address or on one
// address per XCD group (128 bytes apart).
// build: hipcc --offload-arch=gfx950 -O3 -o tests/micro/atomic_latency tests/micro/atomic_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void probe(unsigned* ctr, int per_xcd, int n, unsigned long long* out, unsigned* sink) {
    unsigned* a = ctr + (per_xcd ? (blockIdx.x & 7) * 32 : 0);
    unsigned long long t0, t1;
    unsigned v = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < n; ++i) {
        if (threadIdx.x == 0) v += atomicAdd(a, 1u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = v; }
}

int main() {
    unsigned *ctr, *sink;
    unsigned long long* out;
    CHECK(hipMalloc(&ctr, 4096)); CHECK(hipMalloc(&sink, 4096)); CHECK(hipMalloc(&out, 256 * 8));
    const int n = 64;
    for (int per_xcd = 0; per_xcd < 2; ++per_xcd)
        for (int grid : {1, 8, 32, 64, 256}) {
            CHECK(hipMemset(ctr, 0, 4096));
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(64), 0, nullptr, ctr, per_xcd, n, out, sink);
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(grid);
            CHECK(hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            printf("%-22s %3d workgroups (one lane each): cycles per dependent atomic  min %.0f  median %.0f  max %.0f\n",
                   per_xcd ? "one address per XCD" : "one address", grid, (double)h[0] / n, (double)h[grid / 2] / n, (double)h[grid - 1] / n);
        }
    return 0;
}
