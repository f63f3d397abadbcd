// main.cpp -- driver: device-property report + timed runs of the BASELINE configs through the C ABI.
//
// Counterpart of the reference's main.cpp:5-33, whose check_gpu_props() prints the device limits
// (compute capability, SM count, memory, shared memory, registers, warp size, L2, threads/SM,
// main.cpp:10-25) and whose main() is empty (:30-33), plus the launch half of tests/main.cu:21-103.
// The CPU check half lives in tests/main.cpp (it links the oracle; this driver does not).
//
// usage: fa_main [--props] [--config N]... [--gpus N] [--iters K]
//   configs (BASELINE.json): 1 = bf16 B4 H8 S2048 d64      2 = bf16 B8 H16 S4096 d128 causal
//                            3 = fp8 e4m3fn B1 H16 S16384 d128 (B, H unspecified in BASELINE.json)
//                            4 = bf16 B64 H32 S8192 d128 sharded over --gpus devices (B*H split)
//   Multi-GPU: one host thread + one stream per device, each calling flash_attention on its
//   contiguous slab of B*H heads; no collective is needed (each head is independent).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/flash_attention.h"

#define HIP_CHECK(x)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "HIP error: %s (%s:%d)\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

static void check_gpu_props(int device) {
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    printf("Device %d: %s (%s)\n", device, prop.name, prop.gcnArchName);
    printf("Compute units: %d\n", prop.multiProcessorCount);
    printf("Global memory: %zu MB\n", prop.totalGlobalMem / (1024 * 1024));
    printf("LDS per workgroup: %zu KB\n", prop.sharedMemPerBlock / 1024);
    printf("LDS per CU: %zu KB\n", prop.maxSharedMemoryPerMultiProcessor / 1024);
    printf("Registers per workgroup: %d\n", prop.regsPerBlock);
    printf("Wavefront size: %d\n", prop.warpSize);
    printf("L2 cache size: %d KB\n", prop.l2CacheSize / 1024);
    printf("Max threads per CU: %d\n", prop.maxThreadsPerMultiProcessor);
    printf("Clock: %d MHz, memory clock: %d MHz, bus %d bit\n", prop.clockRate / 1000,
           prop.memoryClockRate / 1000, prop.memoryBusWidth);
}

static inline uint64_t mix(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static inline uint16_t bf16_of(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// N(0,1) bf16 pattern of `n` elements (never zeros or constants: they read 20 % high).
static std::vector<uint16_t> random_bf16(uint64_t seed, size_t n) {
    std::vector<uint16_t> v(n);
    for (size_t i = 0; i < n; i += 2) {
        const uint64_t a = mix(seed + i), b = mix(seed + i + 1);
        const double u1 = ((a >> 11) + 1.0) / 9007199254740993.0, u2 = (b >> 11) / 9007199254740992.0;
        const double r = std::sqrt(-2.0 * std::log(u1));
        v[i] = bf16_of((float)(r * std::cos(6.283185307179586 * u2)));
        if (i + 1 < n) v[i + 1] = bf16_of((float)(r * std::sin(6.283185307179586 * u2)));
    }
    return v;
}

// f32 -> OCP e4m3fn byte, round to nearest even, saturating at +-448 (bias 7, 3 mantissa bits, no infinities).
static uint8_t e4m3fn_of(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    float a = std::fabs(x);
    if (!(a == a)) return (uint8_t)(sign | 0x7fu);
    if (a >= 448.f) return (uint8_t)(sign | 0x7eu);
    if (a < 0.0009765625f) return sign;                          // below half the smallest subnormal (2^-10)
    int e;
    const float m = std::frexp(a, &e);                           // a = m * 2^e, m in [0.5, 1)
    int E = e - 1 + 7;                                           // biased exponent of 1.xxx * 2^(e-1)
    if (E <= 0) {                                                // subnormal: multiples of 2^-9
        const int q = (int)std::nearbyint(a * 512.f);
        return (uint8_t)(sign | (q >= 8 ? 0x08 : q));
    }
    int q = (int)std::nearbyint((m * 2.f - 1.f) * 8.f);          // 3 mantissa bits
    if (q == 8) { q = 0; ++E; }
    if (E > 15 || (E == 15 && q == 7)) return (uint8_t)(sign | 0x7eu);
    return (uint8_t)(sign | (E << 3) | q);
}

struct Config { int id, B, H, S, d, causal, dtype; };   // dtype: FA_DTYPE_BF16 or FA_DTYPE_FP8_E4M3

struct RankResult { double ms_med = 0, ms_min = 0; int rc = 0; };

// One device: allocate its slab of `heads` flattened heads, fill with random bf16, time `iters` launches.
static void run_rank(int dev, const Config& c, int heads, int iters, RankResult* out) {
    HIP_CHECK(hipSetDevice(dev));
    hipStream_t st;
    HIP_CHECK(hipStreamCreate(&st));
    const size_t per_head = (size_t)c.S * c.d, n = per_head * heads;
    const size_t esz = c.dtype == FA_DTYPE_FP8_E4M3 ? 1 : 2;
    void *q, *k, *v, *o;
    HIP_CHECK(hipMalloc(&q, n * esz)); HIP_CHECK(hipMalloc(&k, n * esz));
    HIP_CHECK(hipMalloc(&v, n * esz)); HIP_CHECK(hipMalloc(&o, n * 2));
    // 16 distinct random heads per tensor, tiled over the slab with device-to-device copies
    const int distinct = std::min(heads, 16);
    for (int t = 0; t < 3; ++t) {
        std::vector<uint16_t> h = random_bf16(1000 * (t + 1) + 17 * dev, per_head * distinct);
        char* dst = (char*)(t == 0 ? q : t == 1 ? k : v);
        if (esz == 1) {   // the same N(0,1) draws, rounded to e4m3fn
            std::vector<uint8_t> h8(h.size());
            for (size_t i = 0; i < h.size(); ++i) {
                const uint32_t u = (uint32_t)h[i] << 16;
                float f;
                memcpy(&f, &u, 4);
                h8[i] = e4m3fn_of(f);
            }
            HIP_CHECK(hipMemcpy(dst, h8.data(), per_head * distinct, hipMemcpyHostToDevice));
        } else {
            HIP_CHECK(hipMemcpy(dst, h.data(), per_head * distinct * 2, hipMemcpyHostToDevice));
        }
        for (int g = distinct; g < heads; g += distinct) {
            const int cnt = std::min(distinct, heads - g);
            HIP_CHECK(hipMemcpy(dst + (size_t)g * per_head * esz, dst, per_head * cnt * esz, hipMemcpyDeviceToDevice));
        }
    }
    const float scale = 1.0f / std::sqrt((float)c.d);
    // the slab is a dense [heads, S, d] tensor: pass batchSize = heads, numHeads = 1
    for (int i = 0; i < 3; ++i)
        out->rc |= flash_attention(q, k, v, o, heads, 1, c.S, c.d, scale, c.causal != 0, c.dtype, FA_DTYPE_BF16, st);
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    std::vector<float> ms(iters);
    // batches of REPS back-to-back launches per event pair (a launch from an idle, down-clocked GPU reads slow)
    constexpr int REPS = 5;
    for (int i = 0; i < iters; ++i) {
        HIP_CHECK(hipEventRecord(e0, st));
        for (int r = 0; r < REPS; ++r)
            out->rc |= flash_attention(q, k, v, o, heads, 1, c.S, c.d, scale, c.causal != 0, c.dtype, FA_DTYPE_BF16, st);
        HIP_CHECK(hipEventRecord(e1, st));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipEventElapsedTime(&ms[i], e0, e1));
        ms[i] /= REPS;
    }
    std::sort(ms.begin(), ms.end());
    out->ms_med = ms[iters / 2];
    out->ms_min = ms[0];
    HIP_CHECK(hipFree(q)); HIP_CHECK(hipFree(k)); HIP_CHECK(hipFree(v)); HIP_CHECK(hipFree(o));
    HIP_CHECK(hipStreamDestroy(st));
}

int main(int argc, char** argv) {
    std::vector<int> ids;
    int gpus = 1, iters = 20;
    bool props = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--props") props = true;
        else if (a == "--config" && i + 1 < argc) ids.push_back(atoi(argv[++i]));
        else if (a == "--gpus" && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (a == "--iters" && i + 1 < argc) iters = atoi(argv[++i]);
    }
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    printf("%s, %d device(s) visible\n", flash_attention_version(), ndev);
    if (props || ids.empty()) check_gpu_props(0);
    if (gpus > ndev) {
        printf("requested %d GPUs, only %d visible: running on %d (larger counts NOT measured)\n", gpus, ndev, ndev);
        gpus = ndev;
    }
    const Config all[] = {{1, 4, 8, 2048, 64, 0, FA_DTYPE_BF16}, {2, 8, 16, 4096, 128, 1, FA_DTYPE_BF16},
                          {3, 1, 16, 16384, 128, 0, FA_DTYPE_FP8_E4M3},   // BASELINE cfg3: B, H unspecified there -> 1, 16
                          {4, 64, 32, 8192, 128, 0, FA_DTYPE_BF16}};
    for (int id : ids) {
        const Config* c = nullptr;
        for (const Config& x : all) if (x.id == id) c = &x;
        if (!c) { printf("config %d: not a GPU config of this driver (0 is the CPU-only case)\n", id); continue; }
        const int BH = c->B * c->H;
        const int n = id == 4 ? gpus : 1;
        std::vector<RankResult> res(n);
        std::vector<std::thread> th;
        for (int r = 0; r < n; ++r) {
            const int h0 = (int)((int64_t)BH * r / n), h1 = (int)((int64_t)BH * (r + 1) / n);
            th.emplace_back(run_rank, r, *c, h1 - h0, iters, &res[r]);
        }
        for (auto& t : th) t.join();
        double worst = 0;
        int rc = 0;
        for (auto& r : res) { worst = std::max(worst, r.ms_med); rc |= r.rc; }
        const double flops = (c->causal ? 2.0 : 4.0) * BH * (double)c->S * c->S * c->d;
        const double tf = flops / (worst * 1e-3) / 1e12;
        printf("{\"config\": %d, \"dtype\": \"%s\", \"B\": %d, \"H\": %d, \"S\": %d, \"d\": %d, \"causal\": %d, \"gpus\": %d, \"rc\": %d, "
               "\"ms_median_max_over_gpus\": %.4f, \"tflops\": %.1f, \"frac_of_bf16_peak\": %.4f}\n",
               id, c->dtype == FA_DTYPE_FP8_E4M3 ? "fp8_e4m3fn" : "bf16", c->B, c->H, c->S, c->d, c->causal, n, rc, worst, tf,
               tf / (2516.6 * n));
    }
    return 0;
}
