// valu_rates.hip -- issue-rate microbenchmark for the instructions of the softmax slots (gfx950).
// One wave per SIMD (grid = 1 block of 256 threads), each test = REP x (UNROLL independent instructions)
// timed with s_memtime; prints cycles per instruction per wave.  Build: make tests/micro/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

constexpr int REP = 256;

template <int WHICH>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long* out, float seed) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    f32x16 acc = {0};
    bf16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
    unsigned long long t0 = now();
    for (int r = 0; r < REP; ++r) {
        if constexpr (WHICH == 0) {          // 16 x v_fma_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
        } else if constexpr (WHICH == 1) {   // 16 x v_exp_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if constexpr (WHICH == 2) {   // 8 x v_pk_fma_f32 (16 elements)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x2 v = {a[2 * i], a[2 * i + 1]};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(v));
                a[2 * i] = v[0]; a[2 * i + 1] = v[1];
            }
        } else if constexpr (WHICH == 3) {   // 8 x v_pk_add_f32
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x2 v = {a[2 * i], a[2 * i + 1]};
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(v));
                a[2 * i] = v[0]; a[2 * i + 1] = v[1];
            }
        } else if constexpr (WHICH == 4) {   // 16 x v_cvt_pk_bf16_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
        } else if constexpr (WHICH == 5) {   // 16 x v_dot2_f32_bf16 (acc += a.lo*b.lo + a.hi*b.hi)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_dot2_f32_bf16 %0, %1, %1, %0" : "+v"(a[i]) : "v"(seed));
        } else if constexpr (WHICH == 6) {   // 4 MFMA alone
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        } else if constexpr (WHICH == 7) {   // 4 MFMA + 16 exp (do they overlap?)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(a[4 * i + k]));
            }
        } else if constexpr (WHICH == 8) {   // 4 MFMA + 28 fma (7 per MFMA: fills every issue slot under an 8-pass MFMA)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 7; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[(4 * i + k) & 15]) : "v"(seed));
            }
        } else if constexpr (WHICH == 9) {   // 16 x v_max3_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
        } else if constexpr (WHICH == 10) {  // 16 x v_exp_f16
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
        } else if constexpr (WHICH == 11) {  // 4 MFMA + 8 exp + 8 fma + 8 add + 4 cvt per 4 MFMA ~ one softmax slice
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[4 * i]) : "v"(seed));
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[4 * i + 1]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[4 * i + 2]) : "v"(seed));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[4 * i + 3]) : "v"(seed));
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[4 * i]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[4 * i + 1]) : "v"(seed));
            }
        }
    }
    unsigned long long t1 = now();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678f) out[1000] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int W>
static int run(const char* name, int per_iter, unsigned long long* d, int waves_per_simd) {
    std::vector<unsigned long long> h(8);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((rate_kernel<W>), dim3(1), dim3(256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd), 0, nullptr, d, 1.0f);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(h.data(), d, 32, hipMemcpyDeviceToHost));
    printf("  %-44s %7.2f cycles / instruction (wave 0; %d instr per iteration, %d wave(s) per SIMD) -> %.0f cycles per iteration\n", name,
           (double)h[0] / REP / per_iter, per_iter, waves_per_simd, (double)h[0] / REP);
    return 0;
}

int main() {
    unsigned long long* d;
    HIP_CHECK(hipMalloc(&d, 1 << 16));
    for (int w = 1; w <= 2; ++w) {
        printf("waves per SIMD = %d\n", w);
        run<0>("v_fma_f32", 16, d, w);
        run<1>("v_exp_f32", 16, d, w);
        run<10>("v_exp_f16", 16, d, w);
        run<2>("v_pk_fma_f32 (2 elements each)", 8, d, w);
        run<3>("v_pk_add_f32 (2 elements each)", 8, d, w);
        run<4>("v_cvt_pk_bf16_f32", 16, d, w);
        run<5>("v_dot2_f32_bf16", 16, d, w);
        run<9>("v_max3_f32", 16, d, w);
        run<6>("v_mfma_f32_32x32x16_bf16 (dependent chain)", 4, d, w);
        run<7>("4 MFMA + 16 v_exp_f32 interleaved", 20, d, w);
        run<8>("4 MFMA + 28 v_fma_f32 interleaved", 32, d, w);
        run<11>("4 MFMA + 8 fma + 8 exp + 8 add interleaved", 28, d, w);
    }
    return 0;
}
