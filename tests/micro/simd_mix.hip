// simd_mix.hip -- how close can 2 waves on one SIMD keep the MFMA pipe to 100 % while issuing the
// flash-attention instruction mix?  (gfx950; one workgroup of 8 waves on one CU, no global memory, no barrier.)
// Each "slot" = 1 v_mfma_f32_32x32x16_bf16 + a configurable number of VALU / transcendental / LDS instructions.
// Prints cycles per MFMA per SIMD (32 = pipe saturated), the aggregate TFLOP/s and the clock the chip held.
//   simd_mix [workgroups] [iterations]     table of instruction mixes (1 workgroup = one CU; 256 = whole chip)
//   simd_mix --levers [iterations]         whole chip, random operands: the attention mix with one ingredient removed at a time
//   simd_mix --ceiling [iterations]        one JSON line: whole-chip throughput with RANDOM bf16 operands for
//                                          (a) MFMA only and (b) MFMA + the softmax VALU mix + the K/V^T LDS-read
//                                          mix of the attention kernel.  On MI355X these are set by the power
//                                          limit, not by the 2.4 GHz nominal clock behind the 2516.6 TFLOP/s peak.
// Build: make tests/micro/simd_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

constexpr int SLOTS = 32;   // 32 MFMAs per "tile" per wave

// NFMA/NEXP/NADD/NCVT: VALU instructions per slot (x16 fixed-point: 16 = one per slot, 8 = one every second slot)
// M16: each slot issues TWO v_mfma_f32_16x16x32_bf16 (same FLOPs, same operand bytes as one 32x32x16) -- the guide's
// "DVFS give-back" item 7: on random data the chip holds a higher clock on the 16x16x32 shape.
template <int NFMA, int NEXP, int NADD, int NCVT, int NB128, int NTR, bool CHAIN, bool RANDOM = false, int OPDEP = 0, bool SDEP = false, bool M16 = false>
__global__ __launch_bounds__(512) void mix_kernel(unsigned long long* out, float seed, int REP) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + lane * 1e-3f + i;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    f32x4 acc16[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
    bf16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
    u32x4 kf[4] = {{0}, {0}, {0}, {0}};
    u32x2 vf[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
    const int off = (threadIdx.x * 16) & 32767;
    bf16x8 qf[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qf[k] = fb;
    if constexpr (RANDOM) {
        // bf16 values with random sign / mantissa and exponents around 1: what real Q, K, V, P fragments toggle
        auto rnd = [](unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
        auto two_bf16 = [&](unsigned x) {
            const unsigned r = rnd(x);
            const unsigned lo = (r & 0x80ffu) | ((0x7du + ((r >> 8) & 3u)) << 7), hi = ((r >> 16) & 0x80ffu) | ((0x7du + ((r >> 30) & 3u)) << 7);
            return lo | (hi << 16);
        };
        for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)lds)[i] = two_bf16(i * 2654435761u + blockIdx.x);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            u32x4 v = {two_bf16(threadIdx.x * 64 + k * 4 + 0 + 7777), two_bf16(threadIdx.x * 64 + k * 4 + 1 + 7777), two_bf16(threadIdx.x * 64 + k * 4 + 2 + 7777), two_bf16(threadIdx.x * 64 + k * 4 + 3 + 7777)};
            qf[k] = __builtin_bit_cast(bf16x8, v);
        }
    } else {
        for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)lds)[i] = i;
    }
    __syncthreads();
    unsigned long long t0 = now();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            // operands "loaded" 4 slots ago
            if constexpr (NB128 > 0) if ((s * NB128) % 16 < NB128) {
                asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(3 * (NB128 + NTR) / 16 + 1));
                fa = __builtin_bit_cast(bf16x8, kf[s & 3]);
            }
            if constexpr (OPDEP > 0) {
                // like the attention kernel: EVERY MFMA waits for its own A operand, read OPDEP slots earlier --
                // first half of the slots from one ds_read_b128, second half from two ds_read_b64_tr_b16
                if (s < SLOTS / 2) {
                    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(OPDEP - 1));
                    fa = __builtin_bit_cast(bf16x8, kf[s % OPDEP]);
                } else {
                    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(2 * (OPDEP - 1)));
                    u32x4 t = {vf[(2 * s) % (2 * OPDEP)][0], vf[(2 * s) % (2 * OPDEP)][1], vf[(2 * s + 1) % (2 * OPDEP)][0], vf[(2 * s + 1) % (2 * OPDEP)][1]};
                    fa = __builtin_bit_cast(bf16x8, t);
                }
            }
            if constexpr (RANDOM) fb = qf[s & 7];
            if constexpr (M16) {
                acc16[(2 * s) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc16[(2 * s) & 7], 0, 0, 0);
                acc16[(2 * s + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, RANDOM ? qf[(s + 3) & 7] : fb, acc16[(2 * s + 1) & 7], 0, 0, 0);
            } else
            if constexpr (CHAIN) acc[(s >> 3) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[(s >> 3) & 3], 0, 0, 0);
            else acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[s & 3], 0, 0, 0);
            if constexpr (SDEP) {
                // like the kernel's exponent fma: its input is an accumulator element a DIFFERENT MFMA chain produced
                // (32 MFMAs ago), its output feeds the exp below
                asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(a[(s + 2) & 7]) : "v"(acc[(s + 2) & 3][s & 15]), "v"(seed));
            } else
            if constexpr (NFMA > 0) if ((s * NFMA) % 16 < NFMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[s & 7]) : "v"(seed));
            if constexpr (NEXP > 0) if ((s * NEXP) % 16 < NEXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[(s + 2) & 7]));
            if constexpr (NADD > 0) if ((s * NADD) % 16 < NADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[(s + 4) & 7]) : "v"(seed));
            if constexpr (NCVT > 0) if ((s * NCVT) % 16 < NCVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[(s + 6) & 7]) : "v"(seed));
            if constexpr (OPDEP > 0) {
                // refill the operand slot just consumed (needed again OPDEP slots from now)
                if (s < SLOTS / 2) {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(kf[s % OPDEP]) : "v"(off + 1024 * (s & 15)));
                } else {
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vf[(2 * s) % (2 * OPDEP)]) : "v"((off >> 1) + 512 * (s & 31)));
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vf[(2 * s + 1) % (2 * OPDEP)]) : "v"((off >> 1) + 512 * (s & 31) + 256));
                }
            } else {
            if constexpr (NB128 > 0) if ((s * NB128) % 16 < NB128)
                asm volatile("ds_read_b128 %0, %1" : "=v"(kf[s & 3]) : "v"(off + 1024 * (s & 15)));
            if constexpr (NTR > 0) if ((s * NTR) % 16 < NTR)
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vf[s & 3]) : "v"((off >> 1) + 512 * (s & 31)));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = now();
    float sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += a[i];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += acc[k][i];
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += acc16[k][0] + acc16[k][1] + acc16[k][2] + acc16[k][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) sum += (float)(kf[k][0] + vf[k][0] + vf[k + 4][0]);
    if (sum == 12345.678f) out[1000] = 1;
    if (lane == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}

static int g_grid = 1, g_rep = 128;
static bool g_quiet = false;
static double g_last_tflops = 0, g_last_clock = 0, g_last_pipe = 0;

template <int NFMA, int NEXP, int NADD, int NCVT, int NB128, int NTR, bool CHAIN, bool RANDOM = false, int OPDEP = 0, bool SDEP = false, bool M16 = false>
static int run(const char* name, unsigned long long* d) {
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    for (int waves = 4; waves <= 8; waves += 4) {
        std::vector<unsigned long long> h(8);
        float ms = 0;
        for (int i = 0; i < 3; ++i) {
            HIP_CHECK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL((mix_kernel<NFMA, NEXP, NADD, NCVT, NB128, NTR, CHAIN, RANDOM, OPDEP, SDEP, M16>), dim3(g_grid), dim3(64 * waves), 0, nullptr, d, 1.0f, g_rep);
            HIP_CHECK(hipEventRecord(e1, nullptr));
            HIP_CHECK(hipEventSynchronize(e1));
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        HIP_CHECK(hipMemcpy(h.data(), d, 64, hipMemcpyDeviceToHost));
        unsigned long long mx = 0;
        for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
        const double per_mfma_simd = (double)mx / g_rep / SLOTS / (waves / 4);
        const double tflops = (double)g_grid * waves * g_rep * SLOTS * 32768.0 / (ms * 1e-3) / 1e12;
        g_last_tflops = tflops; g_last_clock = (double)mx / (ms * 1e-3) / 1e9; g_last_pipe = 32.0 / per_mfma_simd;
        if (!g_quiet)
            printf("  %-58s %d wave(s)/SIMD: %6.1f cycles per MFMA per SIMD (pipe %3.0f%%)  %7.1f TFLOP/s  clock %.2f GHz\n", name, waves / 4,
                   per_mfma_simd, 3200.0 / per_mfma_simd, tflops, g_last_clock);
    }
    return 0;
}

int main(int argc, char** argv) {
    unsigned long long* d;
    HIP_CHECK(hipMalloc(&d, 1 << 16));
    if (argc > 1 && std::string(argv[1]) == "--ceiling") {
        // one JSON line for bench.py: power-limited throughput of this device, all CUs busy, random bf16 operands
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        g_grid = cus; g_rep = argc > 2 ? atoi(argv[2]) : 3000; g_quiet = true;
        run<0, 0, 0, 0, 0, 0, false, true>("", d);
        const double a = g_last_tflops, ac = g_last_clock, ap = g_last_pipe;
        run<16, 16, 16, 8, 8, 16, false, true>("", d);
        printf("{\"mfma_only_random_bf16_tflops\": %.1f, \"mfma_only_clock_ghz\": %.3f, \"mfma_only_pipe_busy\": %.3f, "
               "\"attention_mix_random_bf16_tflops\": %.1f, \"attention_mix_clock_ghz\": %.3f, \"attention_mix_pipe_busy\": %.3f, "
               "\"workgroups\": %d, \"mfma_per_wave\": %d}\n", a, ac, ap, g_last_tflops, g_last_clock, g_last_pipe, g_grid, g_rep * SLOTS);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "--levers") {
        // What each ENERGY lever is worth on this device: whole chip, random bf16 operands, the attention kernel's instruction mix with one
        // ingredient taken out at a time (the chip is power-limited under these streams: TFLOP/s follows the clock it holds).
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        g_grid = cus; g_rep = argc > 2 ? atoi(argv[2]) : 3000;
        printf("grid = %d workgroups, %d iterations of %d slots, RANDOM bf16 operands\n", g_grid, g_rep, SLOTS);
        //   fma exp add cvt b128 tr  chain  random
        run<0, 0, 0, 0, 0, 0, false, true>("MFMA only", d);
        run<16, 16, 16, 8, 8, 16, false, true>("attention mix: fma + exp + add + 0.5 cvt, 0.5 b128 + 1 tr64", d);
        run<16, 16, 0, 8, 8, 16, false, true>("  without the row-sum add", d);
        run<0, 16, 16, 8, 8, 16, false, true>("  without the fma (operand prescaled)", d);
        run<16, 16, 16, 8, 4, 8, false, true>("  with half the LDS reads (64 rows per wave)", d);
        run<16, 16, 0, 8, 4, 8, false, true>("  without the add, half the LDS reads", d);
        run<0, 16, 0, 8, 4, 8, false, true>("  without add and fma, half the LDS reads", d);
        run<16, 16, 16, 8, 0, 0, false, true>("  without any LDS read", d);
        run<0, 0, 0, 0, 8, 16, false, true>("  LDS reads only (no softmax instructions)", d);
        run<16, 16, 16, 8, 8, 16, false, true, 0, false, true>("attention mix on 16x16x32 MFMAs (two per slot)", d);
        run<16, 16, 0, 8, 8, 16, false, true, 0, false, true>("  without the row-sum add (16x16x32)", d);
        run<16, 16, 16, 8, 4, 8, false, true, 0, false, true>("  with half the LDS reads (16x16x32)", d);
        return 0;
    }
    if (argc > 1) g_grid = atoi(argv[1]);
    if (argc > 2) g_rep = atoi(argv[2]);
    printf("grid = %d workgroup(s), %d iterations of %d slots\n", g_grid, g_rep, SLOTS);
    //      fma exp add cvt b128 tr  chain
    run<0, 0, 0, 0, 0, 0, false>("MFMA only, 4 independent accumulators", d);
    run<0, 0, 0, 0, 0, 0, true>("MFMA only, chains of 8 on one accumulator", d);
    run<16, 0, 16, 8, 0, 0, false>("+ 1 fma + 1 add + 0.5 cvt per MFMA", d);
    run<0, 16, 0, 0, 0, 0, false>("+ 1 exp per MFMA", d);
    run<16, 16, 16, 8, 0, 0, false>("+ 1 fma + 1 exp + 1 add + 0.5 cvt per MFMA (softmax mix)", d);
    run<0, 0, 0, 0, 8, 16, false>("+ 0.5 ds_read_b128 + 1 ds_read_b64_tr per MFMA", d);
    run<16, 16, 16, 8, 8, 16, false>("softmax mix + LDS mix", d);
    run<16, 16, 16, 8, 8, 16, true>("softmax mix + LDS mix, chained accumulators", d);
    run<8, 8, 8, 4, 8, 16, false>("half softmax mix + LDS mix", d);
    run<0, 0, 0, 0, 0, 0, false, true>("RANDOM bf16 operands: MFMA only (B cycles over 8 fragments)", d);
    run<16, 16, 16, 8, 0, 0, false, true>("RANDOM bf16 operands: MFMA + softmax mix, no LDS", d);
    run<0, 0, 0, 0, 8, 16, false, true>("RANDOM bf16 operands: MFMA + LDS mix (0.5 b128 + 1 tr)", d);
    run<0, 0, 0, 0, 4, 8, false, true>("RANDOM bf16 operands: MFMA + half LDS mix (R=2-like)", d);
    run<0, 0, 0, 0, 16, 0, false, true>("RANDOM bf16 operands: MFMA + 1 ds_read_b128 per MFMA", d);
    run<16, 16, 16, 8, 8, 16, false, true>("RANDOM bf16 operands: softmax mix + LDS mix (A from LDS)", d);
    run<16, 16, 16, 8, 4, 8, false, true>("RANDOM bf16 operands: softmax mix + half LDS mix", d);
    run<16, 16, 16, 8, 0, 0, false, true, 2>("RANDOM: softmax mix, every MFMA waits for an LDS operand read 2 slots earlier", d);
    run<16, 16, 16, 8, 0, 0, false, true, 4>("RANDOM: softmax mix, every MFMA waits for an LDS operand read 4 slots earlier", d);
    run<16, 16, 16, 8, 0, 0, true, true, 2>("RANDOM: same (2 slots), first 16 MFMAs in chains of 8", d);
    run<16, 16, 16, 8, 0, 0, false, true, 2, true>("RANDOM: same (2 slots) + the exponent fma reads MFMA accumulators", d);
    run<0, 0, 0, 0, 0, 0, false, true, 0, false, true>("M16 RANDOM: 2 x 16x16x32 per slot, MFMA only", d);
    run<16, 16, 16, 8, 0, 0, false, true, 0, false, true>("M16 RANDOM: + softmax mix, no LDS", d);
    run<0, 0, 0, 0, 8, 16, false, true, 0, false, true>("M16 RANDOM: + LDS mix (0.5 b128 + 1 tr)", d);
    run<16, 16, 16, 8, 8, 16, false, true, 0, false, true>("M16 RANDOM: softmax mix + LDS mix", d);
    run<12, 16, 8, 8, 8, 16, false, true, 0, false, true>("M16 RANDOM: reduced softmax mix (0.75 fma, 1 exp, 0.5 add) + LDS mix", d);
    run<16, 16, 16, 8, 0, 0, false, true, 2, false, true>("M16 RANDOM: softmax mix, every slot waits for an LDS operand read 2 slots earlier", d);
    run<0, 0, 0, 0, 0, 0, false, false, 0, false, true>("M16 constant operands: MFMA only", d);
    run<16, 16, 16, 8, 8, 16, false, false, 0, false, true>("M16 constant operands: softmax mix + LDS mix", d);
    // the 16x16x32 kernels' own mixes per slot (= 2 MFMAs): row sums come from an MFMA, so no v_add
    run<16, 16, 0, 8, 8, 16, false, true, 0, false, true>("M16 RANDOM: kernel mix, 32 rows/wave (1 fma + 1 exp + .5 cvt + .5 b128 + 1 tr)", d);
    run<16, 16, 0, 8, 4, 8, false, true, 0, false, true>("M16 RANDOM: kernel mix, 64 rows/wave (half the LDS reads)", d);
    run<16, 16, 0, 8, 2, 4, false, true, 0, false, true>("M16 RANDOM: kernel mix, 128 rows/wave-equivalent (quarter LDS reads)", d);
    run<16, 16, 0, 8, 0, 0, false, true, 0, false, true>("M16 RANDOM: kernel mix without LDS reads", d);
    run<0, 16, 0, 8, 8, 16, false, true, 0, false, true>("M16 RANDOM: kernel mix without the exponent fma", d);
    return 0;
}
