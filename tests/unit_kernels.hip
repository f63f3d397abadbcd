// tests/unit_kernels.hip -- unit tests of the two things an end-to-end mismatch cannot localise:
//
//  (1) MFMA fragment layout (the guide's "always A = I with ASYMMETRIC B"): the wrappers of
//      csrc/utils.hip.h are fed A = [I | 0] and B[k][j] = 8k + j (exact in bf16; small integers for fp8), so D must
//      equal the first rows of B -- a swapped row/column map, a wrong acc_row() or a wrong operand k-order shows up as
//      a specific wrong element, not as "max abs err 3e-2".  Covered: bf16 32x32x16, fp8 32x32x16, the block-scaled
//      32x32x64 f8f6f4 form with unit scales, and the accumulator-as-next-operand k permutation the P.V product uses.
//  (2) LDS images (item 7; the intent of the reference's tests/test_loaders.cu:47-110: fill the global tile with
//      i + 1, run the loader, compare shared memory with the expected image): BufStage (csrc/loaders.hip.h) stages a
//      K and a V tile whose 16-bit elements are key*D + j + 1, the raw LDS image is compared with TileGeom's
//      k_lds_off / v_lds_off permutation, and -- what the MFMAs actually consume -- every K fragment read
//      (ds_read_b128) and every V^T fragment read (ds_read_b64_tr_b16) of WaveCompute is compared with the element it
//      must hold: K[32kt + (lane&31)][16u + 8h + j] and V[16s4 + 8(j>>2) + 4h + (j&3)][32db + (lane&31)].
//
// Prints one line per test, "N test(s) failed" at the end, exit status = number of failures.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../flash-attention-cuda-c_amd/csrc/kernel_bf16.hip.h"

#define HIP_CHECK(x)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "HIP error: %s (%s:%d)\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(100);                                                                        \
        }                                                                                     \
    } while (0)

using namespace fa;

static int g_fail = 0;
static void report(const char* name, long long bad, long long n) {
    printf("%s %s (%lld / %lld elements wrong)\n", bad ? "FAIL" : "PASS", name, bad, n);
    if (bad) ++g_fail;
}

// ------------------------------------------------------------------------------------------------ (1) MFMA layouts
static __device__ __host__ inline uint16_t bf16_bits(float x) {   // exact for the small integers used here
    uint32_t u;
#ifdef __HIP_DEVICE_COMPILE__
    u = __float_as_uint(x);
#else
    memcpy(&u, &x, 4);
#endif
    return (uint16_t)(u >> 16);
}

// A[i][k] = (i == k) (32x16: identity on top of zeros), B[k][j] = 8k + j.  Operand maps as documented in utils.hip.h:
// lane l (r = l&31, h = l>>5), element j holds A[r][8h+j] and B[8h+j][r].  out[row][col] row-major 32x32.
__global__ void mfma_bf16_layout(float* out) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    u32x4 a, b;
    uint16_t ae[8], be[8];
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * h + j;
        ae[j] = bf16_bits(r == k ? 1.f : 0.f);
        be[j] = bf16_bits((float)(8 * k + r));          // at most 8*15 + 31 = 151: exact in bf16 (8 significant bits)
    }
    for (int w = 0; w < 4; ++w) { a[w] = ae[2 * w] | ((uint32_t)ae[2 * w + 1] << 16); b[w] = be[2 * w] | ((uint32_t)be[2 * w + 1] << 16); }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = mfma_32x32x16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c);
    for (int i = 0; i < 16; ++i) out[acc_row(i, h) * 32 + r] = c[i];
}

// e4m3fn encodings of the integers 0..15 (exact): value = (1 + m/8) 2^(e-7)
__device__ __host__ inline uint8_t e4m3_of_small_int(int v) {
    const uint8_t t[16] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50, 0x51, 0x52, 0x53, 0x54, 0x55, 0x56, 0x57};
    return t[v];
}
// fp8 32x32x16: A = [I | 0] (32x16), B[k][j] = (k + 3j) % 16 -- asymmetric, every value an exact e4m3fn integer
__global__ void mfma_fp8_layout(float* out) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    uint64_t a = 0, b = 0;
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * h + j;
        a |= (uint64_t)e4m3_of_small_int(r == k ? 1 : 0) << (8 * j);
        b |= (uint64_t)e4m3_of_small_int((k + 3 * r) % 16) << (8 * j);
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = mfma_32x32x16_fp8(a, b, c);
    for (int i = 0; i < 16; ++i) out[acc_row(i, h) * 32 + r] = c[i];
}
// block-scaled 32x32x64, unit scales: a lane supplies 32 bytes per operand; byte j (0..31) of lane half h is contraction
// index k = 32h + j for BOTH operands (any common assignment works -- utils.hip.h).  A = [I(32) | 0], B[k][j] = (k + 3j) % 16.
__global__ void mfma_mx_layout(float* out) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    uint8_t ab[32], bb[32];
    for (int j = 0; j < 32; ++j) {
        const int k = 32 * h + j;
        ab[j] = e4m3_of_small_int(r == k ? 1 : 0);
        bb[j] = e4m3_of_small_int((k + 3 * r) % 16);
    }
    u32x4 alo, ahi, blo, bhi;
    for (int w = 0; w < 4; ++w) {
        alo[w] = ab[4 * w] | (ab[4 * w + 1] << 8) | (ab[4 * w + 2] << 16) | ((uint32_t)ab[4 * w + 3] << 24);
        ahi[w] = ab[16 + 4 * w] | (ab[17 + 4 * w] << 8) | (ab[18 + 4 * w] << 16) | ((uint32_t)ab[19 + 4 * w] << 24);
        blo[w] = bb[4 * w] | (bb[4 * w + 1] << 8) | (bb[4 * w + 2] << 16) | ((uint32_t)bb[4 * w + 3] << 24);
        bhi[w] = bb[16 + 4 * w] | (bb[17 + 4 * w] << 8) | (bb[18 + 4 * w] << 16) | ((uint32_t)bb[19 + 4 * w] << 24);
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = mfma_32x32x64_fp8_unit_scale(alo, ahi, blo, bhi, c);
    for (int i = 0; i < 16; ++i) out[acc_row(i, h) * 32 + r] = c[i];
}

// Accumulator as the next product's B operand (what O^T += V^T . P^T does with the score tile): X = [I|0].Bx gives
// X[i][j] = Bx[i][j] for i < 16 (a 32x32 tile, rows 16..31 zero).  Registers 8s..8s+7 of X, packed to bf16, are the B
// fragment of k-step s; element j of lane half h is then row 16s + 8(j>>2) + 4h + (j&3) of X (computers.hip.h, v_frag).
// With A2[i][k] = 1 iff k == i (32x32 identity, fed in THAT k order) the second product must reproduce X.
__global__ void mfma_acc_as_operand(float* out) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    u32x4 a, b;
    uint16_t ae[8], be[8];
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * h + j;
        ae[j] = bf16_bits(r == k ? 1.f : 0.f);
        be[j] = bf16_bits((float)(8 * k + r));          // X[i][j] = 8i + j for i < 16: at most 8*15+31 = 151, exact in bf16
    }
    for (int w = 0; w < 4; ++w) { a[w] = ae[2 * w] | ((uint32_t)ae[2 * w + 1] << 16); b[w] = be[2 * w] | ((uint32_t)be[2 * w + 1] << 16); }
    f32x16 x;
    for (int i = 0; i < 16; ++i) x[i] = 0.f;
    x = mfma_32x32x16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), x);
    f32x16 y;
    for (int i = 0; i < 16; ++i) y[i] = 0.f;
    for (int s = 0; s < 2; ++s) {
        u32x4 pb, a2;
        uint16_t a2e[8];
        for (int w = 0; w < 4; ++w) pb[w] = pack_bf16(x[8 * s + 2 * w], x[8 * s + 2 * w + 1]);
        for (int j = 0; j < 8; ++j) {
            const int krow = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);     // the X row this operand element multiplies
            a2e[j] = bf16_bits(r == krow ? 1.f : 0.f);
        }
        for (int w = 0; w < 4; ++w) a2[w] = a2e[2 * w] | ((uint32_t)a2e[2 * w + 1] << 16);
        y = mfma_32x32x16(__builtin_bit_cast(bf16x8, a2), __builtin_bit_cast(bf16x8, pb), y);
    }
    for (int i = 0; i < 16; ++i) out[acc_row(i, h) * 32 + r] = y[i];
}

// 16x16x32 bf16: A = [I(16) | 0] (16x32), B[k][j] = 4k + j (32x16) -> D[i][j] = 4i + j.  Operand maps (utils.hip.h): lane l
// (r = l&15, h4 = l>>4), element j holds A[r][8*h4 + j] and B[8*h4 + j][r]; C/D: col = l&15, row = 4*h4 + reg.  out: 16x16.
__global__ void mfma16_bf16_layout(float* out) {
    const int lane = threadIdx.x, r = lane & 15, h4 = lane >> 4;
    u32x4 a, b;
    uint16_t ae[8], be[8];
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * h4 + j;
        ae[j] = bf16_bits(r == k ? 1.f : 0.f);
        be[j] = bf16_bits((float)(4 * k + r));          // at most 4*31 + 15 = 139: exact in bf16
    }
    for (int w = 0; w < 4; ++w) { a[w] = ae[2 * w] | ((uint32_t)ae[2 * w + 1] << 16); b[w] = be[2 * w] | ((uint32_t)be[2 * w + 1] << 16); }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = mfma_16x16x32(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c);
    for (int i = 0; i < 4; ++i) out[(4 * h4 + i) * 32 + r] = c[i];
    for (int i = 0; i < 4; ++i) out[(16 + 4 * h4 + i) * 32 + r] = 0.f;      // (the harness compares a 32x32 array)
    for (int i = 0; i < 4; ++i) { out[(4 * h4 + i) * 32 + 16 + r] = 0.f; out[(16 + 4 * h4 + i) * 32 + 16 + r] = 0.f; }
}
// Accumulator tiles as the next product's B operand, 16x16x32 (computers16.hip.h): X0 = I.B0, X1 = I.B1 are two 16x16 score tiles
// (key groups 2kk, 2kk+1 of one query group: X_g[i][j] = 100*(16g + i) + j); their 8 registers, packed to bf16, are the B fragment
// of ONE k-step of 32 keys in which element j of quarter h4 is key 16*(j>>2) + 4*h4 + (j&3).  With A[i][k] = 1 iff that key == 2*i + 1
// (picks the odd keys 1, 3, .., 31) the product must return rows X[2i + 1].  Also: ONES.P^T = the column sums (the row-sum MFMA),
// and the reductions max_all_quarters / sum_all_quarters over lanes l, l^16, l^32, l^48.
__global__ void mfma16_acc_as_operand(float* out) {
    const int lane = threadIdx.x, r = lane & 15, h4 = lane >> 4;
    f32x4 x[2];
    for (int g = 0; g < 2; ++g) {
        u32x4 a, b;
        uint16_t ae[8], be[8];
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * h4 + j;
            ae[j] = bf16_bits(r == k ? 1.f : 0.f);
            be[j] = bf16_bits(k < 16 ? (float)(4 * (16 * g + k) + r) : 0.f);     // X_g[i][j] = 4*(16g+i) + j <= 139: exact in bf16
        }
        for (int w = 0; w < 4; ++w) { a[w] = ae[2 * w] | ((uint32_t)ae[2 * w + 1] << 16); b[w] = be[2 * w] | ((uint32_t)be[2 * w + 1] << 16); }
        x[g] = mfma_16x16x32(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), f32x4{0.f, 0.f, 0.f, 0.f});
    }
    u32x4 pb = {pack_bf16(x[0][0], x[0][1]), pack_bf16(x[0][2], x[0][3]), pack_bf16(x[1][0], x[1][1]), pack_bf16(x[1][2], x[1][3])};
    u32x4 a2;
    uint16_t a2e[8];
    for (int j = 0; j < 8; ++j) {
        const int key = 16 * (j >> 2) + 4 * h4 + (j & 3);
        a2e[j] = bf16_bits(key == 2 * r + 1 ? 1.f : 0.f);
    }
    for (int w = 0; w < 4; ++w) a2[w] = a2e[2 * w] | ((uint32_t)a2e[2 * w + 1] << 16);
    const f32x4 y = mfma_16x16x32(__builtin_bit_cast(bf16x8, a2), __builtin_bit_cast(bf16x8, pb), f32x4{0.f, 0.f, 0.f, 0.f});
    u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    const f32x4 cs = mfma_16x16x32(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, pb), f32x4{0.f, 0.f, 0.f, 0.f});
    for (int i = 0; i < 4; ++i) out[(4 * h4 + i) * 32 + r] = y[i];              // rows 0..15, cols 0..15: X[2i+1][j]
    for (int i = 0; i < 4; ++i) out[(4 * h4 + i) * 32 + 16 + r] = cs[i];        // cols 16..31: column sums (every row the same)
    const float v = (float)(lane * 3 % 64);                                     // distinct per quarter
    out[(16 + h4) * 32 + r] = max_all_quarters(v);                              // rows 16..19
    out[(20 + h4) * 32 + r] = sum_all_quarters(v);                              // rows 20..23
    for (int i = 24; i < 32; ++i) out[i * 32 + lane % 32] = 0.f;
    for (int i = 16; i < 24; ++i) out[i * 32 + 16 + r] = 0.f;
}

static void test_mfma_layouts() {
    float* d;
    HIP_CHECK(hipMalloc(&d, 32 * 32 * 4));
    std::vector<float> h(32 * 32);
    auto run = [&](const char* name, void (*kern)(float*), auto expect) {
        HIP_CHECK(hipMemset(d, 0xff, 32 * 32 * 4));
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, nullptr, d);
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(h.data(), d, 32 * 32 * 4, hipMemcpyDeviceToHost));
        long long bad = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) bad += !(h[i * 32 + j] == expect(i, j));
        report(name, bad, 32 * 32);
    };
    run("mfma layout: bf16 32x32x16, A = [I|0], B[k][j] = 8k + j", mfma_bf16_layout,
        [](int i, int j) { return i < 16 ? (float)(8 * i + j) : 0.f; });
    run("mfma layout: fp8 e4m3 32x32x16, A = [I|0], B[k][j] = (k + 3j) % 16", mfma_fp8_layout,
        [](int i, int j) { return i < 16 ? (float)((i + 3 * j) % 16) : 0.f; });
    run("mfma layout: MX 32x32x64 f8f6f4 unit scales, A = I, B[k][j] = (k + 3j) % 16", mfma_mx_layout,
        [](int i, int j) { return (float)((i + 3 * j) % 16); });
    run("mfma layout: accumulator tile as the next B operand (P^T of the P.V product)", mfma_acc_as_operand,
        [](int i, int j) { return i < 16 ? (float)(8 * i + j) : 0.f; });
    run("mfma layout: bf16 16x16x32, A = [I|0], B[k][j] = 4k + j", mfma16_bf16_layout,
        [](int i, int j) { return (i < 16 && j < 16) ? (float)(4 * i + j) : 0.f; });
    run("mfma layout: 16x16x32 accumulators as the next B operand, ONES.P^T column sums, 4-quarter reductions", mfma16_acc_as_operand,
        [](int i, int j) {
            if (i < 16 && j < 16) return (float)(4 * (2 * i + 1) + j);                       // X[2i+1][j]
            if (i < 16) { float s = 0; for (int k = 0; k < 32; ++k) s += 4 * k + (j - 16); return s; }   // sum over 32 keys
            if (i < 24 && j < 16) {
                float mx = 0, sm = 0;
                for (int q = 0; q < 4; ++q) { const float v = (float)((16 * q + j) * 3 % 64); mx = v > mx ? v : mx; sm += v; }
                return i < 20 ? mx : sm;
            }
            return 0.f;
        });
    HIP_CHECK(hipFree(d));
}

// ------------------------------------------------------------------------------------------------ (2) LDS images
template <int D, bool DMA>
__global__ __launch_bounds__(512) void lds_image_kernel(const uint16_t* K, const uint16_t* V, int S, uint16_t* img, uint16_t* kfr,
                                                        uint16_t* vfr) {
    using C = KernelCfg<D, false, __bf16, 2, Opt{.pad = !DMA, .m16 = 0}>;   // the 32x32x16 engine; DMA: LDS-DMA staging (production)
    static_assert(C::DMA == DMA);
    using G = TileGeom<D, 2>;
    using W = WaveCompute<C>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_ptr smem = (lds_ptr)smem_raw;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    typename W::Stage st;
    // the ring starts out full of NaN patterns: rows past S must arrive as zeros whatever LDS held before (a masked key's weight is 0,
    // and 0 x NaN would still poison O)
    for (int i = threadIdx.x; i < G::SLOT / 4; i += blockDim.x) *reinterpret_cast<FA_LDS uint32_t*>(smem + 4 * i) = 0xffffffffu;
    __syncthreads();
    st.init((const char*)K, (const char*)V, D * 2, D * 2, S, wave, lane);
    st.load_all_into(0, smem);
    st.write_all(smem);
    st.wait_all();
    __syncthreads();
    for (int i = threadIdx.x; i < G::SLOT / 2; i += blockDim.x) img[i] = *reinterpret_cast<FA_LDS uint16_t*>(smem + 2 * i);
    if (wave == 0) {
        W w;
        const int kbase = DMA ? kd_read_base(lane, G::KBLK) : k_read_base(lane), vbase = v_read_base(lane);
        for (int f = 0; f < W::NF; ++f) {
            const u32x4 kf = w.k_read(smem, kbase, f);
            for (int e = 0; e < 4; ++e) { kfr[(f * 64 + lane) * 8 + 2 * e] = kf[e] & 0xffff; kfr[(f * 64 + lane) * 8 + 2 * e + 1] = kf[e] >> 16; }
        }
        for (int v = 0; v < W::NB; ++v) {
            const u32x4 vf = __builtin_bit_cast(u32x4, w.v_frag(smem + G::K_TILE, vbase, v / W::DB, v % W::DB));
            for (int e = 0; e < 4; ++e) { vfr[(v * 64 + lane) * 8 + 2 * e] = vf[e] & 0xffff; vfr[(v * 64 + lane) * 8 + 2 * e + 1] = vf[e] >> 16; }
        }
    }
}

template <int D, bool DMA>
static void test_lds_image(const int S = 64) {
    using G = TileGeom<D, 2>;
    using W = WaveCompute<KernelCfg<D, false, __bf16, 2, Opt{.pad = !DMA, .m16 = 0}>>;
    std::vector<uint16_t> hk(64 * D, 0), hv(64 * D, 0);   // rows S .. 63 do not exist: expected 0
    for (int i = 0; i < S * D; ++i) { hk[i] = (uint16_t)(i + 1); hv[i] = (uint16_t)(0x8000 + i + 1); }   // "i + 1", V tagged
    uint16_t *dk, *dv, *dimg, *dkf, *dvf;
    HIP_CHECK(hipMalloc(&dk, S * D * 2)); HIP_CHECK(hipMalloc(&dv, S * D * 2));
    HIP_CHECK(hipMalloc(&dimg, G::SLOT)); HIP_CHECK(hipMalloc(&dkf, W::NF * 64 * 8 * 2)); HIP_CHECK(hipMalloc(&dvf, W::NB * 64 * 8 * 2));
    HIP_CHECK(hipMemcpy(dk, hk.data(), S * D * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dv, hv.data(), S * D * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(dimg, 0, G::SLOT));
    HIP_CHECK(hipFuncSetAttribute((const void*)lds_image_kernel<D, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * G::SLOT));
    hipLaunchKernelGGL((lds_image_kernel<D, DMA>), dim3(1), dim3(512), 3 * G::SLOT, nullptr, dk, dv, S, dimg, dkf, dvf);
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<uint16_t> img(G::SLOT / 2), kf(W::NF * 64 * 8), vf(W::NB * 64 * 8);
    HIP_CHECK(hipMemcpy(img.data(), dimg, G::SLOT, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(kf.data(), dkf, kf.size() * 2, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(vf.data(), dvf, vf.size() * 2, hipMemcpyDeviceToHost));
    char name[220];
    // raw image == the documented permutation of the i + 1 tile
    long long bad = 0;
    for (int key = 0; key < 64; ++key)
        for (int ch = 0; ch < D / 8; ++ch)
            for (int e = 0; e < 8; ++e) {
                bad += img[(DMA ? G::kd_lds_off(key, ch, true) : G::k_lds_off(key, ch)) / 2 + e] != hk[key * D + 8 * ch + e];
                bad += img[(G::K_TILE + G::v_lds_off(key, ch)) / 2 + e] != hv[key * D + 8 * ch + e];
            }
    snprintf(name, sizeof name, "lds image d=%d%s: K %s + V [key/8][d/32][key%%8][d%%32] == permutation of the i+1 tile", D,
             DMA ? " (LDS-DMA)" : "", DMA ? "[key/8][chunk^blk][key%8]" : "chunk-major");
    report(name, bad, 2LL * 64 * D);
    // K fragment f = (32-key half kt = f / FPH, k-step u = f % FPH): lane (r, h), element j = K[32kt + r][16u + 8h + j]
    bad = 0;
    for (int f = 0; f < W::NF; ++f)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int kt = f / W::FPH, u = f % W::FPH, r = lane & 31, h = lane >> 5;
                bad += kf[(f * 64 + lane) * 8 + j] != hk[(32 * kt + r) * D + 16 * u + 8 * h + j];
            }
    snprintf(name, sizeof name, "lds image d=%d%s: K A-fragments (ds_read_b128) hold K[32kt+r][16u+8h+j]", D, DMA ? " (LDS-DMA)" : "");
    report(name, bad, (long long)W::NF * 64 * 8);
    // V^T fragment v = (16-key step s4 = v / DB, d block db = v % DB): lane (r, h), element j = V[16s4 + 8(j>>2) + 4h + (j&3)][32db + r]
    bad = 0;
    for (int v = 0; v < W::NB; ++v)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int s4 = v / W::DB, db = v % W::DB, r = lane & 31, h = lane >> 5;
                const int key = 16 * s4 + 8 * (j >> 2) + 4 * h + (j & 3);
                bad += vf[(v * 64 + lane) * 8 + j] != hv[key * D + 32 * db + r];
            }
    snprintf(name, sizeof name, "lds image d=%d%s: V^T A-fragments (ds_read_b64_tr_b16) hold V[16s4+8(j>>2)+4h+(j&3)][32db+r]", D, DMA ? " (LDS-DMA)" : "");
    report(name, bad, (long long)W::NB * 64 * 8);
    HIP_CHECK(hipFree(dk)); HIP_CHECK(hipFree(dv)); HIP_CHECK(hipFree(dimg)); HIP_CHECK(hipFree(dkf)); HIP_CHECK(hipFree(dvf));
}

// The same for the 16x16x32 engine (computers16.hip.h): V image [key/8][d/16][key%8][d%16]; K fragment f = (key group kg = f / KS,
// k-step ks = f % KS): lane (r = l&15, h4 = l>>4), element j = K[16kg + r][32ks + 8h4 + j]; V^T fragment v = (k-step kk = v / DG,
// d group dg = v % DG): element j = V[32kk + 16(j>>2) + 4h4 + (j&3)][16dg + r].
template <int D, bool DMA>
__global__ __launch_bounds__(512) void lds_image16_kernel(const uint16_t* K, const uint16_t* V, int S, uint16_t* img, uint16_t* kfr,
                                                          uint16_t* vfr) {
    using C = KernelCfg<D, false, __bf16, 2, Opt{.pad = !DMA, .m16 = 1}>;
    static_assert(C::M16 && C::DMA == DMA);
    using G = TileGeom<D, 2>;
    using W = WaveCompute16<C>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_ptr smem = (lds_ptr)smem_raw;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    typename W::Stage st;
    // the ring starts out full of NaN patterns: rows past S must arrive as zeros whatever LDS held before (a masked key's weight is 0,
    // and 0 x NaN would still poison O)
    for (int i = threadIdx.x; i < G::SLOT / 4; i += blockDim.x) *reinterpret_cast<FA_LDS uint32_t*>(smem + 4 * i) = 0xffffffffu;
    __syncthreads();
    st.init((const char*)K, (const char*)V, D * 2, D * 2, S, wave, lane);
    st.load_all_into(0, smem);
    st.write_all(smem);
    st.wait_all();
    __syncthreads();
    for (int i = threadIdx.x; i < G::SLOT / 2; i += blockDim.x) img[i] = *reinterpret_cast<FA_LDS uint16_t*>(smem + 2 * i);
    if (wave == 0) {
        W w;
        const int kbase = DMA ? kd16_read_base(lane, G::KBLK) : k16_read_base(lane), vbase = v16_read_base<D>(lane);
        for (int f = 0; f < W::NF; ++f) {
            const u32x4 kf = w.k_read(smem, kbase, f);
            for (int e = 0; e < 4; ++e) { kfr[(f * 64 + lane) * 8 + 2 * e] = kf[e] & 0xffff; kfr[(f * 64 + lane) * 8 + 2 * e + 1] = kf[e] >> 16; }
        }
        for (int v = 0; v < W::NV; ++v) {
            const u32x4 vf = __builtin_bit_cast(u32x4, w.v_frag(smem + G::K_TILE, vbase, v / W::DG, v % W::DG));
            for (int e = 0; e < 4; ++e) { vfr[(v * 64 + lane) * 8 + 2 * e] = vf[e] & 0xffff; vfr[(v * 64 + lane) * 8 + 2 * e + 1] = vf[e] >> 16; }
        }
    }
}

template <int D, bool DMA>
static void test_lds_image16(const int S = 64) {
    using G = TileGeom<D, 2>;
    using W = WaveCompute16<KernelCfg<D, false, __bf16, 2, Opt{.pad = !DMA, .m16 = 1}>>;
    std::vector<uint16_t> hk(64 * D, 0), hv(64 * D, 0);   // rows S .. 63 do not exist: expected 0
    for (int i = 0; i < S * D; ++i) { hk[i] = (uint16_t)(i + 1); hv[i] = (uint16_t)(0x8000 + i + 1); }
    uint16_t *dk, *dv, *dimg, *dkf, *dvf;
    HIP_CHECK(hipMalloc(&dk, S * D * 2)); HIP_CHECK(hipMalloc(&dv, S * D * 2));
    HIP_CHECK(hipMalloc(&dimg, G::SLOT)); HIP_CHECK(hipMalloc(&dkf, W::NF * 64 * 8 * 2)); HIP_CHECK(hipMalloc(&dvf, W::NV * 64 * 8 * 2));
    HIP_CHECK(hipMemcpy(dk, hk.data(), S * D * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dv, hv.data(), S * D * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(dimg, 0, G::SLOT));
    HIP_CHECK(hipFuncSetAttribute((const void*)lds_image16_kernel<D, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * G::SLOT));
    hipLaunchKernelGGL((lds_image16_kernel<D, DMA>), dim3(1), dim3(512), 3 * G::SLOT, nullptr, dk, dv, S, dimg, dkf, dvf);
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<uint16_t> img(G::SLOT / 2), kf(W::NF * 64 * 8), vf(W::NV * 64 * 8);
    HIP_CHECK(hipMemcpy(img.data(), dimg, G::SLOT, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(kf.data(), dkf, kf.size() * 2, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(vf.data(), dvf, vf.size() * 2, hipMemcpyDeviceToHost));
    char name[240];
    long long bad = 0;
    for (int key = 0; key < 64; ++key)
        for (int ch = 0; ch < D / 8; ++ch)
            for (int e = 0; e < 8; ++e) {
                bad += img[(DMA ? G::kd_lds_off(key, ch, false) : G::k_lds_off(key, ch)) / 2 + e] != hk[key * D + 8 * ch + e];
                bad += img[(G::K_TILE + G::v16_lds_off(key, ch)) / 2 + e] != hv[key * D + 8 * ch + e];
            }
    snprintf(name, sizeof name, "lds image (16x16x32 engine) d=%d%s: K %s + V [key/8][d/16][key%%8][d%%16] == permutation of the i+1 tile", D,
             DMA ? " (LDS-DMA)" : "", DMA ? "[key/8][chunk][key%8]" : "chunk-major");
    report(name, bad, 2LL * 64 * D);
    bad = 0;
    for (int f = 0; f < W::NF; ++f)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int kg = f / W::KS, ks = f % W::KS, r = lane & 15, h4 = lane >> 4;
                bad += kf[(f * 64 + lane) * 8 + j] != hk[(16 * kg + r) * D + 32 * ks + 8 * h4 + j];
            }
    snprintf(name, sizeof name, "lds image (16x16x32 engine) d=%d%s: K A-fragments hold K[16kg+r][32ks+8h4+j]", D, DMA ? " (LDS-DMA)" : "");
    report(name, bad, (long long)W::NF * 64 * 8);
    bad = 0;
    for (int v = 0; v < W::NV; ++v)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int kk = v / W::DG, dg = v % W::DG, r = lane & 15, h4 = lane >> 4;
                const int key = 32 * kk + 16 * (j >> 2) + 4 * h4 + (j & 3);
                bad += vf[(v * 64 + lane) * 8 + j] != hv[key * D + 16 * dg + r];
            }
    snprintf(name, sizeof name, "lds image (16x16x32 engine) d=%d%s: V^T A-fragments hold V[32kk+16(j>>2)+4h4+(j&3)][16dg+r]", D, DMA ? " (LDS-DMA)" : "");
    report(name, bad, (long long)W::NV * 64 * 8);
    HIP_CHECK(hipFree(dk)); HIP_CHECK(hipFree(dv)); HIP_CHECK(hipFree(dimg)); HIP_CHECK(hipFree(dkf)); HIP_CHECK(hipFree(dvf));
}

// fp8 inputs (e4m3 bytes, d = 128): the K image and K fragments of the 32x32x16 engine, K by LDS-DMA (production, HybridStageFp8) or through
// registers.  Fragment f = (32-key half kt = f / FPH, 16-byte piece u = f % FPH): lane (r, h) holds bytes K[32kt + r][16(2u + h) .. +15].
// (V is widened to bf16 on its way into LDS: covered end to end by the fp8 parity tests.)
template <bool DMA>
__global__ __launch_bounds__(512) void lds_image_fp8_kernel(const uint8_t* K, const uint8_t* V, int S, uint8_t* kfr) {
    using C = KernelCfg<128, false, __bf16, 1, Opt{.pad = !DMA}>;
    static_assert(C::DMA_K8 == DMA && !C::M16);
    using G = TileGeom<128, 1>;
    using W = WaveCompute<C>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_ptr smem = (lds_ptr)smem_raw;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < G::SLOT / 4; i += blockDim.x) *reinterpret_cast<FA_LDS uint32_t*>(smem + 4 * i) = 0xffffffffu;
    __syncthreads();
    typename W::Stage st;
    st.init((const char*)K, (const char*)V, 128, 128, S, wave, lane);
    st.load_all_into(0, smem);
    st.write_all(smem);
    st.wait_all();
    __syncthreads();
    if (wave == 0) {
        W w;
        const int kbase = DMA ? kd_read_base(lane, G::KBLK) : k_read_base(lane);
        for (int f = 0; f < W::NF; ++f) {
            const u32x4 kf = w.k_read(smem, kbase, f);
            for (int e = 0; e < 4; ++e)
                for (int b = 0; b < 4; ++b) kfr[(f * 64 + lane) * 16 + 4 * e + b] = (kf[e] >> (8 * b)) & 0xff;
        }
    }
}

template <bool DMA>
static void test_lds_image_fp8(const int S = 64) {
    using G = TileGeom<128, 1>;
    using W = WaveCompute<KernelCfg<128, false, __bf16, 1, Opt{.pad = !DMA}>>;
    constexpr int D = 128;
    std::vector<uint8_t> hk(64 * D, 0), hv(S * D, 0);     // rows S .. 63 do not exist: expected 0; e4m3 codes below 0x78 (no NaN / inf patterns)
    for (int i = 0; i < S * D; ++i) hk[i] = (uint8_t)((i * 7 + 1) % 0x77);
    uint8_t *dk, *dv, *dkf;
    HIP_CHECK(hipMalloc(&dk, S * D)); HIP_CHECK(hipMalloc(&dv, S * D)); HIP_CHECK(hipMalloc(&dkf, W::NF * 64 * 16));
    HIP_CHECK(hipMemcpy(dk, hk.data(), S * D, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dv, hv.data(), S * D, hipMemcpyHostToDevice));
    HIP_CHECK(hipFuncSetAttribute((const void*)lds_image_fp8_kernel<DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * G::SLOT));
    hipLaunchKernelGGL((lds_image_fp8_kernel<DMA>), dim3(1), dim3(512), 3 * G::SLOT, nullptr, dk, dv, S, dkf);
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<uint8_t> kf(W::NF * 64 * 16);
    HIP_CHECK(hipMemcpy(kf.data(), dkf, kf.size(), hipMemcpyDeviceToHost));
    long long bad = 0;
    for (int f = 0; f < W::NF; ++f)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 16; ++j) {
                const int kt = f / W::FPH, u = f % W::FPH, r = lane & 31, h = lane >> 5;
                bad += kf[(f * 64 + lane) * 16 + j] != hk[(32 * kt + r) * D + 16 * (2 * u + h) + j];
            }
    char name[200];
    snprintf(name, sizeof name, "lds image fp8 d=128%s, S=%d: K A-fragments hold bytes K[32kt+r][16(2u+h)+j]", DMA ? " (K by LDS-DMA)" : "", S);
    report(name, bad, (long long)W::NF * 64 * 16);
    HIP_CHECK(hipFree(dk)); HIP_CHECK(hipFree(dv)); HIP_CHECK(hipFree(dkf));
}

// ------------------------------------------------------------------------------------------------ (3) fp8 MFMA accumulation
// How exactly does gfx950's fp8 MFMA sum its products?  (A measurement, not a pass/fail test: the fp8 LSE bound of
// tests/test_fuzz_slice.py is derived from the figure printed here.)  One wave computes D = A.B over K = 128 as the kernel
// does -- two chained v_mfma_scale_f32_32x32x64_f8f6f4 with unit scales -- on e4m3fn values drawn as round(N(0, sigma));
// the host forms the exact sums in double.  Printed: max over the 32x32 outputs and all trials of |D - exact| / sum_k |a_k b_k|
// (the natural scale of a summation error) for the fp8 instruction, and for the bf16 32x32x16 instruction on the SAME values.
__global__ void fp8_accum_probe(const uint8_t* A, const uint8_t* B, float* out_mx, float* out_bf16) {
    // A: [32 rows][128 k] e4m3 bytes, B: [128 k][32 cols] e4m3 bytes (row-major)
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    for (int step = 0; step < 2; ++step) {
        uint8_t ab[32], bb[32];
        for (int j = 0; j < 32; ++j) {
            const int k = 64 * step + 32 * h + j;
            ab[j] = A[r * 128 + k];
            bb[j] = B[k * 32 + r];
        }
        u32x4 alo, ahi, blo, bhi;
        for (int w = 0; w < 4; ++w) {
            alo[w] = ab[4 * w] | (ab[4 * w + 1] << 8) | (ab[4 * w + 2] << 16) | ((uint32_t)ab[4 * w + 3] << 24);
            ahi[w] = ab[16 + 4 * w] | (ab[17 + 4 * w] << 8) | (ab[18 + 4 * w] << 16) | ((uint32_t)ab[19 + 4 * w] << 24);
            blo[w] = bb[4 * w] | (bb[4 * w + 1] << 8) | (bb[4 * w + 2] << 16) | ((uint32_t)bb[4 * w + 3] << 24);
            bhi[w] = bb[16 + 4 * w] | (bb[17 + 4 * w] << 8) | (bb[18 + 4 * w] << 16) | ((uint32_t)bb[19 + 4 * w] << 24);
        }
        c = mfma_32x32x64_fp8_unit_scale(alo, ahi, blo, bhi, c);
    }
    for (int i = 0; i < 16; ++i) out_mx[acc_row(i, h) * 32 + r] = c[i];
    // the same values, widened exactly to bf16, through 8 chained 32x32x16 bf16 MFMAs
    f32x16 cb;
    for (int i = 0; i < 16; ++i) cb[i] = 0.f;
    for (int step = 0; step < 8; ++step) {
        uint32_t aw[4], bw[4];
        for (int w = 0; w < 4; ++w) {
            const int k0 = 16 * step + 8 * h + 2 * w;
            const u32x4 a4 = fp8x8_to_bf16x8(A[r * 128 + k0] | ((uint32_t)A[r * 128 + k0 + 1] << 8), 0);
            const u32x4 b4 = fp8x8_to_bf16x8(B[k0 * 32 + r] | ((uint32_t)B[(k0 + 1) * 32 + r] << 8), 0);
            aw[w] = a4[0];
            bw[w] = b4[0];
        }
        u32x4 a = {aw[0], aw[1], aw[2], aw[3]}, b = {bw[0], bw[1], bw[2], bw[3]};
        cb = mfma_32x32x16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), cb);
    }
    for (int i = 0; i < 16; ++i) out_bf16[acc_row(i, h) * 32 + r] = cb[i];
}

static float e4m3fn_decode(uint8_t b) {
    const int sgn = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
    return sgn ? -v : v;
}
static uint8_t e4m3fn_encode_nearest(float x) {   // brute force over the 254 finite codes
    uint8_t best = 0;
    float bd = 1e30f;
    for (int c = 0; c < 256; ++c) {
        if ((c & 0x7f) == 0x7f) continue;
        const float d = std::fabs(e4m3fn_decode((uint8_t)c) - x);
        if (d < bd) { bd = d; best = (uint8_t)c; }
    }
    return best;
}

static void measure_fp8_accumulation() {
    uint8_t *dA, *dB;
    float *dmx, *dbf;
    HIP_CHECK(hipMalloc(&dA, 32 * 128)); HIP_CHECK(hipMalloc(&dB, 128 * 32));
    HIP_CHECK(hipMalloc(&dmx, 32 * 32 * 4)); HIP_CHECK(hipMalloc(&dbf, 32 * 32 * 4));
    std::vector<uint8_t> hA(32 * 128), hB(128 * 32);
    std::vector<float> mx(32 * 32), bf(32 * 32);
    uint64_t state = 0x1234567ull;
    auto rnd = [&]() { state = state * 6364136223846793005ull + 1442695040888963407ull; return (double)(state >> 11) / 9007199254740992.0; };
    auto gauss = [&]() { return std::sqrt(-2.0 * std::log(rnd() + 1e-300)) * std::cos(6.283185307179586 * rnd()); };
    // kreal < 128: only the first kreal contraction elements are non-zero (a head dimension padded onto the 128-wide kernel)
    for (int kreal : {128, 32})
    for (double sigma : {1.0, 3.0, 12.0}) {
        double eps_sum = 0, eps_max = 0, eps_bf = 0;
        for (int trial = 0; trial < 40; ++trial) {
            for (int i = 0; i < 32; ++i)
                for (int k = 0; k < 128; ++k) {
                    hA[i * 128 + k] = k < kreal ? e4m3fn_encode_nearest((float)(sigma * gauss())) : 0;
                    hB[k * 32 + i] = k < kreal ? e4m3fn_encode_nearest((float)(sigma * gauss())) : 0;
                }
            HIP_CHECK(hipMemcpy(dA, hA.data(), hA.size(), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(dB, hB.data(), hB.size(), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(fp8_accum_probe, dim3(1), dim3(64), 0, nullptr, dA, dB, dmx, dbf);
            HIP_CHECK(hipDeviceSynchronize());
            HIP_CHECK(hipMemcpy(mx.data(), dmx, mx.size() * 4, hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(bf.data(), dbf, bf.size() * 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double ex = 0, ab = 0, pmax = 0;
                    for (int k = 0; k < 128; ++k) {
                        const double pr = (double)e4m3fn_decode(hA[i * 128 + k]) * (double)e4m3fn_decode(hB[k * 32 + j]);
                        ex += pr;
                        ab += std::fabs(pr);
                        pmax = std::max(pmax, std::fabs(pr));
                    }
                    if (pmax == 0) continue;
                    eps_sum = std::max(eps_sum, std::fabs(mx[i * 32 + j] - ex) / ab);
                    eps_max = std::max(eps_max, std::fabs(mx[i * 32 + j] - ex) / pmax);
                    eps_bf = std::max(eps_bf, std::fabs(bf[i * 32 + j] - ex) / ab);
                }
        }
        printf("MEASURE fp8 accumulation, K=128 (%3d non-zero), sigma=%4.1f: MX fp8 MFMA max |D-exact| / sum_k|a_k b_k| = %.3e (2^%.2f);  "
               "/ max_k|a_k b_k| = %.3e (2^%.2f);  bf16 MFMA on the same values / sum|ab| = %.3e\n",
               kreal, sigma, eps_sum, std::log2(eps_sum), eps_max, std::log2(eps_max), eps_bf);
    }
    HIP_CHECK(hipFree(dA)); HIP_CHECK(hipFree(dB)); HIP_CHECK(hipFree(dmx)); HIP_CHECK(hipFree(dbf));
}

int main() {
    test_mfma_layouts();
    test_lds_image<128, false>();
    test_lds_image<64, false>();
    test_lds_image16<128, false>();
    test_lds_image16<64, false>();
    test_lds_image<128, true>();     // the LDS-DMA form of both engines' images (production staging for bf16)
    test_lds_image<64, true>();
    test_lds_image16<128, true>();
    test_lds_image16<64, true>();
    printf("ragged tile (S = 41 of 64 keys; LDS prefilled with NaN patterns):\n");
    test_lds_image<128, false>(41);
    test_lds_image<128, true>(41);
    test_lds_image16<128, false>(41);
    test_lds_image16<128, true>(41);
    test_lds_image16<64, true>(41);
    test_lds_image_fp8<false>();
    test_lds_image_fp8<true>();
    test_lds_image_fp8<true>(41);
    measure_fp8_accumulation();
    printf("%d test(s) failed\n", g_fail);
    return g_fail;
}
