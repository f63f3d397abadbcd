// FlashAttention.hip -- kernel entries and the C-ABI launcher (include/flash_attention.h).
//
// Counterpart of the reference's kernels/FlashAttention.cuh:59-84 (kernel entry: role split into
// compute warps + two loader warps around four cuda::pipeline objects) and of the launch code in
// tests/main.cu:51-64.  Written for gfx950 only: hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/flash_attention.h"
#include "../helpers.hpp"
#include "launchers.hip.h"
#include "generic.hip.h"
#include "weights.hip.h"

namespace fa {

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int fill_params(Params& p, const void* Q, const void* K, const void* V, void* O, float* lse, int B, int H,
                       int S, int Sk, int d, float scale, const fa_strides* sQ, const fa_strides* sK,
                       const fa_strides* sV, const fa_strides* sO) {
    p.Q = Q; p.K = K; p.V = V; p.O = O; p.lse = lse;
    const int64_t dS = d, dH = (int64_t)S * d, dB = (int64_t)H * S * d;      // dense Q / O
    const int64_t kH = (int64_t)Sk * d, kB = (int64_t)H * Sk * d;            // dense K / V
    p.qB = sQ ? sQ->strideB : dB; p.qH = sQ ? sQ->strideH : dH; p.qS = sQ ? sQ->strideS : dS;
    p.kB = sK ? sK->strideB : kB; p.kH = sK ? sK->strideH : kH; p.kS = sK ? sK->strideS : dS;
    p.vB = sV ? sV->strideB : kB; p.vH = sV ? sV->strideH : kH; p.vS = sV ? sV->strideS : dS;
    p.oB = sO ? sO->strideB : dB; p.oH = sO ? sO->strideH : dH; p.oS = sO ? sO->strideS : dS;
    p.B = B; p.H = H; p.S = S; p.Sk = Sk; p.d = d;
    p.scale = scale;
    p.scale_log2 = scale * 1.4426950408889634f;
    return FA_OK;
}

static int elem_size(int dtype) {
    switch (dtype) {
        case FA_DTYPE_F32: return 4;
        case FA_DTYPE_BF16: case FA_DTYPE_F16: return 2;
        case FA_DTYPE_FP8_E4M3: return 1;
        default: return 0;
    }
}

static bool strides_ok(const fa_strides* s, int esz, int d) {
    if (!s) return true;
    if (s->strideS < d || s->strideB < 0 || s->strideH < 0) return false;
    return (s->strideS * esz) % 16 == 0 && (s->strideH * esz) % 16 == 0 && (s->strideB * esz) % 16 == 0;
}

static int validate(const void* Q, const void* K, const void* V, void* O, int B, int H, int S, int d,
                    float scale, int dtype, int o_dtype) {
    if (!Q || !K || !V || !O) return FA_ERR_NULL_POINTER;
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O)) return FA_ERR_MISALIGNED;
    if (B <= 0 || H <= 0 || S <= 0 || d <= 0) return FA_ERR_BAD_SHAPE;
    if ((int64_t)B * H > INT32_MAX / 2 || S > (1 << 24)) return FA_ERR_BAD_SHAPE;
    if (!std::isfinite(scale)) return FA_ERR_BAD_SCALE;
    if (dtype != FA_DTYPE_F32 && dtype != FA_DTYPE_BF16 && dtype != FA_DTYPE_FP8_E4M3) return FA_ERR_UNSUPPORTED_DTYPE;
    if (o_dtype != FA_DTYPE_F32 && o_dtype != FA_DTYPE_BF16 && o_dtype != FA_DTYPE_F16) return FA_ERR_UNSUPPORTED_DTYPE;
    if (d > 256) return FA_ERR_UNSUPPORTED_DHEAD;
    if (dtype == FA_DTYPE_FP8_E4M3 && d > 128) return FA_ERR_UNSUPPORTED_DHEAD;   // fp8: MFMA path only ...
    if (dtype == FA_DTYPE_FP8_E4M3 && !(scale > 0.f)) return FA_ERR_BAD_SCALE;    // ... which folds a positive scale into exp2
    if ((d * elem_size(dtype)) % 16 != 0 || (d * elem_size(o_dtype)) % 16 != 0) return FA_ERR_UNSUPPORTED_DHEAD;
    return FA_OK;
}

// Compute units of the current device (persistent grid = one workgroup per CU).  256 on MI355X, which is
// also the answer when no device is visible (flash_attention_plan is callable on a build machine).
static int device_cus() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// bf16 inputs: how many leading query blocks of every head run the fp16-weights kernel (the rest: the bf16-weights kernel).
// bf16 weights carry 2^-9 of relative rounding error each; summed over a row's keys it averages out, but a row that sees few
// keys keeps most of it: at FA_EARLY_KEYS = 1024 visible keys the worst element error measured over 40 random heads is 0.74 of
// the stated tolerance 1e-3 + 1e-3|ref| (0.90 at 256, 1.33 at 64: those rows miss it).  Default: a row that can see fewer than
// FA_EARLY_KEYS keys -- under the causal mask the rows q < FA_EARLY_KEYS, and every row when seqLenK < FA_EARLY_KEYS -- gets
// fp16 weights (11 significant bits; 8 x smaller errors), whole query blocks at a time.
static int early_q_blocks(int S, int Sk, int d, bool causal, unsigned flags, int q_block_rows) {
    const int nQ = getNumCta(S, q_block_rows);
    if (!(d == 64 || d == 128) || (flags & FA_FLAG_BF16_WEIGHTS)) return 0;   // (padded head dimensions have no fp16-weights kernel)
    if (flags & FA_FLAG_F16_WEIGHTS) return nQ;
    if (Sk < FA_EARLY_KEYS) return nQ;
    return causal ? std::min(nQ, FA_EARLY_KEYS / q_block_rows) : 0;
}

// Small problems take the pair kernel (kernel_bf16.hip.h: fwd_mfma_pair_kernel): 128-row units, one per workgroup of four waves.
// Causal, D = 64: two workgroups fit a CU -- at most one 256-row unit per CU (where the launch would last as long as its heaviest
// unit), and every XCD group's 128-row units within its two dispatch rounds.  Causal D = 128, and any D without the mask (equal units:
// nothing to pair): one workgroup per CU -- at most one 256-row unit per TWO CUs (half of the chip would idle), every group's 128-row
// units within one round.
static bool pair_kernel_applies(int B, int H, int S, int d, bool causal, int dtype, float scale) {
    if (!(dtype == FA_DTYPE_BF16 && (d == 64 || d == 128) && scale > 0.f)) return false;
    const int64_t heads = (int64_t)B * H;
    const int cus = device_cus(), jpx = cus / 8, rounds = (causal && d == 64) ? 2 : 1;
    if (heads * getNumCta(S, 256) * 2 > (int64_t)cus * rounds) return false;
    return ((heads + 7) / 8) * getNumCta(S, 128) <= rounds * jpx;
}

static int make_plan(int B, int H, int S, int d, bool causal, int dtype, int o_dtype, float scale,
                     fa_launch_plan* plan) {
    // bf16: d in {64,128} natively; any other multiple of 8 up to 128 runs the next larger instantiation with its
    // rows zero-padded on the fly (d/64 or d/128 of the MFMA work is useful -- still ~1000x the VALU kernel)
    const bool mfma_bf16 = dtype == FA_DTYPE_BF16 && d % 8 == 0 && d <= 128 && scale > 0.f;
    const bool mfma_fp8 = dtype == FA_DTYPE_FP8_E4M3 && d % 16 == 0 && d <= 128 && scale > 0.f;   // d < 128: zero-padded
    // fp32: d in {64,128} natively, other multiples of 4 up to 128 zero-padded onto the next larger instantiation
    const bool mfma_f32 = dtype == FA_DTYPE_F32 && d % 4 == 0 && d <= 128 && scale > 0.f;
    if (mfma_f32) {
        plan->kernel_id = 3;
        plan->q_block_rows = calculateSizeBlockQ(d, dtype);
        plan->kv_block_rows = calculateSizeBlockKV(d, dtype);
        plan->threads = 256;
        plan->lds_bytes = f32_lds_bytes(d > 64 ? 128 : 64);
        const int nQ = getNumCta(S, plan->q_block_rows);
        const int64_t units = (int64_t)B * H * nQ;
        plan->grid = (int)(8 * ((units + 7) / 8));
    } else if (mfma_bf16 || mfma_fp8) {
        plan->kernel_id = mfma_fp8 ? 2 : 1;
        plan->q_block_rows = calculateSizeBlockQ(d, dtype);
        plan->kv_block_rows = calculateSizeBlockKV(d, dtype);
        plan->threads = 512;
        // 3-slot ring of [K image (input type) | V image (bf16)] at the instantiated head dimension, or ring slot 0 + the
        // epilogue's staging regions where those are larger -- as carved up by the instantiation this problem launches (the
        // engine depends on the mask, the staging form on padding: kernel_bf16.hip.h, KernelCfg::LDS_BYTES)
        const int dk = mfma_fp8 ? 128 : paddedDHead(d);
        const bool pad = d != dk;
        plan->lds_bytes = mfma_fp8 ? fp8_d128_lds_bytes(causal, pad, o_dtype)
                                   : (dk == 128 ? bf16_d128_lds_bytes(causal, pad, o_dtype) : bf16_d64_lds_bytes(causal, pad, o_dtype));
        // persistent grid: one workgroup per CU (8 XCD groups x CUs/8), each walking ceil(units/grid) units;
        // with fewer units than CUs, one workgroup per unit
        const int nQ = getNumCta(S, plan->q_block_rows);
        const int64_t units = (int64_t)B * H * nQ;
        plan->grid = (int)(8 * std::min<int64_t>((units + 7) / 8, device_cus() / 8));
        if (mfma_bf16 && pair_kernel_applies(B, H, S, d, causal, dtype, scale)) {
            // 128-row units, one per workgroup of four waves, two workgroups per CU (grid = 16 per XCD-group workgroup slot pair)
            plan->q_block_rows = 128;
            plan->threads = 256;
            plan->lds_bytes = d == 64 ? bf16_pair_d64_lds_bytes(causal, o_dtype) : bf16_pair_d128_lds_bytes(causal, o_dtype);
            // grid: 8 XCD groups x (the largest group's units, or -- more units than CUs in a group -- two workgroups per CU)
            const int jpx = device_cus() / 8;
            const int64_t per_group = (((int64_t)B * H + 7) / 8) * getNumCta(S, 128);
            plan->grid = 8 * (int)(per_group <= jpx ? per_group : 2 * jpx);   // (d = 128: per_group <= jpx always)
        }
    } else {
        plan->kernel_id = 0;
        plan->q_block_rows = GenericCfg::BQ;
        plan->kv_block_rows = GenericCfg::BK;
        plan->threads = GenericCfg::THREADS;
        plan->lds_bytes = generic_lds_bytes(d);
        const int nQ = getNumCta(S, plan->q_block_rows);
        const int64_t units = (int64_t)B * H * nQ;
        plan->grid = (int)(8 * ((units + 7) / 8));
    }
    return FA_OK;
}

template <typename InT, typename OutT>
static hipError_t launch_generic_io(const Params& p, const fa_launch_plan& plan, int d, bool causal, hipStream_t st) {
    dim3 grid(plan.grid), block(plan.threads);
    static std::atomic<bool> done_c[64], done_n[64];
    const int lds_max = generic_lds_bytes(256);   // 100 KiB at the largest supported head dimension
    if (causal) {
        const hipError_t attr = raise_lds_limit(fwd_generic_kernel<InT, OutT, true>, lds_max, done_c);
        if (attr != hipSuccess) return attr;
        hipLaunchKernelGGL((fwd_generic_kernel<InT, OutT, true>), grid, block, plan.lds_bytes, st, p, d);
    } else {
        const hipError_t attr = raise_lds_limit(fwd_generic_kernel<InT, OutT, false>, lds_max, done_n);
        if (attr != hipSuccess) return attr;
        hipLaunchKernelGGL((fwd_generic_kernel<InT, OutT, false>), grid, block, plan.lds_bytes, st, p, d);
    }
    return hipGetLastError();
}

template <typename InT>
static hipError_t launch_generic(const Params& p, const fa_launch_plan& plan, int d, bool causal, int o_dtype, hipStream_t st) {
    if (o_dtype == FA_DTYPE_F32) return launch_generic_io<InT, float>(p, plan, d, causal, st);
    if (o_dtype == FA_DTYPE_BF16) return launch_generic_io<InT, __bf16>(p, plan, d, causal, st);
    return launch_generic_io<InT, _Float16>(p, plan, d, causal, st);
}

static int run(const void* Q, const void* K, const void* V, void* O, float* lse, int B, int H, int S, int Sk, int d,
               float scale, bool causal, int dtype, int o_dtype, const fa_strides* sQ,
               const fa_strides* sK, const fa_strides* sV, const fa_strides* sO, void* stream, unsigned flags = 0) {
    int rc = validate(Q, K, V, O, B, H, S, d, scale, dtype, o_dtype);
    if (rc != FA_OK) return rc;
    if (flags & ~(unsigned)(FA_FLAG_F16_WEIGHTS | FA_FLAG_BF16_WEIGHTS)) return FA_ERR_BAD_FLAGS;
    if ((flags & FA_FLAG_F16_WEIGHTS) && (flags & FA_FLAG_BF16_WEIGHTS)) return FA_ERR_BAD_FLAGS;
    // the fp16-weights kernels exist for bf16 inputs at the natively instantiated head dimensions
    if ((flags & FA_FLAG_F16_WEIGHTS) && !(dtype == FA_DTYPE_BF16 && (d == 64 || d == 128) && scale > 0.f)) return FA_ERR_BAD_FLAGS;
    if ((flags & FA_FLAG_BF16_WEIGHTS) && dtype != FA_DTYPE_BF16) return FA_ERR_BAD_FLAGS;
    if (Sk <= 0 || Sk > (1 << 24)) return FA_ERR_BAD_SHAPE;
    if (lse && !aligned16(lse)) return FA_ERR_MISALIGNED;
    const int esz = elem_size(dtype), osz = elem_size(o_dtype);
    if (!strides_ok(sQ, esz, d) || !strides_ok(sK, esz, d) || !strides_ok(sV, esz, d) || !strides_ok(sO, osz, d))
        return FA_ERR_BAD_STRIDE;
    fa_launch_plan plan;
    make_plan(B, H, S, d, causal, dtype, o_dtype, scale, &plan);
    if (plan.kernel_id != 0) {
        // K/V are fetched through buffer descriptors with 32-bit byte offsets: one head's extent
        // (seqLen x row stride) must stay below 2^31 bytes (two prefetch tiles of slack included)
        const int64_t ks = sK ? sK->strideS : d, vs = sV ? sV->strideS : d;
        if (((int64_t)Sk + 192) * ks * esz >= (1ll << 31) || ((int64_t)Sk + 192) * vs * esz >= (1ll << 31)) return FA_ERR_BAD_SHAPE;
    }
    Params p;
    fill_params(p, Q, K, V, O, lse, B, H, S, Sk, d, scale, sQ, sK, sV, sO);
    const int nQ_total = getNumCta(S, plan.q_block_rows);
    if ((int64_t)B * H * nQ_total > INT32_MAX / 2) return FA_ERR_BAD_SHAPE;   // unit indices are 32-bit
    p.dbg = nullptr;
    p.hp = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // One launch covers the query blocks [qb0, qb0 + nq) of every head.
    auto set_range = [&](int qb0, int nq, bool persistent) {
        p.qb0 = qb0;
        p.nQ = nq;
        p.units = B * H * nq;
        p.cpx = (p.units + 7) / 8;
        if (persistent) plan.grid = 8 * std::min(p.cpx, device_cus() / 8);   // one workgroup per CU, or per unit when there are fewer
        else plan.grid = 8 * p.cpx;
        p.jpx = plan.grid / 8;
    };
    hipError_t e = hipSuccess;
    if (plan.kernel_id == 3) {          // fp32 inputs: exact-fp32 MFMA kernel at D = 128 / 64 (narrower rows zero-padded)
        set_range(0, nQ_total, false);
        e = d > 64 ? launch_f32_d128(p, plan, causal, d != 128, o_dtype, st) : launch_f32_d64(p, plan, causal, d != 64, o_dtype, st);
    } else if (plan.kernel_id == 2) {   // fp8 e4m3fn inputs
        set_range(0, nQ_total, true);
        e = launch_fp8_d128(p, plan, causal, d != 128, o_dtype, st);
    } else if (plan.kernel_id == 1) {   // bf16 inputs
        // Which query blocks take fp16 softmax weights (early_q_blocks): all with FA_FLAG_F16_WEIGHTS, none with
        // FA_FLAG_BF16_WEIGHTS, by default the rows that see few keys.  Both kinds present (a causal problem longer than
        // FA_EARLY_KEYS): ONE launch of ONE kernel over the list of all query blocks, in the single kernel's head-aligned order (all of a
        // head's blocks start in one round of one XCD group, so its K/V is streamed from that XCD's L2); every unit runs in the
        // precision of its block (kernel_bf16.hip.h: KernelCfg::MIX).
        const int hp = early_q_blocks(S, Sk, d, causal, flags, plan.q_block_rows);
        if (pair_kernel_applies(B, H, S, d, causal, dtype, scale)) {   // (make_plan chose 128-row blocks, 256 threads, its grid)
            p.qb0 = 0;
            p.nQ = nQ_total;
            p.units = B * H * nQ_total;
            p.cpx = (p.units + 7) / 8;
            p.jpx = device_cus() / 8;   // (workgroups of one dispatch round per XCD group: what the pairing counts in)
            p.hp = hp;                  // (d = 128: one mixed-precision configuration, the unit's block says which precision)
            e = d == 64 ? launch_bf16_pair_d64(p, hp, p.jpx, plan, causal, o_dtype, st) : launch_bf16_pair_d128(p, hp, p.jpx, plan, causal, o_dtype, st);
        } else if (hp > 0 && causal) {
            // the mixed-precision kernel; hp = nQ_total (FA_FLAG_F16_WEIGHTS, or every row sees fewer than FA_EARLY_KEYS keys) makes it the
            // fp16-weights kernel of the 32x32x16 engine, K by LDS-DMA: faster under the mask than the 16x16x32 one with both tiles through
            // registers (+2.4 % at S = 4096 d = 128, +8 % at S = 2048 d = 64: profiles/r04_tune_f_fp16_everywhere_*.log)
            set_range(0, nQ_total, true);
            p.hp = hp;
            e = launch_bf16_causal_mix(p, plan, d, o_dtype, st);
        } else if (hp > 0) {   // (without the mask early_q_blocks is all or nothing: hp = nQ_total, every unit runs with fp16 weights)
            set_range(0, hp, true);
            p.hp = hp;
            e = launch_bf16_p16(p, plan, causal, d, o_dtype, st);
        } else {
            set_range(0, nQ_total, true);
            e = d > 64 ? launch_bf16_d128(p, plan, causal, d != 128, o_dtype, st) : launch_bf16_d64(p, plan, causal, d != 64, o_dtype, st);
        }
    } else {
        set_range(0, nQ_total, false);
        e = dtype == FA_DTYPE_F32 ? launch_generic<float>(p, plan, d, causal, o_dtype, st) : launch_generic<__bf16>(p, plan, d, causal, o_dtype, st);
    }
    return (int)e;
}

}  // namespace fa

extern "C" {

int flash_attention(const void* Q, const void* K, const void* V, void* O, int batchSize, int numHeads,
                    int seqLen, int dHead, float scale, bool is_causal, int dtype, int o_dtype,
                    void* stream) {
    return fa::run(Q, K, V, O, nullptr, batchSize, numHeads, seqLen, seqLen, dHead, scale, is_causal, dtype, o_dtype,
                   nullptr, nullptr, nullptr, nullptr, stream);
}

int flash_attention_lse(const void* Q, const void* K, const void* V, void* O, float* LSE, int batchSize, int numHeads,
                        int seqLen, int dHead, float scale, bool is_causal, int dtype, int o_dtype, void* stream) {
    return fa::run(Q, K, V, O, LSE, batchSize, numHeads, seqLen, seqLen, dHead, scale, is_causal, dtype, o_dtype,
                   nullptr, nullptr, nullptr, nullptr, stream);
}

int flash_attention_strided(const void* Q, const void* K, const void* V, void* O, int batchSize,
                            int numHeads, int seqLen, int dHead, float scale, bool is_causal, int dtype,
                            int o_dtype, const fa_strides* sQ, const fa_strides* sK,
                            const fa_strides* sV, const fa_strides* sO, void* stream) {
    return fa::run(Q, K, V, O, nullptr, batchSize, numHeads, seqLen, seqLen, dHead, scale, is_causal, dtype, o_dtype, sQ,
                   sK, sV, sO, stream);
}

int flash_attention_cross(const void* Q, const void* K, const void* V, void* O, float* LSE, int batchSize, int numHeads,
                          int seqLenQ, int seqLenK, int dHead, float scale, bool is_causal, int dtype, int o_dtype,
                          const fa_strides* sQ, const fa_strides* sK, const fa_strides* sV, const fa_strides* sO,
                          void* stream) {
    return fa::run(Q, K, V, O, LSE, batchSize, numHeads, seqLenQ, seqLenK, dHead, scale, is_causal, dtype, o_dtype, sQ,
                   sK, sV, sO, stream);
}

int flash_attention_ex(const void* Q, const void* K, const void* V, void* O, float* LSE, int batchSize, int numHeads,
                       int seqLenQ, int seqLenK, int dHead, float scale, bool is_causal, int dtype, int o_dtype,
                       const fa_strides* sQ, const fa_strides* sK, const fa_strides* sV, const fa_strides* sO,
                       unsigned flags, void* stream) {
    return fa::run(Q, K, V, O, LSE, batchSize, numHeads, seqLenQ, seqLenK, dHead, scale, is_causal, dtype, o_dtype, sQ,
                   sK, sV, sO, stream, flags);
}

int flash_attention_weights(const void* Q, const void* K, const float* LSE, float* P, int batchSize, int numHeads,
                            int seqLenQ, int seqLenK, int dHead, float scale, bool is_causal, int dtype,
                            const fa_strides* sQ, const fa_strides* sK, void* stream) {
    using namespace fa;
    if (!Q || !K || !LSE || !P) return FA_ERR_NULL_POINTER;
    if (!aligned16(Q) || !aligned16(K) || !aligned16(LSE) || !aligned16(P)) return FA_ERR_MISALIGNED;
    if (batchSize <= 0 || numHeads <= 0 || seqLenQ <= 0 || seqLenK <= 0 || dHead <= 0) return FA_ERR_BAD_SHAPE;
    if (!std::isfinite(scale)) return FA_ERR_BAD_SCALE;
    if (dtype != FA_DTYPE_F32 && dtype != FA_DTYPE_BF16 && dtype != FA_DTYPE_FP8_E4M3) return FA_ERR_UNSUPPORTED_DTYPE;
    const int esz = elem_size(dtype);
    if (dHead > 256 || (dHead * esz) % 16 != 0) return FA_ERR_UNSUPPORTED_DHEAD;
    if (!strides_ok(sQ, esz, dHead) || !strides_ok(sK, esz, dHead)) return FA_ERR_BAD_STRIDE;
    WeightsParams p;
    p.Q = Q; p.K = K; p.lse = LSE; p.P = P;
    p.qB = sQ ? sQ->strideB : (int64_t)numHeads * seqLenQ * dHead; p.qH = sQ ? sQ->strideH : (int64_t)seqLenQ * dHead;
    p.qS = sQ ? sQ->strideS : dHead;
    p.kB = sK ? sK->strideB : (int64_t)numHeads * seqLenK * dHead; p.kH = sK ? sK->strideH : (int64_t)seqLenK * dHead;
    p.kS = sK ? sK->strideS : dHead;
    p.H = numHeads; p.Sq = seqLenQ; p.Sk = seqLenK; p.d = dHead;
    p.nQ = (seqLenQ + 15) / 16; p.nK = (seqLenK + 63) / 64;
    p.scale = scale; p.causal = is_causal;
    const int64_t blocks = (int64_t)batchSize * numHeads * p.nQ * p.nK;
    if (blocks > INT32_MAX) return FA_ERR_BAD_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int lds = weights_lds_bytes(dHead), lds_max = weights_lds_bytes(256);   // 80 KiB at dHead = 256
    static std::atomic<bool> done[3][64];
    hipError_t attr;
    if (dtype == FA_DTYPE_F32) {
        if ((attr = raise_lds_limit(attn_weights_kernel<float>, lds_max, done[0])) != hipSuccess) return (int)attr;
        hipLaunchKernelGGL((attn_weights_kernel<float>), dim3((unsigned)blocks), dim3(256), lds, st, p);
    } else if (dtype == FA_DTYPE_BF16) {
        if ((attr = raise_lds_limit(attn_weights_kernel<__bf16>, lds_max, done[1])) != hipSuccess) return (int)attr;
        hipLaunchKernelGGL((attn_weights_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), lds, st, p);
    } else {
        if ((attr = raise_lds_limit(attn_weights_kernel<fp8_t>, lds_max, done[2])) != hipSuccess) return (int)attr;
        hipLaunchKernelGGL((attn_weights_kernel<fp8_t>), dim3((unsigned)blocks), dim3(256), lds, st, p);
    }
    return (int)hipGetLastError();
}

int flash_attention_shard_range(int totalHeads, int rank, int nRanks, int* lo, int* hi) {
    if (!lo || !hi) return FA_ERR_NULL_POINTER;
    if (totalHeads < 0 || nRanks <= 0 || rank < 0 || rank >= nRanks) return FA_ERR_BAD_SHAPE;
    *lo = (int)(((int64_t)totalHeads * rank) / nRanks);
    *hi = (int)(((int64_t)totalHeads * (rank + 1)) / nRanks);
    return FA_OK;
}

int flash_attention_sharded(int nDevices, const int* deviceIds, const void* const* Q, const void* const* K,
                            const void* const* V, void* const* O, int batchSize, int numHeads, int seqLen, int dHead,
                            float scale, bool is_causal, int dtype, int o_dtype, void* const* streams) {
    if (!deviceIds || !Q || !K || !V || !O) return FA_ERR_NULL_POINTER;
    if (nDevices <= 0 || batchSize <= 0 || numHeads <= 0 || (int64_t)batchSize * numHeads > INT32_MAX / 2) return FA_ERR_BAD_SHAPE;
    int home = 0;
    hipError_t e = hipGetDevice(&home);
    if (e != hipSuccess) return (int)e;
    int rc = FA_OK;
    for (int r = 0; r < nDevices && rc == FA_OK; ++r) {
        int lo = 0, hi = 0;
        flash_attention_shard_range(batchSize * numHeads, r, nDevices, &lo, &hi);
        if (hi == lo) continue;                       // more devices than heads: nothing for this one
        if ((e = hipSetDevice(deviceIds[r])) != hipSuccess) { rc = (int)e; break; }
        // the slab is a [hi-lo, 1, seqLen, dHead] problem of its own
        rc = fa::run(Q[r], K[r], V[r], O[r], nullptr, hi - lo, 1, seqLen, seqLen, dHead, scale, is_causal, dtype, o_dtype,
                     nullptr, nullptr, nullptr, nullptr, streams ? streams[r] : nullptr);
    }
    (void)hipSetDevice(home);
    return rc;
}

int flash_attention_plan(int batchSize, int numHeads, int seqLen, int dHead, bool is_causal, int dtype,
                         int o_dtype, fa_launch_plan* plan) {
    if (!plan) return FA_ERR_NULL_POINTER;
    if (batchSize <= 0 || numHeads <= 0 || seqLen <= 0 || dHead <= 0) return FA_ERR_BAD_SHAPE;
    if (dtype != FA_DTYPE_F32 && dtype != FA_DTYPE_BF16 && dtype != FA_DTYPE_FP8_E4M3) return FA_ERR_UNSUPPORTED_DTYPE;
    if (dtype == FA_DTYPE_FP8_E4M3 && dHead > 128) return FA_ERR_UNSUPPORTED_DHEAD;
    if (dHead > 256 || (dHead * fa::elem_size(dtype)) % 16 != 0) return FA_ERR_UNSUPPORTED_DHEAD;
    return fa::make_plan(batchSize, numHeads, seqLen, dHead, is_causal, dtype, o_dtype, 1.0f, plan);
}

int flash_attention_plan_ex(int batchSize, int numHeads, int seqLenQ, int seqLenK, int dHead, bool is_causal, int dtype,
                            int o_dtype, unsigned flags, fa_launch_plan_ex* early, fa_launch_plan_ex* main_) {
    fa_launch_plan base;
    const int rc = flash_attention_plan(batchSize, numHeads, seqLenQ, dHead, is_causal, dtype, o_dtype, &base);
    if (rc != FA_OK) return rc;
    if (seqLenK <= 0) return FA_ERR_BAD_SHAPE;
    if (flags & ~(unsigned)(FA_FLAG_F16_WEIGHTS | FA_FLAG_BF16_WEIGHTS)) return FA_ERR_BAD_FLAGS;
    if ((flags & FA_FLAG_F16_WEIGHTS) && ((flags & FA_FLAG_BF16_WEIGHTS) || !(dtype == FA_DTYPE_BF16 && (dHead == 64 || dHead == 128)))) return FA_ERR_BAD_FLAGS;
    if ((flags & FA_FLAG_BF16_WEIGHTS) && dtype != FA_DTYPE_BF16) return FA_ERR_BAD_FLAGS;
    const int nQ = getNumCta(seqLenQ, base.q_block_rows);
    const int hp = base.kernel_id == 1 ? fa::early_q_blocks(seqLenQ, seqLenK, dHead, is_causal, flags, base.q_block_rows) : 0;
    auto fill = [&](fa_launch_plan_ex* out, int qb0, int nq, bool p16) {
        if (!out) return;
        out->launch = base;
        out->first_q_block = qb0;
        out->q_blocks = nq;
        out->unit_lists = 0;
        const int64_t units = (int64_t)batchSize * numHeads * nq;
        if (nq == 0) out->launch.grid = 0;
        else if (base.kernel_id == 1 || base.kernel_id == 2) out->launch.grid = (int)(8 * std::min<int64_t>((units + 7) / 8, fa::device_cus() / 8));
        else out->launch.grid = (int)(8 * ((units + 7) / 8));
        if (p16) out->launch.lds_bytes = is_causal ? fa::bf16_causal_mix_lds_bytes(dHead, o_dtype) : fa::bf16_p16_lds_bytes(false, dHead, o_dtype);
    };
    fill(early, 0, hp, true);
    fill(main_, hp, nQ - hp, false);
    if (base.kernel_id == 1 && base.threads == 256) {   // the pair kernel: one launch whatever the ranges, its own grid and LDS size
        if (early && hp > 0) early->launch = base;
        if (main_ && nQ - hp > 0) main_->launch = base;
    } else if (hp > 0 && hp < nQ) {   // both kinds: ONE launch of the mixed-precision kernel over the list of all query blocks
        const int64_t units = (int64_t)batchSize * numHeads * nQ;
        const int grid = (int)(8 * std::min<int64_t>((units + 7) / 8, fa::device_cus() / 8));
        const int lds = fa::bf16_causal_mix_lds_bytes(dHead, o_dtype);
        if (early) { early->launch.grid = grid; early->launch.lds_bytes = lds; early->unit_lists = 1; }
        if (main_) { main_->launch.grid = grid; main_->launch.lds_bytes = lds; main_->unit_lists = 1; }
    }
    return FA_OK;
}

const char* flash_attention_error_string(int code) {
    switch (code) {
        case FA_OK: return "success";
        case FA_ERR_NULL_POINTER: return "null pointer argument";
        case FA_ERR_MISALIGNED: return "base pointer not 16-byte aligned";
        case FA_ERR_BAD_SHAPE: return "batchSize/numHeads/seqLen/dHead out of range";
        case FA_ERR_UNSUPPORTED_DHEAD: return "unsupported dHead for this dtype";
        case FA_ERR_UNSUPPORTED_DTYPE: return "unsupported dtype / o_dtype";
        case FA_ERR_BAD_SCALE: return "scale is not finite (fp8 inputs: not positive)";
        case FA_ERR_BAD_FLAGS: return "unknown or contradictory flags, or a flag that does not apply to this dtype / dHead";
        case FA_ERR_BAD_STRIDE: return "bad or misaligned stride";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown flash_attention error";
    }
}

const char* flash_attention_version(void) { return "fa-mi355x 0.1 (gfx950)"; }

}  // extern "C"
