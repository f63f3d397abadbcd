// computers.hip.h -- per-wave compute of the fused forward pass: QK^T (MFMA) -> online softmax
// (in registers) -> PV (MFMA).
//
// Counterpart of the reference's kernels/computers.cuh:5-69 (twoLoaderMhaComputeWarp) and the
// device helpers it calls, kernels/utils.cuh:17-45 (computeTileScore), :58-81
// (updateSoftmaxState), :93-113 (multiplyVAccumulateO).  Same algorithm -- blocked scores, running
// (max, sum) per query row, rescaled accumulation of P.V -- re-designed for CDNA4:
//
//   * one WAVE owns 32 query rows (the reference: one warp per row) and a KV tile is 64 keys;
//   * scores come from v_mfma_f32_32x32x16_bf16 in the SWAPPED orientation S^T = K.Q^T, so the
//     32 scores of one query row and key half sit in ONE lane's registers: the row max / row sum
//     are 31 in-lane ops + one v_permlane32_swap (the reference: cg::reduce + Bc-1 shuffles,
//     utils.cuh:66-73);
//   * the recurrence is the FA-2 form: O and l stay un-normalised and are divided once at the
//     end (the reference renormalises every tile, utils.cuh:75-80 -- one divide per weight);
//     exp is exp2 with scale*log2(e) folded into one FMA;
//   * the O rescale is lazy: skipped while the row max grew by < RESCALE_THR (in log2 units)
//     for every row of the wave, so P stays <= 2^THR -- exact in f32 accumulation;
//   * P^T (accumulator layout: key in the register index, query on the lane) is already the B
//     operand of O^T += V^T.P^T; V^T fragments come from ds_read_b64_tr_b16.  O lives in 16*D/32
//     accumulator registers (the reference keeps O in shared memory, utils.cuh:107-111).
#pragma once

#include "loaders.hip.h"

namespace fa {

template <int D>
struct WaveCompute {
    static constexpr int KS = D / 16;   // k-steps of the QK^T contraction
    static constexpr int DB = D / 32;   // 32-wide d blocks of O^T
    static constexpr float RESCALE_THR = 8.0f;  // log2 units; 0 = rescale whenever the max grows

    bf16x8 qf[KS];    // B fragments of this wave's 32 query rows
    f32x16 o[DB];     // O^T accumulators: row = d, col = query
    float m;          // running max used for exponentiation (scaled, log2 domain)
    float l;          // partial row sum (this lane's key half)

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
        m = -INFINITY;
        l = 0.f;
    }

    // Q[q_row][16ks + 8h .. +7] for q_row = row0 + (lane&31), clamped to S-1.
    __device__ __forceinline__ void load_q(const char* Qh, int64_t qS_bytes, int row0, int S, int lane) {
        int r = row0 + (lane & 31);
        r = r < S ? r : S - 1;
        const char* src = Qh + r * qS_bytes + (lane >> 5) * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(src + ks * 32);
    }

    // S^T tile kt (keys 32kt..32kt+31 of the LDS K image) = K . Q^T
    __device__ __forceinline__ f32x16 qk_tile(lds_ptr kimg, int kbase, int kt) const {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            s = mfma_32x32x16(lds_read_b128(kimg, kbase + ks * 2048 + kt * 512), qf[ks], s);
        return s;
    }

    // Mask of the diagonal / tail tile: key index > query index, or key index >= S.
    // s[r] of tile kt holds key kv0 + 32kt + acc_row(r,h), query q_row0 + (lane&31).
    template <bool CAUSAL>
    __device__ __forceinline__ void mask_tile(f32x16& s, int kt, int kv0, int q_row0, int S, int lane) const {
        const int h = lane >> 5;
        const int qi = q_row0 + (lane & 31);
        const int lim = CAUSAL ? (qi < S - 1 ? qi : S - 1) : S - 1;  // last visible key
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kv0 + 32 * kt + acc_row(r, 0) + 4 * h;
            s[r] = key > lim ? -INFINITY : s[r];
        }
    }

    // Online-softmax update for one 64-key tile; returns P (un-normalised, exp2 domain) packed to
    // bf16 as the four B fragments (16 keys each) of the PV contraction.
    __device__ __forceinline__ void softmax_tile(const f32x16& s0, const f32x16& s1, float c, bf16x8 (&pf)[4]) {
        float mx = s0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[r]);
        mx = max_both_halves(mx) * c;
        // lazy rescale (wave-uniform branch)
        if (__any(mx > m + RESCALE_THR)) {
            const float mn = fmaxf(m, mx);
            const float alpha = fast_exp2(m - mn);
            m = mn;
            l *= alpha;
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        float p0[16], p1[16];
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p0[r] = fast_exp2(fmaf(s0[r], c, -m));
            p1[r] = fast_exp2(fmaf(s1[r], c, -m));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += p0[r] + p1[r];
        l += sum;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pf[0][j] = (__bf16)p0[j];
            pf[1][j] = (__bf16)p0[8 + j];
            pf[2][j] = (__bf16)p1[j];
            pf[3][j] = (__bf16)p1[8 + j];
        }
    }

    // O^T += V^T . P^T over the 64 keys of the tile.
    __device__ __forceinline__ void pv_tile(lds_ptr vimg, int vbase, const bf16x8 (&pf)[4]) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const s16x4 lo = lds_read_tr16_b64(vimg, vbase + (2 * s4) * (DB * 512) + db * 512);
                const s16x4 hi = lds_read_tr16_b64(vimg, vbase + (2 * s4 + 1) * (DB * 512) + db * 512);
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o[db] = mfma_32x32x16(__builtin_bit_cast(bf16x8, a), pf[s4], o[db]);
            }
        }
    }

    // Divide by the row sum and store O[q][d].  Lane holds d = 32db + 8g4 + 4h + (0..3) for g4 in
    // 0..3, i.e. 4 consecutive d per register quad.
    template <typename OutT>
    __device__ __forceinline__ void store_o(char* Oh, int64_t oS_bytes, int row0, int S, int lane) {
        const float inv = 1.0f / sum_both_halves(l);
        const int qi = row0 + (lane & 31);
        const int h = lane >> 5;
        if (qi >= S) return;
        char* dst = Oh + qi * oS_bytes;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = 32 * db + 8 * g4 + 4 * h;
                const float a = o[db][4 * g4 + 0] * inv, b = o[db][4 * g4 + 1] * inv;
                const float c2 = o[db][4 * g4 + 2] * inv, e = o[db][4 * g4 + 3] * inv;
                if constexpr (sizeof(OutT) == 4) {
                    f32x4 v = {a, b, c2, e};
                    *reinterpret_cast<f32x4*>(dst + d0 * 4) = v;
                } else if constexpr (__is_same(OutT, __bf16)) {
                    u32x2 v = {pack_bf16(a, b), pack_bf16(c2, e)};
                    *reinterpret_cast<u32x2*>(dst + d0 * 2) = v;
                } else {
                    u32x2 v = {pack_f16(a, b), pack_f16(c2, e)};
                    *reinterpret_cast<u32x2*>(dst + d0 * 2) = v;
                }
            }
    }
};

}  // namespace fa
