// generic.hip.h -- exact-fp32 forward kernel for ANY head dimension <= 256 and ANY sequence length.
//
// This is the literal drop-in for the reference's fp32 signature (const float* Q,K,V; float* O,
// kernels/FlashAttention.cuh:59-63) and the path its own known-answer cases run on
// (tests/main.cu:107: S=16, d=16).  Same blocked algorithm as the reference (score tile ->
// running max/sum -> rescaled P.V, kernels/utils.cuh:58-113) on the f32 VALU, with the
// reference's defects fixed: scores use K (D1), each (b,h) is independent (D2), masked tiles
// cannot NaN (D3), the grid covers all query blocks (D5).
//
//   workgroup = 256 threads = 32 query rows;  8 lanes per row;  KV tile = 32 keys.
//   lane (row r, sub j): scores of keys j, j+8, j+16, j+24; O columns j, j+8, ...
// Inputs may be fp32 or bf16 (converted to fp32 in LDS); all arithmetic is fp32 with expf.
#pragma once

#include "loaders.hip.h"

namespace fa {

struct GenericCfg {
    static constexpr int BQ = 32, BK = 32, THREADS = 256;
};

inline int generic_lds_bytes(int d) {
    return (GenericCfg::BQ * (d + 1) + GenericCfg::BK * (d + 1) + GenericCfg::BK * d +
            GenericCfg::BQ * (GenericCfg::BK + 1)) * 4;
}

template <typename InT, typename OutT, bool CAUSAL>
__global__ __launch_bounds__(256) void fwd_generic_kernel(const Params p, const int d) {
    constexpr int BQ = GenericCfg::BQ, BK = GenericCfg::BK, NT = GenericCfg::THREADS;
    constexpr int NC = 32;  // column chunks of 8 -> d <= 256
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Qs = reinterpret_cast<float*>(smem_raw);   // [BQ][d+1]
    float* Ks = Qs + BQ * (d + 1);                    // [BK][d+1]
    float* Vs = Ks + BK * (d + 1);                    // [BK][d]
    float* Ps = Vs + BK * d;                          // [BQ][BK+1]

    int g, qb;
    if (!unit_of_block(p, CAUSAL, g, qb)) return;
    const int b = g / p.H, h = g - b * p.H;
    const int S = p.S, Sk = p.Sk;
    const InT* Qh = (const InT*)p.Q + b * p.qB + h * p.qH;
    const InT* Kh = (const InT*)p.K + b * p.kB + h * p.kH;
    const InT* Vh = (const InT*)p.V + b * p.vB + h * p.vH;
    OutT* Oh = (OutT*)p.O + b * p.oB + h * p.oH;

    const int tid = threadIdx.x;
    const int row = tid >> 3, sub = tid & 7;
    const int q0 = qb * BQ;
    const int qi = q0 + row;

    for (int idx = tid; idx < BQ * d; idx += NT) {
        const int r = idx / d, c = idx - r * d;
        const int src = min(q0 + r, S - 1);
        Qs[r * (d + 1) + c] = elem_traits<InT>::load(Qh + src * p.qS + c);
    }

    float o[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] = 0.f;
    float m = -INFINITY, l = 0.f;

    const int q_last = min(S, q0 + BQ) - 1;
    const int kv_end = CAUSAL ? min(q_last + 1, Sk) : Sk;       // keys [0, kv_end) are needed by this block
    for (int kv0 = 0; kv0 < kv_end; kv0 += BK) {
        __syncthreads();                              // previous tile fully consumed (and Q visible)
        for (int idx = tid; idx < BK * d; idx += NT) {
            const int r = idx / d, c = idx - r * d;
            const int src = min(kv0 + r, Sk - 1);
            Ks[r * (d + 1) + c] = elem_traits<InT>::load(Kh + src * p.kS + c);
            Vs[r * d + c] = elem_traits<InT>::load(Vh + src * p.vS + c);
        }
        __syncthreads();

        float s[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = 0.f;
        const float* qrow = Qs + row * (d + 1);
        for (int j = 0; j < d; ++j) {
            const float qv = qrow[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = fmaf(qv, Ks[(sub + 8 * i) * (d + 1) + j], s[i]);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = kv0 + sub + 8 * i;
            const bool masked = key >= Sk || (CAUSAL && key > qi);   // utils.cuh:43: k > q
            s[i] = masked ? -INFINITY : s[i] * p.scale;
            mx = fmaxf(mx, s[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        mx = fmaxf(mx, __shfl_xor(mx, 4));
        const float mn = fmaxf(m, mx);
        // rows past S (clamped copies) or a fully masked tile keep mn = -inf only if nothing was
        // ever visible; guard the exp so they stay finite
        const float alpha = (mn == -INFINITY) ? 1.f : expf(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float pv = (mn == -INFINITY) ? 0.f : expf(s[i] - mn);
            Ps[row * (BK + 1) + sub + 8 * i] = pv;
            ps += pv;
        }
        ps += __shfl_xor(ps, 1);
        ps += __shfl_xor(ps, 2);
        ps += __shfl_xor(ps, 4);
        l = l * alpha + ps;
        m = mn;
        __syncthreads();                              // P tile visible to the row's 8 lanes

        const float* prow = Ps + row * (BK + 1);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c * 8 < d) {                          // workgroup-uniform
                const int col = min(c * 8 + sub, d - 1);
                float acc = 0.f;
#pragma unroll 8
                for (int k = 0; k < BK; ++k) acc = fmaf(prow[k], Vs[k * d + col], acc);
                o[c] = fmaf(o[c], alpha, acc);
            }
        }
    }

    if (qi < S) {
        if (p.lse && sub == 0) p.lse[(int64_t)g * S + qi] = m + logf(l);   // natural-log domain: m = max(scale*s)
        const float inv = 1.0f / l;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 8 + sub;
            if (col < d) elem_traits<OutT>::store(Oh + qi * p.oS + col, o[c] * inv);
        }
    }
}

}  // namespace fa
