// weights.hip.h -- the attention matrix itself, for callers that want to look at it.
//
// The reference's oracle returns `attn` next to `output` (check.py:20,25) and its demo prints it
// (check.py:42).  The fused forward kernel never materialises that [B,H,Sq,Sk] matrix; this small
// kernel rebuilds it from Q, K and the log-sum-exp the forward pass already produced:
//     attn[b,h,q,k] = exp(scale * <Q[q], K[k]> - LSE[q])        (0 where the causal mask hides k > q)
// so attn rows sum to 1 and attn @ V reproduces O.  It is a debug / inspection path for small S
// (SURVEY.md section 8f row 4): HBM-bound on the Sq*Sk fp32 output, fp32 FMAs on the VALU.
//
//   workgroup = 256 threads = 16 query rows x 64 keys: thread (r = tid/16, j = tid%16) owns keys
//   j, j+16, j+32, j+48 of row r, so a 16-thread group writes 64 contiguous bytes per store.
#pragma once

#include "loaders.hip.h"

namespace fa {

// one OCP e4m3fn byte (gfx950's v_cvt_f32_fp8 decodes the OCP format)
struct fp8_t { uint8_t v; };
template <> struct elem_traits<fp8_t> {
    __device__ static float load(const fp8_t* p) { return __builtin_amdgcn_cvt_f32_fp8((int)p->v, 0); }
};

struct WeightsParams {
    const void* Q;
    const void* K;
    const float* lse;   // [B,H,Sq] natural log
    float* P;           // [B,H,Sq,Sk] dense
    int64_t qB, qH, qS, kB, kH, kS;
    int H, Sq, Sk, d;
    int nQ, nK;         // tiles per head along q (16) and k (64)
    float scale;
    bool causal;
};

template <typename InT>
__global__ __launch_bounds__(256) void attn_weights_kernel(const WeightsParams p) {
    constexpr int BQ = 16, BK = 64;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int d = p.d;
    float* Qs = reinterpret_cast<float*>(smem_raw);   // [BQ][d+1]
    float* Ks = Qs + BQ * (d + 1);                    // [BK][d+1]
    const int kb = blockIdx.x % p.nK;
    const int qb = (blockIdx.x / p.nK) % p.nQ;
    const int g = blockIdx.x / (p.nK * p.nQ);
    const int b = g / p.H, h = g - b * p.H;
    const InT* Qh = (const InT*)p.Q + b * p.qB + h * p.qH;
    const InT* Kh = (const InT*)p.K + b * p.kB + h * p.kH;
    const int tid = threadIdx.x, q0 = qb * BQ, k0 = kb * BK;
    for (int idx = tid; idx < BQ * d; idx += 256) {
        const int r = idx / d, c = idx - r * d;
        Qs[r * (d + 1) + c] = elem_traits<InT>::load(Qh + (int64_t)min(q0 + r, p.Sq - 1) * p.qS + c);
    }
    for (int idx = tid; idx < BK * d; idx += 256) {
        const int r = idx / d, c = idx - r * d;
        Ks[r * (d + 1) + c] = elem_traits<InT>::load(Kh + (int64_t)min(k0 + r, p.Sk - 1) * p.kS + c);
    }
    __syncthreads();
    const int r = tid >> 4, j = tid & 15;
    const int qi = q0 + r;
    if (qi >= p.Sq) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const float* qrow = Qs + r * (d + 1);
    for (int c = 0; c < d; ++c) {
        const float qv = qrow[c];
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = fmaf(qv, Ks[(j + 16 * i) * (d + 1) + c], s[i]);
    }
    const float lse = p.lse[(int64_t)g * p.Sq + qi];
    float* Prow = p.P + ((int64_t)g * p.Sq + qi) * p.Sk;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = k0 + j + 16 * i;
        if (key < p.Sk) Prow[key] = (p.causal && key > qi) ? 0.f : expf(s[i] * p.scale - lse);   // utils.cuh:43
    }
}

inline int weights_lds_bytes(int d) { return (16 + 64) * (d + 1) * 4; }

}  // namespace fa
