// kernel_bf16_v1.hip.h -- first (unpipelined, 2-slot) bf16 MFMA kernel, kept as the A/B baseline of
// tests/fa_tune.hip while the pipelined kernel (kernel_bf16.hip.h) is tuned.  Not launched by the library.
#pragma once

#include "computers.hip.h"

namespace fa {

// ------------------------------------------------------------------------------------------------
// bf16 MFMA forward kernel.  One workgroup = 8 waves = 256 query rows of one (batch, head);
// each wave owns 32 rows; KV tiles of 64 keys are double-buffered in LDS.
//
// Per tile:   issue global loads of tile t+1 (registers)      <- HBM/L2 latency hides below
//             S^T = K.Q^T      16 (D=128) MFMA 32x32x16, K fragments by ds_read_b128
//             online softmax   in registers
//             O^T += V^T.P^T   16 MFMA, V^T fragments by ds_read_b64_tr_b16
//             write tile t+1 into the other LDS buffer; one barrier
// ------------------------------------------------------------------------------------------------
template <int D, bool CAUSAL, typename OutT>
__global__ __launch_bounds__(512, 2) void fwd_bf16_kernel(const Params p) {
    using Stage = KVStage<D>;
    constexpr int KVBLK = 64, QBLK = 256;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_ptr smem = (lds_ptr)smem_raw;
    // [buf0: K image | V image][buf1: K image | V image]
    constexpr int BUF_BYTES = 2 * Stage::TILE_BYTES;

    int g, qb;
    if (!unit_of_block(p, CAUSAL, g, qb)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = g / p.H, h = g - b * p.H;
    const int S = p.S;

    const char* Qh = (const char*)p.Q + (b * p.qB + h * p.qH) * 2;
    const char* Kh = (const char*)p.K + (b * p.kB + h * p.kH) * 2;
    const char* Vh = (const char*)p.V + (b * p.vB + h * p.vH) * 2;
    char* Oh = (char*)p.O + (b * p.oB + h * p.oH) * (int64_t)sizeof(OutT);
    const int64_t qSb = p.qS * 2, kSb = p.kS * 2, vSb = p.vS * 2, oSb = p.oS * (int64_t)sizeof(OutT);

    const int q_row0 = qb * QBLK + wave * 32;       // first query row of this wave
    const int q_end = min(S, (qb + 1) * QBLK);      // one past the last query row of the block
    const int n_tiles = CAUSAL ? (q_end + KVBLK - 1) / KVBLK : (S + KVBLK - 1) / KVBLK;
    // tiles this wave computes: all (non-causal) or up to its own diagonal (causal)
    const bool wave_live = q_row0 < S;
    const int my_tiles = !wave_live ? 0 : (CAUSAL ? min(n_tiles, (q_row0 + 31) / KVBLK + 1) : n_tiles);

    WaveCompute<D> wc;
    wc.init();
    wc.load_q(Qh, qSb, q_row0, S, lane);

    Stage st;
    st.load(Kh, Vh, kSb, vSb, 0, S, wave, lane);
    st.write(smem, smem + Stage::TILE_BYTES, wave, lane);
    __syncthreads();

    const int kbase = k_read_base(lane);
    const int vbase = v_read_base(lane);
    const float c = p.scale_log2;

    for (int t = 0; t < n_tiles; ++t) {
        lds_ptr kimg = smem + (t & 1) * BUF_BYTES;
        lds_ptr vimg = kimg + Stage::TILE_BYTES;
        const bool more = t + 1 < n_tiles;
        if (more) st.load(Kh, Vh, kSb, vSb, (t + 1) * KVBLK, S, wave, lane);

        if (t < my_tiles) {
            const int kv0 = t * KVBLK;
            f32x16 s0 = wc.qk_tile(kimg, kbase, 0);
            f32x16 s1 = wc.qk_tile(kimg, kbase, 1);
            const bool need_mask = (CAUSAL && kv0 + KVBLK - 1 > q_row0) || (kv0 + KVBLK > S);
            if (need_mask) {
                wc.template mask_tile<CAUSAL>(s0, 0, kv0, q_row0, S, lane);
                wc.template mask_tile<CAUSAL>(s1, 1, kv0, q_row0, S, lane);
            }
            bf16x8 pf[4];
            wc.softmax_tile(s0, s1, c, pf);
            wc.pv_tile(vimg, vbase, pf);
        }

        if (more) {
            lds_ptr knext = smem + ((t + 1) & 1) * BUF_BYTES;
            st.write(knext, knext + Stage::TILE_BYTES, wave, lane);
        }
        __syncthreads();
    }

    if (wave_live) wc.template store_o<OutT>(Oh, oSb, q_row0, S, lane);
}

}  // namespace fa
