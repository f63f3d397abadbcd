// kernel_bf16.hip.h -- the bf16 MFMA forward kernel (software-pipelined, 3-slot LDS ring).
//
// Counterpart of the reference's kernel entry kernels/FlashAttention.cuh:59-84 and of the loop
// nests of kernels/computers.cuh:33-67 / kernels/loaders.cuh:132-156,177-201.  The reference
// commits a tile and immediately waits for it on every pipeline (no load/compute overlap, SURVEY.md
// section 3.1); here three things overlap inside every wave:
//
//   iteration t of a wave (tile = 64 keys, the wave owns 32 query rows):
//     top      issue the global loads of tile t+2 into registers          (HBM/L2 latency)
//     phase A  S(t+1) = K(t+1).Q^T    MFMA   ||  P(t) = exp2(c*S(t) - m), row sum, ->bf16   VALU
//     phase B  O^T   += V(t)^T.P(t)^T MFMA   ||  row max of S(t+1), tail of P(t)            VALU
//     end      lazy rescale decision for tile t+1; write tile t+2 into ring slot (t+2)%3; barrier
//
//   K(t+1) and V(t) are both live in LDS, hence a 3-slot ring (96 KiB at d = 128): slot (t+2)%3 =
//   slot (t-1)%3 was last read in iteration t-1, which every wave left at the previous barrier.
//   One barrier per tile.  The loop is unrolled x2 with ping-pong score registers so S(t+1) never
//   has to be copied into S(t).
#pragma once

#include "computers.hip.h"

namespace fa {

template <int D_, bool CAUSAL_, typename OutT_, int THR_ = 8, int SPLIT_B_ = 8, int NPRE_ = 4, bool SCHED_ = true,
          int VALU_A_ = 5, int VALU_B_ = 4>
struct KernelCfg {
    static constexpr bool SCHED = SCHED_;     // pin the MFMA / LDS-read / VALU interleave with sched_group_barrier
    static constexpr int VALU_A = VALU_A_, VALU_B = VALU_B_;   // VALU instructions per MFMA gap in phase A / B
    static constexpr int NPRE = NPRE_;        // K fragments read ahead of their MFMA
    static constexpr int D = D_;
    static constexpr bool CAUSAL = CAUSAL_;
    using OutT = OutT_;
    static constexpr int THR = THR_;          // lazy-rescale threshold, log2 units
    static constexpr int SPLIT_B = SPLIT_B_;  // how many of the 32 exponentials run in phase B
};

template <class C>
struct PipelinedWave {
    static constexpr int D = C::D;
    static constexpr int KS = D / 16;
    static constexpr int DB = D / 32;
    using Stage = KVStage<D>;

    bf16x8 qf[KS];
    f32x16 o[DB];
    float m, l;

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
        m = -INFINITY;
        l = 0.f;
    }

    __device__ __forceinline__ void load_q(const char* Qh, int64_t qS_bytes, int row0, int S, int lane) {
        int r = row0 + (lane & 31);
        r = r < S ? r : S - 1;
        const char* src = Qh + r * qS_bytes + (lane >> 5) * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(src + ks * 32);
    }
    // Make the Q fragments look "consumed" so hipcc waits for their loads HERE and not with a
    // pessimistic vmcnt inside the main loop (where it would also drain the tile prefetch).
    __device__ __forceinline__ void pin_q() {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 t = __builtin_bit_cast(u32x4, qf[ks]);
            asm volatile("" : "+v"(t));
            qf[ks] = __builtin_bit_cast(bf16x8, t);
        }
    }

    // S^T(both 32-key halves) = K.Q^T from the K image at `kimg`.  K fragments are read through a
    // rolling NPRE-deep register window (reading all 2*KS ahead costs 64 VGPRs at d = 128 and spills).
    __device__ __forceinline__ void qk(lds_ptr kimg, int kbase, f32x16& s0, f32x16& s1) const {
        constexpr int N = 2 * KS, NPRE = C::NPRE < N ? C::NPRE : N;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
        bf16x8 f[NPRE];
        // fragment i: key half kt = i / KS, k-step ks = i % KS
#pragma unroll
        for (int i = 0; i < NPRE; ++i) f[i] = lds_read_b128(kimg, kbase + (i % KS) * 2048 + (i / KS) * 512);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i < KS) s0 = mfma_32x32x16(f[i % NPRE], qf[i % KS], s0);
            else        s1 = mfma_32x32x16(f[i % NPRE], qf[i % KS], s1);
            if (i + NPRE < N)
                f[i % NPRE] = lds_read_b128(kimg, kbase + ((i + NPRE) % KS) * 2048 + ((i + NPRE) / KS) * 512);
        }
    }

    __device__ __forceinline__ void mask(f32x16& s0, f32x16& s1, int kv0, int q_row0, int S, int lane) const {
        const int qi = q_row0 + (lane & 31);
        const int lim = C::CAUSAL ? (qi < S - 1 ? qi : S - 1) : S - 1;
        const int k0 = kv0 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = (k0 + acc_row(r, 0)) > lim ? -INFINITY : s0[r];
            s1[r] = (k0 + 32 + acc_row(r, 0)) > lim ? -INFINITY : s1[r];
        }
    }

    __device__ __forceinline__ float row_max(const f32x16& s0, const f32x16& s1) const {
        float a = fmaxf(s0[0], s0[1]), b = fmaxf(s1[0], s1[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) {
            a = fmaxf(a, fmaxf(s0[r], s0[r + 1]));
            b = fmaxf(b, fmaxf(s1[r], s1[r + 1]));
        }
        return fmaxf(a, b);
    }

    // Decide (wave-uniform) whether the running max has to move for a tile whose raw row max is mx.
    __device__ __forceinline__ void update_max(float mx_raw, float c) {
        const float mx = max_both_halves(mx_raw) * c;
        if (__any(mx > m + (float)C::THR)) {
            const float mn = fmaxf(m, mx);
            const float alpha = fast_exp2(m - mn);
            m = mn;
            l *= alpha;
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
    }

    // p = exp2(c*s - m) for elements [e0, e1) of the 32 scores (0..15 -> s0, 16..31 -> s1); adds the
    // partial row sums and packs pairs into the PV B-fragments pf[e/8].
    template <int E0, int E1>
    __device__ __forceinline__ void exp_range(const f32x16& s0, const f32x16& s1, float c, bf16x8 (&pf)[4],
                                              float& sum_a, float& sum_b) const {
#pragma unroll
        for (int e = E0; e < E1; e += 2) {
            const float x0 = e < 16 ? s0[e & 15] : s1[e & 15];
            const float x1 = e < 16 ? s0[(e + 1) & 15] : s1[(e + 1) & 15];
            const float p0 = fast_exp2(fmaf(x0, c, -m));
            const float p1 = fast_exp2(fmaf(x1, c, -m));
            sum_a += p0;
            sum_b += p1;
            pf[e >> 3][e & 7] = (__bf16)p0;
            pf[e >> 3][(e & 7) + 1] = (__bf16)p1;
        }
    }

    __device__ __forceinline__ bf16x8 v_frag(lds_ptr vimg, int vbase, int s4, int db) const {
        const s16x4 lo = lds_read_tr16_b64(vimg, vbase + (2 * s4) * (DB * 512) + db * 512);
        const s16x4 hi = lds_read_tr16_b64(vimg, vbase + (2 * s4 + 1) * (DB * 512) + db * 512);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }

    template <int S4_0, int S4_1>
    __device__ __forceinline__ void pv_range(lds_ptr vimg, int vbase, const bf16x8 (&pf)[4]) {
#pragma unroll
        for (int s4 = S4_0; s4 < S4_1; ++s4)
#pragma unroll
            for (int db = 0; db < DB; ++db) o[db] = mfma_32x32x16(v_frag(vimg, vbase, s4, db), pf[s4], o[db]);
    }

    // Full pipelined iteration: cur = S(t) (consumed), nxt = S(t+1) (produced).
    // On the wave's last tile (has_next == false) the QK^T of the non-existent next tile is still
    // issued -- its result is never looked at -- so that there is ONE hot code path (a separate tail
    // body doubled the code and pushed lane-constant registers into scratch).
    __device__ __forceinline__ void full_step(lds_ptr k_next, lds_ptr v_cur, int kbase, int vbase, float c,
                                              const f32x16& cur0, const f32x16& cur1, f32x16& nxt0, f32x16& nxt1,
                                              bool has_next, bool mask_next, int kv0_next, int q_row0, int S,
                                              int lane) {
        constexpr int EA = 32 - C::SPLIT_B;
        bf16x8 pf[4];
        float sa = 0.f, sb = 0.f;
        // phase A
        qk(k_next, kbase, nxt0, nxt1);
        exp_range<0, EA>(cur0, cur1, c, pf, sa, sb);
        // phase B
        pv_range<0, EA / 8>(v_cur, vbase, pf);
        exp_range<EA, 32>(cur0, cur1, c, pf, sa, sb);
        float mx = row_max(nxt0, nxt1);
        pv_range<EA / 8, 4>(v_cur, vbase, pf);
        l += sa + sb;
        if constexpr (C::SCHED) {
            // Pin the interleave of this basic block (LLVM SchedGroupMask: VALU 0x2, MFMA 0x8,
            // DS_READ 0x100): K fragments NPRE ahead of their MFMA, ~5 VALU per MFMA gap in phase A
            // (exp work), V^T fragments two ahead in phase B with the row max of S(t+1).
            constexpr int NA = 2 * KS, NB = 4 * DB;
            __builtin_amdgcn_sched_group_barrier(0x100, C::NPRE, 0);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, C::VALU_A, 0);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, C::VALU_B, 0);
            }
        }
        if (has_next) {
            if (mask_next) {
                mask(nxt0, nxt1, kv0_next, q_row0, S, lane);
                mx = row_max(nxt0, nxt1);
            }
            update_max(mx, c);
        }
    }

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

template <class C>
__global__ __launch_bounds__(512, 2) void fwd_bf16_pipelined_kernel(const Params p) {
    constexpr int D = C::D;
    constexpr bool CAUSAL = C::CAUSAL;
    using OutT = typename C::OutT;
    using Stage = KVStage<D>;
    constexpr int KVBLK = 64, QBLK = 256;
    constexpr int TILE = Stage::TILE_BYTES, SLOT = 2 * TILE;   // slot = [K image | V image]
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_ptr smem = (lds_ptr)smem_raw;

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

    const int q_row0 = qb * QBLK + wave * 32;
    const int q_end = min(S, (qb + 1) * QBLK);
    const int n_tiles = CAUSAL ? (q_end + KVBLK - 1) / KVBLK : (S + KVBLK - 1) / KVBLK;
    const bool wave_live = q_row0 < S;
    const int my_tiles = !wave_live ? 0 : (CAUSAL ? min(n_tiles, (q_row0 + 31) / KVBLK + 1) : n_tiles);

    PipelinedWave<C> w;
    w.init();
    w.load_q(Qh, qSb, q_row0, S, lane);

    Stage st;
    st.load(Kh, Vh, kSb, vSb, 0, S, wave, lane);
    st.write(smem, smem + TILE, wave, lane);
    if (n_tiles > 1) {
        st.load(Kh, Vh, kSb, vSb, KVBLK, S, wave, lane);
        st.write(smem + SLOT, smem + SLOT + TILE, wave, lane);
    }
    w.pin_q();
    __syncthreads();

    const int kbase = k_read_base(lane);
    const int vbase = v_read_base(lane);
    const float c = p.scale_log2;

    auto needs_mask = [&](int t) { return (CAUSAL && t * KVBLK + KVBLK - 1 > q_row0) || (t * KVBLK + KVBLK > S); };

    f32x16 sA0, sA1, sB0, sB1;
    if (my_tiles > 0) {
        w.qk(smem, kbase, sA0, sA1);
        if (needs_mask(0)) w.mask(sA0, sA1, 0, q_row0, S, lane);
        w.update_max(w.row_max(sA0, sA1), c);
    }

    // ring slot byte offsets of tiles t, t+1, t+2
    int so_cur = 0, so_nxt = SLOT, so_wr = 2 * SLOT;

    auto step = [&](int t, f32x16& cur0, f32x16& cur1, f32x16& nxt0, f32x16& nxt1) {
        const bool more2 = t + 2 < n_tiles;
        if (more2) st.load(Kh, Vh, kSb, vSb, (t + 2) * KVBLK, S, wave, lane);
        if (t < my_tiles) {
            const bool has_next = t + 1 < my_tiles;
            w.full_step(smem + so_nxt, smem + so_cur + TILE, kbase, vbase, c, cur0, cur1, nxt0, nxt1, has_next,
                        has_next && needs_mask(t + 1), (t + 1) * KVBLK, q_row0, S, lane);
        }
        if (more2) st.write(smem + so_wr, smem + so_wr + TILE, wave, lane);
        __syncthreads();
        const int tmp = so_cur;
        so_cur = so_nxt;
        so_nxt = so_wr;
        so_wr = tmp;
    };

    for (int t = 0; t < n_tiles; t += 2) {
        step(t, sA0, sA1, sB0, sB1);
        if (t + 1 < n_tiles) step(t + 1, sB0, sB1, sA0, sA1);
    }

    if (wave_live) w.template store_o<OutT>(Oh, oSb, q_row0, S, lane);
}

}  // namespace fa
