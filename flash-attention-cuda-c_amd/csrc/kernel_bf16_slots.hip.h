// kernel_bf16_slots.hip.h -- bf16 MFMA forward kernel with a HAND-PLACED per-tile schedule.
//
// Same algorithm, LDS images, 3-slot ring and barrier protocol as kernel_bf16.hip.h (read its
// header first).  The difference is who decides the instruction order inside a tile.  hipcc, left
// alone (or nudged with sched_group_barrier), hoists the whole softmax in front of the MFMAs and
// makes every QK^T MFMA wait on an LDS read issued just before it.  Here a tile is cut into
// NA + NB "slots", one MFMA each, fenced by __builtin_amdgcn_sched_barrier(0):
//
//   phase A, slot i  (NA = 2*d/16):  MFMA  S(t+1) += Kfrag_i . Qfrag            (QK^T of the NEXT tile)
//                                    read  K fragment i+NPRE  (later: the first V^T fragments)
//                                    VALU  exp2 / row-sum / bf16-pack of this slot's share of P(t)
//   phase B, slot j  (NB = 4*d/32):  MFMA  O^T += V^Tfrag_j . P(t)frag
//                                    read  V^T fragment j+VPRE (ds_read_b64_tr_b16 x2)
//                                    VALU  rest of P(t); running row max of S(t+1)
//
// Element e of the 32 scores a lane holds goes to overall slot e*(NA + 3*NB/4)/32, which meets the
// deadlines "P group g (elements 8g..8g+7) complete before PV slot g*NB/4" with an even VALU load
// of <= ~1.25 elements (about 24 issue cycles) per 32-cycle MFMA gap at d = 128.
#pragma once

#include "kernel_bf16.hip.h"

namespace fa {

template <int D_, bool CAUSAL_, typename OutT_, int THR_ = 8, int NPRE_ = 4, int VPRE_ = 2, bool STAMP_ = false>
struct SlotCfg {
    static constexpr bool STAMP = STAMP_;   // diagnostic build: s_memtime stamps around the tile segments
    static constexpr int D = D_;
    static constexpr bool CAUSAL = CAUSAL_;
    using OutT = OutT_;
    static constexpr int THR = THR_;
    static constexpr int NPRE = NPRE_;   // K fragments in flight ahead of their MFMA
    static constexpr int VPRE = VPRE_;   // V^T fragments in flight ahead of their MFMA
    // PipelinedWave<> compatibility (update_max, mask, store_o reuse)
    static constexpr int SPLIT_B = 8;
    static constexpr bool SCHED = false;
    static constexpr int VALU_A = 0, VALU_B = 0;
};

template <class C>
struct SlotWave : PipelinedWave<C> {
    using Base = PipelinedWave<C>;
    static constexpr int D = C::D, KS = D / 16, DB = D / 32;
    static constexpr int NA = 2 * KS, NB = 4 * DB;
    static constexpr int NPRE = C::NPRE < NA ? C::NPRE : NA;
    static constexpr int VPRE = C::VPRE;
    static constexpr int SPAN = NA + (3 * NB) / 4;   // overall slots the exponentials are spread over

    __host__ __device__ static constexpr int elem_slot(int e) { return e * SPAN / 32; }

    // per-tile scratch state (registers)
    bf16x8 kf[NPRE];          // K fragment window
    bf16x8 vf[VPRE + 1];      // V^T fragment window
    uint32_t pw[16];          // P(t) packed bf16 pairs: word 4*g + w = elements 8g+2w, 8g+2w+1
    float sum_a, sum_b, mx_a, mx_b, p_even;
    unsigned long long t_mid = 0, t_end = 0;   // STAMP builds only

    // one asm statement: s_memtime + its own wait, fenced (guide: In-kernel stamps)
    __device__ __forceinline__ static unsigned long long stamp() {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    }

    __device__ __forceinline__ bf16x8 k_read(lds_ptr kimg, int kbase, int i) const {
        return lds_read_b128(kimg, kbase + (i % KS) * 2048 + (i / KS) * 512);
    }

    template <int E>
    __device__ __forceinline__ void exp_elem(const f32x16& c0, const f32x16& c1, float c) {
        const float x = E < 16 ? c0[E & 15] : c1[E & 15];
        const float p = fast_exp2(fmaf(x, c, -this->m));
        if constexpr (E & 1) {
            sum_b += p;
            pw[E >> 1] = pack_bf16(p_even, p);
            asm volatile("" : "+v"(sum_a), "+v"(sum_b));   // keep the adds in this slot (hipcc sinks them)
        } else {
            sum_a += p;
            p_even = p;
        }
    }
    template <int SLOT, int E = 0>
    __device__ __forceinline__ void exp_slot(const f32x16& c0, const f32x16& c1, float c) {
        if constexpr (E < 32) {
            if constexpr (elem_slot(E) == SLOT) exp_elem<E>(c0, c1, c);
            exp_slot<SLOT, E + 1>(c0, c1, c);
        }
    }

    __device__ __forceinline__ bf16x8 p_frag(int g) const {
        u32x4 v = {pw[4 * g], pw[4 * g + 1], pw[4 * g + 2], pw[4 * g + 3]};
        return __builtin_bit_cast(bf16x8, v);
    }

    // running max of S(t+1): PER values per phase-B slot.  The empty asm pins the partial maxima in
    // this slot (otherwise hipcc sinks all 32 max ops behind the MFMAs, into the has_next branch).
    template <int J>
    __device__ __forceinline__ void max_slot(const f32x16& n0, const f32x16& n1) {
        constexpr int PER = 32 / NB;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = J * PER + k;
            const float x = e < 16 ? n0[e & 15] : n1[e & 15];
            if (k & 1) mx_b = fmaxf(mx_b, x);
            else mx_a = fmaxf(mx_a, x);
        }
        asm volatile("" : "+v"(mx_a), "+v"(mx_b));
    }

    template <int I>
    __device__ __forceinline__ void slots_a(lds_ptr k_next, lds_ptr v_cur, int kbase, int vbase, float c,
                                            const f32x16& c0, const f32x16& c1, f32x16& n0, f32x16& n1) {
        if constexpr (I < NA) {
            if constexpr (I < KS) n0 = mfma_32x32x16(kf[I % NPRE], this->qf[I % KS], n0);
            else                  n1 = mfma_32x32x16(kf[I % NPRE], this->qf[I % KS], n1);
            if constexpr (I + NPRE < NA) kf[I % NPRE] = k_read(k_next, kbase, I + NPRE);
            // the last VPRE phase-A slots start the V^T fragment window of phase B
            if constexpr (I >= NA - VPRE) {
                constexpr int J = I - (NA - VPRE);
                vf[J % (VPRE + 1)] = this->v_frag(v_cur, vbase, J / DB, J % DB);
            }
            exp_slot<I>(c0, c1, c);
            __builtin_amdgcn_sched_barrier(0);
            slots_a<I + 1>(k_next, v_cur, kbase, vbase, c, c0, c1, n0, n1);
        }
    }

    template <int J>
    __device__ __forceinline__ void slots_b(lds_ptr v_cur, int vbase, float c, const f32x16& c0, const f32x16& c1,
                                            const f32x16& n0, const f32x16& n1) {
        if constexpr (J < NB) {
            this->o[J % DB] = mfma_32x32x16(vf[J % (VPRE + 1)], p_frag(J / DB), this->o[J % DB]);
            if constexpr (J + VPRE < NB) {
                constexpr int JN = J + VPRE;
                vf[JN % (VPRE + 1)] = this->v_frag(v_cur, vbase, JN / DB, JN % DB);
            }
            exp_slot<NA + J>(c0, c1, c);
            max_slot<J>(n0, n1);
            __builtin_amdgcn_sched_barrier(0);
            slots_b<J + 1>(v_cur, vbase, c, c0, c1, n0, n1);
        }
    }

    __device__ __forceinline__ void slot_step(lds_ptr k_next, lds_ptr v_cur, int kbase, int vbase, float c,
                                              const f32x16& cur0, const f32x16& cur1, f32x16& nxt0, f32x16& nxt1,
                                              bool has_next, bool mask_next, int kv0_next, int q_row0, int S,
                                              int lane) {
        sum_a = sum_b = 0.f;
        mx_a = mx_b = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) { nxt0[r] = 0.f; nxt1[r] = 0.f; }
#pragma unroll
        for (int i = 0; i < NPRE; ++i) kf[i] = k_read(k_next, kbase, i);
        __builtin_amdgcn_sched_barrier(0);
        slots_a<0>(k_next, v_cur, kbase, vbase, c, cur0, cur1, nxt0, nxt1);
        if constexpr (C::STAMP) t_mid = stamp();
        slots_b<0>(v_cur, vbase, c, cur0, cur1, nxt0, nxt1);
        if constexpr (C::STAMP) t_end = stamp();
        this->l += sum_a + sum_b;
        if (has_next) {
            float mx = fmaxf(mx_a, mx_b);
            if (mask_next) {
                this->mask(nxt0, nxt1, kv0_next, q_row0, S, lane);
                mx = this->row_max(nxt0, nxt1);
            }
            this->update_max(mx, c);
        }
    }
};

template <class C>
__global__ __launch_bounds__(512, 2) void fwd_bf16_slots_kernel(const Params p) {
    constexpr int D = C::D;
    constexpr bool CAUSAL = C::CAUSAL;
    using OutT = typename C::OutT;
    using Stage = KVStage<D>;
    constexpr int KVBLK = 64, QBLK = 256;
    constexpr int TILE = Stage::TILE_BYTES, SLOT = 2 * TILE;
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

    SlotWave<C> w;
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

    int so_cur = 0, so_nxt = SLOT, so_wr = 2 * SLOT;

    // STAMP builds: cycles per segment summed over tiles: [0] top->loads issued, [1] phase A, [2] phase B,
    // [3] max update, [4] stage write, [5] barrier, [6] tiles
    unsigned long long acc[7] = {0, 0, 0, 0, 0, 0, 0};
    auto step = [&](int t, f32x16& cur0, f32x16& cur1, f32x16& nxt0, f32x16& nxt1) {
        unsigned long long t0 = 0, t1 = 0, t4 = 0, t5 = 0, t6 = 0;
        if constexpr (C::STAMP) t0 = w.stamp();
        const bool more2 = t + 2 < n_tiles;
        if (more2) st.load(Kh, Vh, kSb, vSb, (t + 2) * KVBLK, S, wave, lane);
        if constexpr (C::STAMP) t1 = w.stamp();
        if (t < my_tiles) {
            const bool has_next = t + 1 < my_tiles;
            w.slot_step(smem + so_nxt, smem + so_cur + TILE, kbase, vbase, c, cur0, cur1, nxt0, nxt1, has_next,
                        has_next && needs_mask(t + 1), (t + 1) * KVBLK, q_row0, S, lane);
        }
        if constexpr (C::STAMP) t4 = w.stamp();
        if (more2) st.write(smem + so_wr, smem + so_wr + TILE, wave, lane);
        if constexpr (C::STAMP) t5 = w.stamp();
        __syncthreads();
        if constexpr (C::STAMP) {
            t6 = w.stamp();
            acc[0] += t1 - t0; acc[1] += w.t_mid - t1; acc[2] += w.t_end - w.t_mid; acc[3] += t4 - w.t_end;
            acc[4] += t5 - t4; acc[5] += t6 - t5; acc[6] += 1;
        }
        const int tmp = so_cur;
        so_cur = so_nxt;
        so_nxt = so_wr;
        so_wr = tmp;
    };

    for (int t = 0; t < n_tiles; t += 2) {
        step(t, sA0, sA1, sB0, sB1);
        if (t + 1 < n_tiles) step(t + 1, sB0, sB1, sA0, sA1);
    }

    if constexpr (C::STAMP) {
        if (lane == 0 && p.dbg) {
#pragma unroll
            for (int k = 0; k < 7; ++k) p.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + k] = acc[k];
        }
    }
    if (wave_live) w.template store_o<OutT>(Oh, oSb, q_row0, S, lane);
}

}  // namespace fa
