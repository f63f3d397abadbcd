// kernel_bf16_v5.hip.h -- bf16 MFMA forward kernel, NW waves per workgroup (4 or 8), separate
// 2-slot K and V rings.
//
// Why: with one 8-wave workgroup per CU (kernel_bf16_v4.hip.h) the two waves that share a SIMD are
// coupled by the workgroup barrier, so they reach the non-MFMA parts of a tile (tile hand-over,
// barrier wait -- measured ~15 % of a tile) at the same time and the matrix pipe idles.  With NW = 4
// a workgroup is one wave per SIMD and TWO workgroups are resident per CU (64 KiB LDS each): the two
// waves of a SIMD belong to different workgroups, drift apart, and each one's stalls are covered by
// the other's MFMAs.
//
// LDS per workgroup = [K slot 0 | K slot 1 | V slot 0 | V slot 1] (4 x 16 KiB at d = 128).  Tile t's K
// lives in K slot t%2, its V in V slot t%2.  In iteration t a wave reads K(t+1) and V(t), so
//     V(t+1) -> V slot (t+1)%2   (held V(t-1): last read in iteration t-1)   written in phase A
//     K(t+2) -> K slot  t%2      (held K(t):   last read in iteration t-1)   written in phase B
// are both safe under ONE barrier per tile.  The loop is unrolled x2, so every slot address is a
// compile-time constant.  One staging register set is time-shared:
//     [V(t+1) data, loaded late in iteration t-1] --ds_write, early phase A-->
//     [K(t+2) loads, late phase A] --ds_write, second half of phase B-->
//     [V(t+2) loads, interleaved right behind those writes] --> next iteration.
#pragma once

#include "kernel_bf16_v4.hip.h"

namespace fa {

template <int D_, bool CAUSAL_, typename OutT_, int NW_ = 4, int THR_ = 8, int NPRE_ = 4, int VPRE_ = 2>
struct V5Cfg {
    static constexpr int D = D_;
    static constexpr bool CAUSAL = CAUSAL_;
    using OutT = OutT_;
    static constexpr int NW = NW_;
    static constexpr int THR = THR_;
    static constexpr int NPRE = NPRE_;
    static constexpr int VPRE = VPRE_;
    static constexpr bool STAMP = false;
    static constexpr int PRIO = 0;
    static constexpr bool OPTIMISTIC = false;
    static constexpr int SPLIT_B = 8;
    static constexpr bool SCHED = false;
    static constexpr int VALU_A = 0, VALU_B = 0;
};

// Staging for NW waves: a 64-key tile is 8 key groups x (D/64) 128-byte column halves = 8*D/64
// wave-instructions of 8 keys x 128 B; each wave issues CPT = 8*(D/64)/NW of them per tensor.
template <int D, int NW>
struct RingStage {
    using Map = KVStage<D>;
    static constexpr int HALVES = D / 64;
    static constexpr int CPT = 8 * HALVES / NW;
    static constexpr int GPW = 8 / NW;          // key groups per wave
    static constexpr int TILE = Map::TILE_BYTES;
    static_assert(CPT >= 1 && 8 % NW == 0, "NW must be 4 or 8");

    __amdgpu_buffer_rsrc_t krsrc, vrsrc;
    int koff, voff;        // per-lane byte offset of instruction 0 inside a tile
    int klds, vlds;        // per-lane LDS byte offset of instruction 0 inside a tile image
    int ktile, vtile;      // bytes per tile step
    int kgrp, vgrp;        // bytes per 8-key group step in global memory
    u32x4 r[CPT];          // ONE staging set, time-shared between K and V

    __device__ __forceinline__ void init(const char* Kh, const char* Vh, int64_t kS_bytes, int64_t vS_bytes, int S,
                                         int wave, int lane) {
        krsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, (int)(S * kS_bytes), 0x00020000);
        vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, (int)(S * vS_bytes), 0x00020000);
        ktile = (int)(64 * kS_bytes);
        vtile = (int)(64 * vS_bytes);
        kgrp = (int)(8 * kS_bytes);
        vgrp = (int)(8 * vS_bytes);
        const int g0 = wave * GPW;   // first key group of this wave
        const int kk = 8 * g0 + (lane & 7), kc = lane >> 3;
        const int vk = 8 * g0 + 2 * ((lane >> 3) & 3) + ((lane >> 2) & 1), vc = 4 * (lane >> 5) + (lane & 3);
        koff = kk * (int)kS_bytes + kc * 16;
        voff = vk * (int)vS_bytes + vc * 16;
        klds = Map::k_lds_off(kk, kc);
        vlds = Map::v_lds_off(vk, vc);
    }
    // instruction i of this wave: key group g0 + i/HALVES, column half i%HALVES
    template <int I>
    __device__ __forceinline__ void load_k(int t) {
        r[I] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                             krsrc, koff + t * ktile + (I / HALVES) * kgrp + (I % HALVES) * 128, 0, 0));
    }
    template <int I>
    __device__ __forceinline__ void load_v(int t) {
        r[I] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                             vrsrc, voff + t * vtile + (I / HALVES) * vgrp + (I % HALVES) * 128, 0, 0));
    }
    template <int I>
    __device__ __forceinline__ void write_k(lds_ptr kslot) const {
        lds_write_b128(kslot, klds + (I / HALVES) * 128 + (I % HALVES) * 8192, r[I]);     // +8 keys: +128 B; +8 chunks: +8 KiB
    }
    template <int I>
    __device__ __forceinline__ void write_v(lds_ptr vslot) const {
        lds_write_b128(vslot, vlds + (I / HALVES) * (Map::DB * 512) + (I % HALVES) * 1024, r[I]);
    }
    template <int I = 0> __device__ __forceinline__ void load_k_all(int t) { if constexpr (I < CPT) { load_k<I>(t); load_k_all<I + 1>(t); } }
    template <int I = 0> __device__ __forceinline__ void load_v_all(int t) { if constexpr (I < CPT) { load_v<I>(t); load_v_all<I + 1>(t); } }
    template <int I = 0> __device__ __forceinline__ void write_k_all(lds_ptr s) const { if constexpr (I < CPT) { write_k<I>(s); write_k_all<I + 1>(s); } }
    template <int I = 0> __device__ __forceinline__ void write_v_all(lds_ptr s) const { if constexpr (I < CPT) { write_v<I>(s); write_v_all<I + 1>(s); } }
};

template <class C>
struct V5Wave : V4Wave<C> {
    static constexpr int D = C::D, KS = D / 16, DB = D / 32;
    static constexpr int NA = 2 * KS, NB = 4 * DB;
    static constexpr int NPRE = V4Wave<C>::NPRE, VPRE = V4Wave<C>::VPRE;
    using Stage = RingStage<D, C::NW>;
    static constexpr int CPT = Stage::CPT;
    static_assert(2 * CPT <= NA / 2 && 2 * CPT <= NB / 2, "staging does not fit the slot plan");

    // phase A slot I: MFMA S(t+1); K frag read; [write V(t+1) #i at I = 1+2i]; [load K(t+2) #i at I = NA/2+1+2i]
    template <int I>
    __device__ __forceinline__ void slots_a(Stage& st, int t, lds_ptr k_next, lds_ptr v_cur, lds_ptr v_wr, int kbase,
                                            int vbase, float c, const f32x16& c0, const f32x16& c1, f32x16& n0, f32x16& n1) {
        if constexpr (I < NA) {
            if constexpr (I < KS) n0 = mfma_32x32x16(this->kf[I % NPRE], this->qf[I % KS], n0);
            else                  n1 = mfma_32x32x16(this->kf[I % NPRE], this->qf[I % KS], n1);
            if constexpr (I + NPRE < NA) this->kf[I % NPRE] = this->k_read(k_next, kbase, I + NPRE);
            if constexpr (I >= NA - VPRE) {
                constexpr int J = I - (NA - VPRE);
                this->vf[J % (VPRE + 1)] = this->v_frag(v_cur, vbase, J / DB, J % DB);
            }
            if constexpr ((I & 1) && I < NA / 2 && (I >> 1) < CPT) st.template write_v<(I >> 1)>(v_wr);
            if constexpr ((I & 1) && I > NA / 2 && ((I - NA / 2) >> 1) < CPT) st.template load_k<((I - NA / 2) >> 1)>(t + 2);
            this->template exp_slot<I>(c0, c1, c);
            __builtin_amdgcn_sched_barrier(0);
            slots_a<I + 1>(st, t, k_next, v_cur, v_wr, kbase, vbase, c, c0, c1, n0, n1);
        }
    }
    // phase B slot J: MFMA O^T; V^T frag reads; max3 (first half); decide; [write K(t+2) #i at J = NB/2+2i];
    // [load V(t+2) #i at J = NB/2+1+2i]
    template <int J>
    __device__ __forceinline__ void slots_b(Stage& st, int t, lds_ptr k_wr, lds_ptr v_cur, int vbase, float c,
                                            const f32x16& c0, const f32x16& c1, const f32x16& n0, const f32x16& n1) {
        if constexpr (J < NB) {
            this->o[J % DB] = mfma_32x32x16(this->vf[J % (VPRE + 1)], this->p_frag(J / DB), this->o[J % DB]);
            if constexpr (J + VPRE < NB) {
                constexpr int JN = J + VPRE;
                this->vf[JN % (VPRE + 1)] = this->v_frag(v_cur, vbase, JN / DB, JN % DB);
            }
            this->template exp_slot<NA + J>(c0, c1, c);
            if constexpr (J < NB / 2) this->template max3_slot<J>(n0, n1);
            if constexpr (J == NB / 2) this->decide(c);
            if constexpr (J >= NB / 2 && ((J - NB / 2) & 1) == 0 && (J - NB / 2) / 2 < CPT) st.template write_k<(J - NB / 2) / 2>(k_wr);
            if constexpr (J > NB / 2 && ((J - NB / 2) & 1) == 1 && (J - NB / 2) / 2 < CPT) st.template load_v<(J - NB / 2) / 2>(t + 2);
            __builtin_amdgcn_sched_barrier(0);
            slots_b<J + 1>(st, t, k_wr, v_cur, vbase, c, c0, c1, n0, n1);
        }
    }

    __device__ __forceinline__ void v5_step(Stage& st, int t, lds_ptr k_next, lds_ptr k_wr, lds_ptr v_cur, lds_ptr v_wr,
                                            int kbase, int vbase, float c, const f32x16& cur0, const f32x16& cur1,
                                            f32x16& nxt0, f32x16& nxt1, bool has_next, bool mask_next, int kv0_next,
                                            int q_row0, int S, int lane) {
        this->sum_a = this->sum_b = 0.f;
        this->mx_a = this->mx_b = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) { nxt0[r] = 0.f; nxt1[r] = 0.f; }
#pragma unroll
        for (int i = 0; i < NPRE; ++i) this->kf[i] = this->k_read(k_next, kbase, i);
        __builtin_amdgcn_sched_barrier(0);
        slots_a<0>(st, t, k_next, v_cur, v_wr, kbase, vbase, c, cur0, cur1, nxt0, nxt1);
        slots_b<0>(st, t, k_wr, v_cur, vbase, c, cur0, cur1, nxt0, nxt1);
        this->l += this->sum_a + this->sum_b;
        if (has_next && mask_next) {
            this->mask(nxt0, nxt1, kv0_next, q_row0, S, lane);
            this->mx_a = this->row_max(nxt0, nxt1);
            this->mx_b = this->mx_a;
            this->decide(c);
        }
        if (has_next && this->need) {
            const float mn = fmaxf(this->m, this->mx_a);
            const float alpha = fast_exp2(this->m - mn);
            this->m = mn;
            this->l *= alpha;
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) this->o[i][r] *= alpha;
        }
    }
};

template <class C>
__global__ __launch_bounds__(C::NW * 64, 2) void fwd_bf16_v5_kernel(const Params p) {
    constexpr int D = C::D, NW = C::NW;
    constexpr bool CAUSAL = C::CAUSAL;
    using OutT = typename C::OutT;
    using Stage = RingStage<D, NW>;
    constexpr int KVBLK = 64, QBLK = 32 * NW;
    constexpr int TILE = Stage::TILE;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_ptr smem = (lds_ptr)smem_raw;
    lds_ptr K0 = smem, K1 = smem + TILE, V0 = smem + 2 * TILE, V1 = smem + 3 * TILE;

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

    V5Wave<C> w;
    w.init();
    w.load_q(Qh, qSb, q_row0, S, lane);

    Stage st;
    st.init(Kh, Vh, kSb, vSb, S, wave, lane);
    st.load_k_all(0); st.write_k_all(K0);
    st.load_k_all(1); st.write_k_all(K1);
    st.load_v_all(0); st.write_v_all(V0);
    st.load_v_all(1);                       // stays in registers: written in phase A of iteration 0
    w.pin_q();
    __syncthreads();

    const int kbase = k_read_base(lane);
    const int vbase = v_read_base(lane);
    const float c = p.scale_log2;

    auto needs_mask = [&](int t) { return (CAUSAL && t * KVBLK + KVBLK - 1 > q_row0) || (t * KVBLK + KVBLK > S); };

    f32x16 sA0, sA1, sB0, sB1;
    if (my_tiles > 0) {
        w.qk(K0, kbase, sA0, sA1);
        if (needs_mask(0)) w.mask(sA0, sA1, 0, q_row0, S, lane);
        w.update_max(w.row_max(sA0, sA1), c);
    }
    __syncthreads();   // every wave has finished reading K slot 0 before iteration 0 refills it with K(2)

    // even t: read K1 / V0, write V1 then K0.   odd t: read K0 / V1, write V0 then K1.
    auto step = [&](int t, lds_ptr k_next, lds_ptr k_wr, lds_ptr v_cur, lds_ptr v_wr, f32x16& cur0, f32x16& cur1,
                    f32x16& nxt0, f32x16& nxt1) {
        if (t < my_tiles) {
            const bool has_next = t + 1 < my_tiles;
            w.v5_step(st, t, k_next, k_wr, v_cur, v_wr, kbase, vbase, c, cur0, cur1, nxt0, nxt1, has_next,
                      has_next && needs_mask(t + 1), (t + 1) * KVBLK, q_row0, S, lane);
        } else {
            // past this wave's causal diagonal: keep staging its share of the tiles
            st.write_v_all(v_wr);
            st.load_k_all(t + 2);
            st.write_k_all(k_wr);
            st.load_v_all(t + 2);
        }
        __syncthreads();
    };

    for (int t = 0; t < n_tiles; t += 2) {
        step(t, K1, K0, V0, V1, sA0, sA1, sB0, sB1);
        if (t + 1 < n_tiles) step(t + 1, K0, K1, V1, V0, sB0, sB1, sA0, sA1);
    }

    if (wave_live) w.template store_o<OutT>(Oh, oSb, q_row0, S, lane);
}

}  // namespace fa
