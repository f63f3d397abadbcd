// kernel_bf16_v4.hip.h -- bf16 MFMA forward kernel, hand-placed slots, everything folded into them.
//
// Measured on MI355X with s_memtime stamps (tests/fa_tune, STAMP variant of kernel_bf16_slots.hip.h):
// per tile and wave, ~890 of ~4270 cycles were segments that all 8 waves run at the same time with
// no MFMA in flight (432: four global loads with 64-bit address arithmetic; 206: row-max ->
// permlane -> compare -> branch chain; 248: vmcnt + four ds_write_b128) and ~710 were barrier
// wait.  This version removes those segments from the critical path:
//
//   * K/V tiles are fetched with BUFFER loads: a per-head descriptor (SGPRs), a per-lane byte
//     offset that never changes, and a scalar tile offset advanced on the SALU -- no per-tile VALU
//     address arithmetic at all; rows past the end of the sequence read as 0 by the hardware range
//     check (no clamping), and the prefetch past the last tile is a harmless out-of-range read;
//   * the four loads of tile t+2 are issued one per slot inside phase A of tile t, and their four
//     ds_write_b128 one per slot inside phase B (ring slot (t+2)%3 is free for the whole iteration);
//   * the row max of S(t+1) uses v_max3 over the first half of phase B, the cross-half exchange and
//     the lazy-rescale decision follow immediately, so the (rare) rescale branch at the end of the
//     tile tests a scalar that has been ready for hundreds of cycles.
//
// Slot contents (d = 128: NA = NB = 16; d = 64: NA = NB = 8), each fenced by sched_barrier(0):
//   A_i : MFMA S(t+1) ; ds_read_b128 K frag i+NPRE ; [buffer_load #i/2 if i odd] ; exp slice
//   B_j : MFMA O^T    ; 2 x ds_read_b64_tr_b16 V frag j+VPRE ; exp slice ; [max3 x2 if j < NB/2]
//         [decision at j = NB/2] ; [ds_write_b128 #k at j = NB/2 + 2k]
#pragma once

#include "kernel_bf16_slots.hip.h"

namespace fa {

template <int D_, bool CAUSAL_, typename OutT_, int THR_ = 8, int NPRE_ = 4, int VPRE_ = 2, bool STAMP_ = false,
          int PRIO_ = 0, bool OPTIMISTIC_ = true>
struct V4Cfg {
    // true: optimistic pass (no per-tile max) + finiteness check + tracked fallback; false: tracked pass only
    static constexpr bool OPTIMISTIC = OPTIMISTIC_;
    static constexpr int D = D_;
    static constexpr bool CAUSAL = CAUSAL_;
    using OutT = OutT_;
    static constexpr int THR = THR_;
    static constexpr int NPRE = NPRE_;
    static constexpr int VPRE = VPRE_;
    static constexpr bool STAMP = STAMP_;
    static constexpr int PRIO = PRIO_;   // 1: waves 4-7 run at s_setprio 1 (static young-half priority)
    static constexpr int SPLIT_B = 8;
    static constexpr bool SCHED = false;
    static constexpr int VALU_A = 0, VALU_B = 0;
};

// K/V tile staging through buffer loads.  Lane -> (key, chunk) maps and LDS images as KVStage<D>.
template <int D>
struct BufStage {
    using Map = KVStage<D>;
    static constexpr int CPT = Map::CPT;       // loads per thread per tensor
    static constexpr int NL = 2 * CPT;         // loads per thread per tile
    __amdgpu_buffer_rsrc_t krsrc, vrsrc;
    // Chunk i of a thread is chunk 0 + 8 (KVStage maps): +128 B in global memory, +8192 B in the K
    // image, +1024 B in the V image -- so ONE per-lane offset per tensor and side, the rest immediates.
    int koff, voff;                            // per-lane byte offset of chunk 0 inside a tile (constant)
    int klds, vlds;                            // per-lane LDS byte offset of chunk 0 inside a tile image
    int ktile, vtile;                          // bytes per 64-key tile step (scalar)
    u32x4 r[NL];                               // staged data: [0,CPT) = K, [CPT,NL) = V

    __device__ __forceinline__ void init(const char* Kh, const char* Vh, int64_t kS_bytes, int64_t vS_bytes, int S,
                                         int wave, int lane) {
        // descriptor inputs are blockIdx / kernarg derived -> wave-uniform; num_records = the head's extent
        krsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, (int)(S * kS_bytes), 0x00020000);
        vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, (int)(S * vS_bytes), 0x00020000);
        ktile = (int)(64 * kS_bytes);
        vtile = (int)(64 * vS_bytes);
        koff = Map::k_key(wave, lane) * (int)kS_bytes + Map::k_chunk(0, lane) * 16;
        voff = Map::v_key(wave, lane) * (int)vS_bytes + Map::v_chunk(0, lane) * 16;
        klds = Map::k_lds_off(Map::k_key(wave, lane), Map::k_chunk(0, lane));
        vlds = Map::v_lds_off(Map::v_key(wave, lane), Map::v_chunk(0, lane));
        static_assert(Map::k_lds_off(0, 8) - Map::k_lds_off(0, 0) == 8192 || CPT == 1, "K image chunk step");
    }
    // load #n of tile `t` (n < CPT: K chunk n, else V chunk n-CPT).  The tile offset goes into the
    // VGPR offset (one v_add with a scalar operand) so the hardware range check certainly covers it.
    template <int N>
    __device__ __forceinline__ void load(int t) {
        if constexpr (N < CPT)
            r[N] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(krsrc, koff + t * ktile + N * 128, 0, 0));
        else
            r[N] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrsrc, voff + t * vtile + (N - CPT) * 128, 0, 0));
    }
    template <int N>
    __device__ __forceinline__ void write(lds_ptr slot_base) const {
        if constexpr (N < CPT) lds_write_b128(slot_base, klds + N * 8192, r[N]);
        else lds_write_b128(slot_base + Map::TILE_BYTES, vlds + (N - CPT) * 1024, r[N]);
    }
    __device__ __forceinline__ void load_all(int t) {
        load<0>(t);
        load<1>(t);
        if constexpr (NL == 4) { load<2>(t); load<3>(t); }
    }
    __device__ __forceinline__ void write_all(lds_ptr slot_base) const {
        write<0>(slot_base);
        write<1>(slot_base);
        if constexpr (NL == 4) { write<2>(slot_base); write<3>(slot_base); }
    }
};

template <class C>
struct V4Wave : SlotWave<C> {
    using SW = SlotWave<C>;
    static constexpr int D = C::D, KS = D / 16, DB = D / 32;
    static constexpr int NA = 2 * KS, NB = 4 * DB;
    static constexpr int NPRE = SW::NPRE, VPRE = SW::VPRE;
    using Stage = BufStage<D>;
    static constexpr int NL = Stage::NL;

    bool need;   // lazy-rescale decision for S(t+1), computed in the middle of phase B (tracked pass)

    // max3 chain over PER values of S(t+1) in slot J (J < NB/2)
    template <int J>
    __device__ __forceinline__ void max3_slot(const f32x16& n0, const f32x16& n1) {
        constexpr int PER = 64 / NB;   // 4 (d=128) or 8 (d=64) values per slot
#pragma unroll
        for (int k = 0; k < PER; k += 2) {
            const int e = J * PER + k;
            const float x0 = e < 16 ? n0[e & 15] : n1[e & 15];
            const float x1 = (e + 1) < 16 ? n0[(e + 1) & 15] : n1[(e + 1) & 15];
            if ((k >> 1) & 1) this->mx_b = fmaxf(fmaxf(this->mx_b, x0), x1);
            else this->mx_a = fmaxf(fmaxf(this->mx_a, x0), x1);
        }
        asm volatile("" : "+v"(this->mx_a), "+v"(this->mx_b));
    }

    __device__ __forceinline__ void decide(float c) {
        const float mx = max_both_halves(fmaxf(this->mx_a, this->mx_b)) * c;
        need = __any(mx > this->m + (float)C::THR);
        this->mx_a = mx;   // keep the scaled row max for the rescale body
    }

    template <int I>
    __device__ __forceinline__ void slots_a(Stage& st, int t_load, lds_ptr k_next, lds_ptr v_cur, int kbase, int vbase,
                                            float c, const f32x16& c0, const f32x16& c1, f32x16& n0, f32x16& n1) {
        if constexpr (I < NA) {
            if constexpr (I < KS) n0 = mfma_32x32x16(this->kf[I % NPRE], this->qf[I % KS], n0);
            else                  n1 = mfma_32x32x16(this->kf[I % NPRE], this->qf[I % KS], n1);
            if constexpr (I + NPRE < NA) this->kf[I % NPRE] = this->k_read(k_next, kbase, I + NPRE);
            if constexpr (I >= NA - VPRE) {
                constexpr int J = I - (NA - VPRE);
                this->vf[J % (VPRE + 1)] = this->v_frag(v_cur, vbase, J / DB, J % DB);
            }
            if constexpr ((I & 1) && (I >> 1) < NL) st.template load<(I >> 1)>(t_load);
            this->template exp_slot<I>(c0, c1, c);
            __builtin_amdgcn_sched_barrier(0);
            slots_a<I + 1>(st, t_load, k_next, v_cur, kbase, vbase, c, c0, c1, n0, n1);
        }
    }

    template <bool TRACK, int J>
    __device__ __forceinline__ void slots_b(const Stage& st, lds_ptr wr_slot, lds_ptr v_cur, int vbase, float c,
                                            const f32x16& c0, const f32x16& c1, const f32x16& n0, const f32x16& n1) {
        if constexpr (J < NB) {
            this->o[J % DB] = mfma_32x32x16(this->vf[J % (VPRE + 1)], this->p_frag(J / DB), this->o[J % DB]);
            if constexpr (J + VPRE < NB) {
                constexpr int JN = J + VPRE;
                this->vf[JN % (VPRE + 1)] = this->v_frag(v_cur, vbase, JN / DB, JN % DB);
            }
            this->template exp_slot<NA + J>(c0, c1, c);
            if constexpr (TRACK && J < NB / 2) max3_slot<J>(n0, n1);
            if constexpr (TRACK && J == NB / 2) decide(c);
            if constexpr (J >= NB / 2 && ((J - NB / 2) & 1) == 0 && (J - NB / 2) / 2 < NL)
                st.template write<(J - NB / 2) / 2>(wr_slot);
            __builtin_amdgcn_sched_barrier(0);
            slots_b<TRACK, J + 1>(st, wr_slot, v_cur, vbase, c, c0, c1, n0, n1);
        }
    }

    // One tile.  TRACK = true: running row max with lazy rescale (always safe).  TRACK = false: the
    // optimistic pass -- m stays the row max of tile 0 and no max / decision / rescale is issued.
    template <bool TRACK>
    __device__ __forceinline__ void v4_step(Stage& st, int t_load, lds_ptr wr_slot, lds_ptr k_next, lds_ptr v_cur,
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
        slots_a<0>(st, t_load, k_next, v_cur, kbase, vbase, c, cur0, cur1, nxt0, nxt1);
        if constexpr (C::STAMP) this->t_mid = this->stamp();
        slots_b<TRACK, 0>(st, wr_slot, v_cur, vbase, c, cur0, cur1, nxt0, nxt1);
        if constexpr (C::STAMP) this->t_end = this->stamp();
        this->l += this->sum_a + this->sum_b;
        // ONE rescale site: the masked (diagonal / ragged) tile only recomputes the scalar decision and the
        // row max.  (Two sites that both multiply O made hipcc copy all 64 accumulator registers twice
        // per tile on the common path.)
        if (has_next && mask_next) {
            this->mask(nxt0, nxt1, kv0_next, q_row0, S, lane);
            if constexpr (TRACK) {
                this->mx_a = this->row_max(nxt0, nxt1);
                this->mx_b = this->mx_a;
                decide(c);
            }
        }
        if constexpr (TRACK) {
            if (has_next && need) {
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
    }

    // True iff this lane's row sum or any of its O accumulators is inf / NaN (x*0 is NaN for both).
    __device__ __forceinline__ bool not_finite() const {
        float acc = this->l * 0.f;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = fmaf(this->o[i][r], 0.f, acc);
        return acc != acc;
    }
};

// One pass over all KV tiles of the workgroup's query block.  Returns (workgroup-uniform) whether
// the result has to be recomputed with max tracking (only ever true for TRACK = false).
template <class C, bool TRACK>
__device__ __forceinline__ bool v4_pass(const Params& p, V4Wave<C>& w, BufStage<C::D>& st, lds_ptr smem, int n_tiles,
                                        int my_tiles, int q_row0, int wave, int lane, unsigned long long (&acc)[7]) {
    constexpr int D = C::D;
    constexpr bool CAUSAL = C::CAUSAL;
    constexpr int KVBLK = 64;
    constexpr int TILE = KVStage<D>::TILE_BYTES, SLOT = 2 * TILE;
    const int S = p.S;
    w.init();
    st.load_all(0);
    st.write_all(smem);
    st.load_all(1);              // past-the-end tiles read as zeros (buffer range check)
    st.write_all(smem + SLOT);
    __syncthreads();

    const int kbase = k_read_base(lane);
    const int vbase = v_read_base(lane);
    const float c = p.scale_log2;
    auto needs_mask = [&](int t) { return (CAUSAL && t * KVBLK + KVBLK - 1 > q_row0) || (t * KVBLK + KVBLK > S); };

    f32x16 sA0, sA1, sB0, sB1;
    if (my_tiles > 0) {
        w.qk(smem, kbase, sA0, sA1);
        if (needs_mask(0)) w.mask(sA0, sA1, 0, q_row0, S, lane);
        w.update_max(w.row_max(sA0, sA1), c);   // m = row max of tile 0 (the reference of the optimistic pass)
    }

    int so_cur = 0, so_nxt = SLOT, so_wr = 2 * SLOT;
    auto step = [&](int t, f32x16& cur0, f32x16& cur1, f32x16& nxt0, f32x16& nxt1) {
        unsigned long long t0 = 0, t4 = 0, t6 = 0;
        if constexpr (C::STAMP) t0 = w.stamp();
        if (t < my_tiles) {
            const bool has_next = t + 1 < my_tiles;
            w.template v4_step<TRACK>(st, t + 2, smem + so_wr, smem + so_nxt, smem + so_cur + TILE, kbase, vbase, c, cur0,
                                      cur1, nxt0, nxt1, has_next, has_next && needs_mask(t + 1), (t + 1) * KVBLK, q_row0, S,
                                      lane);
        } else {
            // wave already past its causal diagonal: it still stages its share of the tile
            st.load_all(t + 2);
            st.write_all(smem + so_wr);
        }
        if constexpr (C::STAMP) t4 = w.stamp();
        __syncthreads();
        if constexpr (C::STAMP) {
            t6 = w.stamp();
            acc[1] += w.t_mid - t0; acc[2] += w.t_end - w.t_mid; acc[3] += t4 - w.t_end; acc[5] += t6 - t4; acc[6] += 1;
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
    if constexpr (TRACK) return false;
    else return __syncthreads_or(my_tiles > 0 && w.not_finite()) != 0;
}

template <class C>
__global__ __launch_bounds__(512, 2) void fwd_bf16_v4_kernel(const Params p) {
    constexpr int D = C::D;
    constexpr bool CAUSAL = C::CAUSAL;
    using OutT = typename C::OutT;
    using Stage = BufStage<D>;
    constexpr int KVBLK = 64, QBLK = 256;
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

    if constexpr (C::PRIO == 1) {
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    }

    V4Wave<C> w;
    w.load_q(Qh, qSb, q_row0, S, lane);
    w.pin_q();
    Stage st;
    st.init(Kh, Vh, kSb, vSb, S, wave, lane);
    unsigned long long acc[7] = {0, 0, 0, 0, 0, 0, 0};

    if constexpr (C::OPTIMISTIC) {
        // Optimistic pass: exponentials relative to the row max of tile 0, no per-tile max tracking.
        // exp2 / bf16 / f32 accumulation have ~2^127 of headroom above that reference; if a later score
        // exceeds it (or P.V overflows) l or O becomes inf/NaN, which the check at the end of the pass
        // catches, and the whole workgroup redoes its block with the tracked (always safe) pass.
        if (v4_pass<C, false>(p, w, st, smem, n_tiles, my_tiles, q_row0, wave, lane, acc))
            v4_pass<C, true>(p, w, st, smem, n_tiles, my_tiles, q_row0, wave, lane, acc);
    } else {
        v4_pass<C, true>(p, w, st, smem, n_tiles, my_tiles, q_row0, wave, lane, acc);
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
