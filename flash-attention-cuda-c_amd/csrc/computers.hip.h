// computers.hip.h -- per-wave compute of the fused forward pass: QK^T (MFMA) -> online softmax
// (in registers) -> PV (MFMA), cut into hand-placed slots.
//
// Counterpart of the reference's kernels/computers.cuh:5-69 (twoLoaderMhaComputeWarp) and the
// device helpers it calls, kernels/utils.cuh:17-45 (computeTileScore), :58-81
// (updateSoftmaxState), :93-113 (multiplyVAccumulateO).  Same algorithm -- blocked scores, running
// (max, sum) per query row, rescaled accumulation of P.V -- re-designed for CDNA4:
//
//   * one WAVE owns 32 query rows (the reference: one warp per row); a KV tile is 64 keys;
//   * scores come from v_mfma_f32_32x32x16_{bf16, fp8_fp8} in the SWAPPED orientation S^T = K.Q^T, so
//     the 32 scores of one query row and key half sit in ONE lane's registers: row max / row sum are
//     in-lane ops plus one v_permlane32_swap (the reference: cg::reduce + Bc-1 shuffles, utils.cuh:66-73);
//   * FA-2 recurrence: O and l stay un-normalised, one divide at the end (the reference renormalises
//     every tile, utils.cuh:75-80); exp is exp2 with scale*log2(e) folded into one FMA;
//   * P^T (accumulator layout: key in the register index, query on the lane), rounded to bf16, IS the
//     B operand of O^T += V^T.P^T; V^T fragments come from ds_read_b64_tr_b16.  O lives in 16*D/32
//     accumulator registers (the reference keeps O in shared memory, utils.cuh:107-111).
//
// Schedule.  hipcc, left alone (or nudged with sched_group_barrier), hoists the whole softmax in front
// of the MFMAs and makes each QK^T MFMA wait on an LDS read issued just before it.  A tile is therefore
// cut into NA + NB slots, one MFMA each, fenced by __builtin_amdgcn_sched_barrier(0); iteration t of a wave:
//
//   A_i (NA = 2*D/16): MFMA  S(t+1) += Kfrag . Qfrag          (QK^T of the NEXT tile, K(t+1) in LDS)
//                      read  K fragment NPRE ahead (later: the first V^T fragments)
//                      [buffer_load #n of tile t+2 at i = 1+2n]
//                      VALU  exp2 / row sum / bf16 pack of this slot's share of P(t)
//   B_j (NB = 4*D/32): MFMA  O^T += V^T(t)frag . P(t)frag
//                      read  V^T fragment VPRE ahead (2 x ds_read_b64_tr_b16)
//                      VALU  rest of P(t); tracked pass: v_max3 of S(t+1) for j < NB/2, decision at j = NB/2
//                      [ds_write_b128 #n of tile t+2 at j = NB/2 + 2n]
//
// Score element e of the 32 a lane holds goes to overall slot e*(NA + 3*NB/4)/32, which meets the
// deadlines "P group g (elements 8g..8g+7) complete before PV slot g*NB/4" with <= 1.25 elements per
// 32-cycle MFMA gap at d = 128.  Empty asm statements pin the partial sums / maxima in their slot
// (hipcc otherwise sinks them behind the MFMAs).
#pragma once

#include "loaders.hip.h"

namespace fa {

template <class C>
struct WaveCompute {
    static constexpr int D = C::D, ESZ = C::ESZ;
    static constexpr int KS = D / 16;              // MFMA k-steps of one 32-key half of QK^T
    static constexpr int DB = D / 32;              // 32-wide d blocks of O^T
    static constexpr int NA = 2 * KS, NB = 4 * DB; // slots of phase A / phase B
    static constexpr int MPF = ESZ == 1 ? 2 : 1;   // MFMAs fed by one 16-byte K (or Q) fragment
    static constexpr int NF = NA / MPF;            // K fragment reads per tile
    static constexpr int FPH = NF / 2;             // ... per 32-key half = Q fragment count
    static constexpr int NPRE = C::NPRE < NF ? C::NPRE : NF;
    static constexpr int VPRE = C::VPRE;
    static constexpr int SPAN = NA + (3 * NB) / 4; // overall slots the exponentials are spread over
    using G = TileGeom<D, ESZ>;
    using Stage = BufStage<D, ESZ>;
    static constexpr int NL = Stage::NL, NW = Stage::NW;

    // ---- state that lives across tiles ----
    u32x4 qf[FPH];     // Q fragments (16 bytes each: one bf16 MFMA operand, or two fp8 operands)
    f32x16 o[DB];      // O^T accumulators: row = d, col = query
    float m;           // reference max used for exponentiation (scaled, log2 domain)
    float l;           // partial row sum (this lane's key half)
    // ---- per-tile scratch ----
    u32x4 kf[NPRE];    // K fragment window
    bf16x8 vf[VPRE + 1];
    uint32_t pw[16];   // P(t) as packed bf16 pairs: word 4g+w = elements 8g+2w, 8g+2w+1
    float sum_a, sum_b, mx_a, mx_b, p_even;
    bool need;         // tracked pass: lazy-rescale decision for S(t+1)
    unsigned long long t_mid = 0, t_end = 0;   // STAMP builds only

    __host__ __device__ static constexpr int elem_slot(int e) { return e * SPAN / 32; }

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
        m = -INFINITY;
        l = 0.f;
    }

    // Q fragment u of row q: 16 bytes at byte 32u + 16h of the row.  bf16: d = 16u + 8h .. +7 (k-step u).
    // fp8: d = 32u + 16h .. +15 -- the contraction order is permuted the same way for K (chunk 2u+h of
    // the K image), so one 16-byte fragment feeds two MFMAs.
    __device__ __forceinline__ void load_q(const char* Qh, int64_t qS_bytes, int row0, int S, int lane) {
        int r = row0 + (lane & 31);
        r = r < S ? r : S - 1;
        const char* src = Qh + r * qS_bytes + (lane >> 5) * 16;
#pragma unroll
        for (int u = 0; u < FPH; ++u) qf[u] = *reinterpret_cast<const u32x4*>(src + u * 32);
    }
    // Make the Q fragments look "consumed" so hipcc waits for their loads HERE and not with a
    // pessimistic vmcnt inside the main loop (where it would also drain the tile prefetch).
    __device__ __forceinline__ void pin_q() {
#pragma unroll
        for (int u = 0; u < FPH; ++u) asm volatile("" : "+v"(qf[u]));
    }

    __device__ __forceinline__ u32x4 k_read(lds_ptr kimg, int kbase, int f) const {
        return __builtin_bit_cast(u32x4, lds_read_b128(kimg, kbase + (f % FPH) * 2048 + (f / FPH) * 512));
    }
    // MFMA #i of QK^T (i = MPF*f + sub) from K fragment kfrag (= fragment f)
    template <int I>
    __device__ __forceinline__ void qk_mfma(const u32x4& kfrag, f32x16& n0, f32x16& n1) const {
        constexpr int f = I / MPF, sub = I % MPF;
        const u32x4& q = qf[f % FPH];
        f32x16& acc = (f < FPH) ? n0 : n1;
        if constexpr (ESZ == 2) {
            acc = mfma_32x32x16(__builtin_bit_cast(bf16x8, kfrag), __builtin_bit_cast(bf16x8, q), acc);
        } else {
            const uint64_t a = (uint64_t)kfrag[2 * sub] | ((uint64_t)kfrag[2 * sub + 1] << 32);
            const uint64_t b = (uint64_t)q[2 * sub] | ((uint64_t)q[2 * sub + 1] << 32);
            acc = mfma_32x32x16_fp8(a, b, acc);
        }
    }

    // S^T(both 32-key halves) = K.Q^T, compiler-scheduled: used once per pass for tile 0.
    template <int I = 0>
    __device__ __forceinline__ void qk_all(lds_ptr kimg, int kbase, f32x16& s0, f32x16& s1) {
        if constexpr (I == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
        }
        if constexpr (I < NA) {
            if constexpr (I % MPF == 0) kf[0] = k_read(kimg, kbase, I / MPF);
            qk_mfma<I>(kf[0], s0, s1);
            qk_all<I + 1>(kimg, kbase, s0, s1);
        }
    }

    // Diagonal / ragged tile: key index > query index, or key index >= S  ->  -inf.
    // s0[r] holds key kv0 + acc_row(r,h), s1[r] key kv0 + 32 + acc_row(r,h); query q_row0 + (lane&31).
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
    // Tile 0 of a pass: m = its row max (m = -inf before, so alpha = 0 and O, l stay 0).
    __device__ __forceinline__ void first_max(float mx_raw, float c) { m = fmaxf(m, max_both_halves(mx_raw) * c); }

    // ---- softmax slices ------------------------------------------------------------------------
    template <int E>
    __device__ __forceinline__ void exp_elem(const f32x16& c0, const f32x16& c1, float c) {
        const float x = E < 16 ? c0[E & 15] : c1[E & 15];
        const float p = fast_exp2(fmaf(x, c, -m));
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
    // tracked pass: v_max3 chain over 64/NB values of S(t+1) in slot J (J < NB/2)
    template <int J>
    __device__ __forceinline__ void max3_slot(const f32x16& n0, const f32x16& n1) {
        constexpr int PER = 64 / NB;
#pragma unroll
        for (int k = 0; k < PER; k += 2) {
            const int e = J * PER + k;
            const float x0 = e < 16 ? n0[e & 15] : n1[e & 15];
            const float x1 = (e + 1) < 16 ? n0[(e + 1) & 15] : n1[(e + 1) & 15];
            if ((k >> 1) & 1) mx_b = fmaxf(fmaxf(mx_b, x0), x1);
            else mx_a = fmaxf(fmaxf(mx_a, x0), x1);
        }
        asm volatile("" : "+v"(mx_a), "+v"(mx_b));
    }
    __device__ __forceinline__ void decide(float c) {
        const float mx = max_both_halves(fmaxf(mx_a, mx_b)) * c;
        need = __any(mx > m + (float)C::THR);
        mx_a = mx;   // keep the scaled row max for the rescale body
    }

    // V^T A-fragment of 16-key step s4, d block db: two transposed reads.  Element j of lane half h is
    // key 16*s4 + 8*(j>>2) + 4h + (j&3) -- exactly the key order of the packed P^T fragment.
    __device__ __forceinline__ bf16x8 v_frag(lds_ptr vimg, int vbase, int s4, int db) const {
        const s16x4 lo = lds_read_tr16_b64(vimg, vbase + (2 * s4) * (DB * 512) + db * 512);
        const s16x4 hi = lds_read_tr16_b64(vimg, vbase + (2 * s4 + 1) * (DB * 512) + db * 512);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }

    // ---- the slots -----------------------------------------------------------------------------
    template <int I>
    __device__ __forceinline__ void slots_a(Stage& st, int t_load, lds_ptr k_next, lds_ptr v_cur, int kbase, int vbase,
                                            float c, const f32x16& c0, const f32x16& c1, f32x16& n0, f32x16& n1) {
        if constexpr (I < NA) {
            constexpr int f = I / MPF;
            qk_mfma<I>(kf[f % NPRE], n0, n1);
            if constexpr (I % MPF == MPF - 1 && f + NPRE < NF) kf[f % NPRE] = k_read(k_next, kbase, f + NPRE);
            if constexpr (I >= NA - VPRE) {   // the last VPRE phase-A slots start the V^T window of phase B
                constexpr int J = I - (NA - VPRE);
                vf[J % (VPRE + 1)] = v_frag(v_cur, vbase, J / DB, J % DB);
            }
            if constexpr ((I & 1) && (I >> 1) < NL) st.template load<(I >> 1)>(t_load);
            exp_slot<I>(c0, c1, c);
            __builtin_amdgcn_sched_barrier(0);
            slots_a<I + 1>(st, t_load, k_next, v_cur, kbase, vbase, c, c0, c1, n0, n1);
        }
    }
    template <bool TRACK, int J>
    __device__ __forceinline__ void slots_b(const Stage& st, lds_ptr wr_slot, lds_ptr v_cur, int vbase, float c,
                                            const f32x16& c0, const f32x16& c1, const f32x16& n0, const f32x16& n1) {
        if constexpr (J < NB) {
            o[J % DB] = mfma_32x32x16(vf[J % (VPRE + 1)], p_frag(J / DB), o[J % DB]);
            if constexpr (J + VPRE < NB) {
                constexpr int JN = J + VPRE;
                vf[JN % (VPRE + 1)] = v_frag(v_cur, vbase, JN / DB, JN % DB);
            }
            exp_slot<NA + J>(c0, c1, c);
            if constexpr (TRACK && J < NB / 2) max3_slot<J>(n0, n1);
            if constexpr (TRACK && J == NB / 2) decide(c);
            if constexpr (J >= NB / 2 && ((J - NB / 2) & 1) == 0 && (J - NB / 2) / 2 < NW)
                st.template write<(J - NB / 2) / 2>(wr_slot);
            __builtin_amdgcn_sched_barrier(0);
            slots_b<TRACK, J + 1>(st, wr_slot, v_cur, vbase, c, c0, c1, n0, n1);
        }
    }

    // One tile: cur = S(t) (consumed), nxt = S(t+1) (produced).  On the wave's last tile (has_next ==
    // false) the QK^T of the non-existent next tile is still issued -- its result is never looked at --
    // so that there is ONE hot code path.
    // TRACK = true: running row max with lazy rescale (always safe).  TRACK = false: the optimistic
    // pass -- m stays the row max of tile 0 and no max / decision / rescale is issued.
    template <bool TRACK>
    __device__ __forceinline__ void tile_step(Stage& st, int t_load, lds_ptr wr_slot, lds_ptr k_next, lds_ptr v_cur,
                                              int kbase, int vbase, float c, const f32x16& cur0, const f32x16& cur1,
                                              f32x16& nxt0, f32x16& nxt1, bool has_next, bool mask_next, int kv0_next,
                                              int q_row0, int S, int lane) {
        sum_a = sum_b = 0.f;
        mx_a = mx_b = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) { nxt0[r] = 0.f; nxt1[r] = 0.f; }
#pragma unroll
        for (int i = 0; i < NPRE; ++i) kf[i] = k_read(k_next, kbase, i);
        __builtin_amdgcn_sched_barrier(0);
        slots_a<0>(st, t_load, k_next, v_cur, kbase, vbase, c, cur0, cur1, nxt0, nxt1);
        if constexpr (C::STAMP) t_mid = cycle_stamp();
        slots_b<TRACK, 0>(st, wr_slot, v_cur, vbase, c, cur0, cur1, nxt0, nxt1);
        if constexpr (C::STAMP) t_end = cycle_stamp();
        l += sum_a + sum_b;
        // ONE rescale site: the masked (diagonal / ragged) tile only recomputes the scalar decision and
        // the row max.  (Two sites that both multiply O made hipcc copy all 64 accumulator registers
        // twice per tile on the common path.)
        if (has_next && mask_next) {
            mask(nxt0, nxt1, kv0_next, q_row0, S, lane);
            if constexpr (TRACK) {
                mx_a = row_max(nxt0, nxt1);
                mx_b = mx_a;
                decide(c);
            }
        }
        if constexpr (TRACK) {
            if (has_next && need) {
                const float mn = fmaxf(m, mx_a);
                const float alpha = fast_exp2(m - mn);
                m = mn;
                l *= alpha;
#pragma unroll
                for (int i = 0; i < DB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            }
        }
    }

    // True iff this lane's row sum or any of its O accumulators is inf / NaN (x*0 is NaN for both).
    __device__ __forceinline__ bool not_finite() const {
        float acc = l * 0.f;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = fmaf(o[i][r], 0.f, acc);
        return acc != acc;
    }

    // ---- epilogues -----------------------------------------------------------------------------
    // Direct form: divide by the row sum and store O[q][d].  A lane holds d = 32db + 8g4 + 4h + (0..3).
    // Used for 4-byte outputs (the reference's float* O).
    // ln sum_k exp(scale*s_k) = (m + log2 l) * ln 2   (m is the reference max in the scaled log2 domain)
    __device__ __forceinline__ void store_lse(float* lse_head, float l_tot, int row0, int S, int lane) const {
        const int qi = row0 + (lane & 31);
        if (lse_head && lane < 32 && qi < S) lse_head[qi] = (m + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
    }

    template <typename OutT>
    __device__ __forceinline__ void store_o(char* Oh, float* lse_head, int64_t oS_bytes, int row0, int S, int lane) {
        const float l_tot = sum_both_halves(l);
        store_lse(lse_head, l_tot, row0, S, lane);
        const float inv = 1.0f / l_tot;
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
    // 2-byte outputs: O^T accumulators -> this wave's private LDS region as a row-major [32 rows][D] tile
    // -> whole rows back out with 16-byte stores (each 16- or 8-lane group writes one full row).  The
    // direct form issues 16 eight-byte stores per lane that touch 32 rows each: ~8k cycles per workgroup,
    // store-issue bound; this form ~3.3k.  16-byte chunk c of row r sits at chunk c ^ (r & mask), so the
    // ds_write_b64 of 16 lanes (16 rows, same column) spread over all banks.  `region` = 32*D*2 bytes
    // private to this wave; the caller guarantees the K/V ring is dead.
    template <typename OutT>
    __device__ __forceinline__ void store_o_lds(lds_ptr region, char* Oh, float* lse_head, int64_t oS_bytes, int row0, int S,
                                                int lane) {
        static_assert(sizeof(OutT) == 2, "LDS epilogue is for bf16 / f16 outputs");
        constexpr int ROWB = D * 2, CHUNKS = ROWB / 16;
        const float l_tot = sum_both_halves(l);
        store_lse(lse_head, l_tot, row0, S, lane);
        const float inv = 1.0f / l_tot;
        const int q = lane & 31, h = lane >> 5;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = 32 * db + 8 * g4 + 4 * h;                  // 4 consecutive d = 8 bytes
                const float a = o[db][4 * g4 + 0] * inv, b = o[db][4 * g4 + 1] * inv;
                const float c2 = o[db][4 * g4 + 2] * inv, e = o[db][4 * g4 + 3] * inv;
                u32x2 v;
                if constexpr (__is_same(OutT, __bf16)) v = u32x2{pack_bf16(a, b), pack_bf16(c2, e)};
                else v = u32x2{pack_f16(a, b), pack_f16(c2, e)};
                const int chunk = (d0 * 2) >> 4, half8 = (d0 * 2) & 8;
                *reinterpret_cast<FA_LDS u32x2*>(region + q * ROWB + (((chunk ^ q) & (CHUNKS - 1)) << 4) + half8) = v;
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): only this wave's own ds_writes have to land
        constexpr int ROWS_PER_INST = 64 / CHUNKS;                          // 4 (D=128) or 8 (D=64)
        const int rr = lane / CHUNKS, cc = lane % CHUNKS;
#pragma unroll
        for (int i = 0; i < 32 / ROWS_PER_INST; ++i) {
            const int r = i * ROWS_PER_INST + rr;
            const u32x4 v = *reinterpret_cast<FA_LDS const u32x4*>(region + r * ROWB + (((cc ^ r) & (CHUNKS - 1)) << 4));
            if (row0 + r < S) *reinterpret_cast<u32x4*>(Oh + (int64_t)(row0 + r) * oS_bytes + cc * 16) = v;
        }
    }
};

}  // namespace fa
