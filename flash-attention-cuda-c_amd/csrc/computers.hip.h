// computers.hip.h -- per-wave compute of the fused forward pass: QK^T (MFMA) -> online softmax
// (in registers) -> PV (MFMA), cut into hand-placed slots.
//
// Counterpart of the reference's kernels/computers.cuh:5-69 (twoLoaderMhaComputeWarp) and the
// device helpers it calls, kernels/utils.cuh:17-45 (computeTileScore), :58-81
// (updateSoftmaxState), :93-113 (multiplyVAccumulateO).  Same algorithm -- blocked scores, running
// (max, sum) per query row, rescaled accumulation of P.V -- re-designed for CDNA4:
//
//   * one WAVE owns 32 query rows (the reference: one warp per row; R = 1 row group below); a KV tile is 64 keys;
//   * scores come from v_mfma_f32_32x32x16_{bf16, fp8_fp8} in the SWAPPED orientation S^T = K.Q^T, so
//     the 32 scores of one query row and key half sit in ONE lane's registers: row max / row sum are
//     in-lane ops plus one v_permlane32_swap (the reference: cg::reduce + Bc-1 shuffles, utils.cuh:66-73);
//   * FA-2 recurrence: O and l stay un-normalised, one divide at the end (the reference renormalises
//     every tile, utils.cuh:75-80); exp is exp2 with scale*log2(e) folded into one FMA;
//   * P^T (accumulator layout: key in the register index, query on the lane), rounded to bf16, IS the
//     B operand of O^T += V^T.P^T; V^T fragments come from ds_read_b64_tr_b16.  O lives in 16*D/32
//     accumulator registers per row group (the reference keeps O in shared memory, utils.cuh:107-111).
//
// Schedule.  hipcc, left alone (or nudged with sched_group_barrier), hoists the whole softmax in front
// of the MFMAs and makes each QK^T MFMA wait on an LDS read issued just before it.  A tile is therefore
// cut into (NA + NB)*R slots, one MFMA each, fenced by __builtin_amdgcn_sched_barrier(0); iteration t:
//
//   A_i (NA*R, NA = 2*D/16): MFMA  S_r(t+1) += Kfrag . Qfrag_r     (QK^T of the NEXT tile, K(t+1) in LDS;
//                                  each K fragment serves the R row groups back to back)
//                            read  K fragment NPRE ahead (later: the first V^T fragments)
//                            [buffer_load #n of tile t+2 at i = 1+2n]
//                            VALU  exp2 / row sum / bf16 pack of this slot's share of P(t)
//   B_j (NB*R, NB = 4*D/32): MFMA  O_r^T += V^T(t)frag . P_r(t)frag  (each V^T fragment serves R groups)
//                            read  V^T fragment VPRE ahead (2 x ds_read_b64_tr_b16)
//                            VALU  rest of P(t); tracked pass: v_max3 of S(t+1) in the first half, then the decision
//                            [ds_write_b128 #n of tile t+2 at j = NB*R/2 + 2n]
//
// The 32*R score elements of a lane are ordered (key group g, row group r, j) and element E goes to
// overall slot E*SPAN/(32R), SPAN = (NA + 3*NB/4)*R, which meets "P group (g, r) complete before its PV
// slot" with <= 1.25 elements per 32-cycle MFMA gap at d = 128.  Empty asm statements pin the partial
// sums / maxima in their slot (hipcc otherwise sinks them behind the MFMAs).
#pragma once

#include "loaders.hip.h"

namespace fa {

template <int R>
struct Scores {          // raw scores of one 64-key tile: [row group][32-key half]
    f32x16 s[R][2];
};

template <class C>
struct WaveCompute {
    static constexpr int D = C::D, ESZ = C::ESZ, R = 1;   // R: 32-row query groups per wave
    static constexpr int KS = D / 16;              // MFMA k-steps of one 32-key half of QK^T
    static constexpr int DB = D / 32;              // 32-wide d blocks of O^T
    static constexpr int NA = 2 * KS, NB = 4 * DB; // MFMAs per row group in phase A / phase B
    static constexpr int SA = NA * R, SB = NB * R; // slots of phase A / phase B
    static constexpr int MPF = ESZ == 1 ? 2 : 1;   // MFMAs (per row group) fed by one 16-byte K / Q fragment
    static constexpr int NF = NA / MPF;            // K fragment reads per tile
    static constexpr int FPH = NF / 2;             // ... per 32-key half = Q fragments per row group
    static constexpr int NPRE = C::NPRE < NF ? C::NPRE : NF;
    static constexpr int VPRE = C::VPRE;
    static constexpr int NE = 32 * R;              // score elements per lane per tile
    static constexpr int SPAN = SA + (3 * SB) / 4; // overall slots the exponentials are spread over
    using G = TileGeom<D, ESZ>;
    using Stage = std::conditional_t<C::MIX, MixStage<D, C::NWAVES>,
                  std::conditional_t<C::DMA, DmaStage<D, C::NWAVES, false, true>,
                                     std::conditional_t<C::DMA_K8, HybridStageFp8<D, C::NWAVES>, BufStage<D, ESZ, C::NWAVES, C::PAD>>>>;
    using ScoresT = Scores<R>;
    static constexpr int NL = Stage::NL, NW = Stage::NW;
    static constexpr int WSTEP = 2 * NW <= SB / 2 + 1 ? 2 : 1;   // LDS writes sit in every WSTEP-th slot of the second half of phase B
    static_assert(2 * NL <= SA && WSTEP * (NW - 1) < SB - SB / 2, "staging does not fit the slot plan");

    // ---- state that lives across tiles ----
    u32x4 qf[R][FPH];  // Q fragments (16 bytes each: one bf16 MFMA operand, or two fp8 operands)
    f32x16 o[R][DB];   // O^T accumulators: row = d, col = query
    float m[R];        // reference max used for exponentiation (scaled, log2 domain)
    float l[R];        // partial row sum (this lane's key half)
    // ---- per-tile scratch ----
    u32x4 kf[NPRE];    // K fragment window
    bf16x8 vf[VPRE + 1];
    uint32_t pw[R][16];   // P(t) as packed bf16 pairs: word 4g+w = elements 8g+2w, 8g+2w+1
    float sum_a[R], sum_b[R], mx_a[R], mx_b[R], p_even, p_last;
    bool need;         // tracked pass: lazy-rescale decision for S(t+1)
    unsigned long long t_mid = 0, t_end = 0;   // STAMP builds only

    __host__ __device__ static constexpr int elem_slot(int E) { return E * SPAN / NE; }
    // overall slot (0 .. SA + SB - 1) that issues staging load / LDS-DMA piece n of the tile two ahead: the odd slots from 1 on.
    // (Later is worse -- the pieces then land after the end-of-step wait: phase A's second half -2.6 %, phase B -9 ... -13 %,
    //  profiles/r03_tune_c_dma_slots_*.log.)
    __host__ __device__ static constexpr int load_slot(int n) { return 1 + 2 * n; }
    // F16W (mixed-precision kernels, C::MIX, only): the unit runs with fp16 softmax weights -- P rounded to fp16, V staged as fp16 through
    // registers (MixStage), P.V on v_mfma_f32_32x32x16_f16.  A property of the pass, handed down as a template argument.
    template <int SLOT, bool F16W = false, int N = 0>
    __device__ __forceinline__ void load_in_slot(Stage& st, int t_load) {
        if constexpr (N < NL) {
            if constexpr (load_slot(N) == SLOT) {
                if constexpr (C::MIX) st.template load<N, F16W>(t_load);
                else st.template load<N>(t_load);
            }
            load_in_slot<SLOT, F16W, N + 1>(st, t_load);
        }
    }

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int k = 0; k < 16; ++k) o[r][i][k] = 0.f;
            m[r] = -INFINITY;
            l[r] = 0.f;
        }
    }

    // Q fragment u of row q: 16 bytes at byte 32u + 16h of the row.  bf16: d = 16u + 8h .. +7 (k-step u).
    // fp8: d = 32u + 16h .. +15 -- the contraction order is permuted the same way for K (chunk 2u+h of
    // the K image), so one 16-byte fragment feeds two MFMAs.
    // row_bytes < D*ESZ (C::PAD): fragments past the end of the row are zero and are never read from memory.
    __device__ __forceinline__ void load_q(const char* Qh, int64_t qS_bytes, int row0, int S, int lane, int row_bytes = D * ESZ) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int row = row0 + 32 * r + (lane & 31);
            row = row < S ? row : S - 1;
            const char* src = Qh + row * qS_bytes + (lane >> 5) * 16;
#pragma unroll
            for (int u = 0; u < FPH; ++u) {
                if constexpr (C::PAD) {
                    qf[r][u] = u32x4{0u, 0u, 0u, 0u};
                    if (u * 32 + (lane >> 5) * 16 < row_bytes) qf[r][u] = *reinterpret_cast<const u32x4*>(src + u * 32);
                } else {
                    qf[r][u] = *reinterpret_cast<const u32x4*>(src + u * 32);
                }
            }
        }
    }
    // Coalesced form (Opt::coalesced_q).  load_q above has every lane read 16-byte pieces of its own row: one
    // instruction touches 32 rows x 2 pieces, 64 separate 16-byte requests.  Here instruction i fetches 64/CH WHOLE
    // rows (CH = 16-byte chunks per row; lane = (row, chunk)), and the fragments are formed by one trip through
    // this wave's private LDS region: chunk c of row q is parked at chunk c ^ (q & (CH-1)), so the 16 rows of a
    // ds_read_b128 lane group land on different banks.  Same instruction count, a quarter of the memory requests.
    static constexpr int QCH = (D * ESZ) / 16;     // 16-byte chunks per Q row
    static constexpr int QRPI = 64 / QCH;          // rows fetched per instruction
    static_assert(32 / QRPI == FPH, "coalesced Q: as many loads as fragments");
    __device__ __forceinline__ void load_q_rows(const char* Qh, int64_t qS_bytes, int row0, int S, int lane) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < FPH; ++i) {
                int row = row0 + 32 * r + i * QRPI + lane / QCH;
                row = row < S ? row : S - 1;
                qf[r][i] = *reinterpret_cast<const u32x4*>(Qh + row * qS_bytes + (lane % QCH) * 16);
            }
    }
    // region: 32*R rows x D*ESZ bytes private to this wave, not aliased by anything live (kernel_bf16.hip.h)
    __device__ __forceinline__ void q_rows_to_fragments(lds_ptr region, int lane) {
        constexpr int ROWB = D * ESZ;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < FPH; ++i) {
                const int q = 32 * r + i * QRPI + lane / QCH, c = lane % QCH;
                lds_write_b128(region, q * ROWB + (((c ^ q) & (QCH - 1)) << 4), qf[r][i]);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes only: LDS executes a wave's accesses in order
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int u = 0; u < FPH; ++u) {
                const int q = 32 * r + (lane & 31), c = 2 * u + (lane >> 5);
                qf[r][u] = __builtin_bit_cast(u32x4, lds_read_b128(region, q * ROWB + (((c ^ q) & (QCH - 1)) << 4)));
            }
    }

    // Make the Q fragments look "consumed" so hipcc waits for their loads HERE and not with a
    // pessimistic vmcnt inside the main loop (where it would also drain the tile prefetch).
    __device__ __forceinline__ void pin_q() {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int u = 0; u < FPH; ++u) asm volatile("" : "+v"(qf[r][u]));
    }

    // Fragment f = (k-step u = f % FPH, 32-key half kt = f / FPH)
    __host__ __device__ static constexpr int frag_u(int f) { return f % FPH; }
    __host__ __device__ static constexpr int frag_kt(int f) { return f / FPH; }
    __device__ __forceinline__ u32x4 k_read(lds_ptr kimg, int kbase, int f) const {
        if constexpr (Stage::K_DMA) return __builtin_bit_cast(u32x4, lds_read_b128(kimg, kbase + frag_kt(f) * (4 * Stage::KBLK) + frag_u(f) * 256));
        else return __builtin_bit_cast(u32x4, lds_read_b128(kimg, kbase + frag_u(f) * 2048 + frag_kt(f) * 512));
    }
    // QK^T MFMA of fragment f, sub-step sub (fp8 only), row group r.  kprev = the fragment before kfrag (MX form only).
    template <int F, int SUB, int RG>
    __device__ __forceinline__ void qk_mfma(const u32x4& kprev, const u32x4& kfrag, Scores<R>& n) const {
        const u32x4& q = qf[RG][frag_u(F)];
        f32x16& acc = n.s[RG][frag_kt(F)];
        if constexpr (C::MXQK) {
            // fragments (F-1, F) = 64 contraction elements in ONE block-scaled MFMA, issued in the last slot of
            // the pair; the pair's other three slots carry only their softmax slice
            static_assert(ESZ == 1, "the MX form is for fp8 inputs");
            if constexpr ((F & 1) && SUB == MPF - 1) acc = mfma_32x32x64_fp8_unit_scale(kprev, kfrag, qf[RG][(F - 1) % FPH], q, acc);
        } else {
            static_assert(ESZ == 2, "the non-scaled form is for bf16 inputs");
            acc = mfma_32x32x16(__builtin_bit_cast(bf16x8, kfrag), __builtin_bit_cast(bf16x8, q), acc);
        }
    }
    __device__ __forceinline__ static void zero(Scores<R>& n) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < 16; ++k) { n.s[r][0][k] = 0.f; n.s[r][1][k] = 0.f; }
    }

    // S^T = K.Q^T for all row groups, compiler-scheduled: used once per pass for tile 0.
    template <int I = 0>
    __device__ __forceinline__ void qk_all(lds_ptr kimg, int kbase, Scores<R>& n) {
        if constexpr (I == 0) zero(n);
        if constexpr (I < SA) {
            constexpr int f = I / (MPF * R), sub = (I % (MPF * R)) / R, rg = I % R;
            if constexpr (I % (MPF * R) == 0) kf[f % 2] = k_read(kimg, kbase, f);
            qk_mfma<f, sub, rg>(kf[(f + 1) % 2], kf[f % 2], n);
            qk_all<I + 1>(kimg, kbase, n);
        }
    }

    // Diagonal / ragged tile: key index > query index, or key index >= S  ->  -inf.
    // s[r][0][k] holds key kv0 + acc_row(k,h), s[r][1][k] key kv0 + 32 + acc_row(k,h); query q_row0 + 32r + (lane&31).
    __device__ __forceinline__ void mask(Scores<R>& n, int kv0, int q_row0, int S, int lane) const {
        const int k0 = kv0 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int qi = q_row0 + 32 * r + (lane & 31);
            const int lim = C::CAUSAL ? (qi < S - 1 ? qi : S - 1) : S - 1;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                n.s[r][0][k] = (k0 + acc_row(k, 0)) > lim ? -INFINITY : n.s[r][0][k];
                n.s[r][1][k] = (k0 + 32 + acc_row(k, 0)) > lim ? -INFINITY : n.s[r][1][k];
            }
        }
    }
    __device__ __forceinline__ float row_max(const Scores<R>& n, int r) const {
        float a = fmaxf(n.s[r][0][0], n.s[r][0][1]), b = fmaxf(n.s[r][1][0], n.s[r][1][1]);
#pragma unroll
        for (int k = 2; k < 16; k += 2) {
            a = fmaxf(a, fmaxf(n.s[r][0][k], n.s[r][0][k + 1]));
            b = fmaxf(b, fmaxf(n.s[r][1][k], n.s[r][1][k + 1]));
        }
        return fmaxf(a, b);
    }
    // Tile 0 of a pass: m = its row max (m = -inf before; O and l are still 0).
    __device__ __forceinline__ void first_max(const Scores<R>& n, float c) {
#pragma unroll
        for (int r = 0; r < R; ++r) m[r] = fmaxf(m[r], max_both_halves(row_max(n, r)) * c);
    }

    // ---- softmax slices ------------------------------------------------------------------------
    // Element E (0..32R-1) = (key group g = E/(8R), row group r = (E/8)%R, j = E%8) -> score 8g+j of row group r.
    template <int E, bool F16W = false>
    __device__ __forceinline__ void exp_elem(const Scores<R>& cur, float c) {
        constexpr int g = E / (8 * R), r = (E / 8) % R, j = E % 8, e = 8 * g + j;
        const float x = cur.s[r][e >> 4][e & 15];
        if constexpr (C::LATE_ADD) {
            // the previous element's weight joins its sum here, in front of this element's exponential (same sums, same order)
            if constexpr (E > 0) {
                constexpr int rp = ((E - 1) / 8) % R, ep = 8 * ((E - 1) / (8 * R)) + (E - 1) % 8;
                if constexpr (ep & 1) sum_b[rp] += p_last;
                else sum_a[rp] += p_last;
                asm volatile("" : "+v"(sum_a[rp]), "+v"(sum_b[rp]));
            }
            const float p = fast_exp2(fmaf(x, c, -m[r]));
            p_last = p;
            if constexpr (e & 1) pw[r][e >> 1] = F16W ? pack_f16(p_even, p) : pack_bf16(p_even, p);
            else p_even = p;
        } else {
            const float p = fast_exp2(fmaf(x, c, -m[r]));
            if constexpr (e & 1) {
                sum_b[r] += p;
                pw[r][e >> 1] = F16W ? pack_f16(p_even, p) : pack_bf16(p_even, p);
                asm volatile("" : "+v"(sum_a[r]), "+v"(sum_b[r]));   // keep the adds in this slot (hipcc sinks them)
            } else {
                sum_a[r] += p;
                p_even = p;
            }
        }
    }
    template <int SLOT, bool F16W = false, int E = 0>
    __device__ __forceinline__ void exp_slot(const Scores<R>& cur, float c) {
        if constexpr (E < NE) {
            if constexpr (elem_slot(E) == SLOT) exp_elem<E, F16W>(cur, c);
            exp_slot<SLOT, F16W, E + 1>(cur, c);
        }
    }
    __device__ __forceinline__ bf16x8 p_frag(int r, int g) const {
        u32x4 v = {pw[r][4 * g], pw[r][4 * g + 1], pw[r][4 * g + 2], pw[r][4 * g + 3]};
        return __builtin_bit_cast(bf16x8, v);
    }
    // tracked pass: v_max3 chains over 64/NB values of S(t+1) in slot J (J < SB/2); 32R values in all
    template <int J>
    __device__ __forceinline__ void max3_slot(const Scores<R>& n) {
        constexpr int PER = 64 / NB;
#pragma unroll
        for (int k = 0; k < PER; k += 2) {
            const int E = J * PER + k, r = E / 32, e = E % 32;
            const float x0 = n.s[r][e >> 4][e & 15];
            const float x1 = n.s[r][(e + 1) >> 4][(e + 1) & 15];
            if ((k >> 1) & 1) mx_b[r] = fmaxf(fmaxf(mx_b[r], x0), x1);
            else mx_a[r] = fmaxf(fmaxf(mx_a[r], x0), x1);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" : "+v"(mx_a[r]), "+v"(mx_b[r]));
    }
    __device__ __forceinline__ void decide(float c) {
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float mx = max_both_halves(fmaxf(mx_a[r], mx_b[r])) * c;
            any = any || (mx > m[r] + (float)C::THR);
            mx_a[r] = mx;   // keep the scaled row max for the rescale body
        }
        need = __any(any);
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
    // phase A slot I: fragment f = I/(MPF*R), then (sub, row group) = ((I % (MPF*R)) / R, I % R)
    template <int I, bool F16W = false>
    __device__ __forceinline__ void slots_a(Stage& st, int t_load, lds_ptr k_next, lds_ptr v_cur, int kbase, int vbase,
                                            float c, const Scores<R>& cur, Scores<R>& nxt) {
        if constexpr (I < SA) {
            constexpr int f = I / (MPF * R), rem = I % (MPF * R), sub = rem / R, rg = rem % R;
            if constexpr (C::VALU_FIRST) exp_slot<I, F16W>(cur, c);   // softmax slice covers the fragment's LDS latency
            qk_mfma<f, sub, rg>(kf[(f + NPRE - 1) % NPRE], kf[f % NPRE], nxt);
            if constexpr (C::MXQK) {
                // both fragments of a pair stay live until the pair's MFMA: refill the two window entries after it
                if constexpr (rem == MPF * R - 1 && (f & 1)) {
                    if constexpr (f - 1 + NPRE < NF) kf[(f - 1) % NPRE] = k_read(k_next, kbase, f - 1 + NPRE);
                    if constexpr (f + NPRE < NF) kf[f % NPRE] = k_read(k_next, kbase, f + NPRE);
                }
            } else if constexpr (rem == MPF * R - 1 && f + NPRE < NF) {
                kf[f % NPRE] = k_read(k_next, kbase, f + NPRE);
            }
            if constexpr (I >= SA - VPRE) {   // the last VPRE phase-A slots start the V^T window of phase B
                constexpr int v = I - (SA - VPRE);
                vf[v % (VPRE + 1)] = v_frag(v_cur, vbase, v / DB, v % DB);
            }
            load_in_slot<I, F16W>(st, t_load);
            if constexpr (!C::VALU_FIRST) exp_slot<I, F16W>(cur, c);
            __builtin_amdgcn_sched_barrier(0);
            slots_a<I + 1, F16W>(st, t_load, k_next, v_cur, kbase, vbase, c, cur, nxt);
        }
    }
    // phase B slot J: V^T fragment v = J/R (16-key step v/DB, d block v%DB), row group J%R
    template <bool TRACK, int J, bool F16W = false>
    __device__ __forceinline__ void slots_b(Stage& st, lds_ptr wr_slot, lds_ptr v_cur, int vbase, float c,
                                            const Scores<R>& cur, const Scores<R>& nxt) {
        if constexpr (J < SB) {
            constexpr int v = J / R, rg = J % R, s4 = v / DB, db = v % DB;
            if constexpr (F16W) o[rg][db] = mfma_32x32x16(__builtin_bit_cast(f16x8, vf[v % (VPRE + 1)]), __builtin_bit_cast(f16x8, p_frag(rg, s4)), o[rg][db]);
            else o[rg][db] = mfma_32x32x16(vf[v % (VPRE + 1)], p_frag(rg, s4), o[rg][db]);
            if constexpr (rg == R - 1 && v + VPRE < NB) {
                constexpr int vn = v + VPRE;
                vf[vn % (VPRE + 1)] = v_frag(v_cur, vbase, vn / DB, vn % DB);
            }
            exp_slot<SA + J, F16W>(cur, c);
            if constexpr (TRACK && J < SB / 2) max3_slot<J>(nxt);
            if constexpr (TRACK && J == SB / 2) decide(c);
            if constexpr (J >= SB / 2 && (J - SB / 2) % WSTEP == 0 && (J - SB / 2) / WSTEP < NW) {
                if constexpr (C::MIX) st.template write<(J - SB / 2) / WSTEP, F16W>(wr_slot);
                else st.template write<(J - SB / 2) / WSTEP>(wr_slot);
            }
            __builtin_amdgcn_sched_barrier(0);
            slots_b<TRACK, J + 1, F16W>(st, wr_slot, v_cur, vbase, c, cur, nxt);
        }
    }

    // One tile: cur = S(t) (consumed), nxt = S(t+1) (produced; on the wave's last tile it is computed from a tile the wave
    // does not need and ignored: one hot code path).
    // TRACK = true: running row max with lazy rescale (always safe).  TRACK = false: the optimistic
    // pass -- m stays the row max of tile 0 and no max / decision / rescale is issued.
    template <bool TRACK, bool F16W = false>
    __device__ __forceinline__ void tile_step(Stage& st, int t_load, lds_ptr wr_slot, lds_ptr k_next, lds_ptr v_cur,
                                              int kbase, int vbase, float c, const Scores<R>& cur, Scores<R>& nxt,
                                              bool has_next, bool mask_next, int kv0_next, int q_row0, int S, int lane) {
        st.set_dst(wr_slot);   // (LDS-DMA staging: where this iteration's loads land)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            sum_a[r] = sum_b[r] = 0.f;
            mx_a[r] = mx_b[r] = -INFINITY;
        }
        zero(nxt);
#pragma unroll
        for (int i = 0; i < NPRE; ++i) kf[i] = k_read(k_next, kbase, i);
        if constexpr (C::PRIO_A) __builtin_amdgcn_s_setprio(1);
        __builtin_amdgcn_sched_barrier(0);
        slots_a<0, F16W>(st, t_load, k_next, v_cur, kbase, vbase, c, cur, nxt);
        if constexpr (C::PRIO_A) {
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (C::STAMP) t_mid = cycle_stamp();
        slots_b<TRACK, 0, F16W>(st, wr_slot, v_cur, vbase, c, cur, nxt);
        if constexpr (C::STAMP) t_end = cycle_stamp();
        if constexpr (C::LATE_ADD) {   // (the tile's last weight: element NE - 1 is an odd key of the last row group)
            static_assert(((NE - 1) % 8) & 1, "the last element feeds sum_b");
            sum_b[((NE - 1) / 8) % R] += p_last;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) l[r] += sum_a[r] + sum_b[r];
        // ONE rescale site: the masked (diagonal / ragged) tile only recomputes the scalar decision and
        // the row max.  (Two sites that both multiply O made hipcc copy all 64 accumulator registers
        // twice per tile on the common path.)
        if (has_next && mask_next) {
            mask(nxt, kv0_next, q_row0, S, lane);
            if constexpr (TRACK) {
#pragma unroll
                for (int r = 0; r < R; ++r) { mx_a[r] = row_max(nxt, r); mx_b[r] = mx_a[r]; }
                decide(c);
            }
        }
        if constexpr (TRACK) {
            if (has_next && need) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float mn = fmaxf(m[r], mx_a[r]);
                    const float alpha = fast_exp2(m[r] - mn);
                    m[r] = mn;
                    l[r] *= alpha;
#pragma unroll
                    for (int i = 0; i < DB; ++i)
#pragma unroll
                        for (int k = 0; k < 16; ++k) o[r][i][k] *= alpha;
                }
            }
        }
    }

    // True iff a row sum or any O accumulator of this lane is inf / NaN (x*0 is NaN for both).  Four independent
    // chains: one dependent chain of 65 fmas costs ~500 cycles per unit.
    __device__ __forceinline__ bool not_finite() const {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[0] = fmaf(l[r], 0.f, acc[0]);
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k & 3] = fmaf(o[r][i][k], 0.f, acc[k & 3]);
        }
        const float a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        return a != a;
    }

    // ---- epilogues -----------------------------------------------------------------------------
    // ln sum_k exp(scale*s_k) = (m + log2 l) * ln 2   (m is the reference max in the scaled log2 domain)
    __device__ __forceinline__ void store_lse(float* lse_head, float l_tot, int r, int row0, int S, int lane) const {
        const int qi = row0 + (lane & 31);
        if (lse_head && lane < 32 && qi < S) lse_head[qi] = (m[r] + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
    }

    // 2-byte outputs: O^T accumulators -> this wave's private LDS region as a row-major [32R rows][D] tile
    // -> whole rows back out with 16-byte stores (each 16- or 8-lane group writes one full row).  The
    // direct form issues 16 eight-byte stores per lane that touch 32 rows each: ~8k cycles per workgroup,
    // store-issue bound; this form ~3.3k.  16-byte chunk c of row q sits at chunk c ^ (q & mask), so the
    // ds_write_b64 of 16 lanes (16 rows, same column) spread over all banks.  `region` = 32*R*D*2 bytes
    // private to this wave; the caller guarantees the K/V ring is dead.
    template <typename OutT>
    __device__ __forceinline__ void store_o_lds(lds_ptr region, char* Oh, float* lse_head, int64_t oS_bytes, int row0, int S,
                                                int lane, int orow_bytes = D * 2) {
        static_assert(sizeof(OutT) == 2, "LDS epilogue is for bf16 / f16 outputs");
        constexpr int ROWB = D * 2, CHUNKS = ROWB / 16;
        const int q = lane & 31, h = lane >> 5;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float l_tot = sum_both_halves(l[r]);
            store_lse(lse_head, l_tot, r, row0 + 32 * r, S, lane);
            const float inv = 1.0f / l_tot;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d0 = 32 * db + 8 * g4 + 4 * h;                  // 4 consecutive d = 8 bytes
                    const float a = o[r][db][4 * g4 + 0] * inv, b = o[r][db][4 * g4 + 1] * inv;
                    const float c2 = o[r][db][4 * g4 + 2] * inv, e = o[r][db][4 * g4 + 3] * inv;
                    u32x2 v;
                    if constexpr (__is_same(OutT, __bf16)) v = u32x2{pack_bf16(a, b), pack_bf16(c2, e)};
                    else v = u32x2{pack_f16(a, b), pack_f16(c2, e)};
                    const int chunk = (d0 * 2) >> 4, half8 = (d0 * 2) & 8;
                    *reinterpret_cast<FA_LDS u32x2*>(region + (32 * r + q) * ROWB + (((chunk ^ q) & (CHUNKS - 1)) << 4) + half8) = v;
                }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): only this wave's own ds_writes have to land
        constexpr int ROWS_PER_INST = 64 / CHUNKS;                          // 4 (D=128) or 8 (D=64)
        const int rr = lane / CHUNKS, cc = lane % CHUNKS;
        if (!C::PAD && row0 + 32 * R <= S) {   // (wave-uniform) every row of the wave exists: reads in flight together, plain stores
            u32x4 v[32 * R / ROWS_PER_INST];
#pragma unroll
            for (int i = 0; i < 32 * R / ROWS_PER_INST; ++i) {
                const int row = i * ROWS_PER_INST + rr;
                v[i] = *reinterpret_cast<FA_LDS const u32x4*>(region + row * ROWB + (((cc ^ row) & (CHUNKS - 1)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 32 * R / ROWS_PER_INST; ++i)
                store_global_b128<C::O_CACHE>(Oh + (int64_t)(row0 + i * ROWS_PER_INST + rr) * oS_bytes + cc * 16, v[i]);
            return;
        }
#pragma unroll
        for (int i = 0; i < 32 * R / ROWS_PER_INST; ++i) {
            const int row = i * ROWS_PER_INST + rr;
            const u32x4 v = *reinterpret_cast<FA_LDS const u32x4*>(region + row * ROWB + (((cc ^ row) & (CHUNKS - 1)) << 4));
            if (row0 + row < S && (!C::PAD || cc * 16 < orow_bytes))
                *reinterpret_cast<u32x4*>(Oh + (int64_t)(row0 + row) * oS_bytes + cc * 16) = v;
        }
    }
    // 4-byte outputs through LDS, 64 columns at a time: O^T accumulators -> this wave's private region as a
    // row-major [32R rows][64 floats] half tile (ds_write_b128, 16-byte chunk c of row q at chunk c ^ (q & 15):
    // the 8 lanes of a write group hit 8 different chunk columns) -> 256 contiguous bytes of a row per 16 lanes
    // back out.  The direct form (store_o) writes 32-byte pieces of 32 different rows per instruction.
    // `region` = 32*R*256 bytes private to this wave; the caller guarantees the K/V ring is dead.
    template <typename OutT>
    __device__ __forceinline__ void store_o_lds32(lds_ptr region, char* Oh, float* lse_head, int64_t oS_bytes, int row0, int S,
                                                  int lane, int orow_bytes = D * 4) {
        static_assert(sizeof(OutT) == 4, "for fp32 outputs");
        const int q = lane & 31, h = lane >> 5;
        float inv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float l_tot = sum_both_halves(l[r]);
            store_lse(lse_head, l_tot, r, row0 + 32 * r, S, lane);
            inv[r] = 1.0f / l_tot;
        }
        const int rr = lane >> 4, cc = lane & 15;
#pragma unroll
        for (int hf = 0; hf < D / 64; ++hf) {
            if (hf > 0) __builtin_amdgcn_s_waitcnt(0xc07f);   // this wave's reads of the previous half are done
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int db = 2 * hf + dd, cidx = 8 * dd + 2 * g4 + h;
                        const f32x4 v = {o[r][db][4 * g4 + 0] * inv[r], o[r][db][4 * g4 + 1] * inv[r],
                                         o[r][db][4 * g4 + 2] * inv[r], o[r][db][4 * g4 + 3] * inv[r]};
                        *reinterpret_cast<FA_LDS f32x4*>(region + (32 * r + q) * 256 + (((cidx ^ q) & 15) << 4)) = v;
                    }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): only this wave's own ds_writes have to land
            if (!C::PAD && row0 + 32 * R <= S) {   // (wave-uniform) every row of the wave exists
                f32x4 v[8 * R];
#pragma unroll
                for (int i = 0; i < 8 * R; ++i) v[i] = *reinterpret_cast<FA_LDS const f32x4*>(region + (4 * i + rr) * 256 + (((cc ^ (4 * i + rr)) & 15) << 4));
#pragma unroll
                for (int i = 0; i < 8 * R; ++i) store_global_b128<C::O_CACHE>(Oh + (int64_t)(row0 + 4 * i + rr) * oS_bytes + hf * 256 + cc * 16, __builtin_bit_cast(u32x4, v[i]));
                continue;
            }
#pragma unroll
            for (int i = 0; i < 8 * R; ++i) {
                const int row = 4 * i + rr;
                const f32x4 v = *reinterpret_cast<FA_LDS const f32x4*>(region + row * 256 + (((cc ^ row) & 15) << 4));
                if (row0 + row < S && (!C::PAD || hf * 256 + cc * 16 < orow_bytes))
                    *reinterpret_cast<f32x4*>(Oh + (int64_t)(row0 + row) * oS_bytes + hf * 256 + cc * 16) = v;
            }
        }
    }
};

}  // namespace fa
