// computers16.hip.h -- per-wave compute of the fused forward pass on the 16x16x32 bf16 MFMA.
//
// Same algorithm and the same slot discipline as computers.hip.h (the counterpart of the reference's
// kernels/computers.cuh:5-69 and kernels/utils.cuh:17-45, :58-81, :93-113), on v_mfma_f32_16x16x32_bf16 instead of
// v_mfma_f32_32x32x16_bf16.  Why a second shape: on real (random) data MI355X is POWER limited, and the 16x16x32 form
// moves half the accumulator registers per FLOP -- MFMA-only microbenchmark on random bf16 operands, whole chip:
// 2353 TFLOP/s @ 2.27 GHz against 1854 @ 1.77 GHz for 32x32x16 (tests/micro/simd_mix, profiles/r02_simd_mix_m16.log);
// swapping only the instruction shape inside the production kernel (a timing-only experiment) was worth +8 %
// non-causal.  The price is issue slots: twice as many MFMA instructions, each holding the SIMD's vector issue 8 cycles.
//
// A wave owns 32 query rows = QG = 2 groups of 16; a KV tile is 64 keys = KG = 4 groups of 16.
//   S^T = K.Q^T   per (key group kg, query group qg): KS = D/32 chained MFMAs; A = K fragment (ds_read_b128 from the
//                 chunk-major K image: lane (key r = l&15, quarter h4 = l>>4) holds d = 32*ks + 8*h4 + j), B = Q fragment
//                 (registers, same (r, h4, j) map on the query's row).  Each K fragment serves both query groups.
//                 Accumulator (f32x4): query on the lane (col = l&15), key = 16*kg + 4*h4 + reg.
//   softmax       a row's 64 scores of a tile sit in the 16 registers of the FOUR lanes l, l^16, l^32, l^48: row max /
//                 row sum are in-lane ops plus one v_permlane16_swap + one v_permlane32_swap, needed once per unit on the
//                 optimistic pass (tile 0's max, the final sum) and once per tile on the tracked pass.
//   O^T += V^T.P^T per (d group dg of 16, query group qg): 2 chained MFMAs (32 keys each).  B = P^T: the accumulators of
//                 key groups 2*kk and 2*kk+1, rounded to bf16, ARE the B fragment of k-step kk (element j of quarter
//                 h4 = key 32*kk + 16*(j>>2) + 4*h4 + (j&3)); A = V^T fragment in that same key order: two
//                 ds_read_b64_tr_b16 from the [key/8][d/16][key%8][d%16] V image (each half-wave reads 256 contiguous
//                 bytes: conflict-free).  Each V^T fragment serves both query groups.  O: 4*D/16*2 = D/2 registers.
//   row sums      l += ONES.P^T: one more 16x16x32 MFMA per (k-step, query group) whose A operand is a constant register
//                 quad of bf16 ones -- every output row is the column sum of P^T over the k-step's 32 keys, i.e. the
//                 COMPLETE row sum (all four lane quarters) lands in every lane: no v_add_f32 per score element (the
//                 kernel is bound by the SIMD's shared instruction-issue port, where 32 adds cost 128 cycles per wave and
//                 tile and 4 MFMAs 32), no cross-lane reduction at the end, and the normaliser sums exactly the
//                 bf16-rounded weights the numerator uses (V = 1 gives O = 1 exactly).  The kernels that also return the
//                 LSE (C::SUM_MFMA = false) keep the fp32 sum of the UNROUNDED weights by v_add_f32 instead: the
//                 log-sum-exp is then exact to fp32 rounding (the bf16-rounded sum is off by up to 2^-9 relative).
//
// Slots: SA = 2*KG*KS (QK^T of tile t+1) + SB = 2*2*D/16 (P.V of tile t) MFMAs per tile, one per slot, fenced by
// sched_barrier(0).  Score element E (0..31) = (k-step kk = E/16, query group qg = (E/8)%2, j = E%8) goes to overall slot
// E*SPAN/32 with SPAN = SA + SB/2, which meets "P fragment (kk, qg) complete before its first P.V slot".
#pragma once

#include <type_traits>

#include "loaders.hip.h"

namespace fa {

struct Scores16 {        // raw scores of one 64-key tile: [16-key group][16-row query group]
    f32x4 s[4][2];
};

template <class C>
struct WaveCompute16 {
    static constexpr int D = C::D, ESZ = C::ESZ;
    static_assert(ESZ == 2, "16x16x32 path: bf16 inputs");
    static constexpr int KS = D / 32;              // 32-wide k-steps of QK^T
    static constexpr int KG = 4, QG = 2;           // 16-key groups per tile, 16-row query groups per wave
    static constexpr int DG = D / 16;              // 16-wide d groups of O^T
    static constexpr int NF = KG * KS;             // K fragments per tile
    static constexpr int NV = 2 * DG;              // V^T fragments per tile (2 k-steps of 32 keys)
    static constexpr int SA = NF * QG, SB = NV * QG;
    static constexpr int NPRE = C::NPRE < NF ? C::NPRE : NF;
    static constexpr int VPRE = C::VPRE;
    static constexpr int NE = 32;                  // score elements per lane per tile
    static constexpr int SPAN = SA + SB / 2;       // overall slots the exponentials are spread over
    using G = TileGeom<D, ESZ>;
    using Stage = std::conditional_t<C::MIX, MixStage<D, C::NWAVES, true, false>,
                  std::conditional_t<C::DMA, DmaStage<D, C::NWAVES, true, false>, BufStage<D, ESZ, C::NWAVES, C::PAD, true, C::P_F16>>>;
    // the P.V operand type: bf16, or fp16 with the fp16-weights option (weights rounded to 11 bits instead of 8; V staged as fp16).
    // F16W is a property of the PASS, not of the kernel: the fp16-weights kernels fall back to a bf16-weights tracked pass when
    // fp16 cannot hold the unit's V (|v| > 65504 becomes inf: kernel_bf16.hip.h, run_units), so every member that touches the
    // weights' or V's type takes it as a template argument that defaults to the kernel's C::P_F16
    template <bool F16W> using pv_of = std::conditional_t<F16W, f16x8, bf16x8>;
    using pv_t = pv_of<C::P_F16>;
    using ScoresT = Scores16;
    static constexpr int NL = Stage::NL, NW = Stage::NW;
    static constexpr int WSTEP = 2 * NW <= SB / 2 + 1 ? 2 : 1;   // LDS writes sit in every WSTEP-th slot of the second half of phase B
    static_assert(2 * NL <= SA && WSTEP * (NW - 1) < SB - SB / 2, "staging does not fit the slot plan");

    // ---- state that lives across tiles ----
    u32x4 qf[QG][KS];   // Q fragments
    f32x4 o[QG][DG];    // O^T accumulators: row = d, col = query
    float m[QG];        // reference max used for exponentiation (scaled, log2 domain)
    f32x4 lsum[QG];     // C::SUM_MFMA: row sums of the bf16-rounded weights, from the ONES.P^T MFMAs (all four registers equal)
    float l[QG];        // !C::SUM_MFMA (the LSE variants): fp32 sum of the UNROUNDED weights of this lane's quarter of the keys
    // ---- per-tile scratch ----
    u32x4 kf[NPRE];
    bf16x8 vf[VPRE + 1];
    uint32_t pw[QG][2][4];   // P(t) as packed bf16 pairs: [query group][k-step][word w = elements 2w, 2w+1]
    float mx_a[QG], mx_b[QG], p_even, sum_a[QG], sum_b[QG];
    bool need;
    unsigned long long t_mid = 0, t_end = 0;   // STAMP builds only

    __host__ __device__ static constexpr int elem_slot(int E) { return E * SPAN / NE; }
    // overall slot (0 .. SA + SB - 1) that issues staging load / LDS-DMA piece n of the tile two ahead: the odd slots from 1 on.
    // (Later is worse -- the pieces then land after the end-of-step wait: phase A's second half -2.6 %, phase B -9 ... -13 %,
    //  profiles/r03_tune_c_dma_slots_*.log.)
    __host__ __device__ static constexpr int load_slot(int n) { return 1 + 2 * n; }
    template <int SLOT, bool F16W = C::P_F16, int N = 0>
    __device__ __forceinline__ void load_in_slot(Stage& st, int t_load) {
        if constexpr (N < NL) {
            if constexpr (load_slot(N) == SLOT) {
                if constexpr (C::MIX) st.template load<N, F16W>(t_load);   // (MixStage: V by DMA or, fp16 units, through registers)
                else st.template load<N>(t_load);
            }
            load_in_slot<SLOT, F16W, N + 1>(st, t_load);
        }
    }

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
#pragma unroll
            for (int i = 0; i < DG; ++i) o[qg][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            m[qg] = -INFINITY;
            lsum[qg] = f32x4{0.f, 0.f, 0.f, 0.f};
            l[qg] = 0.f;
        }
    }
    // complete row sum of query group qg in every lane
    __device__ __forceinline__ float row_sum_total(int qg) const {
        if constexpr (C::SUM_MFMA) return lsum[qg][0];
        else return sum_all_quarters(l[qg]);
    }
    template <bool F16W = C::P_F16>
    __device__ __forceinline__ static pv_of<F16W> ones_frag() {
        constexpr uint32_t one2 = F16W ? 0x3c003c00u : 0x3f803f80u;
        u32x4 v = {one2, one2, one2, one2};
        return __builtin_bit_cast(pv_of<F16W>, v);
    }
    template <bool F16W = C::P_F16>
    __device__ __forceinline__ static uint32_t pack_p(float lo, float hi) { return F16W ? pack_f16(lo, hi) : pack_bf16(lo, hi); }

    // Q fragment (qg, ks) of row q = row0 + 16*qg + (lane&15): 16 bytes at byte 64*ks + 16*h4 of the row.
    __device__ __forceinline__ void load_q(const char* Qh, int64_t qS_bytes, int row0, int S, int lane, int row_bytes = D * ESZ) {
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            int row = row0 + 16 * qg + (lane & 15);
            row = row < S ? row : S - 1;
            const char* src = Qh + row * qS_bytes + (lane >> 4) * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (C::PAD) {
                    qf[qg][ks] = u32x4{0u, 0u, 0u, 0u};
                    if (ks * 64 + (lane >> 4) * 16 < row_bytes) qf[qg][ks] = *reinterpret_cast<const u32x4*>(src + ks * 64);
                } else {
                    qf[qg][ks] = *reinterpret_cast<const u32x4*>(src + ks * 64);
                }
            }
        }
    }
    // Coalesced form (Opt::coalesced_q): instruction i fetches 64/QCH WHOLE rows; the fragments are formed by one trip
    // through this wave's private LDS region (chunk c of row q parked at chunk c ^ (q & (QCH-1))).
    static constexpr int QCH = (D * ESZ) / 16;     // 16-byte chunks per Q row
    static constexpr int QRPI = 64 / QCH;          // rows fetched per instruction
    static constexpr int QLD = 32 / QRPI;          // loads per lane
    static_assert(QLD == QG * KS, "coalesced Q: as many loads as fragments");
    __device__ __forceinline__ void load_q_rows(const char* Qh, int64_t qS_bytes, int row0, int S, int lane) {
#pragma unroll
        for (int i = 0; i < QLD; ++i) {
            int row = row0 + i * QRPI + lane / QCH;
            row = row < S ? row : S - 1;
            qf[i / KS][i % KS] = *reinterpret_cast<const u32x4*>(Qh + row * qS_bytes + (lane % QCH) * 16);
        }
    }
    __device__ __forceinline__ void q_rows_to_fragments(lds_ptr region, int lane) {
        constexpr int ROWB = D * ESZ;
#pragma unroll
        for (int i = 0; i < QLD; ++i) {
            const int q = i * QRPI + lane / QCH, c = lane % QCH;
            lds_write_b128(region, q * ROWB + (((c ^ q) & (QCH - 1)) << 4), qf[i / KS][i % KS]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes only: LDS executes a wave's accesses in order
#pragma unroll
        for (int qg = 0; qg < QG; ++qg)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int q = 16 * qg + (lane & 15), c = 4 * ks + (lane >> 4);
                qf[qg][ks] = __builtin_bit_cast(u32x4, lds_read_b128(region, q * ROWB + (((c ^ q) & (QCH - 1)) << 4)));
            }
    }
    __device__ __forceinline__ void pin_q() {
#pragma unroll
        for (int qg = 0; qg < QG; ++qg)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[qg][ks]));
    }

    // K fragment f = (key group f / KS, k-step f % KS)
    __device__ __forceinline__ u32x4 k_read(lds_ptr kimg, int kbase, int f) const {
        if constexpr (Stage::K_DMA) return __builtin_bit_cast(u32x4, lds_read_b128(kimg, kbase + (f / KS) * (2 * Stage::KBLK) + (f % KS) * 512));
        else return __builtin_bit_cast(u32x4, lds_read_b128(kimg, kbase + (f % KS) * 4096 + (f / KS) * 256));
    }
    template <int F, int QGI>
    __device__ __forceinline__ void qk_mfma(const u32x4& kfrag, Scores16& n) const {
        n.s[F / KS][QGI] = mfma_16x16x32(__builtin_bit_cast(bf16x8, kfrag), __builtin_bit_cast(bf16x8, qf[QGI][F % KS]), n.s[F / KS][QGI]);
    }
    __device__ __forceinline__ static void zero(Scores16& n) {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg)
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) n.s[kg][qg] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // S^T = K.Q^T for the whole tile, compiler-scheduled: used once per pass for tile 0.
    template <int I = 0>
    __device__ __forceinline__ void qk_all(lds_ptr kimg, int kbase, Scores16& n) {
        if constexpr (I == 0) zero(n);
        if constexpr (I < SA) {
            constexpr int f = I / QG, qg = I % QG;
            if constexpr (qg == 0) kf[f % 2] = k_read(kimg, kbase, f);
            qk_mfma<f, qg>(kf[f % 2], n);
            qk_all<I + 1>(kimg, kbase, n);
        }
    }

    // Diagonal / ragged tile: key index > query index, or key index >= S  ->  -inf.
    // s[kg][qg][reg] holds key kv0 + 16*kg + 4*h4 + reg of query q_row0 + 16*qg + (lane&15).
    __device__ __forceinline__ void mask(Scores16& n, int kv0, int q_row0, int S, int lane) const {
        const int k0 = kv0 + 4 * (lane >> 4);
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            const int qi = q_row0 + 16 * qg + (lane & 15);
            const int lim = C::CAUSAL ? (qi < S - 1 ? qi : S - 1) : S - 1;
#pragma unroll
            for (int kg = 0; kg < KG; ++kg)
#pragma unroll
                for (int r = 0; r < 4; ++r) n.s[kg][qg][r] = (k0 + 16 * kg + r) > lim ? -INFINITY : n.s[kg][qg][r];
        }
    }
    __device__ __forceinline__ float row_max(const Scores16& n, int qg) const {
        float a = fmaxf(fmaxf(n.s[0][qg][0], n.s[0][qg][1]), fmaxf(n.s[0][qg][2], n.s[0][qg][3]));
#pragma unroll
        for (int kg = 1; kg < KG; ++kg) a = fmaxf(a, fmaxf(fmaxf(n.s[kg][qg][0], n.s[kg][qg][1]), fmaxf(n.s[kg][qg][2], n.s[kg][qg][3])));
        return a;
    }
    // Tile 0 of a pass: m = its row max (m = -inf before; O and l are still 0).
    __device__ __forceinline__ void first_max(const Scores16& n, float c) {
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) m[qg] = fmaxf(m[qg], max_all_quarters(row_max(n, qg)) * c);
    }

    // ---- softmax slices ------------------------------------------------------------------------
    template <int E, bool F16W = C::P_F16>
    __device__ __forceinline__ void exp_elem(const Scores16& cur, float c) {
        constexpr int kk = E / 16, qg = (E / 8) % 2, j = E % 8, kg = 2 * kk + (j >> 2), reg = j & 3;
        const float p = fast_exp2(fmaf(cur.s[kg][qg][reg], c, -m[qg]));
        if constexpr (!C::SUM_MFMA) {
            if constexpr (j & 1) sum_b[qg] += p;
            else sum_a[qg] += p;
        }
        // (empty asm: pins the value in this slot -- hipcc otherwise sinks the arithmetic towards its first use, behind the MFMAs
        //  or, when the slot sequence holds a branch, into the block behind it)
        if constexpr (j & 1) {
            pw[qg][kk][j >> 1] = pack_p<F16W>(p_even, p);
            if constexpr (!C::SUM_MFMA) asm volatile("" : "+v"(sum_a[qg]), "+v"(sum_b[qg]));
        } else {
            p_even = p;
        }
    }
    template <int SLOT, bool F16W = C::P_F16, int E = 0>
    __device__ __forceinline__ void exp_slot(const Scores16& cur, float c) {
        if constexpr (E < NE) {
            if constexpr (elem_slot(E) == SLOT) exp_elem<E, F16W>(cur, c);
            exp_slot<SLOT, F16W, E + 1>(cur, c);
        }
    }
    template <bool F16W = C::P_F16>
    __device__ __forceinline__ pv_of<F16W> p_frag(int qg, int kk) const {
        u32x4 v = {pw[qg][kk][0], pw[qg][kk][1], pw[qg][kk][2], pw[qg][kk][3]};
        return __builtin_bit_cast(pv_of<F16W>, v);
    }
    // tracked pass: v_max3 chains over 64/SB values of S(t+1) in slot J (J < SB/2); 32 values in all
    template <int J>
    __device__ __forceinline__ void max3_slot(const Scores16& n) {
        constexpr int PER = 64 / SB;
#pragma unroll
        for (int k = 0; k < PER; k += 2) {
            const int E = J * PER + k, qg = E / 16, e = E % 16;      // per query group: 16 values = [kg][reg]
            const float x0 = n.s[e >> 2][qg][e & 3], x1 = n.s[(e + 1) >> 2][qg][(e + 1) & 3];
            if ((k >> 1) & 1) mx_b[qg] = fmaxf(fmaxf(mx_b[qg], x0), x1);
            else mx_a[qg] = fmaxf(fmaxf(mx_a[qg], x0), x1);
        }
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) asm volatile("" : "+v"(mx_a[qg]), "+v"(mx_b[qg]));
    }
    __device__ __forceinline__ void decide(float c) {
        bool any = false;
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            const float mx = max_all_quarters(fmaxf(mx_a[qg], mx_b[qg])) * c;
            any = any || (mx > m[qg] + (float)C::THR);
            mx_a[qg] = mx;   // keep the scaled row max for the rescale body
        }
        need = __any(any);
    }

    // V^T A-fragment v = (k-step kk = v / DG, d group dg = v % DG): two transposed reads (16-key halves jj = 0, 1)
    __device__ __forceinline__ bf16x8 v_frag(lds_ptr vimg, int vbase, int kk, int dg) const {
        const s16x4 lo = lds_read_tr16_b64(vimg, vbase + (4 * kk) * (DG * 256) + dg * 256);
        const s16x4 hi = lds_read_tr16_b64(vimg, vbase + (4 * kk + 2) * (DG * 256) + dg * 256);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }

    // ---- the slots -----------------------------------------------------------------------------
    // phase A slot I: K fragment f = I / QG, query group I % QG
    template <int I, bool F16W = C::P_F16>
    __device__ __forceinline__ void slots_a(Stage& st, int t_load, lds_ptr k_next, lds_ptr v_cur, int kbase, int vbase,
                                            float c, const Scores16& cur, Scores16& nxt) {
        if constexpr (I < SA) {
            constexpr int f = I / QG, qg = I % QG;
            if constexpr (C::VALU_FIRST) exp_slot<I, F16W>(cur, c);
            qk_mfma<f, qg>(kf[f % NPRE], nxt);
            if constexpr (qg == QG - 1 && f + NPRE < NF) kf[f % NPRE] = k_read(k_next, kbase, f + NPRE);
            if constexpr (I >= SA - VPRE) {   // the last VPRE phase-A slots start the V^T window of phase B
                constexpr int v = I - (SA - VPRE);
                vf[v % (VPRE + 1)] = v_frag(v_cur, vbase, v / DG, v % DG);
            }
            load_in_slot<I, F16W>(st, t_load);
            if constexpr (!C::VALU_FIRST) exp_slot<I, F16W>(cur, c);
            __builtin_amdgcn_sched_barrier(0);
            slots_a<I + 1, F16W>(st, t_load, k_next, v_cur, kbase, vbase, c, cur, nxt);
        }
    }
    // phase B slot J: V^T fragment v = J / QG (k-step v / DG, d group v % DG), query group J % QG
    template <bool TRACK, int J, bool F16W = C::P_F16>
    __device__ __forceinline__ void slots_b(Stage& st, lds_ptr wr_slot, lds_ptr v_cur, int vbase, float c,
                                            const Scores16& cur, const Scores16& nxt) {
        if constexpr (J < SB) {
            constexpr int v = J / QG, qg = J % QG, kk = v / DG, dg = v % DG;
            o[qg][dg] = mfma_16x16x32(__builtin_bit_cast(pv_of<F16W>, vf[v % (VPRE + 1)]), p_frag<F16W>(qg, kk), o[qg][dg]);
            if constexpr (C::SUM_MFMA && dg == 1) lsum[qg] = mfma_16x16x32(ones_frag<F16W>(), p_frag<F16W>(qg, kk), lsum[qg]);   // row sums of this k-step
            if constexpr (qg == QG - 1 && v + VPRE < NV) {
                constexpr int vn = v + VPRE;
                vf[vn % (VPRE + 1)] = v_frag(v_cur, vbase, vn / DG, vn % DG);
            }
            exp_slot<SA + J, F16W>(cur, c);
            if constexpr (TRACK && J < SB / 2) max3_slot<J>(nxt);
            if constexpr (TRACK && J == SB / 2) decide(c);
            if constexpr (J >= SB / 2 && (J - SB / 2) % WSTEP == 0 && (J - SB / 2) / WSTEP < NW)
                st.template write<(J - SB / 2) / WSTEP, F16W>(wr_slot);
            __builtin_amdgcn_sched_barrier(0);
            slots_b<TRACK, J + 1, F16W>(st, wr_slot, v_cur, vbase, c, cur, nxt);
        }
    }

    // One tile: cur = S(t) (consumed), nxt = S(t+1) (produced).  Same contract as WaveCompute::tile_step.
    template <bool TRACK, bool F16W = C::P_F16>
    __device__ __forceinline__ void tile_step(Stage& st, int t_load, lds_ptr wr_slot, lds_ptr k_next, lds_ptr v_cur,
                                              int kbase, int vbase, float c, const Scores16& cur, Scores16& nxt,
                                              bool has_next, bool mask_next, int kv0_next, int q_row0, int S, int lane) {
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            mx_a[qg] = mx_b[qg] = -INFINITY;
            sum_a[qg] = sum_b[qg] = 0.f;
        }
        zero(nxt);
        st.set_dst(wr_slot);   // (LDS-DMA staging: where this iteration's loads land)
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
        if constexpr (!C::SUM_MFMA) {
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) l[qg] += sum_a[qg] + sum_b[qg];
        }
        // ONE rescale site (two sites that both multiply O make hipcc copy all accumulator registers per tile)
        if (has_next && mask_next) {
            mask(nxt, kv0_next, q_row0, S, lane);
            if constexpr (TRACK) {
#pragma unroll
                for (int qg = 0; qg < QG; ++qg) { mx_a[qg] = row_max(nxt, qg); mx_b[qg] = mx_a[qg]; }
                decide(c);
            }
        }
        if constexpr (TRACK) {
            if (has_next && need) {
#pragma unroll
                for (int qg = 0; qg < QG; ++qg) {
                    const float mn = fmaxf(m[qg], mx_a[qg]);
                    const float alpha = fast_exp2(m[qg] - mn);
                    m[qg] = mn;
                    lsum[qg] *= alpha;
                    l[qg] *= alpha;
#pragma unroll
                    for (int i = 0; i < DG; ++i) o[qg][i] *= alpha;
                }
            }
        }
    }

    // True iff a row sum or any O accumulator of this lane is inf / NaN (x*0 is NaN for both); four independent chains.
    __device__ __forceinline__ bool not_finite() const {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            acc[0] = fmaf(C::SUM_MFMA ? lsum[qg][0] : l[qg], 0.f, acc[0]);
#pragma unroll
            for (int i = 0; i < DG; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = fmaf(o[qg][i][k], 0.f, acc[k]);
        }
        const float a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        return a != a;
    }

    // ---- epilogues -----------------------------------------------------------------------------
    // ln sum_k exp(scale*s_k) = (m + log2 l) * ln 2
    __device__ __forceinline__ void store_lse(float* lse_head, float l_tot, int qg, int row0, int S, int lane) const {
        const int qi = row0 + 16 * qg + (lane & 15);
        if (lse_head && lane < 16 && qi < S) lse_head[qi] = (m[qg] + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
    }

    // 2-byte outputs through this wave's private LDS region, one query group (16 rows) at a time: row-major [16 rows][D], 16-byte
    // chunk c of row q at chunk c ^ (q & mask); whole rows back out with 16-byte stores.  region: 16*D*2 bytes, not aliased by
    // anything live.  (The LDS executes a wave's accesses in order, so the second group's writes cannot overtake the first
    // group's reads.)
    template <typename OutT>
    __device__ __forceinline__ void store_o_lds(lds_ptr region, char* Oh, float* lse_head, int64_t oS_bytes, int row0, int S,
                                                int lane, int orow_bytes = D * 2) {
        static_assert(sizeof(OutT) == 2, "LDS epilogue is for bf16 / f16 outputs");
        constexpr int ROWB = D * 2, CHUNKS = ROWB / 16;
        constexpr int ROWS_PER_INST = 64 / CHUNKS;                          // 4 (D=128) or 8 (D=64)
        const int h4 = lane >> 4, q = lane & 15;
        const int rr = lane / CHUNKS, cc = lane % CHUNKS;
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            const float l_tot = row_sum_total(qg);
            store_lse(lse_head, l_tot, qg, row0, S, lane);
            const float inv = 1.0f / l_tot;
#pragma unroll
            for (int dg = 0; dg < DG; ++dg) {
                const float a = o[qg][dg][0] * inv, b = o[qg][dg][1] * inv, c2 = o[qg][dg][2] * inv, e = o[qg][dg][3] * inv;
                u32x2 v;
                if constexpr (__is_same(OutT, __bf16)) v = u32x2{pack_bf16(a, b), pack_bf16(c2, e)};
                else v = u32x2{pack_f16(a, b), pack_f16(c2, e)};
                const int chunk = 2 * dg + (h4 >> 1), half8 = (h4 & 1) * 8;          // d0 = 16*dg + 4*h4 -> byte 2*d0
                *reinterpret_cast<FA_LDS u32x2*>(region + q * ROWB + (((chunk ^ q) & (CHUNKS - 1)) << 4) + half8) = v;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): only this wave's own ds_writes have to land
            if (!C::PAD && row0 + 32 <= S) {   // (wave-uniform) every row of the wave exists
                u32x4 v[16 / ROWS_PER_INST];
#pragma unroll
                for (int i = 0; i < 16 / ROWS_PER_INST; ++i) {
                    const int row = i * ROWS_PER_INST + rr;
                    v[i] = *reinterpret_cast<FA_LDS const u32x4*>(region + row * ROWB + (((cc ^ row) & (CHUNKS - 1)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 16 / ROWS_PER_INST; ++i)
                    store_global_b128<C::O_CACHE>(Oh + (int64_t)(row0 + 16 * qg + i * ROWS_PER_INST + rr) * oS_bytes + cc * 16, v[i]);
                continue;
            }
#pragma unroll
            for (int i = 0; i < 16 / ROWS_PER_INST; ++i) {
                const int row = i * ROWS_PER_INST + rr, grow = row0 + 16 * qg + row;
                const u32x4 v = *reinterpret_cast<FA_LDS const u32x4*>(region + row * ROWB + (((cc ^ row) & (CHUNKS - 1)) << 4));
                if (grow < S && (!C::PAD || cc * 16 < orow_bytes))
                    *reinterpret_cast<u32x4*>(Oh + (int64_t)grow * oS_bytes + cc * 16) = v;
            }
        }
    }
    // 4-byte outputs through LDS, one query group and 64 columns at a time ([16 rows][64 floats] pieces, 16-byte chunk c of row
    // q at chunk c ^ q); 256 contiguous bytes of a row per 16 lanes back out.  region: 16*256 bytes.
    template <typename OutT>
    __device__ __forceinline__ void store_o_lds32(lds_ptr region, char* Oh, float* lse_head, int64_t oS_bytes, int row0, int S,
                                                  int lane, int orow_bytes = D * 4) {
        static_assert(sizeof(OutT) == 4, "for fp32 outputs");
        const int h4 = lane >> 4, q = lane & 15;
        const int rr = lane >> 4, cc = lane & 15;
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            const float l_tot = row_sum_total(qg);
            store_lse(lse_head, l_tot, qg, row0, S, lane);
            const float inv = 1.0f / l_tot;
#pragma unroll
            for (int hf = 0; hf < D / 64; ++hf) {
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    const int dg = 4 * hf + dd, cidx = 4 * dd + h4;
                    const f32x4 v = {o[qg][dg][0] * inv, o[qg][dg][1] * inv, o[qg][dg][2] * inv, o[qg][dg][3] * inv};
                    *reinterpret_cast<FA_LDS f32x4*>(region + q * 256 + (((cidx ^ q) & 15) << 4)) = v;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                if (!C::PAD && row0 + 32 <= S) {   // (wave-uniform) every row of the wave exists
                    f32x4 v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<FA_LDS const f32x4*>(region + (4 * i + rr) * 256 + (((cc ^ (4 * i + rr)) & 15) << 4));
#pragma unroll
                    for (int i = 0; i < 4; ++i) store_global_b128<C::O_CACHE>(Oh + (int64_t)(row0 + 16 * qg + 4 * i + rr) * oS_bytes + hf * 256 + cc * 16, __builtin_bit_cast(u32x4, v[i]));
                    continue;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 4 * i + rr, grow = row0 + 16 * qg + row;
                    const f32x4 v = *reinterpret_cast<FA_LDS const f32x4*>(region + row * 256 + (((cc ^ row) & 15) << 4));
                    if (grow < S && (!C::PAD || hf * 256 + cc * 16 < orow_bytes))
                        *reinterpret_cast<f32x4*>(Oh + (int64_t)grow * oS_bytes + hf * 256 + cc * 16) = v;
                }
            }
        }
    }
};

}  // namespace fa
