// kernel_bf16.hip.h -- the MFMA forward kernel (bf16 or fp8-e4m3fn inputs, d in {64,128}).
//
// Counterpart of the reference's kernel entry kernels/FlashAttention.cuh:59-84 and of the loop nests of
// kernels/computers.cuh:33-67 / kernels/loaders.cuh:132-156,177-201.  The reference commits a tile and
// immediately waits for it on every cuda::pipeline (no load/compute overlap).
//
// One workgroup = 8 waves (two per SIMD, 226-240 VGPRs in production) processes UNITS of 256 query rows of one (batch, head), a
// wave owning 32 rows.  The grid is persistent: one workgroup per CU walks a static, balanced list of units
// (work_unit) and requests the next unit's Q and KV tile 0 before it runs the current unit's epilogue.
// KV tiles of 64 keys live in a 3-slot LDS ring [K image | V image] (loaders.hip.h).  Iteration t of a wave reads
// K(t+1) and V(t) and stages tile t+2 (LDS-DMA: loaders.hip.h) into slot (t+2)%3 = slot (t-1)%3, last read in
// iteration t-1, which every wave left at the previous barrier: ONE barrier per tile (behind a vmcnt(0): the wave's own
// DMA pieces have landed).  The loop is unrolled x2 with ping-pong
// score registers so S(t+1) never has to be copied into S(t).  What happens inside a tile: computers.hip.h.
// The alternatives that were built, measured and rejected (64 rows per wave, one unit per workgroup, 4-slot ring, ping-pong
// phases, unit streaming, packed softmax arithmetic, ...) are recorded with their numbers in DESIGN.md section 4 and
// profiles/r01_tune_*, r02_tune_*; their code lives in the history (round-2 tree), not here.
//
// Optimistic max.  exp2 / bf16 / f32 accumulation have ~2^127 of headroom, so the first pass takes
// every exponential relative to the row max of tile 0 and issues no per-tile max, decision or rescale
// (-4.5 % time).  If a later score exceeds that reference by more than the headroom (or P.V overflows),
// l or O becomes inf/NaN; each lane tests that at the end of the pass, block_or makes it
// workgroup-uniform, and the whole block is recomputed by the tracked pass (running max, lazy rescale
// with threshold 2^THR), which is always safe.
#pragma once

#include "computers.hip.h"
#include "computers16.hip.h"

#include <type_traits>

namespace fa {

// What distinguishes the kernel instantiations of the library.  Everything else about the kernel (prefetch depths, slot
// order, ring depth, staging form, ...) is derived from the problem type in KernelCfg.
struct Opt {
    bool stamp = false;          // diagnostic build: s_memtime stamps around the segments (tests/fa_tune)
    bool pad = false;            // the tensors' head dimension is smaller than D: rows are zero-padded on the fly
    int m16 = -1;                // both products on v_mfma_f32_16x16x32_bf16 (computers16.hip.h) instead of 32x32x16: the chip holds
                                 // a higher clock on that shape (power).  -1: on for bf16 inputs
    int sum_mfma = -1;           // 16x16x32 engine: row sums from ONES.P^T MFMAs (sums the bf16-rounded weights) instead of one v_add_f32
                                 // per score.  -1: on (the library turns it off in the kernels that return the LSE)
    int waves = 8;               // waves per workgroup.  4: 128-row units, two workgroups per CU where the LDS allows (d = 64): the small causal
                                 // problems with one 256-row unit per CU or fewer (fwd_mfma_pair_kernel)
    bool p_f16 = false;          // 16x16x32 engine: weights rounded to fp16 (11 significant bits) instead of bf16 (8), V staged as fp16, P.V
                                 // on v_mfma_f32_16x16x32_f16 (needs |V| <= 65504): FA_FLAG_F16_WEIGHTS, and by default the query blocks whose rows see few keys
    bool mix = false;            // 32x32x16 engine, LDS-DMA kernels: the units of the query blocks qb < Params::hp run with fp16 softmax weights (V staged
                                 // as fp16 through registers: MixStage), the others with bf16 weights, in list order inside ONE walk: the causal default
};

template <int D_, bool CAUSAL_, typename OutT_, int ESZ_ = 2, Opt O = Opt{}>
struct KernelCfg {
    static constexpr int D = D_;
    static constexpr bool CAUSAL = CAUSAL_;
    using OutT = OutT_;
    static constexpr int ESZ = ESZ_;                 // bytes per Q/K/V element: 2 = bf16, 1 = fp8 e4m3fn
    static constexpr bool STAMP = O.stamp, PAD = O.pad;
    // Every wave runs phase A of a tile step (QK^T of the next tile: the K fragment reads, the tile's DMA pieces, most of the
    // exponentials) at s_setprio 1 and phase B (P.V) at 0: of the two waves of a SIMD the one still in phase A outranks the one ahead
    // of it.  +0.6 ... +1.3 % (causal and not, 21 / 15 interleaved rounds: profiles/r03_tune_j_phase_a_priority_*.log); static
    // priorities for one half of the waves measured nothing (r03_tune_e_*)
    static constexpr bool PRIO_A = true;
    // Cache policy of the output stores (utils.hip.h: store_global_b128): non-temporal under the causal mask -- O is written once and
    // never read, and every line it leaves in the XCD's L2 evicts K/V lines that the head's other query blocks are about to
    // re-read (causal, fp32 O: plain 1058, sc1 1073, nt 1076, sc0 sc1 1076 TFLOP/s; bf16 O +0.5 %; without the mask -0.3 %:
    // profiles/r03_tune_d_output_store_policy_*.log)
    static constexpr int O_CACHE = CAUSAL_ ? 2 : 0;
    // Q rows fetched whole and turned into fragments through LDS (q_rows_to_fragments): on at d = 128 (+0.8 %), off at d = 64
    // (the 37 us cfg1 loses 1.8 % to the extra LDS trip)
    static constexpr bool COALESCED_Q = D_ == 128 && !O.pad;
    static constexpr bool MXQK = ESZ_ == 1;          // fp8 inputs: QK^T on the block-scaled 32x32x64 MFMA with unit scales
    static constexpr bool M16 = (O.m16 < 0 ? true : O.m16 != 0) && ESZ_ == 2;
    static constexpr bool SUM_MFMA = M16 && (O.sum_mfma < 0 ? true : O.sum_mfma != 0);
    // phase-A slots issue their softmax slice before the MFMA: +2 % on the 32x32x16 engine (it covers the fragment's LDS
    // latency), -1 % on the 16x16x32 engine
    static constexpr bool VALU_FIRST = !M16;
    static constexpr bool P_F16 = M16 && O.p_f16;
    // 32x32x16 engine: a weight joins its row sum in the slot of the NEXT weight -- no v_exp_f32 -> v_add_f32 back to back (the transcendental's
    // result needs a wait state, which hipcc fills with an s_nop 0: ~40 per tile step).  Same sums in the same order: bitwise-equal output;
    // +0.6 ... +1.7 % on the bf16-weights kernel, +0.3 ... +0.4 % on the mixed kernel (profiles/r04_tune_p_late_add_*.log)
    static constexpr bool LATE_ADD = !M16;
    // K/V tiles global -> LDS by `buffer_load ... lds` (loaders.hip.h: DmaStage): no staging registers (-16 to -20 VGPRs), no
    // ds_write; +2.4 ... +4.5 % on both engines.  bf16, unpadded rows; padded / fp16-weights kernels convert or zero-fill between
    // the load and the LDS write and keep the register path.  The epilogue's LDS regions sit behind ring slot 0: the next
    // unit's tile 0 lands there while the epilogue runs
    static constexpr bool DMA = ESZ_ == 2 && !O.pad && !O.p_f16;
    // both weight precisions in one kernel, chosen per unit (run_units); the K/V ring, the K image, the engine and the epilogue are
    // those of the LDS-DMA kernel, only V's way into LDS and the P.V operand type differ per unit
    static constexpr bool MIX = O.mix && DMA;
    static_assert(!O.mix || (ESZ_ == 2 && !O.pad && !O.p_f16 && (O.m16 == 0 || O.sum_mfma == 0)),
                  "the mixed-precision kernels are the LDS-DMA kernels (16x16x32 engine: with fp32 row sums)");
    // fp8 inputs: K alone by LDS-DMA (V is widened to bf16 on its way into LDS and keeps the register path): HybridStageFp8
    static constexpr bool DMA_K8 = ESZ_ == 1 && D_ == 128 && !O.pad && O.waves == 8;
    static constexpr int NPRE = 4;                   // K fragments in flight ahead of their MFMA
    static constexpr int VPRE = 2;                   // V^T fragments in flight ahead of their MFMA
    static constexpr int THR = 8;                    // lazy-rescale threshold of the tracked pass, log2 units
    static constexpr int NWAVES = O.waves;           // waves per workgroup (8: two per SIMD; 4: one per SIMD and workgroup), 32 query rows each
    static_assert(NWAVES == 8 || NWAVES == 4, "8 or 4 waves");
    static constexpr int QBLK = 32 * NWAVES;         // query rows of a unit
    static constexpr int RING = 3;                   // LDS ring slots: tile t+2 is staged in iteration t
    static constexpr int RING_BYTES = RING * TileGeom<D_, ESZ_>::SLOT;
    // LDS-DMA staging: the epilogue regions (QBLK rows of D 2-byte outputs, or of 64 floats) sit behind ring slot 0
    static constexpr int EP_OFF = (DMA || DMA_K8) ? TileGeom<D_, ESZ_>::SLOT : 0;
    static constexpr int EP_NEED = EP_OFF + (sizeof(OutT_) == 2 ? QBLK * D_ * 2 : QBLK * 256);
    static constexpr int FLAG_OFF = EP_NEED > RING_BYTES ? EP_NEED : RING_BYTES;   // behind everything: one word per wave, "my result is not finite" (block_or)
    static constexpr int LDS_BYTES = FLAG_OFF + 64;
    static_assert(LDS_BYTES <= 163840, "160 KiB of LDS per CU");
};

// What the library launches.
// Which engine, by measurement on random data (the chip holds a higher clock on the 16x16x32 shape, which costs twice the MFMA issue
// slots): without the mask 16x16x32 wins (+0.9 ... +4.4 % over five boxes), under the causal mask 32x32x16 (16x16x32: -6.8 ... +1.3 %,
// mean -2.3 %).  What bounds either, and what recovering the seam's cycles does to the clock: DESIGN.md section 5.
template <int D, bool CAUSAL, typename OutT, int ESZ = 2, bool STAMP = false, bool PAD = false, bool LSE = false>
using ProdCfg = KernelCfg<D, CAUSAL, OutT, ESZ, Opt{.stamp = STAMP, .pad = PAD, .m16 = CAUSAL ? 0 : -1, .sum_mfma = LSE ? 0 : -1}>;

// The fp16-weights kernels (FA_FLAG_F16_WEIGHTS, and the early query blocks of the default precision): weights rounded to fp16, V
// staged as fp16, fp32 sum of the unrounded weights (so the LSE is exact too)
template <int D, bool CAUSAL, typename OutT>
using P16Cfg = KernelCfg<D, CAUSAL, OutT, 2, Opt{.sum_mfma = 0, .p_f16 = true}>;

// Workgroup-wide OR of a per-lane predicate through one LDS word per wave and ONE barrier.  (__syncthreads_or takes two barriers, 256
// bytes of static LDS and -- it linearises threadIdx.y / .z -- two registers that hipcc spills to scratch in the largest kernels.)
// A wave rewrites its word only a pass later, with at least two workgroup barriers in between: every wave has read by then.
template <int NWAVES>
__device__ __forceinline__ bool block_or(bool v, lds_ptr flags, int wave) {
    const uint32_t mine = __any(v) ? 1u : 0u;                                    // wave-uniform
    *reinterpret_cast<FA_LDS uint32_t*>(flags + 4 * wave) = mine;               // (all lanes: same address, same value)
    __syncthreads();
    u32x4 a = *reinterpret_cast<FA_LDS const u32x4*>(flags);
    uint32_t r = a[0] | a[1] | a[2] | a[3];
    if constexpr (NWAVES == 8) {
        const u32x4 b = *reinterpret_cast<FA_LDS const u32x4*>(flags + 16);
        r |= b[0] | b[1] | b[2] | b[3];
    }
    return __builtin_amdgcn_readfirstlane(r) != 0;
}

// The causal default: the bf16-weights kernel whose units of the query blocks qb < Params::hp (the rows that see fewer than FA_EARLY_KEYS
// keys) run with fp16 weights -- ONE walk over ONE (head, query block) list, every unit in the precision of its block
template <int D, typename OutT, bool STAMP = false>
using MixCfg = KernelCfg<D, true, OutT, 2, Opt{.stamp = STAMP, .m16 = 0, .mix = true}>;

// The per-wave compute engine of a configuration: 16x16x32 MFMAs (computers16.hip.h) or 32x32x16 (computers.hip.h).
template <class C>
using WaveComputeOf = std::conditional_t<C::M16, WaveCompute16<C>, WaveCompute<C>>;

// One pass over all KV tiles of the workgroup's query block.  Returns (workgroup-uniform) whether the result is not finite and
// has to be recomputed by a safer pass: the optimistic pass (TRACK = false) always reports, the tracked pass of an fp16-weights
// kernel reports too (F16W: its V is fp16, where a finite bf16 |v| > 65504 is inf -- run_units then repeats the unit with bf16
// weights and bf16 V, F16W = false, which is the last resort and reports nothing).
template <class C, bool TRACK, bool F16W = C::P_F16>
__device__ __forceinline__ bool attention_pass(const Params& p, WaveComputeOf<C>& w, typename WaveComputeOf<C>::Stage& st, lds_ptr smem,
                                               int n_tiles, int my_tiles, int q_row0, int lane,
                                               unsigned long long (&acc)[24], bool tile0_in_flight) {
    using G = TileGeom<C::D, C::ESZ>;
    constexpr bool CAUSAL = C::CAUSAL;
    constexpr int KVBLK = 64, SLOT = G::SLOT, KT = G::K_TILE;
    const int S = p.Sk;   // key bound of the masks
    unsigned long long tp0 = 0, tp1 = 0, tp2 = 0;
    if constexpr (C::STAMP) tp0 = cycle_stamp();
    w.init();
    using WC = WaveComputeOf<C>;
    constexpr int KBLK = (G::ROWB / 16) * 128;   // DMA form of the K image: bytes per 8-key block
    const int kbase = WC::Stage::K_DMA ? (C::M16 ? kd16_read_base(lane, KBLK) : kd_read_base(lane, KBLK)) : (C::M16 ? k16_read_base(lane) : k_read_base(lane));
    const int vbase = C::M16 ? v16_read_base<C::D>(lane) : v_read_base(lane);
    const float c = p.scale_log2;
    auto needs_mask = [&](int t) { return (CAUSAL && t * KVBLK + KVBLK - 1 > q_row0) || (t * KVBLK + KVBLK > S); };
    typename WC::ScoresT sA, sB;

    // Prologue: tile 0 (requested by the caller together with Q on the first pass) -> LDS, barrier; then
    // tile 1 is fetched while S(0) = K(0).Q^T and its row max are computed.
    // (MIX / P_F16: the form of the tile -- V as fp16 through registers or as bf16 -- follows the pass)
    if constexpr (C::MIX) {
        if (!tile0_in_flight) st.template load_all_into<F16W>(0, smem);
        st.template write_all<F16W>(smem);
    } else {
        if (!tile0_in_flight) st.load_all_into(0, smem);
        if constexpr (C::P_F16) st.template write_all<0, F16W>(smem);
        else st.write_all(smem);
    }
    unsigned long long tw0 = 0;
    if constexpr (C::STAMP) tw0 = cycle_stamp();
    st.wait_all();
    if constexpr (C::STAMP) acc[16] += cycle_stamp() - tw0;   // (vmcnt(0): tile 0's pieces AND the previous unit's output stores)
    __syncthreads();
    if constexpr (C::MIX) st.template load_all_into<F16W>(1, smem + SLOT);
    else st.load_all_into(1, smem + SLOT);     // past-the-end tiles read as zeros (buffer range check)
    constexpr int AHEAD = C::RING - 1;                 // iteration t stages tile t + AHEAD
    if constexpr (C::STAMP) tp1 = cycle_stamp();
    if (my_tiles > 0) {
        w.qk_all(smem, kbase, sA);
        if (needs_mask(0)) w.mask(sA, 0, q_row0, S, lane);
        w.first_max(sA, c);   // m = row max of tile 0 (the reference of the optimistic pass)
    }
    if constexpr (C::MIX) st.template write_all<F16W>(smem + SLOT);
    else if constexpr (C::P_F16) st.template write_all<0, F16W>(smem + SLOT);
    else st.write_all(smem + SLOT);
    st.wait_all();
    __syncthreads();
    if constexpr (C::STAMP) { tp2 = cycle_stamp(); acc[8] += tp1 - tp0; acc[9] += tp2 - tp1; }

    // ring slot byte offsets of tiles t, t+1, t+AHEAD
    int so_cur = 0, so_nxt = SLOT, so_wr = AHEAD * SLOT;
    // kind: 0 = full step (a next tile exists), 1 = the wave's last tile (its QK^T runs on a tile it does not need), 2 = staging
    // only (the wave is past its causal diagonal but still stages its share of the tiles the other waves need)
    auto step = [&](int t, int kind, typename WC::ScoresT& cur, typename WC::ScoresT& nxt) {
        unsigned long long t0 = 0, t4 = 0, t6 = 0;
        if constexpr (C::STAMP) t0 = cycle_stamp();
        if (kind != 2) {
            const bool has_next = kind == 0;
            if constexpr (C::P_F16 || C::MIX)
                w.template tile_step<TRACK, F16W>(st, t + AHEAD, smem + so_wr, smem + so_nxt, smem + so_cur + KT, kbase, vbase, c, cur, nxt,
                                                  has_next, has_next && needs_mask(t + 1), (t + 1) * KVBLK, q_row0, S, lane);
            else
                w.template tile_step<TRACK>(st, t + AHEAD, smem + so_wr, smem + so_nxt, smem + so_cur + KT, kbase, vbase, c, cur, nxt,
                                            has_next, has_next && needs_mask(t + 1), (t + 1) * KVBLK, q_row0, S, lane);
        } else {
            if constexpr (C::MIX) {
                st.template load_all_into<F16W>(t + AHEAD, smem + so_wr);
                st.template write_all<F16W>(smem + so_wr);
            } else {
                st.load_all_into(t + AHEAD, smem + so_wr);
                if constexpr (C::P_F16) st.template write_all<0, F16W>(smem + so_wr);
                else st.write_all(smem + so_wr);
            }
        }
        if constexpr (C::STAMP) t4 = cycle_stamp();
        st.wait_all();   // (LDS-DMA staging: this wave's pieces of tile t + AHEAD have landed)
        __syncthreads();
        if constexpr (C::STAMP) {
            t6 = cycle_stamp();
            if (kind != 2) { acc[1] += w.t_mid - t0; acc[2] += w.t_end - w.t_mid; acc[3] += t4 - w.t_end; acc[6] += 1; }
            acc[5] += t6 - t4;
        }
        const int tmp = so_cur;
        so_cur = so_nxt;
        so_nxt = so_wr;
        so_wr = tmp;
    };
    auto kind_of = [&](int t) { return t + 1 < my_tiles ? 0 : (t < my_tiles ? 1 : 2); };
    for (int t = 0; t < n_tiles; t += 2) {
        step(t, kind_of(t), sA, sB);
        if (t + 1 < n_tiles) step(t + 1, kind_of(t + 1), sB, sA);
    }
    if constexpr (TRACK && !F16W) return false;
    else {
        unsigned long long tc0 = 0;
        if constexpr (C::STAMP) tc0 = cycle_stamp();
        const bool bad = block_or<C::NWAVES>(my_tiles > 0 && w.not_finite(), smem + C::FLAG_OFF, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
        if constexpr (C::STAMP) acc[10] += cycle_stamp() - tc0;
        return bad;
    }
}

// Work assignment.  The grid is persistent, one workgroup per CU; XCD group x = bid & 7 owns the contiguous units
// [x*cpx, (x+1)*cpx) (all query blocks of a head re-read the same K/V: one XCD's L2), and its jpx workgroups walk them in rounds
// of jpx consecutive units -- at any moment the group works on a few adjacent heads, whose K/V stay in that XCD's L2.  Odd
// rounds run in reverse order ("snake"): under the causal mask a head's query blocks are listed heaviest first, so workgroup j
// gets cost c in one round and (max+1-c) in the next -- a static schedule whose per-workgroup totals are equal when nQ divides
// jpx.  With fewer units than CUs the grid is one workgroup per unit (a single round).  Pure speed choice: any placement is correct.
template <class C>
__device__ __forceinline__ bool work_unit(const UnitList& p, int round, int& g, int& qb) {
    const int bid = blockIdx.x;
    int idx = bid >> 3;
    idx = round * p.jpx + ((round & 1) ? p.jpx - 1 - idx : idx);
    if (idx >= p.cpx) return false;
    const int u = (bid & 7) * p.cpx + idx;
    if (u >= p.units) return false;
    g = u / p.nQ;
    qb = u - g * p.nQ;
    if (C::CAUSAL) qb = p.nQ - 1 - qb;  // heaviest query blocks of a head first
    qb += p.qb0;
    return true;
}

// Everything a wave needs to know about one unit = one 256-row query block of one (batch, head).
template <class C>
struct UnitCtx {
    const char *Qh, *Kh, *Vh;
    char* Oh;
    float* lse_head;
    int q_row0, n_tiles, my_tiles;
    bool wave_live;
    bool early;   // C::MIX: this unit runs with fp16 softmax weights (query block qb < Params::hp)
    __device__ __forceinline__ void set(const Params& p, int g, int qb, int wave) {
        early = C::MIX && qb < p.hp;
        constexpr int ESZ = C::ESZ, KVBLK = 64, QBLK = C::QBLK, WROWS = 32;
        const int b = g / p.H, h = g - b * p.H;
        Qh = (const char*)p.Q + (b * p.qB + h * p.qH) * ESZ;
        Kh = (const char*)p.K + (b * p.kB + h * p.kH) * ESZ;
        Vh = (const char*)p.V + (b * p.vB + h * p.vH) * ESZ;
        Oh = (char*)p.O + (b * p.oB + h * p.oH) * (int64_t)sizeof(typename C::OutT);
        lse_head = p.lse ? p.lse + (int64_t)g * p.S : nullptr;
        q_row0 = qb * QBLK + wave * WROWS;              // first query row of this wave
        const int q_end = min(p.S, (qb + 1) * QBLK);    // one past the last query row of the block
        const int k_tiles = (p.Sk + KVBLK - 1) / KVBLK;
        n_tiles = C::CAUSAL ? min(k_tiles, (q_end + KVBLK - 1) / KVBLK) : k_tiles;
        // tiles this wave computes: all (non-causal) or up to the diagonal of its last row (causal)
        wave_live = q_row0 < p.S;
        my_tiles = !wave_live ? 0 : (C::CAUSAL ? min(n_tiles, (q_row0 + WROWS - 1) / KVBLK + 1) : n_tiles);
    }
};

// Which units a walk takes: all of its list (the persistent kernels), or the single unit (head L.units, query block L.qb0) of a
// workgroup of fwd_mfma_pair_kernel.
enum class Kind { ALL, ONE };
template <class C, Kind KIND>
__device__ __forceinline__ bool next_unit(const UnitList& L, int& round, int& g, int& qb) {
    if constexpr (KIND == Kind::ALL) return work_unit<C>(L, round, g, qb);
    else {
        g = L.units;
        qb = L.qb0;
        return round == 0;
    }
}

// Everything a workgroup does with configuration C: walk its units of list L.
template <class C, Kind KIND = Kind::ALL>
__device__ __forceinline__ void run_units(const Params& p, const UnitList& L, lds_ptr smem) {
    constexpr int D = C::D, ESZ = C::ESZ;
    using OutT = typename C::OutT;

    int g, qb, round = 0;
    if (!next_unit<C, KIND>(L, round, g, qb)) return;
    unsigned long long t_kernel0 = 0, t_real0 = 0;
    if constexpr (C::STAMP) { t_real0 = realtime_stamp(); t_kernel0 = cycle_stamp(); }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = p.S, Sk = p.Sk;
    const int64_t qSb = p.qS * ESZ, kSb = p.kS * ESZ, vSb = p.vS * ESZ, oSb = p.oS * (int64_t)sizeof(OutT);
    constexpr int WROWS = 32;                       // query rows per wave

    UnitCtx<C> cur;
    cur.set(p, g, qb, wave);
    WaveComputeOf<C> w;
    typename WaveComputeOf<C>::Stage st;
    const int row_bytes = C::PAD ? p.d * ESZ : D * ESZ, orow_bytes = C::PAD ? p.d * (int)sizeof(OutT) : D * (int)sizeof(OutT);
    st.init(cur.Kh, cur.Vh, kSb, vSb, Sk, wave, lane, row_bytes);
    // tile 0 and Q travel together (one HBM round trip); C::MIX: in the form the unit's pass takes it (wave-uniform branch)
    if constexpr (C::MIX) {
        if (cur.early) st.template load_all_into<true>(0, smem);
        else st.template load_all_into<false>(0, smem);
    } else {
        st.load_all_into(0, smem);
    }
    if constexpr (C::COALESCED_Q) w.load_q_rows(cur.Qh, qSb, cur.q_row0, S, lane);
    else w.load_q(cur.Qh, qSb, cur.q_row0, S, lane, row_bytes);
    unsigned long long acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (C::STAMP) acc[12] = cycle_stamp() - t_kernel0;   // setup: unit decode, descriptors, first loads issued

    while (true) {
        unsigned long long t_q0 = 0;
        if constexpr (C::STAMP) t_q0 = cycle_stamp();
        w.pin_q();
        if constexpr (C::COALESCED_Q) {
            // staging regions sit behind ring slot 0 (tile 0 is about to be written there by other waves); every
            // wave finishes this round trip before the first barrier of the pass, after which slot 1 is written
            using G = TileGeom<D, ESZ>;
            static_assert(G::SLOT + C::QBLK * D * ESZ <= C::LDS_BYTES, "Q staging regions must fit behind slot 0");
            int lane_q = lane;   // keep the 16 staging addresses inside the unit loop (hoisted, they spill)
            asm volatile("" : "+v"(lane_q));
            w.q_rows_to_fragments(smem + G::SLOT + wave * (WROWS * D * ESZ), lane_q);
            w.pin_q();
        }
        if constexpr (C::STAMP) acc[7] += cycle_stamp() - t_q0;

        // Lane-derived values of the pass (read bases, the 32 mask compares of tile 0, ...) are unit-invariant: hipcc
        // hoists them out of the unit loop, runs out of SGPRs for the compare masks and spills across the tile loop.
        // An opaque copy of the lane id keeps them inside the pass.
        int lane_p = lane;
        asm volatile("" : "+v"(lane_p));
        if constexpr (C::MIX) {
            // both precisions in one walk: the unit's query block says which.  An fp16 unit whose passes come out non-finite (V beyond
            // fp16's range, see below) is repeated by the bf16-weights tracked pass -- the one a bf16 unit falls back to anyway
            if (cur.early) {
                if (attention_pass<C, false, true>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, true))
                    if (attention_pass<C, true, true>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, false))
                        attention_pass<C, true, false>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, false);
            } else {
                if (attention_pass<C, false, false>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, true))
                    attention_pass<C, true, false>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, false);
            }
        } else if constexpr (C::P_F16) {
            // fp16 weights need V in fp16: a finite bf16 |v| > 65504 is inf there (and 0 * inf = NaN poisons rows that do not even see
            // the key).  Both fp16 passes report a non-finite result; the last resort is the bf16-weights tracked pass, which holds
            // whatever bf16 holds (the reference's V is float: kernels/FlashAttention.cuh:60).  Data that never overflows pays nothing.
            if (attention_pass<C, false, true>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, true))
                if (attention_pass<C, true, true>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, false))
                    attention_pass<C, true, false>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, false);
        } else {
            if (attention_pass<C, false>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, true))
                attention_pass<C, true>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, false);
        }

        // The next unit's tile 0 and Q are requested ahead, so that their HBM round trip runs under this unit's epilogue
        // (every wave is past the last tile's barrier: ring slot 0 is free, the epilogue works behind it; the register-staged
        // kernels' staging registers and Q's are dead here).
        unsigned long long t_nx0 = 0;
        if constexpr (C::STAMP) t_nx0 = cycle_stamp();
        UnitCtx<C> nxt;
        ++round;
        const bool more = next_unit<C, KIND>(L, round, g, qb);
        if (more) {
            nxt.set(p, g, qb, wave);
            st.init(nxt.Kh, nxt.Vh, kSb, vSb, Sk, wave, lane, row_bytes);
            if constexpr (C::MIX) {
                if (nxt.early) st.template load_all_into<true>(0, smem);   // (V in registers across the epilogue)
                else st.template load_all_into<false>(0, smem);
            } else {
                st.load_all_into(0, smem);
            }
            // (opaque lane: a hoisted per-lane Q address is spilled across the tile loop, and its reload's vmcnt(0)
            // would make the Q loads wait for the tile-0 loads just issued)
            int lane_n = lane;
            asm volatile("" : "+v"(lane_n));
            if constexpr (C::COALESCED_Q) w.load_q_rows(nxt.Qh, qSb, nxt.q_row0, S, lane_n);
            else w.load_q(nxt.Qh, qSb, nxt.q_row0, S, lane_n, row_bytes);
            __builtin_amdgcn_sched_barrier(0);
        }

        unsigned long long t_ep0 = 0;
        if constexpr (C::STAMP) { t_ep0 = cycle_stamp(); acc[13] += t_ep0 - t_nx0; }   // next unit decode + prefetch issue
        // The epilogue's ~40 per-lane addresses must be recomputed here: hoisted out of the unit loop they
        // would live across the tile loop, spill, and their reload (vmcnt(0)) would wait for the prefetch above.
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        // every pass ends behind a workgroup barrier, so no wave still reads the K/V ring: the epilogue regions reuse it
        if constexpr (sizeof(OutT) == 2) {
            static_assert(C::EP_OFF + C::QBLK * D * 2 <= C::LDS_BYTES, "epilogue regions must fit the ring");
            if (cur.wave_live)
                w.template store_o_lds<OutT>(smem + C::EP_OFF + wave * (WROWS * D * 2), cur.Oh, cur.lse_head, oSb, cur.q_row0, S, lane_e, orow_bytes);
        } else {
            static_assert(C::EP_OFF + C::QBLK * 64 * 4 <= C::LDS_BYTES, "epilogue regions must fit the ring");
            if (cur.wave_live)
                w.template store_o_lds32<OutT>(smem + C::EP_OFF + wave * (WROWS * 256), cur.Oh, cur.lse_head, oSb, cur.q_row0, S, lane_e, orow_bytes);
        }
        if constexpr (C::STAMP) acc[4] += cycle_stamp() - t_ep0;   // epilogue: normalise + store O (issue side)
        if (!more) break;
        // register-staged kernels: the next prologue's ds_writes of tile 0 alias other waves' epilogue regions
        unsigned long long t_b0 = 0;
        if constexpr (C::STAMP) t_b0 = cycle_stamp();
        // (LDS-DMA kernels need no barrier here: a wave's Q staging region IS its epilogue region, tile 0 lands in slot 0 in front of
        //  them, and slot 1 is first written behind the next pass's first barrier -- nothing the next prologue writes before that
        //  barrier aliases another wave's epilogue region.  +1.4 % non-causal with fp32 output, profiles/r03_tune_a_*.log)
        if constexpr (!C::DMA) __syncthreads();
        if constexpr (C::STAMP) acc[15] += cycle_stamp() - t_b0;   // waiting for the slowest wave's epilogue
        cur = nxt;
    }
    if constexpr (C::STAMP) {
        const unsigned long long t_tail0 = cycle_stamp();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // include the store tail
        const unsigned long long t_end = cycle_stamp();
        acc[14] = t_end - t_tail0;
        acc[0] = t_end - t_kernel0;                        // whole workgroup lifetime of this wave
        const unsigned long long t_real1 = realtime_stamp();
        acc[17] = t_real1 - t_real0;                       // ... and in ticks of the constant 100 MHz counter (-> the core clock it ran at),
        acc[18] = t_real0;                                 // whose absolute values place the workgroups of a launch against each other
        acc[19] = t_real1;
        if (lane == 0 && p.dbg) {
            acc[11] = 1;   // (rows of waves that do not exist stay 0)
#pragma unroll
            for (int k = 0; k < 24; ++k) p.dbg[((size_t)blockIdx.x * 8 + wave) * 24 + k] = acc[k];
        }
    }
}

template <class C>
__global__ __launch_bounds__(64 * C::NWAVES, 2) void fwd_mfma_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    run_units<C>(p, unit_list_of(p), (lds_ptr)smem_raw);
}

// Small causal problems: ONE unit per workgroup, 128 query rows, four waves; two workgroups per CU at d = 64, one at d = 128 (where the
// launch simply reaches twice the CUs; jpx then covers every unit and nothing below is paired).  With one 256-row unit per CU (or
// fewer) the launch lasts as long as its heaviest unit while the counted work is the mean -- 4.5 / 8 at BASELINE cfg1.  Here the units
// are half as tall, every CU gets two of them, and the two are the heaviest and the lightest left of its XCD group's heads: workgroups
// are dispatched in index order, one per CU and round, so x + 8 s (s < jpx) and x + 8 (jpx + s) share a CU; the first takes the
// s-th heaviest unit of group x (causal: the highest query blocks of its heads), the second the s-th lightest.  Blocks below hp
// (the rows that see fewer than FA_EARLY_KEYS keys) run configuration CB, the others CA (include/flash_attention.h, "Precision ...").
// Without the mask the units are equal and the order means nothing: the kernel is then used only where 256-row units would leave half of
// the CUs idle (one unit per workgroup, one dispatch round), and hp is all or nothing.
// (where the LDS leaves room for one workgroup per CU only, its four waves have a SIMD each: the whole 512-register file)
template <class CA, class CB>
__global__ __launch_bounds__(64 * CA::NWAVES, (CA::LDS_BYTES > 81920 || CB::LDS_BYTES > 81920) ? 1 : 2) void fwd_mfma_pair_kernel(const Params p, const int hp, const int jpx) {
    static_assert(CA::NWAVES == 4 && CB::NWAVES == 4 && CA::CAUSAL == CB::CAUSAL, "one workgroup shape, one mask");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int x = blockIdx.x & 7, s = blockIdx.x >> 3;
    const int heads = p.B * p.H, hpx = (heads + 7) / 8;
    const int h0 = x * hpx, nh = min(hpx, heads - h0);
    if (nh <= 0) return;
    const int n = nh * p.nQ;                                   // this group's units, heaviest first: block nQ-1 of its heads, nQ-2, ...
    const int idx = s < jpx ? s : n - 1 - (s - jpx);
    if (idx >= n || (s >= jpx && idx < jpx)) return;
    const int blk = idx / nh;
    const UnitList one{p.nQ, p.nQ - 1 - blk, h0 + (idx - blk * nh), 0, 0};
    if constexpr (std::is_same_v<CA, CB>) {   // a mixed-precision configuration: the unit's block says which (Params::hp == hp)
        run_units<CA, Kind::ONE>(p, one, (lds_ptr)smem_raw);
    } else {
        if (one.qb0 < hp) run_units<CB, Kind::ONE>(p, one, (lds_ptr)smem_raw);
        else run_units<CA, Kind::ONE>(p, one, (lds_ptr)smem_raw);
    }
}

}  // namespace fa
