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
// Tuning decisions and the measured alternatives (64 rows per wave, one unit per workgroup, ...) are the
// fields of fa::Opt below.
//
// Optimistic max.  exp2 / bf16 / f32 accumulation have ~2^127 of headroom, so the first pass takes
// every exponential relative to the row max of tile 0 and issues no per-tile max, decision or rescale
// (-4.5 % time).  If a later score exceeds that reference by more than the headroom (or P.V overflows),
// l or O becomes inf/NaN; each lane tests that at the end of the pass, __syncthreads_or makes it
// workgroup-uniform, and the whole block is recomputed by the tracked pass (running max, lazy rescale
// with threshold 2^THR), which is always safe.
#pragma once

#include "computers.hip.h"
#include "computers16.hip.h"

#include <type_traits>

namespace fa {

// Everything about the kernel that is a tuning decision rather than part of the problem.  The defaults ARE the
// production configuration; tests/fa_tune instantiates the alternatives with designated initialisers, e.g.
// KernelCfg<128, true, __bf16, 2, Opt{.persist = false}>.  What each rejected alternative measured: DESIGN.md section 4.
struct Opt {
    bool stamp = false;          // diagnostic build: s_memtime stamps around the segments (tests/fa_tune)
    bool optimistic = true;      // optimistic pass + finiteness check + tracked fallback (false: tracked pass only)
    int npre = 4;                // K fragments in flight ahead of their MFMA
    int vpre = 2;                // V^T fragments in flight ahead of their MFMA
    int thr = 8;                 // lazy-rescale threshold of the tracked pass, log2 units
    int r = 1;                   // 32-row query groups per wave: 1 = 8 waves, two per SIMD; 2 = 4 waves (experimental arm)
    int waves = 0;               // waves per workgroup (0: 8 / r).  4 with r = 1: 128-row units, and where the ring is <= 80 KiB (d = 64) TWO
                                 // workgroups per CU with independent barriers (the grid is then two workgroups per CU)
    bool asm_mfma = false;       // inline-asm MFMAs with dictated register classes (needed by r = 2)
    int valu_first = -1;         // phase-A slots issue their softmax slice before the MFMA: +2 % on the 32x32x16 engine (it covers the
                                 // fragment's LDS latency), -1 % on the 16x16x32 engine.  -1: on for 32x32x16, off for 16x16x32
    bool persist = true;         // one workgroup per CU walks a static list of units (see work_unit)
    bool lds_epilogue32 = true;  // fp32 outputs leave through LDS as whole 256-byte row pieces
    bool pad = false;            // the tensors' head dimension is smaller than D: rows are zero-padded on the fly
    int coalesced_q = -1;        // Q rows fetched whole and turned into fragments through LDS (q_rows_to_fragments);
                                 // -1: on at d = 128 (+0.8 %), off at d = 64 (the 46 us cfg1 loses 1.8 % to the extra LDS trip)
    int mxqk = -1;               // fp8 inputs: QK^T on the block-scaled 32x32x64 MFMA with unit scales (-1: on iff fp8)
    int m16 = -1;                // both products on v_mfma_f32_16x16x32_bf16 (computers16.hip.h) instead of 32x32x16: the chip holds
                                 // a higher clock on that shape (power).  -1: on for bf16 inputs with r = 1
    int sum_mfma = -1;           // 16x16x32 engine: row sums from ONES.P^T MFMAs (sums the bf16-rounded weights) instead of one v_add_f32
                                 // per score.  -1: on (the library turns it off in the kernels that return the LSE)
    int stream = 0;              // REJECTED BY MEASUREMENT (kept as an arm: tests/fa_tune "unit streaming").  16x16x32 engine, persistent grid:
                                 // ONE continuous K/V tile stream across the units of a workgroup -- a unit's last iteration scores the next
                                 // unit's tile 0 (its QK^T phase otherwise runs on garbage), the next unit's tiles 0, 1 and Q arrive under the
                                 // current unit's last iterations, the epilogue gets LDS of its own: no per-unit prologue (stream_units16).
                                 // Bitwise-identical results, per-unit fixed cost 5.1k -> 1.2k cycles, but the tile loop itself got 4-15 %
                                 // slower in every form tried (the seam's conditional code inside the loop body costs hipcc's schedule more
                                 // than the prologues cost): -3 % non-causal, -4 % causal (profiles/r02_tune_g_unit_streaming.log).  With the
                                 // seam's two iterations peeled out of the loop instead (five inlined copies of the tile step) the register
                                 // allocator spills 1 KB per lane into the loop: -33 % (profiles/r02_tune_h_unit_streaming_peeled.log)
    bool dma = true;             // K/V tiles global -> LDS by `buffer_load ... lds` (loaders.hip.h: DmaStage): no staging registers (-16 to -20 VGPRs),
                                 // no ds_write; +2.4 ... +4.5 % on both engines, causal and not (profiles/r02_tune_m_lds_dma.log).  Applies to
                                 // bf16, unpadded rows, 8 waves (KernelCfg::DMA); padded / fp8 / fp16-weights kernels convert or zero-fill
                                 // between the load and the LDS write and keep the register path.  The epilogue's LDS regions sit behind ring
                                 // slot 0: the next unit's tile 0 lands there while the epilogue runs
    bool dma_save_m0 = false;    // (arm) LDS-DMA statements save and restore M0 around themselves (2 more scalar instructions per piece)
    bool early_tile0 = false;    // (arm, measured +-0.7 %: nothing) LDS-DMA kernels: the next unit is decoded and its tile 0 requested BEFORE the finiteness check of the
                                 // optimistic pass (under the check's barrier) instead of after it
    bool pk_fma = false;         // (arm, REJECTED: -7 %) 16x16x32 engine: the exponent arguments s*c - m of two adjacent accumulator registers
                                 // from ONE v_pk_fma_f32 -- 16 VALU instructions fewer per wave-tile, no extra moves in the ISA, bitwise-equal
                                 // results, and 7 % slower at every shape (profiles/r02_tune_s_packed_fma.log): next to MFMAs a packed-fp32
                                 // instruction costs more than the two scalar ones it replaces
    bool pingpong = false;       // (arm, REJECTED: -10 %) 32x32x16 engine, bf16, LDS-DMA: every wave alternates an MFMA phase (QK^T of the next tile +
                                 // P.V of this one, nothing else) with a softmax phase, the two waves of a SIMD in opposite phases between
                                 // workgroup barriers (two per tile), 4-slot ring, one score buffer (attention_pass_pp).  Bitwise-equal results.
                                 // The MFMA phase of a wave takes 1400 cycles for 32 MFMAs (1024 pipe cycles) however deep its fragment
                                 // prefetch: a wave's own LDS reads and waits do not overlap its own MFMAs, so ONE wave cannot keep the pipe
                                 // full, and the interleaved schedule, where both waves feed it, stays ahead (profiles/r02_tune_x_pingpong_phases.log)
    bool qk_pair_order = false;  // 32x32x16 engine, bf16: QK^T fragments ordered so that consecutive MFMAs share their Q fragment
    bool p_f16 = false;          // 16x16x32 engine: weights rounded to fp16 (11 significant bits) instead of bf16 (8), V staged as fp16, P.V
                                 // on v_mfma_f32_16x16x32_f16: the precision option behind FA_FLAG_F16_WEIGHTS (needs |V| <= 65504)
    // ---- rejected by measurement, kept as arms of the tuner (numbers: causal / non-causal headline shape) ----
    int wg = 1;                  // K / V^T fragments consumed per s_waitcnt (1: hipcc's one wait per MFMA; 2: -2 %, 4: 0 %)
    int ring = 3;                // LDS ring slots: 3 (tile t+2 staged in iteration t) or 4 (tile t+3: the next tile's first K
                                 // fragments can be requested BEFORE the barrier: -3..-6 % / -1..-2 %)
    bool early_store = false;    // causal: a wave past its diagonal stores its (final) rows while the others still compute
                                 // (direct scattered stores: -2..-5 %)
    bool pk = false;             // packed-fp32 softmax arithmetic (v_pk_fma_f32 / v_pk_add_f32): -10 %
    bool dot2 = false;           // row sums by v_dot2_f32_bf16 over the packed weights: -3 %
    bool skip_last_qk = false;   // a wave's last tile step without the (unused) QK^T MFMAs: -1..-2 %
    // ---- TIMING EXPERIMENTS ONLY (wrong results by construction): bit 0 no per-tile barrier, bit 1 no global loads in
    // the tile loop, bit 2 P.V takes a constant P (no VALU -> MFMA dependency), bit 3 every 32x32x16 MFMA replaced by two
    // 16x16x32 on the same operand registers (same FLOPs, same dataflow shape: what would that MFMA shape cost / save?) ----
    int dbg = 0;
};

template <int D_, bool CAUSAL_, typename OutT_, int ESZ_ = 2, Opt O = Opt{}>
struct KernelCfg {
    static constexpr int D = D_;
    static constexpr bool CAUSAL = CAUSAL_;
    using OutT = OutT_;
    static constexpr int ESZ = ESZ_;                 // bytes per Q/K/V element: 2 = bf16, 1 = fp8 e4m3fn
    static constexpr bool STAMP = O.stamp, OPTIMISTIC = O.optimistic, ASM_MFMA = O.asm_mfma;
    static constexpr bool PERSIST = O.persist, LDS_EPILOGUE32 = O.lds_epilogue32, PK = O.pk, DOT2 = O.dot2;
    static constexpr bool SKIP_LAST_QK = O.skip_last_qk;
    static constexpr bool COALESCED_Q = (O.coalesced_q < 0 ? D_ == 128 : O.coalesced_q != 0) && O.r == 1 && !O.pad;
    static constexpr bool PAD = O.pad;
    static constexpr bool EARLY_STORE = O.early_store && CAUSAL_ && O.r == 1 && O.optimistic;
    static constexpr bool MXQK = O.mxqk < 0 ? ESZ_ == 1 : O.mxqk != 0;
    static constexpr bool M16 = (O.m16 < 0 ? true : O.m16 != 0) && ESZ_ == 2 && O.r == 1 && !O.asm_mfma && O.ring == 3 && !O.skip_last_qk &&
                                !O.pk && !O.dot2 && O.wg == 1;
    static constexpr bool SUM_MFMA = M16 && (O.sum_mfma < 0 ? true : O.sum_mfma != 0);
    static constexpr bool VALU_FIRST = O.valu_first < 0 ? !M16 : O.valu_first != 0;
    static constexpr bool P_F16 = M16 && O.p_f16;
    static constexpr bool QK_PAIR = O.qk_pair_order && ESZ_ == 2 && !M16;
    static constexpr bool EARLY_TILE0 = O.early_tile0;
    static constexpr bool DMA_SAVE_M0 = O.dma_save_m0;
    static constexpr bool PK_FMA = O.pk_fma && M16;
    static constexpr bool DMA = O.dma && ESZ_ == 2 && !O.pad && O.ring == 3 && !O.p_f16;
    static constexpr bool PP = O.pingpong && DMA && !M16 && O.r == 1 && O.waves == 0 && O.optimistic && !O.asm_mfma;
    // fp8 inputs: K alone by LDS-DMA (V is widened to bf16 on its way into LDS and keeps the register path): loaders.hip.h, HybridStageFp8
    static constexpr bool DMA_K8 = O.dma && ESZ_ == 1 && D_ == 128 && !O.pad && O.r == 1 && O.ring == 3 && O.waves == 0;
    static constexpr bool STREAM = M16 && O.persist && O.optimistic && O.stream != 0;
    static constexpr int NPRE = O.npre, VPRE = O.vpre, THR = O.thr, WG = O.wg;
    static constexpr int R = O.r;                    // 32-row query groups per wave (1 or 2)
    static constexpr int NWAVES = O.waves > 0 ? O.waves : 8 / O.r;   // waves per workgroup
    static constexpr int QBLK = 32 * O.r * NWAVES;   // query rows of a unit (256 in production)
    static_assert(QBLK == 256 || QBLK == 128, "units of 256 or 128 query rows");
    static constexpr int DBG = O.dbg;
    static constexpr bool DBG_NOBAR = DBG & 1, DBG_NOLOAD = DBG & 2, DBG_PCONST = DBG & 4, DBG_M16 = DBG & 8;
    static constexpr int RING = O.ring;
    static_assert(RING == 3 || RING == 4, "3- or 4-slot ring");
    static constexpr int RING_BYTES = (PP ? 4 : RING) * TileGeom<D_, ESZ_>::SLOT;   // (ping-pong schedule: tile t+3 is staged in iteration t)
    // the fp32 LDS epilogue stages 256 rows x 64 floats: more than the ring at d = 64
    // streamed units: the epilogue's staging regions (per wave 16 rows x D 2-byte outputs, or x 64 floats) sit BEHIND the ring
    static constexpr int EP_WAVE_BYTES = 16 * (sizeof(OutT_) == 2 ? D_ * 2 : 256);
    static constexpr int EP_BYTES = NWAVES * EP_WAVE_BYTES;
    // LDS-DMA staging: the epilogue regions (QBLK rows of D 2-byte outputs, or of 64 floats) sit behind ring slot 0
    static constexpr int EP_OFF = (DMA || DMA_K8) ? TileGeom<D_, ESZ_>::SLOT : 0;
    static constexpr int EP_NEED = EP_OFF + (sizeof(OutT_) == 2 ? QBLK * D_ * 2 : (O.lds_epilogue32 ? QBLK * 256 : 0));
    static constexpr int LDS_BYTES = STREAM ? RING_BYTES + EP_BYTES : (EP_NEED > RING_BYTES ? EP_NEED : RING_BYTES);
    static_assert(LDS_BYTES <= 163840 - 256, "160 KiB of LDS per CU, 256 bytes of which __syncthreads_or takes statically");
};

// What the library launches: the defaults of Opt.
// Which engine: without the mask the kernel is power-bound and the 16x16x32 engine's cheaper MFMAs win (+0.9 ... +4.4 % over five
// boxes); under the causal mask the idle stretches (per-unit prologue / epilogue, diagonal block) leave power to spare, cycles
// decide, and the 32x32x16 engine's lower issue pressure wins (16x16x32: -6.8 ... +1.3 %, mean -2.3 %).  DESIGN.md section 4.
template <int D, bool CAUSAL, typename OutT, int ESZ = 2, bool STAMP = false, bool PAD = false, bool LSE = false>
using ProdCfg = KernelCfg<D, CAUSAL, OutT, ESZ, Opt{.stamp = STAMP, .pad = PAD, .m16 = CAUSAL ? 0 : -1, .sum_mfma = LSE ? 0 : -1}>;

// The per-wave compute engine of a configuration: 16x16x32 MFMAs (computers16.hip.h) or 32x32x16 (computers.hip.h).
template <class C>
using WaveComputeOf = std::conditional_t<C::M16, WaveCompute16<C>, WaveCompute<C>>;

// Where a wave's rows go, for the early store of the optimistic causal pass (see attention_pass, step kind 2).
struct RowSink {
    char* Oh;
    float* lse_head;
    int64_t oSb;
    int orow_bytes;
    bool stored;      // this wave's rows of the current unit have already been written
};

// One pass over all KV tiles of the workgroup's query block.  Returns (workgroup-uniform) whether the
// result has to be recomputed with max tracking (only ever true for TRACK = false).
// before_check(): called once, after the last tile's barrier (no wave reads the ring any more) and before the finiteness check of
// the optimistic pass -- the persistent kernel decodes the next unit and starts its tile 0 there, under the check's barrier.
struct NoPreCheck { __device__ __forceinline__ void operator()() const {} };
template <class C, bool TRACK, class PreCheck = NoPreCheck>
__device__ __forceinline__ bool attention_pass(const Params& p, WaveComputeOf<C>& w, typename WaveComputeOf<C>::Stage& st, lds_ptr smem,
                                               int n_tiles, int my_tiles, int q_row0, int lane,
                                               unsigned long long (&acc)[15], bool tile0_in_flight, RowSink& sink,
                                               PreCheck&& before_check = PreCheck{}) {
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
    if (!tile0_in_flight) st.load_all_into(0, smem);
    st.write_all(smem);
    st.wait_all();
    __syncthreads();
    st.load_all_into(1, smem + SLOT);          // past-the-end tiles read as zeros (buffer range check)
    constexpr int AHEAD = C::RING - 1;                 // iteration t stages tile t + AHEAD
    u32x4 r2[WC::Stage::NL];                           // 4-slot ring: tile 2 travels with tile 1
    if constexpr (C::RING == 4) st.load_all_to(r2, 2);
    if constexpr (C::STAMP) tp1 = cycle_stamp();
    if (my_tiles > 0) {
        w.qk_all(smem, kbase, sA);
        if (needs_mask(0)) w.mask(sA, 0, q_row0, S, lane);
        w.first_max(sA, c);   // m = row max of tile 0 (the reference of the optimistic pass)
    }
    st.write_all(smem + SLOT);
    if constexpr (C::RING == 4) st.write_all_from(r2, smem + 2 * SLOT);
    st.wait_all();
    __syncthreads();
    if constexpr (C::RING == 4) w.k_prefetch(smem + SLOT, kbase);   // K(1) fragments of step 0 (later steps: before their barrier)
    if constexpr (C::STAMP) { tp2 = cycle_stamp(); acc[8] += tp1 - tp0; acc[9] += tp2 - tp1; }

    // ring slot byte offsets of tiles t, t+1, [t+2,] t+AHEAD
    int so_cur = 0, so_nxt = SLOT, so_n2 = 2 * SLOT, so_wr = AHEAD * SLOT;
    // kind: 0 = full step (a next tile exists), 1 = the wave's last tile (no QK^T), 2 = staging only (the wave is
    // past its causal diagonal but still stages its share of the tiles the other waves need)
    auto step = [&](int t, int kind, typename WC::ScoresT& cur, typename WC::ScoresT& nxt) {
        unsigned long long t0 = 0, t4 = 0, t6 = 0;
        if constexpr (C::STAMP) t0 = cycle_stamp();
        if (kind == 0 || (kind == 1 && !C::SKIP_LAST_QK)) {
            const bool has_next = kind == 0;
            w.template tile_step<TRACK>(st, t + AHEAD, smem + so_wr, smem + so_nxt, smem + so_cur + KT, kbase, vbase, c, cur, nxt,
                                        has_next, has_next && needs_mask(t + 1), (t + 1) * KVBLK, q_row0, S, lane, smem + so_n2);
        } else if (kind == 1) {
            if constexpr (C::SKIP_LAST_QK)
                w.template tile_step<TRACK, true>(st, t + AHEAD, smem + so_wr, smem + so_nxt, smem + so_cur + KT, kbase, vbase, c, cur,
                                                  nxt, false, false, 0, q_row0, S, lane, smem + so_n2);
        } else {
            st.load_all_into(t + AHEAD, smem + so_wr);
            st.write_all(smem + so_wr);
            if constexpr (C::EARLY_STORE && !TRACK) {
                // This wave is past its causal diagonal: its O and l are final, and it has nothing to do but stage
                // for the others.  Write its rows now (direct form: the LDS staging regions alias the live ring), so
                // the epilogue after the loop is left to the waves on the diagonal.  Should the finiteness check
                // fail afterwards, the tracked pass recomputes the block and every wave stores again.
                if (!sink.stored && my_tiles > 0) {
                    int lane_s = lane;   // keep the store addresses out of the tile loop's live ranges (they would spill)
                    asm volatile("" : "+v"(lane_s));
                    w.template store_o<typename C::OutT>(sink.Oh, sink.lse_head, sink.oSb, q_row0, p.S, lane_s, sink.orow_bytes);
                    sink.stored = true;
                }
            }
        }
        if constexpr (C::STAMP) t4 = cycle_stamp();
        st.wait_all();   // (LDS-DMA staging: this wave's pieces of tile t + AHEAD have landed)
        if constexpr (C::DBG_NOBAR) {
        } else if constexpr (C::RING == 4) {
            // __syncthreads() would drain lgkmcnt(0) and with it the K fragments just requested for the next
            // iteration.  LDS operations of a wave complete in order, and those NPRE reads are the last ones this
            // step issued: waiting until only they are outstanding covers every ds_write of the staged tile.
            if (kind != 2) asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"(WC::NPRE) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            __syncthreads();
        }
        if constexpr (C::STAMP) {
            t6 = cycle_stamp();
            if (kind != 2) { acc[1] += w.t_mid - t0; acc[2] += w.t_end - w.t_mid; acc[3] += t4 - w.t_end; acc[6] += 1; }
            acc[5] += t6 - t4;
        }
        const int tmp = so_cur;
        so_cur = so_nxt;
        if constexpr (C::RING == 4) { so_nxt = so_n2; so_n2 = so_wr; }
        else so_nxt = so_wr;
        so_wr = tmp;
    };
    if constexpr (C::SKIP_LAST_QK) {
        // full steps in ping-pong pairs, then ONE instance of the last-tile step (always on sA: an odd count of
        // full steps copies sB over once per unit), then the staging-only steps.  Every wave runs n_tiles steps.
        const int n_full = my_tiles > 0 ? my_tiles - 1 : 0;
        int t = 0;
        while (t < n_full) {
            step(t, 0, sA, sB);
            ++t;
            if (t < n_full) { step(t, 0, sB, sA); ++t; }
            else sA = sB;
        }
        if (my_tiles > 0) { step(t, 1, sA, sB); ++t; }
        for (; t < n_tiles; ++t) step(t, 2, sA, sB);
    } else {
        auto kind_of = [&](int t) { return t + 1 < my_tiles ? 0 : (t < my_tiles ? 1 : 2); };
        for (int t = 0; t < n_tiles; t += 2) {
            step(t, kind_of(t), sA, sB);
            if (t + 1 < n_tiles) step(t + 1, kind_of(t + 1), sB, sA);
        }
    }
    if constexpr (TRACK) return false;
    else {
        before_check();
        unsigned long long tc0 = 0;
        if constexpr (C::STAMP) tc0 = cycle_stamp();
        const bool bad = __syncthreads_or(my_tiles > 0 && w.not_finite()) != 0;
        if constexpr (C::STAMP) acc[10] += cycle_stamp() - tc0;
        return bad;
    }
}

// The optimistic pass in the ping-pong schedule (Opt::pingpong; computers.hip.h: m_phase / v_phase).  Every wave runs
//     [ M(t): S(t+1) = K(t+1).Q^T, O^T += V(t)^T.P(t)^T | barrier | V: P(t+1) from S(t+1) | barrier ]   per tile,
// and waves 4-7 (group B: the SIMDs' second waves) enter that loop ONE barrier interval later than waves 0-3 (group A) -- they spend
// the first interval forming P(0), which group A does before the loop, and group A spends one idle interval at the end -- so between
// any two barriers one wave of a SIMD is in its MFMA phase and the other in its softmax phase.  4-slot ring: a wave issues its DMA
// pieces of tile t+3 inside M(t), into the slot of tile t-1 (last read by group B's M(t-1), which ended at the previous barrier), and
// waits for them (vmcnt(0)) before the barrier that ends its following V phase; the tile is first read two intervals after that.
// Same contract as attention_pass<C, false>.
template <class C, class PreCheck = NoPreCheck>
__device__ __forceinline__ bool attention_pass_pp(const Params& p, WaveComputeOf<C>& w, typename WaveComputeOf<C>::Stage& st, lds_ptr smem,
                                                  int n_tiles, int my_tiles, int q_row0, int lane, unsigned long long (&acc)[15],
                                                  RowSink& sink, PreCheck&& before_check = PreCheck{}) {
    using G = TileGeom<C::D, C::ESZ>;
    using WC = WaveComputeOf<C>;
    constexpr int KVBLK = 64, SLOT = G::SLOT, KT = G::K_TILE;
    const int S = p.Sk;
    const bool group_b = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= 4;
    w.init();
    const int kbase = kd_read_base(lane, G::KBLK), vbase = v_read_base(lane);
    const float c = p.scale_log2;
    auto needs_mask = [&](int t) { return (C::CAUSAL && t * KVBLK + KVBLK - 1 > q_row0) || (t * KVBLK + KVBLK > S); };
    typename WC::ScoresT s;

    // prologue: tile 0 is on its way (requested with Q).  Slots 1 and 2 may still hold other waves' Q staging regions until the
    // first barrier; then tiles 1 and 2 are requested, and S(0), its row max and P(0) are formed under their flight
    st.wait_all();
    __syncthreads();
    st.load_all_into(1, smem + SLOT);
    st.load_all_into(2, smem + 2 * SLOT);
    if (my_tiles > 0) {
        w.qk_all(smem, kbase, s);
        if (needs_mask(0)) w.mask(s, 0, q_row0, S, lane);
        w.first_max(s, c);
        w.v_phase(s, c);
    }
    st.wait_all();
    __syncthreads();
    if (group_b) __syncthreads();                      // group B's leading interval (group A is in M(0))
    int so_cur = 0, so_nxt = SLOT, so_wr = 3 * SLOT;   // ring slots of tiles t, t+1, t+3
    for (int t = 0; t < n_tiles; ++t) {
        const int kind = t + 1 < my_tiles ? 0 : (t < my_tiles ? 1 : 2);
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        if constexpr (C::STAMP) t0 = cycle_stamp();
        __builtin_amdgcn_s_setprio(2);
        if (kind == 0) w.template m_phase<true>(st, t + 3, smem + so_wr, smem + so_nxt, smem + so_cur + KT, kbase, vbase, s);
        else if (kind == 1) w.template m_phase<false>(st, t + 3, smem + so_wr, smem + so_nxt, smem + so_cur + KT, kbase, vbase, s);
        else st.load_all_into(t + 3, smem + so_wr);
        __builtin_amdgcn_s_setprio(0);
        if constexpr (C::STAMP) t1 = cycle_stamp();
        __syncthreads();
        if constexpr (C::STAMP) t2 = cycle_stamp();
        if (kind == 0) {                               // P(t+1) from the S(t+1) just produced
            if (needs_mask(t + 1)) w.mask(s, (t + 1) * KVBLK, q_row0, S, lane);
            w.v_phase(s, c);
        }
        st.wait_all();                                 // this wave's pieces of tile t+3 have landed
        if constexpr (C::STAMP) t3 = cycle_stamp();
        __syncthreads();
        if constexpr (C::STAMP) {                      // (tuner's columns: "phase A" = MFMA phase, "phase B" = softmax phase, "end-of-tile" = wait after M)
            t4 = cycle_stamp();
            if (kind == 0) { acc[1] += t1 - t0; acc[2] += t3 - t2; acc[3] += t2 - t1; acc[6] += 1; }
            acc[5] += t4 - t3;
        }
        const int nx = so_nxt + SLOT == 4 * SLOT ? 0 : so_nxt + SLOT;
        so_cur = so_nxt;
        so_nxt = nx;
        so_wr = so_wr + SLOT == 4 * SLOT ? 0 : so_wr + SLOT;
    }
    if (!group_b) __syncthreads();                     // group A's trailing interval (group B is in its last V phase)
    before_check();
    (void)sink;
    return __syncthreads_or(my_tiles > 0 && w.not_finite()) != 0;
}

// Work assignment.  Non-persistent: one unit per workgroup (loaders.hip.h: unit_of_block).  Persistent: the
// grid is one workgroup per CU; XCD group x = bid & 7 still owns the contiguous units [x*cpx, (x+1)*cpx), and
// its jpx workgroups walk them in rounds of jpx consecutive units -- at any moment the group works on a few
// adjacent heads, whose K/V stay in that XCD's L2.  Odd rounds run in reverse order ("snake"): under the
// causal mask a head's query blocks are listed heaviest first, so workgroup j gets cost c in one round and
// (max+1-c) in the next -- a static schedule whose per-workgroup totals are equal when nQ divides jpx.
template <class C>
__device__ __forceinline__ bool work_unit(const Params& p, int round, int& g, int& qb) {
    const int bid = blockIdx.x;
    int idx = bid >> 3;
    if constexpr (C::PERSIST) {
        idx = round * p.jpx + ((round & 1) ? p.jpx - 1 - idx : idx);
        if (idx >= p.cpx) return false;
    } else if (round > 0) {
        return false;
    }
    const int u = (bid & 7) * p.cpx + idx;
    if (u >= p.units) return false;
    g = u / p.nQ;
    qb = u - g * p.nQ;
    if (C::CAUSAL) qb = p.nQ - 1 - qb;  // heaviest query blocks of a head first
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
    __device__ __forceinline__ void set(const Params& p, int g, int qb, int wave) {
        constexpr int ESZ = C::ESZ, KVBLK = 64, QBLK = C::QBLK, WROWS = 32 * C::R;
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

// Streamed units (C::STREAM; 16x16x32 engine, optimistic pass).  A workgroup's units form ONE tile stream: iteration t of a unit
// with n tiles stages stream element t+2 -- the next unit's tile t+2-n once t+2 >= n -- and its LAST iteration, whose QK^T phase
// has no tile of its own left to score, scores the next unit's tile 0 against the next unit's Q (requested between the two
// phases of the second-to-last iteration, when this unit's Q is dead).  At the seam only the finiteness check, the epilogue
// (through LDS regions of its own, behind the ring) and the row max of the new tile 0 remain; the ring rotation just continues.
// Per unit this removes: Q load + LDS trip, staging of tiles 0 and 1 with their two barriers, and the unoverlapped S(0) = K(0).Q^T.
// A unit that fails the finiteness check is recomputed from scratch by the tracked pass (attention_pass<C, true>), after which
// -- as after a unit with fewer than two tiles -- the stream restarts with a full prologue.
template <class C>
__device__ __forceinline__ void stream_units16(const Params& p, lds_ptr smem) {
    constexpr int D = C::D, ESZ = C::ESZ, KVBLK = 64, WROWS = 32;
    using OutT = typename C::OutT;
    using WC = WaveCompute16<C>;
    using G = TileGeom<D, ESZ>;
    constexpr int SLOT = G::SLOT, KT = G::K_TILE;
    int g, qb, round = 0;
    if (!work_unit<C>(p, 0, g, qb)) return;
    unsigned long long t_kernel0 = 0;
    if constexpr (C::STAMP) t_kernel0 = cycle_stamp();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = p.S, Sk = p.Sk;
    const int64_t qSb = p.qS * ESZ, kSb = p.kS * ESZ, vSb = p.vS * ESZ, oSb = p.oS * (int64_t)sizeof(OutT);
    const int row_bytes = C::PAD ? p.d * ESZ : D * ESZ, orow_bytes = C::PAD ? p.d * (int)sizeof(OutT) : D * (int)sizeof(OutT);
    const float c = p.scale_log2;

    // A unit's pointers and counts are a pure function of (p, round): they are recomputed where they are needed (seam code)
    // instead of living in registers across the tile loop -- the loop keeps only n_tiles, my_tiles and q_row0.
    auto unit_of = [&](int rnd, UnitCtx<C>& u) {
        int gg, qq;
        if (!work_unit<C>(p, rnd, gg, qq)) return false;
        u.set(p, gg, qq, wave);
        return true;
    };
    WC w;
    typename WC::Stage st;
    {
        UnitCtx<C> u0;
        u0.set(p, g, qb, wave);
        st.init(u0.Kh, u0.Vh, kSb, vSb, Sk, wave, lane, row_bytes);
        st.load_all_into(0, smem);                      // tile 0 and Q travel together (one HBM round trip)
        if constexpr (C::COALESCED_Q) w.load_q_rows(u0.Qh, qSb, u0.q_row0, S, lane);
        else w.load_q(u0.Qh, qSb, u0.q_row0, S, lane, row_bytes);
    }
    bool q_as_rows = C::COALESCED_Q;                 // only the workgroup's first unit takes the coalesced form (the ring is empty then)
    unsigned long long acc[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    Scores16 sA, sB;
    bool fresh = true;                               // this unit starts with a full prologue
    int so_cur = 0, so_nxt = SLOT, so_wr = 2 * SLOT; // ring slot byte offsets of stream elements t, t+1, t+2

    while (true) {
        // lane-derived values are unit-invariant: an opaque copy keeps hipcc from hoisting (and spilling) them
        int lane_u = lane;
        asm volatile("" : "+v"(lane_u));
        const int kbase = C::DMA ? kd16_read_base(lane_u, G::KBLK) : k16_read_base(lane_u), vbase = v16_read_base<D>(lane_u);
        int q_row0, n_tiles, my_tiles;
        {
            UnitCtx<C> cu;
            unit_of(round, cu);
            q_row0 = cu.q_row0; n_tiles = cu.n_tiles; my_tiles = cu.my_tiles;
        }
        auto needs_mask = [&](int t) { return (C::CAUSAL && t * KVBLK + KVBLK - 1 > q_row0) || (t * KVBLK + KVBLK > Sk); };
        unsigned long long t_u0 = 0;
        if constexpr (C::STAMP) t_u0 = cycle_stamp();

        // ---------- start of a unit ----------
        if (fresh) {
            w.pin_q();
            if (q_as_rows) {
                static_assert(!C::COALESCED_Q || G::SLOT + C::QBLK * D * ESZ <= C::LDS_BYTES, "Q staging regions must fit behind slot 0");
                if constexpr (C::COALESCED_Q) {
                    w.q_rows_to_fragments(smem + G::SLOT + wave * (WROWS * D * ESZ), lane_u);
                    w.pin_q();
                }
                q_as_rows = false;
            }
            st.write_all(smem);
            st.wait_all();
            __syncthreads();
            st.load_all_into(1, smem + SLOT);          // past-the-end tiles read as zeros (buffer range check)
            if (my_tiles > 0) w.qk_all(smem, kbase, sA);
            st.write_all(smem + SLOT);
            st.wait_all();
            __syncthreads();
            so_cur = 0; so_nxt = SLOT; so_wr = 2 * SLOT;
        }
        w.init();
        if (my_tiles > 0) {
            if (needs_mask(0)) w.mask(sA, 0, q_row0, Sk, lane_u);
            w.first_max(sA, c);   // m = row max of tile 0 (the reference of the optimistic pass)
        }
        if constexpr (C::STAMP) acc[9] += cycle_stamp() - t_u0;

        // the unit after this one: its heads feed the tail of this unit's tile stream
        bool more, stream, next_live;
        {
            UnitCtx<C> nx;
            more = unit_of(round + 1, nx);
            stream = more && n_tiles >= 2;
            next_live = stream && nx.my_tiles > 0;
            if (stream) st.set_next(nx.Kh, nx.Vh, n_tiles);
        }

        // ---------- tile loop ----------
        auto step = [&](int t, Scores16& cs, Scores16& ns) {
            unsigned long long t0 = 0, t4 = 0, t6 = 0;
            if constexpr (C::STAMP) t0 = cycle_stamp();
            const int kind = t + 1 < my_tiles ? 0 : (t < my_tiles ? 1 : 2);
            const bool q_now = next_live && t == n_tiles - 2, last = t == n_tiles - 1;
            auto fetch_next_q = [&]() {
                if (q_now) {   // wave-uniform, once per unit: the next unit's Q fragments replace this unit's (dead from here on)
                    UnitCtx<C> nx;
                    unit_of(round + 1, nx);
                    w.load_q(nx.Qh, qSb, nx.q_row0, S, lane_u, row_bytes);
                }
            };
            if (kind != 2) {
                const bool has_next = kind == 0;
                w.template tile_step<false, false>(st, t + 2, smem + so_wr, smem + so_nxt, smem + so_cur + KT, kbase, vbase, c, cs, ns, has_next,
                                                   has_next && needs_mask(t + 1), (t + 1) * KVBLK, q_row0, Sk, lane_u);
                st.wait_all();   // (LDS-DMA staging: this wave's pieces of stream element t + 2 have landed; BEFORE the Q loads are issued)
                fetch_next_q();
            } else {
                // past this wave's causal diagonal: it stages its share of the stream, and at the seam fetches / scores for the next unit
                st.load_all_into(st.select(t + 2), smem + so_wr);
                st.write_all(smem + so_wr);
                st.wait_all();
                fetch_next_q();
                if (last && next_live) {
                    w.pin_q();
                    w.qk_all(smem + so_nxt, kbase, ns);
                }
            }
            if constexpr (C::STAMP) t4 = cycle_stamp();
            if constexpr (!C::DBG_NOBAR) __syncthreads();
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
        for (int t = 0; t < n_tiles; t += 2) {
            step(t, sA, sB);
            if (t + 1 < n_tiles) step(t + 1, sB, sA);
        }
        if (next_live && (n_tiles & 1)) sA = sB;   // the next unit's S(0) continues in sA

        // ---------- end of the unit ----------
        unsigned long long tc0 = 0;
        if constexpr (C::STAMP) tc0 = cycle_stamp();
        const bool bad = __syncthreads_or(my_tiles > 0 && w.not_finite()) != 0;
        if constexpr (C::STAMP) acc[10] += cycle_stamp() - tc0;
        UnitCtx<C> cu;
        unit_of(round, cu);
        if (bad) {
            // a score outran the optimistic reference by more than the headroom: the tracked pass recomputes the unit from scratch
            st.init(cu.Kh, cu.Vh, kSb, vSb, Sk, wave, lane_u, row_bytes);
            w.load_q(cu.Qh, qSb, q_row0, S, lane_u, row_bytes);
            w.pin_q();
            RowSink sink{cu.Oh, cu.lse_head, oSb, orow_bytes, false};
            attention_pass<C, true>(p, w, st, smem, n_tiles, my_tiles, q_row0, lane_u, acc, false, sink);
            WC::zero(sA);   // (the stream restarts below: tell the register allocator that no score buffer lives across the tracked pass)
        }
        unsigned long long t_ep0 = 0;
        if constexpr (C::STAMP) t_ep0 = cycle_stamp();
        {
            int lane_e = lane;   // (the epilogue's ~40 per-lane addresses must be recomputed here, not hoisted and spilled)
            asm volatile("" : "+v"(lane_e));
            lds_ptr ep = smem + C::RING_BYTES;
            lds_ptr mine = ep + wave * C::EP_WAVE_BYTES;
            if constexpr (sizeof(OutT) == 2) {
                if (cu.wave_live) w.template store_o_lds<OutT>(mine, cu.Oh, cu.lse_head, oSb, q_row0, S, lane_e, orow_bytes);
            } else {
                if (cu.wave_live) w.template store_o_lds32<OutT>(mine, cu.Oh, cu.lse_head, oSb, q_row0, S, lane_e, orow_bytes);
            }
        }
        if constexpr (C::STAMP) acc[4] += cycle_stamp() - t_ep0;
        if (!more) break;
        ++round;
        if (stream && !bad) {
            st.advance();
            fresh = false;
        } else {
            // (every wave left the ring at the barrier inside __syncthreads_or / the tracked pass: it may be refilled)
            UnitCtx<C> nx;
            unit_of(round, nx);
            st.init(nx.Kh, nx.Vh, kSb, vSb, Sk, wave, lane_u, row_bytes);
            st.load_all_into(0, smem);
            w.load_q(nx.Qh, qSb, nx.q_row0, S, lane_u, row_bytes);
            fresh = true;
        }
    }
    if constexpr (C::STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // include the store tail
        acc[0] = cycle_stamp() - t_kernel0;
        if (lane == 0 && p.dbg) {
#pragma unroll
            for (int k = 0; k < 11; ++k) p.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + k] = acc[k];
            p.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + 11] = 1;
        }
    }
}

template <class C>
__global__ __launch_bounds__(64 * C::NWAVES, C::R == 1 ? 2 : 1) void fwd_mfma_kernel(const Params p) {
    constexpr int D = C::D, ESZ = C::ESZ;
    using OutT = typename C::OutT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_ptr smem = (lds_ptr)smem_raw;
    if constexpr (C::STREAM) {
        stream_units16<C>(p, smem);
        return;
    }

    int g, qb, round = 0;
    if (!work_unit<C>(p, 0, g, qb)) return;
    unsigned long long t_kernel0 = 0;
    if constexpr (C::STAMP) t_kernel0 = cycle_stamp();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = p.S, Sk = p.Sk;
    const int64_t qSb = p.qS * ESZ, kSb = p.kS * ESZ, vSb = p.vS * ESZ, oSb = p.oS * (int64_t)sizeof(OutT);
    constexpr int WROWS = 32 * C::R;                // query rows per wave

    UnitCtx<C> cur;
    cur.set(p, g, qb, wave);
    WaveComputeOf<C> w;
    typename WaveComputeOf<C>::Stage st;
    const int row_bytes = C::PAD ? p.d * ESZ : D * ESZ, orow_bytes = C::PAD ? p.d * (int)sizeof(OutT) : D * (int)sizeof(OutT);
    st.init(cur.Kh, cur.Vh, kSb, vSb, Sk, wave, lane, row_bytes);
    st.load_all_into(0, smem);                      // tile 0 and Q travel together (one HBM round trip)
    if constexpr (C::COALESCED_Q) w.load_q_rows(cur.Qh, qSb, cur.q_row0, S, lane);
    else w.load_q(cur.Qh, qSb, cur.q_row0, S, lane, row_bytes);
    unsigned long long acc[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (C::STAMP) acc[12] = cycle_stamp() - t_kernel0;   // setup: unit decode, descriptors, first loads issued

    while (true) {
        unsigned long long t_q0 = 0;
        if constexpr (C::STAMP) t_q0 = cycle_stamp();
        w.pin_q();
        RowSink sink{cur.Oh, cur.lse_head, oSb, orow_bytes, false};
        if constexpr (C::COALESCED_Q) {
            // staging regions sit behind ring slot 0 (tile 0 is about to be written there by other waves); every
            // wave finishes this round trip before the first barrier of the pass, after which slot 1 is written
            using G = TileGeom<D, ESZ>;
            static_assert(G::SLOT + C::QBLK * D * ESZ <= C::LDS_BYTES, "Q staging regions must fit behind slot 0");
            int lane_q = lane;   // keep the 16 staging addresses inside the unit loop (hoisted, they spill)
            if constexpr (C::PERSIST) asm volatile("" : "+v"(lane_q));
            w.q_rows_to_fragments(smem + G::SLOT + wave * (WROWS * D * ESZ), lane_q);
            w.pin_q();
        }
        if constexpr (C::STAMP) acc[7] += cycle_stamp() - t_q0;

        // Lane-derived values of the pass (read bases, the 32 mask compares of tile 0, ...) are unit-invariant: hipcc
        // hoists them out of the unit loop, runs out of SGPRs for the compare masks and spills across the tile loop.
        // An opaque copy of the lane id keeps them inside the pass.
        int lane_p = lane;
        if constexpr (C::PERSIST) asm volatile("" : "+v"(lane_p));
        // Persistent grid: the next unit's tile 0 and Q are requested ahead, so that their HBM round trip runs under this unit's
        // finiteness check and epilogue.  LDS-DMA staging needs no registers for the tile: it is decoded and started BEFORE the
        // check (every wave is past the last tile's barrier, ring slot 0 is free, the epilogue works behind it); the register
        // form waits until the pass is over (its staging registers and Q's are dead then).
        bool more = false, prefetched = false;
        UnitCtx<C> nxt;
        auto prefetch_tile0 = [&]() {
            more = work_unit<C>(p, ++round, g, qb);
            if (more) {
                nxt.set(p, g, qb, wave);
                st.init(nxt.Kh, nxt.Vh, kSb, vSb, Sk, wave, lane, row_bytes);
                st.load_all_into(0, smem);
            }
            prefetched = true;
        };
        constexpr bool EARLY_PREFETCH = C::PERSIST && C::DMA && C::OPTIMISTIC && C::EARLY_TILE0;
        if constexpr (C::OPTIMISTIC) {
            auto before_check = [&]() { if constexpr (EARLY_PREFETCH) prefetch_tile0(); };
            bool failed;
            if constexpr (C::PP) failed = attention_pass_pp<C>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, sink, before_check);
            else failed = attention_pass<C, false>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, true, sink, before_check);
            if (failed) {
                sink.stored = false;   // whatever was written early came from an overflowed pass
                if constexpr (EARLY_PREFETCH) {
                    st.wait_all();     // the next unit's tile 0 is on its way into slot 0: let it land, then the ring is this unit's again
                    st.init(cur.Kh, cur.Vh, kSb, vSb, Sk, wave, lane, row_bytes);
                }
                attention_pass<C, true>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, false, sink);
                if constexpr (EARLY_PREFETCH) {
                    if (more) {        // ... and request it again
                        st.init(nxt.Kh, nxt.Vh, kSb, vSb, Sk, wave, lane, row_bytes);
                        st.load_all_into(0, smem);
                    }
                }
            }
        } else {
            attention_pass<C, true>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, true, sink);
        }

        unsigned long long t_nx0 = 0;
        if constexpr (C::STAMP) t_nx0 = cycle_stamp();
        if constexpr (C::PERSIST) {
            if (!prefetched) prefetch_tile0();
            if (more) {
                // (opaque lane: a hoisted per-lane Q address is spilled across the tile loop, and its reload's vmcnt(0)
                // would make the Q loads wait for the tile-0 loads just issued)
                int lane_n = lane;
                asm volatile("" : "+v"(lane_n));
                if constexpr (C::COALESCED_Q) w.load_q_rows(nxt.Qh, qSb, nxt.q_row0, S, lane_n);
                else w.load_q(nxt.Qh, qSb, nxt.q_row0, S, lane_n, row_bytes);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        unsigned long long t_ep0 = 0;
        if constexpr (C::STAMP) { t_ep0 = cycle_stamp(); acc[13] += t_ep0 - t_nx0; }   // next unit decode + prefetch issue
        // The epilogue's ~40 per-lane addresses must be recomputed here: hoisted out of the unit loop they
        // would live across the tile loop, spill, and their reload (vmcnt(0)) would wait for the prefetch above.
        int lane_e = lane;
        if constexpr (C::PERSIST) asm volatile("" : "+v"(lane_e));
        if constexpr (sizeof(OutT) == 2) {
            // every pass ends behind a workgroup barrier, so no wave still reads the K/V ring: reuse it
            static_assert(C::EP_OFF + C::QBLK * D * 2 <= C::LDS_BYTES, "epilogue regions must fit the ring");
            if (cur.wave_live && !sink.stored)
                w.template store_o_lds<OutT>(smem + C::EP_OFF + wave * (WROWS * D * 2), cur.Oh, cur.lse_head, oSb, cur.q_row0, S, lane_e, orow_bytes);
        } else if constexpr (C::LDS_EPILOGUE32) {
            static_assert(C::EP_OFF + C::QBLK * 64 * 4 <= C::LDS_BYTES, "epilogue regions must fit the ring");
            if (cur.wave_live && !sink.stored)
                w.template store_o_lds32<OutT>(smem + C::EP_OFF + wave * (WROWS * 256), cur.Oh, cur.lse_head, oSb, cur.q_row0, S, lane_e, orow_bytes);
        } else {
            if (cur.wave_live && !sink.stored) w.template store_o<OutT>(cur.Oh, cur.lse_head, oSb, cur.q_row0, S, lane_e, orow_bytes);
        }
        if constexpr (C::STAMP) acc[4] += cycle_stamp() - t_ep0;   // epilogue: normalise + store O (issue side)
        if (!more) break;
        // the next prologue overwrites ring slots that other waves' epilogue regions alias
        if constexpr (sizeof(OutT) == 2 || C::LDS_EPILOGUE32) __syncthreads();
        cur = nxt;
    }
    if constexpr (C::STAMP) {
        const unsigned long long t_tail0 = cycle_stamp();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // include the store tail
        const unsigned long long t_end = cycle_stamp();
        acc[14] = t_end - t_tail0;
        acc[0] = t_end - t_kernel0;                        // whole workgroup lifetime of this wave
        if (lane == 0 && p.dbg) {
#pragma unroll
            for (int k = 0; k < 11; ++k) p.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + k] = acc[k];
            p.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + 11] = 1;   // (rows of waves that do not exist stay 0)
#pragma unroll
            for (int k = 12; k < 15; ++k) p.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + k] = acc[k];
        }
    }
}

}  // namespace fa
