// kernel_bf16.hip.h -- the MFMA forward kernel (bf16 or fp8-e4m3fn inputs, d in {64,128}).
//
// Counterpart of the reference's kernel entry kernels/FlashAttention.cuh:59-84 and of the loop nests of
// kernels/computers.cuh:33-67 / kernels/loaders.cuh:132-156,177-201.  The reference commits a tile and
// immediately waits for it on every cuda::pipeline (no load/compute overlap).
//
// One workgroup = 8 waves (two per SIMD, <= 256 VGPRs) processes UNITS of 256 query rows of one (batch, head), a
// wave owning 32 rows.  The grid is persistent: one workgroup per CU walks a static, balanced list of units
// (work_unit) and requests the next unit's Q and KV tile 0 before it runs the current unit's epilogue.
// KV tiles of 64 keys live in a 3-slot LDS ring [K image | V image] (loaders.hip.h).  Iteration t of a wave reads
// K(t+1) and V(t) and stages tile t+2 into slot (t+2)%3 = slot (t-1)%3, last read in iteration t-1, which
// every wave left at the previous barrier: ONE barrier per tile.  The loop is unrolled x2 with ping-pong
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
    bool asm_mfma = false;       // inline-asm MFMAs with dictated register classes (needed by r = 2)
    bool valu_first = true;      // phase-A slots issue their softmax slice before the MFMA
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
    static constexpr bool STAMP = O.stamp, OPTIMISTIC = O.optimistic, ASM_MFMA = O.asm_mfma, VALU_FIRST = O.valu_first;
    static constexpr bool PERSIST = O.persist, LDS_EPILOGUE32 = O.lds_epilogue32, PK = O.pk, DOT2 = O.dot2;
    static constexpr bool SKIP_LAST_QK = O.skip_last_qk;
    static constexpr bool COALESCED_Q = (O.coalesced_q < 0 ? D_ == 128 : O.coalesced_q != 0) && O.r == 1 && !O.pad;
    static constexpr bool PAD = O.pad;
    static constexpr bool EARLY_STORE = O.early_store && CAUSAL_ && O.r == 1 && O.optimistic;
    static constexpr bool MXQK = O.mxqk < 0 ? ESZ_ == 1 : O.mxqk != 0;
    static constexpr bool M16 = (O.m16 < 0 ? true : O.m16 != 0) && ESZ_ == 2 && O.r == 1 && !O.asm_mfma && O.ring == 3 && !O.skip_last_qk &&
                                !O.pk && !O.dot2 && O.wg == 1;
    static constexpr bool SUM_MFMA = M16 && (O.sum_mfma < 0 ? true : O.sum_mfma != 0);
    static constexpr bool P_F16 = M16 && O.p_f16;
    static constexpr int NPRE = O.npre, VPRE = O.vpre, THR = O.thr, WG = O.wg;
    static constexpr int R = O.r;                    // 32-row query groups per wave (1 or 2)
    static constexpr int NWAVES = 8 / O.r;           // waves per workgroup (256 query rows)
    static constexpr int DBG = O.dbg;
    static constexpr bool DBG_NOBAR = DBG & 1, DBG_NOLOAD = DBG & 2, DBG_PCONST = DBG & 4, DBG_M16 = DBG & 8;
    static constexpr int RING = O.ring;
    static_assert(RING == 3 || RING == 4, "3- or 4-slot ring");
    static constexpr int RING_BYTES = RING * TileGeom<D_, ESZ_>::SLOT;
    // the fp32 LDS epilogue stages 256 rows x 64 floats: more than the ring at d = 64
    static constexpr int LDS_BYTES = (O.lds_epilogue32 && sizeof(OutT_) == 4 && RING_BYTES < 65536) ? 65536 : RING_BYTES;
};

// What the library launches: the defaults of Opt.
template <int D, bool CAUSAL, typename OutT, int ESZ = 2, bool STAMP = false, bool PAD = false, bool LSE = false>
using ProdCfg = KernelCfg<D, CAUSAL, OutT, ESZ, Opt{.stamp = STAMP, .pad = PAD, .sum_mfma = LSE ? 0 : -1}>;

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
template <class C, bool TRACK>
__device__ __forceinline__ bool attention_pass(const Params& p, WaveComputeOf<C>& w, typename WaveComputeOf<C>::Stage& st, lds_ptr smem,
                                               int n_tiles, int my_tiles, int q_row0, int lane,
                                               unsigned long long (&acc)[12], bool tile0_in_flight, RowSink& sink) {
    using G = TileGeom<C::D, C::ESZ>;
    constexpr bool CAUSAL = C::CAUSAL;
    constexpr int KVBLK = 64, SLOT = G::SLOT, KT = G::K_TILE;
    const int S = p.Sk;   // key bound of the masks
    unsigned long long tp0 = 0, tp1 = 0, tp2 = 0;
    if constexpr (C::STAMP) tp0 = cycle_stamp();
    w.init();
    using WC = WaveComputeOf<C>;
    const int kbase = C::M16 ? k16_read_base(lane) : k_read_base(lane);
    const int vbase = C::M16 ? v16_read_base<C::D>(lane) : v_read_base(lane);
    const float c = p.scale_log2;
    auto needs_mask = [&](int t) { return (CAUSAL && t * KVBLK + KVBLK - 1 > q_row0) || (t * KVBLK + KVBLK > S); };
    typename WC::ScoresT sA, sB;

    // Prologue: tile 0 (requested by the caller together with Q on the first pass) -> LDS, barrier; then
    // tile 1 is fetched while S(0) = K(0).Q^T and its row max are computed.
    if (!tile0_in_flight) st.load_all(0);
    st.write_all(smem);
    __syncthreads();
    st.load_all(1);          // past-the-end tiles read as zeros (buffer range check)
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
            st.load_all(t + AHEAD);
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
        unsigned long long tc0 = 0;
        if constexpr (C::STAMP) tc0 = cycle_stamp();
        const bool bad = __syncthreads_or(my_tiles > 0 && w.not_finite()) != 0;
        if constexpr (C::STAMP) acc[10] += cycle_stamp() - tc0;
        return bad;
    }
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
        constexpr int ESZ = C::ESZ, KVBLK = 64, QBLK = 256, WROWS = 32 * C::R;
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

template <class C>
__global__ __launch_bounds__(64 * C::NWAVES, C::R == 1 ? 2 : 1) void fwd_mfma_kernel(const Params p) {
    constexpr int D = C::D, ESZ = C::ESZ;
    using OutT = typename C::OutT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_ptr smem = (lds_ptr)smem_raw;

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
    st.load_all(0);                                 // tile 0 and Q travel together (one HBM round trip)
    if constexpr (C::COALESCED_Q) w.load_q_rows(cur.Qh, qSb, cur.q_row0, S, lane);
    else w.load_q(cur.Qh, qSb, cur.q_row0, S, lane, row_bytes);
    unsigned long long acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    while (true) {
        unsigned long long t_q0 = 0;
        if constexpr (C::STAMP) t_q0 = cycle_stamp();
        w.pin_q();
        RowSink sink{cur.Oh, cur.lse_head, oSb, orow_bytes, false};
        if constexpr (C::COALESCED_Q) {
            // staging regions sit behind ring slot 0 (tile 0 is about to be written there by other waves); every
            // wave finishes this round trip before the first barrier of the pass, after which slot 1 is written
            using G = TileGeom<D, ESZ>;
            static_assert(G::SLOT + 256 * D * ESZ <= C::LDS_BYTES, "Q staging regions must fit behind slot 0");
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
        if constexpr (C::OPTIMISTIC) {
            if (attention_pass<C, false>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, true, sink)) {
                sink.stored = false;   // whatever was written early came from an overflowed pass
                attention_pass<C, true>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, false, sink);
            }
        } else {
            attention_pass<C, true>(p, w, st, smem, cur.n_tiles, cur.my_tiles, cur.q_row0, lane_p, acc, true, sink);
        }

        // Persistent grid: request the next unit's Q and tile 0 now, so their HBM round trip runs under
        // this unit's epilogue (Q's registers and the staging registers are dead once the pass is over).
        bool more = false;
        UnitCtx<C> nxt;
        if constexpr (C::PERSIST) {
            more = work_unit<C>(p, ++round, g, qb);
            if (more) {
                nxt.set(p, g, qb, wave);
                st.init(nxt.Kh, nxt.Vh, kSb, vSb, Sk, wave, lane, row_bytes);
                st.load_all(0);
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
        if constexpr (C::STAMP) t_ep0 = cycle_stamp();
        // The epilogue's ~40 per-lane addresses must be recomputed here: hoisted out of the unit loop they
        // would live across the tile loop, spill, and their reload (vmcnt(0)) would wait for the prefetch above.
        int lane_e = lane;
        if constexpr (C::PERSIST) asm volatile("" : "+v"(lane_e));
        if constexpr (sizeof(OutT) == 2) {
            // every pass ends behind a workgroup barrier, so no wave still reads the K/V ring: reuse it
            static_assert(256 * D * 2 <= C::LDS_BYTES, "epilogue regions must fit the ring");
            if (cur.wave_live && !sink.stored)
                w.template store_o_lds<OutT>(smem + wave * (WROWS * D * 2), cur.Oh, cur.lse_head, oSb, cur.q_row0, S, lane_e, orow_bytes);
        } else if constexpr (C::LDS_EPILOGUE32) {
            static_assert(256 * 64 * 4 <= C::LDS_BYTES, "epilogue regions must fit the ring");
            if (cur.wave_live && !sink.stored)
                w.template store_o_lds32<OutT>(smem + wave * (WROWS * 256), cur.Oh, cur.lse_head, oSb, cur.q_row0, S, lane_e, orow_bytes);
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // include the store tail
        acc[0] = cycle_stamp() - t_kernel0;                // whole workgroup lifetime of this wave
        if (lane == 0 && p.dbg) {
#pragma unroll
            for (int k = 0; k < 11; ++k) p.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + k] = acc[k];
            p.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + 11] = 1;   // (rows of waves that do not exist stay 0)
        }
    }
}

}  // namespace fa
