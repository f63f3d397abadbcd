// loaders.hip.h -- staging of K/V tiles from HBM into LDS, and the LDS images they land in.
//
// Counterpart of the reference's kernels/loaders.cuh: the LDS carve-up (:23-52), asyncBufferLoad
// (:55-83), asyncWriteO (:85-112) and the two loader warps (:114-203).  Re-designed for CDNA4:
//   * no dedicated loader warps and no cuda::pipeline: every wave stages 1/8 of each tile, the requests issued inside the
//     MFMA slots of the tile two iterations earlier (computers*.hip.h), so HBM/L2 latency hides under compute;
//   * production form (bf16, unpadded rows; fp8: K only): LDS-DMA -- `buffer_load_dwordx4 ... lds` writes the tile straight
//     into LDS, no staging registers and no ds_write (DmaStage, HybridStageFp8 at the end of this file);
//   * register form (BufStage: padded head dimensions, fp8 V, the fp16-weights option -- whatever transforms the tile on its
//     way): buffer loads into registers, ds_write_b128 inside later slots;
//   * either way tiles are fetched through BUFFER descriptors: per-head descriptor in SGPRs, a per-lane byte offset that
//     never changes, the tile offset one scalar-operand add -- no per-tile address arithmetic, and rows past the end of the
//     sequence arrive as 0 through the hardware range check (no clamping);
//   * each wave-instruction fetches 8 rows x 128 contiguous bytes (full cache lines) -- the reference's lane-contiguous
//     fragments (loaders.cuh:57) are 32 rows x frag*4 B per request;
//   * Q is never staged through the K/V ring: each lane ends up with the MFMA B-fragments of its own query row, once;
//   * K image, register form: chunk-major [row bytes/16][64 keys][16 B], so the 32 lanes of a half-wave read 512 contiguous
//     bytes per ds_read_b128; LDS-DMA form: [key/8][chunk slot][key%8][16 B] (what a linear 1-KiB DMA piece can write while
//     reading whole 128-byte lines), with the 32x32x16 engine's chunk slots of odd 8-key blocks swapped pairwise -- both
//     conflict-free without a per-read XOR, one per-lane base + immediates;
//   * the V image is bf16 [key/8][d/32][key%8][d%32] (32x32x16 engine) or [key/8][d/16][key%8][d%16] (16x16x32 engine) so that
//     ds_read_b64_tr_b16 (hardware transpose) feeds V^T straight into the PV MFMA: each half-wave reads 256 contiguous
//     bytes.  fp8 inputs are widened to bf16 (exactly) on their way into this image.
#pragma once

#include "utils.hip.h"

namespace fa {

// Kernel arguments.  Strides are in elements; the last dimension is contiguous.
struct Params {
    const void* Q;
    const void* K;
    const void* V;
    void* O;
    float* lse;       // optional [B,H,S] fp32 log-sum-exp (natural log) of every softmax row, or nullptr
    int64_t qB, qH, qS;
    int64_t kB, kH, kS;
    int64_t vB, vH, vS;
    int64_t oB, oH, oS;
    int B, H, S;      // S = query rows per head
    int Sk;           // keys (rows of K and V) per head; == S for self-attention
    int d;            // head dimension of the tensors (<= the kernel's compile-time D: narrower rows are zero-padded)
    int nQ;           // query blocks per head handled by THIS launch: blocks qb0 .. qb0 + nQ - 1 (a call may be split into two launches)
    int qb0;          // first of them
    int units;        // B*H*nQ
    int cpx;          // ceil(units / 8): work units per XCD group
    int jpx;          // persistent grid only: workgroups per XCD group (grid / 8)
    int hp;           // mixed-precision kernels (KernelCfg::MIX): the query blocks qb < hp of every head take fp16 softmax weights
    float scale_log2; // scale * log2(e)
    float scale;
    unsigned long long* dbg;  // diagnostic builds only (tests/fa_tune): per-wave segment cycle sums
};

// The unit list a workgroup walks (kernel_bf16.hip.h: work_unit): the query blocks qb0 .. qb0 + nQ - 1 of every head.  The single
// kernels take it from Params; the launch that mixes two configurations takes two of them beside ONE Params.
struct UnitList {
    int nQ, qb0, units, cpx, jpx;
};
__host__ __device__ inline UnitList unit_list_of(const Params& p) { return UnitList{p.nQ, p.qb0, p.units, p.cpx, p.jpx}; }

// Workgroup -> (head, query block).  Blocks b and b+8 share an XCD (round-robin dispatch), so
// giving each XCD group a CONTIGUOUS range of units keeps all query blocks of a head -- which
// re-read the same K/V -- on one XCD's L2.  Pure speed choice: any placement is correct.
__device__ __forceinline__ bool unit_of_block(const Params& p, bool causal, int& g, int& qb) {
    const int bid = blockIdx.x;
    const int u = (bid & 7) * p.cpx + (bid >> 3);
    if (u >= p.units) return false;
    g = u / p.nQ;
    qb = u - g * p.nQ;
    if (causal) qb = p.nQ - 1 - qb;  // heaviest query blocks of a head first
    return true;
}

// Geometry of one 64-key tile.  ESZ = bytes per input element (2: bf16, 1: fp8 e4m3fn).
template <int D, int ESZ>
struct TileGeom {
    static constexpr int KVBLK = 64;
    static constexpr int DB = D / 32;
    static constexpr int ROWB = D * ESZ;                 // bytes of one K or V row in global memory
    static constexpr int K_TILE = KVBLK * ROWB;          // K image keeps the input element type
    static constexpr int V_TILE = KVBLK * D * 2;         // V image is always bf16
    static constexpr int SLOT = K_TILE + V_TILE;         // ring slot = [K image | V image]
    static constexpr int LOADS = ROWB / 128;             // 16-byte loads per thread per tensor per tile (512 threads)
    static_assert(ROWB == 128 || ROWB == 256, "tile rows are one or two 128-byte lines");

    // LDS byte offsets inside the images.  K: 16-byte chunk c of key k.  V: bf16 chunk c (8 elements) of key k.
    __host__ __device__ static constexpr int k_lds_off(int key, int chunk) { return chunk * (KVBLK * 16) + key * 16; }
    __host__ __device__ static constexpr int v_lds_off(int key, int chunk) {
        return (key >> 3) * (DB * 512) + (chunk >> 2) * 512 + (key & 7) * 64 + (chunk & 3) * 16;
    }
    // V image of the 16x16x32 kernels: bf16 [key/8][d/16][key%8][d%16] -- one (8 keys x 16 d) subtile is 256
    // contiguous bytes, exactly what one half-wave of a ds_read_b64_tr_b16 of the 16-wide V^T fragment touches.
    static constexpr int DG = D / 16;
    __host__ __device__ static constexpr int v16_lds_off(int key, int chunk) {
        return (key >> 3) * (DG * 256) + (chunk >> 1) * 256 + (key & 7) * 32 + (chunk & 1) * 16;
    }
    // K image of the LDS-DMA staging (DmaStage below): [key/8][chunk slot][key%8][16 B]; with kxor (32x32x16 engine) slot c' of an odd
    // 8-key block holds chunk c' ^ 1
    static constexpr int KBLK = (ROWB / 16) * 128;
    __host__ __device__ static constexpr int kd_lds_off(int key, int chunk, bool kxor) {
        return (key >> 3) * KBLK + (chunk ^ (kxor ? (key >> 3) & 1 : 0)) * 128 + (key & 7) * 16;
    }
};

// Per-lane LDS read bases (everything else is an immediate offset).
//   K A-fragment: 16-byte chunk (2*s + h) of key 32*kt + (lane&31):  k_read_base + s*2048 + kt*512
//   V^T A-fragment of d-block db, 16-key step s4, half jj: v_read_base + (2*s4+jj)*DB*512 + db*512
__device__ __forceinline__ int k_read_base(int lane) { return (lane >> 5) * 1024 + (lane & 31) * 16; }
__device__ __forceinline__ int v_read_base(int lane) {
    const int h = lane >> 5, q = (lane & 15) >> 2, p = lane & 3, g = (lane >> 4) & 1;
    return 256 * h + 64 * q + 32 * g + 8 * p;
}
// 16x16x32 kernels (computers16.hip.h).  K A-fragment of 16-key group kg, 32-wide k-step ks: lane (key r = l&15, quarter
// h4 = l>>4) reads 16-byte chunk 4*ks + h4 of key 16*kg + r:  k16_read_base + ks*4096 + kg*256.
// V^T A-fragment of 16-wide d group dg, 32-key step kk, half jj: the 16 lanes of quarter h4 transpose-read keys
// 32*kk + 16*jj + 4*h4 + (0..3) x d 16*dg .. +15:  v16_read_base + (4*kk + 2*jj)*DG*256 + dg*256.
__device__ __forceinline__ int k16_read_base(int lane) { return (lane >> 4) * 1024 + (lane & 15) * 16; }
template <int D>
__device__ __forceinline__ int v16_read_base(int lane) {
    const int h4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    return (h4 >> 1) * (D / 16 * 256) + (4 * (h4 & 1) + q) * 32 + 8 * p;
}

// K/V tile staging.  One wave-instruction = 8 keys x 128 bytes.  A 64-key tile is 8 key groups x
// (ROWB/128) column halves; with NWAVES waves each wave owns GPW = 8/NWAVES consecutive key groups.
//   K lanes: key 8g + (l&7),                    16-byte chunk (l>>3)          [+8 for the second 128-byte half]
//   V lanes: key 8g + 2*((l>>3)&3) + ((l>>2)&1), 16-byte chunk 4*(l>>5)+(l&3) [+8 ...]
// chosen so that each 8-lane ds_write_b128 group writes 128 contiguous LDS bytes in the respective image.
// PAD: the tensors' rows hold fewer than D elements (row_bytes < ROWB): chunks past the row end are zeroed on
// their way into LDS (the buffer read itself lands in the next row, or past the extent where it returns 0).
// V16: the V image of the 16x16x32 kernels (TileGeom::v16_lds_off); its lanes are
//   V lanes: key 8g + 4*((l>>3)&1) + ((l&7)>>1), 16-byte chunk 2*(l>>4) + (l&1)  [+8 for the second 128-byte half]
// (each 8-lane group still writes 128 contiguous LDS bytes; +8 keys and +8 chunks cost the same strides as above).
// VF16: bf16 V is converted to fp16 on its way into the V image (the fp16-weights option of the 16x16x32 engine).
template <int D, int ESZ, int NWAVES = 8, bool PAD = false, bool V16 = false, bool VF16 = false>
struct BufStage {
    using G = TileGeom<D, ESZ>;
    static constexpr int HALVES = G::ROWB / 128;                 // 128-byte halves of a row
    static constexpr int GPW = 8 / NWAVES;                       // key groups per wave
    static constexpr int LOADS = HALVES * GPW;                   // 16-byte loads per thread per tensor per tile
    static constexpr int NL = 2 * LOADS;                         // loads per thread per tile
    static constexpr int VW = ESZ == 1 ? 2 : 1;                  // ds_write_b128 per V load (fp8 widens to bf16)
    static constexpr int NW = LOADS + LOADS * VW;                // LDS writes per thread per tile
    static constexpr bool K_DMA = false;                         // (the K image is the chunk-major one)
    static_assert(8 % NWAVES == 0, "NWAVES must divide the 8 key groups of a tile");
    __amdgpu_buffer_rsrc_t krsrc, vrsrc;
    int koff, voff;        // per-lane byte offset of load 0 inside a tile (constant)
    int klds, vlds;        // per-lane LDS byte offset of write 0 inside the K / V image
    int ktile, vtile;      // bytes per 64-key tile step (scalar)
    int kgrp, vgrp;        // bytes per 8-key group step in global memory (scalar)
    u32x4 r[NL];           // staged data: [0,LOADS) = K, [LOADS,NL) = V
    bool kok[HALVES], vok[HALVES];   // PAD only: this lane's chunk of each 128-byte half lies inside the row

    __device__ __forceinline__ void init(const char* Kh, const char* Vh, int64_t kS_bytes, int64_t vS_bytes, int S,
                                         int wave, int lane, int row_bytes = G::ROWB) {
        // descriptor inputs are blockIdx / kernarg derived -> wave-uniform; num_records = the head's extent, to
        // the last byte of its last row (a strided view's rows are followed by other heads' data, or by nothing)
        krsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, (int)((S - 1) * kS_bytes + row_bytes), 0x00020000);
        vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, (int)((S - 1) * vS_bytes + row_bytes), 0x00020000);
        ktile = (int)(64 * kS_bytes);
        vtile = (int)(64 * vS_bytes);
        kgrp = (int)(8 * kS_bytes);
        vgrp = (int)(8 * vS_bytes);
        const int g0 = wave * GPW;
        const int kk = 8 * g0 + (lane & 7), kc = lane >> 3;
        const int vk = V16 ? 8 * g0 + 4 * ((lane >> 3) & 1) + ((lane & 7) >> 1) : 8 * g0 + 2 * ((lane >> 3) & 3) + ((lane >> 2) & 1);
        const int vc = V16 ? 2 * (lane >> 4) + (lane & 1) : 4 * (lane >> 5) + (lane & 3);
        koff = kk * (int)kS_bytes + kc * 16;
        voff = vk * (int)vS_bytes + vc * 16;
        klds = G::k_lds_off(kk, kc);
        static_assert(!(V16 && ESZ == 1), "the 16x16x32 V image takes bf16 inputs");
        vlds = V16 ? G::v16_lds_off(vk, vc) : G::v_lds_off(vk, ESZ == 1 ? 2 * vc : vc);   // fp8: 16 input bytes = bf16 chunks 2c, 2c+1
        if constexpr (PAD) {
#pragma unroll
            for (int hf = 0; hf < HALVES; ++hf) {
                kok[hf] = (kc + 8 * hf) * 16 < row_bytes;
                vok[hf] = (vc + 8 * hf) * 16 < row_bytes;
            }
        }
    }
    __device__ __forceinline__ static u32x4 keep_if(bool ok, u32x4 v) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        return ok ? v : z;
    }
    // load #N of tile t (N < LOADS: K, else V): key group g0 + n/HALVES, 128-byte half n%HALVES.  The tile
    // offset goes into the VGPR offset (v_add with scalar operands) so the hardware range check covers it.
    template <int N>
    __device__ __forceinline__ void load(int t) { load_to<N>(r, t); }
    template <int N>
    __device__ __forceinline__ void load_to(u32x4 (&dst)[NL], int t) const {
        constexpr int n = N < LOADS ? N : N - LOADS;
        constexpr int gi = n / HALVES, hf = n % HALVES;
        if constexpr (N < LOADS)
            dst[N] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(krsrc, koff + t * ktile + gi * kgrp + hf * 128, 0, 0));
        else
            dst[N] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrsrc, voff + t * vtile + gi * vgrp + hf * 128, 0, 0));
    }
    // LDS write #N: the K writes, then the V writes.  K image: +8 keys = +128 B, +8 chunks = +8 KiB.
    // V image: +8 keys = +DB*512 B, +8 bf16 chunks = +1 KiB.
    // F16: convert bf16 V to fp16 on the way (the class's VF16 unless the caller says otherwise: the fp16-weights kernels' bf16-weights
    // fallback pass, computers16.hip.h, writes the same image unconverted)
    template <int N, bool F16 = VF16>
    __device__ __forceinline__ void write(lds_ptr slot_base) const { write_from<N, F16>(r, slot_base); }
    template <int N, bool F16 = VF16>
    __device__ __forceinline__ void write_from(const u32x4 (&r)[NL], lds_ptr slot_base) const {
        if constexpr (N < LOADS) {
            constexpr int gi = N / HALVES, hf = N % HALVES;
            lds_write_b128(slot_base, klds + gi * 128 + hf * 8192, PAD ? keep_if(kok[hf], r[N]) : r[N]);
        } else if constexpr (ESZ == 2) {
            constexpr int n = N - LOADS, gi = n / HALVES, hf = n % HALVES;
            const u32x4 v = PAD ? keep_if(vok[hf], r[N]) : r[N];
            lds_write_b128(slot_base + G::K_TILE, vlds + gi * (G::DB * 512) + hf * 1024, F16 ? bf16x8_to_f16x8(v) : v);
        } else {
            // fp8 V (ROWB = 128, one half): 16 e4m3fn bytes -> 16 bf16 (exact), two adjacent 16-byte chunks
            constexpr int n = (N - LOADS) / 2, W = (N - LOADS) % 2;   // load n of this tensor, low / high 8 bytes
            const u32x4 src = PAD ? keep_if(vok[0], r[LOADS + n]) : r[LOADS + n];   // e4m3fn 0x00 = +0 -> bf16 +0
            lds_write_b128(slot_base + G::K_TILE, vlds + n * (G::DB * 512) + W * 16, fp8x8_to_bf16x8(src[2 * W], src[2 * W + 1]));
        }
    }
    template <int N = 0> __device__ __forceinline__ void load_all_to(u32x4 (&dst)[NL], int t) const { if constexpr (N < NL) { load_to<N>(dst, t); load_all_to<N + 1>(dst, t); } }
    template <int N = 0> __device__ __forceinline__ void write_all_from(const u32x4 (&src)[NL], lds_ptr s) const { if constexpr (N < NW) { write_from<N>(src, s); write_all_from<N + 1>(src, s); } }
    template <int N = 0> __device__ __forceinline__ void load_all(int t) { if constexpr (N < NL) { load<N>(t); load_all<N + 1>(t); } }
    template <int N = 0, bool F16 = VF16> __device__ __forceinline__ void write_all(lds_ptr s) const { if constexpr (N < NW) { write<N, F16>(s); write_all<N + 1, F16>(s); } }
    // (interface shared with DmaStage, whose loads need their LDS destination)
    __device__ __forceinline__ void set_dst(lds_ptr) {}
    __device__ __forceinline__ void load_all_into(int t, lds_ptr) { load_all(t); }
    __device__ __forceinline__ void wait_all() const {}
};

// ---- LDS-DMA staging (Opt::dma; bf16, unpadded rows, 8 waves) -------------------------------------------------------------
// K/V tiles go global -> LDS by `buffer_load_dwordx4 ... lds`: no staging registers, no ds_write.  One wave-instruction writes 1 KiB
// of LDS linearly (lane l -> M0 base + 16 l) from per-lane SOURCE addresses, so the images must be cut into 1-KiB pieces whose lane
// order still reads whole 128-byte lines of global memory:
//   K image (DMA form):  [key/8][16-byte chunk c'][key%8][16 B]  (8-key block = ROWB/16 chunks x 128 B), where slot c' of a block
//       holds chunk c' ^ (block & 1) for the 32x32x16 engine (KXOR) and chunk c' for the 16x16x32 engine: with that, the 16-lane groups
//       of both engines' ds_read_b128 fragment reads cover all 64 banks (kd_read_base / kd16_read_base; immediates only, as before).
//       Piece (g, j) = chunks 8j .. 8j+7 of key group g: lane l <- key 8g + (l&7), chunk 8j + ((l>>3) ^ parity) -- the same lane order as
//       the register path's K loads (8 lines of 128 B per instruction, each fully used).
//   V images unchanged (both are made of 512-byte / 256-byte subtiles that are contiguous per 8 keys):
//       32x32x16 image: piece (g, j) = d-blocks 2j, 2j+1 of key group g: lane l <- key 8g + ((l>>2)&7), chunk 8j + 4(l>>5) + (l&3)
//       16x16x32 image: piece (g, j) = d-groups 4j .. 4j+3:              lane l <- key 8g + ((l>>1)&7), chunk 8j + 2(l>>4) + (l&1)
// Rows past the head's extent fail the buffer range check and arrive as zeros, as on the register path.
__device__ __forceinline__ int kd_read_base(int lane, int blk_bytes) {    // 32x32x16 engine: + kt*(4*blk) + u*256
    const int r = lane & 31, h = lane >> 5;
    return (r >> 3) * blk_bytes + ((h ^ ((r >> 3) & 1)) * 128) + (r & 7) * 16;
}
__device__ __forceinline__ int kd16_read_base(int lane, int blk_bytes) {  // 16x16x32 engine: + kg*(2*blk) + ks*512
    const int r = lane & 15, h4 = lane >> 4;
    return (r >> 3) * blk_bytes + h4 * 128 + (r & 7) * 16;
}
template <int D, int NWAVES, bool V16, bool KXOR>
struct DmaStage {
    using G = TileGeom<D, 2>;
    static constexpr int HALVES = G::ROWB / 128;
    static constexpr int GPW = 8 / NWAVES;                        // 8-key groups per wave (8 waves: 1; 4 waves: 2)
    static_assert(GPW == 1 || GPW == 2, "DMA staging: 8 or 4 waves");
    static constexpr int LOADS = HALVES * GPW;                    // 1-KiB pieces per wave per tensor per tile
    static constexpr int NL = 2 * LOADS, NW = 0;
    static constexpr int KBLK = G::KBLK;                          // bytes of one 8-key block of the K image
    static constexpr bool K_DMA = true;
    static constexpr int VBLK = V16 ? G::DG * 256 : G::DB * 512;  // ... of the V image
    u32x4 krsrc, vrsrc;    // raw buffer descriptors (stride 0; word 3 as __builtin_amdgcn_make_buffer_rsrc(..., 0x00020000)) the loads go through
    int koff[GPW], voff;   // per-lane source byte offset of piece 0 of key group gi inside a tile (K: the slot XOR depends on the group's parity)
    int ktile, vtile;      // bytes per 64-key tile step (scalar)
    int kgrp, vgrp;        // bytes per 8-key group step in global memory (scalar)
    int kdst, vdst;        // this wave's first block inside the K / V image (scalar)
    uint32_t dst;          // LDS byte address of the ring slot the next loads go to (scalar)
    __device__ __forceinline__ static u32x4 descriptor(uint64_t a, uint32_t bytes) {
        return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    }
    __device__ __forceinline__ void init(const char* Kh, const char* Vh, int64_t kS_bytes, int64_t vS_bytes, int S, int wave, int lane,
                                         int row_bytes = D * 2) {
        krsrc = descriptor((uint64_t)Kh, (uint32_t)((S - 1) * kS_bytes + row_bytes));
        vrsrc = descriptor((uint64_t)Vh, (uint32_t)((S - 1) * vS_bytes + row_bytes));
        ktile = (int)(64 * kS_bytes);
        vtile = (int)(64 * vS_bytes);
        kgrp = (int)(8 * kS_bytes);
        vgrp = (int)(8 * vS_bytes);
        const int g0 = wave * GPW;
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi)
            koff[gi] = (8 * (g0 + gi) + (lane & 7)) * (int)kS_bytes + ((lane >> 3) ^ (KXOR ? ((g0 + gi) & 1) : 0)) * 16;
        voff = V16 ? (8 * g0 + ((lane >> 1) & 7)) * (int)vS_bytes + (2 * (lane >> 4) + (lane & 1)) * 16
                   : (8 * g0 + ((lane >> 2) & 7)) * (int)vS_bytes + (4 * (lane >> 5) + (lane & 3)) * 16;
        kdst = g0 * KBLK;
        vdst = G::K_TILE + g0 * VBLK;
    }
    __device__ __forceinline__ void set_dst(lds_ptr slot) { dst = (uint32_t)(uintptr_t)slot; }
    // The DMA is issued from inline asm: hipcc then keeps no account of it -- issued through the builtin, every ds_read_b64_tr_b16 that
    // follows waits vmcnt(0) for it (the V^T reads of the SAME iteration), because the waitcnt pass cannot tell the ring slots apart.
    // Ordering is by hand instead: wait_all() before the barrier that publishes the tile.  M0 (the LDS destination) is written in the
    // statement that uses it and restored.  The instruction offset stays 0 (it would be added to the LDS address as well); the tile
    // offset goes into the VGPR offset so that the range check covers it.
    __device__ __forceinline__ static void dma16(const u32x4& rsrc, uint32_t lds_byte, int voffset) {
        // (nothing else in these kernels lives in M0: it is written here, in the statement that uses it, and not restored --
        //  two scalar instructions fewer per piece in the wave's serial instruction stream; the clobber tells hipcc)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voffset), "s"(rsrc), "s"(lds_byte) : "memory", "m0");
#pragma clang diagnostic pop
    }
    // piece #N of tile t (N < LOADS: K, else V) -> ring slot `dst`: key group gi = n / HALVES of this wave, 1-KiB piece j = n % HALVES
    template <int N>
    __device__ __forceinline__ void load(int t) const {
        constexpr int n = N < LOADS ? N : N - LOADS, gi = n / HALVES, j = n % HALVES;
        if constexpr (N < LOADS) dma16(krsrc, dst + kdst + gi * KBLK + j * 1024, koff[gi] + t * ktile + j * 128);
        else dma16(vrsrc, dst + vdst + gi * VBLK + j * 1024, voff + t * vtile + gi * vgrp + j * 128);
    }
    // every piece this wave has issued has landed in LDS (then a barrier publishes it to the other waves)
    __device__ __forceinline__ void wait_all() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    template <int N>
    __device__ __forceinline__ void write(lds_ptr) const {}
    template <int N = 0> __device__ __forceinline__ void load_all(int t) const { if constexpr (N < NL) { load<N>(t); load_all<N + 1>(t); } }
    __device__ __forceinline__ void load_all_into(int t, lds_ptr slot) { set_dst(slot); load_all(t); }
    __device__ __forceinline__ void write_all(lds_ptr) const {}
};

// ---- bf16 inputs, both weight precisions in one kernel (KernelCfg::MIX: the causal default) ------------------------------------------------
// K always by LDS-DMA.  V by LDS-DMA for the units that run with bf16 weights; for the units that run with fp16 weights V has to become
// fp16 on its way (the P.V MFMA then is v_mfma_f32_32x32x16_f16), so it goes through registers as in BufStage: buffer load, convert,
// ds_write_b128 into the SAME V image.  Which form a tile takes is a template argument of every call (F16W), i.e. a property of the pass the
// unit runs, not of the kernel: one workgroup walks units of both kinds in list order, and the next unit's tile 0 is requested in ITS form
// while the current unit's epilogue runs.  As in HybridStageFp8 the K DMA pieces of a tile are issued before its V register loads, so that
// hipcc's counted vmcnt for the V data never waits on a younger DMA.
// V16 / KXOR: the engine's V image and K slot swap (32x32x16: false / true; 16x16x32: true / false), as in DmaStage.
template <int D, int NWAVES, bool V16 = false, bool KXOR = true>
struct MixStage {
    using G = TileGeom<D, 2>;
    using Dma = DmaStage<D, NWAVES, V16, KXOR>;
    using VPath = BufStage<D, 2, NWAVES, false, V16, true>;     // (its K half stays unused)
    static constexpr int LOADS = Dma::LOADS;                     // pieces (DMA) or 16-byte loads per lane (registers) per tensor per tile
    static constexpr int NL = 2 * LOADS, NW = LOADS;             // NW: ds_write_b128 per lane per tile, fp16 form only
    static constexpr int KBLK = G::KBLK;
    static constexpr bool K_DMA = true;
    static_assert(VPath::LOADS == LOADS, "both V paths cut a tile the same way");
    Dma d;
    VPath v;
    __device__ __forceinline__ void init(const char* Kh, const char* Vh, int64_t kS_bytes, int64_t vS_bytes, int S, int wave, int lane,
                                         int row_bytes = D * 2) {
        d.init(Kh, Vh, kS_bytes, vS_bytes, S, wave, lane, row_bytes);
        v.init(Kh, Vh, kS_bytes, vS_bytes, S, wave, lane, row_bytes);
    }
    __device__ __forceinline__ void set_dst(lds_ptr slot) { d.set_dst(slot); }
    template <int N, bool F16W>
    __device__ __forceinline__ void load(int t) {
        if constexpr (N < LOADS || !F16W) d.template load<N>(t);
        else v.template load<N>(t);                              // (BufStage numbers its V loads LOADS .. 2 LOADS - 1 too)
    }
    template <int N, bool F16W>
    __device__ __forceinline__ void write(lds_ptr slot_base) const {
        if constexpr (F16W) v.template write<LOADS + N, true>(slot_base);
    }
    __device__ __forceinline__ void wait_all() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    template <bool F16W, int N = 0> __device__ __forceinline__ void load_all(int t) { if constexpr (N < NL) { load<N, F16W>(t); load_all<F16W, N + 1>(t); } }
    template <bool F16W> __device__ __forceinline__ void load_all_into(int t, lds_ptr slot) { set_dst(slot); load_all<F16W>(t); }
    template <bool F16W, int N = 0> __device__ __forceinline__ void write_all(lds_ptr s) const { if constexpr (N < NW) { write<N, F16W>(s); write_all<F16W, N + 1>(s); } }
};

// ---- fp8 inputs (Opt::dma, unpadded rows, 8 waves): K by LDS-DMA, V through registers -------------------------------------------
// K rows are 128 bytes of e4m3: the tile's K image (8 KiB, DMA form with the 32x32x16 engine's slot swap) is eight 1-KiB pieces, one per
// wave.  V has to be widened to bf16 between the load and the LDS write, so it keeps BufStage's register path (1 load, 2 ds_write_b128).
// The DMA is issued BEFORE the V load of the same tile: hipcc's counted vmcnt for the V data then never waits on a younger DMA.
template <int D, int NWAVES>
struct HybridStageFp8 {
    using G = TileGeom<D, 1>;
    using VPath = BufStage<D, 1, NWAVES, false>;
    static_assert(NWAVES == 8 && G::ROWB == 128, "fp8 K by DMA: 8 waves, 128-byte rows");
    static constexpr int LOADS = 1, NL = 2, NW = 2;
    static constexpr int KBLK = G::KBLK;
    static constexpr bool K_DMA = true;
    VPath v;               // (its K half stays unused: only load<1> / write<1>, write<2> are called)
    u32x4 krsrc;
    int koff, ktile, kdst;
    uint32_t dst;
    __device__ __forceinline__ void init(const char* Kh, const char* Vh, int64_t kS_bytes, int64_t vS_bytes, int S, int wave, int lane,
                                         int row_bytes = D) {
        v.init(Kh, Vh, kS_bytes, vS_bytes, S, wave, lane, row_bytes);
        const uint64_t a = (uint64_t)Kh;
        krsrc = u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, (uint32_t)((S - 1) * kS_bytes + row_bytes), 0x00020000u};
        ktile = (int)(64 * kS_bytes);
        koff = (8 * wave + (lane & 7)) * (int)kS_bytes + ((lane >> 3) ^ (wave & 1)) * 16;
        kdst = wave * KBLK;
    }
    __device__ __forceinline__ void set_dst(lds_ptr slot) { dst = (uint32_t)(uintptr_t)slot; }
    template <int N>
    __device__ __forceinline__ void load(int t) {
        if constexpr (N == 0) DmaStage<128, 8, false, true>::dma16(krsrc, dst + kdst, koff + t * ktile);
        else v.template load<1>(t);
    }
    template <int N>
    __device__ __forceinline__ void write(lds_ptr slot_base) const { v.template write<N + 1>(slot_base); }
    __device__ __forceinline__ void wait_all() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ void load_all(int t) { load<0>(t); load<1>(t); }
    __device__ __forceinline__ void load_all_into(int t, lds_ptr slot) { set_dst(slot); load_all(t); }
    __device__ __forceinline__ void write_all(lds_ptr s) const { write<0>(s); write<1>(s); }
};

}  // namespace fa
