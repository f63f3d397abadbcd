// loaders.hip.h -- HBM -> registers -> LDS staging of K/V tiles, Q fragment loads, O epilogue.
//
// Counterpart of the reference's kernels/loaders.cuh: the LDS carve-up (:23-52), asyncBufferLoad
// (:55-83), asyncWriteO (:85-112) and the two loader warps (:114-203).  Re-designed for CDNA4:
//   * no dedicated loader warps and no cuda::pipeline: every wave stages 1/8 of each tile with
//     issue-early / write-late register staging (global_load_dwordx4 before the tile's compute,
//     ds_write_b128 after it), so HBM/L2 latency hides under the MFMA phases;
//   * each wave-instruction fetches 8 rows x 128 contiguous bytes (full cache lines) -- the
//     reference's lane-contiguous fragments (loaders.cuh:57) are 32 rows x frag*4 B per request;
//   * Q is never staged in LDS: each lane loads the MFMA B-fragments of its own query row once;
//   * the K image is chunk-major  [d/8][key][8 x bf16]  so the 32 lanes of a half-wave read 512
//     contiguous bytes per ds_read_b128 (conflict-free without an XOR swizzle, immediates only);
//   * the V image is [key/8][d/32][key%8][d%32] so that ds_read_b64_tr_b16 (hardware transpose)
//     feeds V^T straight into the PV MFMA: each half-wave reads 256 contiguous bytes.
#pragma once

#include "utils.hip.h"

namespace fa {

// Kernel arguments.  Strides are in elements; the last dimension is contiguous.
struct Params {
    const void* Q;
    const void* K;
    const void* V;
    void* O;
    int64_t qB, qH, qS;
    int64_t kB, kH, kS;
    int64_t vB, vH, vS;
    int64_t oB, oH, oS;
    int B, H, S;
    int nQ;           // query blocks per head
    int units;        // B*H*nQ
    int cpx;          // ceil(units / 8): work units per XCD group
    float scale_log2; // scale * log2(e)
    float scale;
    unsigned long long* dbg;  // diagnostic builds only (tests/fa_tune): per-wave segment cycle sums
};

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

// ------------------------------------------------------------------------------------------------
// K/V tile staging for the bf16 MFMA kernel: KVBLK = 64 keys, 512 threads.
// ------------------------------------------------------------------------------------------------
template <int D>
struct KVStage {
    static constexpr int KVBLK = 64;
    static constexpr int ROW_BYTES = D * 2;
    static constexpr int TILE_BYTES = KVBLK * ROW_BYTES;       // 16 KiB (D=128) / 8 KiB (D=64)
    static constexpr int CPT = TILE_BYTES / 16 / 512;          // 16-B chunks per thread: 2 / 1
    static constexpr int DB = D / 32;

    u32x4 k[CPT];
    u32x4 v[CPT];

    // lane -> (key, chunk) of wave-instruction i.  One instruction = 8 keys x 128 B.
    __device__ static int k_key(int wave, int lane) { return 8 * wave + (lane & 7); }
    __device__ static int k_chunk(int i, int lane) { return 8 * i + (lane >> 3); }
    __device__ static int v_key(int wave, int lane) { return 8 * wave + 2 * ((lane >> 3) & 3) + ((lane >> 2) & 1); }
    __device__ static int v_chunk(int i, int lane) { return 8 * i + 4 * (lane >> 5) + (lane & 3); }

    // LDS byte offsets inside one K / V tile image.
    __host__ __device__ static constexpr int k_lds_off(int key, int chunk) { return chunk * (KVBLK * 16) + key * 16; }
    __host__ __device__ static constexpr int v_lds_off(int key, int chunk) {
        return (key >> 3) * (DB * 512) + (chunk >> 2) * 512 + (key & 7) * 64 + (chunk & 3) * 16;
    }

    // Issue the global loads of the tile starting at key row kv0 (rows clamped to S-1: pad,
    // don't mask -- the padded keys are masked to -inf in the softmax).
    __device__ __forceinline__ void load(const char* Kh, const char* Vh, int64_t kS_bytes,
                                         int64_t vS_bytes, int kv0, int S, int wave, int lane) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            int kr = kv0 + k_key(wave, lane);
            kr = kr < S ? kr : S - 1;
            k[i] = *reinterpret_cast<const u32x4*>(Kh + kr * kS_bytes + k_chunk(i, lane) * 16);
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            int vr = kv0 + v_key(wave, lane);
            vr = vr < S ? vr : S - 1;
            v[i] = *reinterpret_cast<const u32x4*>(Vh + vr * vS_bytes + v_chunk(i, lane) * 16);
        }
    }

    // Write the staged registers into the LDS images (ds_write_b128, 128 contiguous bytes per
    // 8-lane group -> conflict-free).
    __device__ __forceinline__ void write(lds_ptr kimg, lds_ptr vimg, int wave, int lane) const {
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            lds_write_b128(kimg, k_lds_off(k_key(wave, lane), k_chunk(i, lane)), k[i]);
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            lds_write_b128(vimg, v_lds_off(v_key(wave, lane), v_chunk(i, lane)), v[i]);
    }
};

// Per-lane LDS read bases (everything else is an immediate offset).
//   K A-fragment of k-step ks, key tile kt:  k_read_base + ks*2048 + kt*512
//   V^T A-fragment of d-block db, 16-key step s4, half jj: v_read_base + (2*s4+jj)*DB*512 + db*512
__device__ __forceinline__ int k_read_base(int lane) { return (lane >> 5) * 1024 + (lane & 31) * 16; }
__device__ __forceinline__ int v_read_base(int lane) {
    const int h = lane >> 5, q = (lane & 15) >> 2, p = lane & 3, g = (lane >> 4) & 1;
    return 256 * h + 64 * q + 32 * g + 8 * p;
}

}  // namespace fa
