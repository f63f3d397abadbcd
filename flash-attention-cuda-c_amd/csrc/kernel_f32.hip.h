// kernel_f32.hip.h -- exact-fp32 forward kernel on the f32-input MFMA (d in {64,128}).
//
// The reference's tensors are `const float*` (kernels/FlashAttention.cuh:59-63) and its arithmetic is
// IEEE fp32 on CUDA cores (kernels/utils.cuh:23-31,102-111).  gfx950 has no TF32/xf32 path, but
// v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fp32 fmaf chain at the fp32 vector peak
// (64 FLOP/clk/SIMD = 157 TFLOP/s): this kernel keeps fp32 end to end -- Q, K, V, scores, P and O --
// with no rounding of P.  Same algorithm and orientation as the bf16 kernel:
//
//   workgroup = 4 waves = 128 query rows; a wave owns 32 rows; KV tile = 32 keys, 2 LDS buffers;
//   S^T = K.Q^T : lane (key, h) supplies A[key][k=h], lane (q, h) supplies B[k=h][q]  (one f32 each).
//       The contraction order is permuted so that one ds_read_b128 of the chunk-major K image
//       [d/4][32 keys][4 floats] feeds 4 MFMAs: step 4v+i uses d = 8v + 4h + i (Q is loaded to match).
//   P^T register r of lane (q, h) is key (r&3) + 8(r>>2) + 4h -- exactly the B operand of PV step r;
//   O^T += V^T.P^T: the V image is stored TRANSPOSED and padded, [d][32 keys + 4], so the 4 keys
//       8g+4h .. +3 of one d are one ds_read_b128 (rows of 144 B: conflict-free for 16 lanes).
//   Online softmax: running max with lazy rescale (threshold 2^8), exp2 with scale*log2(e) folded in.
// MFMA-bound by construction (256 MFMA-cycles of 64 per 32x32 score tile per k/d unit): the softmax
// VALU and the staging are a few per cent, so the loop is left to the compiler's scheduler.
#pragma once

#include "loaders.hip.h"

namespace fa {

// PAD_: the tensors' head dimension p.d is smaller than D_ (a multiple of 4): rows are zero-padded on the fly,
// exactly as in the bf16 kernel (loaders.hip.h: BufStage<..., PAD>).
template <int D_, bool CAUSAL_, typename OutT_, bool PAD_ = false>
struct F32Cfg {
    static constexpr bool PAD = PAD_;
    static constexpr int D = D_;
    static constexpr bool CAUSAL = CAUSAL_;
    using OutT = OutT_;
    static constexpr int KVBLK = 32, QBLK = 128, NWAVES = 4;
    static constexpr int KROW = D * 4;                     // bytes of one K / V row in global memory
    static constexpr int K_TILE = KVBLK * KROW;            // K image [D/4 chunks][32 keys][16 B]
    static constexpr int VT_ROW = (KVBLK + 4) * 4;         // V^T image row: 32 keys + 4 pad floats = 144 B
    static constexpr int V_TILE = D * VT_ROW;
    static constexpr int SLOT = K_TILE + V_TILE;
    static constexpr int LDS_BYTES = 2 * SLOT;
    static constexpr int LOADS = K_TILE / 16 / (64 * NWAVES);   // 16-byte loads per thread per tensor per tile
    static constexpr int QUARTERS = KROW / 128;                 // 128-byte pieces of a row
};

template <class C>
__global__ __launch_bounds__(256, 2) void fwd_f32_mfma_kernel(const Params p) {
    constexpr int D = C::D, KVBLK = C::KVBLK, QBLK = C::QBLK, DB = D / 32, NV = D / 8;   // NV: 16-byte K reads per tile
    constexpr bool CAUSAL = C::CAUSAL;
    using OutT = typename C::OutT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    lds_ptr smem = (lds_ptr)smem_raw;

    int g, qb;
    if (!unit_of_block(p, CAUSAL, g, qb)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = g / p.H, hd = g - b * p.H;
    const int S = p.S, Sk = p.Sk;
    const int h = lane >> 5, r31 = lane & 31;

    const char* Qh = (const char*)p.Q + (b * p.qB + hd * p.qH) * 4;
    const char* Kh = (const char*)p.K + (b * p.kB + hd * p.kH) * 4;
    const char* Vh = (const char*)p.V + (b * p.vB + hd * p.vH) * 4;
    char* Oh = (char*)p.O + (b * p.oB + hd * p.oH) * (int64_t)sizeof(OutT);
    const int64_t qSb = p.qS * 4, kSb = p.kS * 4, vSb = p.vS * 4, oSb = p.oS * (int64_t)sizeof(OutT);

    const int row_bytes = C::PAD ? p.d * 4 : D * 4, orow_bytes = C::PAD ? p.d * (int)sizeof(OutT) : D * (int)sizeof(OutT);
    const int q_row0 = qb * QBLK + wave * 32;
    const int q_end = min(S, (qb + 1) * QBLK);
    const int k_tiles = (Sk + KVBLK - 1) / KVBLK;
    const int n_tiles = CAUSAL ? min(k_tiles, (q_end + KVBLK - 1) / KVBLK) : k_tiles;
    const bool wave_live = q_row0 < S;
    const int my_tiles = !wave_live ? 0 : (CAUSAL ? min(n_tiles, (q_row0 + 31) / KVBLK + 1) : n_tiles);

    // Q fragments: qv[v] = Q[row][8v + 4h .. +3]  (B operands of steps 4v .. 4v+3)
    f32x4 qv[NV];
    {
        int row = q_row0 + r31;
        row = row < S ? row : S - 1;
        const char* src = Qh + row * qSb + h * 16;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if constexpr (C::PAD) {
                qv[v] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (v * 32 + h * 16 < row_bytes) qv[v] = *reinterpret_cast<const f32x4*>(src + v * 32);   // never read past the row
            } else {
                qv[v] = *reinterpret_cast<const f32x4*>(src + v * 32);
            }
        }
    }

    // staging: a wave-instruction = 8 keys x 128 bytes; wave w owns keys 8w..8w+7; load i = 128-byte piece i
    const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, (int)((Sk - 1) * kSb + row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, (int)((Sk - 1) * vSb + row_bytes), 0x00020000);
    const int skey = 8 * wave + (lane & 7), schunk = lane >> 3;      // 16-byte chunk within the 128-byte piece
    const int koff = skey * (int)kSb + schunk * 16, voff = skey * (int)vSb + schunk * 16;
    const int ktile = (int)(KVBLK * kSb), vtile = (int)(KVBLK * vSb);
    const int klds = schunk * (KVBLK * 16) + skey * 16;              // K image: chunk c at c*512, key at +16*key
    const int vlds = (4 * schunk) * C::VT_ROW + skey * 4;            // V^T image: float d at d*144, key at +4*key
    u32x4 kr[C::LOADS], vr[C::LOADS];
    auto stage_load = [&](int t) {
#pragma unroll
        for (int i = 0; i < C::LOADS; ++i) {
            kr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(krsrc, koff + t * ktile + i * 128, 0, 0));
            vr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrsrc, voff + t * vtile + i * 128, 0, 0));
        }
    };
    auto stage_write = [&](lds_ptr slot) {
#pragma unroll
        for (int i = 0; i < C::LOADS; ++i) {
            const bool ok = !C::PAD || (schunk + 8 * i) * 16 < row_bytes;             // chunk inside the (narrower) row?
            const u32x4 z = {0u, 0u, 0u, 0u};
            const u32x4 kk = ok ? kr[i] : z, vv = ok ? vr[i] : z;
            lds_write_b128(slot, klds + i * 8 * (KVBLK * 16), kk);                      // +8 chunks per 128-byte piece
#pragma unroll
            for (int j = 0; j < 4; ++j)                                                   // transpose: 4 floats -> 4 rows
                *reinterpret_cast<FA_LDS uint32_t*>(slot + C::K_TILE + vlds + (32 * i + j) * C::VT_ROW) = vv[j];
        }
    };

    f32x16 o[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) o[i][k] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float c = p.scale_log2;

    stage_load(0);
    stage_write(smem);
    __syncthreads();

    const int kread = h * (KVBLK * 16) + r31 * 16;                   // K chunk 2v+h of key r31: + v*1024
    const int vread = r31 * C::VT_ROW + h * 16;                      // V^T row d = 32db + r31, keys 8g+4h: + db*32*VT_ROW + g*32

    for (int t = 0; t < n_tiles; ++t) {
        lds_ptr cur = smem + (t & 1) * C::SLOT;
        const bool more = t + 1 < n_tiles;
        if (more) stage_load(t + 1);

        if (t < my_tiles) {
            const int kv0 = t * KVBLK;
            // ---- S^T = K.Q^T ----
            f32x16 s;
#pragma unroll
            for (int k = 0; k < 16; ++k) s[k] = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const f32x4 kf = *reinterpret_cast<FA_LDS const f32x4*>(cur + kread + v * (2 * KVBLK * 16));
#pragma unroll
                for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[i], qv[v][i], s, 0, 0, 0);
            }
            // ---- mask (diagonal / ragged tile) ----
            if ((CAUSAL && kv0 + KVBLK - 1 > q_row0) || kv0 + KVBLK > Sk) {
                const int qi = q_row0 + r31;
                const int lim = CAUSAL ? (qi < Sk - 1 ? qi : Sk - 1) : Sk - 1;
#pragma unroll
                for (int k = 0; k < 16; ++k) s[k] = (kv0 + acc_row(k, h)) > lim ? -INFINITY : s[k];
            }
            // ---- online softmax (lazy rescale) ----
            float mx = fmaxf(s[0], s[1]);
#pragma unroll
            for (int k = 2; k < 16; ++k) mx = fmaxf(mx, s[k]);
            mx = max_both_halves(mx) * c;
            if (__any(mx > m + 8.0f)) {
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);   // m = -inf on the first tile: alpha = 0, O and l are 0
                m = mn;
                l *= alpha;
#pragma unroll
                for (int i = 0; i < DB; ++i)
#pragma unroll
                    for (int k = 0; k < 16; ++k) o[i][k] *= alpha;
            }
            float pr[16];
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                pr[k] = fast_exp2(fmaf(s[k], c, -m));
                sum += pr[k];
            }
            l += sum;
            // ---- O^T += V^T.P^T : step r = 4g+i uses key 8g + 4h + i of this lane half ----
            lds_ptr vimg = cur + C::K_TILE;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 vf = *reinterpret_cast<FA_LDS const f32x4*>(vimg + vread + db * 32 * C::VT_ROW + gq * 32);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[i], pr[4 * gq + i], o[db], 0, 0, 0);
                }
        }

        if (more) stage_write(smem + ((t + 1) & 1) * C::SLOT);
        __syncthreads();
    }

    if (wave_live) {
        const float l_tot = sum_both_halves(l);
        const int qi = q_row0 + r31;
        if (p.lse && lane < 32 && qi < S) p.lse[(int64_t)g * S + qi] = (m + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
        const float inv = 1.0f / l_tot;
        if (qi < S) {
            char* dst = Oh + qi * oSb;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d0 = 32 * db + 8 * g4 + 4 * h;
                    if (C::PAD && d0 * (int)sizeof(OutT) >= orow_bytes) continue;       // columns past the real head dimension
                    const float a = o[db][4 * g4 + 0] * inv, bq = o[db][4 * g4 + 1] * inv;
                    const float c2 = o[db][4 * g4 + 2] * inv, e = o[db][4 * g4 + 3] * inv;
                    if constexpr (sizeof(OutT) == 4) {
                        f32x4 v = {a, bq, c2, e};
                        *reinterpret_cast<f32x4*>(dst + d0 * 4) = v;
                    } else if constexpr (__is_same(OutT, __bf16)) {
                        u32x2 v = {pack_bf16(a, bq), pack_bf16(c2, e)};
                        *reinterpret_cast<u32x2*>(dst + d0 * 2) = v;
                    } else {
                        u32x2 v = {pack_f16(a, bq), pack_f16(c2, e)};
                        *reinterpret_cast<u32x2*>(dst + d0 * 2) = v;
                    }
                }
        }
    }
}

}  // namespace fa
