// helpers.hpp -- launch-geometry policy for gfx950.
//
// Counterpart of the reference's helpers.hpp:8-36.  There, calculateSizeBlockQ / calculateSizeBlockKV
// sketch formulas in comments (Br <= regs/(d*4), helpers.hpp:9-14; Bc <= L2/(2*d*4), :22-25) and then
// both `return 64` (:18,:29) -- sizes the reference kernel cannot even launch with.
// Here the sizes are what the MI355X kernels are built around:
//
//   Br (query rows / workgroup): 8 waves x 32 rows = 256 for the MFMA kernels.  32 rows per wave
//      are two 16-row query groups of the 16x16x32 engine (bf16) or one 32x32 tile of the 32x32x16
//      engine (fp8); 8 waves = 2 per SIMD share every K/V tile staged in LDS, which halves L2->LDS
//      traffic per FLOP against a 4-wave workgroup.  Register budget per lane at 2 waves/SIMD is
//      256: O^T accumulators d/2, Q fragments d/4, scores 2 x 32 (ping-pong), P 16, K / V^T fragment
//      windows 28, staging 16 -- about 250 at d = 128.
//   Bc (keys / tile): 64.  Two 32-key score tiles per wave; K + V tile = 32 KiB at d = 128,
//      a 3-slot ring = 96 KiB of the CU's 160 KiB LDS.
//   Grid: persistent, one workgroup per CU walking ceil(units / CUs) units (kernel_bf16.hip.h: work_unit).
//   Exception, decided per problem in csrc/FlashAttention.hip (pair_kernel_applies) because it depends on the shape, not on d and
//      dtype alone: a small bf16 problem (causal at d = 64: at most one 256-row unit per CU; otherwise per two CUs) runs Br = 128
//      (4 waves), one unit per workgroup -- at d = 64 two workgroups per CU paired heaviest + lightest (kernel_bf16.hip.h:
//      fwd_mfma_pair_kernel); flash_attention_plan() reports that.
//   Head dimensions other than 64 / 128 (<= 128; multiples of 8 for bf16, 16 for fp8, 4 for fp32) run the next
//      larger instantiation with their rows zero-padded on the fly.
//   exact-fp32 MFMA kernel (fp32 inputs, d <= 128): Br = 128 (4 waves x 32 rows, 2 workgroups per CU),
//      Bc = 32 (K image 16 KiB + transposed V image 18 KiB per buffer, two buffers).
//   generic exact-fp32 kernel: Br = Bc = 32 (VALU path, 128 < d <= 256, any seqLen).
#pragma once

#include "../include/flash_attention.h"

// bf16 head dimensions the MFMA kernel is instantiated for; other multiples of 8 up to 128 run the next larger one
// with their rows zero-padded on the fly.
inline bool bf16MfmaDHead(int d_head) { return d_head > 0 && d_head <= 128 && d_head % 8 == 0; }
inline int paddedDHead(int d_head) { return d_head <= 64 ? 64 : 128; }

inline int calculateSizeBlockQ(int d_head, int dtype) {
    if (dtype == FA_DTYPE_BF16 && bf16MfmaDHead(d_head)) return 256;
    if (dtype == FA_DTYPE_FP8_E4M3 && d_head <= 128) return 256;
    if (dtype == FA_DTYPE_F32 && d_head > 0 && d_head <= 128 && d_head % 4 == 0) return 128;   // exact-fp32 MFMA kernel: 4 waves x 32 rows
    return 32;
}

inline int calculateSizeBlockKV(int d_head, int dtype) {
    if (dtype == FA_DTYPE_BF16 && bf16MfmaDHead(d_head)) return 64;
    if (dtype == FA_DTYPE_FP8_E4M3 && d_head <= 128) return 64;
    return 32;
}

// Number of query blocks.  The reference asserts q_dim % q_block_size == 0 (helpers.hpp:34); the
// MI355X kernels handle a ragged last block, so this is a ceiling division.
inline int getNumCta(int q_dim, int q_block_size) { return (q_dim + q_block_size - 1) / q_block_size; }
