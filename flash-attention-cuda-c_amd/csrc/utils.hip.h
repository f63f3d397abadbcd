// utils.hip.h -- fragment types, wave-64 cross-lane helpers and numeric conversions for gfx950.
//
// Counterpart of the reference's kernels/utils.cuh:10-15,49-55,84-90 (pipe_t / WARP / lane masks /
// cooperative-groups reductions).  Nothing of that file transfers: a CDNA4 wavefront is 64 lanes,
// there is no cuda::pipeline and no partial-warp shuffle mask.  Reductions here are in-register:
// with the swapped QK^T orientation (computers.hip.h) a score row lives in ONE lane pair
// (l, l+32), so the row max / row sum need a single v_permlane32_swap instead of the
// reference's cg::reduce + Bc-1 __shfl_down_sync chain (kernels/utils.cuh:66-73).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

constexpr int WAVE = 64;  // hard-coded: gfx950 wavefront (reference: #define WARP 32, utils.cuh:12)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) int i32x8;

#define FA_LDS __attribute__((address_space(3)))
typedef FA_LDS char* lds_ptr;

// ---- MFMA wrappers: D = A*B + C on one wave, 32x32 output tile, K = 16 ------------------------
// Operand lane maps (guide cdna_hip_programming.md section 3): lane l, r = l&31, h = l>>5 holds
// A[row r][k = 8h+j], B[k = 8h+j][col r], j = 0..7; C/D: col = l&31, row = (reg&3)+8*(reg>>2)+4h.
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// 16x16 output tile, K = 32 (computers16.hip.h).  Operand lane maps: lane l, r = l&15, h4 = l>>4 holds A[row r][k = 8*h4 + j],
// B[k = 8*h4 + j][col r], j = 0..7; C/D: col = l&15, row = 4*h4 + reg (reg 0..3).  Same FLOPs per cycle as the 32x32x16
// form, but half the accumulator traffic per FLOP: on random data the chip holds a ~10-25 % higher clock on it.
__device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_16x16x32(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// fp8 e4m3fn (OCP on gfx950) operands: 8 one-byte elements per lane, same (row, k = 8h+j) lane map,
// same rate as the bf16 instruction (the 2x rate needs the block-scaled f8f6f4 form).
__device__ __forceinline__ f32x16 mfma_32x32x16_fp8(uint64_t a, uint64_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8((long)a, (long)b, c, 0, 0, 0);
}

// Block-scaled form (MX): v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit scales (E8M0 byte 127 =
// 2^0) contracts 64 elements per instruction in 16 passes -- twice the rate of the 32x32x16 fp8 form.  A lane
// supplies 32 bytes per operand; the contraction pairs byte j of lane half h of A with byte j of lane half h of
// B, so any assignment of d indices to (h, j) works as long as Q and K use the same one.
__device__ __forceinline__ f32x16 mfma_32x32x64_fp8_unit_scale(u32x4 a_lo, u32x4 a_hi, u32x4 b_lo, u32x4 b_hi, f32x16 c) {
    const i32x8 a = {(int)a_lo[0], (int)a_lo[1], (int)a_lo[2], (int)a_lo[3], (int)a_hi[0], (int)a_hi[1], (int)a_hi[2], (int)a_hi[3]};
    const i32x8 b = {(int)b_lo[0], (int)b_lo[1], (int)b_lo[2], (int)b_lo[3], (int)b_hi[0], (int)b_hi[1], (int)b_hi[2], (int)b_hi[3]};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// Row of the 32x32 accumulator tile held in register `reg` of lane half `h`.
__host__ __device__ constexpr int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ---- LDS accessors ------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 lds_read_b128(lds_ptr base, int byte_off) {
    return *reinterpret_cast<FA_LDS const bf16x8*>(base + byte_off);
}
__device__ __forceinline__ void lds_write_b128(lds_ptr base, int byte_off, u32x4 v) {
    *reinterpret_cast<FA_LDS u32x4*>(base + byte_off) = v;
}
// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of 16-bit elements is returned
// column-major: lane i of the group gets column i, rows 0..3 in elements 0..3.  Lane 4q+p supplies
// the address of row q, columns 4p..4p+3 (8-byte aligned).  EXEC must be all ones.
__device__ __forceinline__ s16x4 lds_read_tr16_b64(lds_ptr base, int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<FA_LDS s16x4*>(base + byte_off));
}

// 16-byte global store with a cache policy (KernelCfg::O_CACHE): 0 plain; 1 sc1 (write-through: the line does not stay in
// the XCD's L2); 2 nt; 3 sc0 sc1.  The asm forms end in s_nop 1: hipcc must not reuse the data registers before the store has read them.
template <int POLICY>
__device__ __forceinline__ void store_global_b128(void* p, u32x4 v) {
    if constexpr (POLICY == 0) *reinterpret_cast<u32x4*>(p) = v;
    else if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// ---- cross-half exchange: value of lane l^32 ---------------------------------------------------
__device__ __forceinline__ float other_half(float x) {
    // v_permlane32_swap vdst, src swaps vdst[32..63] with src[0..31]; with both = x the pair
    // (r[0], r[1]) holds {own, partner} in some order on every lane.
    const uint32_t u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const bool lo = (threadIdx.x & 32) == 0;
    return __uint_as_float(lo ? r[1] : r[0]);
}
__device__ __forceinline__ float max_both_halves(float x) {
    const uint32_t u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// ---- reductions over the four 16-lane quarters (lanes l, l^16, l^32, l^48): a softmax row of the 16x16x32 kernels ----
// v_permlane16_swap vdst, src swaps vdst's odd 16-lane rows with src's even rows: with both = x the pair holds
// {x[row^1] on one side, own on the other}; v_permlane32_swap then does the same for the two 32-lane halves.
__device__ __forceinline__ float max_all_quarters(float x) {
    const uint32_t u = __float_as_uint(x);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const uint32_t v = __float_as_uint(y);
    auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float sum_all_quarters(float x) {
    const uint32_t u = __float_as_uint(x);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const uint32_t v = __float_as_uint(y);
    auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float sum_both_halves(float x) {
    const uint32_t u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ---- conversions ---------------------------------------------------------------------------------
// bf16 -> f32 is exact: place the 16 bits in the high half.
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }

// Plain casts: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) for adjacent pairs.
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t pack_f16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    f16x2 v;
    v[0] = (_Float16)lo;
    v[1] = (_Float16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// 8 bf16 -> 8 fp16 (the fp16-weights option: V is staged as fp16).  Exact for every bf16 value whose magnitude lies in fp16's
// normal range [2^-14, 65504] (8 significant bits fit 11); smaller magnitudes round to fp16 subnormals / zero (absolute error
// < 2^-25), larger ones become inf -- v_cvt_pk_f16_f32 rounds to nearest even and does not saturate.
__device__ __forceinline__ u32x4 bf16x8_to_f16x8(u32x4 v) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = pack_f16(bf16_lo(v[i]), bf16_hi(v[i]));
    return r;
}

// 8 fp8 e4m3fn bytes (two dwords) -> 8 bf16, exactly (e4m3fn has 3 mantissa bits): 4 v_cvt_pk_f32_fp8 +
// 4 v_cvt_pk_bf16_f32.
__device__ __forceinline__ u32x4 fp8x8_to_bf16x8(uint32_t lo, uint32_t hi) {
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true);
    const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true);
    return u32x4{pack_bf16(a[0], a[1]), pack_bf16(b[0], b[1]), pack_bf16(c[0], c[1]), pack_bf16(d[0], d[1])};
}

// s_memtime stamp for diagnostic builds: one asm statement with its own wait, fenced (guide: In-kernel stamps).
// the constant 100 MHz counter: with cycle_stamp() around the same stretch it gives the core clock the stretch ran at
__device__ __forceinline__ unsigned long long realtime_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__device__ __forceinline__ unsigned long long cycle_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

template <typename T> struct elem_traits;
template <> struct elem_traits<float> {
    __device__ static float load(const float* p) { return *p; }
    __device__ static void store(float* p, float v) { *p = v; }
};
template <> struct elem_traits<__bf16> {
    __device__ static float load(const __bf16* p) { return (float)*p; }
    __device__ static void store(__bf16* p, float v) { *p = (__bf16)v; }
};
template <> struct elem_traits<_Float16> {
    __device__ static float load(const _Float16* p) { return (float)*p; }
    __device__ static void store(_Float16* p, float v) { *p = (_Float16)v; }
};

}  // namespace fa
