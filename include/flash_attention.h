/*
 * flash_attention.h -- C ABI of the MI355X-native FlashAttention forward path.
 *
 * Drop-in boundary for the ONE hot path of GMichailov/Flash-Attention-CUDA-C: the fused
 * QK^T -> online softmax -> PV forward kernel
 *
 *     template<int D_HEAD,int Q_TILE_ROWS,int KV_TILE_ROWS> __global__
 *     void twoLoaderMhaFlashAttentionKernel(const float* Q, const float* K, const float* V,
 *                                           float* O, int batchSize, int numHeads, int seqLen,
 *                                           float scale, bool is_causal)
 *                                                   (reference kernels/FlashAttention.cuh:59-63)
 *
 * and its only host-side launcher test_flash_attention<...>() (reference tests/main.cu:21-103).
 * The reference has no symbol literally named flash_attention; BASELINE.json's north_star gives
 * that name to the launch signature above, so this library exports it with the kernel's
 * parameters in the kernel's order (Q,K,V,O,batchSize,numHeads,seqLen,...,scale,is_causal).
 * D_HEAD becomes a runtime argument; tile sizes are an internal policy (helpers.hpp), not ABI.
 *
 * Contract (same as the reference, tests/main.cu:39-48,60-64,99-102):
 *   - Q,K,V,O are DEVICE pointers to dense row-major [batchSize, numHeads, seqLen, dHead]
 *     tensors, element (b,h,s,j) at ((b*numHeads+h)*seqLen+s)*dHead+j
 *     (reference kernels/loaders.cuh:57,92).  Base pointers 16-byte aligned.
 *   - The caller owns all four buffers.  The library allocates nothing, frees nothing, keeps no
 *     global state, never synchronises the host and never prints: the call enqueues work on
 *     `stream` and returns (graph-capturable, re-entrant).  O is fully overwritten.
 *   - is_causal masks key k > query q (reference kernels/utils.cuh:43, tests/main.cu:81).
 *     Every query row keeps at least key 0, so no row is fully masked (the reference's NaN on
 *     fully-masked tiles, SURVEY.md defect D3, is not reproduced).
 *   - Each (b,h) pair is an independent problem (the reference mixes them: defect D2).
 *
 * Return value: 0 on success; > 0 a hipError_t from the launch; < 0 one of FA_ERR_* below.
 * The library never calls exit() (the reference's CUDA_CHECK does, tests/main.cu:12-19).
 *
 * There is NO CPU fallback: on a machine without a gfx950 device the call fails with a HIP error.
 */
#ifndef FLASH_ATTENTION_H
#define FLASH_ATTENTION_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Element types of Q/K/V (`dtype`) and of O (`o_dtype`). */
enum {
    FA_DTYPE_F32 = 0,      /* IEEE fp32 -- the reference's own type (const float*) */
    FA_DTYPE_BF16 = 1,     /* bfloat16, fp32 accumulation on MFMA */
    FA_DTYPE_FP8_E4M3 = 2, /* OCP e4m3fn (not fnuz); inputs only */
    FA_DTYPE_F16 = 3       /* IEEE fp16; output only */
};

/* Argument-validation errors (negative so they cannot collide with hipError_t). */
enum {
    FA_OK = 0,
    FA_ERR_NULL_POINTER = -1,
    FA_ERR_MISALIGNED = -2,       /* a base pointer is not 16-byte aligned */
    FA_ERR_BAD_SHAPE = -3,        /* batchSize/numHeads/seqLen/dHead <= 0 or too large */
    FA_ERR_UNSUPPORTED_DHEAD = -4,/* dHead not supported for this dtype */
    FA_ERR_UNSUPPORTED_DTYPE = -5,
    FA_ERR_BAD_SCALE = -6,        /* scale is NaN or infinite (fp8 inputs: or not positive) */
    FA_ERR_BAD_STRIDE = -7,
    FA_ERR_BAD_FLAGS = -8         /* flash_attention_ex: unknown flag, contradictory flags, or one that does not apply to this dtype / dHead */
};

/*
 * Precision of the softmax weights on the bf16 path (bf16 inputs, dHead 64 or 128).
 *
 * The weights P = exp(scale*S - max) are rounded before the P.V product: to bf16 (8 significant bits), or to fp16 (11 bits) with V
 * converted bf16 -> fp16 on its way into LDS.  The rounding errors of a row's weights average out over the keys that carry its
 * mass, so how close O comes to check.py (reference check.py:19-21) depends on the DATA, not only on the kernel:
 *
 *   tolerance stated by BASELINE.json: |O - ref| <= 1e-3 + 1e-3 |ref|, fp32 output, fraction of elements inside it
 *                                               bf16 weights   fp16 weights   default (flags = 0)
 *   N(0,1) Q, K, V, S = 4096, d = 128, no mask      100 %          100 %          100 %   (= bf16 weights: every row sees 4096 keys)
 *   same, causal                                   99.994 %        100 %          100 %   (the misses of bf16: rows that see few keys)
 *   Q, K x 3 (scores ~ N(0, 9^2): a SHARP softmax, a row's mass on a handful of keys), S = 4096, d = 128:
 *        no mask                                    97.9 %         100 %          97.9 %
 *        causal                                     88.3 %         100 %          90.9 %
 *   (measured: tests/test_flash_attention.py::test_parity_at_stated_tolerance_*, test_sharp_softmax_parity_is_what_it_measures;
 *    a float64 emulation of the bf16-weights arithmetic over 600 N(0,1) heads -- tests/micro/bf16_weight_error_by_row.py,
 *    profiles/r04_bf16_weight_error_by_row.txt -- puts the worst element of the rows that see 1024-1280 keys at 0.70-0.98 of the
 *    tolerance (two draws), of the rows that see 512-1024 keys at 1.07-1.31 x: hence FA_EARLY_KEYS, and its thin margin.)
 *
 *   default (flags = 0)    rows that can see fewer than FA_EARLY_KEYS keys take fp16 weights, all others bf16 weights: under the
 *                          causal mask the query rows q < FA_EARLY_KEYS of every head (whole query blocks; ONE kernel walks all
 *                          query blocks and runs each in the precision of its rows), and every row when seqLenK < FA_EARLY_KEYS.
 *                          Meets the stated tolerance on every element on N(0,1)-like data (the benchmark's); on data whose
 *                          softmax is sharp it is as accurate as bf16 weights are there (table above).  Costs ~0.5 % at seqLen 4096.
 *   FA_FLAG_F16_WEIGHTS    fp16 weights on every row (-7 % throughput; 8-13 x smaller errors): the choice for data with a sharp
 *                          softmax, or whenever the stated tolerance must hold whatever the data.
 *   FA_FLAG_BF16_WEIGHTS   bf16 weights on every row: the fastest form.
 *
 * Range of V.  Any finite bf16 V is valid input for every form (the reference's V is float: kernels/FlashAttention.cuh:60).  fp16
 * holds |v| <= 65504; a unit whose fp16-weights passes come out non-finite (a larger |v| is inf in fp16, and 0 * inf = NaN would
 * even reach rows that do not see that key) is repeated with bf16 weights and bf16 V -- so beyond 65504 the fp16 forms are as
 * accurate as FA_FLAG_BF16_WEIGHTS, never inf / NaN where the exact result is finite.  All forms accumulate un-normalised sums of
 * up to 2^8 x |v| per key in fp32: |V| up to 2^95 (4e28) is safe for any seqLen.
 * Zero-padded head dimensions (dHead not 64 / 128) have no fp16-weights kernel: their rows that see few keys keep bf16 weights.
 * Other dHead, fp8 and fp32 inputs have one form each: the two flags are rejected (FA_ERR_BAD_FLAGS) where they cannot apply,
 * except FA_FLAG_BF16_WEIGHTS on any bf16 problem.
 */
#define FA_EARLY_KEYS 1024
enum {
    FA_FLAG_F16_WEIGHTS = 1,
    FA_FLAG_BF16_WEIGHTS = 2
};

/*
 * flash_attention -- replaces the <<<grid,block,smem>>> launch of
 * twoLoaderMhaFlashAttentionKernel at reference tests/main.cu:60-61.
 *
 *   Q,K,V,O     device pointers, dense [batchSize,numHeads,seqLen,dHead]
 *   dHead       reference template parameter D_HEAD (kernels/FlashAttention.cuh:59)
 *   scale       multiplies QK^T before the softmax; the reference passes 1/sqrt(dHead)
 *               (tests/main.cu:27, check.py:19)
 *   dtype       element type of Q,K,V (FA_DTYPE_*)
 *   o_dtype     element type of O; FA_DTYPE_F32 matches the reference's float* O
 *   stream      hipStream_t (passed as void* so this header needs no HIP include); NULL = the
 *               default stream
 *
 * Supported: f32 inputs, any dHead <= 256 with 16-byte rows, any seqLen (exact fp32: dHead <= 128 on the
 *            f32-input MFMA -- 64 and 128 natively, other multiples of 4 zero-padded on the fly -- larger dHead on
 *            the generic VALU kernel);
 *            bf16 inputs, any dHead <= 128 that is a multiple of 8 on the MFMA path (64 and 128 natively, the
 *            others on the next larger instantiation with rows zero-padded on the fly; any seqLen >= 1), larger
 *            dHead <= 256 on the generic path; fp8 e4m3fn inputs, dHead <= 128 in multiples of 16 (QK^T on the
 *            block-scaled MFMA; dHead < 128 zero-padded on the fly).
 *            One head's K/V extent (seqLen x row stride) must stay below 2^31 bytes on the MFMA paths.
 */
int flash_attention(const void* Q, const void* K, const void* V, void* O,
                    int batchSize, int numHeads, int seqLen, int dHead,
                    float scale, bool is_causal,
                    int dtype, int o_dtype, void* stream);

/*
 * flash_attention_strided -- same path for tensors that are views of a (B,S,H*d_k) model-layout
 * buffer (reference check.py:14-16,24) or any other layout whose last dimension is contiguous.
 * Strides are in ELEMENTS: element (b,h,s,j) of X lives at b*strideB + h*strideH + s*strideS + j.
 * This is the strided API the reference sketched and left commented out
 * (kernels/FlashAttention.cuh:22-27: strideBatch / strideHead per tensor).
 * Row starts must stay 16-byte aligned (strides multiples of 16 bytes).
 */
typedef struct fa_strides {
    int64_t strideB, strideH, strideS;
} fa_strides;

int flash_attention_strided(const void* Q, const void* K, const void* V, void* O,
                            int batchSize, int numHeads, int seqLen, int dHead,
                            float scale, bool is_causal, int dtype, int o_dtype,
                            const fa_strides* sQ, const fa_strides* sK, const fa_strides* sV,
                            const fa_strides* sO, void* stream);

/*
 * flash_attention_lse -- flash_attention() that also returns the log-sum-exp of every softmax row:
 *     LSE[b,h,q] = ln( sum over visible keys k of exp(scale * <Q[b,h,q], K[b,h,k]>) )      (natural log)
 * in a dense fp32 [batchSize, numHeads, seqLen] device buffer (16-byte aligned).  This is the L / M
 * statistic of the reference's commented-out first API (kernels/FlashAttention.cuh:21,
 * archive/archive.cu:34-42,201-204) and what a backward pass, split-KV or ring composition needs.
 * LSE may be NULL (then identical to flash_attention()).
 * O with and without an LSE request: the kernels that return the LSE normalise by the fp32 sum of the UNROUNDED softmax weights
 * (so that the LSE is exact to fp32 rounding); bf16 inputs without the causal mask and without an LSE request normalise by the
 * sum of the ROUNDED weights instead (it comes out of the matrix cores with the P.V product).  The two differ by the weights'
 * rounding averaged over a row: at most one ulp of a bf16 output, <= 2^-9 relative in fp32.  A caller that compares a sharded
 * run with a whole one bit for bit asks for the LSE on BOTH sides: a shard small enough for the 128-row pair kernel always takes
 * the fp32-sum normaliser, which the whole problem's persistent kernel takes only with an LSE request
 * (tests/test_flash_attention.py::test_small_noncausal_shard_against_the_whole_problem; under the causal mask both sides always
 * sum in fp32, and shards of equal kernel choice -- BASELINE cfg4's 256-head slabs -- are bit for bit either way).
 */
int flash_attention_lse(const void* Q, const void* K, const void* V, void* O, float* LSE,
                        int batchSize, int numHeads, int seqLen, int dHead,
                        float scale, bool is_causal, int dtype, int o_dtype, void* stream);

/*
 * flash_attention_cross -- the full argument list of the reference's first (commented-out) API,
 * kernels/FlashAttention.cuh:18-28: separate seqLenQ / seqLenK, the L/M statistic, per-tensor strides.
 *   Q, O   [batchSize, numHeads, seqLenQ, dHead]      K, V   [batchSize, numHeads, seqLenK, dHead]
 *   LSE    fp32 [batchSize, numHeads, seqLenQ] or NULL;   sQ..sO  element strides or NULL (dense)
 * Cross-attention, decode against a longer key/value cache (seqLenQ < seqLenK) and chunked prefill all
 * go through here.  is_causal keeps the reference's predicate on ABSOLUTE row indices -- key k is masked
 * when k > q (kernels/utils.cuh:43) -- i.e. the mask is top-left aligned; query q sees keys
 * 0..min(q, seqLenK-1).  (A caller that wants the last query aligned with the last key offsets its K/V
 * view or runs non-causal over the prefix it may see.)
 */
int flash_attention_cross(const void* Q, const void* K, const void* V, void* O, float* LSE,
                          int batchSize, int numHeads, int seqLenQ, int seqLenK, int dHead,
                          float scale, bool is_causal, int dtype, int o_dtype,
                          const fa_strides* sQ, const fa_strides* sK, const fa_strides* sV,
                          const fa_strides* sO, void* stream);

/*
 * flash_attention_ex -- flash_attention_cross() plus option flags (FA_FLAG_*); flags = 0 is flash_attention_cross().
 * flash_attention(), _lse(), _strided(), _cross() and _sharded() all run with flags = 0.
 */
int flash_attention_ex(const void* Q, const void* K, const void* V, void* O, float* LSE,
                       int batchSize, int numHeads, int seqLenQ, int seqLenK, int dHead,
                       float scale, bool is_causal, int dtype, int o_dtype,
                       const fa_strides* sQ, const fa_strides* sK, const fa_strides* sV,
                       const fa_strides* sO, unsigned flags, void* stream);

/*
 * flash_attention_weights -- the attention matrix the reference's oracle returns next to its output
 * (check.py:20,25 `attn`, printed by its demo at :42).  The fused kernel never stores it; this call
 * rebuilds it from Q, K and the LSE a flash_attention_lse / flash_attention_cross call produced:
 *     P[b,h,q,k] = exp(scale * <Q[b,h,q], K[b,h,k]> - LSE[b,h,q]),   0 where is_causal hides k > q
 * P is a dense fp32 [batchSize, numHeads, seqLenQ, seqLenK] device buffer (mind its size: this is an
 * inspection path for small seqLen).  Any dHead <= 256 with 16-byte rows; sQ / sK as above or NULL.
 */
int flash_attention_weights(const void* Q, const void* K, const float* LSE, float* P,
                            int batchSize, int numHeads, int seqLenQ, int seqLenK, int dHead,
                            float scale, bool is_causal, int dtype,
                            const fa_strides* sQ, const fa_strides* sK, void* stream);

/*
 * Multi-GPU: every (b,h) pair is an independent problem, so the forward pass shards
 * over the flattened head index g = b*numHeads + h with no data-path collective.
 *
 * flash_attention_shard_range -- rank `rank` of `nRanks` owns heads [*lo, *hi): contiguous ranges that tile
 * [0, totalHeads) exactly, sizes differing by at most one.  Returns 0 or FA_ERR_BAD_SHAPE.
 *
 * flash_attention_sharded -- one host thread drives nDevices devices: device deviceIds[r] holds, as its own
 * dense [hi-lo, seqLen, dHead] slabs Q[r], K[r], V[r], O[r], the head range of rank r, and gets the same
 * kernel enqueued on streams[r] (NULL array or NULL entry = that device's default stream).  Asynchronous
 * like flash_attention(); the caller's current device is restored.  Returns the first error, else 0.
 * (One process per GPU -- bench.py under torchrun -- just calls flash_attention() on its own slab.)
 */
int flash_attention_shard_range(int totalHeads, int rank, int nRanks, int* lo, int* hi);

int flash_attention_sharded(int nDevices, const int* deviceIds,
                            const void* const* Q, const void* const* K, const void* const* V, void* const* O,
                            int batchSize, int numHeads, int seqLen, int dHead,
                            float scale, bool is_causal, int dtype, int o_dtype, void* const* streams);

/*
 * Launch-geometry policy -- the counterpart of the reference's helpers.hpp:8-36
 * (calculateSizeBlockQ / calculateSizeBlockKV / getNumCta, which return constants there).
 * Fills the tile sizes and grid the library will use for this problem; returns 0 or FA_ERR_*.
 */
typedef struct fa_launch_plan {
    int q_block_rows;    /* Br: query rows per workgroup          (helpers.hpp:8-19); 256 on the MFMA paths' persistent kernels, 128 (and
                            threads = 256) for small bf16 problems at dHead 64 / 128: the pair kernel */
    int kv_block_rows;   /* Bc: keys per inner-loop tile           (helpers.hpp:21-30) */
    int threads;         /* threads per workgroup                  (tests/main.cu:52)  */
    int grid;            /* number of workgroups                   (helpers.hpp:33-36) */
    int lds_bytes;       /* dynamic LDS per workgroup              (tests/main.cu:55)  */
    int kernel_id;       /* which internal kernel: 0 generic fp32 VALU, 1 bf16 MFMA, 2 fp8 MFMA, 3 exact-fp32 MFMA */
} fa_launch_plan;

int flash_attention_plan(int batchSize, int numHeads, int seqLen, int dHead, bool is_causal,
                         int dtype, int o_dtype, fa_launch_plan* plan);

/*
 * flash_attention_plan_ex -- what a flash_attention_ex() call with these arguments launches.  A bf16 problem may be split in two
 * ranges of query blocks (see "Precision of the softmax weights"): `early` describes the fp16-weights kernel over the first
 * early->q_blocks query blocks of every head, `main` the bf16-weights kernel over the remaining main->q_blocks; a range that
 * does not exist has q_blocks = 0 and grid = 0.  When both exist they run in ONE launch of one kernel (every workgroup walks its
 * share of the list of all query blocks; a unit runs in the precision of its block): both descriptions then carry that launch's
 * grid and LDS size.  lds_bytes is the launched
 * instantiation's own figure (it depends on the engine, the staging form and the output type).  flash_attention_plan() is
 * this call with seqLenK = seqLen, flags = FA_FLAG_BF16_WEIGHTS (one range) and only `main` returned.  Either pointer may be NULL.
 */
typedef struct fa_launch_plan_ex {
    fa_launch_plan launch;
    int q_blocks;        /* query blocks of every head this range covers */
    int first_q_block;   /* ... starting at this one */
    int unit_lists;      /* 1 = both ranges run in ONE launch of one kernel that walks ONE (head, query block) list over all query blocks,
                            every unit in the precision of its range; 0 = this is the only range (or the pair kernel's launch) */
} fa_launch_plan_ex;

int flash_attention_plan_ex(int batchSize, int numHeads, int seqLenQ, int seqLenK, int dHead, bool is_causal,
                            int dtype, int o_dtype, unsigned flags, fa_launch_plan_ex* early, fa_launch_plan_ex* main);

/* Human-readable text for a return code of the functions above (static storage). */
const char* flash_attention_error_string(int code);

/* Library version, e.g. "fa-mi355x 0.1 (gfx950)". */
const char* flash_attention_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FLASH_ATTENTION_H */
