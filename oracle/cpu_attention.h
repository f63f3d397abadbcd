/*
 * oracle/cpu_attention.h -- CPU restatement of the reference's attention math.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into it, and only as the
 * checker (or the timed CPU baseline), never as the thing shipped.  The product path
 * (libflash_attention.so) has no CPU fallback and never links this file.
 *
 * Parity pin: this restatement is checked against golden vectors minted in the build container
 * by importing the reference's own check.py (tests/golden/make_golden.py -> the .npy files under tests/golden),
 * and against the reference's two known-answer cases (tests/main.cu:24-36,107 all-ones S=16 d=16
 * -> O == 1; check.py:30-43 all-ones (1,4,8) H=2 -> attn == 0.25, output == 1).
 *
 * What it follows in /root/reference:
 *   check.py:14-16   (B,S,H*d_k) -> view (B,S,H,d_k) -> transpose (B,H,S,d_k)
 *   check.py:19      scores = Q K^T / sqrt(d_k)
 *   check.py:20      attn = softmax(scores, dim=-1)          (max-subtracted, as F.softmax)
 *   check.py:21      output = attn V
 *   check.py:24      (B,H,S,d_k) -> transpose -> (B,S,H*d_k)
 *   tests/main.cu:74-91   the reference's own 3-loop CPU check on one [S,d] head
 *   tests/main.cu:81      causal rule: key index k > query index q is masked
 *   kernels/utils.cuh:43  same rule on absolute rows (top-left aligned)
 *   kernels/FlashAttention.cuh:59-63  dense [B,H,S,d] row-major operands, scale passed in
 */
#ifndef ORACLE_CPU_ATTENTION_H
#define ORACLE_CPU_ATTENTION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Dense [B,H,S,d] attention, fp32 storage, fp32 arithmetic (float accumulators, expf).
 * Stable softmax (row max subtracted).  causal != 0 masks key k > query q.
 * Threads: OpenMP over (b*h, query row); nthreads <= 0 means "all cores".
 * Returns the number of threads actually used. */
int oracle_attention_f32(const float* Q, const float* K, const float* V, float* O,
                         int batchSize, int numHeads, int seqLen, int dHead,
                         float scale, int causal, int nthreads);

/* Same contract, but every accumulation (dot products, row sum, PV) in double and exp() in
 * double.  This is the tight reference used to measure the GPU kernel's error. */
int oracle_attention_f64acc(const float* Q, const float* K, const float* V, float* O,
                            int batchSize, int numHeads, int seqLen, int dHead,
                            float scale, int causal, int nthreads);

/* Only rows [row0,row1) of heads [head0,head1) (flattened b*H+h) -- for sampled checks at the
 * large BASELINE configs.  O has the full [B*H,S,d] shape; untouched rows are left alone. */
int oracle_attention_f64acc_rows(const float* Q, const float* K, const float* V, float* O,
                                 int numBH, int seqLen, int dHead, float scale, int causal,
                                 int head0, int head1, int row0, int row1, int nthreads);

/* Literal restatement of tests/main.cu:74-91 for ONE head [S,d]: un-stabilised expf(dot*scale),
 * causal by dot = -1e9f BEFORE scaling, single thread.  Kept to show the stable form above
 * agrees with the reference's own check wherever that check does not overflow. */
void oracle_attention_maincu_single_head(const float* Q, const float* K, const float* V,
                                         float* O, int seqLen, int dHead, float scale,
                                         int causal);

/* check.py:4-25 in its own layout: Q,K,V (B,S,d_model) -> output (B,S,d_model) and, when
 * attn != NULL, attn (B,H,S,S).  scale is 1/sqrt(d_model/H) as check.py:19. */
int oracle_multi_head_attention(const float* Q, const float* K, const float* V,
                                float* output, float* attn,
                                int batch, int seqLen, int dModel, int numHeads, int nthreads);

/* Round-to-nearest-even conversions used to mint low-precision inputs (NaN-preserving). */
uint16_t oracle_f32_to_bf16(float x);
float    oracle_bf16_to_f32(uint16_t h);
uint8_t  oracle_f32_to_e4m3fn(float x);   /* OCP e4m3fn, saturating to +-448 */
float    oracle_e4m3fn_to_f32(uint8_t b);
void     oracle_round_to_bf16(float* x, int64_t n);     /* in place: x <- bf16(x) */
void     oracle_round_to_e4m3fn(float* x, int64_t n);   /* in place: x <- e4m3fn(x) */

#ifdef __cplusplus
}
#endif
#endif
