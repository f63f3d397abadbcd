/*
 * oracle/cpu_attention.c -- CPU restatement of the reference's attention math (plain C, gcc).
 *
 * TEST INFRASTRUCTURE ONLY -- see cpu_attention.h for the rules and the reference file:line
 * each function follows.  Pinned by tests/test_oracle.py against tests/golden/ (vectors minted
 * from the reference's check.py) and the reference's all-ones known-answer cases.
 */
#include "cpu_attention.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int pick_threads(int nthreads) {
#ifdef _OPENMP
    int maxt = omp_get_num_procs();
    if (nthreads <= 0 || nthreads > maxt) nthreads = maxt;
    return nthreads;
#else
    (void)nthreads;
    return 1;
#endif
}

/* ---- one query row, float arithmetic (check.py:19-21 / tests/main.cu:76-90 loop structure) ---- */
static void row_f32(const float* q, const float* Kh, const float* Vh, float* o, float* attn_row,
                    int S, int d, float scale, int nkeys, float* sc) {
    float m = -INFINITY;
    for (int k = 0; k < nkeys; ++k) {                 /* tests/main.cu:77-80 */
        float dot = 0.f;
        const float* kr = Kh + (int64_t)k * d;
        for (int j = 0; j < d; ++j) dot += q[j] * kr[j];
        sc[k] = dot * scale;                          /* check.py:19 */
        if (sc[k] > m) m = sc[k];
    }
    float sum = 0.f;
    for (int k = 0; k < nkeys; ++k) {                 /* check.py:20, max-subtracted */
        sc[k] = expf(sc[k] - m);
        sum += sc[k];
    }
    for (int j = 0; j < d; ++j) o[j] = 0.f;
    const float inv = 1.0f / sum;
    for (int k = 0; k < nkeys; ++k) {                 /* check.py:21 / tests/main.cu:84-90 */
        const float w = sc[k] * inv;
        if (attn_row) attn_row[k] = w;
        const float* vr = Vh + (int64_t)k * d;
        for (int j = 0; j < d; ++j) o[j] += w * vr[j];
    }
    if (attn_row) for (int k = nkeys; k < S; ++k) attn_row[k] = 0.f;
}

/* ---- one query row, double accumulation ---- */
static void row_f64(const float* q, const float* Kh, const float* Vh, float* o,
                    int d, float scale, int nkeys, double* sc, double* acc) {
    double m = -INFINITY;
    for (int k = 0; k < nkeys; ++k) {
        double dot = 0.0;
        const float* kr = Kh + (int64_t)k * d;
        for (int j = 0; j < d; ++j) dot += (double)q[j] * (double)kr[j];
        sc[k] = dot * (double)scale;
        if (sc[k] > m) m = sc[k];
    }
    double sum = 0.0;
    for (int k = 0; k < nkeys; ++k) {
        sc[k] = exp(sc[k] - m);
        sum += sc[k];
    }
    for (int j = 0; j < d; ++j) acc[j] = 0.0;
    for (int k = 0; k < nkeys; ++k) {
        const double w = sc[k];
        const float* vr = Vh + (int64_t)k * d;
        for (int j = 0; j < d; ++j) acc[j] += w * (double)vr[j];
    }
    for (int j = 0; j < d; ++j) o[j] = (float)(acc[j] / sum);
}

int oracle_attention_f32(const float* Q, const float* K, const float* V, float* O,
                         int batchSize, int numHeads, int seqLen, int dHead,
                         float scale, int causal, int nthreads) {
    const int64_t BH = (int64_t)batchSize * numHeads;
    const int64_t rows = BH * seqLen;
    nthreads = pick_threads(nthreads);
#pragma omp parallel num_threads(nthreads)
    {
        float* sc = (float*)malloc(sizeof(float) * (size_t)(seqLen > 0 ? seqLen : 1));
#pragma omp for schedule(dynamic, 16)
        for (int64_t r = 0; r < rows; ++r) {
            const int64_t g = r / seqLen;             /* flattened (b,h): kernels/loaders.cuh:57 layout */
            const int qi = (int)(r % seqLen);
            const float* Kh = K + g * seqLen * dHead;
            const float* Vh = V + g * seqLen * dHead;
            const int nkeys = causal ? qi + 1 : seqLen;   /* tests/main.cu:81: k > q masked */
            row_f32(Q + r * dHead, Kh, Vh, O + r * dHead, NULL, seqLen, dHead, scale, nkeys, sc);
        }
        free(sc);
    }
    return nthreads;
}

int oracle_attention_f64acc_rows(const float* Q, const float* K, const float* V, float* O,
                                 int numBH, int seqLen, int dHead, float scale, int causal,
                                 int head0, int head1, int row0, int row1, int nthreads) {
    if (head0 < 0) head0 = 0;
    if (head1 > numBH) head1 = numBH;
    if (row0 < 0) row0 = 0;
    if (row1 > seqLen) row1 = seqLen;
    if (head1 <= head0 || row1 <= row0) return 0;
    const int64_t nr = (int64_t)(head1 - head0) * (row1 - row0);
    nthreads = pick_threads(nthreads);
#pragma omp parallel num_threads(nthreads)
    {
        double* sc = (double*)malloc(sizeof(double) * (size_t)seqLen);
        double* acc = (double*)malloc(sizeof(double) * (size_t)dHead);
#pragma omp for schedule(dynamic, 8)
        for (int64_t i = 0; i < nr; ++i) {
            const int64_t g = head0 + i / (row1 - row0);
            const int qi = row0 + (int)(i % (row1 - row0));
            const int64_t r = g * seqLen + qi;
            const int nkeys = causal ? qi + 1 : seqLen;
            row_f64(Q + r * dHead, K + g * seqLen * dHead, V + g * seqLen * dHead,
                    O + r * dHead, dHead, scale, nkeys, sc, acc);
        }
        free(sc);
        free(acc);
    }
    return nthreads;
}

int oracle_attention_f64acc(const float* Q, const float* K, const float* V, float* O,
                            int batchSize, int numHeads, int seqLen, int dHead,
                            float scale, int causal, int nthreads) {
    return oracle_attention_f64acc_rows(Q, K, V, O, batchSize * numHeads, seqLen, dHead, scale,
                                        causal, 0, batchSize * numHeads, 0, seqLen, nthreads);
}

void oracle_attention_maincu_single_head(const float* Q, const float* K, const float* V,
                                         float* O, int seqLen, int dHead, float scale,
                                         int causal) {
    float* scores = (float*)malloc(sizeof(float) * (size_t)seqLen);
    for (int q = 0; q < seqLen; ++q) {                       /* tests/main.cu:75 */
        for (int k = 0; k < seqLen; ++k) {                   /* :77 */
            float dot = 0.f;
            for (int j = 0; j < dHead; ++j) dot += Q[q * dHead + j] * K[k * dHead + j];  /* :80 */
            if (causal && k > q) dot = -1e9f;                /* :81 */
            scores[k] = expf(dot * scale);                   /* :82 */
        }
        float sum = 0.f;
        for (int k = 0; k < seqLen; ++k) sum += scores[k];   /* :84-85 */
        for (int j = 0; j < dHead; ++j) O[q * dHead + j] = 0.f;
        for (int k = 0; k < seqLen; ++k) {                   /* :86-90 */
            const float w = scores[k] / sum;
            for (int j = 0; j < dHead; ++j) O[q * dHead + j] += w * V[k * dHead + j];
        }
    }
    free(scores);
}

int oracle_multi_head_attention(const float* Q, const float* K, const float* V,
                                float* output, float* attn,
                                int batch, int seqLen, int dModel, int numHeads, int nthreads) {
    if (numHeads <= 0 || dModel % numHeads != 0) return -1;
    const int dk = dModel / numHeads;                        /* check.py:11 */
    const float scale = (float)(1.0 / sqrt((double)dk));     /* check.py:19 */
    const int64_t n = (int64_t)batch * numHeads * seqLen * dk;
    float* q = (float*)malloc(sizeof(float) * (size_t)n * 4);
    if (!q) return -2;
    float *k = q + n, *v = k + n, *o = v + n;
    /* check.py:14-16: (B,S,H,d_k) -> (B,H,S,d_k) */
    for (int b = 0; b < batch; ++b)
        for (int s = 0; s < seqLen; ++s)
            for (int h = 0; h < numHeads; ++h) {
                const int64_t src = ((int64_t)b * seqLen + s) * dModel + (int64_t)h * dk;
                const int64_t dst = (((int64_t)b * numHeads + h) * seqLen + s) * dk;
                memcpy(q + dst, Q + src, sizeof(float) * (size_t)dk);
                memcpy(k + dst, K + src, sizeof(float) * (size_t)dk);
                memcpy(v + dst, V + src, sizeof(float) * (size_t)dk);
            }
    const int64_t rows = (int64_t)batch * numHeads * seqLen;
    nthreads = pick_threads(nthreads);
#pragma omp parallel num_threads(nthreads)
    {
        float* sc = (float*)malloc(sizeof(float) * (size_t)seqLen);
#pragma omp for schedule(dynamic, 16)
        for (int64_t r = 0; r < rows; ++r) {
            const int64_t g = r / seqLen;
            row_f32(q + r * dk, k + g * seqLen * dk, v + g * seqLen * dk, o + r * dk,
                    attn ? attn + r * seqLen : NULL, seqLen, dk, scale, seqLen, sc);
        }
        free(sc);
    }
    /* check.py:24: back to (B,S,H*d_k) */
    for (int b = 0; b < batch; ++b)
        for (int h = 0; h < numHeads; ++h)
            for (int s = 0; s < seqLen; ++s) {
                const int64_t src = (((int64_t)b * numHeads + h) * seqLen + s) * dk;
                const int64_t dst = ((int64_t)b * seqLen + s) * dModel + (int64_t)h * dk;
                memcpy(output + dst, o + src, sizeof(float) * (size_t)dk);
            }
    free(q);
    return nthreads;
}

/* ---------------- low-precision rounding helpers ---------------- */
uint16_t oracle_f32_to_bf16(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);  /* quiet NaN */
    u += 0x7fffu + ((u >> 16) & 1u);                                              /* RNE */
    return (uint16_t)(u >> 16);
}

float oracle_bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

float oracle_e4m3fn_to_f32(uint8_t b) {
    const int sign = b >> 7, e = (b >> 3) & 0xf, m = b & 7;
    float v;
    if (e == 0xf && m == 7) return NAN;                      /* the only NaN; no inf in e4m3fn */
    if (e == 0) v = ldexpf((float)m, -9);                    /* subnormal: m * 2^-3 * 2^-6 */
    else v = ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return sign ? -v : v;
}

uint8_t oracle_f32_to_e4m3fn(float x) {
    if (isnan(x)) return 0x7f;
    const uint8_t sign = signbit(x) ? 0x80 : 0;
    float a = fabsf(x);
    if (a >= 448.0f) return sign | 0x7e;                     /* saturate (max finite 448) */
    if (a < ldexpf(1.0f, -10)) return sign;                  /* below half the min subnormal */
    int e;
    (void)frexpf(a, &e);                                     /* a = f * 2^e, f in [0.5,1) */
    int E = e - 1;                                           /* a = 1.xxx * 2^E */
    if (E < -6) E = -6;                                      /* subnormal range shares 2^-6 */
    const float q = ldexpf(1.0f, E - 3);                     /* quantum */
    float r = nearbyintf(a / q);                             /* RNE (default rounding mode) */
    float v = r * q;
    if (v >= 448.0f) return sign | 0x7e;
    if (v < ldexpf(1.0f, -6)) return sign | (uint8_t)(int)(v * 512.0f);
    (void)frexpf(v, &e);
    E = e - 1;
    const int mant = (int)((v / ldexpf(1.0f, E) - 1.0f) * 8.0f + 0.5f);
    return sign | (uint8_t)(((E + 7) << 3) | mant);
}

void oracle_round_to_bf16(float* x, int64_t n) {
    for (int64_t i = 0; i < n; ++i) x[i] = oracle_bf16_to_f32(oracle_f32_to_bf16(x[i]));
}

void oracle_round_to_e4m3fn(float* x, int64_t n) {
    for (int64_t i = 0; i < n; ++i) x[i] = oracle_e4m3fn_to_f32(oracle_f32_to_e4m3fn(x[i]));
}
