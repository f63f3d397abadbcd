// tests/fa_tune.hip -- A/B harness for kernel variants: instantiates several configurations of the
// bf16 kernels side by side, checks each against the first (and a sampled oracle check), and times
// them in INTERLEAVED rounds inside one process on random N(0,1) data (devices and separate runs
// differ by more than the deltas being measured).  Test/tuning infrastructure, not product.
//
// usage: fa_tune [B H S d causal] [--rounds R] [--only i,j,...]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../flash-attention-cuda-c_amd/csrc/kernel_bf16.hip.h"
#include "../oracle/cpu_attention.h"

#define HIP_CHECK(x)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "HIP error: %s (%s:%d)\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                          \
        }                                                                                     \
    } while (0)

using namespace fa;

static inline uint64_t mix(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static void fill_randn_bf16(std::vector<uint16_t>& v, uint64_t seed, double mul = 1.0) {
    for (size_t i = 0; i < v.size(); i += 2) {
        const uint64_t a = mix(seed * 0x9e3779b1ull + i), b = mix(seed * 0x85ebca6bull + i + 1);
        const double u1 = ((a >> 11) + 1.0) / 9007199254740993.0, u2 = (b >> 11) / 9007199254740992.0;
        const double r = std::sqrt(-2.0 * std::log(u1));
        v[i] = oracle_f32_to_bf16((float)(mul * r * std::cos(6.283185307179586 * u2)));
        if (i + 1 < v.size()) v[i + 1] = oracle_f32_to_bf16((float)(mul * r * std::sin(6.283185307179586 * u2)));
    }
}

static int g_cus = 256, g_jpx = 0;   // --jpx N: workgroups per XCD group of the persistent variants (default CUs/8)

struct Variant {
    std::string name;
    std::function<void(const Params&, int grid)> launch;
    int osz = 2;   // bytes per output element (2: bf16, 4: fp32)
};

template <class K>
static void launch_cfg(const Params& p, int grid) {
    static bool once = [] {
        HIP_CHECK(hipFuncSetAttribute((const void*)fwd_mfma_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
        return true;
    }();
    (void)once;
    (void)grid;
    // persistent grid: one workgroup per CU (256 on MI355X), or one per unit when there are fewer units than CUs
    Params q = p;
    q.jpx = std::min(q.cpx, g_jpx > 0 ? g_jpx : g_cus / 8);
    hipLaunchKernelGGL((fwd_mfma_kernel<K>), dim3(8 * q.jpx), dim3(64 * K::NWAVES), K::LDS_BYTES, nullptr, q);
}

// Every variant is a library configuration, or the production configuration with one knob under study changed.  (The arms of
// rounds 1 and 2 -- 64-row waves, 4-slot ring, ping-pong phases, unit streaming, packed softmax math, wait grouping, ... -- were
// measured, rejected and removed; their logs are profiles/r01_tune_*, r02_tune_*.)
template <int D, bool CAUSAL>
static std::vector<Variant> make_variants() {
    using T = __bf16;
    std::vector<Variant> v;
    constexpr int M = CAUSAL ? 0 : -1;   // production engine choice
    v.push_back({"production (16x16x32 non-causal, 32x32x16 causal), bf16 O", launch_cfg<ProdCfg<D, CAUSAL, T>>});
    v.push_back({"fp32 O (production)", launch_cfg<ProdCfg<D, CAUSAL, float>>, 4});
    // the default precision as two launches of the single-precision kernels (what the mixed kernel replaces)
    if constexpr (CAUSAL) {
        v.push_back({"fp32 O, default precision as TWO launches: fp16-weights kernel on query blocks 0-3, bf16-weights kernel on the rest",
                     [](const Params& p, int grid) {
                         const int hp = std::min(p.nQ, 1024 / 256);
                         Params a = p, b = p;
                         a.qb0 = 0; a.nQ = hp; a.units = p.B * p.H * hp; a.cpx = (a.units + 7) / 8;
                         b.qb0 = hp; b.nQ = p.nQ - hp; b.units = p.B * p.H * b.nQ; b.cpx = (b.units + 7) / 8;
                         launch_cfg<KernelCfg<D, CAUSAL, float, 2, Opt{.sum_mfma = 0, .p_f16 = true}>>(a, grid);
                         if (b.nQ > 0) launch_cfg<ProdCfg<D, CAUSAL, float>>(b, grid);
                     }, 4});
        v.push_back({"fp32 O, library default: the MIXED kernel (one list, fp16 weights on query blocks 0-3)",
                     [](const Params& p, int grid) { Params q = p; q.hp = std::min(p.nQ, 1024 / 256); launch_cfg<MixCfg<D, float>>(q, grid); }, 4});
        v.push_back({"fp32 O, MIXED kernel, hp = 0 (bf16 weights everywhere: must equal the production kernel)",
                     [](const Params& p, int grid) { Params q = p; q.hp = 0; launch_cfg<MixCfg<D, float>>(q, grid); }, 4});
        v.push_back({"STAMP fp32 O, MIXED kernel",
                     [](const Params& p, int grid) { Params q = p; q.hp = std::min(p.nQ, 1024 / 256); launch_cfg<MixCfg<D, float, true>>(q, grid); }, 4});
        v.push_back({"fp32 O, fp16 weights on query blocks 0-3 ONLY (the first of the two launches)",
                     [](const Params& p, int grid) {
                         const int hp = std::min(p.nQ, 1024 / 256);
                         Params a = p;
                         a.qb0 = 0; a.nQ = hp; a.units = p.B * p.H * hp; a.cpx = (a.units + 7) / 8;
                         launch_cfg<KernelCfg<D, CAUSAL, float, 2, Opt{.sum_mfma = 0, .p_f16 = true}>>(a, grid);
                     }, 4});
    }
    {
        // small problems: 128-row units, one per workgroup of four waves (fwd_mfma_pair_kernel; causal d = 64: two workgroups per CU, paired)
        auto pair = [](const Params& p, unsigned flags) {
            // (the library's configurations: inst_bf16_pair_d64.hip; inst_bf16_pair_d128.hip: one mixed-precision configuration for both)
            using CA = std::conditional_t<D == 128, KernelCfg<D, CAUSAL, float, 2, Opt{.m16 = CAUSAL ? 0 : -1, .sum_mfma = 0, .waves = 4, .mix = true}>,
                                          KernelCfg<D, CAUSAL, float, 2, Opt{.m16 = CAUSAL ? 0 : -1, .sum_mfma = 0, .waves = 4}>>;
            using CB = std::conditional_t<D == 128, CA, KernelCfg<D, CAUSAL, float, 2, Opt{.sum_mfma = 0, .waves = 4, .p_f16 = true}>>;
            constexpr int lds = CA::LDS_BYTES > CB::LDS_BYTES ? CA::LDS_BYTES : CB::LDS_BYTES;
            static bool once = [] {
                HIP_CHECK(hipFuncSetAttribute((const void*)fwd_mfma_pair_kernel<CA, CB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                return true;
            }();
            (void)once;
            Params q = p;
            q.nQ = (p.S + 127) / 128;
            const int hp = (flags == 2 || !CAUSAL) ? 0 : std::min(q.nQ, 1024 / 128);
            q.hp = hp;
            const int jpx = g_cus / 8;
            const int per_group = ((p.B * p.H + 7) / 8) * q.nQ;
            hipLaunchKernelGGL((fwd_mfma_pair_kernel<CA, CB>), dim3(8 * (per_group <= jpx ? per_group : 2 * jpx)), dim3(256), lds, nullptr, q, hp, jpx);
        };
        v.push_back({"fp32 O, PAIR kernel: 128-row units, 2 workgroups per CU, default precision", [pair](const Params& p, int) { pair(p, 0); }, 4});
        v.push_back({"fp32 O, PAIR kernel: 128-row units, 2 workgroups per CU, bf16 weights", [pair](const Params& p, int) { pair(p, 2); }, 4});
    }
    v.push_back({"production STAMP, bf16 O", launch_cfg<ProdCfg<D, CAUSAL, T, 2, true>>});
    v.push_back({"production STAMP, fp32 O", launch_cfg<ProdCfg<D, CAUSAL, float, 2, true>>, 4});
    v.push_back({"the other engine", launch_cfg<KernelCfg<D, CAUSAL, T, 2, Opt{.m16 = CAUSAL ? 1 : 0}>>});
    v.push_back({"fp16 weights (FA_FLAG_F16_WEIGHTS)", launch_cfg<KernelCfg<D, CAUSAL, T, 2, Opt{.sum_mfma = 0, .p_f16 = true}>>});
    v.push_back({"exact row sums (the LSE instantiation)", launch_cfg<ProdCfg<D, CAUSAL, T, 2, false, false, true>>});
    v.push_back({"fp32 O, the other engine", launch_cfg<KernelCfg<D, CAUSAL, float, 2, Opt{.m16 = CAUSAL ? 1 : 0}>>, 4});
    v.push_back({"fp32 O, fp16 weights (FA_FLAG_F16_WEIGHTS: 16x16x32 engine, K and V through registers)", launch_cfg<KernelCfg<D, CAUSAL, float, 2, Opt{.sum_mfma = 0, .p_f16 = true}>>, 4});
    v.push_back({"fp32 O, fp16 weights on EVERY unit through the mixed kernel, 16x16x32 engine (K by LDS-DMA, V through registers)",
                 [](const Params& p, int grid) { Params q = p; q.hp = p.nQ; launch_cfg<KernelCfg<D, CAUSAL, float, 2, Opt{.sum_mfma = 0, .mix = true}>>(q, grid); }, 4});
    v.push_back({"fp32 O, fp16 weights on EVERY unit through the mixed kernel (32x32x16 engine, K by LDS-DMA, V through registers)",
                 [](const Params& p, int grid) { Params q = p; q.hp = p.nQ; launch_cfg<KernelCfg<D, CAUSAL, float, 2, Opt{.m16 = 0, .mix = true}>>(q, grid); }, 4});
    return v;
}

template <bool CAUSAL>
static std::vector<Variant> make_variants_fp8() {
    std::vector<Variant> v;
    v.push_back({"fp8 production (MX: QK^T on 32x32x64 f8f6f4, unit scales; K by LDS-DMA)", launch_cfg<ProdCfg<128, CAUSAL, __bf16, 1>>});
    return v;
}

int main(int argc, char** argv) {
    int B = 8, H = 16, S = 4096, d = 128, causal = 0, rounds = 7;
    bool fp8 = false;
    double qkscale = 1.0;   // multiplies Q and K: 12 makes later tiles exceed the tile-0 row max by > 2^127 (fallback path)
    std::vector<int> only;
    std::vector<int> pos;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--rounds" && i + 1 < argc) rounds = atoi(argv[++i]);
        else if (a == "--qkscale" && i + 1 < argc) qkscale = atof(argv[++i]);
        else if (a == "--jpx" && i + 1 < argc) g_jpx = atoi(argv[++i]);
        else if (a == "--fp8") fp8 = true;   // OCP e4m3fn inputs (d = 128): the fp8 kernel variants
        else if (a == "--only" && i + 1 < argc) {
            char* s = argv[++i];
            for (char* t = strtok(s, ","); t; t = strtok(nullptr, ",")) only.push_back(atoi(t));
        } else pos.push_back(atoi(argv[i]));
    }
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 8)
            g_cus = cus;
    }
    if (pos.size() >= 5) { B = pos[0]; H = pos[1]; S = pos[2]; d = pos[3]; causal = pos[4]; }
    const int BH = B * H;
    const size_t per_head = (size_t)S * d, n = per_head * BH;
    const int distinct = std::min(BH, 8);
    std::vector<uint16_t> hq(per_head * distinct), hk(per_head * distinct), hv(per_head * distinct);
    fill_randn_bf16(hq, 1, qkscale); fill_randn_bf16(hk, 2, qkscale); fill_randn_bf16(hv, 3);
    const int esz = fp8 ? 1 : 2;
    std::vector<uint8_t> eq, ek, ev;
    if (fp8) {   // round the bf16 draws to e4m3fn; the host copies (hq..) keep the rounded values as bf16 (exact)
        eq.resize(hq.size()); ek.resize(hk.size()); ev.resize(hv.size());
        for (size_t i = 0; i < hq.size(); ++i) {
            eq[i] = oracle_f32_to_e4m3fn(oracle_bf16_to_f32(hq[i])); hq[i] = oracle_f32_to_bf16(oracle_e4m3fn_to_f32(eq[i]));
            ek[i] = oracle_f32_to_e4m3fn(oracle_bf16_to_f32(hk[i])); hk[i] = oracle_f32_to_bf16(oracle_e4m3fn_to_f32(ek[i]));
            ev[i] = oracle_f32_to_e4m3fn(oracle_bf16_to_f32(hv[i])); hv[i] = oracle_f32_to_bf16(oracle_e4m3fn_to_f32(ev[i]));
        }
    }
    void *dq, *dk, *dv, *dref, *dout;
    HIP_CHECK(hipMalloc(&dq, n * 2)); HIP_CHECK(hipMalloc(&dk, n * 2)); HIP_CHECK(hipMalloc(&dv, n * 2));
    HIP_CHECK(hipMalloc(&dref, n * 2)); HIP_CHECK(hipMalloc(&dout, n * 4));   // room for fp32-output variants
    for (int t = 0; t < 3; ++t) {
        char* dst = (char*)(t == 0 ? dq : t == 1 ? dk : dv);
        const void* src = fp8 ? (const void*)(t == 0 ? eq : t == 1 ? ek : ev).data() : (const void*)(t == 0 ? hq : t == 1 ? hk : hv).data();
        for (int g = 0; g < BH; g += distinct) {
            const int cnt = std::min(distinct, BH - g);
            HIP_CHECK(hipMemcpy(dst + (size_t)g * per_head * esz, src, per_head * cnt * esz, hipMemcpyHostToDevice));
        }
    }
    Params p{};
    p.Q = dq; p.K = dk; p.V = dv; p.O = dout;
    p.qS = p.kS = p.vS = p.oS = d;
    p.qH = p.kH = p.vH = p.oH = (int64_t)S * d;
    p.qB = p.kB = p.vB = p.oB = (int64_t)H * S * d;
    p.B = B; p.H = H; p.S = S; p.Sk = S;
    p.nQ = (S + 255) / 256;
    p.units = BH * p.nQ;
    p.cpx = (p.units + 7) / 8;
    p.scale = 1.0f / std::sqrt((float)d);
    p.scale_log2 = p.scale * 1.4426950408889634f;
    const int grid = 8 * p.cpx;
    unsigned long long* ddbg;
    HIP_CHECK(hipMalloc(&ddbg, (size_t)grid * 192 * 8));
    HIP_CHECK(hipMemset(ddbg, 0, (size_t)grid * 192 * 8));
    p.dbg = ddbg;

    std::vector<Variant> vars;
#if defined(FA_TUNE_CAUSAL_D128)
    if (fp8) { fprintf(stderr, "this build holds the causal d = 128 bf16 variants only\n"); return 2; }
#else
    if (fp8 && d == 128) vars = causal ? make_variants_fp8<true>() : make_variants_fp8<false>();
#endif
    else if (fp8) { fprintf(stderr, "--fp8 needs d = 128\n"); return 2; }
#if defined(FA_TUNE_CAUSAL_D128)     // (a build with one problem class: a third of the compile time)
    else if (d == 128 && causal) vars = make_variants<128, true>();
#else
    else if (d == 128) vars = causal ? make_variants<128, true>() : make_variants<128, false>();
    else if (d == 64) vars = causal ? make_variants<64, true>() : make_variants<64, false>();
#endif
    else { fprintf(stderr, "d must be 128 or 64\n"); return 2; }
    if (!only.empty()) {
        std::vector<Variant> sel;
        for (int i : only) if (i >= 0 && i < (int)vars.size()) sel.push_back(vars[i]);
        vars = sel;
    }

    // correctness: oracle on head 0 (all rows) and the last head's last 64 rows; variants vs variant 0
    std::vector<float> fq(per_head), fk(per_head), fv(per_head), ref(per_head);
    auto host_head = [&](int g) {
        const size_t off = (size_t)(g % distinct) * per_head;
        for (size_t i = 0; i < per_head; ++i) {
            fq[i] = oracle_bf16_to_f32(hq[off + i]); fk[i] = oracle_bf16_to_f32(hk[off + i]); fv[i] = oracle_bf16_to_f32(hv[off + i]);
        }
    };
    std::vector<uint16_t> out0(n), outv(n);
    std::vector<float> outf(n);
    printf("problem: B=%d H=%d S=%d d=%d causal=%d  grid=%d  rounds=%d qkscale=%g jpx=%d\n", B, H, S, d, causal, grid, rounds, qkscale, g_jpx);
    for (size_t vi = 0; vi < vars.size(); ++vi) {
        HIP_CHECK(hipMemset(dout, 0xff, n * 4));
        vars[vi].launch(p, grid);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        if (vars[vi].osz == 4) {   // fp32 output: round to bf16 on the host so the comparisons below stay uniform
            HIP_CHECK(hipMemcpy(outf.data(), dout, n * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n; ++i) outv[i] = oracle_f32_to_bf16(outf[i]);
        } else {
            HIP_CHECK(hipMemcpy(outv.data(), dout, n * 2, hipMemcpyDeviceToHost));
        }
        double max_err = 0;
        size_t bad = 0;
        for (int g : {0, BH - 1}) {
            host_head(g);
            const int r0 = g == 0 ? 0 : S - 64;
            oracle_attention_f64acc_rows(fq.data(), fk.data(), fv.data(), ref.data(), 1, S, d, p.scale, causal, 0, 1, r0, S, 0);
            for (size_t i = (size_t)r0 * d; i < per_head; ++i) {
                const double e = std::fabs((double)oracle_bf16_to_f32(outv[(size_t)g * per_head + i]) - ref[i]);
                if (!(e <= 8e-3 + 8e-3 * std::fabs(ref[i]))) ++bad;
                if (e == e) max_err = std::max(max_err, e);
            }
        }
        double max_diff0 = 0;
        if (vi == 0) out0 = outv;
        else
            for (size_t i = 0; i < n; ++i)
                max_diff0 = std::max(max_diff0, (double)std::fabs(oracle_bf16_to_f32(outv[i]) - oracle_bf16_to_f32(out0[i])));
        printf("  [%zu] %-32s oracle max_abs_err=%.3e bad=%zu  max|diff vs [0]|=%.3e %s\n", vi, vars[vi].name.c_str(), max_err,
               bad, max_diff0, bad ? "FAIL" : "ok");
    }

    // timing: interleaved rounds
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    const int reps = 5;
    std::vector<std::vector<float>> ms(vars.size());
    for (size_t vi = 0; vi < vars.size(); ++vi) { vars[vi].launch(p, grid); vars[vi].launch(p, grid); }
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemset(ddbg, 0, (size_t)grid * 192 * 8));   // stamp rows: only the STAMP variant(s) timed below write them
    // every round visits the variants in a fresh (seeded) random order: a variant's clock depends on what ran just before it
    // (a fixed order gave identical kernels 3.6 % apart), so the predecessor must not be the same in every round
    std::vector<size_t> order(vars.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    unsigned long long lcg = 0x9E3779B97F4A7C15ull;
    for (int r = 0; r < rounds; ++r) {
        for (size_t i = order.size(); i > 1; --i) {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(order[i - 1], order[(lcg >> 33) % i]);
        }
        for (size_t oi = 0; oi < order.size(); ++oi) {
            const size_t vi = order[oi];
            HIP_CHECK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < reps; ++i) vars[vi].launch(p, grid);
            HIP_CHECK(hipEventRecord(e1, nullptr));
            HIP_CHECK(hipEventSynchronize(e1));
            float t;
            HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
            ms[vi].push_back(t / reps);
        }
    }
    {   // segment stamps of the STAMP variant (if it ran)
        std::vector<unsigned long long> h((size_t)grid * 192);
        HIP_CHECK(hipMemcpy(h.data(), ddbg, h.size() * 8, hipMemcpyDeviceToHost));
        double seg[24] = {0};
        for (size_t i = 0; i < (size_t)grid * 8; ++i)
            for (int k = 0; k < 24; ++k) seg[k] += (double)h[i * 24 + k];
        if (seg[6] > 0) {
            const double nt = seg[6], nw = seg[11] > 0 ? seg[11] : 1;
            printf("  STAMP build (each stamp costs ~40-60 cycles):\n");
            printf("    per tile per wave (%.0f wave-tiles): phase A %.1f | phase B %.1f | end-of-tile %.1f | barrier %.1f | sum %.1f\n", nt,
                   seg[1] / nt, seg[2] / nt, seg[3] / nt, seg[5] / nt, (seg[1] + seg[2] + seg[3] + seg[5]) / nt);
            for (int wv = 0; wv < 8; ++wv) {   // per wave index: does the older half (waves 0-3) wait at the barrier?
                double a = 0, b = 0, bar = 0, n = 0, ep = 0, epb = 0, w0 = 0, nwv = 0;
                for (size_t g = 0; g < (size_t)grid; ++g) {
                    const unsigned long long* r = &h[(g * 8 + wv) * 24];
                    a += (double)r[1]; b += (double)r[2]; bar += (double)r[5]; n += (double)r[6];
                    ep += (double)r[4]; epb += (double)r[15]; w0 += (double)r[16]; nwv += (double)r[11];
                }
                if (n > 0) printf("      wave %d: phase A %.0f  phase B %.0f  barrier wait %.0f   | per workgroup-wave: epilogue issue %.0f  post-epilogue barrier %.0f  prologue vmcnt(0) %.0f\n",
                                  wv, a / n, b / n, bar / n, ep / nwv, epb / nwv, w0 / nwv);
            }
            {   // spread of workgroup lifetimes (wave 0 of each): a static schedule ends with its slowest workgroup
                double mn = 1e30, mx = 0, sum = 0; int cnt = 0;
                double xsum[8] = {0}; int xcnt[8] = {0};
                for (size_t g = 0; g < (size_t)grid; ++g) {
                    const double life = (double)h[(g * 8) * 24 + 0];
                    if (life <= 0) continue;
                    mn = std::min(mn, life); mx = std::max(mx, life); sum += life; ++cnt;
                    xsum[g & 7] += life; ++xcnt[g & 7];
                }
                if (cnt > 0) {
                    printf("    workgroup lifetimes (%d workgroups): min %.0f  mean %.0f  max %.0f cycles  (max/mean %.3f); per XCD mean:", cnt, mn, sum / cnt, mx, mx / (sum / cnt));
                    for (int x = 0; x < 8; ++x) printf(" %.0f", xcnt[x] ? xsum[x] / xcnt[x] : 0.0);
                    printf("\n");
                    // the same lifetimes on the constant 100 MHz counter: the core clock each XCD ran at, and when its workgroups ended
                    double ticks = 0, s0 = 1e30, s1 = 0, e0 = 1e30, e1 = 0;
                    for (size_t g = 0; g < (size_t)grid; ++g) {
                        const double a = (double)h[(g * 8) * 24 + 18], b = (double)h[(g * 8) * 24 + 19];
                        if (b <= 0) continue;
                        ticks += (double)h[(g * 8) * 24 + 17];
                        s0 = std::min(s0, a); s1 = std::max(s1, a); e0 = std::min(e0, b); e1 = std::max(e1, b);
                    }
                    if (ticks > 0) {
                        printf("    per XCD (workgroup index & 7): mean lifetime us / clock GHz / last end after the first start, us:");
                        for (int x = 0; x < 8; ++x) {
                            double tk = 0, cy = 0, le = 0; int n = 0;
                            for (size_t g = x; g < (size_t)grid; g += 8) {
                                const double b = (double)h[(g * 8) * 24 + 19];
                                if (b <= 0) continue;
                                tk += (double)h[(g * 8) * 24 + 17]; cy += (double)h[(g * 8) * 24 + 0]; le = std::max(le, b); ++n;
                            }
                            if (n) printf("  %.1f/%.3f/%.1f", tk / n * 0.01, cy / tk * 0.1, (le - s0) * 0.01);
                        }
                        printf("\n");
                        printf("    last launch, 100 MHz counter: first workgroup start -> last workgroup end %.1f us; starts spread over %.1f us, ends over %.1f us\n",
                               (e1 - s0) * 0.01, (s1 - s0) * 0.01, (e1 - e0) * 0.01);
                        printf("    core clock over the workgroup lifetimes (s_memtime / s_memrealtime at 100 MHz): %.3f GHz; mean lifetime %.1f us\n",
                               sum / ticks * 0.1, ticks / cnt * 0.01);
                    }
                }
            }
            const double loop = (seg[1] + seg[2] + seg[3] + seg[5]) / nw, tot = seg[0] / nw;
            printf("    per workgroup-wave (%.0f waves): lifetime %.0f = Q load+pin %.0f | stage tiles 0,1 + barrier %.0f | QK(0)+max %.0f | tile loop %.0f (%.1f%%) | finite check %.0f | epilogue %.0f | unaccounted %.0f\n",
                   nw, tot, seg[7] / nw, seg[8] / nw, seg[9] / nw, loop, 100 * loop / tot, seg[10] / nw, seg[4] / nw,
                   tot - loop - (seg[7] + seg[8] + seg[9] + seg[10] + seg[4]) / nw);
            printf("      of the unaccounted: setup (unit decode, descriptors, first loads issued) %.0f | next unit decode + prefetch issue %.0f | store tail (vmcnt(0) at exit) %.0f | post-epilogue barrier %.0f\n",
                   seg[12] / nw, seg[13] / nw, seg[14] / nw, seg[15] / nw);
            printf("      inside 'stage tiles 0,1 + barrier': the first vmcnt(0) of the prologue %.0f\n", seg[16] / nw);
        }
    }
    const double flops = (causal ? 2.0 : 4.0) * BH * (double)S * S * d;
    for (size_t vi = 0; vi < vars.size(); ++vi) {
        std::sort(ms[vi].begin(), ms[vi].end());
        const double med = ms[vi][ms[vi].size() / 2], mn = ms[vi][0];
        printf("  [%zu] %-32s med %.4f ms  min %.4f ms  %.1f TFLOP/s (%.1f%% of peak)  best %.1f\n", vi, vars[vi].name.c_str(), med,
               mn, flops / med / 1e9, 100.0 * flops / med / 1e9 / 2516.6, flops / mn / 1e9);
    }
    return 0;
}
