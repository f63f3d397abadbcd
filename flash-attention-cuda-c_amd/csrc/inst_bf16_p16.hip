// inst_bf16_p16.hip -- bf16 inputs WITHOUT the causal mask, fp16 softmax weights on every row (FA_FLAG_F16_WEIGHTS, or seqLenK <
// FA_EARLY_KEYS), D = 128 / 64: the mixed-precision kernel of the 16x16x32 engine with Params::hp = all query blocks -- K by LDS-DMA, V
// as fp16 through registers (MixStage); +1.5 ... +2.4 % over both tiles through registers (profiles/r04_tune_g_*.log).  Under the
// mask the 32x32x16 mixed-precision kernel of inst_bf16_mix.hip serves these calls.
// (one translation unit of libflash_attention.so: see launchers.hip.h).
#include "kernel_bf16.hip.h"
#include "launchers.hip.h"

namespace fa {
namespace {

template <class Cfg>
hipError_t launch_mfma(const Params& p, const fa_launch_plan& plan, hipStream_t st) {
    static std::atomic<bool> done[64];
    const hipError_t attr = raise_lds_limit(fwd_mfma_kernel<Cfg>, Cfg::LDS_BYTES, done);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((fwd_mfma_kernel<Cfg>), dim3(plan.grid), dim3(plan.threads), Cfg::LDS_BYTES, st, p);
    return hipGetLastError();
}

// fp32 sum of the unrounded weights (the LSE is exact either way), every unit "early"
template <int D, typename OutT>
using F16Cfg = KernelCfg<D, false, OutT, 2, Opt{.sum_mfma = 0, .mix = true}>;

template <int D, bool CAUSAL>
hipError_t by_out(const Params& p, const fa_launch_plan& plan, int o_dtype, hipStream_t st) {
    static_assert(!CAUSAL, "without the mask only");
    if (o_dtype == FA_DTYPE_F32) return launch_mfma<F16Cfg<D, float>>(p, plan, st);
    if (o_dtype == FA_DTYPE_BF16) return launch_mfma<F16Cfg<D, __bf16>>(p, plan, st);
    return launch_mfma<F16Cfg<D, _Float16>>(p, plan, st);
}

}  // namespace

hipError_t launch_bf16_p16(const Params& p, const fa_launch_plan& plan, bool causal, int d, int o_dtype, hipStream_t st) {
    if (causal) return hipErrorInvalidValue;   // (never requested: FlashAttention.hip sends causal problems to launch_bf16_causal_mix)
    return d == 128 ? by_out<128, false>(p, plan, o_dtype, st) : by_out<64, false>(p, plan, o_dtype, st);
}

namespace {
template <int D, bool CAUSAL>
int lds_by_out(int o_dtype) {
    if (o_dtype == FA_DTYPE_F32) return F16Cfg<D, float>::LDS_BYTES;
    if (o_dtype == FA_DTYPE_BF16) return F16Cfg<D, __bf16>::LDS_BYTES;
    return F16Cfg<D, _Float16>::LDS_BYTES;
}
}  // namespace

int bf16_p16_lds_bytes(bool causal, int d, int o_dtype) {
    (void)causal;   // (the carve-up does not depend on the mask)
    return d == 128 ? lds_by_out<128, false>(o_dtype) : lds_by_out<64, false>(o_dtype);
}

}  // namespace fa
