// inst_bf16_d64.hip -- bf16 inputs, MFMA kernel at D = 64 (one translation unit of libflash_attention.so: see launchers.hip.h).
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

template <bool CAUSAL, bool PAD, bool LSE>
hipError_t by_out(const Params& p, const fa_launch_plan& plan, int o_dtype, hipStream_t st) {
    constexpr int D = 64, ESZ = 2;
    if (o_dtype == FA_DTYPE_F32) return launch_mfma<ProdCfg<D, CAUSAL, float, ESZ, false, PAD, LSE>>(p, plan, st);
    if (o_dtype == FA_DTYPE_BF16) return launch_mfma<ProdCfg<D, CAUSAL, __bf16, ESZ, false, PAD, LSE>>(p, plan, st);
    return launch_mfma<ProdCfg<D, CAUSAL, _Float16, ESZ, false, PAD, LSE>>(p, plan, st);
}

template <bool PAD, int ESZ_ = 2>
hipError_t by_causal_lse(const Params& p, const fa_launch_plan& plan, bool causal, int o_dtype, hipStream_t st) {
    // bf16 inputs without the mask (16x16x32 engine): a call that also wants the LSE runs the instantiation that sums the
    // unrounded weights.  The causal and the fp8 kernels (32x32x16 engine) sum unrounded weights anyway.
    if constexpr (ESZ_ == 2) {
        if (p.lse != nullptr && !causal) return by_out<false, PAD, true>(p, plan, o_dtype, st);
    }
    return causal ? by_out<true, PAD, false>(p, plan, o_dtype, st) : by_out<false, PAD, false>(p, plan, o_dtype, st);
}

}  // namespace

hipError_t launch_bf16_d64(const Params& p, const fa_launch_plan& plan, bool causal, bool pad, int o_dtype, hipStream_t st) {
    return pad ? by_causal_lse<true>(p, plan, causal, o_dtype, st) : by_causal_lse<false>(p, plan, causal, o_dtype, st);
}

namespace {
template <bool CAUSAL, bool PAD>
int lds_by_out(int o_dtype) {
    constexpr int D = 64, ESZ = 2;
    if (o_dtype == FA_DTYPE_F32) return ProdCfg<D, CAUSAL, float, ESZ, false, PAD, false>::LDS_BYTES;
    if (o_dtype == FA_DTYPE_BF16) return ProdCfg<D, CAUSAL, __bf16, ESZ, false, PAD, false>::LDS_BYTES;
    return ProdCfg<D, CAUSAL, _Float16, ESZ, false, PAD, false>::LDS_BYTES;
}
}  // namespace

int bf16_d64_lds_bytes(bool causal, bool pad, int o_dtype) {
    return causal ? (pad ? lds_by_out<true, true>(o_dtype) : lds_by_out<true, false>(o_dtype))
                  : (pad ? lds_by_out<false, true>(o_dtype) : lds_by_out<false, false>(o_dtype));
}

}  // namespace fa
