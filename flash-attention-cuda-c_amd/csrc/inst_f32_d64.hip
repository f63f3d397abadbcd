// inst_f32_d64.hip -- fp32 inputs, exact-fp32 MFMA kernel at D = 64 (one translation unit of libflash_attention.so: see launchers.hip.h).
#include "kernel_f32.hip.h"
#include "launchers.hip.h"

namespace fa {
namespace {

template <class Cfg>
hipError_t launch_f32(const Params& p, const fa_launch_plan& plan, hipStream_t st) {
    static std::atomic<bool> done[64];
    const hipError_t attr = raise_lds_limit(fwd_f32_mfma_kernel<Cfg>, Cfg::LDS_BYTES, done);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((fwd_f32_mfma_kernel<Cfg>), dim3(plan.grid), dim3(plan.threads), Cfg::LDS_BYTES, st, p);
    return hipGetLastError();
}

template <bool CAUSAL, bool PAD>
hipError_t by_out(const Params& p, const fa_launch_plan& plan, int o_dtype, hipStream_t st) {
    if (o_dtype == FA_DTYPE_F32) return launch_f32<F32Cfg<64, CAUSAL, float, PAD>>(p, plan, st);
    if (o_dtype == FA_DTYPE_BF16) return launch_f32<F32Cfg<64, CAUSAL, __bf16, PAD>>(p, plan, st);
    return launch_f32<F32Cfg<64, CAUSAL, _Float16, PAD>>(p, plan, st);
}

}  // namespace

hipError_t launch_f32_d64(const Params& p, const fa_launch_plan& plan, bool causal, bool pad, int o_dtype, hipStream_t st) {
    if (pad) return causal ? by_out<true, true>(p, plan, o_dtype, st) : by_out<false, true>(p, plan, o_dtype, st);
    return causal ? by_out<true, false>(p, plan, o_dtype, st) : by_out<false, false>(p, plan, o_dtype, st);
}

}  // namespace fa
