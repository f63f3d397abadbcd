// inst_bf16_p16.hip -- bf16 inputs with the fp16-weights precision option (FA_FLAG_F16_WEIGHTS), D = 128 / 64
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

template <int D, bool CAUSAL>
hipError_t by_out(const Params& p, const fa_launch_plan& plan, int o_dtype, hipStream_t st) {
    if (o_dtype == FA_DTYPE_F32) return launch_mfma<P16Cfg<D, CAUSAL, float>>(p, plan, st);
    if (o_dtype == FA_DTYPE_BF16) return launch_mfma<P16Cfg<D, CAUSAL, __bf16>>(p, plan, st);
    return launch_mfma<P16Cfg<D, CAUSAL, _Float16>>(p, plan, st);
}

}  // namespace

hipError_t launch_bf16_p16(const Params& p, const fa_launch_plan& plan, bool causal, int d, int o_dtype, hipStream_t st) {
    if (d == 128) return causal ? by_out<128, true>(p, plan, o_dtype, st) : by_out<128, false>(p, plan, o_dtype, st);
    return causal ? by_out<64, true>(p, plan, o_dtype, st) : by_out<64, false>(p, plan, o_dtype, st);
}

namespace {
template <int D, bool CAUSAL>
int lds_by_out(int o_dtype) {
    if (o_dtype == FA_DTYPE_F32) return P16Cfg<D, CAUSAL, float>::LDS_BYTES;
    if (o_dtype == FA_DTYPE_BF16) return P16Cfg<D, CAUSAL, __bf16>::LDS_BYTES;
    return P16Cfg<D, CAUSAL, _Float16>::LDS_BYTES;
}
}  // namespace

int bf16_p16_lds_bytes(bool causal, int d, int o_dtype) {
    if (d == 128) return causal ? lds_by_out<128, true>(o_dtype) : lds_by_out<128, false>(o_dtype);
    return causal ? lds_by_out<64, true>(o_dtype) : lds_by_out<64, false>(o_dtype);
}

}  // namespace fa
