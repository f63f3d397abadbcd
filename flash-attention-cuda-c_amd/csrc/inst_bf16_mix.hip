// inst_bf16_mix.hip -- bf16 inputs under the causal mask at the library's default precision, D = 128 / 64: the bf16-weights kernel whose
// units of the first Params::hp query blocks of every head (the rows that see fewer than FA_EARLY_KEYS keys) run with fp16 softmax
// weights -- one walk over one (head, query block) list, every unit in the precision of its block (kernel_bf16.hip.h: KernelCfg::MIX,
// MixCfg; one translation unit of libflash_attention.so: see launchers.hip.h).
#include "kernel_bf16.hip.h"
#include "launchers.hip.h"

namespace fa {
namespace {

template <class Cfg>
hipError_t launch_mix(const Params& p, const fa_launch_plan& plan, hipStream_t st) {
    static std::atomic<bool> done[64];
    const hipError_t attr = raise_lds_limit(fwd_mfma_kernel<Cfg>, Cfg::LDS_BYTES, done);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((fwd_mfma_kernel<Cfg>), dim3(plan.grid), dim3(plan.threads), Cfg::LDS_BYTES, st, p);
    return hipGetLastError();
}

template <int D>
hipError_t by_out(const Params& p, const fa_launch_plan& plan, int o_dtype, hipStream_t st) {
    if (o_dtype == FA_DTYPE_F32) return launch_mix<MixCfg<D, float>>(p, plan, st);
    if (o_dtype == FA_DTYPE_BF16) return launch_mix<MixCfg<D, __bf16>>(p, plan, st);
    return launch_mix<MixCfg<D, _Float16>>(p, plan, st);
}

template <int D>
int lds_by_out(int o_dtype) {
    if (o_dtype == FA_DTYPE_F32) return MixCfg<D, float>::LDS_BYTES;
    if (o_dtype == FA_DTYPE_BF16) return MixCfg<D, __bf16>::LDS_BYTES;
    return MixCfg<D, _Float16>::LDS_BYTES;
}

}  // namespace

// p.hp: how many leading query blocks of every head take fp16 weights (0 < hp < p.nQ; p covers all query blocks)
hipError_t launch_bf16_causal_mix(const Params& p, const fa_launch_plan& plan, int d, int o_dtype, hipStream_t st) {
    return d == 128 ? by_out<128>(p, plan, o_dtype, st) : by_out<64>(p, plan, o_dtype, st);
}

int bf16_causal_mix_lds_bytes(int d, int o_dtype) { return d == 128 ? lds_by_out<128>(o_dtype) : lds_by_out<64>(o_dtype); }

}  // namespace fa
