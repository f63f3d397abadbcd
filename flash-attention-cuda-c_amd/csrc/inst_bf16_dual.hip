// inst_bf16_dual.hip -- bf16 inputs under the causal mask at the library's default precision, D = 128 / 64: ONE launch that runs the
// bf16-weights kernel on the late query blocks and then the fp16-weights kernel on the early ones, over one unit list (kernel_bf16.hip.h:
// fwd_mfma_dual_kernel; one translation unit of libflash_attention.so: see launchers.hip.h).
#include "kernel_bf16.hip.h"
#include "launchers.hip.h"

namespace fa {
namespace {

template <class CA, class CB>
hipError_t launch_dual(const Params& p, const UnitList& la, const UnitList& lb, int hp, const fa_launch_plan& plan, hipStream_t st) {
    constexpr int lds = CA::LDS_BYTES > CB::LDS_BYTES ? CA::LDS_BYTES : CB::LDS_BYTES;
    static std::atomic<bool> done[64];
    const hipError_t attr = raise_lds_limit(fwd_mfma_dual_kernel<CA, CB>, lds, done);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((fwd_mfma_dual_kernel<CA, CB>), dim3(plan.grid), dim3(plan.threads), lds, st, p, la, lb, hp);
    return hipGetLastError();
}

template <int D>
hipError_t by_out(const Params& p, const UnitList& la, const UnitList& lb, int hp, const fa_launch_plan& plan, int o_dtype, hipStream_t st) {
    if (o_dtype == FA_DTYPE_F32) return launch_dual<ProdCfg<D, true, float>, P16Cfg<D, true, float>>(p, la, lb, hp, plan, st);
    if (o_dtype == FA_DTYPE_BF16) return launch_dual<ProdCfg<D, true, __bf16>, P16Cfg<D, true, __bf16>>(p, la, lb, hp, plan, st);
    return launch_dual<ProdCfg<D, true, _Float16>, P16Cfg<D, true, _Float16>>(p, la, lb, hp, plan, st);
}

template <int D>
int lds_by_out(int o_dtype) {
    auto mx = [](int a, int b) { return a > b ? a : b; };
    if (o_dtype == FA_DTYPE_F32) return mx(ProdCfg<D, true, float>::LDS_BYTES, P16Cfg<D, true, float>::LDS_BYTES);
    if (o_dtype == FA_DTYPE_BF16) return mx(ProdCfg<D, true, __bf16>::LDS_BYTES, P16Cfg<D, true, __bf16>::LDS_BYTES);
    return mx(ProdCfg<D, true, _Float16>::LDS_BYTES, P16Cfg<D, true, _Float16>::LDS_BYTES);
}

}  // namespace

// la: the list the bf16-weights kernel walks (it takes the units of blocks >= hp), lb: the list the fp16-weights kernel walks (blocks
// < hp) -- two lists of their own, or twice the list over all blocks (kernel_bf16.hip.h: fwd_mfma_dual_kernel); plan.grid covers the larger
hipError_t launch_bf16_causal_dual(const Params& p, const UnitList& la, const UnitList& lb, int hp, const fa_launch_plan& plan, int d, int o_dtype,
                                   hipStream_t st) {
    return d == 128 ? by_out<128>(p, la, lb, hp, plan, o_dtype, st) : by_out<64>(p, la, lb, hp, plan, o_dtype, st);
}

int bf16_causal_dual_lds_bytes(int d, int o_dtype) { return d == 128 ? lds_by_out<128>(o_dtype) : lds_by_out<64>(o_dtype); }

}  // namespace fa
