// inst_bf16_pair_d64.hip -- bf16 inputs, D = 64, causal mask, small problems (at most one 256-row unit per CU): 128-row units, one per
// workgroup of four waves, two workgroups per CU paired heaviest + lightest (kernel_bf16.hip.h: fwd_mfma_pair_kernel; one translation
// unit of libflash_attention.so: see launchers.hip.h).  D = 128 (one workgroup per CU): inst_bf16_pair_d128.hip.
#include "kernel_bf16.hip.h"
#include "launchers.hip.h"

namespace fa {
namespace {

template <typename OutT>
using PairA = KernelCfg<64, true, OutT, 2, Opt{.m16 = 0, .waves = 4}>;                           // bf16 weights (the 32x32x16 engine, LDS-DMA staging)
template <typename OutT>
using PairB = KernelCfg<64, true, OutT, 2, Opt{.sum_mfma = 0, .waves = 4, .p_f16 = true}>;      // fp16 weights

template <typename OutT>
constexpr int pair_lds() { return PairA<OutT>::LDS_BYTES > PairB<OutT>::LDS_BYTES ? PairA<OutT>::LDS_BYTES : PairB<OutT>::LDS_BYTES; }
static_assert(pair_lds<float>() <= 80 * 1024 - 256 && pair_lds<__bf16>() <= 80 * 1024 - 256, "two workgroups per CU");

template <typename OutT>
hipError_t launch_pair(const Params& p, int hp, int jpx, const fa_launch_plan& plan, hipStream_t st) {
    constexpr int lds = pair_lds<OutT>();
    static std::atomic<bool> done[64];
    const hipError_t attr = raise_lds_limit(fwd_mfma_pair_kernel<PairA<OutT>, PairB<OutT>>, lds, done);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((fwd_mfma_pair_kernel<PairA<OutT>, PairB<OutT>>), dim3(plan.grid), dim3(plan.threads), lds, st, p, hp, jpx);
    return hipGetLastError();
}

}  // namespace

// p.nQ = 128-row query blocks per head; hp of them (the first ones) take fp16 weights; plan.grid = 8 x (a group's units, at most 2 jpx)
// workgroups of 256 threads
hipError_t launch_bf16_causal_pair_d64(const Params& p, int hp, int jpx, const fa_launch_plan& plan, int o_dtype, hipStream_t st) {
    if (o_dtype == FA_DTYPE_F32) return launch_pair<float>(p, hp, jpx, plan, st);
    if (o_dtype == FA_DTYPE_BF16) return launch_pair<__bf16>(p, hp, jpx, plan, st);
    return launch_pair<_Float16>(p, hp, jpx, plan, st);
}

int bf16_causal_pair_d64_lds_bytes(int o_dtype) {
    if (o_dtype == FA_DTYPE_F32) return pair_lds<float>();
    if (o_dtype == FA_DTYPE_BF16) return pair_lds<__bf16>();
    return pair_lds<_Float16>();
}

}  // namespace fa
