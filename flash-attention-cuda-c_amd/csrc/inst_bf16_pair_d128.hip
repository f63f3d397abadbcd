// inst_bf16_pair_d128.hip -- bf16 inputs, D = 128, small problems, with or without the causal mask (at most one 256-row unit per TWO CUs): 128-row units, one per
// workgroup of four waves, ONE workgroup per CU (a d = 128 ring leaves no room for a second): the launch occupies twice the CUs
// (kernel_bf16.hip.h: fwd_mfma_pair_kernel; one translation unit of libflash_attention.so: see launchers.hip.h).
#include "kernel_bf16.hip.h"
#include "launchers.hip.h"

namespace fa {
namespace {

// ONE configuration for both weight precisions: the four-wave form of the persistent kernels' mixed-precision kernel (KernelCfg::MIX) on their
// engine (32x32x16 under the mask, 16x16x32 without) -- K by LDS-DMA, V by LDS-DMA (bf16 weights) or as fp16 through registers (the blocks
// qb < Params::hp), exact fp32 row sums whether or not the call asks for the LSE.  (Until round 4 the fp16 blocks ran the register-staged
// 16x16x32 kernel, whose 32 staging registers at four waves spilled 31 VGPRs into its tile loop.)
template <bool CAUSAL, typename OutT>
using PairA = KernelCfg<128, CAUSAL, OutT, 2, Opt{.m16 = CAUSAL ? 0 : -1, .sum_mfma = 0, .waves = 4, .mix = true}>;
template <bool CAUSAL, typename OutT>
using PairB = PairA<CAUSAL, OutT>;

template <bool CAUSAL, typename OutT>
constexpr int pair_lds() {
    return PairA<CAUSAL, OutT>::LDS_BYTES > PairB<CAUSAL, OutT>::LDS_BYTES ? PairA<CAUSAL, OutT>::LDS_BYTES : PairB<CAUSAL, OutT>::LDS_BYTES;
}

template <bool CAUSAL, typename OutT>
hipError_t launch_pair(const Params& p, int hp, int jpx, const fa_launch_plan& plan, hipStream_t st) {
    constexpr int lds = pair_lds<CAUSAL, OutT>();
    static std::atomic<bool> done[64];
    const hipError_t attr = raise_lds_limit(fwd_mfma_pair_kernel<PairA<CAUSAL, OutT>, PairB<CAUSAL, OutT>>, lds, done);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((fwd_mfma_pair_kernel<PairA<CAUSAL, OutT>, PairB<CAUSAL, OutT>>), dim3(plan.grid), dim3(plan.threads), lds, st, p, hp, jpx);
    return hipGetLastError();
}
template <bool CAUSAL>
hipError_t by_out(const Params& p, int hp, int jpx, const fa_launch_plan& plan, int o_dtype, hipStream_t st) {
    if (o_dtype == FA_DTYPE_F32) return launch_pair<CAUSAL, float>(p, hp, jpx, plan, st);
    if (o_dtype == FA_DTYPE_BF16) return launch_pair<CAUSAL, __bf16>(p, hp, jpx, plan, st);
    return launch_pair<CAUSAL, _Float16>(p, hp, jpx, plan, st);
}
template <bool CAUSAL>
int lds_by_out(int o_dtype) {
    if (o_dtype == FA_DTYPE_F32) return pair_lds<CAUSAL, float>();
    if (o_dtype == FA_DTYPE_BF16) return pair_lds<CAUSAL, __bf16>();
    return pair_lds<CAUSAL, _Float16>();
}

}  // namespace

// p.nQ = 128-row query blocks per head; hp of them (the first ones) take fp16 weights; plan.grid = 8 x (a group's units, at most 2 jpx)
// workgroups of 256 threads
hipError_t launch_bf16_pair_d128(const Params& p, int hp, int jpx, const fa_launch_plan& plan, bool causal, int o_dtype, hipStream_t st) {
    return causal ? by_out<true>(p, hp, jpx, plan, o_dtype, st) : by_out<false>(p, hp, jpx, plan, o_dtype, st);
}

int bf16_pair_d128_lds_bytes(bool causal, int o_dtype) { return causal ? lds_by_out<true>(o_dtype) : lds_by_out<false>(o_dtype); }

}  // namespace fa
