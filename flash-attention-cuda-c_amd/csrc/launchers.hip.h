// launchers.hip.h -- how the host side starts a kernel instantiation, and the per-group entry points that the translation units
// inst_*.hip define.  The kernels are spread over several translation units only so that they compile in parallel (one hipcc
// process per group); FlashAttention.hip holds the C ABI, validation and geometry and calls the group entry points below.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/flash_attention.h"
#include "loaders.hip.h"

namespace fa {

// The MFMA kernels need up to 96 KiB of dynamic LDS: above the 64 KiB default, so the limit is raised once per
// (kernel instantiation, device) -- function attributes are per device, and the multi-GPU driver calls in from one host
// thread per device.  Not a stream operation; a repeated set is harmless.
template <typename Kernel>
static hipError_t raise_lds_limit(Kernel kernel, int bytes, std::atomic<bool> (&done)[64]) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && done[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
    return e;
}

// ---- group entry points (each defined in exactly one inst_*.hip) ----
// bf16 inputs, MFMA kernel instantiated at D = 128 / 64; pad: the tensors' head dimension is smaller than D;
// lse: the instantiation that keeps the fp32 sum of the unrounded weights (computers16.hip.h)
hipError_t launch_bf16_d128(const Params& p, const fa_launch_plan& plan, bool causal, bool pad, int o_dtype, hipStream_t st);
hipError_t launch_bf16_d64(const Params& p, const fa_launch_plan& plan, bool causal, bool pad, int o_dtype, hipStream_t st);
// bf16 inputs, fp16-weights precision option (d = 128 or 64 exactly)
hipError_t launch_bf16_p16(const Params& p, const fa_launch_plan& plan, bool causal, int d, int o_dtype, hipStream_t st);
// bf16 inputs, causal, default precision, d = 128 or 64 exactly: one kernel, one (head, query block) list; the units of the first p.hp
// query blocks of every head run with fp16 weights, the rest with bf16 weights (inst_bf16_mix.hip)
hipError_t launch_bf16_causal_mix(const Params& p, const fa_launch_plan& plan, int d, int o_dtype, hipStream_t st);
int bf16_causal_mix_lds_bytes(int d, int o_dtype);
// small problems: 128-row units, one per workgroup of four waves -- causal at D = 64: two workgroups per CU, paired; else one per CU
// (inst_bf16_pair_d64.hip, inst_bf16_pair_d128.hip)
hipError_t launch_bf16_pair_d64(const Params& p, int hp, int jpx, const fa_launch_plan& plan, bool causal, int o_dtype, hipStream_t st);
hipError_t launch_bf16_pair_d128(const Params& p, int hp, int jpx, const fa_launch_plan& plan, bool causal, int o_dtype, hipStream_t st);
int bf16_pair_d64_lds_bytes(bool causal, int o_dtype);
int bf16_pair_d128_lds_bytes(bool causal, int o_dtype);
// fp8 e4m3fn inputs (always the D = 128 instantiation)
hipError_t launch_fp8_d128(const Params& p, const fa_launch_plan& plan, bool causal, bool pad, int o_dtype, hipStream_t st);
// fp32 inputs, exact-fp32 MFMA kernel
hipError_t launch_f32_d128(const Params& p, const fa_launch_plan& plan, bool causal, bool pad, int o_dtype, hipStream_t st);
hipError_t launch_f32_d64(const Params& p, const fa_launch_plan& plan, bool causal, bool pad, int o_dtype, hipStream_t st);
int f32_lds_bytes(int d_padded);
// dynamic LDS of the instantiation the bf16 / fp8 entry points above would launch (the engine, and with it the ring and epilogue
// carve-up, depends on the mask, the output type and whether the rows are padded): what flash_attention_plan reports
int bf16_d128_lds_bytes(bool causal, bool pad, int o_dtype);
int bf16_d64_lds_bytes(bool causal, bool pad, int o_dtype);
int fp8_d128_lds_bytes(bool causal, bool pad, int o_dtype);
int bf16_p16_lds_bytes(bool causal, int d, int o_dtype);

}  // namespace fa
