"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, validates arguments before touching the GPU, and reports the launch plan (host logic).
No compute call is made here."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as entry

fa = entry.load_package()
ROOT = entry.ROOT


def header_functions():
    text = open(os.path.join(ROOT, "include", "flash_attention.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(flash_attention\w*)\s*\(", text))


def test_library_exports_every_declared_symbol():
    L = fa.lib()
    declared = header_functions()
    assert declared == set(fa.EXPORTS), declared ^ set(fa.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None
    assert "gfx950" in fa.version()


def test_argument_validation_happens_before_any_launch():
    L = fa.lib()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p = (p + 15) & ~15
    ok_args = dict(B=1, H=1, S=16, d=16, scale=0.25, causal=False, dtype=0, o=0)

    def call(Q=p, K=p, V=p, O=p, **kw):
        a = dict(ok_args, **kw)
        return L.flash_attention(Q, K, V, O, a["B"], a["H"], a["S"], a["d"], a["scale"], a["causal"], a["dtype"],
                                 a["o"], None)

    assert call(Q=None) == -1 and call(O=None) == -1                 # FA_ERR_NULL_POINTER
    assert call(K=p + 4) == -2                                       # FA_ERR_MISALIGNED
    assert call(S=0) == -3 and call(B=-1) == -3                      # FA_ERR_BAD_SHAPE
    assert call(d=512) == -4 and call(d=3) == -4                     # FA_ERR_UNSUPPORTED_DHEAD
    assert call(dtype=2, d=144) == -4 and call(dtype=2, d=24) == -4  # fp8 e4m3fn: d <= 128, 16-byte rows
    assert call(dtype=2, d=64, scale=-1.0) == -6                     # fp8: MFMA path only, which needs scale > 0 -> FA_ERR_BAD_SCALE
    assert call(dtype=9) == -5 and call(o=2) == -5                   # FA_ERR_UNSUPPORTED_DTYPE
    assert call(scale=float("nan")) == -6 and call(scale=float("inf")) == -6
    for code in range(-8, 1):
        assert fa.error_string(code) and "unknown flash_attention error" not in fa.error_string(code)
    assert "null" in fa.error_string(-1)


def test_strided_validation():
    L = fa.lib()
    buf = (ctypes.c_char * 4096)()
    p = (ctypes.addressof(buf) + 15) & ~15
    bad = fa.FaStrides(64, 16, 8)       # strideS < d
    good = fa.FaStrides(1024, 16, 64)
    rc = L.flash_attention_strided(p, p, p, p, 1, 1, 16, 16, 0.25, False, 0, 0, ctypes.byref(bad),
                                   ctypes.byref(good), ctypes.byref(good), ctypes.byref(good), None)
    assert rc == -7
    odd = fa.FaStrides(1024, 16, 18)    # 72-byte rows: not 16-byte aligned
    rc = L.flash_attention_strided(p, p, p, p, 1, 1, 16, 16, 0.25, False, 0, 0, ctypes.byref(odd),
                                   ctypes.byref(good), ctypes.byref(good), ctypes.byref(good), None)
    assert rc == -7


@pytest.mark.parametrize("B,H,S,d,causal,dtype,kid,br,bc", [
    (8, 16, 4096, 128, True, fa.FA_DTYPE_BF16, 1, 256, 64),    # BASELINE cfg2
    (4, 8, 2048, 64, False, fa.FA_DTYPE_BF16, 1, 256, 64),     # BASELINE cfg1
    (1, 1, 16, 16, False, fa.FA_DTYPE_F32, 3, 128, 32),        # tests/main.cu:107: d = 16 zero-padded onto the d = 64 fp32 MFMA kernel
    (1, 2, 50, 200, False, fa.FA_DTYPE_F32, 0, 32, 32),        # fp32, d > 128: generic kernel
    (1, 1, 128, 64, False, fa.FA_DTYPE_F32, 3, 128, 32),       # BASELINE cfg0's shape, exact-fp32 MFMA kernel
    (2, 4, 4096, 128, True, fa.FA_DTYPE_F32, 3, 128, 32),
    (1, 2, 200, 80, True, fa.FA_DTYPE_BF16, 1, 256, 64),       # bf16, d = 80: MFMA kernel of d = 128, rows zero-padded
    (1, 2, 200, 40, False, fa.FA_DTYPE_BF16, 1, 256, 64),      # bf16, d = 40: MFMA kernel of d = 64
    (1, 2, 200, 136, True, fa.FA_DTYPE_BF16, 0, 32, 32),       # bf16, d > 128: generic kernel
    (1, 16, 16384, 128, False, fa.FA_DTYPE_FP8_E4M3, 2, 256, 64),  # BASELINE cfg3 (fp8 e4m3fn)
    (1, 4, 500, 64, True, fa.FA_DTYPE_FP8_E4M3, 2, 256, 64),       # fp8 d = 64: zero-padded onto the d = 128 instantiation
])
def test_launch_plan(B, H, S, d, causal, dtype, kid, br, bc):
    p = fa.plan(B, H, S, d, causal, dtype, fa.FA_DTYPE_F32)
    assert p["kernel_id"] == kid and p["q_block_rows"] == br and p["kv_block_rows"] == bc
    n_q = -(-S // br)                                  # getNumCta: ceil, not the reference's assert
    units = B * H * n_q
    if kid in (1, 2):      # persistent grid: one workgroup per CU (256 on MI355X) walking ceil(units/grid) units
        assert p["grid"] == min(8 * (-(-units // 8)), 256)
    else:
        assert p["grid"] == 8 * (-(-units // 8)) and p["grid"] >= units
    assert p["threads"] % 64 == 0 and p["lds_bytes"] <= 160 * 1024
    if kid in (1, 2):
        # 3-slot ring of [K image (input type) | V image (bf16)] tiles at the instantiated head dimension; the fp32 epilogue
        # stages 256 rows x 64 floats -- behind ring slot 0 in the LDS-DMA kernels (unpadded rows: the next unit's tile 0 lands in
        # slot 0 meanwhile), from the start of the ring in the register-staged ones (padded rows)
        dk = 128 if kid == 2 or d > 64 else 64
        slot = 64 * dk * ((1 if kid == 2 else 2) + 2)
        ep_off = slot if d == dk else 0
        flags = 64                                     # behind everything: one "not finite" word per wave (kernel_bf16.hip.h: block_or)
        assert p["lds_bytes"] == max(3 * slot, ep_off + 256 * 256) + flags
        # the plan is the launched instantiation's own figure: same problem with a bf16 output
        pb = fa.plan(B, H, S, d, causal, dtype, fa.FA_DTYPE_BF16)
        assert pb["lds_bytes"] == max(3 * slot, ep_off + 256 * dk * 2) + flags


def test_no_cpu_fallback_in_binding():
    torch = pytest.importorskip("torch")
    x = torch.zeros(1, 1, 16, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fa.flash_attention(x, x, x)


def test_shard_range_matches_the_python_partition():
    """flash_attention_shard_range (C ABI) == shard.shard_heads: contiguous, exhaustive, sizes within one."""
    from flash_attention_cuda_c_amd import shard
    for total in (0, 1, 7, 128, 2048, 2049):
        for world in (1, 2, 3, 8):
            ranges = [fa.shard_range(total, r, world) for r in range(world)]
            assert ranges == [shard.shard_heads(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(fa.FlashAttentionError):
        fa.shard_range(8, 3, 2)


def test_plan_ex_describes_both_launches():
    """flash_attention_plan_ex: the launches a call makes.  bf16, d = 128, causal, S = 4096: the first FA_EARLY_KEYS / 256 = 4 query
    blocks of every head go to the fp16-weights kernel, the other 12 to the bf16-weights kernel; the flags move the split."""
    early, main = fa.plan_ex(8, 16, 4096, 4096, 128, True, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32, 0)
    assert (early["first_q_block"], early["q_blocks"], main["first_q_block"], main["q_blocks"]) == (0, 4, 4, 12)
    assert early["grid"] == 256 and main["grid"] == 256 and early["threads"] == main["threads"] == 512      # ONE launch runs both ranges
    assert early["unit_lists"] == main["unit_lists"] == 1      # one kernel, one list over all 16 query blocks per head: every unit in its block's precision
    e2, m2 = fa.plan_ex(4, 16, 8192, 8192, 128, True, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32, 0)
    assert (e2["q_blocks"], m2["q_blocks"], e2["unit_lists"], m2["unit_lists"]) == (4, 28, 1, 1)      # whatever the shape
    assert fa.plan_ex(8, 16, 4096, 4096, 128, False)[1]["unit_lists"] == 0
    assert early["lds_bytes"] == main["lds_bytes"] >= fa.plan(8, 16, 4096, 128, True, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32)["lds_bytes"]
    early, main = fa.plan_ex(8, 16, 4096, 4096, 128, True, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32, fa.FA_FLAG_BF16_WEIGHTS)
    assert (early["q_blocks"], early["grid"], main["q_blocks"]) == (0, 0, 16)
    early, main = fa.plan_ex(8, 16, 4096, 4096, 128, True, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32, fa.FA_FLAG_F16_WEIGHTS)
    assert (early["q_blocks"], main["q_blocks"], main["grid"]) == (16, 0, 0)
    # without the mask only short key sequences are "early"
    assert [x["q_blocks"] for x in fa.plan_ex(8, 16, 4096, 4096, 128, False)] == [0, 16]
    assert [x["q_blocks"] for x in fa.plan_ex(8, 16, 4096, 1000, 128, False)] == [16, 0]
    assert [x["q_blocks"] for x in fa.plan_ex(32, 16, 600, 600, 128, False)] == [3, 0]
    # small causal problems: the pair kernel -- 128-row blocks, 256 threads.  d = 64: at most one 256-row unit per CU, two workgroups per CU
    e3, m3 = fa.plan_ex(4, 8, 2048, 2048, 64, True, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32, 0)
    assert (e3["q_blocks"], m3["first_q_block"], m3["q_blocks"]) == (8, 8, 8)          # rows < 1024 = eight 128-row blocks take fp16 weights
    assert e3["q_block_rows"] == m3["q_block_rows"] == 128 and e3["threads"] == m3["threads"] == 256 and e3["grid"] == m3["grid"] == 512
    assert e3["lds_bytes"] == m3["lds_bytes"] <= 80 * 1024                              # two workgroups share a CU's 160 KiB
    assert [x["q_blocks"] for x in fa.plan_ex(1, 2, 600, 600, 64, True)] == [5, 0]
    # twice the heads: two 256-row units per CU, the persistent kernels again
    assert fa.plan(8, 8, 2048, 64, True, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32)["threads"] == 512
    assert fa.plan(4, 8, 2048, 64, False, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32)["threads"] == 512   # (BASELINE cfg1: no mask, nothing to pair)
    # d = 128: one workgroup per CU (a ring of 96 KiB), so at most one 256-row unit per two CUs -- the launch then covers twice the CUs
    e4, m4 = fa.plan_ex(1, 8, 4096, 4096, 128, True, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32, 0)
    assert (e4["q_blocks"], m4["q_blocks"], e4["q_block_rows"], e4["threads"], e4["grid"]) == (8, 24, 128, 256, 256)
    assert 80 * 1024 < e4["lds_bytes"] == m4["lds_bytes"] <= 160 * 1024
    assert [x["q_blocks"] for x in fa.plan_ex(1, 2, 600, 600, 128, True)] == [5, 0]
    assert fa.plan(1, 16, 4096, 128, True, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32)["threads"] == 512    # 256 units: one per CU, the persistent kernels
    # without the mask the units are equal: the same 128-row units only where 256-row units would leave half of the CUs idle
    e5, m5 = fa.plan_ex(1, 8, 4096, 4096, 128, False, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32, 0)
    assert (e5["q_blocks"], m5["q_blocks"], m5["q_block_rows"], m5["threads"], m5["grid"]) == (0, 32, 128, 256, 256)
    assert fa.plan(2, 8, 2048, 64, False, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32)["threads"] == 256      # 128 units of 256 rows
    assert fa.plan(4, 8, 2048, 64, False, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32)["grid"] == 256          # BASELINE cfg1: 256 units, one per CU: unchanged
    # padded head dimensions, fp32 and fp8 inputs have one form
    assert [x["q_blocks"] for x in fa.plan_ex(1, 2, 600, 600, 80, True)] == [0, 3]
    assert [x["q_blocks"] for x in fa.plan_ex(1, 2, 600, 600, 128, True, fa.FA_DTYPE_F32)][0] == 0
    L = fa.lib()
    assert L.flash_attention_plan_ex(1, 1, 64, 64, 128, False, fa.FA_DTYPE_BF16, 0, 3, None, None) == -8      # contradictory flags
    assert L.flash_attention_plan_ex(1, 1, 64, 64, 80, False, fa.FA_DTYPE_BF16, 0, 1, None, None) == -8       # no fp16-weights kernel at d = 80
    assert L.flash_attention_plan_ex(1, 1, 64, 64, 128, False, fa.FA_DTYPE_F32, 0, 2, None, None) == -8       # flag on fp32 inputs
    assert L.flash_attention_plan_ex(1, 1, 64, 64, 128, False, fa.FA_DTYPE_BF16, 0, 8, None, None) == -8      # unknown flag
