"""GPU parity tests: the HIP path behind the C ABI against the CPU oracle, the golden vectors
minted from the reference's check.py, and size-independent properties at the BASELINE sizes.

Tolerances (written here, justified in DESIGN.md "Tolerance"):
  fp32 inputs  (exact-fp32 kernels):  |O - ref| <= 2e-5 + 1e-4 |ref|
  bf16 inputs, fp32 O, the DEFAULT call on N(0,1) tensors:  the tolerance BASELINE.json / BASELINE.md state against check.py
               (/root/reference/check.py:19-21),  |O - ref| <= 1e-3 + 1e-3 |ref|  on EVERY element (STATED below);
  bf16 inputs, fp32 O, where the test says why: 4e-3 + 4e-3 |ref| (BF16W below) -- bf16 weights forced on every row
               (weights_dtype=torch.bfloat16), head dimensions that run zero-padded (no fp16-weights kernel: their early rows keep
               bf16 weights), fp8 inputs, or data spiked / scaled away from N(0,1) (the bound is a property of the data:
               test_sharp_softmax_parity_is_what_it_measures);
  bf16 / f16 O add the output rounding (2^-9 / 2^-11 relative).
"""
import numpy as np
import pytest

import __graft_entry__ as entry

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

fa = entry.load_package()
import oracle  # noqa: E402  (checker only)

DEV = "cuda:0"


STATED = (1e-3, 1e-3)     # what north_star states; the default call on N(0,1) data is held to it element-wise
BF16W = (4e-3, 4e-3)      # bf16 weights on rows that see few keys / data away from N(0,1): each use says which


def tol_for(in_dtype, out_dtype, weights=None, padded=False):
    """(atol, rtol) of a call: fp32 inputs exact; bf16 / fp8 inputs at the default precision on randn data the STATED tolerance,
    with bf16 weights forced (weights=torch.bfloat16), a zero-padded head dimension or fp8 inputs BF16W; plus the output's rounding."""
    if in_dtype == torch.float32:
        atol, rtol = 2e-5, 1e-4
    elif weights == torch.bfloat16 or padded or in_dtype not in (torch.bfloat16,):
        atol, rtol = BF16W
    else:
        atol, rtol = STATED
    if out_dtype == torch.bfloat16:
        atol, rtol = atol + 4e-3, rtol + 4e-3
    if out_dtype == torch.float16:
        atol, rtol = atol + 5e-4, rtol + 1e-3
    return atol, rtol


def randn(shape, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32).to(dtype)


def run_gpu(Q, K, V, causal, out_dtype=torch.float32, scale=None):
    O = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, out_dtype=out_dtype, scale=scale)
    torch.cuda.synchronize()
    return O.float().cpu().numpy()


def check(O, ref, atol, rtol, rms=None):
    assert np.isfinite(O).all()
    err = np.abs(O - ref)
    bad = err > atol + rtol * np.abs(ref)
    assert not bad.any(), f"{bad.sum()} / {bad.size} outside tolerance, max abs err {err.max():.3e}"
    if rms is not None:
        assert np.sqrt(np.mean(err ** 2)) <= rms


# ------------------------------------------------------------------ reference known-answer cases
def test_known_answer_maincu_all_ones():
    """tests/main.cu:24-36,107: B=H=1, S=16, d=16, fp32 all-ones, scale 1/sqrt(16) -> O == 1."""
    one = torch.ones(1, 1, 16, 16)
    for causal in (False, True):
        np.testing.assert_allclose(run_gpu(one, one, one, causal), 1.0, atol=1e-6)


def test_known_answer_checkpy_demo(golden):
    """check.py:30-43 through the check.py-shaped API: ones (1,4,8), H=2 -> output == 1."""
    one = torch.from_numpy(golden.load("F1", "Q")).to(DEV)
    out, attn = fa.multi_head_attention(one, one, one, 2)
    torch.cuda.synchronize()
    assert attn is None
    np.testing.assert_allclose(out.cpu().numpy(), golden.load("F1", "out"), atol=1e-6)


def test_attn_matrix_against_checkpy(golden):
    """check.py:25 returns (output, attn).  attn rebuilt on the GPU from the fused kernel's LSE against the
    matrices check.py itself produced: F1 (demo, attn == 0.25), F3 (two heads, random), F0 (row sums == 1)."""
    one = torch.from_numpy(golden.load("F1", "Q")).to(DEV)
    out, attn = fa.multi_head_attention(one, one, one, 2, return_attn=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(attn.cpu().numpy(), golden.load("F1", "attn"), atol=1e-6)       # check.py:42 prints this
    np.testing.assert_allclose(out.cpu().numpy(), golden.load("F1", "out"), atol=1e-6)
    Q, K, V = (torch.from_numpy(golden.load("F3", k)).to(DEV) for k in "QKV")
    out, attn = fa.multi_head_attention(Q, K, V, 2, return_attn=True)
    torch.cuda.synchronize()
    assert attn.shape == golden.load("F3", "attn").shape
    np.testing.assert_allclose(attn.cpu().numpy(), golden.load("F3", "attn"), rtol=2e-4, atol=1e-7)
    check(out.cpu().numpy(), golden.load("F3", "out"), 2e-5, 1e-4)
    Q, K, V = (torch.from_numpy(golden.load("F0", k)).to(DEV) for k in "QKV")
    out, attn = fa.multi_head_attention(Q, K, V, 1, return_attn=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(attn.sum(-1).cpu().numpy(), golden.load("F0", "attn_rowsum"), atol=2e-5)
    # attn @ V reproduces the fused kernel's output (check.py:21)
    check(torch.matmul(attn, V.view(1, 128, 1, 64).transpose(1, 2)).transpose(1, 2).reshape(1, 128, 64).cpu().numpy(),
          out.cpu().numpy(), 2e-5, 1e-4)


def test_attn_matrix_causal_and_rectangular():
    """Causal (k > q entries exactly 0) and seqLenQ != seqLenK, bf16 inputs, against the float64 oracle."""
    B, H, Sq, Sk, d = 1, 2, 70, 150, 128
    Q, K, V = randn((B, H, Sq, d), 50, torch.bfloat16), randn((B, H, Sk, d), 51, torch.bfloat16), randn((B, H, Sk, d), 52, torch.bfloat16)
    for causal in (False, True):
        O, lse = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, out_dtype=torch.float32, return_lse=True)
        P = fa.attention_weights(Q.to(DEV), K.to(DEV), lse, is_causal=causal)
        torch.cuda.synchronize()
        s = Q.double().numpy() @ K.double().numpy().swapaxes(-1, -2) / np.sqrt(d)
        if causal:
            s = np.where(np.triu(np.ones((Sq, Sk), dtype=bool), 1), -np.inf, s)
        ref = np.exp(s - s.max(-1, keepdims=True)); ref /= ref.sum(-1, keepdims=True)
        np.testing.assert_allclose(P.cpu().numpy(), ref, rtol=3e-3, atol=1e-6)    # LSE carries the bf16-P row-sum error
        if causal:
            assert (P.cpu().numpy()[..., np.triu(np.ones((Sq, Sk), dtype=bool), 1)] == 0).all()


def test_attn_matrix_largest_head_dim():
    """dHead = 256: the generic forward kernel and the weights kernel both take > 64 KiB of dynamic LDS."""
    B, H, S, d = 1, 2, 150, 256
    Q, K, V = (randn((B, H, S, d), s, torch.float32) for s in (53, 54, 55))
    O, lse = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=True, return_lse=True)
    P = fa.attention_weights(Q.to(DEV), K.to(DEV), lse, is_causal=True)
    torch.cuda.synchronize()
    check(O.cpu().numpy(), oracle.attention(Q.numpy(), K.numpy(), V.numpy(), causal=True), 2e-5, 1e-4)
    np.testing.assert_allclose(P.sum(-1).cpu().numpy(), 1.0, atol=2e-5)
    check(torch.matmul(P, V.to(DEV)).cpu().numpy(), O.cpu().numpy(), 2e-5, 1e-4)


# ------------------------------------------------------------------ golden vectors (from check.py)
@pytest.mark.parametrize("name", ["F0", "F3", "F4", "F6"])
def test_golden_fp32_through_checkpy_api(golden, name):
    """(B,S,H*d_k) tensors in, strided kernel launch, against check.py's recorded output."""
    meta = golden.meta(name)
    Q, K, V = (torch.from_numpy(golden.load(name, k)).to(DEV) for k in "QKV")
    out, _ = fa.multi_head_attention(Q, K, V, meta["num_heads"], is_causal=meta["causal"])
    torch.cuda.synchronize()
    atol, rtol = (2e-4, 2e-4) if name == "F6" else (2e-5, 1e-4)
    check(out.cpu().numpy(), golden.load(name, "out"), atol, rtol)


def test_golden_bf16_inputs(golden):
    """F5: check.py on bf16-rounded inputs; the MFMA kernel gets the same values as bf16.
    d=64, S=128 (BASELINE cfg0's shape) on the bf16 path."""
    Q, K, V = (torch.from_numpy(golden.load("F5bf16", k)).to(torch.bfloat16) for k in "QKV")
    for t, k in zip((Q, K, V), "QKV"):
        np.testing.assert_array_equal(t.float().numpy(), golden.load("F5bf16", k))   # exact in bf16
    out, _ = fa.multi_head_attention(Q.to(DEV), K.to(DEV), V.to(DEV), 1, out_dtype=torch.float32)
    torch.cuda.synchronize()
    check(out.cpu().numpy(), golden.load("F5bf16", "out"), *STATED, rms=1e-3)


def test_golden_layout_bf16_two_heads(golden):
    """F3 pins (B,S,H*d_k) <-> [B,H,S,d] (check.py:14-16,24) on the MFMA path (d_k = 64)."""
    Q, K, V = (torch.from_numpy(golden.load("F3", k)).to(torch.bfloat16) for k in "QKV")
    ref, _ = oracle.multi_head_attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), 2)
    out, _ = fa.multi_head_attention(Q.to(DEV), K.to(DEV), V.to(DEV), 2, out_dtype=torch.float32)
    torch.cuda.synchronize()
    check(out.cpu().numpy(), ref, *STATED, rms=1e-3)


# ------------------------------------------------------------------ random parity vs the oracle
FP32_CASES = [(1, 1, 128, 64, False), (2, 3, 77, 40, True), (1, 2, 300, 256, False), (1, 1, 1, 16, True),
              (1, 2, 33, 8, True), (3, 1, 65, 100, False),
              # exact-fp32 MFMA kernel (d in {64,128})
              (1, 2, 128, 128, False), (2, 2, 1000, 128, True), (1, 3, 333, 64, True), (1, 1, 1, 128, True),
              (1, 2, 2048, 64, False), (1, 1, 31, 128, False)]


@pytest.mark.parametrize("B,H,S,d,causal", FP32_CASES)
def test_fp32_path_matches_oracle(B, H, S, d, causal):
    Q, K, V = (randn((B, H, S, d), s, torch.float32) for s in (1, 2, 3))
    ref = oracle.attention(Q.numpy(), K.numpy(), V.numpy(), causal=causal)
    check(run_gpu(Q, K, V, causal), ref, *tol_for(torch.float32, torch.float32))


BF16_CASES = [(1, 1, 64, 128, False), (1, 2, 256, 128, False), (2, 2, 512, 128, True), (1, 3, 1000, 128, True),
              (1, 3, 333, 64, False), (1, 1, 1, 128, True), (1, 2, 63, 64, True), (1, 2, 65, 128, False),
              (2, 1, 257, 128, True), (1, 1, 2048, 64, True), (1, 2, 200, 80, True), (1, 9, 320, 128, False),
              (1, 2, 150, 136, True), (1, 1, 70, 256, False)]      # d > 128: generic kernel on bf16 inputs (exact fp32 math)


@pytest.mark.parametrize("B,H,S,d,causal", BF16_CASES)
def test_bf16_path_matches_oracle(B, H, S, d, causal):
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16) for s in (4, 5, 6))
    ref = oracle.attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=causal)
    # d <= 128: MFMA path, default precision; d = 80: zero-padded onto the d = 128 instantiation, whose early rows keep bf16 weights
    atol, rtol = tol_for(torch.bfloat16 if d <= 128 else torch.float32, torch.float32, padded=d not in (64, 128))
    check(run_gpu(Q, K, V, causal), ref, atol, rtol, rms=1e-3)


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float16])
def test_low_precision_outputs(out_dtype):
    Q, K, V = (randn((2, 2, 512, 128), s, torch.bfloat16) for s in (7, 8, 9))
    ref = oracle.attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=True)
    check(run_gpu(Q, K, V, True, out_dtype=out_dtype), ref, *tol_for(torch.bfloat16, out_dtype))


def test_causal_edges():
    """Row 0 sees one key (O[0] == V[0]); tile-boundary rows; a fully masked tile is skipped, never NaN
    (reference defect D3)."""
    Q, K, V = (randn((1, 2, 512, 128), s, torch.bfloat16) for s in (10, 11, 12))
    O = run_gpu(Q, K, V, True)
    np.testing.assert_allclose(O[:, :, 0], V.float().numpy()[:, :, 0], rtol=1e-6, atol=1e-6)
    ref = oracle.attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=True)
    for r in (31, 32, 63, 64, 255, 256, 511):
        check(O[:, :, r], ref[:, :, r], *STATED)


def test_online_softmax_rescale_is_forced():
    """Spike one key against one query so the row max jumps by far more than the lazy-rescale
    threshold in the middle of the sequence (guide rule: a rare data-dependent branch needs its own test)."""
    B, H, S, d = 1, 2, 1024, 128
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16) for s in (13, 14, 15))
    for row, key in ((3, S // 2 + 5), (700, 130), (1023, 1000)):
        K[:, :, key] = (6.0 * Q[:, :, row].float()).to(torch.bfloat16)
    for causal in (False, True):
        ref = oracle.attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=causal)
        check(run_gpu(Q, K, V, causal), ref, *BF16W, rms=1e-3)      # (spiked data: a row's mass on ONE key, its rounding does not average out)


def test_large_score_range_no_overflow():
    """Scores of magnitude ~ +-300: exp must be taken relative to the running max."""
    Q, K, V = (randn((1, 1, 256, 128), s, torch.bfloat16) for s in (16, 17, 18))
    Q, K = Q * 6, K * 6
    ref = oracle.attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=False)
    check(run_gpu(Q, K, V, False), ref, 8e-3, 8e-3)


@pytest.mark.parametrize("d,causal", [(128, True), (64, False)])
def test_optimistic_pass_falls_back_to_tracked_max(d, causal):
    """The bf16 kernel first runs an optimistic pass (exponentials relative to the row max of the FIRST
    key tile, no per-tile max).  Q, K x12 makes later tiles exceed that reference by more than 2^127, so
    the pass produces inf/NaN, its finiteness check fires and the workgroup recomputes with max tracking.
    A second case keeps the excess below 2^127 (P up to ~2^100 in the optimistic pass, no fallback)."""
    B, H, S = 1, 3, 1536
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16) for s in (50, 51, 52))
    for mul in (12.0, 4.0):
        Qm, Km = (Q.float() * mul).to(torch.bfloat16), (K.float() * mul).to(torch.bfloat16)
        ref = oracle.attention(Qm.float().numpy(), Km.float().numpy(), V.float().numpy(), causal=causal)
        check(run_gpu(Qm, Km, V, causal), ref, 1.6e-2, 1.6e-2)
    # spike far beyond 2^127 at one (row, key) pair in the middle of the sequence
    K2 = K.clone()
    K2[:, :, 900] = (12.0 * Q[:, :, 1000].float()).to(torch.bfloat16)
    ref = oracle.attention(Q.float().numpy(), K2.float().numpy(), V.float().numpy(), causal=causal)
    check(run_gpu(Q, K2, V, causal), ref, 4e-3, 4e-3, rms=1e-3)


FP8 = getattr(torch, "float8_e4m3fn", None)


@pytest.mark.skipif(FP8 is None, reason="torch build without float8_e4m3fn")
@pytest.mark.parametrize("B,H,S,causal", [(1, 1, 64, False), (2, 2, 512, True), (1, 3, 1000, False), (1, 2, 1024, True),
                                          (1, 1, 1, True), (1, 2, 4096, False)])
def test_fp8_e4m3_inputs_match_oracle(B, H, S, causal):
    """BASELINE cfg3's dtype: OCP e4m3fn Q/K/V (d = 128).  QK^T runs on v_mfma_f32_32x32x16_fp8_fp8, V is
    widened to bf16 exactly, so the only rounding inside the kernel is P -> bf16: same tolerance as bf16."""
    d = 128
    Q, K, V = (randn((B, H, S, d), s, torch.float32).to(FP8) for s in (60, 61, 62))
    ref = oracle.attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=causal)
    np.testing.assert_array_equal(oracle.round_e4m3fn(Q.float().numpy()), Q.float().numpy())   # same e4m3fn grid
    check(run_gpu(Q, K, V, causal), ref, 4e-3, 4e-3, rms=1e-3)
    O = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal)     # default output: bf16
    torch.cuda.synchronize()
    assert O.dtype == torch.bfloat16
    check(O.float().cpu().numpy(), ref, 8e-3, 8e-3)


@pytest.mark.skipif(FP8 is None, reason="torch build without float8_e4m3fn")
def test_fp8_golden_inputs(golden):
    """F5e4m3: check.py on inputs rounded to float8_e4m3fn by torch (d = 64 there, so the 128-wide kernel
    cannot take it directly): two heads are concatenated into one d = 128 head whose second half of Q is
    zero -- scores and therefore the first 64 output columns are unchanged."""
    Q, K, V = (torch.from_numpy(golden.load("F5e4m3", k)) for k in "QKV")        # (1,128,64)
    z = torch.zeros_like(Q)
    Q2, K2, V2 = torch.cat([Q, z], -1), torch.cat([K, K], -1), torch.cat([V, V], -1)
    scale = 1.0 / 64 ** 0.5
    O = fa.flash_attention(Q2.to(FP8).to(DEV)[:, None], K2.to(FP8).to(DEV)[:, None], V2.to(FP8).to(DEV)[:, None],
                           scale=scale, out_dtype=torch.float32)
    torch.cuda.synchronize()
    check(O[0, 0, :, :64].cpu().numpy(), golden.load("F5e4m3", "out")[0], 4e-3, 4e-3, rms=1e-3)


@pytest.mark.parametrize("dtype,d,S,causal", [(torch.bfloat16, 128, 777, True), (torch.bfloat16, 64, 512, False),
                                              (torch.float32, 40, 100, True), (torch.bfloat16, 128, 1536, False)])
def test_lse_output(dtype, d, S, causal):
    """flash_attention_lse: LSE[b,h,q] = ln sum_k exp(scale*q.k) (the L/M statistic of the reference's
    commented-out first API, kernels/FlashAttention.cuh:21).  Also through the optimistic pass with a
    large reference offset (Q, K x4) and its fallback (x12)."""
    B, H = 2, 2
    Q, K, V = (randn((B, H, S, d), s, dtype) for s in (70, 71, 72))
    for mul in ((1.0, 4.0, 12.0) if dtype == torch.bfloat16 else (1.0,)):
        Qm, Km = (Q.float() * mul).to(dtype), (K.float() * mul).to(dtype)
        O, lse = fa.flash_attention(Qm.to(DEV), Km.to(DEV), V.to(DEV), is_causal=causal, out_dtype=torch.float32,
                                    return_lse=True)
        torch.cuda.synchronize()
        ref = oracle.lse_numpy(Qm.float().numpy(), Km.float().numpy(), causal=causal)
        np.testing.assert_allclose(lse.cpu().numpy(), ref, rtol=2e-5, atol=2e-4 * mul * mul)
        refO = oracle.attention(Qm.float().numpy(), Km.float().numpy(), V.float().numpy(), causal=causal)
        check(O.cpu().numpy(), refO, 1.6e-2, 1.6e-2)


def test_fp32_mfma_path_spike_and_outputs():
    """fp32 inputs, d = 128: the f32-input MFMA kernel keeps fp32 end to end.  Forced rescale (spike), large
    score range, bf16 output and LSE."""
    B, H, S, d = 1, 2, 768, 128
    Q, K, V = (randn((B, H, S, d), s, torch.float32) for s in (90, 91, 92))
    K[:, :, 500] = 6.0 * Q[:, :, 3]
    K[:, :, 40] = 12.0 * Q[:, :, 700]
    for causal in (False, True):
        ref = oracle.attention(Q.numpy(), K.numpy(), V.numpy(), causal=causal)
        O, lse = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, return_lse=True)
        torch.cuda.synchronize()
        check(O.cpu().numpy(), ref, 2e-5, 1e-4)
        np.testing.assert_allclose(lse.cpu().numpy(), oracle.lse_numpy(Q.numpy(), K.numpy(), causal=causal), rtol=2e-6, atol=2e-4)
        check(run_gpu(Q, K, V, causal, out_dtype=torch.bfloat16), ref, 8e-3, 8e-3)
    # scores 36x larger (sigma 36, spikes in the thousands): an fp32 score carries ~|s|*2^-24*sqrt(d) of rounding
    # noise which exp() turns into relative error, so the bound is looser here -- the same allowance the
    # peaky golden fixture F6 gets in test_oracle.py
    Q6, K6 = Q * 6, K * 6
    check(run_gpu(Q6, K6, V, False), oracle.attention(Q6.numpy(), K6.numpy(), V.numpy()), 2e-4, 2e-4)


CROSS_CASES = [  # dtype, B, H, Sq, Sk, d, causal
    (torch.bfloat16, 1, 2, 1, 1000, 128, False),      # decode: one query against a key/value cache
    (torch.bfloat16, 2, 2, 300, 1111, 128, False),
    (torch.bfloat16, 1, 2, 700, 130, 128, True),      # more queries than keys: rows q >= Sk see every key
    (torch.bfloat16, 1, 3, 100, 517, 64, True),       # top-left mask: only keys 0..q
    (torch.float32, 1, 2, 260, 90, 128, True),        # fp32 MFMA kernel
    (torch.float32, 2, 1, 33, 400, 64, False),
    (torch.float32, 1, 2, 50, 77, 40, True),          # generic kernel
    (torch.bfloat16, 1, 2, 64, 200, 80, False),       # d = 80 padded onto the d = 128 MFMA kernel
    (torch.bfloat16, 1, 2, 40, 90, 160, True),        # generic kernel, bf16 inputs
]


@pytest.mark.parametrize("dtype,B,H,Sq,Sk,d,causal", CROSS_CASES)
def test_cross_lengths(dtype, B, H, Sq, Sk, d, causal):
    """seqLenQ != seqLenK (the reference's first API, kernels/FlashAttention.cuh:23, never live there: the
    reference holds no fixture for it, so parity here is against the oracle's float64 restatement of the same
    formula with the k > q mask of kernels/utils.cuh:43 on absolute indices).  O and LSE."""
    Q = randn((B, H, Sq, d), 60, dtype)
    K = randn((B, H, Sk, d), 61, dtype)
    V = randn((B, H, Sk, d), 62, dtype)
    Qf, Kf, Vf = (t.float().numpy() for t in (Q, K, V))
    ref = oracle.attention_numpy(Qf, Kf, Vf, causal=causal)
    O, lse = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, out_dtype=torch.float32, return_lse=True)
    torch.cuda.synchronize()
    atol, rtol = tol_for(dtype, torch.float32)
    check(O.cpu().numpy(), ref, atol, rtol)
    np.testing.assert_allclose(lse.cpu().numpy(), oracle.lse_numpy(Qf, Kf, causal=causal), rtol=2e-6,
                               atol=2e-4 if dtype == torch.float32 else 2e-3)


@pytest.mark.skipif(FP8 is None, reason="torch build without float8_e4m3fn")
def test_cross_lengths_fp8_and_strided_cache():
    """fp8 inputs with Sq < Sk, and K/V given as a strided window of a longer cache buffer."""
    B, H, Sq, Sk, d = 1, 2, 130, 900, 128
    Q = randn((B, H, Sq, d), 70, torch.float32).to(FP8)
    cache_k = randn((B, H, 1024, d), 71, torch.float32).to(FP8)
    cache_v = randn((B, H, 1024, d), 72, torch.float32).to(FP8)
    Kd, Vd = cache_k.to(DEV)[:, :, :Sk], cache_v.to(DEV)[:, :, :Sk]     # views: head stride 1024*d
    assert not Kd.is_contiguous()
    O = fa.flash_attention(Q.to(DEV), Kd, Vd, out_dtype=torch.float32)
    torch.cuda.synchronize()
    ref = oracle.attention_numpy(Q.float().numpy(), cache_k[:, :, :Sk].float().numpy(), cache_v[:, :, :Sk].float().numpy())
    check(O.cpu().numpy(), ref, 4e-3, 4e-3, rms=1e-3)


def test_cross_equals_self_attention_on_the_same_prefix():
    """Causal self-attention rows [0, n) depend only on keys [0, n): running the first n queries against all
    Sk keys (cross entry point, causal) must reproduce the first n rows of the square problem bit for bit
    when n is a multiple of the query block."""
    B, H, S, d, n = 1, 2, 1024, 128, 512
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16).to(DEV) for s in (80, 81, 82))
    full = fa.flash_attention(Q, K, V, is_causal=True, out_dtype=torch.float32)
    part = fa.flash_attention(Q[:, :, :n].contiguous(), K, V, is_causal=True, out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert torch.equal(part, full[:, :, :n])


def test_sharded_entry_point_single_device():
    """flash_attention_sharded with every "rank" on cuda:0 (the box has one GPU): 3 slabs of a B=2, H=5 problem
    on 3 streams reproduce the unsharded call bit for bit."""
    B, H, S, d = 2, 5, 300, 128
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16).to(DEV) for s in (110, 111, 112))
    full = fa.flash_attention(Q, K, V, is_causal=True)
    n = 3
    flat = lambda t: t.view(B * H, S, d)
    rng = [fa.shard_range(B * H, r, n) for r in range(n)]
    Qs, Ks, Vs = ([flat(t)[lo:hi].contiguous() for lo, hi in rng] for t in (Q, K, V))
    Os = [torch.empty_like(q) for q in Qs]
    streams = [torch.cuda.Stream() for _ in range(n)]
    torch.cuda.synchronize()
    fa.flash_attention_sharded(Qs, Ks, Vs, Os, B, H, is_causal=True, streams=streams)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(Os), flat(full))


def _fuzz_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        dtype = [torch.bfloat16, torch.float32, "fp8"][int(rng.integers(0, 3))]
        d = 128 if dtype == "fp8" else int(rng.choice([64, 128, 128, 32, 80, 256] if dtype == torch.float32 else [64, 128, 128, 80]))
        B, H = int(rng.integers(1, 4)), int(rng.integers(1, 6))
        Sq = int(rng.choice([1, 7, 63, 64, 65, 200, 256, 257, 511, 700, 1030]))
        Sk = Sq if rng.random() < 0.6 else int(rng.choice([1, 5, 64, 129, 300, 1000]))
        cases.append((i, dtype, B, H, Sq, Sk, d, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)),
                      [torch.float32, torch.bfloat16][int(rng.integers(0, 2))]))
    return cases


@pytest.mark.parametrize("i,dtype,B,H,Sq,Sk,d,causal,strided,out_dtype", _fuzz_cases(36, 2024))
def test_fuzz_shapes_layouts_dtypes(i, dtype, B, H, Sq, Sk, d, causal, strided, out_dtype):
    """Seeded sweep over (dtype, B, H, Sq, Sk, d, causal, model-layout strides, output dtype): every dispatch
    branch (bf16 / fp8 / fp32 MFMA kernels, generic kernel, LDS and direct epilogues, cross lengths, ragged
    tails) against the float64 oracle on the same rounded inputs."""
    if dtype == "fp8":
        if FP8 is None:
            pytest.skip("torch build without float8_e4m3fn")
        dtype = FP8
    mk = lambda S, seed: randn((B, S, H * d), seed, torch.float32).to(dtype)         # (B, S, H*d) model layout
    Qm, Km, Vm = mk(Sq, 1000 + 3 * i), mk(Sk, 1001 + 3 * i), mk(Sk, 1002 + 3 * i)
    view = lambda t, S: t.view(B, S, H, d).transpose(1, 2)                            # [B, H, S, d] strided view
    Qd, Kd, Vd = view(Qm.to(DEV), Sq), view(Km.to(DEV), Sk), view(Vm.to(DEV), Sk)
    if not strided:
        Qd, Kd, Vd = Qd.contiguous(), Kd.contiguous(), Vd.contiguous()
    O, lse = fa.flash_attention(Qd, Kd, Vd, is_causal=causal, out_dtype=out_dtype, return_lse=True)
    torch.cuda.synchronize()
    f = lambda t, S: view(t, S).float().numpy()
    ref = oracle.attention_numpy(f(Qm, Sq), f(Km, Sk), f(Vm, Sk), causal=causal)
    # bf16 at d in {64, 128}: the default precision; other d run zero-padded (bf16 weights on the early rows), d > 128 the exact generic kernel
    atol, rtol = tol_for(torch.float32 if (dtype == torch.float32 or d > 128) else torch.bfloat16, out_dtype, padded=d not in (64, 128) or dtype == FP8)   # (fp8 inputs: one weight precision, bf16)
    check(O.float().cpu().numpy(), ref, atol, rtol)
    np.testing.assert_allclose(lse.cpu().numpy(), oracle.lse_numpy(f(Qm, Sq), f(Km, Sk), causal=causal), rtol=2e-6,
                               atol=2e-4 if dtype == torch.float32 else 3e-3)


@pytest.mark.parametrize("d", [8, 16, 24, 40, 56, 72, 80, 96, 104, 120])
@pytest.mark.parametrize("causal", [False, True])
def test_bf16_head_dims_padded_onto_the_mfma_kernel(d, causal):
    """bf16 with d not in {64,128} (multiples of 8): runs the d = 64 / 128 MFMA instantiation with rows zero-padded
    on the fly.  Model-layout strides (rows of other heads right behind each row, the tensor ending right behind the
    last head's last row), ragged S, cross lengths, bf16 and fp32 outputs, LSE."""
    B, H, Sq, Sk = 2, 3, 333, 400 if causal else 333
    mk = lambda S, seed: randn((B, S, H * d), seed, torch.bfloat16)
    Qm, Km, Vm = mk(Sq, 120 + d), mk(Sk, 121 + d), mk(Sk, 122 + d)
    view = lambda t, S: t.view(B, S, H, d).transpose(1, 2)
    f = lambda t, S: view(t, S).float().numpy()
    ref = oracle.attention_numpy(f(Qm, Sq), f(Km, Sk), f(Vm, Sk), causal=causal)
    for out_dtype in (torch.float32, torch.bfloat16):
        for strided in (True, False):
            Qd, Kd, Vd = view(Qm.to(DEV), Sq), view(Km.to(DEV), Sk), view(Vm.to(DEV), Sk)
            if not strided:
                Qd, Kd, Vd = Qd.contiguous(), Kd.contiguous(), Vd.contiguous()
            O, lse = fa.flash_attention(Qd, Kd, Vd, is_causal=causal, out_dtype=out_dtype, return_lse=True)
            torch.cuda.synchronize()
            atol, rtol = tol_for(torch.bfloat16, out_dtype, padded=True)     # (no fp16-weights kernel for zero-padded head dimensions)
            check(O.float().cpu().numpy(), ref, atol, rtol)
            np.testing.assert_allclose(lse.cpu().numpy(), oracle.lse_numpy(f(Qm, Sq), f(Km, Sk), causal=causal), rtol=2e-6, atol=3e-3)
    assert fa.plan(B, H, Sq, d, causal, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32)["kernel_id"] == 1


@pytest.mark.skipif(FP8 is None, reason="torch build without float8_e4m3fn")
@pytest.mark.parametrize("d", [16, 48, 64, 96, 112])
def test_fp8_head_dims_padded_onto_the_mfma_kernel(d):
    """fp8 e4m3fn with d < 128 (multiples of 16): the d = 128 instantiation (MX QK^T) with zero-padded rows."""
    B, H, Sq, Sk = 2, 3, 300, 513
    mk = lambda S, seed: randn((B, S, H * d), seed, torch.float32).to(FP8)
    Qm, Km, Vm = mk(Sq, 160 + d), mk(Sk, 161 + d), mk(Sk, 162 + d)
    view = lambda t, S: t.view(B, S, H, d).transpose(1, 2)
    f = lambda t, S: view(t, S).float().numpy()
    for causal in (False, True):
        ref = oracle.attention_numpy(f(Qm, Sq), f(Km, Sk), f(Vm, Sk), causal=causal)
        for strided in (True, False):
            Qd, Kd, Vd = view(Qm.to(DEV), Sq), view(Km.to(DEV), Sk), view(Vm.to(DEV), Sk)
            if not strided:
                Qd, Kd, Vd = Qd.contiguous(), Kd.contiguous(), Vd.contiguous()
            O = fa.flash_attention(Qd, Kd, Vd, is_causal=causal, out_dtype=torch.float32)
            torch.cuda.synchronize()
            check(O.cpu().numpy(), ref, 4e-3, 4e-3, rms=1e-3)


@pytest.mark.parametrize("d", [4, 16, 36, 60, 68, 100, 124])
def test_fp32_head_dims_padded_onto_the_fp32_mfma_kernel(d):
    """fp32 with d not in {64,128} (multiples of 4, e.g. the reference's own d = 16 of tests/main.cu:107): the exact
    fp32 MFMA kernel with zero-padded rows; same tolerance as the native head dimensions."""
    B, H, Sq, Sk = 2, 2, 150, 201
    mk = lambda S, seed: randn((B, S, H * d), seed, torch.float32)
    Qm, Km, Vm = mk(Sq, 150 + d), mk(Sk, 151 + d), mk(Sk, 152 + d)
    view = lambda t, S: t.view(B, S, H, d).transpose(1, 2)
    f = lambda t, S: view(t, S).numpy()
    for causal in (False, True):
        ref = oracle.attention_numpy(f(Qm, Sq), f(Km, Sk), f(Vm, Sk), causal=causal)
        for strided in (True, False):
            Qd, Kd, Vd = view(Qm.to(DEV), Sq), view(Km.to(DEV), Sk), view(Vm.to(DEV), Sk)
            if not strided:
                Qd, Kd, Vd = Qd.contiguous(), Kd.contiguous(), Vd.contiguous()
            O, lse = fa.flash_attention(Qd, Kd, Vd, is_causal=causal, return_lse=True)
            torch.cuda.synchronize()
            check(O.cpu().numpy(), ref, 2e-5, 1e-4)
            np.testing.assert_allclose(lse.cpu().numpy(), oracle.lse_numpy(f(Qm, Sq), f(Km, Sk), causal=causal), rtol=2e-6, atol=2e-4)
    assert fa.plan(B, H, Sq, d, False, fa.FA_DTYPE_F32, fa.FA_DTYPE_F32)["kernel_id"] == 3


def test_padded_output_rows_are_not_overrun():
    """d = 72 into an O buffer whose rows are exactly 72 elements, with a guard band behind every row of a wider
    allocation: the padded kernel must not write past column d."""
    B, H, S, d, W = 1, 2, 130, 72, 128
    for in_dtype, out_dtype in ((torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float32, torch.float32),
                                (torch.float32, torch.bfloat16)):
        Q, K, V = (randn((B, H, S, d), s, in_dtype).to(DEV) for s in (140, 141, 142))
        buf = torch.full((B, H, S, W), 7.0, dtype=out_dtype, device=DEV)
        O = buf[..., :d]                                    # row stride W, d columns
        fa.flash_attention(Q, K, V, O=O, is_causal=True)
        torch.cuda.synchronize()
        assert bool((buf[..., d:] == 7.0).all())
        ref = oracle.attention(Q.float().cpu().numpy(), K.float().cpu().numpy(), V.float().cpu().numpy(), causal=True)
        atol, rtol = tol_for(in_dtype, out_dtype, padded=True)      # d = 72: zero-padded instantiation
        check(O.float().cpu().numpy(), ref, atol, rtol)


def test_heads_are_independent():
    """Reference defect D2 (every query attends to every batch/head) must not be reproduced:
    a head computed alone equals the same head computed inside a batch, bit for bit."""
    Q, K, V = (randn((2, 3, 256, 128), s, torch.bfloat16) for s in (19, 20, 21))
    full = run_gpu(Q, K, V, True)
    for b in range(2):
        for h in range(3):
            one = run_gpu(Q[b:b + 1, h:h + 1], K[b:b + 1, h:h + 1], V[b:b + 1, h:h + 1], True)
            np.testing.assert_array_equal(one[0, 0], full[b, h])


def test_custom_scale_and_nonpositive_scale_route():
    Q, K, V = (randn((1, 2, 192, 64), s, torch.bfloat16) for s in (22, 23, 24))
    for scale in (0.5, 0.03, -0.2, 0.0):   # scale <= 0 takes the generic kernel
        ref = oracle.attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), scale=scale)
        check(run_gpu(Q, K, V, False, scale=scale), ref, *BF16W)   # (scale 0.5 = 4 / sqrt(d): a sharp softmax; scale <= 0: exact fp32 math)


def test_every_output_element_is_written():
    Q, K, V = (randn((1, 2, 300, 128), s, torch.bfloat16).to(DEV) for s in (25, 26, 27))
    O = torch.full((1, 2, 300, 128), float("nan"), device=DEV)
    fa.flash_attention(Q, K, V, O, is_causal=True)
    torch.cuda.synchronize()
    assert torch.isfinite(O).all()


def test_runs_on_a_side_stream_and_is_deterministic():
    Q, K, V = (randn((2, 4, 1024, 128), s, torch.bfloat16).to(DEV) for s in (28, 29, 30))
    a = fa.flash_attention(Q, K, V, is_causal=True)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        b = fa.flash_attention(Q, K, V, is_causal=True)
    st.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_call_is_graph_capturable():
    """The launcher allocates nothing and never synchronises (include/flash_attention.h contract), so a call
    can be captured into a HIP graph and replayed on new data."""
    Q, K, V = (randn((2, 4, 512, 128), s, torch.bfloat16).to(DEV) for s in (80, 81, 82))
    O = torch.empty_like(Q)
    fa.flash_attention(Q, K, V, O, is_causal=True)           # warm-up outside capture (raises the LDS limit once)
    torch.cuda.synchronize()
    expected = O.clone()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        O.zero_()
        with torch.cuda.graph(graph, stream=st):
            fa.flash_attention(Q, K, V, O, is_causal=True)
    torch.cuda.current_stream().wait_stream(st)
    O.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(O, expected)
    Q.copy_(randn((2, 4, 512, 128), 83, torch.bfloat16))     # new data through the same graph
    graph.replay()
    torch.cuda.synchronize()
    ref = oracle.attention(Q.float().cpu().numpy(), K.float().cpu().numpy(), V.float().cpu().numpy(), causal=True)
    check(O.float().cpu().numpy(), ref, 8e-3, 8e-3)


# ------------------------------------------------------------------ BASELINE sizes
def _sampled_check(B, H, S, d, causal, seeds, heads, rows):
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16) for s in seeds)
    O = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, out_dtype=torch.float32)
    torch.cuda.synchronize()
    Of = O.cpu().numpy().reshape(B * H, S, d)
    Qf, Kf, Vf = (t.float().numpy() for t in (Q, K, V))
    for h0 in heads:
        for (r0, r1) in rows:
            ref = oracle.attention_rows(Qf, Kf, Vf, (h0, h0 + 1), (r0, r1), causal=causal)
            check(Of[h0:h0 + 1, r0:r1], ref, *STATED, rms=1e-3)
    return Q, K, V, O


@pytest.mark.parametrize("B,H,S,d,causal,out_dtype", [
    (3, 37, 700, 128, True, torch.bfloat16),     # 333 units on 256 workgroups: 2 ragged rounds, LDS epilogue + barrier
    (2, 150, 300, 64, False, torch.float32),     # 600 units, 3 rounds, direct fp32 epilogue
    (5, 61, 257, 128, True, torch.float32),      # 610 units, second query block has one live row
])
def test_persistent_grid_every_unit_is_computed(B, H, S, d, causal, out_dtype):
    """More units than CUs: each workgroup walks several units (snake schedule, next unit's Q and tile 0
    prefetched under the epilogue).  Full-tensor compare, so a skipped or doubly-assigned unit cannot hide."""
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16) for s in (95, 96, 97))
    ref = oracle.attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=causal)
    atol, rtol = tol_for(torch.bfloat16, out_dtype)           # (S < FA_EARLY_KEYS: fp16 weights on every row)
    check(run_gpu(Q, K, V, causal, out_dtype=out_dtype), ref, atol, rtol)


def test_persistent_grid_fallback_inside_a_walk():
    """One head in the middle of a workgroup's unit list overflows the optimistic pass (scores x12): that unit
    is recomputed by the tracked pass and the units before / after it on the same workgroup stay right."""
    B, H, S, d = 1, 300, 512, 128      # 600 units -> every workgroup walks 2-3 units
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16) for s in (98, 99, 100))
    for h in (5, 140, 141, 299):
        Q[0, h] *= 12
        K[0, h] *= 12
    ref = oracle.attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=True)
    check(run_gpu(Q, K, V, True), ref, *BF16W)      # (four heads x12: scores ~ N(0, 144), far from the N(0,1) data the stated tolerance is for)


def test_baseline_cfg1_full_tensor():
    """BASELINE cfg1: bf16, B=4, H=8, S=2048, d=64, non-causal -- full-tensor compare (34 GFLOP on the CPU)."""
    B, H, S, d = 4, 8, 2048, 64
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16) for s in (31, 32, 33))
    ref = oracle.attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=False)
    check(run_gpu(Q, K, V, False), ref, *STATED, rms=1e-3)


def test_baseline_cfg2_sampled_and_properties():
    """BASELINE cfg2 (headline): bf16, B=8, H=16, S=4096, d=128, causal.  Oracle on sampled heads/rows
    (first, middle, last head; first rows, tile edges, last rows) + size-independent properties."""
    B, H, S, d = 8, 16, 4096, 128
    Q, K, V, O = _sampled_check(B, H, S, d, True, (34, 35, 36), heads=(0, 77, 127),
                                rows=((0, 96), (2016, 2112), (4000, 4096)))
    Qd, Kd, Vd = Q.to(DEV), K.to(DEV), V.to(DEV)
    # (1) softmax rows sum to one: V == 1 gives O == 1 whatever Q, K are
    ones = torch.ones_like(Vd)
    O1 = fa.flash_attention(Qd, Kd, ones, is_causal=True, out_dtype=torch.float32)
    assert float((O1 - 1).abs().max()) <= 4e-3
    # (2) linearity in V: attn(V1 + V2) == attn(V1) + attn(V2) up to rounding (V2 = a second draw)
    V2 = randn((B, H, S, d), 37, torch.bfloat16).to(DEV)
    Vs = (Vd.float() + V2.float()).to(torch.bfloat16)
    lhs = fa.flash_attention(Qd, Kd, Vs, is_causal=True, out_dtype=torch.float32)
    rhs = O + fa.flash_attention(Qd, Kd, V2, is_causal=True, out_dtype=torch.float32)
    assert float((lhs - rhs).abs().max()) <= 3e-2 and float((lhs - rhs).pow(2).mean().sqrt()) <= 2e-3
    # (3) a head shard equals the unsharded result bit for bit (the multi-GPU partition, section 8e)
    lo, hi = 48, 80
    Os = fa.flash_attention(Qd.view(B * H, 1, S, d)[lo:hi], Kd.view(B * H, 1, S, d)[lo:hi],
                            Vd.view(B * H, 1, S, d)[lo:hi], is_causal=True, out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert torch.equal(Os.view(hi - lo, S, d), O.view(B * H, S, d)[lo:hi])
    # (4) row 0 of every head under the causal mask is V[0] (one key, weight 1; the FMA in the exponent
    #     leaves l = 1 + O(1e-7), so equality holds to fp32 rounding, not bitwise)
    torch.testing.assert_close(O[:, :, 0], Vd[:, :, 0].float(), rtol=1e-6, atol=1e-6)


def test_noncausal_key_permutation_invariance():
    """Non-causal attention does not depend on the order of the (key, value) pairs."""
    B, H, S, d = 2, 4, 4096, 128
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16).to(DEV) for s in (38, 39, 40))
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(41)).to(DEV)
    a = fa.flash_attention(Q, K, V, out_dtype=torch.float32)
    b = fa.flash_attention(Q, K[:, :, perm].contiguous(), V[:, :, perm].contiguous(), out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert float((a - b).abs().max()) <= 4e-3


# ------------------------------------------------------------------ parity at the STATED tolerance
def _parity_table(tag, O, ref):
    from parity import parity_report
    rep = parity_report(O, ref)
    print(f"PARITY {tag}: max_abs {rep['max_abs_err']:.3e}  max_rel(|ref|>=1e-3) {rep['max_rel_err']:.3e}  rms {rep['rms_err']:.3e}  "
          f"pass fraction at |O-ref| <= 1e-3 + 1e-3|ref|: {rep['pass_frac_at_1e-3']:.6f}  (n = {rep['n']})")
    return rep


@pytest.mark.parametrize("causal", [False, True])
def test_parity_at_stated_tolerance_cfg2_all_weight_precisions(causal):
    """BASELINE.json / BASELINE.md section 4 state the tolerance |O - ref| <= 1e-3 + 1e-3|ref| against check.py
    (/root/reference/check.py:19-21).  On the headline shape (cfg2: S = 4096, d = 128), WHOLE heads, fp32 output:
      * the default call (what bench.py times): fp16 weights on the query rows that see fewer than FA_EARLY_KEYS keys, bf16 weights
        elsewhere -- every element inside the tolerance, causal and not;
      * FA_FLAG_F16_WEIGHTS (weights_dtype=torch.float16): every element inside it;
      * FA_FLAG_BF16_WEIGHTS (weights_dtype=torch.bfloat16), weights rounded to bf16 on every row (2^-9 relative each): the rounding
        error of a row averages out over its keys -- non-causal rows (4096 keys) sit well inside the tolerance, the first rows of a
        causal problem (a handful of keys) do not: held to what it measures (99.994 %), and to 4e-3 element-wise.
    The figures are printed (DESIGN.md section 7 quotes them)."""
    B, H, S, d = 8, 16, 4096, 128
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16) for s in (34, 35, 36))
    Qd, Kd, Vd = Q.to(DEV), K.to(DEV), V.to(DEV)
    heads = (0, 41, 77, 127)
    Qf, Kf, Vf = (t.float().numpy() for t in (Q, K, V))
    refs = {h: oracle.attention_rows(Qf, Kf, Vf, (h, h + 1), (0, S), causal=causal)[0] for h in heads}   # whole heads
    floors = {None: 1.0, torch.float16: 1.0, torch.bfloat16: 0.9999 if causal else 1.0}
    outs = {}
    for wd in (None, torch.float16, torch.bfloat16):
        O = fa.flash_attention(Qd, Kd, Vd, is_causal=causal, out_dtype=torch.float32, weights_dtype=wd)
        torch.cuda.synchronize()
        Of = O.cpu().numpy().reshape(B * H, S, d)
        outs[wd] = Of
        got = np.stack([Of[h] for h in heads])
        ref = np.stack([refs[h] for h in heads])
        name = {None: "default (fp16 on early rows)", torch.float16: "fp16", torch.bfloat16: "bf16"}[wd]
        rep = _parity_table(f"cfg2 causal={causal} weights={name}", got, ref)
        assert rep["pass_frac_at_1e-3"] >= floors[wd], rep
        if wd == torch.bfloat16:
            assert (np.abs(got - ref) <= 4e-3 + 4e-3 * np.abs(ref)).all(), rep
    # the default at this size is the mixed-precision kernel: its bf16-weights units ARE the bf16-weights kernel's (bit for bit), its
    # fp16-weights units the same arithmetic as FA_FLAG_F16_WEIGHTS on the other MFMA shape (32x32x16 against 16x16x32: the fp32 sums
    # are taken in another order)
    E = fa.FA_EARLY_KEYS
    if causal:
        assert np.array_equal(outs[None][:, E:], outs[torch.bfloat16][:, E:])
        assert np.abs(outs[None][:, :E] - outs[torch.float16][:, :E]).max() <= 2e-5 and not np.array_equal(outs[None][:, :E], outs[torch.bfloat16][:, :E])
    else:
        assert np.array_equal(outs[None], outs[torch.bfloat16])


def test_default_weight_precision_is_the_two_kernels_on_disjoint_rows():
    """flags = 0 on a bf16 problem = fp16 weights on the query blocks whose rows see fewer than FA_EARLY_KEYS = 1024 keys + bf16 weights
    on the rest.  Rows with bf16 weights: bit for bit the FA_FLAG_BF16_WEIGHTS call (the same code), O and LSE, every head.  Rows with
    fp16 weights: bit for bit the FA_FLAG_F16_WEIGHTS call where the whole problem is "early" (the same kernel); in a causal problem
    longer than FA_EARLY_KEYS they come from the mixed-precision kernel's fp16 units -- the same arithmetic on the 32x32x16 MFMA instead
    of the 16x16x32 one, equal to fp32 summation order."""
    E = fa.FA_EARLY_KEYS

    def three(Q, K, V, causal, lse, out_dtype=torch.float32):
        outs = []
        for wd in (None, torch.float16, torch.bfloat16):
            r = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, out_dtype=out_dtype, return_lse=lse, weights_dtype=wd)
            torch.cuda.synchronize()
            outs.append(tuple(t.float().cpu().numpy() for t in r) if lse else (r.float().cpu().numpy(),))
        return outs

    for d, lse, out_dtype in ((128, True, torch.float32), (64, False, torch.bfloat16)):
        # causal, S = 1500 > E: rows [0, 1024) early, [1024, 1500) main -- the mixed-precision kernel
        Q, K, V = (randn((2, 3, 1500, d), s, torch.bfloat16) for s in (601, 602, 603))
        dflt, f16, b16 = three(Q, K, V, True, lse, out_dtype)
        for k in range(len(dflt)):
            assert np.array_equal(dflt[k][:, :, E:], b16[k][:, :, E:])
            close = 2e-5 if out_dtype == torch.float32 else 2.0 ** -7      # (bf16 output: one ulp at |O| ~ 1)
            assert np.abs(dflt[k][:, :, :E] - f16[k][:, :, :E]).max() <= close
        assert not np.array_equal(f16[0][:, :, E:], b16[0][:, :, E:])            # (the two precisions do differ)
        assert np.abs(dflt[0][:, :, :E] - f16[0][:, :, :E]).max() < np.abs(dflt[0][:, :, :E] - b16[0][:, :, :E]).max()
        # causal cross attention against FEWER keys than FA_EARLY_KEYS: every row sees < E keys -> all early
        dflt, f16, b16 = three(Q, K[:, :, :700].contiguous(), V[:, :, :700].contiguous(), True, lse, out_dtype)
        assert all(np.array_equal(dflt[k], f16[k]) for k in range(len(dflt)))
        # no mask: early iff seqLenK < E
        dflt, f16, b16 = three(Q, K[:, :, :1023].contiguous(), V[:, :, :1023].contiguous(), False, lse, out_dtype)
        assert all(np.array_equal(dflt[k], f16[k]) for k in range(len(dflt)))
        dflt, f16, b16 = three(Q, K[:, :, :1024].contiguous(), V[:, :, :1024].contiguous(), False, lse, out_dtype)
        assert all(np.array_equal(dflt[k], b16[k]) for k in range(len(dflt)))
    # padded head dimension: one form, the flag for it accepted, the fp16 one refused
    Q, K, V = (randn((1, 2, 300, 80), s, torch.bfloat16) for s in (604, 605, 606))
    a = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=True, out_dtype=torch.float32)
    b = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=True, out_dtype=torch.float32, weights_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_early_rows_meet_the_stated_tolerance_on_short_and_ragged_problems():
    """The rows the default computes with fp16 weights, where bf16 weights miss the stated tolerance: short sequences, ragged lengths,
    both head dimensions, causal and not -- every element inside 1e-3 + 1e-3|ref| (fp32 output)."""
    for (B, H, Sq, Sk, d, causal) in ((2, 4, 300, 300, 128, True), (1, 3, 1024, 1024, 64, True), (2, 2, 513, 129, 128, False),
                                      (1, 2, 64, 64, 128, False), (1, 4, 1100, 1100, 128, True)):
        Q, K, V = randn((B, H, Sq, d), 611, torch.bfloat16), randn((B, H, Sk, d), 612, torch.bfloat16), randn((B, H, Sk, d), 613, torch.bfloat16)
        O = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, out_dtype=torch.float32)
        torch.cuda.synchronize()
        ref = oracle.attention_numpy(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=causal)
        rep = _parity_table(f"B{B} H{H} Sq{Sq} Sk{Sk} d{d} causal={causal}", O.cpu().numpy(), ref)
        assert rep["pass_frac_at_1e-3"] == 1.0, rep


PAIR_CASES = [
    # (B, H, Sq, Sk, d, causal, out dtype, weights): small problems take fwd_mfma_pair_kernel -- causal d = 64: at most one 256-row unit per CU
    (4, 8, 2048, 2048, 64, True, torch.float32, None),             # BASELINE cfg1's shape under the mask: both precisions in the launch, 512 workgroups
    (4, 8, 2048, 2048, 64, True, torch.bfloat16, torch.bfloat16),  # bf16 weights on every row
    (1, 3, 1000, 1000, 64, True, torch.float32, None),             # ragged last block, 3 heads on 8 XCD groups, every row "early"
    (3, 5, 1500, 1500, 64, True, torch.float16, torch.float16),    # 15 heads: groups of 2 and 1 head(s), ragged, fp16 weights on every row
    (2, 4, 700, 1900, 64, True, torch.float32, None),              # more keys than queries: rows see up to 700 keys of 1900
    (1, 16, 4000, 4000, 64, True, torch.float32, None),            # 16 heads x 32 blocks = all 64 slots of every XCD group
    # causal d = 128: one workgroup per CU, at most one 256-row unit per two CUs
    (1, 8, 4096, 4096, 128, True, torch.float32, None),            # 128 units of 256 rows -> 256 of 128: every CU
    (1, 8, 4096, 4096, 128, True, torch.bfloat16, torch.bfloat16),
    (2, 3, 1100, 1100, 128, True, torch.float32, None),            # ragged, 6 heads on 8 XCD groups
    (1, 5, 900, 3000, 128, True, torch.float16, torch.float16),    # more keys than queries, fp16 weights on every row
    # without the mask (equal units): one workgroup per CU at either head dimension
    (1, 8, 4096, 4096, 128, False, torch.float32, None),
    (2, 8, 2048, 2048, 64, False, torch.float32, None),
    (1, 3, 700, 700, 128, False, torch.float32, None),             # Sk < 1024: fp16 weights on every row
    (3, 3, 1300, 2100, 64, False, torch.bfloat16, torch.bfloat16),
    (1, 7, 1000, 5000, 128, False, torch.float16, torch.float16),
]


@pytest.mark.parametrize("B,H,Sq,Sk,d,causal,out_dtype,wd", PAIR_CASES)
def test_pair_kernel_small_problems(B, H, Sq, Sk, d, causal, out_dtype, wd):
    """The small-problem kernel (128-row units, one per workgroup of four waves; causal d = 64: two workgroups per CU paired heaviest +
    lightest, otherwise one per CU): O and LSE against the oracle, the plan says which kernel ran, and the result equals the persistent
    kernels' on the same heads -- the same configurations compute the same rows (without the mask the persistent kernel that does not
    return the LSE normalises by the MFMA sum of the ROUNDED weights, this one always by the fp32 sum: 2^-9 relative)."""
    plan = fa.plan(B, H, Sq, d, causal, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32)
    assert plan["threads"] == 256 and plan["q_block_rows"] == 128
    Q, K, V = randn((B, H, Sq, d), 71, torch.bfloat16), randn((B, H, Sk, d), 72, torch.bfloat16), randn((B, H, Sk, d), 73, torch.bfloat16)
    Qf, Kf, Vf = (t.float().numpy() for t in (Q, K, V))
    ref = oracle.attention_numpy(Qf, Kf, Vf, causal=causal)
    O, lse = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, out_dtype=out_dtype, return_lse=True, weights_dtype=wd)
    O1 = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, out_dtype=out_dtype, weights_dtype=wd)
    torch.cuda.synchronize()
    assert torch.equal(O, O1)                                    # (one instantiation whether or not the LSE is asked for)
    atol, rtol = tol_for(torch.bfloat16, out_dtype, weights=wd)
    check(O.float().cpu().numpy(), ref, atol, rtol)
    np.testing.assert_allclose(lse.cpu().numpy(), oracle.lse_numpy(Qf, Kf, causal=causal), rtol=2e-6, atol=2e-3)
    if out_dtype == torch.float32 and wd is None:
        rep = _parity_table(f"pair kernel B{B} H{H} Sq{Sq} Sk{Sk} d{d} causal={causal}", O.cpu().numpy(), ref)
        assert rep["pass_frac_at_1e-3"] == 1.0, rep           # the default precision's promise holds here too
    # the same heads inside a problem too large for the pair kernel (more heads): the persistent kernels compute them
    reps = 256 // (H * ((Sq + 255) // 256)) + 1
    big = fa.plan(B * reps, H, Sq, d, causal, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32)
    assert big["threads"] == 512
    Ob, _ = fa.flash_attention(Q.repeat(reps, 1, 1, 1).to(DEV), K.repeat(reps, 1, 1, 1).to(DEV), V.repeat(reps, 1, 1, 1).to(DEV),
                               is_causal=causal, out_dtype=out_dtype, weights_dtype=wd, return_lse=True)
    torch.cuda.synchronize()
    diff = (Ob[:B].float() - O.float()).abs().max().item()
    assert diff <= (1e-6 if out_dtype == torch.float32 else 8e-3), diff   # (same math on 128- against 256-row blocks)


def test_lse_request_changes_o_by_at_most_one_ulp():
    """bf16 inputs without the mask: a call that also asks for the LSE runs the instantiation that sums the UNROUNDED weights in
    fp32 (so the LSE is exact to fp32 rounding), a call that does not takes the row sums from the MFMA over the ROUNDED weights
    (include/flash_attention.h, flash_attention_lse).  The two normalisers differ by the rounding of the weights averaged over the
    row: O differs by at most one ulp of a bf16 output, and by <= 2^-9 relative in fp32."""
    Q, K, V = (randn((8, 8, 2048, 128), s, torch.bfloat16) for s in (621, 622, 623))   # (512 units: the persistent kernels)
    assert fa.plan(8, 8, 2048, 128, False, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32)["threads"] == 512
    for causal in (False, True):
        o1 = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, out_dtype=torch.float32, weights_dtype=torch.bfloat16)
        o2, _ = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, out_dtype=torch.float32, weights_dtype=torch.bfloat16, return_lse=True)
        torch.cuda.synchronize()
        a, b = o1.cpu().numpy(), o2.cpu().numpy()
        if causal:
            assert np.array_equal(a, b)       # (32x32x16 engine: fp32 sum of the unrounded weights either way)
        else:
            assert np.abs(a - b).max() <= 2.0 ** -9 * np.abs(a).max() and not np.array_equal(a, b)
        b1 = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, weights_dtype=torch.bfloat16)
        b2, _ = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, weights_dtype=torch.bfloat16, return_lse=True)
        torch.cuda.synchronize()
        ulp = np.abs(b1.view(torch.int16).cpu().numpy().astype(np.int32) - b2.view(torch.int16).cpu().numpy().astype(np.int32))
        assert ulp.max() <= 1


def test_parity_at_stated_tolerance_cfg1_full_tensor():
    """BASELINE cfg1 (S = 2048, d = 64, non-causal), the whole tensor, both weight precisions."""
    B, H, S, d = 4, 8, 2048, 64
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16) for s in (31, 32, 33))
    ref = oracle.attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=False)
    for wd in (None, torch.float16):
        O = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), out_dtype=torch.float32, weights_dtype=wd)
        torch.cuda.synchronize()
        rep = _parity_table(f"cfg1 weights={'fp16' if wd else 'bf16'}", O.cpu().numpy(), ref)
        assert rep["pass_frac_at_1e-3"] == 1.0, rep


def test_f16_weights_option_edges():
    """FA_FLAG_F16_WEIGHTS: LSE, bf16 output, forced rescale / fallback, ragged S, strided layout; and the flag is refused where
    it does not apply (fp32 / fp8 inputs, padded head dimensions)."""
    B, H, S, d = 2, 3, 777, 128
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16) for s in (201, 202, 203))
    K[:, :, 500] = (6.0 * Q[:, :, 3].float()).to(torch.bfloat16)
    for causal in (False, True):
        for mul in (1.0, 12.0):      # 12: the optimistic pass overflows and the tracked pass takes over
            Qm, Km = (Q.float() * mul).to(torch.bfloat16), (K.float() * mul).to(torch.bfloat16)
            ref = oracle.attention(Qm.float().numpy(), Km.float().numpy(), V.float().numpy(), causal=causal)
            O, lse = fa.flash_attention(Qm.to(DEV), Km.to(DEV), V.to(DEV), is_causal=causal, out_dtype=torch.float32, return_lse=True,
                                        weights_dtype=torch.float16)
            torch.cuda.synchronize()
            check(O.cpu().numpy(), ref, 2e-3 * mul, 2e-3 * mul)
            np.testing.assert_allclose(lse.cpu().numpy(), oracle.lse_numpy(Qm.float().numpy(), Km.float().numpy(), causal=causal),
                                       rtol=2e-5, atol=2e-4 * mul * mul)
    Ob = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=True, weights_dtype=torch.float16)       # bf16 output
    torch.cuda.synchronize()
    check(Ob.float().cpu().numpy(), oracle.attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=True), 6e-3, 6e-3)
    # d = 64, model-layout strides
    Qm, Km, Vm = (randn((2, 300, 4 * 64), s, torch.bfloat16) for s in (204, 205, 206))
    view = lambda t: t.view(2, 300, 4, 64).transpose(1, 2)
    O = fa.flash_attention(view(Qm.to(DEV)), view(Km.to(DEV)), view(Vm.to(DEV)), out_dtype=torch.float32, weights_dtype=torch.float16)
    torch.cuda.synchronize()
    check(O.cpu().numpy(), oracle.attention_numpy(*(view(t).float().numpy() for t in (Qm, Km, Vm))), 2e-3, 2e-3)
    for bad in (randn((1, 1, 64, 128), 1, torch.float32), randn((1, 1, 64, 80), 1, torch.bfloat16)):
        with pytest.raises(fa.FlashAttentionError) as e:
            fa.flash_attention(bad.to(DEV), bad.to(DEV), bad.to(DEV), weights_dtype=torch.float16)
        assert e.value.code == -8   # FA_ERR_BAD_FLAGS


# ------------------------------------------------------------------ outside N(0,1): |V| beyond fp16, sharp softmax
def _rel_check(O, ref, scale, atol, rtol):
    """|O - ref| <= scale * (atol + rtol |ref / scale|): the stated form of the tolerance for values of magnitude `scale`."""
    assert np.isfinite(O).all(), f"{(~np.isfinite(O)).sum()} non-finite outputs"
    err = np.abs(O - ref)
    bad = err > scale * atol + rtol * np.abs(ref)
    assert not bad.any(), f"{bad.sum()} / {bad.size} outside tolerance, max abs err {err.max():.3e} (scale {scale:g})"


@pytest.mark.parametrize("B,H,S,Sk,d,causal,k_big", [
    (2, 64, 2048, 2048, 128, True, 700),     # persistent kernels, the fused two-precision launch: the early units that touch key 700's tile fall back
    (2, 64, 2048, 2048, 128, True, 2047),    # ... a key only the late (bf16-weights) units ever load: nothing to fall back from
    (4, 40, 512, 900, 128, False, 100),      # no mask, seqLenK < FA_EARLY_KEYS: every unit runs the fp16-weights kernel and every row sees the key
    (1, 3, 600, 600, 64, True, 300),         # the pair kernel (small problem), d = 64
    (2, 96, 1024, 1024, 64, True, 1023),     # d = 64 persistent, all rows early; only the last row sees the key, 63 rows share its tile masked
])
def test_default_call_is_finite_and_right_for_any_finite_bf16_v(B, H, S, Sk, d, causal, k_big):
    """The reference's V is `const float*` (kernels/FlashAttention.cuh:60): any finite value is valid input.  The default call's
    fp16-weights kernels hold V as fp16, where a finite bf16 |v| > 65504 is inf -- and 0 * inf = NaN would poison rows that do not even
    see the key (a causally masked key in the row's own tile).  A unit whose fp16 passes come out non-finite is repeated with bf16
    weights and bf16 V (kernel_bf16.hip.h: run_units): V[k_big] = 1e5 (masked for some rows of its tile under the mask, visible to
    others) must give a finite O inside the bf16-weights tolerance everywhere, and inside the stated one on the units that never load it."""
    Q, K, V = randn((B, H, S, d), 701, torch.bfloat16), randn((B, H, Sk, d), 702, torch.bfloat16), randn((B, H, Sk, d), 703, torch.bfloat16)
    V[:, :, k_big] = 1.0e5
    V[:, 0, k_big] = -1.0e28                              # (head 0: the largest magnitude the header promises, 2^95 ~ 4e28)
    Qf, Kf, Vf = (t.float().numpy() for t in (Q, K, V))
    ref = oracle.attention_numpy(Qf, Kf, Vf, causal=causal)
    O, lse = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, out_dtype=torch.float32, return_lse=True)
    O1 = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, out_dtype=torch.float32)
    torch.cuda.synchronize()
    for got in (O.cpu().numpy(), O1.cpu().numpy()):
        assert np.isfinite(got).all()
        # rows that see the key: |O| is the key's magnitude times its weight, the relative term rules; elsewhere the absolute one (bf16 weights)
        err = np.abs(got - ref)
        bad = err > 4e-3 + 4e-3 * np.abs(ref)
        assert not bad.any(), f"{bad.sum()} / {bad.size} outside 4e-3 + 4e-3|ref|, worst {err[bad].max():.3e}"
        if causal:          # rows in 256-row blocks entirely in front of the key's tile never load it: the default precision's promise holds
            clean = (k_big // 64 * 64) // 256 * 256
            if clean > 0:
                rep = _parity_table(f"|V|=1e5 at key {k_big}: rows < {clean}", got[:, :, :clean], ref[:, :, :clean])
                assert rep["pass_frac_at_1e-3"] == 1.0, rep
    np.testing.assert_allclose(lse.cpu().numpy(), oracle.lse_numpy(Qf, Kf, causal=causal), rtol=2e-5, atol=2e-3)   # (the LSE never sees V)


def test_v_of_magnitude_1e5_everywhere():
    """V ~ 1e5 * N(0,1) on a causal S = 2048 problem: every fp16-weights unit overflows and is repeated with bf16 weights; the result is
    finite and inside the bf16-weights tolerance at V's scale.  FA_FLAG_F16_WEIGHTS on the same data goes the same way."""
    B, H, S, d = 1, 160, 2048, 128
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16) for s in (711, 712, 713))
    V = (V.float() * 1.0e5).to(torch.bfloat16)
    ref = oracle.attention(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=True)
    for wd in (None, torch.float16, torch.bfloat16):
        O = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=True, out_dtype=torch.float32, weights_dtype=wd)
        torch.cuda.synchronize()
        _rel_check(O.cpu().numpy(), ref, 1.0e5, 4e-3, 4e-3)


# (causal, weights) -> floor of the pass fraction at the stated tolerance; MEASURED (profiles/r04_pytest_gpu.log, three whole heads, 1.57 M
# elements): no mask: default = bf16 weights 0.97914, fp16 1.0; causal: default 0.90906 (fp16 on rows < 1024, bf16 above), bf16 0.88290, fp16 1.0
SHARP_FLOORS = {(False, "default"): 0.975, (False, "bf16"): 0.975, (False, "fp16"): 1.0,
                (True, "default"): 0.90, (True, "bf16"): 0.875, (True, "fp16"): 1.0}


@pytest.mark.parametrize("causal", [False, True])
def test_sharp_softmax_parity_is_what_it_measures(causal):
    """The stated tolerance is met by the default precision on N(0,1) tensors (tests above); it is a property of the DATA: a sharper
    softmax (Q, K x 3: scores ~ N(0, 9^2)) puts a row's mass on a handful of keys, whose bf16 rounding errors (2^-9 each) no longer
    average out over the thousands of keys the row sees.  S = 4096, d = 128, whole heads, fp32 output: the default is held to the
    fraction it measures (floor below it; include/flash_attention.h quotes it), FA_FLAG_F16_WEIGHTS to every element."""
    B, H, S, d = 1, 160, 4096, 128
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16) for s in (721, 722, 723))
    Q, K = (Q.float() * 3).to(torch.bfloat16), (K.float() * 3).to(torch.bfloat16)
    heads = (0, 79, 159)
    Qf, Kf, Vf = (t.float().numpy() for t in (Q, K, V))
    ref = np.stack([oracle.attention_rows(Qf, Kf, Vf, (h, h + 1), (0, S), causal=causal)[0] for h in heads])
    for wd, name in ((None, "default"), (torch.float16, "fp16"), (torch.bfloat16, "bf16")):
        O = fa.flash_attention(Q.to(DEV), K.to(DEV), V.to(DEV), is_causal=causal, out_dtype=torch.float32, weights_dtype=wd)
        torch.cuda.synchronize()
        got = np.stack([O.cpu().numpy().reshape(B * H, S, d)[h] for h in heads])
        rep = _parity_table(f"SHARP softmax (Q, K x 3) S=4096 causal={causal} weights={name}", got, ref)
        assert np.isfinite(got).all()
        assert (np.abs(got - ref) <= 8e-3 + 8e-3 * np.abs(ref)).all(), rep
        assert rep["pass_frac_at_1e-3"] >= SHARP_FLOORS.get((causal, name), 0.0), rep


def test_small_noncausal_shard_against_the_whole_problem():
    """A shard small enough for the pair kernel against the unsharded run (persistent kernel), no mask, no LSE request: the pair
    kernel normalises by the fp32 sum of the unrounded weights, the persistent kernel by the MFMA sum of the rounded ones -- equal to
    2^-9 relative, NOT bit for bit (include/flash_attention.h says so next to flash_attention_lse); with an LSE request on both sides
    the two are the same arithmetic."""
    B, H, S, d = 1, 64, 2048, 128
    Q, K, V = (randn((B, H, S, d), s, torch.bfloat16).to(DEV) for s in (731, 732, 733))
    assert fa.plan(B, H, S, d, False, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32)["threads"] == 512 and fa.plan(1, 8, S, d, False, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32)["threads"] == 256
    whole = fa.flash_attention(Q, K, V, out_dtype=torch.float32)
    part = fa.flash_attention(Q[:, 8:16].contiguous(), K[:, 8:16].contiguous(), V[:, 8:16].contiguous(), out_dtype=torch.float32)
    whole_l, _ = fa.flash_attention(Q, K, V, out_dtype=torch.float32, return_lse=True)
    part_l, _ = fa.flash_attention(Q[:, 8:16].contiguous(), K[:, 8:16].contiguous(), V[:, 8:16].contiguous(), out_dtype=torch.float32, return_lse=True)
    torch.cuda.synchronize()
    a, b = whole[:, 8:16].cpu().numpy(), part.cpu().numpy()
    assert np.abs(a - b).max() <= 2.0 ** -9 * np.abs(a).max()
    assert float((whole_l[:, 8:16] - part_l).abs().max()) <= 1e-6
