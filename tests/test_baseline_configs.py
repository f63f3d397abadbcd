"""GPU tests at the sizes of BASELINE.json configs 3 and 4 (configs 0-2: test_flash_attention.py).

cfg3  1xMI355X fp8 e4m3 on the CDNA4 fp8 MFMA, seq=16384, d=128   (B=1, H=16: BASELINE leaves them open)
cfg4  8xMI355X batch x head shard, B=64, H=32, seq=8192, d=128     -- 2^31 elements = 4 GiB per bf16 tensor

cfg4 is the overflow case: the reference indexes with 32-bit `int` (kernels/loaders.cuh:57,92:
`(batchIdx*numHeads + headIdx)*seqLen*dHead + ...`), which wraps at head 1024 in bytes and at the tensor's
end in elements.  Here the whole B*H = 2048-head problem is allocated on the ONE GPU of the box (16 GiB for
Q, K, V, O), launched once into a NaN-prefilled O, and heads on both sides of the 2^31-byte line and the last
head are checked against the oracle; then rank 7's slab of the 8-way shard (heads 1792..2047) is run as its
own problem and must equal the same heads of the full run bit for bit (section 8e: the shard IS the
multi-GPU path).  Inputs are generated on the device (a host copy would be 3 x 8 GiB of fp32).
"""
import numpy as np
import pytest

import __graft_entry__ as entry

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

fa = entry.load_package()
import oracle  # noqa: E402  (checker only)
from parity import parity_report  # noqa: E402

DEV = "cuda:0"
FP8 = getattr(torch, "float8_e4m3fn", None)


def _device_randn(shape, seed, dtype):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(shape, generator=g, device=DEV, dtype=torch.float32).to(dtype)


def _head_to_host(t, g):
    """flattened head g of a [BH, 1, S, d] device tensor -> float32 numpy [1, 1, S, d]"""
    return t[g:g + 1].float().cpu().numpy()


@pytest.mark.skipif(FP8 is None, reason="torch build without float8_e4m3fn")
def test_baseline_cfg3_sampled():
    """BASELINE cfg3 at its own size: fp8 e4m3fn, B=1, H=16, S=16384, d=128 (256 key tiles per unit, 64 units per
    head).  Oracle (float64 accumulation, same e4m3-rounded inputs) on sampled heads and rows: first / middle /
    last head; first rows, a tile edge in the middle, the last rows.  Non-causal and causal."""
    B, H, S, d = 1, 16, 16384, 128
    Q, K, V = (_device_randn((B, H, S, d), s, FP8) for s in (301, 302, 303))
    for causal in (False, True):
        O, lse = fa.flash_attention(Q, K, V, is_causal=causal, out_dtype=torch.float32, return_lse=True)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(O).all()) and bool(torch.isfinite(lse).all())
        for h in (0, 7, 15):
            Qh, Kh, Vh = (t[:, h:h + 1].float().cpu().numpy() for t in (Q, K, V))
            for (r0, r1) in ((0, 64), (8160, 8256), (16320, 16384)):
                ref = oracle.attention_rows(Qh, Kh, Vh, (0, 1), (r0, r1), causal=causal)[0]
                got = O[0, h, r0:r1].cpu().numpy()
                rep = parity_report(got, ref)
                print(f"cfg3 causal={causal} head {h} rows [{r0},{r1}): {rep}")
                err = np.abs(got - ref)
                # the stated tolerance: rows of a non-causal S = 16384 problem see 16384 keys (measured max-abs 1.9e-4); under the mask the
                # first rows see a handful, with bf16 weights (fp8 inputs have one weight precision): 4e-3 there, and said so
                tol = 1e-3 if (not causal or r0 >= 8160) else 4e-3
                assert (err <= tol + tol * np.abs(ref)).all(), rep
                assert np.sqrt(np.mean(err ** 2)) <= 1e-3
            # LSE of the sampled rows (float64 restatement on the same inputs)
            r0, r1 = 16320, 16384
            s = (Qh[0, 0, r0:r1].astype(np.float64) @ Kh[0, 0].astype(np.float64).T) / np.sqrt(d)
            if causal:
                s = np.where(np.arange(S)[None, :] > np.arange(r0, r1)[:, None], -np.inf, s)
            m = s.max(-1)
            ref_lse = m + np.log(np.exp(s - m[:, None]).sum(-1))
            np.testing.assert_allclose(lse[0, h, r0:r1].cpu().numpy(), ref_lse, rtol=2e-6, atol=3e-3)


def test_baseline_cfg4_offsets():
    """BASELINE cfg4 whole, on one GPU: B*H = 2048 heads, S=8192, d=128, bf16 -> 2^31 elements / 4 GiB per tensor."""
    BH, S, d = 64 * 32, 8192, 128
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2 ** 30:
        pytest.skip("needs ~26 GiB of device memory")
    Q, K, V = (_device_randn((BH, 1, S, d), s, torch.bfloat16) for s in (401, 402, 403))
    assert Q.numel() == 2 ** 31
    O = torch.full((BH, 1, S, d), float("nan"), device=DEV, dtype=torch.bfloat16)
    _, lse = fa.flash_attention(Q, K, V, O, is_causal=False, return_lse=True)
    torch.cuda.synchronize()
    # every one of the 2^31 output elements was written (NaN prefill), in chunks to bound the temporaries
    for g0 in range(0, BH, 256):
        assert bool(torch.isfinite(O[g0:g0 + 256]).all()), f"unwritten / non-finite output in heads [{g0},{g0 + 256})"
    assert bool(torch.isfinite(lse).all())
    # heads on both sides of the 2^31-byte line (head 1024 starts at byte 2^31) and the last head (ends at byte 2^32)
    for g in (0, 1023, 1024, 2047):
        Qh, Kh, Vh = (_head_to_host(t, g) for t in (Q, K, V))
        for (r0, r1) in ((0, 32), (4080, 4112), (8160, 8192)):
            ref = oracle.attention_rows(Qh, Kh, Vh, (0, 1), (r0, r1), causal=False)[0]
            got = O[g, 0, r0:r1].float().cpu().numpy()
            err = np.abs(got - ref)
            rep = parity_report(got, ref)
            print(f"cfg4 head {g} rows [{r0},{r1}): {rep}")
            assert (err <= 8e-3 + 8e-3 * np.abs(ref)).all(), rep      # bf16 P + bf16 O rounding
    # the whole problem again with fp32 OUTPUT (8 GiB), at the stated tolerance 1e-3 + 1e-3|ref| (the bf16 output above cannot meet
    # it: its half-ulp is 2^-9): heads 0, 1023, 1024, 2047, first / middle / last rows
    O32 = fa.flash_attention(Q, K, V, is_causal=False, out_dtype=torch.float32)
    torch.cuda.synchronize()
    for g in (0, 1023, 1024, 2047):
        Qh, Kh, Vh = (_head_to_host(t, g) for t in (Q, K, V))
        for (r0, r1) in ((0, 64), (4064, 4128), (8128, 8192)):
            ref = oracle.attention_rows(Qh, Kh, Vh, (0, 1), (r0, r1), causal=False)[0]
            got = O32[g, 0, r0:r1].cpu().numpy()
            rep = parity_report(got, ref)
            print(f"cfg4 fp32 O head {g} rows [{r0},{r1}): {rep}")
            assert rep["pass_frac_at_1e-3"] == 1.0, rep
    del O32
    # heads are distinct draws: a wrapped offset would make head g alias head g - 1024 (or 0)
    assert not torch.equal(O[2047], O[1023]) and not torch.equal(O[1024], O[0])
    # rank 7's slab of the 8-way shard, as its own dense problem == the same heads of the full run, bit for bit
    lo, hi = fa.shard_range(BH, 7, 8)
    assert (lo, hi) == (1792, 2048)
    # (same entry point as the full run: a call that also returns the LSE sums the unrounded weights, one that does not
    #  takes its row sums from an MFMA over the bf16-rounded ones -- the two differ in the last bit)
    Os, lse_s = fa.flash_attention(Q[lo:hi], K[lo:hi], V[lo:hi], is_causal=False, return_lse=True)
    torch.cuda.synchronize()
    assert torch.equal(Os, O[lo:hi]) and torch.equal(lse_s, lse[lo:hi])
    del Os
    # causal, last heads only (the slab again): last row block of the last head against the oracle
    Oc = fa.flash_attention(Q[lo:hi], K[lo:hi], V[lo:hi], is_causal=True, out_dtype=torch.float32)
    torch.cuda.synchronize()
    Qh, Kh, Vh = (_head_to_host(t, 2047) for t in (Q, K, V))
    for (r0, r1) in ((0, 40), (8128, 8192)):
        ref = oracle.attention_rows(Qh, Kh, Vh, (0, 1), (r0, r1), causal=True)[0]
        got = Oc[hi - lo - 1, 0, r0:r1].cpu().numpy()
        assert (np.abs(got - ref) <= 4e-3 + 4e-3 * np.abs(ref)).all(), parity_report(got, ref)
