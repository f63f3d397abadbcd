"""How exact is QK^T on the fp8 MFMA?  The same e4m3fn-representable values go through the fp8 path (QK^T on
v_mfma_scale_f32_32x32x64_f8f6f4, unit scales -- bit-identical to the non-scaled 32x32x16 fp8 form) and, widened
exactly to bf16, through the bf16 path; LSE is a direct read-out of the scores.  Measured on MI355X
(profiles/r01_micro_fp8_accumulation_error.log): the fp8 MFMA's scores are off by ~5e-6 of their magnitude
(about 2^-17.6), the bf16 MFMA's by ~5e-9 -- the fp8 datapath accumulates with reduced internal precision.
Run by hand on a GPU box:  python tests/micro/fp8_accumulation_error.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as entry
fa = entry.load_package()
import oracle
FP8 = torch.float8_e4m3fn
for boost in (1.0, 3.0, 12.0):
    for S in (1, 64, 512):
        g = torch.Generator().manual_seed(5)
        B, H, d = 1, 2, 128
        Q = (torch.randn(B, H, S, d, generator=g) * boost).to(FP8); K = (torch.randn(B, H, S, d, generator=g) * boost).to(FP8); V = torch.randn(B, H, S, d, generator=g).to(FP8)
        O8, l8 = fa.flash_attention(Q.cuda(), K.cuda(), V.cuda(), out_dtype=torch.float32, return_lse=True)
        Ob, lb = fa.flash_attention(Q.float().bfloat16().cuda(), K.float().bfloat16().cuda(), V.float().bfloat16().cuda(), out_dtype=torch.float32, return_lse=True)
        torch.cuda.synchronize()
        qn, kn = Q.float().numpy(), K.float().numpy()
        lref = oracle.lse_numpy(qn, kn)
        smax = np.abs(qn.astype(np.float64) @ np.swapaxes(kn.astype(np.float64), -1, -2)).max() / np.sqrt(d)
        print(f"boost {boost:5.1f} S {S:4d}: max|scaled score| {smax:9.2f}  LSE err fp8 path {np.abs(l8.cpu().numpy()-lref).max():.3e}  bf16 path on the same values {np.abs(lb.cpu().numpy()-lref).max():.3e}  "
              f"O fp8 vs bf16 path max diff {float((O8-Ob).abs().max()):.3e}")
