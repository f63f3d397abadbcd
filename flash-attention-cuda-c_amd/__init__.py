"""Host-side binding of the MI355X FlashAttention forward path.

The product is ``libflash_attention.so`` (hand-written HIP for gfx950 behind the C ABI declared in
``include/flash_attention.h``).  This module is the thin Python mirror of the reference's two
entry points for that path:

* ``flash_attention(Q, K, V, O, ...)`` -- the launch signature of
  ``twoLoaderMhaFlashAttentionKernel`` (reference ``kernels/FlashAttention.cuh:59-63``; launched at
  ``tests/main.cu:60-61``): dense ``[B, H, S, d]`` device tensors, ``scale``, ``is_causal``.
* ``multi_head_attention(Q, K, V, num_heads)`` -- the reference's Python oracle API
  (``check.py:4-25``): ``(B, S, d_model)`` tensors; the ``(B,S,H,d_k) -> (B,H,S,d_k)`` transposes of
  ``check.py:14-16,24`` are done by strides inside the kernel, not by copies.

PyTorch is used only for device memory and streams.  There is NO fallback: if the shared library
is missing or the tensors are not on a GPU the call raises.  (The directory name contains '-', so
the package is loaded through ``__graft_entry__.load_package()`` under the module name
``flash_attention_cuda_c_amd``.)
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FA_LIB_PATH: another build of the SAME library (the host-sanitizer build of `make asan`); there is still no other implementation
LIB_PATH = os.environ.get("FA_LIB_PATH") or os.path.join(_HERE, "libflash_attention.so")

FA_DTYPE_F32, FA_DTYPE_BF16, FA_DTYPE_FP8_E4M3, FA_DTYPE_F16 = 0, 1, 2, 3
FA_FLAG_F16_WEIGHTS = 1     # flash_attention_ex: softmax weights rounded to fp16 on every row (bf16 inputs, d = 64 / 128)
FA_FLAG_BF16_WEIGHTS = 2    # ... to bf16 on every row; flags = 0: fp16 on the rows that see fewer than FA_EARLY_KEYS keys, bf16 elsewhere
FA_EARLY_KEYS = 1024

# every symbol include/flash_attention.h declares
EXPORTS = ("flash_attention", "flash_attention_strided", "flash_attention_lse", "flash_attention_cross", "flash_attention_ex", "flash_attention_weights", "flash_attention_shard_range", "flash_attention_sharded",
           "flash_attention_plan", "flash_attention_plan_ex",
           "flash_attention_error_string", "flash_attention_version")


class FaStrides(ctypes.Structure):
    _fields_ = [("strideB", ctypes.c_int64), ("strideH", ctypes.c_int64), ("strideS", ctypes.c_int64)]


class FaLaunchPlan(ctypes.Structure):
    _fields_ = [("q_block_rows", ctypes.c_int), ("kv_block_rows", ctypes.c_int),
                ("threads", ctypes.c_int), ("grid", ctypes.c_int), ("lds_bytes", ctypes.c_int),
                ("kernel_id", ctypes.c_int)]


class FaLaunchPlanEx(ctypes.Structure):
    _fields_ = [("launch", FaLaunchPlan), ("q_blocks", ctypes.c_int), ("first_q_block", ctypes.c_int), ("unit_lists", ctypes.c_int)]


class FlashAttentionError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"flash_attention failed with code {code}: {text}")
        self.code = code


_lib = None


def lib() -> ctypes.CDLL:
    """Load libflash_attention.so (in-tree build).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} not built: run `make` (or __graft_entry__.build()); there is no fallback path")
        L = ctypes.CDLL(LIB_PATH)
        vp, i, f, b = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_bool
        L.flash_attention.argtypes = [vp, vp, vp, vp, i, i, i, i, f, b, i, i, vp]
        L.flash_attention.restype = i
        sp = ctypes.POINTER(FaStrides)
        L.flash_attention_strided.argtypes = [vp, vp, vp, vp, i, i, i, i, f, b, i, i, sp, sp, sp, sp, vp]
        L.flash_attention_strided.restype = i
        L.flash_attention_lse.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, f, b, i, i, vp]
        L.flash_attention_lse.restype = i
        L.flash_attention_cross.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, f, b, i, i, sp, sp, sp, sp, vp]
        L.flash_attention_cross.restype = i
        L.flash_attention_ex.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, f, b, i, i, sp, sp, sp, sp, ctypes.c_uint, vp]
        L.flash_attention_ex.restype = i
        L.flash_attention_weights.argtypes = [vp, vp, vp, vp, i, i, i, i, i, f, b, i, sp, sp, vp]
        L.flash_attention_weights.restype = i
        ip, pp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p)
        L.flash_attention_shard_range.argtypes = [i, i, i, ip, ip]
        L.flash_attention_shard_range.restype = i
        L.flash_attention_sharded.argtypes = [i, ip, pp, pp, pp, pp, i, i, i, i, f, b, i, i, pp]
        L.flash_attention_sharded.restype = i
        L.flash_attention_plan.argtypes = [i, i, i, i, b, i, i, ctypes.POINTER(FaLaunchPlan)]
        L.flash_attention_plan.restype = i
        px = ctypes.POINTER(FaLaunchPlanEx)
        L.flash_attention_plan_ex.argtypes = [i, i, i, i, i, b, i, i, ctypes.c_uint, px, px]
        L.flash_attention_plan_ex.restype = i
        L.flash_attention_error_string.argtypes = [i]
        L.flash_attention_error_string.restype = ctypes.c_char_p
        L.flash_attention_version.argtypes = []
        L.flash_attention_version.restype = ctypes.c_char_p
        _lib = L
    return _lib


def version() -> str:
    return lib().flash_attention_version().decode()


def error_string(code: int) -> str:
    return lib().flash_attention_error_string(int(code)).decode()


def _check(code: int):
    if code != 0:
        raise FlashAttentionError(code, error_string(code))


def _dtype_code(t):
    import torch
    table = {torch.float32: FA_DTYPE_F32, torch.bfloat16: FA_DTYPE_BF16, torch.float16: FA_DTYPE_F16}
    if hasattr(torch, "float8_e4m3fn"):
        table[torch.float8_e4m3fn] = FA_DTYPE_FP8_E4M3
    if t not in table:
        raise TypeError(f"unsupported dtype {t}")
    return table[t]


def _default_out_dtype(in_dtype):
    """fp32 in -> fp32 out (the reference's float* O); fp8 in -> bf16 out; otherwise the input type."""
    import torch
    if in_dtype == torch.float32:
        return torch.float32
    if in_dtype == getattr(torch, "float8_e4m3fn", None):
        return torch.bfloat16
    return in_dtype


def _stream_ptr(stream):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def plan(batchSize, numHeads, seqLen, dHead, is_causal=False, dtype=FA_DTYPE_BF16, o_dtype=FA_DTYPE_F32):
    """Launch geometry the library will use (counterpart of the reference's helpers.hpp:8-36)."""
    p = FaLaunchPlan()
    _check(lib().flash_attention_plan(batchSize, numHeads, seqLen, dHead, bool(is_causal), dtype, o_dtype,
                                      ctypes.byref(p)))
    return {k: getattr(p, k) for k, _ in FaLaunchPlan._fields_}


def plan_ex(batchSize, numHeads, seqLenQ, seqLenK, dHead, is_causal=False, dtype=FA_DTYPE_BF16, o_dtype=FA_DTYPE_F32, flags=0):
    """The launches a flash_attention_ex() call makes: (early, main) dicts; a launch that does not happen has q_blocks = 0."""
    e, m = FaLaunchPlanEx(), FaLaunchPlanEx()
    _check(lib().flash_attention_plan_ex(batchSize, numHeads, seqLenQ, seqLenK, dHead, bool(is_causal), dtype, o_dtype, flags,
                                         ctypes.byref(e), ctypes.byref(m)))
    conv = lambda x: dict({k: getattr(x.launch, k) for k, _ in FaLaunchPlan._fields_}, q_blocks=x.q_blocks, first_q_block=x.first_q_block, unit_lists=x.unit_lists)
    return conv(e), conv(m)


def flash_attention(Q, K, V, O=None, scale=None, is_causal=False, out_dtype=None, stream=None, return_lse=False, weights_dtype=None):
    """O = softmax(scale * Q K^T [+ causal mask]) V on [B, H, S, d] device tensors.

    Argument order and meaning follow the reference kernel (Q, K, V, O, batchSize, numHeads,
    seqLen, scale, is_causal -- kernels/FlashAttention.cuh:59-63); batchSize/numHeads/seqLen/dHead
    are read from Q.shape, ``scale`` defaults to 1/sqrt(d) (tests/main.cu:27).  K and V may hold a
    different number of rows than Q (``[B, H, Sk, d]``: the seqLenQ / seqLenK of the reference's first
    API, kernels/FlashAttention.cuh:23); the causal mask stays ``k > q`` on absolute indices.
    Asynchronous on ``stream`` (default: torch's current stream).  Returns O, or ``(O, LSE)`` with
    ``return_lse=True`` (LSE: fp32 [B, H, S], natural-log sum of exp(scale * scores) over the visible keys).
    ``weights_dtype`` (bf16 inputs): None = the library default (fp16 softmax weights on the rows that see fewer than
    FA_EARLY_KEYS keys, bf16 weights elsewhere); ``torch.float16`` = FA_FLAG_F16_WEIGHTS (fp16 weights on every row; d = 64
    or 128); ``torch.bfloat16`` = FA_FLAG_BF16_WEIGHTS (bf16 weights on every row: the fastest form; accepted and ignored for
    fp32 / fp8 inputs, which have one form).
    """
    import torch
    if not (Q.is_cuda and K.is_cuda and V.is_cuda):
        raise RuntimeError("flash_attention needs device tensors (no CPU fallback)")
    if Q.dim() != 4 or K.dim() != 4 or K.shape != V.shape or Q.shape[:2] != K.shape[:2] or Q.shape[3] != K.shape[3]:
        raise ValueError("Q must be [B, H, S, d] and K, V [B, H, Sk, d]")
    if not (Q.dtype == K.dtype == V.dtype):
        raise TypeError("Q, K, V must share a dtype")
    B, H, S, d = Q.shape
    Sk = K.shape[2]
    if scale is None:
        scale = 1.0 / float(d) ** 0.5
    if O is None:
        O = torch.empty((B, H, S, d), dtype=out_dtype or _default_out_dtype(Q.dtype), device=Q.device)
    elif O.shape != Q.shape or not O.is_cuda:
        raise ValueError("O must be a device tensor shaped like Q")
    dense = all(t.is_contiguous() for t in (Q, K, V, O))
    lse = torch.empty((B, H, S), dtype=torch.float32, device=Q.device) if return_lse else None
    common = (float(scale), bool(is_causal), _dtype_code(Q.dtype), _dtype_code(O.dtype))
    ptrs = (Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr())
    flags = 0
    if weights_dtype is not None:
        if weights_dtype == torch.float16:
            flags |= FA_FLAG_F16_WEIGHTS
        elif weights_dtype == torch.bfloat16:
            # bf16 weights are what fp32 / fp8 inputs get anyway (one form each): the request is a no-op there, not an error --
            # the library rejects the FLAG on non-bf16 inputs (FA_ERR_BAD_FLAGS), so it is only set where it selects something
            if Q.dtype == torch.bfloat16:
                flags |= FA_FLAG_BF16_WEIGHTS
        else:
            raise TypeError("weights_dtype must be torch.float16 or torch.bfloat16")
    with torch.cuda.device(Q.device):
        if flags:
            st = []
            for t in (Q, K, V, O):
                if t.stride(3) != 1:
                    raise ValueError("last dimension must be contiguous")
                st.append(FaStrides(t.stride(0), t.stride(1), t.stride(2)))
            refs = [ctypes.byref(x) for x in st]
            rc = lib().flash_attention_ex(*ptrs, lse.data_ptr() if lse is not None else None, B, H, S, Sk, d, *common, *refs,
                                          flags, _stream_ptr(stream))
        elif dense and Sk == S and lse is not None:
            rc = lib().flash_attention_lse(*ptrs, lse.data_ptr(), B, H, S, d, *common, _stream_ptr(stream))
        elif dense and Sk == S:
            rc = lib().flash_attention(*ptrs, B, H, S, d, *common, _stream_ptr(stream))
        else:
            st = []
            for t in (Q, K, V, O):
                if t.stride(3) != 1:
                    raise ValueError("last dimension must be contiguous")
                st.append(FaStrides(t.stride(0), t.stride(1), t.stride(2)))
            refs = [ctypes.byref(x) for x in st]
            if Sk == S and lse is None:
                rc = lib().flash_attention_strided(*ptrs, B, H, S, d, *common, *refs, _stream_ptr(stream))
            else:
                rc = lib().flash_attention_cross(*ptrs, lse.data_ptr() if lse is not None else None, B, H, S, Sk, d,
                                                 *common, *refs, _stream_ptr(stream))
    _check(rc)
    return (O, lse) if return_lse else O


def shard_range(total_heads, rank, world):
    """[lo, hi) of flattened heads g = b*H + h owned by `rank` of `world` (C ABI twin of shard.shard_heads)."""
    lo, hi = ctypes.c_int(), ctypes.c_int()
    _check(lib().flash_attention_shard_range(total_heads, rank, world, ctypes.byref(lo), ctypes.byref(hi)))
    return lo.value, hi.value


def flash_attention_sharded(Qs, Ks, Vs, Os, batchSize, numHeads, scale=None, is_causal=False, streams=None):
    """One host thread, several devices: Qs[r], Ks[r], Vs[r], Os[r] are rank r's dense [hi-lo, S, d] slabs of the
    flattened heads (shard_range), each resident on its own device.  No collective; asynchronous."""
    n = len(Qs)
    S, d = Qs[0].shape[-2:]
    if scale is None:
        scale = 1.0 / float(d) ** 0.5
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    devs = (ctypes.c_int * n)(*[t.device.index for t in Qs])
    st = (ctypes.c_void_p * n)(*[s.cuda_stream if s is not None else None for s in (streams or [None] * n)])
    _check(lib().flash_attention_sharded(n, devs, arr(Qs), arr(Ks), arr(Vs), arr(Os), batchSize, numHeads, S, d, float(scale),
                                         bool(is_causal), _dtype_code(Qs[0].dtype), _dtype_code(Os[0].dtype), st))
    return Os


def attention_weights(Q, K, lse, scale=None, is_causal=False, stream=None):
    """attn[b,h,q,k] = exp(scale * <Q[q], K[k]> - lse[q]) as a dense fp32 [B, H, Sq, Sk] device tensor -- the
    matrix check.py:20,25 returns as ``attn`` -- from the LSE of a ``flash_attention(..., return_lse=True)``
    call.  Inspection path for small sequences (the fused kernel itself never stores it)."""
    import torch
    B, H, S, d = Q.shape
    Sk = K.shape[2]
    if scale is None:
        scale = 1.0 / float(d) ** 0.5
    if Q.stride(3) != 1 or K.stride(3) != 1 or not lse.is_contiguous() or lse.dtype != torch.float32:
        raise ValueError("last dimension must be contiguous; lse must be dense fp32 [B, H, S]")
    P = torch.empty((B, H, S, Sk), dtype=torch.float32, device=Q.device)
    sq, sk = (FaStrides(t.stride(0), t.stride(1), t.stride(2)) for t in (Q, K))
    with torch.cuda.device(Q.device):
        rc = lib().flash_attention_weights(Q.data_ptr(), K.data_ptr(), lse.data_ptr(), P.data_ptr(), B, H, S, Sk, d,
                                           float(scale), bool(is_causal), _dtype_code(Q.dtype), ctypes.byref(sq),
                                           ctypes.byref(sk), _stream_ptr(stream))
    _check(rc)
    return P


def multi_head_attention(Q, K, V, num_heads, is_causal=False, return_attn=False, out_dtype=None):
    """Drop-in for the reference's ``check.py:multi_head_attention(Q, K, V, num_heads)``.

    Q, K, V: (batch, seq_len, d_model) device tensors.  Returns ``(output, attn)`` like check.py:25,
    with output (batch, seq_len, d_model).  The fused kernel never materialises the (B,H,S,S) attention
    matrix, so ``attn`` is None unless ``return_attn=True``, which rebuilds it (fp32) from the kernel's
    log-sum-exp with a second small kernel -- meant for the small shapes check.py's demo prints.
    The head split / merge of check.py:14-16,24 is done with strides: no transpose copies.
    """
    import torch
    if Q.dim() != 3:
        raise ValueError("Q, K, V must be (batch, seq_len, d_model)")
    B, S, dm = Q.shape
    if dm % num_heads != 0:
        raise ValueError("d_model must be divisible by num_heads")
    dk = dm // num_heads                                                 # check.py:11
    out = torch.empty((B, S, dm), device=Q.device, dtype=out_dtype or _default_out_dtype(Q.dtype))
    view = lambda t: t.view(B, S, num_heads, dk).transpose(1, 2)         # check.py:14-16 (views only)
    scale = 1.0 / float(dk) ** 0.5                                       # check.py:19
    if not return_attn:
        flash_attention(view(Q), view(K), view(V), view(out), scale=scale, is_causal=is_causal)
        return out, None
    _, lse = flash_attention(view(Q), view(K), view(V), view(out), scale=scale, is_causal=is_causal, return_lse=True)
    return out, attention_weights(view(Q), view(K), lse, scale=scale, is_causal=is_causal)
