"""oracle -- CPU restatement of the reference's attention math.  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker / the timed CPU baseline.  The product path
(``libflash_attention.so`` + its host binding) never imports it and has no CPU fallback.

Two restatements live here:

* ``liboracle_attention.so`` (``cpu_attention.c``, plain C + OpenMP) -- bound below with ctypes.
* ``mha_numpy`` / ``attention_numpy`` -- a numpy restatement of ``check.py:14-24`` used to
  cross-check the C code on small cases.

Both are pinned by ``tests/test_oracle.py`` against ``tests/golden/`` (vectors minted in the
build container by importing the reference's ``check.py``; generator:
``tests/golden/make_golden.py``) and the reference's all-ones known-answer cases
(``tests/main.cu:24-36,107``; ``check.py:30-43``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("ORACLE_LIB_PATH") or os.path.join(_HERE, "liboracle_attention.so")   # (`make asan`: the sanitizer build)
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)


def build(force: bool = False) -> str:
    """Compile cpu_attention.c with gcc (make).  Building the checker is not using it."""
    src = os.path.join(_HERE, "cpu_attention.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        i, f = ctypes.c_int, ctypes.c_float
        L.oracle_attention_f32.argtypes = [_f32p] * 4 + [i, i, i, i, f, i, i]
        L.oracle_attention_f32.restype = i
        L.oracle_attention_f64acc.argtypes = [_f32p] * 4 + [i, i, i, i, f, i, i]
        L.oracle_attention_f64acc.restype = i
        L.oracle_attention_f64acc_rows.argtypes = [_f32p] * 4 + [i, i, i, f, i, i, i, i, i, i]
        L.oracle_attention_f64acc_rows.restype = i
        L.oracle_attention_maincu_single_head.argtypes = [_f32p] * 4 + [i, i, f, i]
        L.oracle_attention_maincu_single_head.restype = None
        L.oracle_multi_head_attention.argtypes = [_f32p] * 5 + [i, i, i, i, i]
        L.oracle_multi_head_attention.restype = i
        L.oracle_round_to_bf16.argtypes = [_f32p, ctypes.c_int64]
        L.oracle_round_to_bf16.restype = None
        L.oracle_round_to_e4m3fn.argtypes = [_f32p, ctypes.c_int64]
        L.oracle_round_to_e4m3fn.restype = None
        L.oracle_f32_to_e4m3fn.argtypes = [f]
        L.oracle_f32_to_e4m3fn.restype = ctypes.c_uint8
        L.oracle_e4m3fn_to_f32.argtypes = [ctypes.c_uint8]
        L.oracle_e4m3fn_to_f32.restype = f
        _lib = L
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(_f32p)


def _c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def attention(Q, K, V, scale=None, causal=False, f64=True, nthreads=0) -> np.ndarray:
    """Dense [B,H,S,d] attention on the CPU (C oracle).  f64=True: double accumulation."""
    Q, K, V = _c(Q), _c(K), _c(V)
    B, H, S, d = Q.shape
    if scale is None:
        scale = 1.0 / float(np.sqrt(d))
    O = np.empty_like(Q)
    fn = lib().oracle_attention_f64acc if f64 else lib().oracle_attention_f32
    fn(_p(Q), _p(K), _p(V), _p(O), B, H, S, d, float(scale), int(bool(causal)), int(nthreads))
    return O


def attention_rows(Q, K, V, heads, rows, scale=None, causal=False, nthreads=0) -> np.ndarray:
    """Rows [rows[0],rows[1]) of flattened heads [heads[0],heads[1]) only -> [nh, nr, d]."""
    Q, K, V = _c(Q), _c(K), _c(V)
    B, H, S, d = Q.shape
    if scale is None:
        scale = 1.0 / float(np.sqrt(d))
    O = np.zeros_like(Q)
    lib().oracle_attention_f64acc_rows(_p(Q), _p(K), _p(V), _p(O), B * H, S, d, float(scale),
                                       int(bool(causal)), heads[0], heads[1], rows[0], rows[1],
                                       int(nthreads))
    return O.reshape(B * H, S, d)[heads[0]:heads[1], rows[0]:rows[1]].copy()


def attention_maincu(Q, K, V, scale, causal=False) -> np.ndarray:
    """tests/main.cu:74-91, literal, one head [S,d]."""
    Q, K, V = _c(Q), _c(K), _c(V)
    S, d = Q.shape
    O = np.empty_like(Q)
    lib().oracle_attention_maincu_single_head(_p(Q), _p(K), _p(V), _p(O), S, d, float(scale),
                                              int(bool(causal)))
    return O


def multi_head_attention(Q, K, V, num_heads, nthreads=0):
    """check.py:4-25 in its own (B,S,d_model) layout -> (output, attn)."""
    Q, K, V = _c(Q), _c(K), _c(V)
    B, S, dm = Q.shape
    out = np.empty_like(Q)
    attn = np.empty((B, num_heads, S, S), dtype=np.float32)
    rc = lib().oracle_multi_head_attention(_p(Q), _p(K), _p(V), _p(out), _p(attn), B, S, dm,
                                           int(num_heads), int(nthreads))
    if rc < 0:
        raise ValueError("d_model must be divisible by num_heads")
    return out, attn


def round_bf16(x) -> np.ndarray:
    x = _c(x).copy()
    lib().oracle_round_to_bf16(_p(x), x.size)
    return x


def round_e4m3fn(x) -> np.ndarray:
    x = _c(x).copy()
    lib().oracle_round_to_e4m3fn(_p(x), x.size)
    return x


# ---------------------------------------------------------------- numpy restatement
def mha_numpy(Q, K, V, num_heads):
    """check.py:4-25 in numpy (float64 internally); small cases only."""
    Q, K, V = (np.asarray(a, dtype=np.float64) for a in (Q, K, V))
    B, S, dm = Q.shape
    dk = dm // num_heads                                            # check.py:11
    q = Q.reshape(B, S, num_heads, dk).transpose(0, 2, 1, 3)        # check.py:14
    k = K.reshape(B, S, num_heads, dk).transpose(0, 2, 1, 3)        # check.py:15
    v = V.reshape(B, S, num_heads, dk).transpose(0, 2, 1, 3)        # check.py:16
    scores = q @ k.transpose(0, 1, 3, 2) / (dk ** 0.5)              # check.py:19
    scores = scores - scores.max(axis=-1, keepdims=True)
    attn = np.exp(scores)
    attn /= attn.sum(axis=-1, keepdims=True)                        # check.py:20
    out = attn @ v                                                  # check.py:21
    out = out.transpose(0, 2, 1, 3).reshape(B, S, dm)               # check.py:24
    return out, attn


def lse_numpy(Q, K, scale=None, causal=False):
    """ln sum_k exp(scale * <q, k>) over the visible keys, float64 -> [..., S]."""
    Q, K = (np.asarray(a, dtype=np.float64) for a in (Q, K))
    S, d = Q.shape[-2:]
    if scale is None:
        scale = 1.0 / np.sqrt(d)
    s = Q @ np.swapaxes(K, -1, -2) * scale
    if causal:
        s = np.where(np.triu(np.ones((S, K.shape[-2]), dtype=bool), 1), -np.inf, s)
    mx = s.max(axis=-1, keepdims=True)
    return (mx + np.log(np.exp(s - mx).sum(axis=-1, keepdims=True)))[..., 0]


def attention_numpy(Q, K, V, scale=None, causal=False):
    """Dense [B,H,S,d] attention in numpy float64; causal masks key k > query q
    (tests/main.cu:81, kernels/utils.cuh:43).  K, V may have a different row count than Q (the
    seqLenQ / seqLenK of kernels/FlashAttention.cuh:23); the mask stays on absolute indices."""
    Q, K, V = (np.asarray(a, dtype=np.float64) for a in (Q, K, V))
    S, d = Q.shape[-2:]
    if scale is None:
        scale = 1.0 / np.sqrt(d)
    s = Q @ np.swapaxes(K, -1, -2) * scale
    if causal:
        s = np.where(np.triu(np.ones((S, K.shape[-2]), dtype=bool), 1), -np.inf, s)
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(axis=-1, keepdims=True)
    return p @ V
