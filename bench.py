#!/usr/bin/env python3
"""bench.py -- forward attention TFLOP/s on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg2nc|cfg1|cfg3|cfg4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (one flash_attention launch) over one batch of synthetic
N(0,1) bf16 tensors already resident in HBM.  Default workload = BASELINE.json configs[2], the
configuration the metric is quoted on: bf16, B=8, H=16, S=4096, d=128, causal.  With N > 1 every
rank runs that same per-GPU batch on its own heads (weak scaling: the path shards over batch x head
with no data-path collective; RCCL carries only the MAX of elapsed times, outside the timed region).
FLOPs: 4*B*H*S^2*d non-causal, 2*B*H*S^2*d causal (only unmasked work counts) -- SURVEY.md 8(d).

Rank 0 prints ONE JSON line with `roofline` (HIP-event kernel time vs the 2516.6 TFLOP/s bf16 MFMA
peak) and, at N=1, `cpu_baseline` (the oracle's naive fp32 attention timed on the host cores over a
bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2516.6   # 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (B, H, S, d, causal, description)
    "cfg2": (8, 16, 4096, 128, True, "BASELINE cfg2: bf16 B=8 H=16 S=4096 d=128 causal"),
    "cfg2nc": (8, 16, 4096, 128, False, "BASELINE cfg2 shape, non-causal: bf16 B=8 H=16 S=4096 d=128"),
    "cfg1": (4, 8, 2048, 64, False, "BASELINE cfg1: bf16 B=4 H=8 S=2048 d=64 non-causal"),
    "cfg4": (64, 32, 8192, 128, False, "BASELINE cfg4: bf16 B=64 H=32 S=8192 d=128, B*H sharded over the ranks"),
    "cfg3": (1, 16, 16384, 128, False, "BASELINE cfg3: fp8 e4m3fn B=1 H=16 S=16384 d=128 non-causal (B, H chosen: unspecified)"),
}


def flops_of(BH, S, d, causal):
    return (2.0 if causal else 4.0) * BH * float(S) * S * d


def measured_traffic(workload):
    """HBM bytes per launch from the committed PMC run of this same command (separate FETCH_SIZE / WRITE_SIZE
    passes, FETCH_SIZE doubled as the gfx950 guide prescribes) -- profiles/r01_hbm_traffic_<workload>.json;
    None when no such measurement is committed for the workload."""
    path = os.path.join(ROOT, "profiles", f"r01_hbm_traffic_{workload}.json")
    try:
        with open(path) as f:
            return int(json.load(f)["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def power_limited_ceiling():
    """This device's throughput ceiling as the power limit sets it: tests/micro/simd_mix --ceiling runs, on every
    CU, 2 waves per SIMD of (a) back-to-back v_mfma_f32_32x32x16_bf16 and (b) the attention kernel's own
    per-MFMA instruction mix (softmax VALU ops + K / V^T LDS reads), both on RANDOM bf16 operands, pipe >= 90 %
    busy -- the chip then holds 1.3-1.7 GHz, not the 2.4 GHz behind the 2516.6 TFLOP/s nominal peak.  A separate
    child process started after the timed region; None if the microbenchmark is not built."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "micro", "simd_mix")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "--ceiling"], capture_output=True, text=True, timeout=120, check=True).stdout
        return json.loads(out.strip().splitlines()[-1])
    except Exception:   # noqa: BLE001 -- a diagnostic extra must never fail the benchmark
        return None


def cpu_baseline(S, d, causal, budget_s=12.0):
    """Naive fp32 attention (the oracle, a port of tests/main.cu:74-91 / check.py:19-21) on the host
    cores, on a bounded sample: as many whole heads of the workload's (S, d) as fit ~budget_s."""
    import numpy as np
    import oracle
    rng = np.random.default_rng(0)

    def timed(nh):
        Q, K, V = (rng.standard_normal((1, nh, S, d), dtype=np.float32) for _ in range(3))
        O = np.empty_like(Q)
        L = oracle.lib()
        t0 = time.perf_counter()
        thr = L.oracle_attention_f32(oracle._p(Q), oracle._p(K), oracle._p(V), oracle._p(O), 1, nh, S, d,
                                     float(1.0 / np.sqrt(d)), int(causal), 0)
        return time.perf_counter() - t0, thr

    timed(1)                       # warm-up: thread pool start, first touch
    t1, thr = timed(1)
    nh = int(max(1, min(512, budget_s / max(t1, 1e-3))))
    t, thr = timed(nh)
    return {"value": round(flops_of(nh, S, d, causal) / t / 1e12, 5), "unit": "TFLOP/s", "cores": int(thr),
            "kind": "port",
            "sample": f"{nh} head(s) of the workload (S={S}, d={d}, causal={causal}), fp32 naive attention, "
                      f"{t:.2f} s wall, OpenMP over query rows"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the power-limited-ceiling microbenchmark")
    ap.add_argument("--out-dtype", default="bf16", choices=["bf16", "f32"])
    args = ap.parse_args()

    # The ceiling microbenchmark is a separate GPU program: run it as a child BEFORE this process touches the GPU
    # (no exec from a process that has initialised HIP).  Single-GPU runs only.
    ceil = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_ceiling and args.workload != "cfg3":
        ceil = power_limited_ceiling()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    fa = entry.load_package()
    from flash_attention_cuda_c_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n_gpus = world

    B, H, S, d, causal, desc = WORKLOADS[args.workload]
    if args.workload == "cfg4":     # fixed total problem, B*H split over the ranks (strong scaling)
        lo, hi = shard.shard_heads(B * H, rank, world)
        heads_local, scaling = hi - lo, "strong"
        total_heads = B * H
    else:                            # same batch on every rank (weak scaling)
        heads_local, scaling = B * H, "weak"
        total_heads = B * H * world

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    shape = (heads_local, 1, S, d)   # the rank's slab as a dense [heads,1,S,d] tensor
    in_dtype = torch.float8_e4m3fn if args.workload == "cfg3" else torch.bfloat16
    Q, K, V = (torch.randn(shape, generator=g, device=dev, dtype=torch.float32).to(in_dtype) for _ in range(3))
    O = torch.empty(shape, device=dev, dtype=torch.bfloat16 if args.out_dtype == "bf16" else torch.float32)
    scale = 1.0 / d ** 0.5

    def step():
        fa.flash_attention(Q, K, V, O, scale=scale, is_causal=causal)

    # Device priming (setup, untimed, before the W warmup steps): a GPU coming out of idle needs tens of milliseconds
    # of work before its power state and clocks settle -- with a small W the timed steps would otherwise run on the
    # ramp (measured: 855 instead of ~980 TFLOP/s at W=5, K=20).  ~100 ms of the same launches, at most 3000.
    step(); torch.cuda.synchronize()
    t_p = time.perf_counter(); step(); step(); torch.cuda.synchronize()
    t_step = max((time.perf_counter() - t_p) / 2, 1e-6)
    for _ in range(max(0, min(3000, int(0.1 / t_step)))):
        step()
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                     # same stream the kernel is launched on (torch's current stream)
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps
    elapsed = shard.reduce_max(elapsed)
    kernel_ms_max = shard.reduce_max(kernel_ms)

    # sanity: the output is finite and row 0 of a causal head equals V[0]
    ok = bool(torch.isfinite(O.float()).all())
    if causal:
        ok = ok and bool(torch.allclose(O[:, :, 0].float(), V[:, :, 0].float(), rtol=1e-2, atol=1e-2))
    ok_all = shard.reduce_sum(0.0 if ok else 1.0) == 0.0

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        flops_job = flops_of(total_heads, S, d, causal)
        value = flops_job / (ms_per_step * 1e-3) / 1e12
        flops_launch = flops_of(heads_local, S, d, causal)
        achieved = flops_launch / (kernel_ms_max * 1e-3) / 1e12
        # fp8 inputs: QK^T (half the FLOPs) runs on the block-scaled MX MFMA at twice the bf16 rate, P.V on the bf16
        # MFMA: the bound is the time-weighted mix 1 / (0.5/5033.2 + 0.5/2516.6) = 3355.5 TFLOP/s
        peak = 1.0 / (0.5 / (2 * PEAK_BF16_TFLOPS) + 0.5 / PEAK_BF16_TFLOPS) if args.workload == "cfg3" else PEAK_BF16_TFLOPS
        line = {
            "metric": "fwd attention TFLOP/s (bf16, seq=4096, d=128) + % MFMA peak" if args.workload.startswith("cfg2")
                      else "fwd attention TFLOP/s",
            "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "fp8_e4m3fn" if args.workload == "cfg3" else "bf16",
            "data": "synthetic",
            "config": {"workload": desc, "B": B, "H": H, "S": S, "d": d, "causal": causal,
                       "out_dtype": args.out_dtype, "heads_per_gpu": heads_local,
                       "flop_convention": "2*B*H*S^2*d causal / 4*B*H*S^2*d non-causal",
                       "parallelism": f"batch x head shard over {n_gpus} GPU(s), no data-path collective"},
            "pct_of_bf16_mfma_peak": round(100.0 * value / (PEAK_BF16_TFLOPS * n_gpus), 2),
            # causal FLOPs count only the unmasked half; the full-count figure alongside, labelled
            "value_if_masked_half_counted_too": round(value * (2.0 if causal else 1.0), 2),
            "output_ok": ok_all,
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1),
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": measured_traffic(args.workload) if n_gpus == 1 else None,
                         "kernel": "fa::fwd_mfma_kernel", "kernel_ms": round(kernel_ms_max, 5),
                         "algorithmic_hbm_bytes": heads_local * S * d * (3 * Q.element_size() + O.element_size()),
                         "algorithmic_hbm_GBps": round(heads_local * S * d * (3 * Q.element_size() + O.element_size())
                                                       / (kernel_ms_max * 1e-3) / 1e9, 1)},
        }
        if ceil:
            line["roofline"]["power_limited"] = dict(
                ceil, frac_of_mfma_only=round(achieved / ceil["mfma_only_random_bf16_tflops"], 4),
                frac_of_attention_mix=round(achieved / ceil["attention_mix_random_bf16_tflops"], 4))
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(S, d, causal)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
