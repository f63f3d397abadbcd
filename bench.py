#!/usr/bin/env python3
"""bench.py -- forward attention TFLOP/s on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg2nc|cfg1|cfg1c|cfg3|cfg4|anchor]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process (which has not touched the GPU) starts the N ranks
itself -- a child `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same
arguments>` -- relays rank 0's single JSON line and exits with the children's status.  Started under torchrun by someone else
(RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* set) it is one of the ranks.

A "step" is one pass of the hot path (one flash_attention call) over one batch of synthetic N(0,1) bf16 tensors already
resident in HBM.  Default workload = BASELINE.json configs[2], the configuration the metric is quoted on: bf16, B=8, H=16,
S=4096, d=128, causal, with fp32 output (the reference's O is float*, kernels/FlashAttention.cuh:61).  With N > 1 every rank
runs that same per-GPU batch on its own heads (weak scaling: the path shards over batch x head with no data-path collective;
RCCL carries only the MAX of elapsed times, outside the timed region).  Every line also carries a `cfg4` sub-record: BASELINE
configs[4] (B=64, H=32, S=8192, d=128) with its 2048 heads split over the N ranks (strong scaling; N = 1 is its anchor), a
few steps, and a `bf16_out` sub-record: the same workload with bf16 output.
FLOPs: 4*B*H*S^2*d non-causal, 2*B*H*S^2*d causal (only unmasked work counts) -- SURVEY.md 8(d).

Rank 0 prints ONE JSON line with
  `value`         whole-job TFLOP/s from the host clock around exactly K steps between barriers, on a PRIMED device (~100 ms of
                  the same launches first; `value_unprimed` is the same K steps from a device just out of idle);
                  `ms_median` / `ms_min`: per-step HIP-event times of those K steps,
  `roofline`      HIP-event kernel time against the dense MFMA peak of the workload's arithmetic, the HBM traffic measured by
                  the committed PMC run IF it was taken on the same kernel sources (provenance hash), and
  `cpu_baseline`  (N=1) the oracle's naive fp32 attention timed on the host cores over a bounded sample of THIS run's own
                  tensors, whose result is also the checker for `parity`: max-abs / max-rel error and the fraction of
                  elements inside the stated tolerance |O-ref| <= 1e-3 + 1e-3|ref|.  A pass fraction
                  under the floor (1.0 with fp32 output) sets `output_ok` false and the exit status 3.
`--dry-run` (CPU, gloo): no kernel is launched and nothing is measured -- the line's STRUCTURE for N ranks (tests).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))

PEAK_BF16_TFLOPS = 2516.6   # 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md)
METRIC = "fwd attention TFLOP/s/GPU (bf16, seq=4096, d=128) + % MFMA peak"   # BASELINE.json, verbatim
PROFILE_ROUND = "r04"       # profiles/<round>_hbm_traffic_<workload>.json is where roofline.traffic comes from

WORKLOADS = {
    # name: (B, H, S, d, causal, description)
    "cfg2": (8, 16, 4096, 128, True, "BASELINE cfg2: bf16 B=8 H=16 S=4096 d=128 causal"),
    "cfg2nc": (8, 16, 4096, 128, False, "BASELINE cfg2 shape, non-causal: bf16 B=8 H=16 S=4096 d=128"),
    "cfg1": (4, 8, 2048, 64, False, "BASELINE cfg1: bf16 B=4 H=8 S=2048 d=64 non-causal"),
    "cfg1c": (4, 8, 2048, 64, True, "BASELINE cfg1's shape under the causal mask (not a BASELINE config): the pair kernel's case"),
    "cfg4": (64, 32, 8192, 128, False, "BASELINE cfg4: bf16 B=64 H=32 S=8192 d=128, B*H sharded over the ranks"),
    "cfg3": (1, 16, 16384, 128, False, "BASELINE cfg3: fp8 e4m3fn B=1 H=16 S=16384 d=128 non-causal (B, H chosen: unspecified)"),
    # not a BASELINE config: the shape class of the environment's best known-good structure (cdna_hip_programming.md, "4-wave,
    # one-wave-per-SIMD, persistent structure": 1.25 PFLOP/s on random data with bf16 I/O at N = 2048, D = 128) -- bf16 in AND out
    "anchor": (16, 16, 2048, 128, False, "anchor (not a BASELINE config): bf16 in/out B=16 H=16 S=2048 d=128 non-causal, the guide's 1.25 PFLOP/s shape class"),
}
# element type of O when --out-dtype is not given: fp32 (the reference's float* O) except for the anchor, whose yardstick is bf16 I/O
DEFAULT_OUT_DTYPE = {"anchor": "bf16"}
ANCHOR_GUIDE_TFLOPS = 1250.0   # cdna_hip_programming.md, Appendix B "Fused attention prefill": the hand-placed 4-wave kernel, random data
# parity.pass_frac_at_1e-3 below this fails the run: every sampled element with fp32 output (the stated tolerance, met by the
# default weight precisions: include/flash_attention.h); 2-byte outputs add their own rounding (2^-9 relative for bf16)
PARITY_FLOOR = {"f32": 1.0, "bf16": 0.998}


def flops_of(BH, S, d, causal):
    return (2.0 if causal else 4.0) * BH * float(S) * S * d


def bound_for(workload):
    """(bound, peak TFLOP/s, derivation, issue_bound) of the dominant kernel for a workload.

    `peak` is the dense MFMA peak of the arithmetic the kernel runs -- the metric's own denominator ("% MFMA peak"): 2516.6 for
    bf16; for fp8 inputs (cfg3) QK^T runs on the block-scaled MX MFMA at twice the bf16 rate and P.V at the bf16 rate, so the
    time-weighted mix 1 / (0.5 / 5033.2 + 0.5 / 2516.6).
    `issue_bound` (a labelled extra, never the denominator of `frac`): the SIMD's vector-issue port serves BOTH waves of a SIMD
    one instruction at a time -- an MFMA holds it 8 cycles, v_exp_f32 8, v_fma_f32 / v_add_f32 / v_cvt_pk 4
    (MI355X_MICROARCH.md 'vector-instruction ISSUE cost') -- so per wave and 64-key tile the engine the library launches issues
      32x32x16 (causal bf16):      d/4 MFMAs of 32 pipe cycles + 32 x (fma + exp + add) + 16 cvt_pk
      16x16x32 (non-causal bf16):  d/2 + 4 MFMAs of 16 pipe cycles (4 of them the row sums) + 32 x (fma + exp) + 16 cvt_pk
    and where the issue cycles exceed the MFMA pipe cycles, vector issue and not the pipe is the ceiling of that engine."""
    if workload == "cfg3":
        return ("mfma", 1.0 / (0.5 / (2 * PEAK_BF16_TFLOPS) + 0.5 / PEAK_BF16_TFLOPS),
                "0.5 of the FLOPs at the MX fp8 rate (2x), 0.5 at the bf16 rate", None)
    d, causal = WORKLOADS[workload][3], WORKLOADS[workload][4]
    if causal:      # 32x32x16 engine, fp32 row sums by v_add_f32
        n_mfma, cyc = d // 4, 32
        pipe, useful, issue = cyc * n_mfma, cyc * n_mfma, 8 * n_mfma + 32 * (4 + 8 + 4) + 16 * 4
        eng = "32x32x16"
    else:           # 16x16x32 engine, row sums by 4 ONES.P^T MFMAs
        n_mfma, cyc = d // 2 + 4, 16
        pipe, useful, issue = cyc * n_mfma, cyc * (n_mfma - 4), 8 * n_mfma + 32 * (4 + 8) + 16 * 4
        eng = "16x16x32"
    ib = {"engine": eng, "mfma_pipe_cycles_per_wave_tile": pipe, "vector_issue_cycles_per_wave_tile": issue,
          "ceiling_tflops": round(PEAK_BF16_TFLOPS * useful / max(pipe, issue), 1),
          "binds": "vector issue" if issue > pipe else "mfma pipe"}
    return "mfma", PEAK_BF16_TFLOPS, "nominal dense bf16 MFMA peak: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz", ib


def launched_kernel(fa, B, H, S, d, causal, dtype_code, o_code, flags):
    """Name of the kernel the library launches for this problem, from the plan the library itself reports
    (flash_attention_plan_ex): 256 threads = the pair kernel, else the persistent kernel fa::fwd_mfma_kernel -- for a causal bf16
    problem longer than FA_EARLY_KEYS its mixed-precision instantiation (both ranges of the plan in one launch, unit_lists = 1).
    (profiles/*_kernel_stats_*.csv carry the same names from rocprofv3; round 3's two-kernel launch was fa::fwd_mfma_dual_kernel.)"""
    early, main = fa.plan_ex(B, H, S, S, d, causal, dtype_code, o_code, flags)
    live = main if main["q_blocks"] > 0 else early
    if live["kernel_id"] == 3:
        return "fa::fwd_f32_mfma_kernel"
    if live["kernel_id"] == 0:
        return "fa::fwd_generic_kernel"
    if live["threads"] == 256:
        return "fa::fwd_mfma_pair_kernel"
    return "fa::fwd_mfma_kernel"


def visible_devices():
    """GPUs this process may use.  torch.cuda.device_count() does not initialise the GPU on this image; the test suite injects a
    count through FA_BENCH_VISIBLE_DEVICES (CPU box: tests/test_bench_contract.py)."""
    fake = os.environ.get("FA_BENCH_VISIBLE_DEVICES")
    if fake is not None:
        return int(fake)
    import torch
    return int(torch.cuda.device_count())


def measured_traffic(workload):
    """HBM bytes per launch from the committed PMC run of this same command (separate FETCH_SIZE / WRITE_SIZE passes,
    FETCH_SIZE doubled as the gfx950 guide prescribes) -- profiles/<round>_hbm_traffic_<workload>.json -- but ONLY if that run
    was taken on the kernel sources that are built now (sha256 over csrc/, profiles/provenance.py): a measurement of
    other code is not this run's traffic.  Returns (bytes or None, provenance dict)."""
    import provenance
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_hbm_traffic_{workload}.json")
    now = provenance.csrc_sha256()
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None, {"file": None, "csrc_sha256_now": now}
    same = rec.get("csrc_sha256") == now
    prov = {"file": os.path.relpath(path, ROOT), "csrc_sha256_measured": rec.get("csrc_sha256"), "csrc_sha256_now": now,
            "matches_built_sources": same}
    try:
        return (int(rec["traffic_bytes_per_launch"]) if same else None), prov
    except (KeyError, ValueError):
        return None, prov


def power_limited_ceiling():
    """This device's throughput ceiling as the power limit sets it: tests/micro/simd_mix --ceiling runs, on every
    CU, 2 waves per SIMD of (a) back-to-back v_mfma_f32_32x32x16_bf16 and (b) the attention kernel's own
    per-MFMA instruction mix (softmax VALU ops + K / V^T LDS reads), both on RANDOM bf16 operands, pipe >= 90 %
    busy -- the chip then holds 1.3-1.7 GHz, not the 2.4 GHz behind the 2516.6 TFLOP/s nominal peak.  A separate
    child process started BEFORE this process touches the GPU; None if the microbenchmark is not built."""
    exe = os.path.join(ROOT, "tests", "micro", "simd_mix")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "--ceiling"], capture_output=True, text=True, timeout=120, check=True).stdout
        return json.loads(out.strip().splitlines()[-1])
    except Exception:   # noqa: BLE001 -- a diagnostic extra must never fail the benchmark
        return None


def cpu_baseline_and_parity(Q, K, V, O, S, d, causal, budget_s=12.0, heads_at=None):
    """Naive fp32 attention (the oracle, a port of tests/main.cu:74-91 / check.py:19-21) on the host cores, on a bounded
    sample of THIS run's tensors: as many whole heads of the workload as fit ~budget_s.  The same result is the checker
    for the GPU output of those heads (`parity`).  Q, K, V, O: the rank's [heads, 1, S, d] device tensors."""
    import numpy as np
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity import parity_report

    def timed(h0, nh):
        q, k, v = (np.ascontiguousarray(t[h0:h0 + nh].float().cpu().numpy().reshape(1, nh, S, d)) for t in (Q, K, V))
        ref = np.empty_like(q)
        L = oracle.lib()
        t0 = time.perf_counter()
        thr = L.oracle_attention_f32(oracle._p(q), oracle._p(k), oracle._p(v), oracle._p(ref), 1, nh, S, d,
                                     float(1.0 / np.sqrt(d)), int(causal), 0)
        return time.perf_counter() - t0, thr, ref

    heads = Q.shape[0]
    timed(0, 1)                      # warm-up: thread pool start, first touch
    if heads_at is not None:         # parity only, on the given whole heads (cfg4: either side of the 2^31-byte line, first and last)
        from parity import parity_report as _pr
        gots, refs, t_all, thr = [], [], 0.0, 1
        for h in heads_at:
            t, thr, ref = timed(h, 1)
            t_all += t
            refs.append(ref)
            gots.append(O[h:h + 1].float().cpu().numpy().reshape(1, 1, S, d))
        par = _pr(np.concatenate(gots, 1), np.concatenate(refs, 1))
        par["checked"] = f"GPU output of whole heads {list(heads_at)} of this rank's slab (every row) against the oracle's result on the same (rounded) inputs; {t_all:.2f} s on {int(thr)} host cores"
        return None, par
    t1, thr, _ = timed(0, 1)
    nh = int(max(1, min(heads, budget_s / max(t1, 1e-3))))
    h0 = heads - nh                  # the LAST heads of the slab (the first are the easy ones to get right)
    t, thr, ref = timed(h0, nh)
    got = O[h0:h0 + nh].float().cpu().numpy().reshape(1, nh, S, d)
    base = {"value": round(flops_of(nh, S, d, causal) / t / 1e12, 5), "unit": "TFLOP/s", "cores": int(thr), "kind": "port",
            "sample": f"the last {nh} head(s) of this run's tensors (S={S}, d={d}, causal={causal}), fp32 naive attention, "
                      f"{t:.2f} s wall, OpenMP over query rows"}
    par = parity_report(got, ref)
    par["checked"] = f"GPU output of the same {nh} head(s) (whole heads, every row) against the oracle's result on the same (rounded) inputs"
    return base, par


def run_workload(fa, shard, torch, dist, args, workload, world, rank, dev, steps, warmup, dry, want_parity, out_dtype, prime=True, parity_heads=None):
    """One workload on this rank; returns the rank-0 record (None elsewhere)."""
    B, H, S, d, causal, desc = WORKLOADS[workload]
    if workload == "cfg4":           # fixed total problem, B*H split over the ranks (strong scaling)
        lo, hi = shard.shard_heads(B * H, rank, world)
        heads_local, scaling, total_heads = hi - lo, "strong", B * H
    else:                            # same batch on every rank (weak scaling)
        heads_local, scaling, total_heads = B * H, "weak", B * H * world
    esz = 1 if workload == "cfg3" else 2
    osz = 2 if out_dtype == "bf16" else 4
    Q = K = V = O = None
    if not dry:
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        shape = (heads_local, 1, S, d)   # the rank's slab as a dense [heads,1,S,d] tensor
        in_dtype = torch.float8_e4m3fn if workload == "cfg3" else torch.bfloat16
        Q, K, V = (torch.randn(shape, generator=g, device=dev, dtype=torch.float32).to(in_dtype) for _ in range(3))
        O = torch.empty(shape, device=dev, dtype=torch.bfloat16 if out_dtype == "bf16" else torch.float32)
    scale = 1.0 / d ** 0.5
    wdt = None if (dry or esz != 2) else {"default": None, "bf16": torch.bfloat16, "f16": torch.float16}[args.weights]

    def step():
        if dry:
            time.sleep(2e-4)
        else:
            fa.flash_attention(Q, K, V, O, scale=scale, is_causal=causal, weights_dtype=wdt)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def timed(n):
        """n steps between barriers; (host seconds, per-step HIP-event ms on the launch stream)"""
        sync()
        if dist.is_initialized():
            dist.barrier()
        sync()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)] if not dry else None
        per = []
        t0 = time.perf_counter()
        for i in range(n):
            if ev:
                ev[i].record()       # same stream the kernel is launched on (torch's current stream)
            ts = time.perf_counter()
            step()
            if not ev:
                per.append((time.perf_counter() - ts) * 1e3)
        if ev:
            ev[n].record()
        sync()
        if dist.is_initialized():
            dist.barrier()
        sync()
        el = time.perf_counter() - t0
        if ev:
            per = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
        return el, per

    # first touch (LDS limit raised, code object loaded), then the UNPRIMED figure: the same K steps from a device that has
    # just come out of idle -- what a caller sees on its first few calls (clock ramp; reported beside `value`, never as it)
    step(); sync()
    el_cold, _ = timed(steps)
    # Device priming (setup, untimed, before the W warmup steps): a GPU coming out of idle needs tens of milliseconds of
    # work before its power state and clocks settle.  ~100 ms of the same launches, at most 3000.
    if prime:
        t_p = time.perf_counter(); step(); step(); sync()
        t_step = max((time.perf_counter() - t_p) / 2, 1e-6)
        for _ in range(max(0, min(3000, int(0.1 / t_step)))):
            step()
        sync()
    for _ in range(warmup):
        step()
    elapsed, per_step = timed(steps)
    elapsed = shard.reduce_max(elapsed)
    el_cold = shard.reduce_max(el_cold)
    kernel_ms = sum(per_step) / len(per_step)
    kernel_ms_max = shard.reduce_max(kernel_ms)
    ms_median = shard.reduce_max(statistics.median(per_step))
    ms_min = shard.reduce_max(min(per_step))

    ok = True
    if not dry:                      # sanity: the output is finite and row 0 of a causal head equals V[0]
        ok = bool(torch.isfinite(O.float()).all()) if heads_local <= 256 else bool(torch.isfinite(O[:64].float()).all() and torch.isfinite(O[-64:].float()).all())
        if causal:
            ok = ok and bool(torch.allclose(O[:, :, 0].float(), V[:, :, 0].float(), rtol=1e-2, atol=1e-2))
    ok_all = shard.reduce_sum(0.0 if ok else 1.0) == 0.0
    if rank != 0:
        return None

    ms_per_step = elapsed / steps * 1e3
    value = flops_of(total_heads, S, d, causal) / (ms_per_step * 1e-3) / 1e12
    achieved = flops_of(heads_local, S, d, causal) / (kernel_ms_max * 1e-3) / 1e12
    bound, peak, why, issue_bound = bound_for(workload)
    # (the committed PMC run is of the default call: fp32 output, default weight precision)
    default_call = out_dtype == DEFAULT_OUT_DTYPE.get(workload, "f32") and args.weights == "default"
    traffic, prov = measured_traffic(workload) if world == 1 and default_call else (None, None)
    algo_bytes = heads_local * S * d * (3 * esz + osz)
    flags = {"default": 0, "bf16": fa.FA_FLAG_BF16_WEIGHTS, "f16": fa.FA_FLAG_F16_WEIGHTS}[args.weights] if esz == 2 else 0
    dcode, ocode = (fa.FA_DTYPE_FP8_E4M3 if esz == 1 else fa.FA_DTYPE_BF16), (fa.FA_DTYPE_BF16 if out_dtype == "bf16" else fa.FA_DTYPE_F32)
    kernel = launched_kernel(fa, heads_local, 1, S, d, causal, dcode, ocode, flags)
    pe, pm = fa.plan_ex(heads_local, 1, S, S, d, causal, dcode, ocode, flags)
    kernel_variant = ("mixed precision in one walk: fp16 weights on query blocks [0, %d), bf16 on [%d, %d)" % (pe["q_blocks"], pe["q_blocks"], pe["q_blocks"] + pm["q_blocks"])
                      if pe["q_blocks"] and pm["q_blocks"] and pm["unit_lists"] == 1 and pm["threads"] == 512
                      else ("fp16 weights" if pe["q_blocks"] else "one weight precision (bf16)") + (", 128-row units (pair kernel)" if (pm if pm["q_blocks"] else pe)["threads"] == 256 else ""))
    rec = {
        "metric": METRIC if workload.startswith("cfg2") else f"fwd attention TFLOP/s/GPU ({desc}) + % MFMA peak",
        "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "fp8_e4m3fn" if workload == "cfg3" else "bf16",
        "data": "synthetic",
        "config": {"workload": desc, "B": B, "H": H, "S": S, "d": d, "causal": causal,
                   "kernel_variant": kernel_variant,
                   "out_dtype": out_dtype, "heads_per_gpu": heads_local,
                   "softmax_weights": ({"default": "library default: fp16 on the query rows that see fewer than 1024 keys, bf16 elsewhere",
                                        "bf16": "FA_FLAG_BF16_WEIGHTS: bf16 on every row", "f16": "FA_FLAG_F16_WEIGHTS: fp16 on every row"}[args.weights]
                                       if esz == 2 else "bf16"),
                   "flop_convention": "2*B*H*S^2*d causal / 4*B*H*S^2*d non-causal",
                   "parallelism": f"batch x head shard over {world} GPU(s), no data-path collective"},
        "rccl_ranks": world,
        "value_per_gpu": round(value / world, 2),
        "pct_of_bf16_mfma_peak": round(100.0 * value / (PEAK_BF16_TFLOPS * world), 2),
        # per-step HIP-event times of the K timed steps (median and min beside the mean that defines `value`)
        "ms_median": round(ms_median, 5), "ms_min": round(ms_min, 5),
        "value_at_ms_median": round(flops_of(total_heads, S, d, causal) / (ms_median * 1e-3) / 1e12, 2),
        "device_primed": bool(prime),
        # the same K steps timed BEFORE the ~100 ms priming loop (device just out of idle): not the metric, the caveat
        "value_unprimed": round(flops_of(total_heads, S, d, causal) / (el_cold / steps) / 1e12, 2),
        # causal FLOPs count only the unmasked half; the full-count figure alongside, labelled
        "value_if_masked_half_counted_too": round(value * (2.0 if causal else 1.0), 2),
        "output_ok": ok_all,
        "roofline": {"bound": bound, "achieved": round(achieved, 2), "peak": round(peak, 1),
                     "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "peak_derivation": why,
                     "issue_bound": issue_bound,
                     "traffic": traffic, "traffic_provenance": prov,
                     # HBM bytes the counters saw per algorithmic byte (1.0 = every byte moved once; > 1 = re-reads beyond the XCD's L2)
                     "traffic_ratio": round(traffic / algo_bytes, 3) if traffic else None,
                     # from the library's own plan for this call (flash_attention_plan_ex), not a literal
                     "kernel": kernel, "kernel_ms": round(kernel_ms_max, 5),
                     "algorithmic_hbm_bytes": algo_bytes,
                     "algorithmic_hbm_GBps": round(algo_bytes / (kernel_ms_max * 1e-3) / 1e9, 1)},
    }
    if dry:
        rec["dry_run"] = True
    if workload == "anchor":
        rec["anchor"] = {"guide_best_known_tflops": ANCHOR_GUIDE_TFLOPS, "ratio_to_guide": round(value / world / ANCHOR_GUIDE_TFLOPS, 4),
                         "source": "cdna_hip_programming.md Appendix B: 4-wave one-wave-per-SIMD persistent kernel, bf16 I/O, N=2048 D=128, random data"}
    if (want_parity or parity_heads) and not dry:
        hs = [h for h in (parity_heads or []) if 0 <= h < heads_local] or None
        base, rec["parity"] = cpu_baseline_and_parity(Q, K, V, O, S, d, causal, budget_s=args.cpu_budget_s, heads_at=hs)
        if base is not None:
            rec["cpu_baseline"] = base
        floor = PARITY_FLOOR[out_dtype]
        rec["parity"]["floor"] = floor
        if rec["parity"]["pass_frac_at_1e-3"] < floor:
            rec["output_ok"] = False
    return rec


def launch_ranks(n, argv):
    """Parent of an N-rank run: start N fresh processes (one per GPU) through torch.distributed.run, relay their output, return
    their exit status.  This process never initialises the GPU (no exec from a process that has)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py")] + argv
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the power-limited-ceiling microbenchmark")
    ap.add_argument("--no-cfg4", action="store_true", help="skip the cfg4 strong-scaling sub-record")
    ap.add_argument("--no-bf16-out", action="store_true", help="skip the bf16-output sub-record")
    ap.add_argument("--out-dtype", default=None, choices=["bf16", "f32"],
                    help="element type of O; default f32 = the reference's float* O (kernels/FlashAttention.cuh:61); the anchor workload: bf16")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0, help="seconds of host-core time for the CPU baseline / parity sample (whole heads)")
    ap.add_argument("--weights", default="default", choices=["default", "bf16", "f16"],
                    help="softmax-weight precision (bf16 inputs): the library default, or one precision on every row (FA_FLAG_*_WEIGHTS)")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo: no launches, no measurements -- the line's structure only")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.out_dtype is None:
        args.out_dtype = DEFAULT_OUT_DTYPE.get(args.workload, "f32")

    # More ranks than visible GPUs: say so in one line and leave with a status of its own, BEFORE any rendezvous (the parent checks
    # before it starts the ranks, a rank started by someone else's torchrun before init_process_group).  Not for --dry-run (CPU, gloo)
    # unless a count is injected (FA_BENCH_VISIBLE_DEVICES: the test suite).
    if not args.dry_run or "FA_BENCH_VISIBLE_DEVICES" in os.environ:
        n_vis = visible_devices()
        if n_vis < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {n_vis} GPU(s) visible to this process: nothing started", file=sys.stderr, flush=True)
            return 4

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py with --gpus equal to the number of ranks")

    # The ceiling microbenchmark is a separate GPU program: run it as a child BEFORE this process touches the GPU
    # (no exec from a process that has initialised HIP).  Single-GPU runs only.
    ceil = None
    if world == 1 and not args.no_ceiling and not args.dry_run and args.workload != "cfg3":
        ceil = power_limited_ceiling()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    fa = entry.load_package()
    from flash_attention_cuda_c_amd import shard

    dev = None
    # FA_BENCH_RCCL_SINGLE=1: a ONE-rank process group over RCCL -- the N > 1 code path (init with device_id, barriers, the MAX / SUM
    # all-reduces) on a box that has a single GPU (tests/test_bench_gpu.py); the timed region is the same
    rccl_single = world == 1 and os.environ.get("FA_BENCH_RCCL_SINGLE") == "1" and not args.dry_run
    if args.dry_run:
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1 or rccl_single:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if rccl_single:
                import socket
                sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(port))
            # RCCL prints a version banner on STDOUT when its communicator comes up: keep stdout for the one JSON line (the banner goes to
            # stderr: file descriptor 1 points at 2 while the group is created and the first collective runs)
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                dist.barrier()
                torch.cuda.synchronize()
            finally:
                os.dup2(saved, 1)
                os.close(saved)
    ranks = dist.get_world_size() if dist.is_initialized() else 1

    line = run_workload(fa, shard, torch, dist, args, args.workload, world, rank, dev, args.steps, args.warmup, args.dry_run,
                        want_parity=world == 1 and not args.no_cpu_baseline, out_dtype=args.out_dtype)
    if rank == 0:
        line["rccl_ranks"] = ranks
        if ceil:
            a = line["roofline"]["achieved"]
            line["roofline"]["power_limited"] = dict(
                ceil, frac_of_mfma_only=round(a / ceil["mfma_only_random_bf16_tflops"], 4),
                frac_of_attention_mix=round(a / ceil["attention_mix_random_bf16_tflops"], 4))
    few = max(2, min(5, args.steps))
    # the same workload with bf16 output (half the output bytes; its rounding alone exceeds the stated tolerance for |O| > 1)
    if not args.no_bf16_out and args.out_dtype != "bf16" and args.workload != "cfg4":
        sub = run_workload(fa, shard, torch, dist, args, args.workload, world, rank, dev, args.steps, args.warmup, args.dry_run, False, "bf16", prime=False)
        if rank == 0:
            line["bf16_out"] = {k: sub[k] for k in ("value", "unit", "ms_per_step", "ms_median", "ms_min", "output_ok")}
            line["bf16_out"]["roofline"] = {k: sub["roofline"][k] for k in ("achieved", "peak", "frac", "kernel_ms")}
    # BASELINE configs[4] itself, its 2048 heads split over the ranks (the driver never passes --workload cfg4); N = 1: the anchor
    if not args.no_cfg4 and args.workload != "cfg4":
        # parity of cfg4 at the stated tolerance on whole heads of rank 0's slab: the first, the two either side of the 2^31-byte
        # line of the whole problem (heads 1023 | 1024; on a shard: its middle), and the last
        n_loc = shard.shard_heads(64 * 32, rank, world)
        n_loc = n_loc[1] - n_loc[0]
        c4_heads = sorted({0, n_loc // 2 - 1, n_loc // 2, n_loc - 1}) if (world == 1 and not args.no_cpu_baseline) else None
        sub = run_workload(fa, shard, torch, dist, args, "cfg4", world, rank, dev, few, 1, args.dry_run, False, args.out_dtype, prime=False,
                           parity_heads=c4_heads)
        if rank == 0:
            line["cfg4"] = {k: sub[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_median", "ms_min", "scaling", "config",
                                                 "value_per_gpu", "pct_of_bf16_mfma_peak", "output_ok")}
            line["cfg4"]["roofline"] = {k: sub["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_ms")}
            if "parity" in sub:
                line["cfg4"]["parity"] = sub["parity"]
    rc = 0
    if rank == 0:
        if not line["output_ok"]:
            rc = 3
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
