"""CPU checks of bench.py's side of the measurement contract: the workloads are BASELINE.json's configs, the FLOP
convention is SURVEY.md section 8d's, and the committed profile inputs it reads are where it expects them."""
import json
import os
import re

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_workloads_are_the_baseline_configs():
    cfgs = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]

    def shape(text):
        g = lambda k: int(re.search(rf"{k}=(\d+)", text).group(1))
        return g("B"), g("H"), g("seq"), g("d")

    for name, idx in (("cfg1", 1), ("cfg2", 2), ("cfg4", 4)):
        B, H, S, d, causal, desc = bench.WORKLOADS[name]
        assert (B, H, S, d) == shape(cfgs[idx]) and causal == ("causal" in cfgs[idx] and "non-causal" not in cfgs[idx])
    B, H, S, d, causal, desc = bench.WORKLOADS["cfg3"]                    # BASELINE.json gives only seq and d for cfg3
    assert (S, d) == (16384, 128) and "fp8" in cfgs[3] and "unspecified" in desc
    assert bench.WORKLOADS["cfg2nc"][:4] == bench.WORKLOADS["cfg2"][:4] and not bench.WORKLOADS["cfg2nc"][4]


def test_flop_convention_and_peak():
    # 4*B*H*S^2*d non-causal (QK^T + PV), half of it under the causal mask
    assert bench.flops_of(128, 4096, 128, False) == 4.0 * 128 * 4096 * 4096 * 128 == 1.099511627776e12
    assert bench.flops_of(128, 4096, 128, True) == 0.5 * bench.flops_of(128, 4096, 128, False)
    assert bench.PEAK_BF16_TFLOPS == 256 * 4 * 1024 * 2.4e9 / 1e12 == 2516.5824 or abs(bench.PEAK_BF16_TFLOPS - 2516.58) < 0.05


def test_provenance_hash_ignores_comments_and_white_space_only():
    """profiles/provenance.py: a measurement stays attached to the sources while only comments / layout change."""
    import provenance
    a = 'int a = 1;  // one\n/* block\n comment */ const char* s = "// kept /* kept */";\nchar q = \'"\';\n'
    b = 'int a=1;\nconst char* s   =   "// kept /* kept */"; // trailing\nchar q = \'"\';'
    assert provenance.code_only(a) == provenance.code_only(b)
    assert provenance.code_only(a) != provenance.code_only(a.replace("1", "2"))
    assert provenance.code_only(a) != provenance.code_only(a.replace("// kept", "// changed"))


def test_traffic_is_reported_only_with_matching_provenance(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC run, and only while the kernel sources are the ones it was taken on."""
    import provenance
    now = provenance.csrc_sha256()
    assert len(now) == 64 and now == provenance.csrc_sha256()
    t, prov = bench.measured_traffic("no-such-workload")
    assert t is None and prov["file"] is None
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    name = f"{bench.PROFILE_ROUND}_hbm_traffic_cfg2.json"
    (prof / name).write_text(json.dumps({"traffic_bytes_per_launch": 600000000.0, "csrc_sha256": now}))
    t, prov = bench.measured_traffic("cfg2")
    assert t == 600000000 and prov["matches_built_sources"] is True
    (prof / name).write_text(json.dumps({"traffic_bytes_per_launch": 600000000.0, "csrc_sha256": "0" * 64}))
    t, prov = bench.measured_traffic("cfg2")
    assert t is None and prov["matches_built_sources"] is False


def test_metric_string_and_bounds():
    assert bench.METRIC == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    # `peak` (the denominator of roofline.frac) is the nominal dense MFMA peak of the arithmetic, whatever the workload; the
    # vector-issue ceiling of the engine the library launches is a labelled extra
    b, peak, _, ib = bench.bound_for("cfg2")      # causal: 32x32x16 engine, 32 MFMAs of 32 cycles against 32*16 + 16*4 + 32*8 issue cycles
    assert b == "mfma" and abs(peak - 2516.6) < 0.1 and ib["engine"] == "32x32x16"
    assert (ib["mfma_pipe_cycles_per_wave_tile"], ib["vector_issue_cycles_per_wave_tile"]) == (1024, 832) and ib["binds"] == "mfma pipe"
    b, peak, _, ib = bench.bound_for("cfg2nc")    # 16x16x32 engine: 68 MFMAs of 16 cycles, 4 of them row sums
    assert abs(peak - 2516.6) < 0.1 and (ib["mfma_pipe_cycles_per_wave_tile"], ib["vector_issue_cycles_per_wave_tile"]) == (1088, 992)
    b, peak, _, ib = bench.bound_for("cfg1")      # d = 64: vector issue (736 cycles) outweighs the MFMA pipe (576) per wave-tile
    assert b == "mfma" and abs(peak - 2516.6) < 0.1 and ib["binds"] == "vector issue" and abs(ib["ceiling_tflops"] - 2516.6 * 512 / 736) < 0.1
    b, peak, _, ib = bench.bound_for("cfg3")
    assert b == "mfma" and abs(peak - 3355.5) < 0.1 and ib is None


def _bench_rank(rank, world, port, q):
    import io
    import contextlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1", "--dry-run", "--no-ceiling"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.main()
    q.put((rank, buf.getvalue() if rc == 0 else f"rc={rc}"))


def test_two_rank_dry_run_line_carries_the_cfg4_sub_record():
    """world_size 2 over gloo, no GPU: the N > 1 line's structure -- weak-scaling headline + the cfg4 strong-scaling sub-record
    (BASELINE configs[4]: 2048 heads split over the ranks)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs[1].strip() == ""                               # only rank 0 prints
    lines = [l for l in outs[0].splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["scaling"] == "weak" and line["metric"] == bench.METRIC
    assert line["config"]["out_dtype"] == "f32" and {"ms_median", "ms_min", "value_unprimed", "device_primed"} <= set(line) and "bf16_out" in line
    assert line["config"]["heads_per_gpu"] == 128 and line["steps"] == 3
    sub = line["cfg4"]
    assert sub["scaling"] == "strong" and sub["n_gpus"] == 2 and sub["config"]["heads_per_gpu"] == 1024
    assert (sub["config"]["B"], sub["config"]["H"], sub["config"]["S"], sub["config"]["d"]) == (64, 32, 8192, 128)
    assert {"bound", "achieved", "peak", "frac"} <= set(sub["roofline"])


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE",
                                                            "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    return env


def test_gpus_n_starts_n_ranks_itself():
    """`bench.py --gpus 2` from a clean environment (no RANK / WORLD_SIZE): the parent starts the two ranks itself
    (torch.distributed.run, 127.0.0.1), relays rank 0's ONE JSON line and exits 0.  CPU / gloo dry run: structure only."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--no-ceiling"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dry_run"] is True
    assert line["cfg4"]["config"]["heads_per_gpu"] == 1024 and line["cfg4"]["n_gpus"] == 2 and line["cfg4"]["scaling"] == "strong"


def test_single_rank_line_carries_the_cfg4_anchor():
    """N = 1 (dry run): the line still carries BASELINE cfg4 -- all 2048 heads on the one rank: the strong-scaling anchor."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--dry-run", "--no-ceiling"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["cfg4"]["config"]["heads_per_gpu"] == 2048
    assert line["config"]["out_dtype"] == "f32" and line["bf16_out"]["unit"] == "TFLOP/s"


def test_gpus_must_match_the_world():
    import subprocess
    import sys
    env = dict(_clean_env(), RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_roofline_kernel_name_comes_from_the_plan():
    """roofline.kernel is what the library's own plan says the call launches (flash_attention_plan_ex), per workload: the persistent
    kernel (for the causal headline its mixed-precision instantiation -- config.kernel_variant says so; round 3 launched
    fa::fwd_mfma_dual_kernel there: profiles/r03_kernel_stats_cfg2.csv), the pair kernel for cfg1's shape under the mask."""
    import __graft_entry__ as entry
    fa = entry.load_package()
    want = {"cfg2": "fa::fwd_mfma_kernel", "cfg2nc": "fa::fwd_mfma_kernel", "cfg1": "fa::fwd_mfma_kernel", "cfg1c": "fa::fwd_mfma_pair_kernel",
            "cfg4": "fa::fwd_mfma_kernel", "cfg3": "fa::fwd_mfma_kernel", "anchor": "fa::fwd_mfma_kernel"}
    for wl, name in want.items():
        B, H, S, d, causal, _ = bench.WORKLOADS[wl]
        dt = fa.FA_DTYPE_FP8_E4M3 if wl == "cfg3" else fa.FA_DTYPE_BF16
        assert bench.launched_kernel(fa, B * H, 1, S, d, causal, dt, fa.FA_DTYPE_F32, 0) == name, wl
    # one precision on every row: a single kernel again
    B, H, S, d, causal, _ = bench.WORKLOADS["cfg2"]
    assert bench.launched_kernel(fa, B * H, 1, S, d, causal, fa.FA_DTYPE_BF16, fa.FA_DTYPE_F32, fa.FA_FLAG_BF16_WEIGHTS) == "fa::fwd_mfma_kernel"
    # and the dry-run line carries it, with the traffic ratio's slot, for the headline and the cfg4 sub-record
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--dry-run", "--no-ceiling"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["roofline"]["kernel"] == "fa::fwd_mfma_kernel" and "traffic_ratio" in line["roofline"]
    assert line["config"]["kernel_variant"].startswith("mixed precision in one walk: fp16 weights on query blocks [0, 4), bf16 on [4, 16)")
    assert line["cfg4"]["roofline"]["kernel"] == "fa::fwd_mfma_kernel"


def test_anchor_workload_line():
    """`--workload anchor`: bf16 in AND out, non-causal, S = 2048, d = 128, 2048 units -- the shape class of the guide's best known-good
    structure (1.25 PFLOP/s); the line says how far from it the run is."""
    import subprocess
    import sys
    B, H, S, d, causal, _ = bench.WORKLOADS["anchor"]
    assert (S, d, causal) == (2048, 128, False) and B * H * (S // 256) >= 2048
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "anchor", "--steps", "2", "--warmup", "1", "--dry-run", "--no-ceiling",
                        "--no-cfg4"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["config"]["out_dtype"] == "bf16" and line["anchor"]["guide_best_known_tflops"] == 1250.0 and "ratio_to_guide" in line["anchor"]
    assert "bf16_out" not in line


def test_more_ranks_than_visible_gpus_is_refused_before_any_rendezvous():
    """`--gpus N` with fewer than N GPUs visible: one stated line, exit status 4, nothing started (no torchrun child, no
    init_process_group) -- from the parent of an N-rank run and from a rank started under someone else's torchrun alike.  The count is
    injected (this box has no GPU): FA_BENCH_VISIBLE_DEVICES."""
    import subprocess
    import sys
    import time
    exe = [sys.executable, os.path.join(ROOT, "bench.py")]
    t0 = time.time()
    r = subprocess.run(exe + ["--gpus", "8", "--dry-run"], env=dict(_clean_env(), FA_BENCH_VISIBLE_DEVICES="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 4 and "--gpus 8 but only 1 GPU(s) visible" in r.stderr and r.stdout.strip() == "", (r.returncode, r.stderr[-500:])
    assert time.time() - t0 < 60
    env = dict(_clean_env(), RANK="1", WORLD_SIZE="2", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29998", FA_BENCH_VISIBLE_DEVICES="1")
    r = subprocess.run(exe + ["--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)   # (would hang in the rendezvous otherwise)
    assert r.returncode == 4 and "only 1 GPU(s) visible" in r.stderr
    # enough devices: goes on as before
    r = subprocess.run(exe + ["--gpus", "1", "--steps", "2", "--warmup", "1", "--dry-run", "--no-ceiling", "--no-cfg4", "--no-bf16-out"],
                       env=dict(_clean_env(), FA_BENCH_VISIBLE_DEVICES="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
