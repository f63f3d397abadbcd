"""bench.py on the GPU box: the RCCL code path on the one GPU there is, and the refusal of more ranks than GPUs.  Both runs are made by
tests/conftest.py at session start (children of a process that has not initialised the GPU); these tests read what they printed."""
import json

import pytest

pytestmark = pytest.mark.gpu


def _run(driver_runs, name):
    if name not in driver_runs:
        pytest.skip("bench.py was not run at session start (no GPU session)")
    return driver_runs[name]


def test_one_rank_over_rccl_runs_the_multi_gpu_code_path(driver_runs):
    """`bench.py --gpus N` for N > 1 initialises a process group over RCCL (`nccl`, device_id), brackets the timed region with
    barriers and reduces elapsed time (MAX) and the ok flag (SUM) over the ranks.  The box has one GPU, so that path runs here with
    ONE rank (FA_BENCH_RCCL_SINGLE=1): the same calls on real hardware, `rccl_ranks` = 1 in the line, parity still green."""
    r = _run(driver_runs, "bench_rccl_single")
    assert r["rc"] == 0, r["stderr"][-3000:]
    lines = [l for l in r["stdout"].splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r["stdout"][-2000:]      # ONE line on stdout: RCCL's banner goes to stderr
    line = json.loads(lines[0])
    assert line["rccl_ranks"] == 1 and line["n_gpus"] == 1 and line["output_ok"] is True
    assert line["parity"]["pass_frac_at_1e-3"] == 1.0 and line["value"] > 100
    assert line["roofline"]["kernel"] == "fa::fwd_mfma_kernel" and line["config"]["kernel_variant"].startswith("mixed precision in one walk")


def test_more_ranks_than_gpus_on_the_real_box(driver_runs):
    """--gpus 2 on the one-GPU box: refused with one line and exit status 4 before anything is started (the real device count)."""
    r = _run(driver_runs, "bench_gpus2")
    if r["rc"] == 0:
        pytest.skip("this box has several GPUs: the two ranks ran")
    assert r["rc"] == 4 and "only 1 GPU(s) visible" in r["stderr"] and r["stdout"].strip() == ""
