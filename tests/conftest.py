"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Output of the C++ driver (flash-attention-cuda-c_amd/fa_main) and the C++ harness (tests/fa_test), run as CHILD
# processes at session start -- before this process initialises the GPU -- when the session selects the gpu tests
# on a box that has a GPU.  tests/test_driver.py asserts on what they printed.
DRIVER_RUNS = {}


def _run_child(name, argv, timeout, env=None):
    import subprocess
    try:
        r = subprocess.run(argv, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
        DRIVER_RUNS[name] = {"argv": argv, "rc": r.returncode, "stdout": r.stdout, "stderr": r.stderr}
    except Exception as e:  # noqa: BLE001 -- recorded, the test reports it
        DRIVER_RUNS[name] = {"argv": argv, "rc": None, "stdout": "", "stderr": f"{type(e).__name__}: {e}"}
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"session_{name}.log"), "w") as f:
            f.write(f"$ {' '.join(argv)}\nrc = {DRIVER_RUNS[name]['rc']}\n{DRIVER_RUNS[name]['stdout']}\n{DRIVER_RUNS[name]['stderr']}")
    except OSError:
        pass


def pytest_sessionstart(session):
    expr = session.config.option.markexpr or ""
    if "gpu" not in expr or "not gpu" in expr or not os.path.exists("/dev/kfd"):
        return
    fa_main = os.path.join(ROOT, "flash-attention-cuda-c_amd", "fa_main")
    fa_test = os.path.join(ROOT, "tests", "fa_test")
    unit = os.path.join(ROOT, "tests", "unit_kernels")
    if os.path.exists(fa_main):
        _run_child("fa_main", [fa_main, "--props", "--config", "1", "--config", "2", "--gpus", "1", "--iters", "10"], 600)
    if os.path.exists(fa_test):
        _run_child("fa_test", [fa_test, "--quick"], 600)
    if os.path.exists(unit):
        _run_child("unit_kernels", [unit], 300)
    # bench.py itself (tests/test_bench_gpu.py): one rank over RCCL -- the N > 1 code path on the one GPU there is -- and the refusal of
    # more ranks than GPUs.  Children of a process that has not touched the GPU yet, like the programs above.
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE",
                                                              "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    bench = [sys.executable, os.path.join(ROOT, "bench.py")]
    _run_child("bench_rccl_single", bench + ["--steps", "5", "--warmup", "2", "--no-ceiling", "--no-cfg4", "--no-bf16-out", "--cpu-budget-s", "3"], 600,
               env=dict(clean, FA_BENCH_RCCL_SINGLE="1"))
    _run_child("bench_gpus2", bench + ["--gpus", "2", "--steps", "2", "--warmup", "1"], 300, env=clean)


@pytest.fixture(scope="session")
def driver_runs():
    return DRIVER_RUNS


class Golden:
    """Loader for tests/golden (data minted from the reference's check.py; see make_golden.py)."""

    def __init__(self):
        with open(os.path.join(GOLDEN, "manifest.json")) as f:
            self.manifest = json.load(f)["fixtures"]

    def meta(self, name):
        return self.manifest[name]

    def load(self, name, key):
        entry = self.manifest[name]["files"][key]
        return np.load(os.path.join(GOLDEN, entry["file"]), allow_pickle=False)

    @staticmethod
    def to_bhsd(x, num_heads):
        """(B,S,H*d_k) -> [B,H,S,d_k] (check.py:14-16)."""
        B, S, dm = x.shape
        return np.ascontiguousarray(x.reshape(B, S, num_heads, dm // num_heads).transpose(0, 2, 1, 3))

    @staticmethod
    def to_bsd(x):
        """[B,H,S,d_k] -> (B,S,H*d_k) (check.py:24)."""
        B, H, S, d = x.shape
        return np.ascontiguousarray(x.transpose(0, 2, 1, 3).reshape(B, S, H * d))


@pytest.fixture(scope="session")
def golden():
    return Golden()
