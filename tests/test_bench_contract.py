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


def test_committed_traffic_measurement_is_found():
    t = bench.measured_traffic("cfg2")
    assert isinstance(t, int) and 4 * 8 * 16 * 4096 * 128 * 2 <= t < 2 * 4 * 8 * 16 * 4096 * 128 * 2   # >= algorithmic, < 2x
    assert bench.measured_traffic("no-such-workload") is None
