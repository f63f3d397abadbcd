"""Turn the rocprofv3 output directories written by profiles/collect.sh into the small files kept under profiles/:
  gpurun_out/<tag>_kernel_stats.csv   the --stats table, our kernels only (torch's kernel names run to kilobytes)
  gpurun_out/<tag>_hbm_traffic.json   FETCH_SIZE / WRITE_SIZE per launch of the attention kernel, FETCH doubled as
                                      MI355X_MICROARCH.md's HBM section prescribes for gfx950, counters in KiB
  gpurun_out/<tag>_sq_counters.json   MFMA pipe busy fraction, sustained clock, LDS bank conflicts
"""
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
# Q, K, V read once + O written once; bench.py's default output type is fp32 (4 bytes)
ALGO = {"cfg2": 8 * 16 * 4096 * 128 * (3 * 2 + 4), "cfg2nc": 8 * 16 * 4096 * 128 * (3 * 2 + 4), "cfg1": 4 * 8 * 2048 * 64 * (3 * 2 + 4),
        "cfg3": 1 * 16 * 16384 * 128 * (3 * 1 + 4)}
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import provenance  # noqa: E402


def one(pattern):
    hits = glob.glob(os.path.join(OUT, pattern), recursive=True)
    return hits[0] if hits else None


def counters(tag, suffix):
    path = one(f"{tag}_{suffix}/**/*counter_collection.csv")
    rows = {}
    if not path:
        return rows
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if "fwd_mfma_" in r["Kernel_Name"] or "fwd_f32_mfma" in r["Kernel_Name"]:   # (fwd_mfma_kernel, fwd_mfma_dual_kernel)
                rows.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]),
                                                               int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return rows


def main():
    tag, wl = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "cfg2")
    stats = one(f"{tag}_stats/**/*kernel_stats.csv")
    if stats:
        with open(stats, newline="") as f, open(os.path.join(OUT, f"{tag}_kernel_stats.csv"), "w", newline="") as g:
            w = csv.writer(g)
            for i, r in enumerate(csv.reader(f)):
                if i == 0 or r[0].startswith("_ZN2fa") or "fa::" in r[0]:
                    w.writerow([r[0][:160]] + r[1:])
    fetch, write = counters(tag, "fetch").get("FETCH_SIZE", []), counters(tag, "write").get("WRITE_SIZE", [])
    if fetch and write:
        rd = 2.0 * 1024 * sum(v for v, _ in fetch) / len(fetch)
        wr = 1024.0 * sum(v for v, _ in write) / len(write)
        json.dump({"workload": wl, "launches": len(fetch), "note": "FETCH_SIZE doubled (gfx950), counters in KiB, separate --pmc passes",
                   "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "traffic_bytes_per_launch": rd + wr,
                   "algorithmic_bytes_per_launch": ALGO.get(wl), "csrc_sha256": provenance.csrc_sha256()}, open(os.path.join(OUT, f"{tag}_hbm_traffic.json"), "w"), indent=1)
    sq = counters(tag, "sq")
    if sq:
        n = len(sq["GRBM_GUI_ACTIVE"])
        avg = {k: sum(v for v, _ in rows) / len(rows) for k, rows in sq.items()}
        dur_ns = sum(d for _, d in sq["GRBM_GUI_ACTIVE"]) / n
        cyc = avg["GRBM_GUI_ACTIVE"] / 8.0                       # summed over the 8 XCDs
        json.dump({"workload": wl, "launches": n, "kernel_us_under_pmc": dur_ns / 1e3, "cycles_per_launch": cyc,
                   "clock_ghz": cyc / dur_ns, "mfma_busy_frac": avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc,
                   "lds_bank_conflict_cycles": avg.get("SQ_LDS_BANK_CONFLICT"), "raw": avg},
                  open(os.path.join(OUT, f"{tag}_sq_counters.json"), "w"), indent=1)
    print("summaries written under gpurun_out/ for", tag)


if __name__ == "__main__":
    main()
