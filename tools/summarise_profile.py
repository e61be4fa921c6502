#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_gpu.sh into the tracked summaries under profiles/.

    python tools/summarise_profile.py r01            # reads gpurun_out/prof_r01, writes profiles/r01_*

Writes
  profiles/<tag>_bench_kernel_stats.csv    the --kernel-trace --stats per-kernel table, verbatim
  profiles/<tag>_bench_pmc_summary.json    mean per launch of every PMC counter, dominant kernel only
  profiles/<tag>_traffic_bytes_per_launch.json   HBM bytes per launch (bench.py reads this for roofline.traffic)
Every summary carries nrldpc_build_id / nrldpc_kernel_id of the library that was profiled (gpurun_out/prof_<tag>/ids.json).
The kernel-trace average excludes the first (warm-up) launch of the dominant kernel.

HBM correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB; on
gfx950 FETCH_SIZE counts each 128-byte request as 64 bytes, so fetched bytes = 2 * FETCH_SIZE * 1024.
The factor was cross-checked on this kernel: 2 * FETCH_SIZE matches the known LLR input size, and
WRITE_SIZE matches the hard-bit output size exactly.
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_KEY = "nrldpc_decode"


def per_launch(path, kernel_name):
    acc = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Kernel_Name"] != kernel_name:
                continue
            d = acc.setdefault(row["Counter_Name"], {})
            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    return {k: {"mean_per_launch": sum(v.values()) / len(v), "launches": len(v)} for k, v in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = os.path.join(src, "stats", "stats_kernel_stats.csv")
    shutil.copy(stats, os.path.join(dst, tag + "_bench_kernel_stats.csv"))
    with open(stats, newline="") as f:
        rows = [r for r in csv.DictReader(f) if KERNEL_KEY in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    kernel = rows[0]["Name"]
    print("dominant kernel:", kernel, "avg ns", rows[0]["AverageNs"], "calls", rows[0]["Calls"])

    summary = {}
    for name in ("sqA", "sqB", "grbm", "fetch", "write"):
        p = os.path.join(src, name, name + "_counter_collection.csv")
        if os.path.exists(p):
            summary.update(per_launch(p, kernel))
    summary["_kernel"] = kernel
    summary["_avg_ns_kernel_trace_all_launches"] = float(rows[0]["AverageNs"])
    ids = {}
    try:
        ids = json.load(open(os.path.join(src, "ids.json")))
    except (OSError, ValueError):
        pass
    summary["_nrldpc_build_id"] = ids.get("nrldpc_build_id")
    summary["_nrldpc_kernel_id"] = ids.get("nrldpc_kernel_id")
    # per-launch durations of the dominant kernel from the kernel trace, first (warm-up) launch excluded
    trace = os.path.join(src, "stats", "stats_kernel_trace.csv")
    if os.path.exists(trace):
        with open(trace, newline="") as f:
            d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(f)
                 if r["Kernel_Name"] == kernel]
        d.sort()
        dur = [x[1] for x in d][1:] if len(d) > 1 else [x[1] for x in d]
        summary["_kernel_trace"] = {"launches": len(d), "avg_ns_without_first_launch": sum(dur) / len(dur), "min_ns": min(dur),
                                    "max_ns": max(dur), "first_launch_ns": d[0][1]}
        print("kernel trace without the warm-up launch: avg %.0f ns over %d launches" % (sum(dur) / len(dur), len(dur)))
    with open(os.path.join(dst, tag + "_bench_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)

    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        fk, wk = summary["FETCH_SIZE"]["mean_per_launch"], summary["WRITE_SIZE"]["mean_per_launch"]
        batch = int(os.environ.get("NRLDPC_PROFILE_BATCH", "4096"))
        hbm = int(round(2 * fk * 1024 + wk * 1024))
        out = {
            "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), "
                      "tools/profile_gpu.sh %s, bench.py batch %d; tools/summarise_profile.py" % (tag, batch),
            "kernel": kernel,
            "FETCH_SIZE_KB": fk,
            "WRITE_SIZE_KB": wk,
            "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM) -> x2; "
                          "WRITE_SIZE is exact",
            "hbm_bytes_per_launch": hbm,
            "hbm_bytes_per_codeword": hbm / batch,
        }
        out["nrldpc_build_id"], out["nrldpc_kernel_id"] = ids.get("nrldpc_build_id"), ids.get("nrldpc_kernel_id")
        with open(os.path.join(dst, tag + "_traffic_bytes_per_launch.json"), "w") as f:
            json.dump(out, f, indent=1)
        print("HBM bytes per launch:", hbm, "per codeword:", hbm / batch)


if __name__ == "__main__":
    main()
