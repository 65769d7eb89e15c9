#!/usr/bin/env python3
"""Condenses gpurun_out/profile_<tag>/ into the files kept under profiles/.

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the bench command
  profiles/<tag>_pmc.txt            per-kernel PMC totals and derived ratios
  profiles/<tag>_traffic_<key>.json HBM bytes per launch of the dominant kernel of the profiled workload (k_extend
                                    or k_trace), read by bench.py for roofline.traffic / roofline.hbm; <key> names
                                    the workload as bench.py's workload_key does (e.g. cornell_1920x1080_8b)

  usage: tools/profile_digest.py <tag> [<key>]      (tag may carry a suffix: r03_outside -> profiles/r03_outside_*)

HBM bytes follow the gfx950 rule for rocprofv3: FETCH_SIZE and WRITE_SIZE are in KiB, and FETCH_SIZE counts
128-byte read requests as 64 bytes, so reads are doubled: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import glob
import io
import json
import os
import shutil
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_summary  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
key = sys.argv[2] if len(sys.argv) > 2 else "cornell_1920x1080_8b"
round_tag = tag.split("_")[0]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"profile_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench_traced.json")):
    shutil.copy(os.path.join(src, "bench_traced.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))

buf = io.StringIO()
with redirect_stdout(buf):
    pmc_summary.main(src)
open(os.path.join(dst, f"{tag}_pmc.txt"), "w").write(buf.getvalue())


def per_dispatch(counter, kernel_substr):
    tot, n = 0.0, 0
    for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                tot += float(row["Counter_Value"])
                n += 1
    return (tot / n if n else None), n


# HBM bytes per launch of the kernel that dominates the profiled run (k_trace: persistent pipeline,
# k_extend: wavefront pipeline); bench.py reads the file of the kernel it ran for roofline.traffic
for kernel in ("k_extend", "k_trace"):
    fname = f"{round_tag}_traffic_{key}.json"
    fetch, nf = per_dispatch("FETCH_SIZE", kernel)
    write, nw = per_dispatch("WRITE_SIZE", kernel)
    if fetch is None or write is None:
        continue
    segs, pipe = None, None
    try:
        b = json.load(open(os.path.join(src, "fetch.json")))
        segs = (b.get("roofline") or {}).get("segments_per_launch") or b["kernel_segments"] / b["extend_launches_rank0"]
        pipe = {"wavefront": 0, "persistent": 2, "hybrid": 4}.get(b["config"]["pipeline"].split()[0])
    except Exception:
        pass
    # how busy the units that actually bound the kernel were: texture addresser / data return (the L1
    # gather path, counters summed over 256 CUs) and vector issue (4 cycles per wave64 instruction on each
    # of 1024 SIMDs), against the kernel's cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    busy = {}
    cyc, _ = per_dispatch("GRBM_GUI_ACTIVE", kernel)
    if cyc:
        cyc /= 8.0
        for name, counter, scale in (("ta_busy_frac", "TA_TA_BUSY_sum", 1.0 / 256), ("td_busy_frac", "TD_TD_BUSY_sum", 1.0 / 256),
                                     ("valu_issue_frac", "SQ_INSTS_VALU", 4.0 / 1024)):
            v, _ = per_dispatch(counter, kernel)
            if v is not None:
                busy[name] = v * scale / cyc
    if pipe is not None and (kernel == "k_extend") != (pipe == 0):
        continue  # not the kernel whose launches the run timed
    traffic = {
        "kernel": kernel,
        "workload": key,
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline"
                   + " --no-secondary [workload arguments]",
        "fetch_size_kib_per_launch": fetch, "write_size_kib_per_launch": write, "launches": nf,
        "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
        "segments_per_launch": segs,
        "pipeline": pipe,
        "units": busy,
        "rule": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-B reads as 64 B)",
    }
    json.dump(traffic, open(os.path.join(dst, fname), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
    break
print(buf.getvalue()[:3000])
