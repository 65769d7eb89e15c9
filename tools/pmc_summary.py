#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output (one row per dispatch and counter) per kernel."""
import csv
import glob
import sys
from collections import defaultdict


def main(root):
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
            k = row["Kernel_Name"].split("(lpt")[0].split("::")[-1] if "k_" in row["Kernel_Name"] else k
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]] += 1
    for k in sorted(agg):
        print(f"== {k}")
        for c in sorted(agg[k]):
            n = calls[k][c]
            print(f"   {c:28s} total {agg[k][c]:18.0f}  per-dispatch {agg[k][c] / n:16.1f}  ({n} dispatches)")
        a = agg[k]
        if "SQ_THREAD_CYCLES_VALU" in a and a.get("SQ_ACTIVE_INST_VALU"):
            print(f"   -> VALU lane utilisation  {100 * a['SQ_THREAD_CYCLES_VALU'] / (64 * a['SQ_ACTIVE_INST_VALU']):.1f} %")
        if a.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
                if c in a:
                    print(f"   -> {c}/SQ_WAVE_CYCLES {100 * a[c] / a['SQ_WAVE_CYCLES']:.1f} %")
        if "TCC_HIT_sum" in a:
            print(f"   -> L2 hit rate {100 * a['TCC_HIT_sum'] / max(1, a['TCC_HIT_sum'] + a['TCC_MISS_sum']):.1f} %")


if __name__ == "__main__":
    main(sys.argv[1])
