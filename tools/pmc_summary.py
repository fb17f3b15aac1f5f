"""Summarise rocprofv3 --pmc counter_collection CSVs: mean per-dispatch value of every
counter for kernels whose name contains a substring.  Usage:
    python tools/pmc_summary.py <dir-or-csv> [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else "adc_scan_kernel"
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(float))
    for f in files:
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for name in sorted(acc):
        v = list(acc[name].values())
        # full-batch dispatches only: the bench also launches a few small ones (checker rows)
        big = [x for x in v if x >= 0.5 * max(v)] if max(v) > 0 else v
        print("%-28s dispatches=%d mean=%.4g  (full-batch dispatches=%d mean=%.4g)"
              % (name, len(v), sum(v) / len(v), len(big), sum(big) / len(big)))


if __name__ == "__main__":
    main()
