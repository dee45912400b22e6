"""Sums rocprofv3 --pmc counter_collection CSVs per kernel and counter.
    python3 tools/pmc_summary.py DIR [DIR ...] > summary.json
Every DIR is the -d directory of one `rocprofv3 --kernel-trace --pmc <counters> --output-format csv` pass.  Output: per
kernel (template arguments kept, rocPRIM kernels folded into one name) the number of dispatches and, per counter, the
total over all dispatches and counter instances -- divide by the dispatches for a per-launch figure."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    if "rocprim" in name:
        return "rocprim (sorts, scans)"
    name = name.replace("em2::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    out = defaultdict(lambda: {"dispatches": set(), "counters": defaultdict(float)})
    for directory in sys.argv[1:]:
        for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    kernel = short(row["Kernel_Name"])
                    out[kernel]["dispatches"].add((path, row["Dispatch_Id"]))
                    out[kernel]["counters"][row["Counter_Name"]] += float(row["Counter_Value"])
    result = {}
    for kernel, data in sorted(out.items()):
        passes = defaultdict(int)
        for path, _ in data["dispatches"]:
            passes[path] += 1
        result[kernel] = {"dispatches_per_pass": sorted(set(passes.values())), "totals": dict(data["counters"])}
    json.dump(result, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
