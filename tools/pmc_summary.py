#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters (CSV output): one line per (counter, kernel) with the mean
value per dispatch and the number of dispatches.  Kernel names are shortened to their function name.

  python tools/pmc_summary.py <dir-with-*_counter_collection.csv> [...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)              # drop the argument list
    name = re.sub(r"<.*$", "", name)
    name = name.split("::")[-1].strip()
    name = re.sub(r"^void ", "", name)
    return name.replace(" [clone .kd]", "").replace(".kd", "")


def main():
    acc = defaultdict(lambda: defaultdict(float))   # (counter, kernel) -> dispatch id -> value
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row.get("Kernel_Name") or row.get("Kernel Name") or "?")
                    c = row.get("Counter_Name") or row.get("Counter Name")
                    v = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
                    disp = row.get("Dispatch_Id") or row.get("Dispatch Id") or row.get("Correlation_Id")
                    acc[(c, k)][disp] += v          # sum over dimensions (XCD / instance) of one dispatch
    for (c, k) in sorted(acc):
        vals = list(acc[(c, k)].values())
        print(f"{c:12s} {k:28s} mean/dispatch {sum(vals) / len(vals):14.1f}   dispatches {len(vals)}")


if __name__ == "__main__":
    main()
