#!/usr/bin/env python3
"""Mean counter value per launch and kernel from rocprofv3 `--pmc ... --output-format csv` runs.

    python tools/pmc_summary.py out.json run1_counter_collection.csv [run2_counter_collection.csv ...]
"""
import collections
import csv
import json
import sys


def main():
    out, files = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])       # one row per XCD / dimension: sum them
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (disp, ctr), v in per_dispatch.items():
            agg[names[disp]][ctr].append(v)
    res = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in ctrs.items()}
           for k, ctrs in agg.items()}
    json.dump(res, open(out, "w"), indent=1)
    for k, ctrs in res.items():
        print(k[:100], {c: round(v["mean_per_launch"], 1) for c, v in ctrs.items()})


if __name__ == "__main__":
    main()
