#!/usr/bin/env python3
"""Per-kernel averages of the two SQ counter passes of profiles/collect.sh (instruction classes; cycles) ->
profiles/r02_pmc_sq_summary.json.  The raw per-dispatch csv files (6 MB each) are not kept.

    python3 profiles/sq_summary.py <pmc_insts counter_collection.csv> <pmc_cycles counter_collection.csv> profiles/r02_pmc_sq_summary.json
"""
import csv
import json
import sys
from collections import defaultdict

SKIP_FIRST = 3


def reduce(path, out):
    seen = defaultdict(int)                  # (kernel, counter) -> dispatches seen
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if "gsr::" not in name:
            continue
        key = (name, r["Counter_Name"])
        seen[key] += 1
        if seen[key] <= SKIP_FIRST:
            continue
        a = acc[key]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for (name, ctr), (s, n) in acc.items():
        k = out.setdefault(name, {"launches_sampled": n})
        k[ctr] = round(s / n, 1)


def main():
    out = {}
    reduce(sys.argv[1], out)
    reduce(sys.argv[2], out)
    json.dump({"source": "rocprofv3 --pmc <8 SQ counters per pass> -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline "
                         "--no-secondary (two passes: profiles/collect.sh); first 3 launches of every kernel dropped",
               "kernels": dict(sorted(out.items()))}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
