#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel."""
import collections
import csv
import json
import sys


def summarise(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


if __name__ == "__main__":
    out = {}
    for p in sys.argv[1:]:
        for k, d in summarise(p).items():
            if k.startswith("m2s::"):
                out.setdefault(k, {}).update(d)
    print(json.dumps(out, indent=1))
