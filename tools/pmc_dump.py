#!/usr/bin/env python
"""Per-kernel means of every counter in a rocprofv3 counter_collection.csv:  python tools/pmc_dump.py file.csv [substring]"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
want = sys.argv[2] if len(sys.argv) > 2 else ""
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
    if want in name:
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, counters in acc.items():
    print(name[:70])
    for c, v in sorted(counters.items()):
        print("   %-34s %14.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
