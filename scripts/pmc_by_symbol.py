#!/usr/bin/env python3
"""Per kernel symbol: dispatches and the sum of one counter from a rocprofv3 --pmc counter_collection.csv.
usage: pmc_by_symbol.py counter_collection.csv COUNTER [substring ...]   (KiB counters are printed in GB as well)"""
import csv
import collections
import sys

path, counter, subs = sys.argv[1], sys.argv[2], sys.argv[3:]
tot, n = collections.defaultdict(float), collections.Counter()
for row in csv.DictReader(open(path)):
    if row["Counter_Name"] != counter:
        continue
    name = row["Kernel_Name"].split("(")[0]
    if subs and not any(s in name for s in subs):
        continue
    tot[name] += float(row["Counter_Value"])
    n[name] += 1
for name in sorted(tot, key=lambda k: -tot[k]):
    print(f"{name[:90]:90s} dispatches {n[name]:5d}  {counter} {tot[name]:16.1f}  (x 1 KiB = {tot[name] * 1024 / 1e9:9.3f} GB)")
