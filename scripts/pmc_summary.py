"""Summarise a rocprofv3 --pmc csv (counter_collection.csv): per kernel name, mean counter value per dispatch."""
import csv, sys, collections
path = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(path) as f:
    for row in csv.DictReader(f):
        name = row.get("Kernel_Name", "")[:60]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, ctrs in acc.items():
    n = max(len(v) for v in ctrs.values())
    print(f"{name}  dispatches={n}")
    for c, v in sorted(ctrs.items()):
        print(f"    {c:28s} mean={sum(v)/len(v):.4g} total={sum(v):.4g}")
