"""Per-kernel durations (ms) from a rocprofv3 --kernel-trace csv."""
import csv, collections, sys
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"][:56]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for n, v in d.items():
    v.sort()
    print(f"{n:58s} n={len(v):4d} med={v[len(v)//2]:8.3f} max={v[-1]:8.3f} sum={sum(v):8.1f}")
