"""Every dispatch of kernels whose name contains argv[2], in start order: duration in ms."""
import csv, sys
rows = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"][:40]) for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
for s, d, n in sorted(rows): print(f"{d:8.3f} ms  {n}")
