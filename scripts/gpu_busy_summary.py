"""GPU occupancy of a time window of a rocprofv3 --kernel-trace csv: union-busy fraction, mean number of kernels in flight and
the per-family sums. Usage: gpu_busy_summary.py kernel_trace.csv window_ms skip_ms   (window ends skip_ms before the last kernel)"""
import csv, sys, collections
path, win_ms, skip_ms = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
rows = []
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    fam = "fm_search" if "fm_search" in n else "ed_align_trace" if "true>" in n else "ed_align_exists" if "false>" in n else "ed_traceback" if "traceback" in n else "other"
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam))
t_end = max(r[1] for r in rows) - int(skip_ms * 1e6); t0 = t_end - int(win_ms * 1e6)
rows = sorted((max(s, t0), min(e, t_end), f) for s, e, f in rows if e > t0 and s < t_end)
busy = 0; cur_s = cur_e = None
for s, e, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
fam = collections.Counter()
for s, e, f in rows: fam[f] += e - s
tot = sum(fam.values())
print(f"window {win_ms:.0f} ms: union busy {busy / 1e6:.1f} ms ({100 * busy / (win_ms * 1e6):.1f}%), kernels in flight while busy {tot / busy:.2f}")
for f, v in fam.most_common(): print(f"  {f:18s} {v / 1e6:8.1f} ms summed")
