"""Timeline of GPU occupancy from a rocprofv3 --kernel-trace csv: busy fraction (union of kernel intervals) per time bin,
with the kernel family that holds most of the bin. Usage: gpu_busy.py kernel_trace.csv [bin_ms] [last_ms] [skip_ms]"""
import csv, sys, collections
path = sys.argv[1]; bin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0; last_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 400.0
skip_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0      # ignore this much at the end of the trace
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        fam = "search" if "fm_search" in n else "locate" if "fm_locate" in n else "exists/trace" if "ed_band" in n or "ed_align" in n else "traceback" if "traceback" in n else "peq" if "peq" in n else "other"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam))
rows.sort()
t_end = max(r[1] for r in rows) - int(skip_ms * 1e6); t0 = t_end - int(last_ms * 1e6)
rows = [(s, min(e, t_end), f) for s, e, f in rows if e > t0 and s < t_end]
nb = int(last_ms / bin_ms)
busy = [0.0] * nb; fam_t = [collections.Counter() for _ in range(nb)]
# union of intervals
cur_s = cur_e = None; merged = []
for s, e, _ in rows:
    s = max(s, t0)
    if cur_e is None or s > cur_e:
        if cur_e is not None: merged.append((cur_s, cur_e))
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
merged.append((cur_s, cur_e))
def spread(s, e, fn):
    b = int((s - t0) / (bin_ms * 1e6))
    while b < nb and s < e:
        be = t0 + int((b + 1) * bin_ms * 1e6); x = min(e, be) - s
        fn(b, x); s = be; b += 1
for s, e in merged: spread(s, e, lambda b, x: busy.__setitem__(b, busy[b] + x))
for s, e, fam in rows: spread(max(s, t0), e, lambda b, x, fam=fam: fam_t[b].update({fam: x}))
tot = sum(busy) / 1e6
print(f"busy {tot:.1f} ms of the last {last_ms:.0f} ms")
for b in range(nb):
    top = ", ".join(f"{k}:{v / 1e6:.1f}" for k, v in fam_t[b].most_common(3))
    print(f"{b * bin_ms:6.0f} ms  busy {busy[b] / (bin_ms * 1e4):5.1f}%  {top}")
