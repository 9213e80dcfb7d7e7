#!/usr/bin/env python3
"""Concurrency of the kernels in a rocprofv3 kernel trace (CSV): for the time of the last n search launches, how long k launches of
the search kernel ran at once, how long nothing ran, and every kernel family's busy time (union of its launches).
usage: trace_concurrency.py kernel_trace.csv [n_search_launches]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 64          # window = from the start of the n-th last search launch to the end of the last
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
fm = sorted((s, e) for s, e, k in ev if "fm_search_filter" in k or "fm_search_kernel" in k)
t0 = fm[-min(n_last, len(fm))][0]
t_end = max(e for _, e in fm)
ev = [(s, min(e, t_end), k) for s, e, k in ev if s < t_end]
ev = [(max(s, t0), e, k) for s, e, k in ev if e > t0]
span = t_end - t0
def family(k):
    for f in ("fm_search_filter", "fm_search_text", "fm_search", "ed_trace_block", "ed_exists_lane", "ed_band_kernel", "ed_exists_block", "ed_align_kernel", "ed_traceback", "seed_select", "seed_rows", "hit_scatter",
              "vr2_request", "vr2_apply", "vr_", "lastrow", "rocprim", "hipcub", "fillBuffer", "copyBuffer", "peq_build", "pack_pool", "seed_compact"):
        if f in k: return f
    return "other"
def union(iv):
    iv.sort(); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot
fam = collections.defaultdict(list)
for s, e, k in ev: fam[family(k)].append((s, e))
print(f"window {span / 1e6:.1f} ms, {len(ev)} launches")
print(f"any kernel running: {union([(s, e) for s, e, _ in ev]) / span:.3f} of the time")
for f, iv in sorted(fam.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    print(f"  {f:18s} launches {len(iv):6d}  sum {sum(e - s for s, e in iv) / 1e6:9.1f} ms  avg {sum(e - s for s, e in iv) / 1e3 / len(iv):9.1f} us  union {union(list(iv)) / span:.3f} of the time")
# distribution of concurrent search launches
pts = []
for s, e in fam["fm_search_filter"] + fam["fm_search"]: pts += [(s, 1), (e, -1)]
pts.sort(); cur = 0; last = t0; hist = collections.Counter()
for t, d in pts:
    hist[cur] += t - last; last = t; cur += d
hist[cur] += t_end - last
print("concurrent fm_search launches: " + "  ".join(f"{k}: {v / span:.3f}" for k, v in sorted(hist.items())))

# ---- what the launches ask of the chip while they run: wave slots, registers and LDS, as shares of the chip over the window.
# A launch is taken to keep min(its waves, what the chip can hold of it) resident for its whole duration (an upper bound: tails are
# emptier), 256 CUs x 4 SIMDs, 8 wave slots and 512 VGPRs per SIMD lane, 160 KB LDS per CU.
if rows and "VGPR_Count" in rows[0] and "LDS_Block_Size" in rows[0]:
    CUS, SIMDS = 256, 1024
    demand = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        s, e = max(s, t0), min(e, t_end)
        if e <= s: continue
        wg = max(1, int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 64))
        waves_per_wg = (wg + 63) // 64
        n_wg = max(1, int(r.get("Grid_Size") or r.get("Grid_Size_X") or wg) // wg)
        vg = max(8, int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count", 0) or 0))
        lds = int(r["LDS_Block_Size"])
        per_simd = max(1, min(8, 512 // vg))
        cap_waves = SIMDS * per_simd
        if lds: cap_waves = min(cap_waves, CUS * max(1, (160 * 1024) // lds) * waves_per_wg)
        w = min(n_wg * waves_per_wg, cap_waves)
        d = demand[family(r["Kernel_Name"])]
        d[0] += w * (e - s) / (SIMDS * 8 * span)
        d[1] += w * vg * (e - s) / (SIMDS * 512 * span)
        d[2] += (w / waves_per_wg) * lds * (e - s) / (CUS * 160 * 1024 * span)
    print("resident demand (share of the chip over the window; > 1 in total = launches wait for room):  wave slots   VGPRs   LDS")
    tot = [0.0, 0.0, 0.0]
    for f, d in sorted(demand.items(), key=lambda kv: -max(kv[1])):
        print(f"  {f:18s} {d[0]:10.3f} {d[1]:8.3f} {d[2]:8.3f}")
        for i in range(3): tot[i] += d[i]
    print(f"  {'total':18s} {tot[0]:10.3f} {tot[1]:8.3f} {tot[2]:8.3f}")
