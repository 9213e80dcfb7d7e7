"""Builds profiles/<round>_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the
MI355X guide prescribes) of `bench.py --isolated-only`. gfx950 corrections (MI355X_MICROARCH.md, HBM): counters are in KiB;
FETCH_SIZE reports half of the bytes of wide coalesced reads (doubled here); WRITE_SIZE is exact for 16-B-per-lane stores."""
import csv, json, sys
fetch_csv, write_csv, kernel_substr, out_path, reads_per_step, algorithmic_bytes_per_pass, kernel_name = sys.argv[1:8]
def total(path, counter):
    t, n = 0.0, 0
    for row in csv.DictReader(open(path)):
        if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == counter:
            t += float(row["Counter_Value"]); n += 1
    return t, n
f, nf = total(fetch_csv, "FETCH_SIZE")
w, nw = total(write_csv, "WRITE_SIZE")
passes = 2           # --isolated-only runs the batch twice (warm + measured)
hbm = (2.0 * f + w) * 1024.0 / passes
alg = float(algorithmic_bytes_per_pass)
json.dump({"kernel": kernel_name, "kernel_symbol": kernel_substr, "reads_per_step": int(reads_per_step),
           "dispatches_seen": {"fetch_pass": nf, "write_pass": nw}, "FETCH_SIZE_KiB_total": f, "WRITE_SIZE_KiB_total": w,
           "corrections": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
           "hbm_bytes_per_pass": hbm, "algorithmic_bytes_per_pass": alg, "traffic_over_algorithmic": hbm / alg},
          open(out_path, "w"), indent=1)
print(open(out_path).read())
