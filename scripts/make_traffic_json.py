"""Builds profiles/<round>_pmc_traffic_<kernel>.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the
MI355X guide prescribes) of `bench.py --isolated-only` and that run's JSON line. gfx950 corrections (MI355X_MICROARCH.md, HBM):
counters are in KiB; FETCH_SIZE reports half of the bytes of wide coalesced reads (doubled here); WRITE_SIZE is exact for
16-B-per-lane stores.
Usage: make_traffic_json.py fetch.csv write.csv isolated.json out_dir prefix"""
import csv, json, os, sys
fetch_csv, write_csv, iso_json, out_dir, prefix = sys.argv[1:6]
iso = json.load(open(iso_json))
# accounting name -> substrings of the kernel symbols it covers (fm_search = the filter walk + the text walk)
KERNELS = {"fm_search": ("fm_search_filter_kernel", "fm_search_text_kernel"), "ed_align_trace": ("ed_trace_block_kernel",), "ed_align_exists": ("ed_exists_block_kernel", "ed_exists_lane_kernel"),
           "ed_traceback": ("traceback",)}
# bytes per FETCH_SIZE unit / 1024: 2 for wide coalesced reads (128-B requests tallied at 64 B, MI355X guide); 1 for fm_search, whose reads are
# random 32-byte blocks fetched as 64-byte requests and tallied exactly (calibration: scripts/micro/gather_cost.hip ... calib,
# profiles/r02_gather_calib.txt)
FETCH_FACTOR = {"fm_search": 1.0}
def total(path, subs, counter):
    t, n = 0.0, 0
    for row in csv.DictReader(open(path)):
        if any(sub in row["Kernel_Name"] for sub in subs) and row["Counter_Name"] == counter:
            t += float(row["Counter_Value"]); n += 1
    return t, n
passes = 2           # --isolated-only runs the batch twice (warm + measured)
for name, sub in KERNELS.items():
    st = iso["kernels_isolated"].get(name)
    if not st:
        continue
    alg = st["GBps"] * 1e6 * st["device_ms"]           # algorithmic bytes of the measured pass
    f, nf = total(fetch_csv, sub, "FETCH_SIZE")
    w, nw = total(write_csv, sub, "WRITE_SIZE")
    ff = FETCH_FACTOR.get(name, 2.0)
    hbm = (ff * f + w) * 1024.0 / passes
    units = st["work_units"]                              # work units of the measured pass (fm_search: rank pairs; DP kernels: word-steps)
    out = {"kernel": name, "kernel_symbol": list(sub), "genome": iso["config"].get("genome"), "reads_per_step": iso["config"]["reads_per_step_per_gpu"],
           "read_length": int(round(iso["config"]["mean_read_length"], -2)),
           "dispatches_seen": {"fetch_pass": nf, "write_pass": nw}, "FETCH_SIZE_KiB_total": f, "WRITE_SIZE_KiB_total": w,
           "corrections": f"bytes = ({ff:g}*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per request; wide coalesced reads are 128-B requests)",
           "hbm_bytes_per_pass": hbm, "algorithmic_bytes_per_pass": alg, "traffic_over_algorithmic": hbm / alg if alg else None,
           "work_units_per_pass": units, "traffic_bytes_per_work_unit": hbm / units if units else None}
    json.dump(out, open(os.path.join(out_dir, f"{prefix}_pmc_traffic_{name}.json"), "w"), indent=1)
    print(name, "HBM bytes per pass", int(hbm), "per work unit", round(hbm / units, 3) if units else None)
