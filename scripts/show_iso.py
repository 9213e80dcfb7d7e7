import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], d.get("value"), {k: round(v["device_ms"] / max(1, v["launches"]), 2) if k == "fm_search" else round(v["device_ms"], 2) for k, v in d["kernels_isolated"].items()})
