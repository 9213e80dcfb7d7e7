"""K1 at a larger reference: one lane, kernel timing on. usage: search_scale.py genome_bp n_reads read_len"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import floxer_amd as F
from floxer_amd import simulate as S
G, NR, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
genome = S.make_genome(G, 1, seed=S.DEFAULT_SEED)
t = time.time(); idx = F.fmindex(genome, device=0); print("index build", round(time.time() - t, 1), "s", flush=True)
os.environ["FLX_LANES"] = "1"
ctx = F.context(idx)
al = F.aligner(ctx, F.params(error_probability=0.08, interval_optimization=True))
reads, _, _ = S.make_reads(genome, NR, L, 0.08, seed=5)
rr = F.resident_reads(ctx, reads)
al.align_reads(rr)
ctx.enable_kernel_timing(True); ctx.reset_kernel_stats()
t = time.time(); al.align_reads(rr); dt = time.time() - t
print(f"one lane: {NR / dt:.0f} reads/s")
for k, v in ctx.kernel_stats().items():
    print(" ", k, "launches", v["launches"], "ms", round(v["device_ms"], 2), "alg GB/s", round(v["algorithmic_bytes"] / 1e6 / max(v["device_ms"], 1e-9), 1))
