import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import floxer_amd as F
from floxer_amd import simulate as S
genome = S.make_genome(4_600_000, 1, seed=S.DEFAULT_SEED)
idx = F.fmindex(genome)
os.environ["FLX_LANES"] = "1"
ctx = F.context(idx)
al = F.aligner(ctx, F.params(error_probability=0.08))
reads, _, _ = S.make_reads(genome, int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 5000, 0.08, seed=5)
rr = F.resident_reads(ctx, reads)
al.align_reads(rr)
for it in range(3):
    ctx.enable_kernel_timing(True); ctx.reset_kernel_stats()
    al.align_reads(rr)
    st = ctx.kernel_stats()
    t = st["ed_align_trace"]
    print("trace: launches", t["launches"], "ms", round(t["device_ms"], 2), "GB/s", round(t["algorithmic_bytes"] / 1e6 / t["device_ms"], 1),
          "Gws/s", round(t["work_units"] / 1e6 / t["device_ms"], 1), "| exists ms", round(st["ed_align_exists"]["device_ms"], 2), flush=True)
