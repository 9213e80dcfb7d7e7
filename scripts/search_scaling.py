import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import floxer_amd as F
from floxer_amd import simulate as S
genome = S.make_genome(4_600_000, 1, seed=S.DEFAULT_SEED)
idx = F.fmindex(genome)
os.environ["FLX_LANES"] = "1"
ctx = F.context(idx)
p = F.params(error_probability=0.08, interval_optimization=True)
al = F.aligner(ctx, p)
for n in (256, 1024, 4096):
    reads, _, _ = S.make_reads(genome, n, 5000, 0.08, seed=5)
    rr = F.resident_reads(ctx, reads)
    al.align_reads(rr)
    ctx.enable_kernel_timing(True); ctx.reset_kernel_stats()
    al.align_reads(rr)
    st = ctx.kernel_stats()
    ctx.enable_kernel_timing(False)
    print(n, {k: round(v["device_ms"], 3) for k, v in st.items()}, flush=True)
    rr.close()
