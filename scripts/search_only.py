import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import floxer_amd as F
from floxer_amd import simulate as S
genome = S.make_genome(4_600_000, 1, seed=S.DEFAULT_SEED)
idx = F.fmindex(genome)
os.environ["FLX_LANES"] = "1"
ctx = F.context(idx)
al = F.aligner(ctx, F.params(error_probability=0.08, interval_optimization=True))
reads, _, _ = S.make_reads(genome, 4096, 5000, 0.08, seed=5)
rr = F.resident_reads(ctx, reads)
al.align_reads(rr)
al.align_reads(rr)
