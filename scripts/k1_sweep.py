"""K1 launch-shape sweep at a large reference: one index, several (seeds per wave, max waves) settings in child processes is not
possible (the settings are read once per process), so this script runs ONE setting; the caller loops.
usage: k1_sweep.py genome_bp n_reads read_len"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import floxer_amd as F
from floxer_amd import simulate as S
G, NR, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pool, genome = S.make_genome_fast(G, 1, seed=S.DEFAULT_SEED)
idx = F.fmindex(genome, device=0)
os.environ["FLX_LANES"] = "1"
ctx = F.context(idx)
al = F.aligner(ctx, F.params(error_probability=0.08, interval_optimization=True))
reads, _ = S.make_reads_fast(pool, [G], NR, L, 0.08, seed=5)
rr = F.resident_reads(ctx, reads)
al.align_reads(rr)
ctx.enable_kernel_timing(True); ctx.reset_kernel_stats()
al.align_reads(rr)
st = ctx.kernel_stats()
v = st["fm_search"]
print(f"SPW={os.environ.get('FLX_FM_SEEDS_PER_WAVE','-')} MAXW={os.environ.get('FLX_FM_MAX_WAVES','-')} fm_search launches {v['launches']} ms {v['device_ms']:.2f} ext {v['work_units']} Gext/s {v['work_units']/v['device_ms']/1e6:.2f} frac {v['algorithmic_bytes']/v['device_ms']/1e6/8000:.3f} | select ms {st['fm_select']['device_ms']:.2f}", flush=True)
