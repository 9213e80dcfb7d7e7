import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
import floxer_amd as F
from floxer_amd import simulate as S, capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
genome = S.make_genome(4_600_000, 1, seed=S.DEFAULT_SEED)
reads, _, _ = S.make_reads(genome, n, 5000, 0.08, seed=5)
idx = F.fmindex(genome); ctx = F.context(idx)
p = F.params(error_probability=0.08); al = F.aligner(ctx, p)
rr = F.resident_reads(ctx, reads)
al.align_reads(rr)
for it in range(3):
    t0 = time.perf_counter()
    run = C.c_void_p()
    capi.check(capi.lib().flx_align_reads_resident(ctx.h, C.byref(p), rr.h, C.byref(run)))
    t1 = time.perf_counter()
    res = F._collect_run(run, rr.n)
    t2 = time.perf_counter()
    print(f"C call {1e3*(t1-t0):.1f} ms, python collect {1e3*(t2-t1):.1f} ms, records {len(res.rows)}, cigar words {len(res.cigars)}")
