"""Cost of every launch shape (words per lane W, lanes per job R) of the existence kernel at the node sizes of a 10-kb read's PEX
tree: N identical-size jobs (window = node + 2k + 1 columns, query = the window's middle with ~k/2 edits), one shape forced per run
(FLX_FORCE_SHAPE), kernel time from the context's kernel statistics. usage: shape_cost.py [total_rows]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import floxer_amd as F
from floxer_amd import simulate as S

TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24          # rows x jobs per size
os.environ["FLX_LANES"] = "1"
os.environ["FLX_ALIGN_FEW_WAVES"] = "0"
rng = np.random.default_rng(7)
genome = S.make_genome(4_000_000, 1, seed=3)
idx = F.fmindex(genome, device=0)
ctx = F.context(idx)
ctx.enable_kernel_timing(True)
text = np.asarray(genome[0]) if isinstance(genome, (list, tuple)) else np.asarray(genome)
WORDS = [1, 2, 3, 4, 5, 6, 8, 13, 25]
print("m k n jobs | shape: ms (ns per job-row)")
for m in (75, 150, 300, 600, 1250, 2500, 5000):
    k = int(np.ceil(0.08 * m))
    n = m + 2 * k + 1
    jobs_n = max(64, TOTAL // m)
    starts = rng.integers(0, len(text) - n - 1, size=jobs_n)
    qpool = np.empty(jobs_n * m, dtype=np.uint8)
    for i, st in enumerate(starts):
        q = text[st + k: st + k + m].copy()
        pos = rng.integers(0, m, size=k // 2)
        q[pos] = (q[pos] % 4) + 1
        qpool[i * m:(i + 1) * m] = q
    jobs = [(int(st), n, i * m, m, k, F.MODE_EXISTS) for i, st in enumerate(starts)]
    nw, width = (m + 63) // 64, n - m + 2 * k
    rows = []
    for w in WORDS:
        for r in (1, 2, 4, 8, 16, 32, 64):
            if not ((nw + w - 1) // w <= r or 64 * w * (r - 1) + r + 1 > width):
                continue
            if w * r > 4 * max(nw, 1) and r > 1:
                continue                                             # far more lanes x words than the job has
            os.environ["FLX_FORCE_SHAPE"] = f"{w},{r}"
            F.align_batch(ctx, qpool, jobs)                          # warm
            ctx.reset_kernel_stats()
            res = F.align_batch(ctx, qpool, jobs)
            st = ctx.kernel_stats()
            ms = sum(v["device_ms"] for kname, v in st.items() if kname.startswith("ed_align"))
            found = sum(1 for x in res if x is not None)
            rows.append((ms, w, r, found))
    os.environ.pop("FLX_FORCE_SHAPE", None)
    F.align_batch(ctx, qpool, jobs)
    ctx.reset_kernel_stats()
    F.align_batch(ctx, qpool, jobs)
    base = sum(v["device_ms"] for kname, v in ctx.kernel_stats().items() if kname.startswith("ed_align"))
    rows.sort()
    print(f"{m} {k} {n} {jobs_n} | default {base:.2f} ms | " + "  ".join(f"({w},{r}): {ms:.2f}" for ms, w, r, _ in rows[:8]) +
          f" | found {rows[0][3]}/{jobs_n}", flush=True)
