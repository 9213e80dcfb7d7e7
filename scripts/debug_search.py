import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import floxer_amd as F
import oracle_lib as O
import test_gpu_parity as T
rng = np.random.default_rng(3)
refs = [rng.integers(1, 5, size=60000, dtype=np.uint8), rng.integers(1, 5, size=20000, dtype=np.uint8)]
idx = F.fmindex(refs); ctx = F.context(idx); oidx = O.Index(refs)
rng = np.random.default_rng(21)
pool, seeds = T._make_seeds(rng, refs, 400)
sr = F.searcher(ctx)
got = sr.search_groups(pool, seeds, max_hits=10**6)
bad = {}
tot = {}
for i, (off, ln, k, _) in enumerate(seeds):
    exp, _ = oidx.search_groups(pool[off:off + ln], k, n=10**6)
    mine = got[got[:, 0] == i][:, 1:]
    tot[k] = tot.get(k, 0) + 1
    if mine.tolist() != exp.tolist():
        bad.setdefault(k, []).append((i, ln, len(mine), len(exp)))
print("per k totals", tot)
for k, v in bad.items(): print("k", k, "bad", len(v), v[:6])
i = bad[min(bad)][0][0] if bad else None
if i is not None:
    off, ln, k, _ = seeds[i]
    exp, _ = oidx.search_groups(pool[off:off + ln], k, n=10**6)
    mine = got[got[:, 0] == i][:, 1:]
    print("seed", i, ln, k, "exp", exp.tolist()[:10], "got", mine.tolist()[:10])
