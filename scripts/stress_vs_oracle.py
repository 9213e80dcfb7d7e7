"""One-off stress: many reads of several lengths and error rates, several references, default flags and -I, every record
compared with the oracle (the checker). usage: stress_vs_oracle.py [n_reads]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import floxer_amd as F
import oracle_lib as O
from floxer_amd import simulate as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
genome = S.make_genome(3_000_000, 5, seed=101)
# a few repeats: copy segments around so that seeds have many hits
rng = np.random.default_rng(7)
for g in genome:
    for _ in range(6):
        a, b, ln = rng.integers(0, len(g) - 5000), rng.integers(0, len(g) - 5000), int(rng.integers(500, 4000))
        g[b:b + ln] = g[a:a + ln]
idx = F.fmindex(genome); ctx = F.context(idx); oidx = O.Index(genome)
bad = 0
for (length, rate, seed) in [(1500, 0.05, 1), (3000, 0.08, 2), (800, 0.10, 3), (5000, 0.03, 4)]:
    reads, _, _ = S.make_reads(genome, n, length, rate, seed=200 + seed)
    for kw, okw in [(dict(), dict()), (dict(interval_optimization=True), dict(interval_opt=True))]:
        t = time.time(); got = F.aligner(ctx, F.params(error_probability=rate, **kw)).align_reads(reads); tg = time.time() - t
        t = time.time(); exp = oidx.run(reads, O.params(error_probability=rate, **okw), threads=16); to = time.time() - t
        same = got.records() == exp.records() and got.skipped.tolist() == exp.skipped.tolist()
        print(f"len {length} rate {rate} {kw}: {len(got.raw)} records, gpu {tg:.2f} s, oracle {to:.2f} s, {'equal' if same else 'DIFFERENT'}", flush=True)
        bad += not same
print("all equal" if not bad else f"{bad} configurations differ")
sys.exit(1 if bad else 0)
