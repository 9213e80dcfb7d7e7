#!/usr/bin/env python3
"""torch started in a process in which the library has already worked (host-built index, a run, the oracle's threads, a closed context).
usage: torch_after_lib.py E|F|G   E: host-built index + run + oracle run (8 threads) + close, then torch; F: the same without the oracle;
G: as E, with a second context alive while torch starts"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
mode = sys.argv[1]
import floxer_amd as F
from floxer_amd import simulate as S
import oracle_lib as O
genome = S.make_genome(600000, 2, seed=71)
reads, _, _ = S.make_reads(genome, 40, 10000, 0.08, seed=72)
idx = F.fmindex(genome)
ctx = F.context(idx)
recs = F.aligner(ctx, F.params(error_probability=0.08)).align_reads(reads).records()
if mode in "EG":
    exp = O.Index(genome).run(reads, O.params(error_probability=0.08), threads=8)
    assert recs == exp.records()
ctx.close()
del ctx, idx
def libs():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if any(k in l for k in ("hip64", "hsa-runtime", "hsakmt"))})
print(mode, "before torch:", libs())
g2 = S.make_genome(300000, 3, seed=51)
idx2 = F.fmindex(g2, device=0)
ctx2 = F.context(idx2) if mode == "G" else None
import torch
print(mode, "after import torch:", libs())
try:
    t = torch.empty(1000, dtype=torch.uint8, device="cuda:0"); torch.cuda.synchronize()
    print(mode, "torch after the library: ok,", len(recs), "records")
except Exception as e:
    print(mode, "FAILED:", type(e).__name__, str(e)[:200])
