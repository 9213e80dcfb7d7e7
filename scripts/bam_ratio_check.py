#!/usr/bin/env python3
"""What the BAM writer makes of records that share their CIGAR (default flags: the windows of one locus all take their union's alignment):
1500 synthetic reads with 43 records each through flx_sam_write; prints bytes per read and the writer's rate. No GPU involved.
usage: bam_ratio_check.py [threads]"""
import ctypes as C, hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from floxer_amd import capi
L = capi.lib()
so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "floxer_amd", "libfloxer_amd.so")
print("library", hashlib.md5(open(so, "rb").read()).hexdigest(), os.path.getsize(so))
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(1)
NR, RPR = 1500, 43
reads = [rng.integers(1, 5, size=10000).astype(np.uint8) for _ in range(NR)]
names = [f"id_{i}_chromosome_{i % 5}_position_{int(rng.integers(0, 5e7))}_max_errors_800" for i in range(NR)]
cigs, rows, off = [], [], 0
for i in range(NR):
    ops = []
    while len(ops) < 1500:
        ops.append((int(rng.geometric(1 / 12.0)) << 4) | 7)
        ops.append((1 << 4) | int(rng.choice([8, 1, 2])))
    c = np.array(ops[:1500], dtype=np.uint32)
    cigs.append(c)
    pos, flag = int(rng.integers(0, 4e7)), 16 if i % 2 else 0
    for k in range(RPR):
        rows.append((i, flag | (256 if k else 0), i % 5, pos, 770, off, len(c)))
    off += len(c)
cig = np.concatenate(cigs)
n = len(rows)
recs = (capi.Record * n)(*[capi.Record(*r, 0) for r in rows])
ids = (C.c_char_p * NR)(*[nm.encode() for nm in names])
pool = np.concatenate(reads)
offs = np.zeros(NR + 1, dtype=np.uint64)
offs[1:] = np.cumsum([len(r) for r in reads])
quals = (C.c_char_p * NR)(*[b"I" * len(r) for r in reads])
ref_ids = (C.c_char_p * 5)(*[f"chromosome_{i}".encode() for i in range(5)])
ref_lens = np.array([50_000_000] * 5, dtype=np.uint64)
w = C.c_void_p()
path = "/tmp/flx_bam_ratio_check.bam"
capi.check(L.flx_sam_open(path.encode(), ref_ids, capi.ptr(ref_lens, capi.u64p), 5, C.byref(w)))
capi.check(L.flx_sam_set_threads(w, threads))
t0 = time.time()
capi.check(L.flx_sam_write(w, ids, capi.ptr(pool, capi.u8p), capi.ptr(offs, capi.u64p), quals, recs, n, capi.ptr(cig, capi.u32p)))
capi.check(L.flx_sam_close(w))
dt = time.time() - t0
sz = os.path.getsize(path)
os.remove(path)
print(f"{n} records of {NR} reads, {threads} threads: {sz} bytes = {sz / NR:.0f} per read; {dt:.2f} s = {n * 6.15e3 / 1e9 / dt:.2f} GB/s of records")
