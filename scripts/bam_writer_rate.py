"""Rate of the BAM / SAM writer alone (host code, no GPU): N reads of 10 kb with R records each whose CIGARs look like those of
8 % error reads (~1600 operations), written with T threads. usage: bam_writer_rate.py [reads] [records_per_read] [threads]"""
import ctypes as C, os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from floxer_amd import capi

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
per_read = int(sys.argv[2]) if len(sys.argv) > 2 else 43
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
L = capi.lib()
rng = np.random.default_rng(5)
read_len = 10000
names = [f"read_{i}".encode() for i in range(n_reads)]
ids = (C.c_char_p * n_reads)(*names)
offs = (np.arange(n_reads + 1, dtype=np.uint64) * read_len)
pool = rng.integers(1, 5, size=n_reads * read_len).astype(np.uint8)
qual = b"I" * read_len
quals = (C.c_char_p * n_reads)(*[qual] * n_reads)
# one CIGAR per read (its records share the shape, as the records of one locus do): runs of = broken by X / I / D
cigs, cig_off = [], [0]
for i in range(n_reads):
    ops, left = [], read_len
    while left > 0:
        run = int(min(left, rng.geometric(0.08)))
        ops.append((run << 4) | 7); left -= run
        if left > 0:
            op = int(rng.choice([8, 1, 2]))
            ops.append((1 << 4) | op)
            if op != 2: left -= 1
    cigs.append(np.array(ops, dtype=np.uint32)); cig_off.append(cig_off[-1] + len(ops))
cig = np.concatenate(cigs)
recs = (capi.Record * (n_reads * per_read))()
k = 0
for i in range(n_reads):
    for r in range(per_read):
        recs[k] = capi.Record(i, 0 if r == 0 else 256, 0, int(rng.integers(0, 1_000_000)), 700 + r, cig_off[i], cig_off[i + 1] - cig_off[i], 0)
        k += 1
ref_ids = (C.c_char_p * 1)(b"chr1")
ref_lens = np.array([2_000_000], dtype=np.uint64)
for ext in ("bam", "sam"):
    path = os.path.join(tempfile.gettempdir(), f"flx_writer_rate.{ext}")
    w = C.c_void_p()
    capi.check(L.flx_sam_open(path.encode(), ref_ids, capi.ptr(ref_lens, capi.u64p), 1, C.byref(w)))
    capi.check(L.flx_sam_set_threads(w, threads))
    t0 = time.time()
    capi.check(L.flx_sam_write(w, ids, capi.ptr(pool, capi.u8p), capi.ptr(offs, capi.u64p), quals, recs, len(recs), capi.ptr(cig, capi.u32p)))
    capi.check(L.flx_sam_close(w))
    dt = time.time() - t0
    size = os.path.getsize(path)
    os.remove(path)
    print(f"{ext}: {n_reads} reads x {per_read} records, {threads} threads: {dt:.2f} s = {n_reads / dt:.0f} reads/s, {size / 1e6:.0f} MB written, "
          f"level {os.environ.get('FLX_BGZF_LEVEL', '1')}")
