#!/usr/bin/env python3
"""First records of a BAM file: name, flag, CIGAR operations, a hash of the CIGAR bytes, record size - to see what the writer's deflate has to
work with. usage: bam_peek.py file.bam [n_records]"""
import gzip, hashlib, struct, sys
f = gzip.open(sys.argv[1], "rb")
n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 60
assert f.read(4) == b"BAM\x01"
l_text, = struct.unpack("<i", f.read(4)); f.read(l_text)
n_ref, = struct.unpack("<i", f.read(4))
for _ in range(n_ref):
    l, = struct.unpack("<i", f.read(4)); f.read(l + 4)
prev = None
for i in range(n_show):
    h = f.read(4)
    if len(h) < 4: break
    bs, = struct.unpack("<i", h)
    rec = f.read(bs)
    ref, pos, l_name, mapq, bin_, n_cig, flag, l_seq = struct.unpack("<iiBBHHHi", rec[:20])
    name = rec[32:32 + l_name - 1].decode()
    cig = rec[32 + l_name: 32 + l_name + 4 * n_cig]
    same = prev is not None and cig == prev
    print(f"{i:4d} size {bs + 4:6d} name {name[:28]:28s} flag {flag:4d} ref {ref} pos {pos:9d} ops {n_cig:5d} seq {l_seq:5d} cigar {hashlib.md5(cig).hexdigest()[:8]} {'= previous' if same else ''}")
    prev = cig
