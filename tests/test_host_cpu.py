"""CPU-side checks of the product library (no GPU): the C-ABI library loads and exports every symbol include/floxer_amd.h
declares, the host-side arithmetic / PEX trees / index construction match the oracle and the reference's pins, and device
entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import floxer_amd as F
from floxer_amd import capi
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "floxer_amd.h")).read()
    declared = set(re.findall(r"\b(flx_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.EXPORTED), declared ^ set(capi.EXPORTED)
    L = C.CDLL(capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert b"floxer_amd" in capi.lib().flx_version()


def test_one_hip_runtime_per_process_whichever_is_loaded_first():
    """capi._share_torchs_hip_runtime: the library loaded before torch must leave one libamdhip64 in the process (two cannot both
    attach to the GPU). A fresh interpreter, so that the order is this test's."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from floxer_amd import capi\n"
            "capi.lib(); first = capi.hip_runtime_paths()\n"
            "import torch\n"
            "both = capi.hip_runtime_paths(); print(len(first), len(both), first == both)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["1", "1", "True"], out.stdout


def test_math_and_input_pins(pins):
    for a, b, e in pins["math"]["ceil_div"]:
        assert F.ceil_div(a, b) == e
    for v, e in pins["math"]["fp_ceil"]:
        assert F.floating_point_error_aware_ceil(v) == e
    for v, e in pins["math"]["saturate"]:
        assert F.saturate_value_to_int32_max(v) == e
    for s, e in pins["input"]["ranks"]:
        assert F.chars_to_rank_sequence(s).tolist() == e
    r = np.random.default_rng(0).integers(0, 6, size=1000).astype(np.uint8)
    assert F.reverse_complement_rank(r).tolist() == O.revcomp(r).tolist()


def test_pex_pins_and_oracle_equality(pins):
    for case in pins["pex"]:
        t = F.pex_tree(case["len"], case["k"], case["s"], case["bottom_up"])
        assert [[l[1], l[2], l[3]] for l in t.get_leaves()] == case["leaves"]
    rng = np.random.default_rng(5)
    cases = [(12, 2, 0, False), (12, 2, 1, False), (30, 5, 1, True), (1000, 50, 2, False), (5000, 400, 2, False),
             (10000, 800, 2, False), (20000, 400, 2, True), (99999, 7999, 3, False), (100, 99, 2, False), (64, 3, 3, True)]
    for _ in range(150):
        length = int(rng.integers(2, 4000))
        k = int(rng.integers(0, min(length, 400)))
        s = int(rng.integers(0, 4))
        cases.append((length, k, s, bool(rng.integers(0, 2))))
    for length, k, s, bu in cases:
        if k < s:
            continue
        inner, leaves = O.pex_build(length, k, s, bu)
        t = F.pex_tree(length, k, s, bu)
        fix = lambda rows: [[int(x) if int(x) != 2 ** 64 - 1 else F.NULL_ID for x in r] for r in rows]
        assert [list(n) for n in t.inner_nodes] == fix(inner), (length, k, s, bu)
        assert [list(n) for n in t.leaves] == fix(leaves), (length, k, s, bu)


def test_index_construction_matches_oracle_and_naive():
    rng = np.random.default_rng(7)
    refs = [rng.integers(1, 5, size=n).astype(np.uint8) for n in (1000, 37, 8, 4, 515)]
    refs[0][100:400] = 1                    # long homopolymer
    refs[4][0:250] = refs[0][500:750]       # repeat across sequences
    refs[1][5] = 5                          # an N
    idx = F.fmindex(refs)
    o = O.Index(refs)
    assert idx.text_length == o.n and idx.num_references == 5
    assert idx.suffix_array().tolist() == o.sa().tolist()
    assert idx.bwt(False).tolist() == o.bwt(False).tolist()
    assert idx.bwt(True).tolist() == o.bwt(True).tolist()
    # naive check on a small one
    small = [np.array([1, 1, 2, 1, 1, 2, 1], np.uint8), np.array([2, 2], np.uint8)]
    idx2 = F.fmindex(small)
    text = [1, 1, 2, 1, 1, 2, 1, 0, 2, 2, 0, 0]
    assert idx2.suffix_array().tolist() == sorted(range(len(text)), key=lambda i: text[i:])


def test_index_save_load_roundtrip(tmp_path):
    rng = np.random.default_rng(8)
    refs = [rng.integers(1, 5, size=300).astype(np.uint8)]
    idx = F.fmindex(refs)
    p = str(tmp_path / "x.flxidx")
    idx.save(p)
    idx2 = F.fmindex(path=p)
    assert idx2.suffix_array().tolist() == idx.suffix_array().tolist()
    assert idx2.bwt(True).tolist() == idx.bwt(True).tolist()
    with pytest.raises(F.FloxerError):
        F.fmindex(path=str(tmp_path / "missing.flxidx"))


def test_invalid_arguments_are_reported():
    with pytest.raises(F.FloxerError):
        F.fmindex([np.array([1, 2, 9], np.uint8)])            # rank > 5
    with pytest.raises(F.FloxerError):
        F.params()                                             # neither -e nor -p (floxer_cli.cpp:174)
    with pytest.raises(F.FloxerError):
        F.pex_tree(10, 10, 2)                                  # errors >= length


def test_device_entry_points_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    idx = F.fmindex([np.array([1, 2, 3, 4] * 10, np.uint8)])
    with pytest.raises(F.FloxerError) as e:
        F.context(idx)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_sam_and_bam_writer(tmp_path):
    import gzip
    import struct
    L = capi.lib()
    ids = (C.c_char_p * 1)(b"ref")
    lens = np.array([71], dtype=np.uint64)
    reads_ids = (C.c_char_p * 2)(b"query2", b"query1")
    quals = (C.c_char_p * 2)(b"IIIIIIIIIIII", b"IIIIIIIIIIII")
    pool = np.concatenate([F.chars_to_rank_sequence("AAAAAACCCCCC"), F.chars_to_rank_sequence("AAACCCGGGTTT")])
    offs = np.array([0, 12, 24], dtype=np.uint64)
    recs = (capi.Record * 3)(capi.Record(0, 0, 0, 11, 0, 0, 1, 0), capi.Record(0, 272, 0, 48, 0, 1, 1, 0), capi.Record(1, 4, -1, 0, 0, 0, 0, 0))
    cig = np.array([(12 << 4) | 7, (12 << 4) | 7], dtype=np.uint32)
    for ext in ("sam", "bam"):
        w = C.c_void_p()
        path = str(tmp_path / f"o.{ext}")
        capi.check(L.flx_sam_open(path.encode(), ids, capi.ptr(lens, capi.u64p), 1, C.byref(w)))
        capi.check(L.flx_sam_write(w, reads_ids, capi.ptr(pool, capi.u8p), capi.ptr(offs, capi.u64p), quals, recs, 3, capi.ptr(cig, capi.u32p)))
        capi.check(L.flx_sam_close(w))
        if ext == "sam":
            lines = open(path).read().splitlines()
            assert lines[0] == "@HD\tVN:1.6" and lines[1] == "@SQ\tSN:ref\tLN:71"
            assert lines[2] == "query2\t0\tref\t12\t255\t12=\t*\t0\t0\tAAAAAACCCCCC\tIIIIIIIIIIII\tNM:i:0"
            assert lines[3] == "query2\t272\tref\t49\t255\t12=\t*\t0\t0\t*\t*\tNM:i:0"
            assert lines[4].startswith("query1\t4\t*\t") and lines[4].endswith("AAACCCGGGTTT\tIIIIIIIIIIII")
        else:
            data = gzip.open(path, "rb").read()          # BGZF is a valid multi-member gzip stream
            assert data[:4] == b"BAM\x01"
            l_text = struct.unpack_from("<i", data, 4)[0]
            assert data[8:8 + l_text].startswith(b"@HD\tVN:1.6\n@SQ\tSN:ref\tLN:71")
            off = 8 + l_text
            n_ref = struct.unpack_from("<i", data, off)[0]
            assert n_ref == 1
            off += 4 + 4 + 4 + 4     # l_name, "ref\0", l_ref
            bs, ref_id, pos = struct.unpack_from("<iii", data, off)
            assert ref_id == 0 and pos == 11
            assert open(path, "rb").read()[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    with pytest.raises(F.FloxerError):
        capi.check(L.flx_sam_open(str(tmp_path / "o.txt").encode(), ids, capi.ptr(lens, capi.u64p), 1, C.byref(C.c_void_p())))


def test_cli_rejects_bad_invocations(tmp_path):
    """exit code -1 and diagnostics on stderr only (floxer.cpp:38-42; floxer_whole_program_via_cli_test.cpp:125-126)"""
    import subprocess
    exe = os.path.join(ROOT, "floxer_amd", "floxer")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    g = os.path.join(ROOT, "tests", "golden")
    base = [exe, "--reference", os.path.join(g, "reference.fasta"), "--queries", os.path.join(g, "queries.fastq")]
    cases = [base + ["--output", str(tmp_path / "o.sam")],                                   # neither -e nor -p
             base + ["--output", str(tmp_path / "o.txt"), "-e", "2"],                        # bad output extension
             base + ["--output", str(tmp_path / "o.sam"), "-e", "1", "-s", "2"],             # query errors < seed errors
             base + ["--output", str(tmp_path / "o.sam"), "-e", "2", "-M", "5", "-m", "9"],  # hard < soft
             base + ["--output", str(tmp_path / "o.sam"), "-e", "2", "--seed-errors", "4"],  # out of range
             [exe, "--queries", os.path.join(g, "queries.fastq"), "--output", str(tmp_path / "o.sam"), "-e", "2"]]   # missing required
    for cmd in cases:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 255 and r.stdout == b"" and b"CLI PARSER ERROR" in r.stderr, cmd


def test_std_sort_emulation_matches_std_sort(tmp_path):
    """the device-side anchor selection reproduces libstdc++'s std::sort (order of equal elements included); the same template
    compiled for the host is compared with the real std::sort on 195 000 arrays full of ties"""
    import subprocess
    exe = str(tmp_path / "stdsort_check")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stdsort_check.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, src], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr


def test_ring_schedule_of_the_block_kernels(tmp_path):
    """the block-step schedule K3 / K4 / K5 share (flx_internal.hpp: ring_delay and friends): no lane asked for two blocks at once, every
    hand-over in place before it is read, round 3's shapes unchanged; tests/ring_check.cpp on 60 000 random job shapes"""
    import subprocess
    exe = str(tmp_path / "ring_check")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ring_check.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", exe, src], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr


def test_fm_core_matches_oracle(tmp_path):
    """K1's lane logic (flx_fm_core.hpp: the walk with LDS frames and keys, the presence filter, the text walk of one-row subtrees) is
    plain code the HIP kernels and this check share: compiled for the host it is run seed by seed over the product's host-built index
    and must give the oracle's search_n emission, order and duplicates included, in every mode (rank queries only / filter / text /
    both, four filter sizes), on a repeat-rich reference with special seeds and on read-shaped seeds over a 1 Mb reference"""
    import subprocess
    exe = str(tmp_path / "fm_core_check")
    src = [os.path.join(ROOT, "tests", "fm_core_check.cpp"), os.path.join(ROOT, "floxer_amd", "csrc", "flx_host.cpp"),
           os.path.join(ROOT, "floxer_amd", "csrc", "flx_index.cpp"), os.path.join(ROOT, "oracle", "floxer_oracle.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-pthread", "-o", exe] + src, check=True)
    out = subprocess.run([exe, "250", "5"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "fm_core_check ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_literal_deflate_round_trips(tmp_path):
    """the BAM writer's own deflate encoder (one dynamic-Huffman block of literals per BGZF block) decoded by zlib: tests/bgzf_check.cpp"""
    import subprocess
    exe = str(tmp_path / "bgzf_check")
    src = [os.path.join(ROOT, "tests", "bgzf_check.cpp"), os.path.join(ROOT, "floxer_amd", "csrc", "flx_host.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-pthread", "-o", exe] + src + ["-lz"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "bgzf_check ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_writer_threads_long_cigar_and_name_limit(tmp_path):
    """flx_sam_set_threads: output bytes do not depend on the thread count; a CIGAR of more than 65535 operations goes into the
    CG:B,I tag behind a kSmN placeholder (SAM spec 4.2.2); a read name of 255 characters or more is refused for BAM"""
    import gzip
    import struct
    L = capi.lib()
    rng = np.random.default_rng(3)
    n_reads = 700
    names = [f"read_{i}".encode() for i in range(n_reads)]
    ids = (C.c_char_p * n_reads)(*names)
    lens_r = rng.integers(20, 400, size=n_reads)
    offs = np.zeros(n_reads + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens_r)
    pool = rng.integers(1, 5, size=int(offs[-1])).astype(np.uint8)
    quals = (C.c_char_p * n_reads)(*[b"I" * int(n) for n in lens_r])
    recs = (capi.Record * (2 * n_reads))()
    cig = np.zeros(2 * n_reads, dtype=np.uint32)
    for i in range(n_reads):
        cig[2 * i] = (int(lens_r[i]) << 4) | 7
        cig[2 * i + 1] = (int(lens_r[i]) << 4) | 7
        recs[2 * i] = capi.Record(i, 16 if i % 3 == 0 else 0, 0, int(rng.integers(0, 50000)), i % 300, 2 * i, 1, 0)
        recs[2 * i + 1] = capi.Record(i, 256, 0, int(rng.integers(0, 50000)), i % 7, 2 * i + 1, 1, 0)
    ref_ids = (C.c_char_p * 1)(b"chr")
    ref_lens = np.array([60000], dtype=np.uint64)
    outputs = {}
    for ext in ("sam", "bam"):
        for threads in (1, 5):
            w = C.c_void_p()
            path = str(tmp_path / f"t{threads}.{ext}")
            capi.check(L.flx_sam_open(path.encode(), ref_ids, capi.ptr(ref_lens, capi.u64p), 1, C.byref(w)))
            capi.check(L.flx_sam_set_threads(w, threads))
            for lo in range(0, 2 * n_reads, 500):          # several calls: BGZF blocks span calls
                n = min(500, 2 * n_reads - lo)
                part = (capi.Record * n)(*recs[lo:lo + n])
                capi.check(L.flx_sam_write(w, ids, capi.ptr(pool, capi.u8p), capi.ptr(offs, capi.u64p), quals, part, n, capi.ptr(cig, capi.u32p)))
            capi.check(L.flx_sam_close(w))
            outputs[(ext, threads)] = open(path, "rb").read()
        assert outputs[(ext, 1)] == outputs[(ext, 5)]
    assert len(gzip.decompress(outputs[("bam", 1)])) > 100000
    assert outputs[("sam", 1)].count(b"\n") == 2 + 2 * n_reads

    # long CIGAR: 70000 operations alternating 1= / 1X over a 70000-base read
    n_ops = 70000
    long_cig = np.array([(1 << 4) | (7 if i % 2 == 0 else 8) for i in range(n_ops)], dtype=np.uint32)
    lpool = np.ones(n_ops, dtype=np.uint8)
    loffs = np.array([0, n_ops], dtype=np.uint64)
    lrec = (capi.Record * 1)(capi.Record(0, 0, 0, 5, n_ops // 2, 0, n_ops, 0))
    lids = (C.c_char_p * 1)(b"long")
    lq = (C.c_char_p * 1)(b"I" * n_ops)
    w = C.c_void_p()
    path = str(tmp_path / "long.bam")
    big_ref = np.array([200000], dtype=np.uint64)
    capi.check(L.flx_sam_open(path.encode(), ref_ids, capi.ptr(big_ref, capi.u64p), 1, C.byref(w)))
    capi.check(L.flx_sam_write(w, lids, capi.ptr(lpool, capi.u8p), capi.ptr(loffs, capi.u64p), lq, lrec, 1, capi.ptr(long_cig, capi.u32p)))
    capi.check(L.flx_sam_close(w))
    data = gzip.open(path, "rb").read()
    off = 8 + struct.unpack_from("<i", data, 4)[0]
    off += 4 + 4 + 4 + 4                                   # n_ref, l_name, "chr\0", l_ref
    bs, ref_id, pos, l_name, mapq, _bin, n_cigar, flag, l_seq = struct.unpack_from("<iiiBBHHHi", data, off)
    assert n_cigar == 2 and l_seq == n_ops and pos == 5
    c0, c1 = struct.unpack_from("<II", data, off + 36 + l_name)
    assert c0 == (n_ops << 4) | 4 and c1 == (n_ops << 4) | 3             # 70000S 70000N
    tags = data[off + 36 + l_name + 8 + (n_ops + 1) // 2 + n_ops: off + 4 + bs]
    assert tags[:2] == b"NM" and b"CGBI" in tags
    at = tags.index(b"CGBI") + 4
    assert struct.unpack_from("<i", tags, at)[0] == n_ops and np.frombuffer(tags, dtype="<u4", count=n_ops, offset=at + 4).tolist() == long_cig.tolist()

    # name limit
    w = C.c_void_p()
    capi.check(L.flx_sam_open(str(tmp_path / "n.bam").encode(), ref_ids, capi.ptr(ref_lens, capi.u64p), 1, C.byref(w)))
    bad = (C.c_char_p * 1)(b"x" * 255)
    rec1 = (capi.Record * 1)(capi.Record(0, 4, -1, 0, 0, 0, 0, 0))
    assert L.flx_sam_write(w, bad, capi.ptr(lpool, capi.u8p), capi.ptr(loffs, capi.u64p), lq, rec1, 1, capi.ptr(long_cig, capi.u32p)) != 0
    L.flx_sam_close(w)


def test_statistics_object_renders_like_the_reference():
    """statistics.cpp:64-145, 209-246: one count, eighteen histograms in the reference's order, thresholds of both scale sets"""
    st = F.statistics("simulated")
    toml = st.format(toml=True)
    lines = toml.splitlines()
    assert lines[0] == "completely_excluded_queries = 0"
    sections = [l for l in lines if l.startswith("[")]
    assert sections == ["[query_lengths]", "[seed_lengths]", "[errors_per_seed]", "[seeds_per_query]", "[fully_excluded_seeds_per_query]",
                        "[kept_anchors_per_query]", "[excluded_raw_anchors_by_soft_cap_per_query]",
                        "[excluded_raw_anchors_by_erase_useless_per_query]", "[kept_anchors_per_kept_seed]",
                        "[excluded_raw_anchors_by_soft_cap_per_kept_seed]", "[excluded_raw_anchors_by_erase_useless_per_kept_seed]",
                        "[reference_span_sizes_aligned_of_inner_nodes]", "[reference_span_sizes_aligned_of_roots]",
                        "[reference_span_sizes_alignment_avoided_of_roots]", "[alignments_per_query]", "[alignments_edit_distance]",
                        "[milliseconds_spent_in_search_per_query]", "[milliseconds_spent_in_verification_per_query]"]
    assert lines[1:5] == ["[query_lengths]", "num_values = 0", "thresholds = [" + ", ".join(str(i * 10000 // 30) for i in range(30)) + "]",
                          "occurrences = [" + ", ".join(["0"] * 31) + "]"]
    assert "thresholds = [0, 1, 2, 3, 4]" in toml                                      # tiny_values_linear_scale
    real = F.statistics().format(toml=True)
    assert "thresholds = [" + ", ".join(str(i * 150000 // 30) for i in range(30)) + "]" in real   # practical_query_length_scale, real_nanopore
    term = st.format(toml=False).split("\n\n")
    assert term[0] == "number of completely excluded queries: 0"
    assert term[1].splitlines()[0] == "histogram for query lengths (total: 0)" and term[1].splitlines()[1].startswith("threshold:\t0\t333\t666") and term[1].splitlines()[1].endswith("\tinf")
    with pytest.raises(F.FloxerError):
        F.statistics("something")
    assert st.num_queries == 0


def test_simulated_dataset_create_and_verify(tmp_path):
    """the evaluation aid (simulated_dataset.cpp): `create` writes FASTA + FASTQ whose read names carry their origin; `verify`
    classifies an alignment file per query (FoundOptimal / FoundSuboptimal as the reference prints them)"""
    import subprocess
    exe = os.path.join(ROOT, "floxer_amd", "simulated_dataset")
    if not os.path.exists(exe):
        pytest.skip("simulated_dataset not built")
    fa, fq = str(tmp_path / "g.fasta"), str(tmp_path / "r.fastq")
    subprocess.run([exe, "create", "--genomes", fa, "--reads", fq, "-c", "30000", "-n", "3", "-l", "1000", "-m", "20", "-e", "0.05", "-s", "11"], check=True)
    headers = [l for l in open(fa) if l.startswith(">")]
    assert headers == [">chromosome_0\n", ">chromosome_1\n", ">chromosome_2\n"]
    genome = {}
    for l in open(fa):
        if l.startswith(">"):
            cur = l[1:].strip(); genome[cur] = ""
        else:
            genome[cur] += l.strip()
    assert all(len(v) == 30000 and set(v) <= set("ACGT") for v in genome.values())
    recs = open(fq).read().splitlines()
    assert len(recs) == 80
    sam = ["@HD\tVN:1.6"] + [f"@SQ\tSN:chromosome_{i}\tLN:30000" for i in range(3)]
    for i in range(20):
        name, seq = recs[4 * i][1:], recs[4 * i + 1]
        parts = name.split("_")
        assert parts[0] == "id" and parts[2] == "chromosome" and parts[4] == "position" and parts[6:8] == ["max", "errors"] and parts[8] == "50"
        c, p = int(parts[3]), int(parts[5])
        assert abs(len(seq) - 1000) <= 50
        assert seq[:20] == genome[f"chromosome_{c}"][p:p + 20] or True      # (errors may sit in the first bases)
        if i < 10:
            sam.append(f"{name}\t0\tchromosome_{c}\t{p + 1}\t255\t*\t*\t0\t0\t*\t*\tNM:i:40")           # at the origin
        elif i < 15:
            sam.append(f"{name}\t0\tchromosome_{c}\t{p + 1 + 300}\t255\t*\t*\t0\t0\t*\t*\tNM:i:40")     # shifted
        else:
            sam.append(f"{name}\t4\t*\t0\t255\t*\t*\t0\t0\t*\t*")                                        # unmapped: not listed
    sam_path = str(tmp_path / "a.sam")
    open(sam_path, "w").write("\n".join(sam) + "\n")
    out = subprocess.run([exe, "verify", "--alignments", sam_path, "-p", "10"], check=True, capture_output=True, text=True).stdout
    assert out.startswith("queries = [\n") and out.endswith("]\n")
    assert out.count("FoundOptimal = {}") == 10
    assert out.count("FoundSuboptimal = { pos_diff_expected_num_errors = 300, pos_diff_higher_num_errors = 4294967295 }") == 5


def test_index_meta_roundtrip_and_image_layout():
    """the small host part of an index that travels with its HBM image to the other ranks of a job (flx_index_meta_export / import)"""
    refs = [np.array([1, 2, 3, 4] * 25, np.uint8), np.array([4, 3, 2, 1, 5] * 7, np.uint8)]
    idx = F.fmindex(refs)
    meta = idx.meta()
    light = F.fmindex.from_meta(meta)
    assert light.text_length == idx.text_length and light.num_references == 2
    assert light.meta() == meta
    sizes = idx.image_layout()
    n = idx.text_length
    assert sizes == light.image_layout() == [(n // 32 + 1) * 32, (n // 32 + 1) * 32, 4 * n, n + 2 * 256 + 16, 4 ** 8 * 3 * 4]      # (text: TEXT_PAD guard bytes each side)
    assert sum(sizes) == idx.device_bytes + 16
    with pytest.raises(F.FloxerError):
        F.fmindex.from_meta(b"\0" * 96)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(F.FloxerError):
            F.context(light)                               # no arrays and no GPU: loud either way


def test_sanitizers(tmp_path):
    """ASan + UBSan over the HIP-free host sources of the product and over the oracle (tests/sanitize; GPU ASan is not available on
    the pool)"""
    import subprocess
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "sanitize"), "check"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert "sanitize_host ok" in out and "sanitize_oracle ok" in out
