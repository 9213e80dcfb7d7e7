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
