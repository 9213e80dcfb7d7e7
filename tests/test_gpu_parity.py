"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs, plus the reference's pins
run through the GPU path. Bit-exact: integer / byte / index work only. Needs an MI355X (-m gpu)."""
import os

import numpy as np
import pytest

import floxer_amd as F
from floxer_amd import capi
from floxer_amd import simulate as S
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _recs_of(result, read):
    """the records of one read of a run as comparable tuples (CIGAR words as bytes)"""
    return [(int(r[1]), int(r[2]), int(r[3]), int(r[4]), result.cigars[r[5]: r[5] + r[6]].tobytes()) for r in result.rows[result.rows[:, 0] == read]]


@pytest.fixture(scope="module")
def small_genome():
    rng = np.random.default_rng(11)
    refs = [rng.integers(1, 5, size=n).astype(np.uint8) for n in (60000, 20000, 777)]
    refs[0][1000:1400] = 1                          # homopolymer: many hits, exercises the hard cap
    refs[1][500:3500] = refs[0][20000:23000]        # 3 kb repeat across references
    refs[2][100:110] = 5                            # a run of N
    idx = F.fmindex(refs)
    ctx = F.context(idx)
    yield refs, idx, ctx, O.Index(refs)
    ctx.close()


# ---------------------------------------------------------------- K1: search_n emission
def _make_seeds(rng, refs, n, kmax=3, lens=(8, 60)):
    pool, seeds = [], []
    for i in range(n):
        r = refs[int(rng.integers(0, len(refs)))]
        L = int(rng.integers(lens[0], lens[1]))
        st = int(rng.integers(0, len(r) - L))
        s = r[st:st + L].copy()
        k = int(rng.integers(0, kmax + 1))
        for _ in range(int(rng.integers(0, k + 2))):
            p = int(rng.integers(0, len(s)))
            kind = rng.integers(0, 3)
            if kind == 0:
                s[p] = rng.integers(1, 5)
            elif kind == 1 and len(s) > 6:
                s = np.delete(s, p)
            else:
                s = np.insert(s, p, rng.integers(1, 5))
        if i % 17 == 0:
            s = rng.integers(1, 5, size=len(s)).astype(np.uint8)     # unrelated seed
        if i % 29 == 0:
            s[:] = 1                                                   # poly-A seed (hits the 400-A run)
        seeds.append((sum(len(x) for x in pool), len(s), k, i))
        pool.append(s.astype(np.uint8))
    return np.concatenate(pool), seeds


def test_search_groups_match_oracle(small_genome):
    refs, idx, ctx, oidx = small_genome
    rng = np.random.default_rng(21)
    pool, seeds = _make_seeds(rng, refs, 400)
    sr = F.searcher(ctx)
    for cap in (501, 50, 10 ** 6):
        got = sr.search_groups(pool, seeds, max_hits=cap)
        for i, (off, ln, k, _) in enumerate(seeds):
            exp, _ctr = oidx.search_groups(pool[off:off + ln], k, n=cap)
            mine = got[got[:, 0] == i][:, 1:]
            assert mine.tolist() == exp.tolist(), (cap, i, ln, k)


def test_search_groups_of_the_filtered_walk_match_oracle(small_genome, monkeypatch):
    """the default search path (stack in LDS, presence filter, one-row subtrees against the text; hits put back into emission order by
    their keys) asked for its raw groups: every group of every seed, in the oracle's order; also with the filter or the text walk off"""
    refs, idx, ctx, oidx = small_genome
    rng = np.random.default_rng(23)
    pool, seeds = _make_seeds(rng, refs, 400)
    # (no cap: this walk finds the groups in another order than the oracle, so a cap would cut another group short)
    exp = [oidx.search_groups(pool[off:off + ln], k, n=2 ** 40)[0].tolist() for off, ln, k, _ in seeds]
    sr = F.searcher(ctx)
    monkeypatch.setenv("FLX_FM_KEYED_RAW", "1")
    for env in ({}, {"FLX_FM_NO_FILTER": "1"}, {"FLX_FM_NO_TEXT": "1"}):
        for k_, v in env.items():
            monkeypatch.setenv(k_, v)
        got = sr.search_groups(pool, seeds, max_hits=2 ** 31)
        for i in range(len(seeds)):
            assert got[got[:, 0] == i][:, 1:].tolist() == exp[i], (env, i, seeds[i])
        for k_ in env:
            monkeypatch.delenv(k_)


def test_search_short_and_degenerate_seeds(small_genome):
    refs, idx, ctx, oidx = small_genome
    sr = F.searcher(ctx)
    pool = np.concatenate([refs[0][:40], np.array([5, 5, 5, 5, 5, 5], np.uint8), refs[2][95:115]])
    seeds = [(0, 1, 0, 0), (0, 2, 1, 1), (0, 3, 2, 2), (0, 4, 3, 3), (0, 5, 3, 4), (40, 6, 1, 5), (46, 20, 2, 6), (3, 4, 2, 7)]
    got = sr.search_groups(pool, seeds, max_hits=501)
    for i, (off, ln, k, _) in enumerate(seeds):
        exp, _ = oidx.search_groups(pool[off:off + ln], k, n=501)
        assert got[got[:, 0] == i][:, 1:].tolist() == exp.tolist(), i


# ---------------------------------------------------------------- seam 1: search_seeds
@pytest.mark.parametrize("order,choice,erase,hard,soft", [
    ("count_first", "round_robin", True, 60, 7), ("errors_first", "full_groups", True, 60, 7), ("none", "first_reported", False, 60, 7),
    ("count_first", "full_groups", False, 60, 7),
    # the combinations the device-side selection takes (default order and choice), with caps on either side of its eight slots
    ("count_first", "round_robin", False, 60, 7), ("count_first", "round_robin", True, 500, 50), ("count_first", "round_robin", True, 5, 3),
    ("count_first", "round_robin", True, 3, 3)])
def test_search_seeds_match_oracle(small_genome, order, choice, erase, hard, soft):
    refs, idx, ctx, oidx = small_genome
    rng = np.random.default_rng(22)
    pool, seeds = _make_seeds(rng, refs, 300, kmax=2, lens=(12, 50))
    cfg = F.search_config(hard, soft, order, choice, erase)
    anchors, stats = F.searcher(ctx, cfg).search_seeds(pool, seeds)
    exp_a, exp_s = oidx.search_seeds(pool, [(o, l, k, leaf) for o, l, k, leaf in seeds], hard=hard, soft=soft, order=F.ORDER[order],
                                     choice=F.CHOICE[choice], erase=erase)
    assert stats.tolist() == exp_s.tolist()
    assert anchors.tolist() == exp_a.tolist()


def test_search_seeds_reference_setup(pins):
    s = pins["search_seeds_setup"]
    idx = F.fmindex(s["references"])
    ctx = F.context(idx)
    cfg = F.search_config(10, 10, "count_first", "round_robin", True)
    anchors, stats = F.searcher(ctx, cfg).search_seeds(s["query"], [tuple(x) for x in s["seeds"]])
    assert stats[:, 3].sum() == 0                      # search_test.cpp:74
    oa, os_ = O.Index(s["references"]).search_seeds(s["query"], s["seeds"], hard=10, soft=10)
    assert anchors.tolist() == oa.tolist() and stats.tolist() == os_.tolist()
    ctx.close()


# ---------------------------------------------------------------- seam 2: align
def _rand_align_case(rng, m, err, slack):
    q = rng.integers(1, 5, size=m).astype(np.uint8)
    core = []
    for c in q:
        r = rng.random()
        if r < err / 3:
            continue
        if r < 2 * err / 3:
            core.append(rng.integers(1, 5))
        core.append(rng.integers(1, 5) if r < err else c)
    left = int(rng.integers(0, slack + 1))
    ref = np.concatenate([rng.integers(1, 5, size=left), np.array(core, dtype=np.uint8), rng.integers(1, 5, size=slack - left)]).astype(np.uint8)
    return ref, q


def test_align_pins_on_gpu(pins, small_genome):
    _, _, ctx, _ = small_genome
    a = pins["alignment"]
    assert F.align(ctx, a["reference"], a["query"], a["k"]) == (a["nm"], a["start"], a["cigar"])
    assert F.align(ctx, a["reference"], a["query"], 0, F.MODE_EXISTS) is None
    assert F.align(ctx, a["reference"], a["query"], a["k"], F.MODE_WITHOUT_CIGAR)[:2] == (a["nm"], a["start"])
    ref = O.chars_to_ranks("A" * 17 + "C" * 19 + "G" * 18 + "T" * 17)
    rc = lambda s: O.revcomp(O.chars_to_ranks(s))
    assert F.align(ctx, ref, O.chars_to_ranks("GGGGAAGGGGGG"), 2) == (2, 44, "4=2I6=")
    assert F.align(ctx, ref, rc("GGGGAAGGGGGG"), 2) == (2, 26, "6=2I4=")
    assert F.align(ctx, ref, O.chars_to_ranks("TTTTTTTTTTGG"), 2) == (2, 61, "10=2I")
    assert F.align(ctx, ref, rc("TTTTTTTTTTGG"), 2) == (2, 7, "2I10=")
    t = pins["try_align_node"]
    w = np.array(t["reference"], np.uint8)[t["span"][0]: t["span"][0] + t["span"][1]]
    q = np.array(t["query"], np.uint8)[t["node"][0]: t["node"][1] + 1]
    r = F.align(ctx, w, q, t["node"][2])
    assert r[0] == t["nm"] and t["span"][0] + r[1] == t["start"]


def test_align_batch_random_all_shapes(small_genome):
    """every (words per lane, lanes per job) shape of the kernel, all three modes, ragged sizes in one batch"""
    _, _, ctx, _ = small_genome
    rng = np.random.default_rng(31)
    sizes = [1, 2, 7, 63, 64, 65, 127, 128, 129, 200, 256, 257, 300, 511, 513, 700, 1000, 1025, 1500, 2047, 2049, 3000, 4097,
             5000, 6200, 8200, 12500, 16500, 26000]
    refs, queries, jobs = [], [], []
    ro = qo = 0
    for m in sizes:
        for rep in range(2):
            err = 0.08 if rep == 0 else 0.3
            ref, q = _rand_align_case(rng, m, err, int(rng.integers(0, m // 4 + 10)))
            if m in (64, 200) and rep == 1:      # low complexity -> many ties
                ref = (ref % 2 + 1).astype(np.uint8)
                q = (q % 2 + 1).astype(np.uint8)
            k = int(m * (0.15 if rep == 0 else 0.25)) + int(rng.integers(0, 3))
            for mode in (0, 1, 2):
                jobs.append((ro, len(ref), qo, len(q), k, mode))
            refs.append(ref)
            queries.append(q)
            ro += len(ref)
            qo += len(q)
    rpool, qpool = np.concatenate(refs), np.concatenate(queries)
    got = F.align_batch(ctx, qpool, jobs, reference_pool=rpool)
    for (ro_, rl, qo_, ql, k, mode), g in zip(jobs, got):
        exp = O.align(rpool[ro_:ro_ + rl], qpool[qo_:qo_ + ql], k, mode=mode, algo=1)
        if mode == 0 and exp is not None:
            exp = (exp[0], 0, "")
        if mode == 1 and exp is not None:
            exp = (exp[0], exp[1], "")
        assert g == exp, (ql, rl, k, mode)


@pytest.mark.parametrize("shape", ["1,2", "1,8", "2,4", "3,4", "5,2", "5,8", "8,2"])
def test_align_batch_on_rings_that_wait(small_genome, monkeypatch, shape):
    """launch shapes with fewer lanes than a job's band asks for: every revolution of the ring waits (flx_internal.hpp: ring_delay), what the
    last lane hands to the first goes through the queue in LDS; existence, score / end and traced alignments (K4's slots, K5's reading of
    them) against the oracle, sizes from one word group to 26 000 rows"""
    _, _, ctx, _ = small_genome
    monkeypatch.setenv("FLX_FORCE_SHAPE", shape)
    rng = np.random.default_rng(33)
    refs, queries, jobs = [], [], []
    ro = qo = 0
    for m in [130, 257, 700, 1025, 1500, 2049, 3000, 4097, 5000, 6200, 8200, 10000, 12500, 16500, 26000]:
        for rep in range(2):
            err = 0.08 if rep == 0 else 0.25
            ref, q = _rand_align_case(rng, m, err, int(rng.integers(0, m // 4 + 10)))
            k = int(m * (0.1 if rep == 0 else 0.2)) + int(rng.integers(0, 3))
            for mode in (0, 1, 2):
                jobs.append((ro, len(ref), qo, len(q), k, mode))
            refs.append(ref)
            queries.append(q)
            ro += len(ref)
            qo += len(q)
    rpool, qpool = np.concatenate(refs), np.concatenate(queries)
    got = F.align_batch(ctx, qpool, jobs, reference_pool=rpool)
    for (ro_, rl, qo_, ql, k, mode), g in zip(jobs, got):
        exp = O.align(rpool[ro_:ro_ + rl], qpool[qo_:qo_ + ql], k, mode=mode, algo=1)
        if mode == 0 and exp is not None:
            exp = (exp[0], 0, "")
        if mode == 1 and exp is not None:
            exp = (exp[0], exp[1], "")
        assert g == exp, (shape, ql, rl, k, mode)


def test_align_against_context_text(small_genome):
    refs, idx, ctx, _ = small_genome
    rng = np.random.default_rng(32)
    q = refs[0][30000:30500].copy()
    q[100] = q[100] % 4 + 1
    q = np.delete(q, 300)
    jobs = [(29950, 620, 0, len(q), 10, 2), (0, 600, 0, len(q), 10, 0)]
    got = F.align_batch(ctx, q, jobs)
    exp = O.align(refs[0][29950:29950 + 620], q, 10, mode=2)
    assert got[0] == exp and got[1] is None


# ---------------------------------------------------------------- seam 3: whole path
def _records_equal(got, exp):
    assert got.skipped.tolist() == exp.skipped.tolist()
    assert got.records() == exp.records()


@pytest.mark.parametrize("kw", [dict(), dict(interval_optimization=True), dict(without_cigar=True),
                                dict(direct_full_verification=True, interval_optimization=True), dict(bottom_up_pex_tree=True),
                                dict(seed_errors=1, anchor_choice_strategy="full_groups"), dict(seed_sampling_step_size=2, num_anchors_per_task=3)])
def test_whole_path_matches_oracle(kw):
    genome = S.make_genome(200000, 2, seed=5)
    reads, names, truth = S.make_reads(genome, 24, 1200, 0.06, seed=6)
    reads += [np.zeros(0, np.uint8), np.array([1, 2, 3], np.uint8), np.random.default_rng(1).integers(1, 5, size=900).astype(np.uint8)]
    idx = F.fmindex(genome)
    ctx = F.context(idx)
    oidx = O.Index(genome)
    p = F.params(error_probability=0.06, **kw)
    got = F.aligner(ctx, p).align_reads(reads)
    okw = dict(error_probability=0.06, seed_errors=kw.get("seed_errors", 2), choice=kw.get("anchor_choice_strategy", "round_robin"),
               seed_step=kw.get("seed_sampling_step_size", 1), bottom_up=kw.get("bottom_up_pex_tree", False),
               interval_opt=kw.get("interval_optimization", False), direct_full=kw.get("direct_full_verification", False),
               anchors_per_task=kw.get("num_anchors_per_task", 3000), without_cigar=kw.get("without_cigar", False))
    exp = oidx.run(reads, O.params(**okw), threads=4)
    _records_equal(got, exp)
    mapped = sum(1 for r in got.records() if not r[1] & 4 and not r[1] & 256)
    assert mapped >= 20
    ctx.close()


@pytest.mark.parametrize("seed_errors", [0, 1])
def test_whole_program_pins_on_gpu(pins, seed_errors):
    """floxer_whole_program_via_cli_test.cpp:47-93 through the HIP path"""
    from test_oracle_pins import _read_fasta, _read_fastq
    g = os.path.join(os.path.dirname(__file__), "golden")
    refs = _read_fasta(os.path.join(g, "reference.fasta"))
    reads = _read_fastq(os.path.join(g, "queries.fastq"))
    idx = F.fmindex([F.chars_to_rank_sequence(s) for _, s in refs])
    ctx = F.context(idx)
    p = F.params(query_errors=2, seed_errors=seed_errors, extra_verification_ratio=2.0, interval_optimization=True)
    res = F.aligner(ctx, p).align_reads([F.chars_to_rank_sequence(s) for _, s, _ in reads])
    wp = pins["whole_program"]
    ids = [r[0] for r in reads]
    recs = res.records()
    assert {ids[r[0]] for r in recs} == set(wp["ids"])
    for ridx, flag, ref_id, pos, nm, cig in recs:
        name = ids[ridx]
        if name in wp["unmapped"]:
            assert flag == 4
            continue
        for ename, erev, pmin, pmax, enm, ecig in wp["expect"]:
            if ename == name and erev == bool(flag & 16):
                assert pmin <= pos <= pmax and nm == enm and cig == ecig
    seen = {(ids[r[0]], bool(r[1] & 16)) for r in recs if not r[1] & 4}
    for ename, erev, *_ in wp["expect"]:
        assert (ename, erev) in seen
    exp = O.Index([O.chars_to_ranks(s) for _, s in refs]).run([O.chars_to_ranks(s) for _, s, _ in reads],
                                                               O.params(query_errors=2, seed_errors=seed_errors, extra_ratio=2.0, interval_opt=True))
    assert recs == exp.records()
    ctx.close()


def test_full_size_reads_properties():
    """BASELINE config-1 read shape (5 kb @ 8 %): size-independent properties + equality with the oracle on a sample."""
    import re
    genome = S.make_genome(1000000, 1, seed=9)
    reads, names, truth = S.make_reads(genome, 48, 5000, 0.08, seed=10)
    idx = F.fmindex(genome)
    ctx = F.context(idx)
    p = F.params(error_probability=0.08)
    res = F.aligner(ctx, p).align_reads(reads)
    recs = res.records()
    g = genome[0]
    for ridx, flag, ref_id, pos, nm, cig in recs:
        if flag & 4:
            continue
        q = reads[ridx] if not flag & 16 else F.reverse_complement_rank(reads[ridx])
        i, j, cost = 0, pos, 0
        for ln, op in re.findall(r"(\d+)([=XID])", cig):
            ln = int(ln)
            if op == "=":
                assert (q[i:i + ln] == g[j:j + ln]).all(); i += ln; j += ln
            elif op == "X":
                assert (q[i:i + ln] != g[j:j + ln]).all(); i += ln; j += ln; cost += ln
            elif op == "I":
                i += ln; cost += ln
            else:
                j += ln; cost += ln
        assert i == len(q) and cost == nm and nm <= F.floating_point_error_aware_ceil(len(q) * 0.08)
    for i, (c, start, rev) in enumerate(truth):
        prim = [r for r in recs if r[0] == i and not r[1] & 256]
        assert len(prim) == 1 and not prim[0][1] & 4 and abs(prim[0][3] - start) <= 400 and bool(prim[0][1] & 16) == rev
    exp = O.Index(genome).run(reads[:12], O.params(error_probability=0.08), threads=8)
    assert [r for r in recs if r[0] < 12] == exp.records()
    ctx.close()


@pytest.mark.parametrize("seed_errors,ext", [(0, "sam"), (1, "sam"), (1, "bam")])
def test_cli_whole_program(pins, tmp_path, seed_errors, ext):
    """the reference's end-to-end test (floxer_whole_program_via_cli_test.cpp:17-143) against the drop-in CLI"""
    import gzip
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "floxer_amd", "floxer")
    g = os.path.join(root, "tests", "golden")
    out = str(tmp_path / f"out.{ext}")
    cmd = [exe, "--reference", os.path.join(g, "reference.fasta"), "--queries", os.path.join(g, "queries.fastq"), "--output", out,
           "--interval-optimization", "--console-debug-logs", "--query-errors", "2", "--seed-errors", str(seed_errors),
           "--extra-verification-ratio", "2", "--threads", "1"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == b""                                    # all diagnostics on stderr
    wp = pins["whole_program"]
    recs = []
    if ext == "sam":
        lines = open(out).read().splitlines()
        assert lines[0].startswith("@HD") and lines[1] == "@SQ\tSN:ref\tLN:71" and lines[2] == "@SQ\tSN:*extra_snippet(\")\tLN:8"
        for l in lines[3:]:
            f = l.split("\t")
            nm = [int(t[5:]) for t in f[11:] if t.startswith("NM:i:")]
            recs.append((f[0], int(f[1]), f[2], int(f[3]) - 1, f[5], nm[0] if nm else None, f[9], f[10]))
    else:
        data = gzip.open(out, "rb").read()
        l_text = struct.unpack_from("<i", data, 4)[0]
        off = 8 + l_text
        n_ref = struct.unpack_from("<i", data, off)[0]
        off += 4
        names = []
        for _ in range(n_ref):
            ln = struct.unpack_from("<i", data, off)[0]
            names.append(data[off + 4: off + 4 + ln - 1].decode())
            off += 4 + ln + 4
        while off < len(data):
            bs, ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiiBBHHHi", data, off)
            p = off + 36
            name = data[p: p + l_name - 1].decode()
            p += l_name
            cig = "".join(f"{w >> 4}{'MIDNSHP=X'[w & 15]}" for w in struct.unpack_from(f"<{n_cig}I", data, p))
            p += 4 * n_cig + (l_seq + 1) // 2 + l_seq
            nm = None
            if p < off + 4 + bs:
                assert data[p:p + 2] == b"NM"
                nm = data[p + 3]
            recs.append((name, flag, names[ref_id] if ref_id >= 0 else "*", pos, cig or "*", nm, None, None))
            off += 4 + bs
    assert {r[0] for r in recs} == set(wp["ids"])
    for name, flag, rname, pos, cig, nm, seq, qual in recs:
        if name in wp["unmapped"]:
            assert flag == 4 and rname == "*"
            continue
        assert rname == "ref" and not flag & 4
        for ename, erev, pmin, pmax, enm, ecig in wp["expect"]:
            if ename == name and erev == bool(flag & 16):
                assert pmin <= pos <= pmax and nm == enm and cig == ecig
        if seq is not None:
            if flag & 256:
                assert seq == "*" and qual == "*"            # secondary records carry no SEQ/QUAL (output.cpp:69-90)
            else:
                assert len(seq) == 12 and qual == "I" * 12


def _check_cigars(genome, reads, recs, rate):
    import re
    for ridx, flag, ref_id, pos, nm, cig in recs:
        if flag & 4:
            continue
        g = genome[ref_id]
        q = reads[ridx] if not flag & 16 else F.reverse_complement_rank(reads[ridx])
        i, j, cost = 0, pos, 0
        for ln, op in re.findall(r"(\d+)([=XID])", cig):
            ln = int(ln)
            if op == "=":
                assert (q[i:i + ln] == g[j:j + ln]).all(); i += ln; j += ln
            elif op == "X":
                assert (q[i:i + ln] != g[j:j + ln]).all(); i += ln; j += ln; cost += ln
            elif op == "I":
                i += ln; cost += ln
            else:
                j += ln; cost += ln
        assert i == len(q) and cost == nm and nm <= F.floating_point_error_aware_ceil(len(q) * rate)


@pytest.mark.parametrize("length,rate,n_reads", [(10000, 0.08, 12), (20000, 0.02, 8)])
def test_baseline_read_shapes(length, rate, n_reads):
    """BASELINE.json configs[2]/[4] read shapes (10 kb @ 8 %, 20 kb @ 2 %) on a small reference: CIGAR consistency, truth position,
    and record-for-record equality with the oracle."""
    genome = S.make_genome(600000, 2, seed=71)
    reads, names, truth = S.make_reads(genome, n_reads, length, rate, seed=72)
    idx = F.fmindex(genome)
    ctx = F.context(idx)
    res = F.aligner(ctx, F.params(error_probability=rate)).align_reads(reads)
    recs = res.records()
    _check_cigars(genome, reads, recs, rate)
    for i, (c, start, rev) in enumerate(truth):
        prim = [r for r in recs if r[0] == i and not r[1] & 256]
        assert len(prim) == 1 and not prim[0][1] & 4 and prim[0][2] == c and abs(prim[0][3] - start) <= 0.1 * length
        assert bool(prim[0][1] & 16) == rev
    exp = O.Index(genome).run(reads, O.params(error_probability=rate), threads=8)
    assert recs == exp.records()
    ctx.close()


def test_maximum_length_read_and_reference_edges():
    """a read close to the 100 000 bp limit (input.hpp:42), one read over it (skipped), and reads hanging over both reference
    ends (clipped windows)"""
    genome = S.make_genome(150000, 1, seed=81)
    g = genome[0]
    rng = np.random.default_rng(82)
    long_read = g[20000:20000 + 99000].copy()
    for p in rng.choice(len(long_read), size=600, replace=False):
        long_read[p] = long_read[p] % 4 + 1
    too_long = g[:100001].copy()
    head = np.concatenate([rng.integers(1, 5, size=40).astype(np.uint8), g[:1500]])          # overhangs the start
    tail = np.concatenate([g[-1500:], rng.integers(1, 5, size=40).astype(np.uint8)])          # overhangs the end
    reads = [long_read, too_long, head, tail]
    idx = F.fmindex(genome)
    ctx = F.context(idx)
    p = F.params(error_probability=0.01, interval_optimization=True)       # -I keeps the oracle's root alignments of the 99 kb read few
    res = F.aligner(ctx, p).align_reads(reads)
    assert res.skipped.tolist() == [0, 1, 0, 0]
    exp = O.Index(genome).run(reads, O.params(error_probability=0.01, interval_opt=True), threads=4)
    assert res.records() == exp.records()
    prim = [r for r in res.records() if r[0] == 0 and not r[1] & 256][0]
    assert not prim[1] & 4 and abs(prim[3] - 20000) <= 10
    p2 = F.params(error_probability=0.04)
    res2 = F.aligner(ctx, p2).align_reads([head, tail])
    exp2 = O.Index(genome).run([head, tail], O.params(error_probability=0.04), threads=2)
    assert res2.records() == exp2.records()
    assert all(not r[1] & 4 for r in res2.records())
    ctx.close()


def test_many_references_and_n_runs():
    """many short references (sentinel padding, locate across sequence boundaries), N runs in reads and reference"""
    rng = np.random.default_rng(91)
    refs = [rng.integers(1, 5, size=int(rng.integers(700, 4000))).astype(np.uint8) for _ in range(40)]
    refs[3][100:160] = 5
    reads = []
    for i in range(30):
        r = refs[int(rng.integers(0, len(refs)))]
        st = int(rng.integers(0, len(r) - 600))
        q = r[st:st + 600].copy()
        for pnt in rng.choice(600, size=12, replace=False):
            q[pnt] = q[pnt] % 4 + 1
        if i % 7 == 0:
            q[50:55] = 5
        if i % 2:
            q = F.reverse_complement_rank(q)
        reads.append(q)
    idx = F.fmindex(refs)
    ctx = F.context(idx)
    res = F.aligner(ctx, F.params(error_probability=0.05, interval_optimization=True)).align_reads(reads)
    exp = O.Index(refs).run(reads, O.params(error_probability=0.05, interval_opt=True), threads=4)
    assert res.records() == exp.records()
    assert sum(1 for r in res.records() if not r[1] & 4 and not r[1] & 256) >= 25
    ctx.close()


# ---------------------------------------------------------------- scheduling must not change results
def _run_with_env(genome, reads, env, **kw):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        idx = F.fmindex(genome)
        ctx = F.context(idx)                      # FLX_LANES is read here, the other variables per call
        res = F.aligner(ctx, F.params(error_probability=0.07, **kw)).align_reads(reads)
        out = (res.skipped.tolist(), res.records())
        ctx.close()
        return out
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("kw", [dict(), dict(interval_optimization=True)])
def test_lanes_chunks_and_launch_forms_give_the_same_records(kw):
    """one lane / several lanes with small chunks; alignment launches forced to the wave-slot-minimal or to the parallel form"""
    genome = S.make_genome(300000, 2, seed=31)
    reads, _, _ = S.make_reads(genome, 200, 1500, 0.07, seed=32)
    base = _run_with_env(genome, reads, {"FLX_LANES": "1"}, **kw)
    assert sum(1 for r in base[1] if not r[1] & 4) >= 150
    assert _run_with_env(genome, reads, {"FLX_LANES": "4", "FLX_CHUNK_READS": "16"}, **kw) == base
    assert _run_with_env(genome, reads, {"FLX_LANES": "3", "FLX_CHUNK_READS": "70"}, **kw) == base
    assert _run_with_env(genome, reads, {"FLX_LANES": "2", "FLX_NO_UNION": "1"}, **kw) == base          # every root window on its own
    assert _run_with_env(genome, reads, {"FLX_LANES": "2", "FLX_ALIGN_FEW_WAVES": "0"}, **kw) == base
    assert _run_with_env(genome, reads, {"FLX_LANES": "2", "FLX_ALIGN_FEW_WAVES": "1000000000"}, **kw) == base
    exp = O.Index(genome).run(reads[:40], O.params(error_probability=0.07, interval_opt=kw.get("interval_optimization", False)), threads=8)
    assert [r for r in base[1] if r[0] < 40] == exp.records()


def test_overlapping_calls_on_one_context():
    """compute calls from several host threads share the context's lanes; every call returns what it returns alone"""
    from concurrent.futures import ThreadPoolExecutor
    genome = S.make_genome(300000, 1, seed=41)
    batches = [S.make_reads(genome, 150, 1200, 0.06, seed=42 + b)[0] for b in range(4)]
    old = os.environ.get("FLX_LANES")
    os.environ["FLX_LANES"] = "3"
    try:
        idx = F.fmindex(genome)
        ctx = F.context(idx)
    finally:
        if old is None:
            os.environ.pop("FLX_LANES", None)
        else:
            os.environ["FLX_LANES"] = old
    al = F.aligner(ctx, F.params(error_probability=0.06))
    resident = [F.resident_reads(ctx, b) for b in batches]
    alone = [al.align_reads(r).records() for r in resident]
    with ThreadPoolExecutor(max_workers=4) as pool:
        together = [f.result().records() for f in [pool.submit(al.align_reads, r) for r in resident]]
        uploads = [f.result() for f in [pool.submit(F.resident_reads, ctx, b) for b in batches]]      # uploads while nothing else runs
        mixed = [f.result().records() for f in [pool.submit(al.align_reads, r) for r in uploads + resident]]
    assert together == alone
    assert mixed == alone + alone
    for r in resident + uploads:
        r.close()
    ctx.close()


def test_full_size_reads_both_launch_forms_match_oracle():
    """5 kb @ 8 % reads through the wave-slot-minimal launch form (what a full batch uses: five words per lane, eight lanes per
    root) and through the parallel form (what a small batch uses), both record-for-record equal to the oracle."""
    genome = S.make_genome(800000, 1, seed=51)
    reads, _, _ = S.make_reads(genome, 16, 5000, 0.08, seed=52)
    exp = O.Index(genome).run(reads, O.params(error_probability=0.08), threads=8)
    old = os.environ.get("FLX_ALIGN_FEW_WAVES")
    try:
        for few_waves in ("0", "1000000000"):
            os.environ["FLX_ALIGN_FEW_WAVES"] = few_waves
            ctx = F.context(F.fmindex(genome))
            res = F.aligner(ctx, F.params(error_probability=0.08)).align_reads(reads)
            assert res.skipped.tolist() == exp.skipped.tolist()
            assert res.records() == exp.records(), few_waves
            ctx.close()
    finally:
        if old is None:
            os.environ.pop("FLX_ALIGN_FEW_WAVES", None)
        else:
            os.environ["FLX_ALIGN_FEW_WAVES"] = old


def test_root_unions_and_their_fallback():
    """root windows of one locus share one DP over their union; a member whose path leaves its window is aligned on its own (the
    hook FLX_UNION_ALIGN_OWN sends every member that way): identical CIGAR slabs are shared, records equal the oracle's either way"""
    import subprocess, sys, json
    genome = S.make_genome(500000, 2, seed=61)
    reads, _, _ = S.make_reads(genome, 40, 3000, 0.08, seed=62)
    exp = O.Index(genome).run(reads, O.params(error_probability=0.08), threads=8)
    ctx = F.context(F.fmindex(genome))
    res = F.aligner(ctx, F.params(error_probability=0.08)).align_reads(reads)
    assert res.records() == exp.records()
    mapped = res.rows[(res.rows[:, 1] & 4) == 0]
    assert len(mapped) > 3 * len(reads)                                    # several records per read ...
    assert len(np.unique(mapped[:, 5])) < len(mapped) / 2                  # ... that share their CIGAR words
    ctx.close()
    # the hook is read once per process: run the forced variant in a child
    code = ("import sys, json; sys.path.insert(0, %r); import numpy as np, floxer_amd as F; from floxer_amd import simulate as S;"
            "g = S.make_genome(500000, 2, seed=61); r, _, _ = S.make_reads(g, 40, 3000, 0.08, seed=62);"
            "c = F.context(F.fmindex(g)); print(json.dumps(F.aligner(c, F.params(error_probability=0.08)).align_reads(r).records()))"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FLX_UNION_ALIGN_OWN="1"), capture_output=True, text=True, check=True)
    forced = [tuple(r) for r in json.loads(out.stdout.strip().split("\n")[-1])]
    assert forced == exp.records()


def test_existence_kernel_forms_give_the_same_records():
    """the existence tests run as a ring of lanes per job over the static band, 16 columns per step (ed_exists_block_kernel), by default;
    one column per step with FLX_EXISTS_STEPWISE=1 (ed_band_kernel); one lane per job with Ukkonen's cutoff with FLX_EXISTS_LANES=1
    (ed_exists_lane_kernel), also with the rounds queued without the host in between (FLX_ROUNDS_QUEUED=1). The switches of the ring forms
    are read once per process, so the other forms run in children: same records, all equal to the oracle's, on reads whose trees reach
    every launch shape of the lower levels and the ring-scheduled upper ones"""
    import subprocess, sys, json
    genome = S.make_genome(500000, 2, seed=161)
    reads, _, _ = S.make_reads(genome, 24, 6000, 0.08, seed=162)
    exp = O.Index(genome).run(reads, O.params(error_probability=0.08), threads=8)
    ctx = F.context(F.fmindex(genome))
    assert F.aligner(ctx, F.params(error_probability=0.08)).align_reads(reads).records() == exp.records()
    ctx.close()
    code = ("import sys, json; sys.path.insert(0, %r); import floxer_amd as F; from floxer_amd import simulate as S;"
            "g = S.make_genome(500000, 2, seed=161); r, _, _ = S.make_reads(g, 24, 6000, 0.08, seed=162);"
            "c = F.context(F.fmindex(g)); print(json.dumps(F.aligner(c, F.params(error_probability=0.08)).align_reads(r).records()))"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for env in ({"FLX_EXISTS_STEPWISE": "1"}, {"FLX_EXISTS_LANES": "1"}, {"FLX_EXISTS_LANES": "1", "FLX_ROUNDS_QUEUED": "1"},
                {"FLX_EXISTS_LANES": "1", "FLX_EXISTS_TEAM": "1"}, {"FLX_EXISTS_LANES": "1", "FLX_EXISTS_TEAM": "8"}):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, check=True)
        assert [tuple(r) for r in json.loads(out.stdout.strip().split("\n")[-1])] == exp.records(), env


@pytest.mark.parametrize("team", [None, "2", "4", "8", "16"])
def test_existence_with_cutoff_matches_oracle(small_genome, monkeypatch, team):
    """ed_exists_lane_kernel / ed_exists_team_kernel<P> (FLX_EXISTS_LANES=1; team: P lanes per job, each on every P-th word group one block
    behind the lane of the group above - also with more lanes than a job has groups) against the oracle's Myers on what the cutoff has to get right: several occurrences in one window (tandem
    repeats with periods around the group height, a second occurrence far to the right of the first), budgets from 0 to more than the
    query is long, low-complexity sequence (every diagonal alive), occurrences at either edge of the window, windows shorter than the
    query, N runs; ragged sizes across word-group boundaries in one launch"""
    _, _, ctx, _ = small_genome
    monkeypatch.setenv("FLX_EXISTS_LANES", "1")
    if team:
        monkeypatch.setenv("FLX_EXISTS_TEAM", team)
    rng = np.random.default_rng(77)
    refs, queries, jobs = [], [], []
    ro = qo = 0

    def add(ref, q, k):
        nonlocal ro, qo
        ref = np.asarray(ref, np.uint8)
        q = np.asarray(q, np.uint8)
        jobs.append((ro, len(ref), qo, len(q), int(k), 0))
        refs.append(ref)
        queries.append(q)
        ro += len(ref)
        qo += len(q)

    for m in (1, 5, 63, 64, 65, 128, 130, 190, 257, 400, 640, 1000, 1500, 2300):
        for err in (0.0, 0.05, 0.12, 0.3):
            ref, q = _rand_align_case(rng, m, err, int(rng.integers(0, m // 2 + 40)))
            for k in sorted({0, int(m * err * 0.6), int(m * err) + 1, int(m * 0.1) + 2, int(m * 0.3) + 1, m + 3}):
                add(ref, q, k)
        # tandem repeats: the query is a few copies of a unit, the window many more of them (an occurrence every `period` columns)
        for period in (7, 50, 64, 100, 150, 333):
            unit = rng.integers(1, 5, size=period).astype(np.uint8)
            q = np.tile(unit, m // period + 2)[:m]
            mut = q.copy()
            flip = rng.random(m) < 0.04
            mut[flip] = (mut[flip] % 4 + 1).astype(np.uint8)
            ref = np.tile(unit, (2 * m) // period + 6)
            add(ref, mut, int(m * 0.06) + 1)
            add(ref, mut, int(m * 0.02))
        # two occurrences far apart, the better one second; and one at either edge
        ref1, q = _rand_align_case(rng, m, 0.10, 0)
        ref2, _ = _rand_align_case(rng, m, 0.0, 0)
        ref2 = np.array([c for c in q], np.uint8)          # exact copy
        gap = rng.integers(1, 5, size=3 * m + 50).astype(np.uint8)
        add(np.concatenate([ref1, gap, ref2]), q, int(m * 0.12) + 1)
        add(np.concatenate([ref2, gap, ref1]), q, int(m * 0.12) + 1)
        add(np.concatenate([gap, ref2]), q, 2)
        add(np.concatenate([ref2, gap]), q, 2)
        # low complexity and N runs
        add((rng.integers(0, 2, size=m + 60) + 1).astype(np.uint8), (rng.integers(0, 2, size=m) + 1).astype(np.uint8), int(m * 0.2) + 1)
        rn, qn = _rand_align_case(rng, m, 0.05, 20)
        rn = rn.copy()
        rn[len(rn) // 3: len(rn) // 3 + max(1, m // 10)] = 5
        add(rn, qn, int(m * 0.15) + 2)
        # window shorter than the query
        add(q[: max(1, m - 5)], q, 3)
        add(q[: max(1, m - 5)], q, 6)
    rpool, qpool = np.concatenate(refs), np.concatenate(queries)
    got = F.align_batch(ctx, qpool, jobs, reference_pool=rpool)
    n_yes = 0
    for (ro_, rl, qo_, ql, k, mode), g in zip(jobs, got):
        exp = O.align(rpool[ro_:ro_ + rl], qpool[qo_:qo_ + ql], k, mode=0, algo=1)
        if exp is not None:
            exp = (exp[0], 0, "")
            n_yes += 1
        assert g == exp, (ql, rl, k)
    assert 0.2 * len(jobs) < n_yes < 0.95 * len(jobs)
    # the same jobs ask for the end positions too (the cutoff keeps the rightmost minimum of the last row)
    jobs1 = [j[:5] + (1,) for j in jobs]
    got1 = F.align_batch(ctx, qpool, jobs1, reference_pool=rpool)
    for (ro_, rl, qo_, ql, k, mode), g in zip(jobs1, got1):
        exp = O.align(rpool[ro_:ro_ + rl], qpool[qo_:qo_ + ql], k, mode=1, algo=1)
        if exp is not None:
            exp = (exp[0], exp[1], "")
        assert g == exp, (ql, rl, k)


def test_caller_owned_stream():
    """flx_ctx_set_stream: every launch goes to the caller's HIP stream (one lane); results do not change"""
    import ctypes
    capi.lib()
    hip = ctypes.CDLL(capi.hip_runtime_paths()[0])          # the HIP runtime this process already has (capi._share_torchs_hip_runtime), not a second one
    genome = S.make_genome(200000, 1, seed=71)
    reads, _, _ = S.make_reads(genome, 60, 1500, 0.06, seed=72)
    ctx = F.context(F.fmindex(genome))
    al = F.aligner(ctx, F.params(error_probability=0.06))
    base = al.align_reads(reads).records()
    stream = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(stream)) == 0
    ctx.set_stream(stream)
    rr = F.resident_reads(ctx, reads)
    assert al.align_reads(rr).records() == base
    assert al.align_reads(reads).records() == base
    assert hip.hipStreamSynchronize(stream) == 0
    ctx.set_stream(None)
    assert al.align_reads(rr).records() == base
    rr.close()
    ctx.close()
    assert hip.hipStreamDestroy(stream) == 0


@pytest.mark.parametrize("kw,okw", [(dict(), dict()), (dict(interval_optimization=True), dict(interval_opt=True))])
def test_repeats_and_several_references_match_oracle(kw, okw):
    """five references with copied segments: seeds with many hits (hard and soft cap, the device-side selection and its host
    remainder, std::sort on more than 16 equal keys), loci that share windows, anchors on several references"""
    genome = S.make_genome(1_000_000, 5, seed=101)
    rng = np.random.default_rng(7)
    for g in genome:
        for _ in range(6):
            a, b, ln = rng.integers(0, len(g) - 5000), rng.integers(0, len(g) - 5000), int(rng.integers(500, 4000))
            g[b:b + ln] = g[a:a + ln]
    ctx = F.context(F.fmindex(genome))
    oidx = O.Index(genome)
    for length, rate, seed in [(1500, 0.05, 1), (3000, 0.08, 2), (600, 0.10, 3)]:
        reads, _, _ = S.make_reads(genome, 250, length, rate, seed=300 + seed)
        got = F.aligner(ctx, F.params(error_probability=rate, **kw)).align_reads(reads)
        exp = oidx.run(reads, O.params(error_probability=rate, **okw), threads=8)
        assert got.skipped.tolist() == exp.skipped.tolist()
        assert got.records() == exp.records(), (length, rate)
    ctx.close()


def test_index_built_on_the_device_is_the_host_index(tmp_path):
    """flx_index_build_on_device: suffix arrays by prefix doubling on the GPU; the index (SA, both BWTs) equals the host-built one,
    for references with repeats, N runs, several sequences and a sequence that is a prefix of another"""
    rng = np.random.default_rng(81)
    base = rng.integers(1, 5, size=30000, dtype=np.uint8)
    refs = [base.copy(), base[:12000].copy(), np.concatenate([base[5000:9000], base[5000:9000], base[5000:9000]]),
            np.concatenate([np.full(700, 5, np.uint8), rng.integers(1, 5, size=4001, dtype=np.uint8), np.full(300, 1, np.uint8)]),
            np.array([3], np.uint8)]
    host = F.fmindex(refs)
    dev = F.fmindex(refs, device=0)
    n = capi.lib().flx_index_text_length(host.h)
    assert capi.lib().flx_index_text_length(dev.h) == n

    def arrays(ix):
        sa = np.zeros(n, dtype=np.uint64)
        capi.check(capi.lib().flx_index_copy_sa(ix.h, capi.ptr(sa, capi.u64p)))
        bwts = []
        for rev in (0, 1):
            b = np.zeros(n, dtype=np.uint8)
            capi.check(capi.lib().flx_index_copy_bwt(ix.h, rev, capi.ptr(b, capi.u8p)))
            bwts.append(b)
        return sa, bwts
    sa_h, bwt_h = arrays(host)
    sa_d, bwt_d = arrays(dev)
    assert (sa_h == sa_d).all()
    assert (bwt_h[0] == bwt_d[0]).all() and (bwt_h[1] == bwt_d[1]).all()
    p1, p2 = str(tmp_path / "h.idx"), str(tmp_path / "d.idx")
    host.save(p1); dev.save(p2)
    assert open(p1, "rb").read() == open(p2, "rb").read()


def _padded_text(refs):
    """the text the index is over: every sequence followed by 4 - (len % 4) delimiters (BiFMIndex's layout, floxer.cpp:93-97)"""
    parts = []
    for r in refs:
        parts.append(np.asarray(r, dtype=np.uint8))
        parts.append(np.zeros(4 - len(r) % 4, np.uint8))
    return np.concatenate(parts)


def test_index_at_scale_checked_without_the_products_arrays():
    """250 Mb in 5 sequences, index built on the device. Nothing of the product checks it here: the suffix array is a permutation,
    2 M sampled neighbouring rows are in suffix order (texts compared with numpy, the end of the text sorting first), the BWT is the
    symbol in front of every suffix (all rows), and the BWT of the reversed text has the text's symbol counts and every sampled LF walk
    over it spells a string that occurs in the text (found with the suffix array just checked)."""
    chrom = 50_000_000
    pool, genome = S.make_genome_fast(chrom, 5, seed=4242)
    idx = F.fmindex(genome, device=0)
    text = _padded_text(genome)
    n = len(text)
    assert idx.text_length == n
    sa = idx.suffix_array_u32()
    seen = np.zeros(n, dtype=bool)
    seen[sa] = True
    assert seen.all()                                                         # a permutation of 0 .. n-1
    del seen
    rng = np.random.default_rng(99)

    def suffix_cmp(a, b):
        """-1 / 0 / +1 per pair: text[a:] against text[b:], a suffix that ends first being the smaller one"""
        res = np.zeros(len(a), dtype=np.int8)
        todo = np.arange(len(a))
        a, b = a.astype(np.int64), b.astype(np.int64)
        for d in range(0, 1 << 20):
            if len(todo) == 0:
                break
            pa, pb = a[todo] + d, b[todo] + d
            ea, eb = pa >= n, pb >= n
            ca = np.where(ea, -1, text[np.minimum(pa, n - 1)].astype(np.int16))
            cb = np.where(eb, -1, text[np.minimum(pb, n - 1)].astype(np.int16))
            diff = ca != cb
            res[todo[diff]] = np.where(ca[diff] < cb[diff], -1, 1)
            todo = todo[~diff & ~(ea & eb)]
        return res

    rows = rng.integers(0, n - 1, size=2_000_000)
    assert (suffix_cmp(sa[rows], sa[rows + 1]) < 0).all()
    # rows around the delimiters too (suffixes that start with runs of the delimiter sort first)
    assert (suffix_cmp(sa[:5000], sa[1:5001]) < 0).all()
    bwt = idx.bwt(False)
    assert (bwt == text[(sa.astype(np.int64) - 1) % n]).all()
    del bwt
    # ---- the BWT of the reversed text
    bwt_r = idx.bwt(True)
    counts = np.bincount(text, minlength=6)
    assert (np.bincount(bwt_r, minlength=6) == counts).all()
    C = np.concatenate([[0], np.cumsum(counts)])[:6]
    step = 256
    ck = np.zeros((n // step + 1, 6), dtype=np.int64)                       # symbol counts in front of every 256th row
    for c in range(6):
        ck[1:, c] = np.cumsum(np.add.reduceat((bwt_r == c).astype(np.int32), np.arange(0, n, step)))[: n // step]

    def rank(c, i):                                                          # occurrences of c[k] in bwt_r[0, i[k])
        base = ck[i // step, c]
        out = base.copy()
        for d in range(step):
            at = (i // step) * step + d
            live = at < i
            out += (live & (bwt_r[np.minimum(at, n - 1)] == c)).astype(np.int64)
        return out

    walkers = rng.integers(0, n, size=400)
    spelled = []
    for _ in range(24):
        c = bwt_r[walkers]
        spelled.append(c)
        walkers = C[c] + rank(c.astype(np.int64), walkers)
    # an LF step on the reversed text's BWT prepends a symbol of the reversed text = appends one of the text: the walk spells text forwards
    strings = np.stack(spelled, axis=1)
    sa64 = sa.astype(np.int64)
    for s_ in strings:
        if (s_ == 0).any():
            continue                                                         # (walks through a delimiter wrap around the text's end)
        lo, hi = 0, n
        for d, c in enumerate(s_):                                           # narrow [lo, hi) by the d-th symbol with binary searches on the suffix array
            def sym(row):
                p = sa64[row] + d
                return -1 if p >= n else int(text[p])
            l, h = lo, hi
            while l < h:
                m = (l + h) // 2
                if sym(m) < c: l = m + 1
                else: h = m
            first = l
            h = hi
            while l < h:
                m = (l + h) // 2
                if sym(m) <= c: l = m + 1
                else: h = m
            lo, hi = first, l
            assert lo < hi, "an LF walk over the reverse BWT spelled a string the text does not hold"


# ---------------------------------------------------------------- '$' in a read, repeat-rich text, scale
def test_dollar_in_reads_and_seeds_matches_oracle(small_genome):
    """input.cpp:165-176 maps '$' to rank 0, the rank of the sequence delimiters: search_ng21 extends a cursor with it like with any
    other query symbol (it can only ever be a match child: the mismatch loops run over symbols 1..5), the aligner compares it like
    any other rank. K1 emission and the whole path on reads holding rank 0, several references (so delimiters exist in the text)."""
    refs, idx, ctx, oidx = small_genome
    rng = np.random.default_rng(5)
    pool, seeds = _make_seeds(rng, refs, 120, kmax=2, lens=(10, 40))
    pool = pool.copy()
    for off, ln, k, i in seeds[::3]:
        pool[off + int(rng.integers(0, ln))] = 0                     # a '$' somewhere in every third seed
    # seeds that run over the end of a reference into its delimiter: 'xxxx$' really occurs in the text
    extra = []
    for r in refs[:2]:
        tail = np.concatenate([r[-14:], np.zeros(1, np.uint8)])
        extra.append((len(pool), len(tail), 1, len(seeds) + len(extra)))
        pool = np.concatenate([pool, tail])
    seeds = seeds + extra
    got = F.searcher(ctx).search_groups(pool, seeds, max_hits=501)
    n_with_dollar_hits = 0
    for si, (off, ln, k, _) in enumerate(seeds):
        exp, _ = oidx.search_groups(pool[off:off + ln], k, 501)
        mine = got[got[:, 0] == si][:, 1:]
        assert mine.tolist() == exp.tolist(), (si, ln, k)
        n_with_dollar_hits += int(len(exp) > 0 and (pool[off:off + ln] == 0).any())
    assert n_with_dollar_hits >= 2                                   # the delimiter seeds are found
    anchors, stats = F.searcher(ctx).search_seeds(pool, seeds)
    exp_a, exp_s = oidx.search_seeds(pool, seeds)
    assert anchors.tolist() == exp_a.tolist() and stats.tolist() == exp_s.tolist()
    # whole path: reads with a '$' inside
    genome = S.make_genome(150000, 3, seed=31)
    reads, _, _ = S.make_reads(genome, 40, 900, 0.06, seed=32)
    for i in range(0, 40, 4):
        reads[i] = reads[i].copy()
        reads[i][int(rng.integers(0, len(reads[i])))] = 0
    ctx2 = F.context(F.fmindex(genome))
    for kw, okw in [(dict(), dict()), (dict(interval_optimization=True), dict(interval_opt=True)), (dict(without_cigar=True), dict(without_cigar=True))]:
        got_r = F.aligner(ctx2, F.params(error_probability=0.06, **kw)).align_reads(reads)
        exp_r = O.Index(genome).run(reads, O.params(error_probability=0.06, **okw), threads=8)
        assert got_r.records() == exp_r.records(), kw
    ctx2.close()


def _repeat_rich_reference(rng, n_bases):
    """tandem repeats (units of 1..60 bp in arrays of up to 6 kb), a 300-bp family with thousands of diverged copies, low-complexity
    stretches and runs of N between unique sequence"""
    family = rng.integers(1, 5, size=300).astype(np.uint8)
    parts, total = [], 0
    while total < n_bases:
        kind = int(rng.integers(0, 10))
        if kind < 3:                                               # a family copy, 1..12 % diverged (mismatches and small indels)
            c = family.copy()
            n_mut = int(rng.integers(3, 36))
            for _ in range(n_mut):
                p = int(rng.integers(0, len(c)))
                t = int(rng.integers(0, 3))
                if t == 0:
                    c[p] = rng.integers(1, 5)
                elif t == 1:
                    c = np.delete(c, p)
                else:
                    c = np.insert(c, p, rng.integers(1, 5))
            piece = c
        elif kind < 5:                                             # tandem repeat
            unit = rng.integers(1, 5, size=int(rng.integers(1, 61))).astype(np.uint8)
            piece = np.tile(unit, int(rng.integers(200, 6000)) // len(unit) + 1)
        elif kind == 5:                                            # run of N
            piece = np.full(int(rng.integers(5, 400)), 5, np.uint8)
        elif kind == 6:                                            # two-letter low complexity
            piece = rng.choice(np.array([1, 4], np.uint8), size=int(rng.integers(100, 1500)))
        else:                                                      # unique sequence
            piece = rng.integers(1, 5, size=int(rng.integers(300, 4000))).astype(np.uint8)
        parts.append(piece.astype(np.uint8))
        total += len(piece)
    return np.concatenate(parts)[:n_bases]


@pytest.mark.parametrize("kw,okw", [(dict(), dict()), (dict(interval_optimization=True), dict(interval_opt=True))])
def test_repeat_rich_reference_matches_oracle(kw, okw, capsys):
    """hg38 is repeat-rich: a text where a large share of the seeds runs into the soft cap (50 rows kept) or the hard cap (seed
    excluded over 500 rows) and loci repeat. Records equal the oracle's, default flags and -I; the selection stays on the device
    (the share of seeds the host selected for is printed, and kept in profiles/ by the round's log: 0)."""
    rng = np.random.default_rng(2024)
    genome = [_repeat_rich_reference(rng, 2_500_000), _repeat_rich_reference(rng, 1_500_000)]
    fidx = F.fmindex(genome)
    ctx = F.context(fidx)
    oidx = O.Index(genome, imported=(fidx.suffix_array_u32(), fidx.bwt(False), fidx.bwt(True)))   # (oracle SA == product SA: test_host_cpu)
    lines = []
    for length, rate, n_reads, seed in [(5000, 0.08, 60, 1), (2000, 0.05, 120, 2)]:
        reads, _, _ = S.make_reads(genome, n_reads, length, rate, seed=900 + seed)
        ctx.path_counters(reset=True)
        got = F.aligner(ctx, F.params(error_probability=rate, **kw)).align_reads(reads)
        pc = ctx.path_counters()
        exp = oidx.run(reads, O.params(error_probability=rate, **okw), threads=8)
        assert got.skipped.tolist() == exp.skipped.tolist()
        assert got.records() == exp.records(), (length, rate)
        lines.append(f"repeat-rich {length} bp @ {rate:.0%} {kw or 'defaults'}: seeds {pc['seeds']}, with anchors {pc['seeds_with_anchors']}, excluded by the hard cap "
                     f"{pc['seeds_excluded_by_hard_cap']} ({pc['seeds_excluded_by_hard_cap'] / pc['seeds']:.1%}), selected on the host "
                     f"{pc['seeds_selected_on_host']} ({pc['seeds_selected_on_host'] / pc['seeds']:.2%}), anchors {pc['anchors']}, records {pc['records']}")
        # the caps really bite here, and at the default caps every seed is selected on the device (heavy seeds: one wave each)
        assert pc["seeds_excluded_by_hard_cap"] > 0.02 * pc["seeds"] and pc["seeds_selected_on_host"] == 0
    with capsys.disabled():
        print("\n" + "\n".join(lines))
    ctx.close()


def test_scale_250mb_multi_sequence():
    """BASELINE.json configs[2] size: 250 Mb in 5 sequences, index built on the GPU, 1024 reads of 10 kb @ 8 % (floxer defaults).
    All reads: one primary at the simulated sequence / position / strand; CIGARs of a sample consistent with the texts; and
    record-for-record equality with the oracle (its index laid out around the imported suffix array) on a sample of the reads."""
    G, NSEQ, NR, L, rate = 250_000_000, 5, 1024, 10000, 0.08
    pool, genome = S.make_genome_fast(G // NSEQ, NSEQ, seed=77)
    (rpool, offs), (chrom, pos, rev) = S.make_reads_fast(pool, [G // NSEQ] * NSEQ, NR, L, rate, seed=78)
    fidx = F.fmindex(genome, device=0)
    ctx = F.context(fidx)
    res = F.aligner(ctx, F.params(error_probability=rate)).align_reads((rpool, offs))
    rows = res.rows
    prim = rows[(rows[:, 1] & 256) == 0]
    assert len(prim) == NR and (prim[:, 0] == np.arange(NR)).all()
    assert ((prim[:, 1] & 4) == 0).all()
    assert (prim[:, 2] == chrom).all() and (np.abs(prim[:, 3] - pos.astype(np.int64)) <= 0.1 * L).all()
    assert (((prim[:, 1] & 16) != 0) == (rev != 0)).all()
    sample = list(range(0, NR, 32))
    reads = [rpool[int(offs[i]):int(offs[i + 1])] for i in sample]
    sub = F.aligner(ctx, F.params(error_probability=rate)).align_reads(reads)
    # the records of a read do not depend on its batch (compared as arrays: 45 k records as strings would take minutes)
    def recs_of(result, read):
        out = []
        for r in result.rows[result.rows[:, 0] == read]:
            out.append((int(r[1]), int(r[2]), int(r[3]), int(r[4]), result.cigars[r[5]: r[5] + r[6]].tobytes()))
        return out
    for j, i in enumerate(sample):
        assert recs_of(sub, j) == recs_of(res, i)
    sub_recs = sub.records()
    _check_cigars(genome, reads, sub_recs, rate)
    oidx = O.Index(genome, imported=(fidx.suffix_array_u32(), fidx.bwt(False), fidx.bwt(True)), pool=pool)
    exp = oidx.run(reads, O.params(error_probability=rate), threads=16)
    assert sub_recs == exp.records()
    ctx.close()


def test_scale_250mb_repeat_rich_matches_oracle(capsys):
    """A repeat-rich reference at BASELINE.json configs[2] size (250 Mb in 5 sequences, flx_sim_genome_repeats: interspersed families,
    tandem repeats, low complexity, N runs, segmental duplications - the bench's --repeat-rich generator): 512 reads of 10 kb @ 8 %,
    floxer defaults. Here the search is the tail of its heaviest seeds and the lanes of a wave share their subtrees
    (fm_search_filter_kernel); the records must not notice: a read's records do not depend on its batch (another batch = other seeds
    next to it on a wave = other hand-overs), and a sample of reads equals the oracle record for record, defaults and -I."""
    G, NSEQ, NR, L, rate = 250_000_000, 5, 512, 10000, 0.08
    pool, genome = S.make_genome_fast(G // NSEQ, NSEQ, seed=177, repeat_rich=True)
    (rpool, offs), (chrom, pos, rev) = S.make_reads_fast(pool, [G // NSEQ] * NSEQ, NR, L, rate, seed=178)
    fidx = F.fmindex(genome, device=0)
    ctx = F.context(fidx)
    ctx.path_counters(reset=True)
    res = F.aligner(ctx, F.params(error_probability=rate)).align_reads((rpool, offs))
    pc = ctx.path_counters()
    sample = list(range(0, NR, 16))
    reads = [rpool[int(offs[i]):int(offs[i + 1])] for i in sample]
    sub = F.aligner(ctx, F.params(error_probability=rate)).align_reads(reads)
    for j, i in enumerate(sample):
        assert _recs_of(sub, j) == _recs_of(res, i)
    oidx = O.Index(genome, imported=(fidx.suffix_array_u32(), fidx.bwt(False), fidx.bwt(True)), pool=pool)
    exp = oidx.run(reads, O.params(error_probability=rate), threads=16)
    assert sub.skipped.tolist() == exp.skipped.tolist()
    assert np.array_equal(sub.rows[:, :5], exp.rows[:, :5]) and sub.records() == exp.records()
    sub_i = F.aligner(ctx, F.params(error_probability=rate, interval_optimization=True)).align_reads(reads)
    exp_i = oidx.run(reads, O.params(error_probability=rate, interval_opt=True), threads=16)
    assert sub_i.records() == exp_i.records()
    with capsys.disabled():
        print(f"\nrepeat-rich 250 Mb: seeds {pc['seeds']}, excluded by the hard cap {pc['seeds_excluded_by_hard_cap']} "
              f"({pc['seeds_excluded_by_hard_cap'] / pc['seeds']:.1%}), selected on the host {pc['seeds_selected_on_host']}, rank pairs per read "
              f"{pc['cursor_extensions'] / NR:.0f}, search reruns {pc['search_reruns']}, records {pc['records']}")
    assert pc["seeds_excluded_by_hard_cap"] > 0.01 * pc["seeds"]
    ctx.close()


def test_cli_devices_stats_fastq_forms_and_accuracy(tmp_path):
    """The drop-in CLI beyond the reference's own test: (1) a FASTQ as plain text, gzip, with CR LF line ends, without a final
    line end, and through a named pipe (plain and gzip) gives the same SAM; (2) two contexts (--devices 0,0: batches dealt in turn, written in input order) and several I/O
    threads give the same records as one; SAM and BAM hold the same records; (3) --stats writes the reference's TOML (query count,
    histogram totals consistent with the records); (4) --index is saved, reused, and refused for another reference;
    (5) simulated_dataset verify finds every read of the simulated set at its origin."""
    import gzip
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe, sim = os.path.join(root, "floxer_amd", "floxer"), os.path.join(root, "floxer_amd", "simulated_dataset")
    fa, fq = str(tmp_path / "g.fasta"), str(tmp_path / "r.fastq")
    subprocess.run([sim, "create", "--genomes", fa, "--reads", fq, "-c", "200000", "-n", "3", "-l", "2000", "-m", "300", "-e", "0.06", "-s", "5", "--revcomp-fraction", "0.5"], check=True)
    text = open(fq).read()
    forms = {"plain": fq, "gz": str(tmp_path / "r2.fastq.gz"), "crlf": str(tmp_path / "r3.fastq"), "noeol": str(tmp_path / "r4.fastq")}
    with gzip.open(forms["gz"], "wt") as f:
        f.write(text)
    open(forms["crlf"], "w", newline="").write(text.replace("\n", "\r\n"))
    open(forms["noeol"], "w").write(text.rstrip("\n"))

    def run(queries, out, *extra, env=None):
        cmd = [exe, "--reference", fa, "--queries", queries, "--output", out, "--error-probability", "0.06", *extra]
        e = dict(os.environ, FLX_BATCH_READS="64", **(env or {}))       # several batches even for 300 reads
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=e)
        assert r.stdout == b""
        return r

    def sam_records(path):
        return [l for l in open(path).read().splitlines() if not l.startswith("@")]

    base_out = str(tmp_path / "base.sam")
    r = run(fq, base_out, "--threads", "1")
    assert r.returncode == 0, r.stderr.decode()
    base = sam_records(base_out)
    assert len(base) >= 300
    for name, path in forms.items():
        if name == "plain":
            continue
        o = str(tmp_path / f"{name}.sam")
        r = run(path, o, "--threads", "3")
        assert r.returncode == 0, (name, r.stderr.decode())
        assert sam_records(o) == base, name

    # the queries through a named pipe, plain and gzip (the reference reads a FIFO named *.fastq through its ifstream just as well)
    import threading
    for name in ("plain", "gz"):
        fifo = str(tmp_path / (f"pipe_{name}.fastq" + (".gz" if name == "gz" else "")))
        os.mkfifo(fifo)
        feeder = threading.Thread(target=lambda src=forms[name], dst=fifo: open(dst, "wb").write(open(src, "rb").read()))
        feeder.start()
        o = str(tmp_path / f"pipe_{name}.sam")
        r = run(fifo, o, "--threads", "3")
        feeder.join()
        assert r.returncode == 0, (name, r.stderr.decode())
        assert sam_records(o) == base, f"fifo {name}"

    # two contexts on the one GPU, five I/O threads, statistics, index saved
    o2, toml, idx = str(tmp_path / "two.sam"), str(tmp_path / "stats.toml"), str(tmp_path / "g.index")
    r = run(fq, o2, "--devices", "0,0", "--threads", "5", "--stats", toml, "--stats-input-hint", "simulated", "--index", idx)
    assert r.returncode == 0, r.stderr.decode()
    assert sam_records(o2) == base
    assert b"2 HIP device contexts" in r.stderr
    st = open(toml).read()
    sect = {}
    cur = None
    for l in st.splitlines():
        if l.startswith("["):
            cur = l[1:-1]; sect[cur] = {}
        elif cur and " = " in l:
            k, v = l.split(" = ", 1); sect[cur][k] = v
    assert st.splitlines()[0].startswith("completely_excluded_queries = ")
    assert sect["query_lengths"]["num_values"] == "300" and sect["alignments_per_query"]["num_values"] == "300"
    n_mapped = sum(1 for l in base if not int(l.split("\t")[1]) & 4)
    assert int(sect["alignments_edit_distance"]["num_values"]) == n_mapped
    assert int(sect["reference_span_sizes_aligned_of_roots"]["num_values"]) >= n_mapped
    assert int(sect["seeds_per_query"]["min_value"]) > 0 and "mean" in sect["seed_lengths"]

    # BAM holds the same records; the saved index is loaded (no build) and gives the same output
    ob = str(tmp_path / "two.bam")
    r = run(fq, ob, "--index", idx, "--threads", "4")
    assert r.returncode == 0 and b"loading index" in r.stderr and b"building index" not in r.stderr
    v_sam = subprocess.run([sim, "verify", "--alignments", base_out, "-p", "120"], check=True, capture_output=True, text=True).stdout
    v_bam = subprocess.run([sim, "verify", "--alignments", ob, "-p", "120"], check=True, capture_output=True, text=True).stdout
    assert v_sam == v_bam
    assert v_sam.count("FoundOptimal = {}") == 300                                        # every simulated read at its origin

    # the index of one reference is refused for another
    fa2, fq2 = str(tmp_path / "h.fasta"), str(tmp_path / "h.fastq")
    subprocess.run([sim, "create", "--genomes", fa2, "--reads", fq2, "-c", "200000", "-n", "3", "-l", "2000", "-m", "4", "-e", "0.06", "-s", "6"], check=True)
    cmd = [exe, "--reference", fa2, "--queries", fq2, "--output", str(tmp_path / "x.sam"), "--error-probability", "0.06", "--index", idx]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"was not built from the reference" in r.stderr

    # a malformed query file is an error, not a silent truncation
    bad = str(tmp_path / "bad.fastq")
    open(bad, "w").write("\n".join(text.splitlines()[:-1]) + "\n")                        # the last record lacks its quality line
    r = run(bad, str(tmp_path / "bad.sam"))
    assert r.returncode != 0


def test_empty_batch_is_an_empty_run(small_genome):
    refs, idx, ctx, oidx = small_genome
    for kw in (dict(), dict(without_cigar=True)):
        res = F.aligner(ctx, F.params(error_probability=0.05, **kw)).align_reads([])
        assert res.n_records == 0 and len(res.cigars) == 0
    rr = F.resident_reads(ctx, [])
    assert F.aligner(ctx, F.params(error_probability=0.05)).align_reads(rr).n_records == 0
    rr.close()


def test_context_on_an_index_image_and_two_contexts():
    """flx_ctx_create_on_image: the index's HBM image in caller-owned device buffers (what floxer_amd.distributed.replicate_index
    hands to every rank after the RCCL broadcast), with the full index and with the array-less index made from its meta block;
    two contexts on one image at once (the CLI's --devices / a rank's second context): identical records, also with --without-cigar
    (the reversed text then comes from HBM)"""
    import torch
    genome = S.make_genome(300000, 3, seed=51)
    reads, _, _ = S.make_reads(genome, 80, 1500, 0.06, seed=52)
    idx = F.fmindex(genome, device=0)
    base_ctx = F.context(idx)
    light = F.fmindex.from_meta(idx.meta())
    image = [torch.empty(n, dtype=torch.uint8, device="cuda:0") for n in idx.image_layout()]
    idx.image_upload(0, [b.data_ptr() for b in image])
    a, b = F.context(idx, image=image), F.context(light, image=image)
    for kw in (dict(), dict(without_cigar=True), dict(interval_optimization=True)):
        p = F.params(error_probability=0.06, **kw)
        base = F.aligner(base_ctx, p).align_reads(reads).records()
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(2) as pool:
            fa, fb = pool.submit(F.aligner(a, p).align_reads, reads), pool.submit(F.aligner(b, p).align_reads, reads)
            assert fa.result().records() == base and fb.result().records() == base
    with pytest.raises(F.FloxerError):
        F.context(light)                                   # an index without arrays cannot upload itself
    with pytest.raises(F.FloxerError):                     # buffers that are not this index's layout (a stale image) are refused
        F.context(idx, image=image[:4] + [torch.empty(int(image[4].numel()) + 64, dtype=torch.uint8, device="cuda:0")])
    for c in (a, b, base_ctx):
        c.close()


def test_torch_started_after_the_library_has_worked_finds_the_gpu():
    """the order a test run of this file alone has (the library works, torch comes later): one HIP runtime in the process, so
    torch's initialisation finds the GPU (scripts/torch_after_lib.py; a child process, so that the order is this test's)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "torch_after_lib.py"), "G"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "torch after the library: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_device_rounds_equal_host_rounds():
    """The inner PEX levels with the anchors' state on the device (requests, de-duplication, clusters, moves up the trees as kernels)
    against the host form of the same rounds (taken when a statistics object is attached: it wants every request's window), on a
    text with copied segments (loci that share windows, clusters that are not decided by their intersection / union), several
    references, three read shapes, defaults / -I / -d / bottom-up trees"""
    genome = S.make_genome(400_000, 4, seed=61)
    rng = np.random.default_rng(9)
    for g in genome:
        for _ in range(8):
            a, b, ln = rng.integers(0, len(g) - 6000), rng.integers(0, len(g) - 6000), int(rng.integers(300, 5000))
            g[b:b + ln] = g[a:a + ln]
            if ln > 1000:                                    # a diverged copy next to the exact one
                c = int(rng.integers(0, len(g) - 6000))
                g[c:c + ln] = g[a:a + ln]
                for p in rng.integers(0, ln, size=ln // 25):
                    g[c + p] = rng.integers(1, 5)
    idx = F.fmindex(genome)
    dev_ctx, host_ctx = F.context(idx), F.context(idx)
    stats = F.statistics("simulated").attach(host_ctx)
    n_total = 0
    for length, rate, n_reads, seed in [(3000, 0.08, 120, 1), (800, 0.05, 200, 2), (6000, 0.04, 40, 3)]:
        reads, _, _ = S.make_reads(genome, n_reads, length, rate, seed=700 + seed)
        for kw in (dict(), dict(interval_optimization=True), dict(direct_full_verification=True), dict(bottom_up_pex_tree=True)):
            p = F.params(error_probability=rate, **kw)
            dev_ctx.path_counters(reset=True); host_ctx.path_counters(reset=True)
            a = F.aligner(dev_ctx, p).align_reads(reads)
            b = F.aligner(host_ctx, p).align_reads(reads)
            assert a.records() == b.records(), (length, kw)
            assert dev_ctx.path_counters()["inner_tests_requested"] == host_ctx.path_counters()["inner_tests_requested"]
            n_total += n_reads
    assert stats.num_queries == n_total
    dev_ctx.close(); host_ctx.close()


@pytest.mark.gpu
def test_seeds_written_on_the_device_equal_the_host_list(monkeypatch):
    """A chunk's seeds are written by a kernel from the reads' description (seed = function of read, orientation and sampled leaf;
    FLX_HOST_SEEDS=1: the host lists them and copies them over): same records, ragged read lengths (many trees and seed classes in one
    chunk), seed sampling steps 1 and 3, 1 and 2 seed errors, statistics attached (they read per-seed counters back)"""
    genome = S.make_genome(300_000, 3, seed=71)
    rng = np.random.default_rng(5)
    reads = []
    for i, length in enumerate(rng.integers(400, 4000, size=96)):
        r, _, _ = S.make_reads(genome, 1, int(length), 0.06, seed=900 + i)
        reads += r
    ctx = F.context(F.fmindex(genome))
    for kw in (dict(), dict(seed_sampling_step_size=3), dict(seed_errors=1), dict(interval_optimization=True), dict(max_anchors_soft=501, max_anchors_hard=100000)):
        p = F.params(error_probability=0.06, **kw)
        monkeypatch.delenv("FLX_HOST_SEEDS", raising=False)
        dev = F.aligner(ctx, p).align_reads(reads).records()
        monkeypatch.setenv("FLX_HOST_SEEDS", "1")
        host = F.aligner(ctx, p).align_reads(reads).records()
        assert dev == host and len(dev) >= len(reads) // 2, kw
    monkeypatch.delenv("FLX_HOST_SEEDS", raising=False)
    exp = O.Index(genome).run(reads, O.params(error_probability=0.06), threads=8)
    assert F.aligner(ctx, F.params(error_probability=0.06)).align_reads(reads).records() == exp.records()
    ctx.close()


# ---------------------------------------------------------------- BASELINE.json configs[0]: the reference's own CPU-runnable case
def test_config0_1mb_reference_1k_reads_of_1kb_match_oracle():
    """BASELINE.json configs[0] at its own shape: 1 Mb random reference + 1 000 simulated reads of 1 kb @ 5 %, floxer defaults and -I:
    every record equal to the oracle's (flag, reference, position, NM, CIGAR), every read mapped at its simulated place."""
    genome = S.make_genome(1_000_000, 1, seed=S.DEFAULT_SEED)
    reads, names, truth = S.make_reads(genome, 1000, 1000, 0.05, seed=S.DEFAULT_SEED + 1)
    fidx = F.fmindex(genome)
    ctx = F.context(fidx)
    oidx = O.Index(genome, imported=(fidx.suffix_array_u32(), fidx.bwt(False), fidx.bwt(True)))      # (oracle SA == product SA: test_host_cpu)
    for kw, okw in ((dict(), dict()), (dict(interval_optimization=True), dict(interval_opt=True))):
        got = F.aligner(ctx, F.params(error_probability=0.05, **kw)).align_reads(reads)
        exp = oidx.run(reads, O.params(error_probability=0.05, **okw), threads=16)
        assert got.skipped.tolist() == exp.skipped.tolist()
        assert np.array_equal(got.rows[:, :5], exp.rows[:, :5]) and got.records() == exp.records(), kw
        prim = got.rows[(got.rows[:, 1] & 256) == 0]
        assert len(prim) == 1000 and ((prim[:, 1] & 4) == 0).all()
        assert (np.abs(prim[:, 3] - np.array([t[1] for t in truth])) <= 100).all()
    ctx.close()


# ---------------------------------------------------------------- the metric's size: 3.1 Gb in 25 sequences, against the oracle
@pytest.fixture(scope="module")
def grch38_size():
    """the bench's reference (BASELINE.json configs[3] / [4]: 3.1 Gb in 25 sequences, flx_sim_genome with the default seed), its index built
    on the GPU, a context on it, and the oracle's index laid out around the product's suffix array and BWTs (a text has one suffix array;
    oracle SA == product SA is tested on small inputs, the device-built index against the text alone at 250 Mb)"""
    chrom = 124_000_000
    pool, genome = S.make_genome_fast(chrom, 25, seed=S.DEFAULT_SEED)
    fidx = F.fmindex(genome, device=0)
    ctx = F.context(fidx)
    oidx = O.Index(genome, imported=(fidx.suffix_array_u32(), fidx.bwt(False), fidx.bwt(True)), pool=pool)
    yield pool, genome, [chrom] * 25, ctx, oidx
    ctx.close()


def test_scale_grch38_size_matches_oracle_truth_and_batch_invariance(grch38_size):
    """The metric's configuration (BASELINE.json configs[3] shape): 3.1 Gb in 25 sequences, 10-kb reads @ 8 %, floxer defaults and -I.
    All 512 reads: exactly one primary, at the simulated sequence / position / strand; CIGARs of a sample consistent with the two texts
    and their NM; a read's records do not depend on its batch. And record for record the ORACLE's output on a sample of 32 reads, without
    and with -I (what floxer_whole_program_via_cli_test.cpp:40-94 compares, at the size the metric is quoted on)."""
    pool, genome, chrom_lens, ctx, oidx = grch38_size
    NR, L, rate = 512, 10000, 0.08
    (rpool, offs), (chrom, pos, rev) = S.make_reads_fast(pool, chrom_lens, NR, L, rate, seed=92)
    res = F.aligner(ctx, F.params(error_probability=rate)).align_reads((rpool, offs))
    rows = res.rows
    prim = rows[(rows[:, 1] & 256) == 0]
    assert len(prim) == NR and (prim[:, 0] == np.arange(NR)).all()
    assert ((prim[:, 1] & 4) == 0).all()
    assert (prim[:, 2] == chrom).all() and (np.abs(prim[:, 3] - pos.astype(np.int64)) <= 0.1 * L).all()
    assert (((prim[:, 1] & 16) != 0) == (rev != 0)).all()
    sample = list(range(0, NR, 16))
    reads = [rpool[int(offs[i]):int(offs[i + 1])] for i in sample]
    sub = F.aligner(ctx, F.params(error_probability=rate)).align_reads(reads)
    for j, i in enumerate(sample):
        assert _recs_of(sub, j) == _recs_of(res, i)
    _check_cigars(genome, reads[:8], [r for r in sub.records() if r[0] < 8], rate)
    exp = oidx.run(reads, O.params(error_probability=rate), threads=16)
    assert sub.skipped.tolist() == exp.skipped.tolist()
    assert np.array_equal(sub.rows[:, :5], exp.rows[:, :5]) and sub.records() == exp.records()
    opt = F.aligner(ctx, F.params(error_probability=rate, interval_optimization=True)).align_reads((rpool, offs))
    oprim = opt.rows[(opt.rows[:, 1] & 256) == 0]
    assert len(oprim) == NR and (oprim[:, 2] == chrom).all() and (np.abs(oprim[:, 3] - pos.astype(np.int64)) <= 0.1 * L).all()
    sub_i = F.aligner(ctx, F.params(error_probability=rate, interval_optimization=True)).align_reads(reads)
    exp_i = oidx.run(reads, O.params(error_probability=rate, interval_opt=True), threads=16)
    assert sub_i.records() == exp_i.records()


def test_hifi_shape_at_grch38_size_matches_oracle(grch38_size):
    """BASELINE.json configs[4]'s shape at full size: 20 kb reads @ 2 % against the 3.1 Gb reference (25 sequences), floxer defaults.
    Every read has one primary at its simulated sequence, place and strand; the CIGARs of a sample are consistent with both texts and
    their NM; batch invariance; -I keeps every primary; and the oracle's records, one for one, on a sample of 32 reads (defaults and -I)."""
    pool, genome, chrom_lens, ctx, oidx = grch38_size
    (rpool, offs), (tc, tp, tr) = S.make_reads_fast(pool, chrom_lens, 192, 20000, 0.02, seed=77)
    reads = [rpool[int(offs[i]):int(offs[i + 1])] for i in range(192)]
    p = F.params(error_probability=0.02)
    got = F.aligner(ctx, p).align_reads(reads)
    recs = got.records()
    prim = {r[0]: r for r in recs if not r[1] & 256 and not r[1] & 4}
    assert len(prim) == 192
    for i in range(192):
        r = prim[i]
        assert r[2] == int(tc[i]) and bool(r[1] & 16) == bool(tr[i]) and abs(r[3] - int(tp[i])) <= 2000, (i, r[:5], int(tc[i]), int(tp[i]))
    _check_cigars(genome, reads, [r for r in recs if r[0] < 24], 0.02)
    half = F.aligner(ctx, p).align_reads(reads[:96]).records()
    assert half == [r for r in recs if r[0] < 96]
    with_i = F.aligner(ctx, F.params(error_probability=0.02, interval_optimization=True)).align_reads(reads).records()
    prim_i = {r[0]: r for r in with_i if not r[1] & 256 and not r[1] & 4}
    assert {k: v[:5] for k, v in prim_i.items()} == {k: v[:5] for k, v in prim.items()}
    exp = oidx.run(reads[:32], O.params(error_probability=0.02), threads=16)
    assert [r for r in recs if r[0] < 32] == exp.records()
    exp_i = oidx.run(reads[:32], O.params(error_probability=0.02, interval_opt=True), threads=16)
    assert [r for r in with_i if r[0] < 32] == exp_i.records()


# ---------------------------------------------------------------- N > 1: the bench's own control flow, two ranks on the one GPU
def test_bench_two_ranks_share_one_gpu_and_gather_what_one_rank_computes(tmp_path):
    """`bench.py --gpus 2` as the driver starts it (child ranks before any GPU call, rendezvous on 127.0.0.1), rehearsed on one GPU:
    FLX_BENCH_BACKEND=gloo lets both ranks use device 0. Rank 0 builds the index and hands its HBM image to rank 1
    (distributed.replicate_index), the reads shard by rank (parallelization.cpp:77-87: reads are the independent units), every step's
    counts are exchanged and its records gathered to rank 0 inside the timed region. Checked: the line says two GPUs and carries
    gather_s; its record count is the sum of both shards; and the table rank 0 gathered for step 0 is, row for row, what one context
    in this process computes for rank 0's and rank 1's reads of that step (global read indices, rank order)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = str(tmp_path / "gathered.npy")
    B, steps, warm = 1024, 2, 1
    env = dict(os.environ, FLX_BENCH_BACKEND="gloo", FLX_LANES="4")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "ecoli", "--steps", str(steps), "--warmup", str(warm), "--reads-per-step", str(B),
           "--no-cpu-baseline", "--no-isolated-pass", "--no-host-inputs-leg", "--dump-gathered", dump]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["gather_s"] is not None and line["scaling"] == "weak"
    assert line["config"]["reads_per_step_per_gpu"] == B and line["value"] > 0
    table = np.load(dump)
    # the same reads here: bench.py gives rank r, batch b the simulator seed DEFAULT_SEED + 1 + 1000 r + b; step 0 is batch `warm`
    cfg = dict(genome=4_600_000, read_length=5000, error_rate=0.08)
    pool, genome = S.make_genome_fast(cfg["genome"], 1, seed=S.DEFAULT_SEED)
    ctx = F.context(F.fmindex(genome, device=0))
    al = F.aligner(ctx, F.params(error_probability=cfg["error_rate"]))
    parts, total = [], 0
    for rank in range(2):
        reads = S.make_reads_fast(pool, [cfg["genome"]], B, cfg["read_length"], cfg["error_rate"], seed=S.DEFAULT_SEED + 1 + rank * 1000 + warm)[0]
        res = al.align_reads(reads)
        rows = res.rows.copy()
        rows[:, 0] += rank * B                    # (step 0: global read index = rank * B + local)
        parts.append(rows)
    expected = np.concatenate(parts)
    # (column 5 is the CIGAR's offset within its owner's part: where a chunk's CIGARs land depends on the lanes' timing)
    keep = [0, 1, 2, 3, 4, 6]
    assert table.shape == expected.shape and np.array_equal(table[:, keep], expected[:, keep])
    # records of the whole run = both shards of both steps; step 0's share is known exactly
    assert line["records"] >= len(expected) and line["records"] > 2 * B * steps * 0.9
    ctx.close()


# ---------------------------------------------------------------- --stats: every bin of the reference's histograms
def _parse_stats_toml(text):
    """{section: {key: value}} of `floxer --stats` output (statistics.cpp:70-74, 126-145); '' = the counts in front of the first section"""
    import ast
    out, cur = {"": {}}, ""
    for line in text.splitlines():
        line = line.strip()
        if not line:
            continue
        if line.startswith("["):
            cur = line.strip("[]")
            out[cur] = {}
        else:
            k, v = line.split(" = ", 1)
            out[cur][k] = ast.literal_eval(v)
    return out


@pytest.mark.parametrize("kw,okw", [(dict(), dict()), (dict(interval_optimization=True), dict(interval_opt=True)),
                                     (dict(direct_full_verification=True), dict(direct_full=True))])
def test_statistics_bins_match_oracle(kw, okw):
    """statistics::search_and_alignment_statistics (statistics.hpp:24-172): the count and the sixteen histograms that do not measure
    wall-clock time, bin by bin. The oracle keeps the raw values the reference's call sites insert (parallelization.cpp:107-110, 262-269;
    verification.cpp:130, 238-242); this test bins them itself (statistics.cpp:80-94: the first threshold >= value, else the last bin) over
    the reference's "simulated" scales (statistics.cpp:35-62) and compares with the TOML the product writes: thresholds, occurrences,
    num_values, min, max, mean. A text with repeats (caps bite, windows avoided under -I), two read shapes."""
    genome = S.make_genome(300_000, 3, seed=31)
    rng = np.random.default_rng(8)
    for g in genome:
        for _ in range(10):
            a, b, ln = rng.integers(0, len(g) - 4000), rng.integers(0, len(g) - 4000), int(rng.integers(200, 3000))
            g[b:b + ln] = g[a:a + ln]
    genome[0][5000:5600] = 2
    reads = S.make_reads(genome, 90, 2500, 0.07, seed=41)[0] + S.make_reads(genome, 60, 700, 0.05, seed=42)[0]
    fidx = F.fmindex(genome)
    ctx = F.context(fidx)
    stats = F.statistics("simulated").attach(ctx)
    # (one error probability for both shapes: the statistics object sees one run, as `floxer --stats` does)
    got = F.aligner(ctx, F.params(error_probability=0.07, **kw)).align_reads(reads)
    exp = O.Index(genome).run(reads, O.params(error_probability=0.07, **okw), threads=8)
    assert got.records() == exp.records()
    toml = _parse_stats_toml(stats.format(toml=True))

    def linear_range(steps, mx):
        return [i * mx // steps for i in range(steps)]
    small, medium, tiny = linear_range(30, 100), linear_range(30, 1000), [0, 1, 2, 3, 4]
    qlen, anchor, per_seed, edit = linear_range(30, 10_000), linear_range(30, 1000), linear_range(30, 200), linear_range(30, 1000)
    sections = [("query_lengths", qlen), ("seed_lengths", small), ("errors_per_seed", tiny), ("seeds_per_query", medium),
                ("fully_excluded_seeds_per_query", medium), ("kept_anchors_per_query", anchor), ("excluded_raw_anchors_by_soft_cap_per_query", anchor),
                ("excluded_raw_anchors_by_erase_useless_per_query", anchor), ("kept_anchors_per_kept_seed", per_seed),
                ("excluded_raw_anchors_by_soft_cap_per_kept_seed", per_seed), ("excluded_raw_anchors_by_erase_useless_per_kept_seed", per_seed),
                ("reference_span_sizes_aligned_of_inner_nodes", qlen), ("reference_span_sizes_aligned_of_roots", qlen),
                ("reference_span_sizes_alignment_avoided_of_roots", qlen), ("alignments_per_query", small), ("alignments_edit_distance", edit)]
    assert toml[""]["completely_excluded_queries"] == int(exp.stat_values[16][0])
    for i, (name, thresholds) in enumerate(sections):
        values = [int(v) for v in exp.stat_values[i]]
        occ = [0] * (len(thresholds) + 1)
        for v in values:
            occ[next((j for j, t in enumerate(thresholds) if v <= t), len(thresholds))] += 1
        sec = toml[name]
        assert sec["thresholds"] == thresholds, name
        assert sec["num_values"] == len(values), (name, sec["num_values"], len(values))
        assert sec["occurrences"] == occ, (name, sec["occurrences"], occ)
        if values:
            assert sec["min_value"] == min(values) and sec["max_value"] == max(values), name
            assert sec["mean"] == float(f"{sum(values) / len(values):.2f}"), name
    if kw.get("interval_optimization"):
        assert len(exp.stat_values[13]) > 0          # windows were avoided: the histogram is exercised
    # the two wall-clock histograms exist with one value per query
    for name in ("milliseconds_spent_in_search_per_query", "milliseconds_spent_in_verification_per_query"):
        assert toml[name]["num_values"] == len(exp.stat_values[0])
    ctx.close()
