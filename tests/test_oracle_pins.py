"""Pins the CPU oracle (oracle/) against every known-answer vector the reference's own tests hold for the hot path
(tests/golden/reference_pins.json, re-typed from /root/reference/test/*.cpp with file:line), and against slow
brute-force definitions. CPU only."""
import itertools

import numpy as np
import pytest

import oracle_lib as O


# ---------------------------------------------------------------- math / input (math_test.cpp, input_test.cpp)
def test_math_pins(pins):
    L = O.lib()
    for a, b, e in pins["math"]["ceil_div"]:
        assert L.orc_ceil_div(a, b) == e
    for v, e in pins["math"]["fp_ceil"]:
        assert L.orc_fp_ceil(v) == e
    for v, e in pins["math"]["saturate"]:
        assert L.orc_saturate_i32(v) == e


def test_input_pins(pins):
    for s, e in pins["input"]["ranks"]:
        assert O.chars_to_ranks(s).tolist() == e
    assert O.revcomp([1, 2, 3, 4, 5, 0, 1]).tolist() == [4, 0, 5, 1, 2, 3, 4]


# ---------------------------------------------------------------- PEX (pex_test.cpp)
def test_pex_pins(pins):
    for case in pins["pex"]:
        inner, leaves = O.pex_build(case["len"], case["k"], case["s"], case["bottom_up"])
        got = [[int(l[1]), int(l[2]), int(l[3])] for l in leaves]
        assert got == case["leaves"]


@pytest.mark.parametrize("length,k,s", [(1000, 50, 2), (5000, 400, 2), (10000, 800, 2), (20000, 400, 2), (12, 2, 1), (12, 2, 0)])
def test_pex_structure(length, k, s):
    """leaf/inner counts and invariants of the recursive builder (pex.cpp:102-107 asserts)."""
    inner, leaves = O.pex_build(length, k, s, False)
    null = np.uint64(2 ** 64 - 1)
    root = inner[0] if len(inner) else leaves[0]
    assert root[0] == null and root[1] == 0 and root[2] == length - 1
    assert k <= root[3] <= k + s
    # leaves tile the query
    assert leaves[0][1] == 0 and leaves[-1][2] == length - 1
    assert all(int(leaves[i][2]) + 1 == int(leaves[i + 1][1]) for i in range(len(leaves) - 1))
    assert all(l[3] <= s for l in leaves)
    expected = {(1000, 50): (19, 18), (5000, 400): (145, 144), (10000, 800): (289, 288), (20000, 400): (145, 144)}
    if (length, k) in expected:
        assert (len(leaves), len(inner)) == expected[(length, k)]


def test_pex_bottom_up_root_errors():
    inner, leaves = O.pex_build(30, 5, 1, True)   # verification_test.cpp:46-51
    assert len(leaves) == 3 and [int(l[3]) for l in leaves] == [1, 1, 1]
    assert int(inner[0][3]) == 5 and int(inner[0][1]) == 0 and int(inner[0][2]) == 29


# ---------------------------------------------------------------- intervals (intervals_test.cpp)
def test_interval_pins(pins):
    iv = pins["intervals"]
    named, codes = iv["named"], iv["relationship_codes"]
    for a, b, rel in iv["relationship"]:
        assert O.lib().orc_relationship(*named[a], *named[b]) == codes[rel], (a, b, rel)
    import ctypes as C
    for s, e, amt, es, ee in iv["trim"]:
        os_, oe = C.c_uint64(), C.c_uint64()
        O.lib().orc_trim(s, e, amt, C.byref(os_), C.byref(oe))
        assert (os_.value, oe.value) == (es, ee)
    ivls = O.Intervals(True)
    for step in iv["verified_intervals_steps"]:
        for name in step["insert"]:
            ivls.insert(*named[name])
        for name in step.get("contains_self", []):
            assert ivls.contains(*named[name])
        got = [ivls.contains(*named[o]) for o in iv["others"]]
        assert got == step["expect"], step
    off = O.Intervals(False)
    off.insert(0, 100)
    assert not off.contains(5, 6) and len(off) == 0     # intervals.cpp:85, 95-97


# ---------------------------------------------------------------- search.cpp pins
def test_erase_useless_pin(pins):
    p = pins["erase_useless_anchors"]
    assert O.erase_useless(p["in"]).tolist() == p["out"]


def test_search_seeds_reference_setup(pins):
    """search_test.cpp:6-75 asserts only num_fully_excluded_seeds == 0; the listed expectations there are never compared.
    We additionally check the seeds find what the comments in that test describe."""
    s = pins["search_seeds_setup"]
    idx = O.Index(s["references"])
    anchors, stats = idx.search_seeds(s["query"], s["seeds"], hard=10, soft=10, order=1, choice=0, erase=True)
    assert stats[:, 3].sum() == 0                                  # nothing fully excluded
    by_seed = {i: [tuple(int(x) for x in a[2:]) for a in anchors if a[0] == i] for i in range(4)}
    assert (0, 0, 0) in by_seed[0]                                  # "matches exactly": ref 0 pos 0, 0 errors
    assert any(r == 0 and e == 1 and 5 <= p <= 7 for r, p, e in by_seed[1])   # one mismatch inside the C block
    assert any(r == 1 and e == 1 for r, p, e in by_seed[2])        # one deletion, second reference
    assert by_seed[3] == []                                          # "does not match"


# ---------------------------------------------------------------- alignment (alignment_test.cpp + seqan3 tie rules)
@pytest.mark.parametrize("algo", [0, 1])
def test_alignment_pin(pins, algo):
    a = pins["alignment"]
    assert O.align(a["reference"], a["query"], a["k"], mode=2, algo=algo) == (a["nm"], a["start"], a["cigar"])
    assert O.align(a["reference"], a["query"], 0, mode=0, algo=algo) is None
    r = O.align(a["reference"], a["query"], a["k"], mode=1, algo=algo)
    assert r[0] == a["nm"] and r[1] == a["start"]


@pytest.mark.parametrize("algo", [0, 1])
def test_alignment_tie_rules_from_whole_program_pins(algo):
    """floxer_whole_program_via_cli_test.cpp:47-93: rightmost end column, I preferred over X."""
    ref = O.chars_to_ranks("A" * 17 + "C" * 19 + "G" * 18 + "T" * 17)
    rc = lambda s: O.revcomp(O.chars_to_ranks(s))
    assert O.align(ref, O.chars_to_ranks("GGGGAAGGGGGG"), 2, algo=algo) == (2, 44, "4=2I6=")
    assert O.align(ref, rc("GGGGAAGGGGGG"), 2, algo=algo) == (2, 26, "6=2I4=")
    assert O.align(ref, O.chars_to_ranks("TTTTTTTTTTGG"), 2, algo=algo) == (2, 61, "10=2I")
    assert O.align(ref, rc("TTTTTTTTTTGG"), 2, algo=algo) == (2, 7, "2I10=")
    assert O.align(ref, O.chars_to_ranks("AAAAAACCCCCC"), 2, algo=algo) == (0, 11, "12=")


def _rand_pair(rng, m, n, err):
    q = rng.integers(1, 5, size=m).astype(np.uint8)
    core = []
    for c in q:
        r = rng.random()
        if r < err / 3:
            continue
        if r < 2 * err / 3:
            core.append(rng.integers(1, 5))
        if r < err:
            core.append(rng.integers(1, 5))
        else:
            core.append(c)
    core = np.array(core, dtype=np.uint8)
    pad = max(0, n - len(core))
    left = rng.integers(0, pad + 1)
    ref = np.concatenate([rng.integers(1, 5, size=left), core, rng.integers(1, 5, size=pad - left)]).astype(np.uint8)
    return ref, q


def test_myers_equals_dp_random():
    rng = np.random.default_rng(1)
    for it in range(300):
        m = int(rng.integers(1, 200))
        n = int(rng.integers(1, 260))
        ref, q = _rand_pair(rng, m, n, 0.15 if it % 3 else 0.5)
        if it % 5 == 0:   # low-complexity sequences create many ties
            ref = (ref % 2 + 1).astype(np.uint8)
            q = (q % 2 + 1).astype(np.uint8)
        k = int(rng.integers(0, m + 1))
        for mode in (0, 1, 2):
            assert O.align(ref, q, k, mode=mode, algo=0) == O.align(ref, q, k, mode=mode, algo=1), (it, mode)


def _cigar_check(ref, q, res):
    import re
    nm, begin, cig = res
    i, j, cost = 0, begin, 0
    for ln, op in re.findall(r"(\d+)([=XID])", cig):
        for _ in range(int(ln)):
            if op == "=":
                assert q[i] == ref[j]; i += 1; j += 1
            elif op == "X":
                assert q[i] != ref[j]; i += 1; j += 1; cost += 1
            elif op == "I":
                i += 1; cost += 1
            else:
                j += 1; cost += 1
    assert i == len(q) and cost == nm


def test_cigar_is_consistent_and_optimal():
    rng = np.random.default_rng(2)
    for it in range(100):
        m = int(rng.integers(5, 150))
        ref, q = _rand_pair(rng, m, m + 40, 0.2)
        res = O.align(ref, q, m, mode=2, algo=1)
        assert res is not None
        _cigar_check(ref, q, res)
        # optimal: brute-force semi-global minimum
        D = np.zeros((m + 1, len(ref) + 1), dtype=np.int32)
        D[:, 0] = np.arange(m + 1)
        for i in range(1, m + 1):
            for j in range(1, len(ref) + 1):
                D[i, j] = min(D[i - 1, j - 1] + (q[i - 1] != ref[j - 1]), D[i - 1, j] + 1, D[i, j - 1] + 1)
        assert res[0] == D[m].min()


# ---------------------------------------------------------------- verification (verification_test.cpp)
def test_span_pins(pins):
    s = pins["span"]
    for ratio, exp in s["cases"]:
        assert list(O.span(s["anchor_pos"], s["node"][0], s["node"][1], s["node"][2], s["leaf_from"], s["reflen"], ratio)) == exp


def test_verify_pin(pins):
    v = pins["verification_verify"]
    pv = O.params(query_errors=v["k"], seed_errors=v["s"], bottom_up=True, interval_opt=True, extra_ratio=v["extra_ratio"])
    ivs = O.Intervals(True)
    args = (len(v["query"]), v["k"], v["s"], True, v["anchor"]["leaf"], v["anchor"]["pos"], v["anchor"]["errors"])
    got = O.verify_anchor(*args, v["query"], True, v["reference"], pv, ivs)
    assert got == [(v["start"], v["nm"], True, v["cigar"])]
    assert O.verify_anchor(*args, v["query"], True, v["reference"], pv, ivs) == []       # cached (verification_test.cpp:88-91)
    pv_direct = O.params(query_errors=v["k"], seed_errors=v["s"], bottom_up=True, interval_opt=False, extra_ratio=v["extra_ratio"],
                         direct_full=True)
    assert O.verify_anchor(*args, v["query"], True, v["reference"], pv_direct, None) == got
    q = list(v["query"])
    for i, c in v["mutations_for_no_alignment"]:
        q[i] = c
    assert O.verify_anchor(*args, q, True, v["reference"], pv_direct, None) == []


@pytest.mark.parametrize("algo", [0, 1])
def test_try_align_node_pin(pins, algo):
    t = pins["try_align_node"]
    ref = np.array(t["reference"], dtype=np.uint8)[t["span"][0]: t["span"][0] + t["span"][1]]
    q = np.array(t["query"], dtype=np.uint8)
    node = t["node"]
    res = O.align(ref, q[node[0]: node[1] + 1], node[2], mode=2, algo=algo)
    assert res[0] == t["nm"] and t["span"][0] + res[1] == t["start"]
    assert O.align(ref, q[node[0]: node[1] + 1], node[2], mode=0, algo=algo) is not None
    q[t["extra_error"][0]] = t["extra_error"][1]
    assert O.align(ref, q[node[0]: node[1] + 1], node[2], mode=0, algo=algo) is None


# ---------------------------------------------------------------- end to end (floxer_whole_program_via_cli_test.cpp)
def _read_fasta(path):
    recs, name, seq = [], None, []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if name is not None:
                recs.append((name, "".join(seq)))
            name, seq = line[1:], []
        else:
            seq.append(line)
    recs.append((name, "".join(seq)))
    return recs


def _read_fastq(path):
    lines = [l.rstrip("\n") for l in open(path)]
    return [(lines[i][1:], lines[i + 1], lines[i + 3]) for i in range(0, len(lines) - 3, 4)]


@pytest.mark.parametrize("seed_errors", [0, 1])
def test_whole_program_pins(pins, seed_errors):
    import os
    g = os.path.join(os.path.dirname(__file__), "golden")
    refs = _read_fasta(os.path.join(g, "reference.fasta"))
    reads = _read_fastq(os.path.join(g, "queries.fastq"))
    idx = O.Index([O.chars_to_ranks(s) for _, s in refs])
    wp = pins["whole_program"]
    pv = O.params(query_errors=2, seed_errors=seed_errors, extra_ratio=2.0, interval_opt=True)
    res = idx.run([O.chars_to_ranks(s) for _, s, _ in reads], pv)
    recs = res.records()
    ids = [r[0] for r in reads]
    assert {ids[r[0]] for r in recs} == set(wp["ids"])
    for ridx, flag, ref_id, pos, nm, cig in recs:
        name = ids[ridx]
        if name in wp["unmapped"]:
            assert flag == 4
            continue
        assert not (flag & 4)
        assert ref_id == 0
        for ename, erev, pmin, pmax, enm, ecig in wp["expect"]:
            if ename == name and erev == bool(flag & 16):
                assert pmin <= pos <= pmax and nm == enm and cig == ecig, (name, flag, pos, nm, cig)
    # every expected (id, strand) combination is present at least once
    seen = {(ids[r[0]], bool(r[1] & 16)) for r in recs if not r[1] & 4}
    for ename, erev, *_ in wp["expect"]:
        assert (ename, erev) in seen
    # exactly one primary record per mapped read, carrying the best NM (output.cpp:66-67)
    for i, name in enumerate(ids):
        mine = [r for r in recs if r[0] == i]
        prim = [r for r in mine if not r[1] & 256]
        assert len(prim) == 1
        if not prim[0][1] & 4:
            assert prim[0][4] == min(r[4] for r in mine)


# ---------------------------------------------------------------- FM index + search against brute force
def test_suffix_array_and_locate_bruteforce():
    rng = np.random.default_rng(3)
    refs = [rng.integers(1, 5, size=n).astype(np.uint8) for n in (37, 8, 101, 4)]
    refs[2][10:40] = 1    # a repeat
    idx = O.Index(refs)
    text = []
    starts = []
    for r in refs:
        starts.append(len(text))
        text += r.tolist() + [0] * (4 - len(r) % 4)
    n = len(text)
    assert idx.n == n
    naive = sorted(range(n), key=lambda i: text[i:])
    sa = idx.sa()
    assert sa.tolist() == naive
    assert idx.bwt().tolist() == [text[(i - 1) % n] for i in naive]
    rtext = text[::-1]
    rnaive = sorted(range(n), key=lambda i: rtext[i:])
    assert idx.bwt(True).tolist() == [rtext[(i - 1) % n] for i in rnaive]
    for row in range(n):
        p = int(sa[row])
        if text[p] == 0:
            continue
        s = max(i for i in range(len(starts)) if starts[i] <= p)
        assert idx.locate(row) == (s, p - starts[s])


def _edit_distance(a, b):
    D = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        prev, D[0] = D[0], i
        for j in range(1, len(b) + 1):
            cur = min(prev + (a[i - 1] != b[j - 1]), D[j] + 1, D[j - 1] + 1)
            prev, D[j] = D[j], cur
    return D[len(b)]


@pytest.mark.parametrize("k", [0, 1, 2, 3])
def test_search_finds_every_occurrence_within_k(k):
    """Soundness: everything reported is within k edits and carries an error count >= its true edit distance.
    Completeness for substitution-only occurrences with matching outer characters."""
    rng = np.random.default_rng(10 + k)
    ref = rng.integers(1, 5, size=600).astype(np.uint8)
    ref[100:130] = ref[300:330]                    # a repeat so that groups have count > 1
    idx = O.Index([ref])
    sa = idx.sa()
    text = ref.tolist()
    for trial in range(6):
        L = int(rng.integers(max(6, k + 3), 16))
        start = int(rng.integers(0, len(ref) - L))
        seed = ref[start:start + L].copy()
        for _ in range(int(rng.integers(0, k + 1))):
            seed[int(rng.integers(1, L - 1))] = rng.integers(1, 5)
        groups, _ = idx.search_groups(seed, k, n=10 ** 9)
        reported = {}
        for lb, ln, e in groups.tolist():
            assert e <= k
            for row in range(lb, lb + ln):
                reported.setdefault(int(sa[row]), []).append(e)
        # soundness: some substring starting at each reported position is within e edits
        for pos, errs in reported.items():
            best = min(_edit_distance(seed.tolist(), text[pos:pos + ln2]) for ln2 in range(max(1, L - k), L + k + 1)
                       if pos + ln2 <= len(text))
            assert best <= min(errs)
        # completeness (Hamming case): every same-length substring within k substitutions whose outer characters match
        # must be reported. (Alignments that need two edit operations on one query character at the end of a part
        # can be pruned by the per-character lower bound, so indel completeness is not asserted.)
        sl = seed.tolist()
        for pos in range(len(text) - L + 1):
            sub = text[pos:pos + L]
            if sub[0] == sl[0] and sub[-1] == sl[-1] and sum(a != b for a, b in zip(sl, sub)) <= k:
                assert pos in reported, (trial, pos)


def test_scheme_is_complete_and_disjoint():
    """Every error distribution over the parts is covered by exactly one search (search scheme validity)."""
    schemes = {
        0: [([0], [0], [0])],
        1: [([0, 1], [0, 0], [0, 1]), ([1, 0], [0, 1], [0, 1])],
        2: [([0, 1, 2, 3], [0, 0, 1, 1], [0, 0, 2, 2]), ([2, 1, 0, 3], [0, 0, 0, 0], [0, 1, 1, 2]),
            ([3, 2, 1, 0], [0, 0, 0, 2], [0, 1, 2, 2])],
        3: [([0, 1, 2, 3, 4], [0, 0, 0, 0, 0], [0, 0, 3, 3, 3]), ([2, 1, 0, 3, 4], [0, 0, 1, 1, 1], [0, 1, 1, 2, 3]),
            ([3, 2, 1, 0, 4], [0, 0, 0, 2, 2], [0, 1, 2, 2, 3]), ([4, 3, 2, 1, 0], [0, 0, 0, 0, 3], [0, 2, 2, 3, 3])],
    }
    for k, searches in schemes.items():
        P = len(searches[0][0])
        for dist in itertools.product(range(k + 1), repeat=P):
            if sum(dist) > k:
                continue
            covering = 0
            for pi, l, u in searches:
                c, ok = 0, True
                for i in range(P):
                    c += dist[pi[i]]
                    ok &= l[i] <= c <= u[i]
                covering += ok
            assert covering == 1, (k, dist)


def test_hard_cap_and_truncation():
    ref = np.ones(3000, dtype=np.uint8)          # poly-A: every seed has > 500 raw anchors
    idx = O.Index([ref])
    groups, _ = idx.search_groups(np.ones(20, np.uint8), 0, n=501)
    assert groups[:, 1].sum() == 501              # truncated to hard+1 (search.cpp:177-179)
    anchors, stats = idx.search_seeds(np.ones(20, np.uint8), [[0, 20, 0, 0]])
    assert len(anchors) == 0 and stats[0, 3] == 1   # fully excluded (search.cpp:190-202)
    anchors, stats = idx.search_seeds(np.ones(20, np.uint8), [[0, 20, 0, 0]], choice=2, erase=False)
    assert len(anchors) == 50                      # first_reported ignores the hard cap, keeps soft many
