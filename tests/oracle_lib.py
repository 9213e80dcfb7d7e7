"""ctypes binding of the CPU oracle (oracle/). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package (floxer_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "_build", "liboracle.so")


def build(force=False):
    srcs = [os.path.join(_ORACLE_DIR, f) for f in ("floxer_oracle.cpp", "oracle_capi.cpp", "floxer_oracle.hpp")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _SO


_lib = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.orc_ceil_div.restype = C.c_uint64
        L.orc_ceil_div.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_fp_ceil.restype = C.c_uint64
        L.orc_fp_ceil.argtypes = [C.c_double]
        L.orc_saturate_i32.restype = C.c_int32
        L.orc_saturate_i32.argtypes = [C.c_uint64]
        L.orc_chars_to_ranks.argtypes = [C.c_char_p, C.c_uint64, u8p]
        L.orc_revcomp.argtypes = [u8p, C.c_uint64, u8p]
        L.orc_pex_build.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, u64p, C.c_uint64, u64p, u64p]
        L.orc_index_build.restype = C.c_void_p
        L.orc_index_build.argtypes = [u8p, u64p, C.c_uint32, C.c_uint32]
        L.orc_index_import.restype = C.c_void_p
        L.orc_index_import.argtypes = [u8p, u64p, C.c_uint32, C.c_uint32, u32p, u8p, u8p]
        L.orc_index_free.argtypes = [C.c_void_p]
        L.orc_index_size.restype = C.c_uint64
        L.orc_index_size.argtypes = [C.c_void_p]
        L.orc_index_sa.argtypes = [C.c_void_p, i64p]
        L.orc_index_bwt.argtypes = [C.c_void_p, C.c_int, u8p]
        L.orc_index_locate.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p]
        L.orc_search_groups.restype = C.c_int64
        L.orc_search_groups.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_uint32, C.c_uint64, u64p, C.c_uint64, u64p]
        L.orc_search_seeds.restype = C.c_int64
        L.orc_search_seeds.argtypes = [C.c_void_p, u8p, u64p, C.c_uint64, u64p, u64p, C.c_uint64, u64p]
        L.orc_erase_useless.restype = C.c_uint64
        L.orc_erase_useless.argtypes = [u64p, C.c_uint64]
        L.orc_align.restype = C.c_int
        L.orc_align.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, u64p, u64p, u32p, u64p]
        L.orc_relationship.restype = C.c_int
        L.orc_relationship.argtypes = [C.c_uint64] * 4
        L.orc_trim.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, u64p, u64p]
        L.orc_intervals_new.restype = C.c_void_p
        L.orc_intervals_new.argtypes = [C.c_int]
        L.orc_intervals_free.argtypes = [C.c_void_p]
        L.orc_intervals_insert.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.orc_intervals_contains.restype = C.c_int
        L.orc_intervals_contains.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.orc_intervals_count.restype = C.c_uint64
        L.orc_intervals_count.argtypes = [C.c_void_p]
        L.orc_span.argtypes = [C.c_uint64] * 6 + [C.c_double, u64p]
        L.orc_verify_anchor.restype = C.c_int64
        L.orc_verify_anchor.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, u8p,
                                        C.c_int, u8p, C.c_uint64, f64p, C.c_void_p, u64p, C.c_uint64, u32p, C.c_uint64]
        L.orc_run.restype = C.c_void_p
        L.orc_run.argtypes = [C.c_void_p, u8p, u64p, C.c_uint64, f64p, C.c_uint32]
        L.orc_run_free.argtypes = [C.c_void_p]
        L.orc_run_seconds.restype = C.c_double
        L.orc_run_seconds.argtypes = [C.c_void_p]
        L.orc_run_num_records.restype = C.c_uint64
        L.orc_run_num_records.argtypes = [C.c_void_p]
        L.orc_run_num_cigar_words.restype = C.c_uint64
        L.orc_run_num_cigar_words.argtypes = [C.c_void_p]
        L.orc_run_get.argtypes = [C.c_void_p, i64p, u32p, u8p]
        L.orc_run_counters.argtypes = [C.c_void_p, u64p]
        L.orc_run_stat_values.restype = C.c_uint64
        L.orc_run_stat_values.argtypes = [C.c_void_p, C.c_uint32, u64p]
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def as_u8(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint8))


CIGAR_OPS = "MIDNSHP=X"
N_STAT_LISTS = 16          # run_statistics, oracle/floxer_oracle.hpp


def cigar_str(words):
    return "".join(f"{int(w) >> 4}{CIGAR_OPS[int(w) & 15]}" for w in words)


def chars_to_ranks(s):
    b = s.encode() if isinstance(s, str) else s
    out = np.zeros(len(b), dtype=np.uint8)
    lib().orc_chars_to_ranks(b, len(b), _p(out, u8p))
    return out


def revcomp(r):
    r = as_u8(r)
    out = np.zeros_like(r)
    lib().orc_revcomp(_p(r, u8p), len(r), _p(out, u8p))
    return out


def pex_build(length, k, s, bottom_up=False):
    cap = 4 * (k + 2) + 16
    nodes = np.zeros((cap, 4), dtype=np.uint64)
    ni, nl = C.c_uint64(), C.c_uint64()
    rc = lib().orc_pex_build(length, k, s, int(bottom_up), _p(nodes, u64p), cap, C.byref(ni), C.byref(nl))
    assert rc == 0
    return nodes[: ni.value].copy(), nodes[ni.value: ni.value + nl.value].copy()


# default parameter vector (floxer_cli.hpp:41-70)
def params(error_probability=-1.0, query_errors=0, seed_errors=2, hard=500, soft=50, group_order="count_first",
           choice="round_robin", erase=True, seed_step=1, bottom_up=False, interval_opt=False, extra_ratio=0.05,
           direct_full=False, anchors_per_task=3000, without_cigar=False, align_algo=1):
    order = {"errors_first": 0, "count_first": 1, "none": 2}[group_order]
    ch = {"round_robin": 0, "full_groups": 1, "first_reported": 2}[choice]
    return np.array([error_probability, query_errors, seed_errors, hard, soft, order, ch, int(erase), seed_step, int(bottom_up),
                     int(interval_opt), extra_ratio, int(direct_full), anchors_per_task, int(without_cigar), align_algo],
                    dtype=np.float64)


class Index:
    def __init__(self, refs, sampling=4, imported=None, pool=None):
        """imported = (suffix array u32, BWT, BWT of the reversed text) of the padded text: the index is then laid out around
        these arrays instead of sorting the suffixes here (a text has one suffix array; the arrays are spot-checked)."""
        self.refs = [as_u8(r) for r in refs]
        if pool is None:
            pool = np.concatenate(self.refs) if self.refs else np.zeros(0, np.uint8)
        lens = np.array([len(r) for r in self.refs], dtype=np.uint64)
        if imported is None:
            self.h = lib().orc_index_build(_p(pool, u8p), _p(lens, u64p), len(self.refs), sampling)
        else:
            sa, bwt, bwt_rev = imported
            sa = np.ascontiguousarray(sa, dtype=np.uint32)
            bwt, bwt_rev = as_u8(bwt), as_u8(bwt_rev)
            self.h = lib().orc_index_import(_p(pool, u8p), _p(lens, u64p), len(self.refs), sampling, _p(sa, u32p), _p(bwt, u8p), _p(bwt_rev, u8p))
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_index_free(self.h)
            self.h = None

    @property
    def n(self):
        return lib().orc_index_size(self.h)

    def sa(self):
        out = np.zeros(self.n, dtype=np.int64)
        lib().orc_index_sa(self.h, _p(out, i64p))
        return out

    def bwt(self, rev=False):
        out = np.zeros(self.n, dtype=np.uint8)
        lib().orc_index_bwt(self.h, int(rev), _p(out, u8p))
        return out

    def locate(self, row):
        a, b = C.c_uint64(), C.c_uint64()
        lib().orc_index_locate(self.h, row, C.byref(a), C.byref(b))
        return a.value, b.value

    def search_groups(self, seq, k, n=501):
        seq = as_u8(seq)
        cap = 4096
        while True:
            out = np.zeros((cap, 3), dtype=np.uint64)
            ctr = np.zeros(2, dtype=np.uint64)
            r = lib().orc_search_groups(self.h, _p(seq, u8p), len(seq), k, n, _p(out, u64p), cap, _p(ctr, u64p))
            if r >= 0:
                return out[:r].copy(), ctr
            cap = -r

    def search_seeds(self, pool, seeds, hard=500, soft=50, order=1, choice=0, erase=True):
        """seeds: rows {offset, len, errors, leaf_index}. Returns (anchors rows {seed,leaf,ref,pos,err}, stats rows)."""
        pool = as_u8(pool)
        seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).reshape(-1, 4))
        cfg = np.array([hard, soft, order, choice, int(erase)], dtype=np.uint64)
        cap = max(64, len(seeds) * (soft + 1))
        anchors = np.zeros((cap, 5), dtype=np.uint64)
        stats = np.zeros((len(seeds), 4), dtype=np.uint64)
        n = lib().orc_search_seeds(self.h, _p(pool, u8p), _p(seeds, u64p), len(seeds), _p(cfg, u64p), _p(anchors, u64p), cap,
                                   _p(stats, u64p))
        assert 0 <= n <= cap
        return anchors[:n].copy(), stats

    def run(self, reads, pv, threads=1):
        reads = [as_u8(r) for r in reads]
        offs = np.zeros(len(reads) + 1, dtype=np.uint64)
        for i, r in enumerate(reads):
            offs[i + 1] = offs[i] + len(r)
        pool = np.concatenate(reads) if reads else np.zeros(0, np.uint8)
        if len(pool) == 0:
            pool = np.zeros(1, np.uint8)
        pv = np.ascontiguousarray(pv, dtype=np.float64)
        h = lib().orc_run(self.h, _p(pool, u8p), _p(offs, u64p), len(reads), _p(pv, f64p), threads)
        try:
            n = lib().orc_run_num_records(h)
            nc = lib().orc_run_num_cigar_words(h)
            rows = np.zeros((n, 7), dtype=np.int64)
            cig = np.zeros(max(nc, 1), dtype=np.uint32)
            skipped = np.zeros(max(len(reads), 1), dtype=np.uint8)
            lib().orc_run_get(h, _p(rows, i64p), _p(cig, u32p), _p(skipped, u8p))
            ctr = np.zeros(9, dtype=np.uint64)
            lib().orc_run_counters(h, _p(ctr, u64p))
            secs = lib().orc_run_seconds(h)
            stats = []
            for i in range(N_STAT_LISTS + 1):
                n = lib().orc_run_stat_values(h, i, None)
                v = np.zeros(max(n, 1), dtype=np.uint64)
                lib().orc_run_stat_values(h, i, _p(v, u64p))
                stats.append(v[:n])
        finally:
            lib().orc_run_free(h)
        res = RunResult(rows, cig[:nc], skipped[: len(reads)], ctr, secs)
        res.stat_values = stats          # the raw values behind the reference's 16 count histograms + [16] = (completely excluded queries,)
        return res


class RunResult:
    def __init__(self, rows, cigars, skipped, counters, seconds):
        self.rows, self.cigars, self.skipped, self.counters, self.seconds = rows, cigars, skipped, counters, seconds

    def records(self):
        """list of (read_index, flag, ref_id, pos, nm, cigar string)"""
        out = []
        for r in self.rows:
            out.append((int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4]), cigar_str(self.cigars[r[5]: r[5] + r[6]])))
        return out


def erase_useless(rows):
    rows = np.ascontiguousarray(np.asarray(rows, dtype=np.uint64).reshape(-1, 2))
    k = lib().orc_erase_useless(_p(rows, u64p), len(rows))
    return rows[:k].copy()


def align(ref, query, k, mode=2, algo=1):
    """mode 0 exists / 1 without cigar / 2 with cigar. Returns None or (nm, begin, cigar string)."""
    ref, query = as_u8(ref), as_u8(query)
    nm, begin = C.c_uint64(), C.c_uint64()
    cap = len(ref) + len(query) + 8
    cig = np.zeros(cap, dtype=np.uint32)
    clen = C.c_uint64(cap)
    ok = lib().orc_align(_p(ref, u8p), len(ref), _p(query, u8p), len(query), k, mode, algo, C.byref(nm), C.byref(begin),
                         _p(cig, u32p), C.byref(clen))
    if not ok:
        return None
    return nm.value, begin.value, cigar_str(cig[: clen.value])


def span(anchor_pos, node_from, node_to, node_errors, leaf_from, reflen, ratio):
    out = np.zeros(3, dtype=np.uint64)
    lib().orc_span(anchor_pos, node_from, node_to, node_errors, leaf_from, reflen, ratio, _p(out, u64p))
    return tuple(int(x) for x in out)


class Intervals:
    def __init__(self, active=True):
        self.h = lib().orc_intervals_new(int(active))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_intervals_free(self.h)
            self.h = None

    def insert(self, s, e):
        lib().orc_intervals_insert(self.h, s, e)

    def contains(self, s, e):
        return bool(lib().orc_intervals_contains(self.h, s, e))

    def __len__(self):
        return lib().orc_intervals_count(self.h)


def verify_anchor(qlen, k, s, bottom_up, leaf_index, anchor_pos, anchor_errors, query, reverse, reference, pv, intervals=None):
    query, reference = as_u8(query), as_u8(reference)
    rows = np.zeros((8, 5), dtype=np.uint64)
    cig = np.zeros(4 * (len(query) + len(reference)) + 16, dtype=np.uint32)
    pv = np.ascontiguousarray(pv, dtype=np.float64)
    n = lib().orc_verify_anchor(qlen, k, s, int(bottom_up), leaf_index, anchor_pos, anchor_errors, _p(query, u8p), int(reverse),
                                _p(reference, u8p), len(reference), _p(pv, f64p), intervals.h if intervals is not None else None,
                                _p(rows, u64p), 8, _p(cig, u32p), len(cig))
    out = []
    for i in range(n):
        out.append((int(rows[i, 0]), int(rows[i, 1]), bool(rows[i, 2]), cigar_str(cig[rows[i, 3]: rows[i, 3] + rows[i, 4]])))
    return out
