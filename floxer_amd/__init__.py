"""floxer_amd — MI355X-native seed-and-verify path of the floxer long-read aligner.

Python mirror of the reference's interfaces for the hot path (same names, argument meaning, error behaviour), over the C ABI of
libfloxer_amd.so:

    pex_tree(config)                       pex::pex_tree            (pex.hpp:57-126)
    fmindex(references)                    fmindex(refs, 4, threads) (floxer.cpp:93-97)
    searcher(ctx, config).search_seeds()   search::searcher         (search.hpp:104-112)
    align(reference, query, config)        alignment::align         (alignment.hpp:73-77), batched as align_batch
    aligner(ctx, params).align_reads()     spawn_search_task + query_verifier::verify + write_alignments_for_query

The compute runs in hand-written HIP kernels; nothing here falls back to a CPU implementation.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import FloxerError, check, lib, ptr, as_u8, u8p, u32p, u64p

CIGAR_OPS = "MIDNSHP=X"
NULL_ID = 0xFFFFFFFF

ORDER = {"errors_first": 0, "count_first": 1, "none": 2}          # search.cpp:59-69
CHOICE = {"round_robin": 0, "full_groups": 1, "first_reported": 2}  # search.cpp:71-81
MODE_EXISTS, MODE_WITHOUT_CIGAR, MODE_WITH_CIGAR = 0, 1, 2


def cigar_string(words):
    return "".join(f"{int(w) >> 4}{CIGAR_OPS[int(w) & 15]}" for w in words)


# ------------------------------------------------------------------------------------------------ math / input
def ceil_div(a, b):
    return lib().flx_ceil_div(a, b)


def floating_point_error_aware_ceil(v):
    return lib().flx_floating_point_error_aware_ceil(float(v))


def saturate_value_to_int32_max(v):
    return lib().flx_saturate_value_to_int32_max(v)


def chars_to_rank_sequence(s):
    b = s.encode() if isinstance(s, str) else bytes(s)
    out = np.zeros(len(b), dtype=np.uint8)
    lib().flx_chars_to_rank_sequence(b, len(b), ptr(out, u8p))
    return out


def reverse_complement_rank(r):
    r = as_u8(r)
    out = np.zeros_like(r)
    lib().flx_reverse_complement_rank(ptr(r, u8p), len(r), ptr(out, u8p))
    return out


# ------------------------------------------------------------------------------------------------ PEX tree
class pex_tree:
    """pex::pex_tree built from (total_query_length, query_num_errors, leaf_max_num_errors, build_strategy)."""

    def __init__(self, total_query_length, query_num_errors, leaf_max_num_errors, bottom_up=False):
        cap = 4 * (query_num_errors + 2) + 16
        nodes = (capi.PexNode * cap)()
        ni, nl = C.c_uint64(), C.c_uint64()
        check(lib().flx_pex_tree_build(total_query_length, query_num_errors, leaf_max_num_errors, int(bottom_up), nodes, cap,
                                       C.byref(ni), C.byref(nl)))
        rows = [(n.parent_id, n.from_, n.to, n.num_errors) for n in nodes[: ni.value + nl.value]]
        self.inner_nodes = rows[: ni.value]
        self.leaves = rows[ni.value:]

    def root(self):
        return self.inner_nodes[0] if self.inner_nodes else self.leaves[0]

    def get_leaves(self):
        return self.leaves

    def generate_seeds(self, step=1):
        """[(offset, length, num_errors, pex_leaf_index)] as pex_tree::generate_seeds (pex.cpp:258-277)."""
        return [(l[1], l[2] - l[1] + 1, l[3], i) for i, l in enumerate(self.leaves)][::step]


# ------------------------------------------------------------------------------------------------ index + context
class fmindex:
    def __init__(self, references=None, path=None, device=None):
        """device: HIP device ordinal to build the suffix arrays on (None: on the host, as flx_index_build)"""
        self.h = C.c_void_p()
        if path is not None and references is None:
            check(lib().flx_index_load(path.encode(), C.byref(self.h)))
        else:
            refs = [as_u8(r) for r in references]
            pool = np.concatenate(refs) if refs else np.zeros(0, np.uint8)
            lens = np.array([len(r) for r in refs], dtype=np.uint64)
            if device is None:
                check(lib().flx_index_build(ptr(pool, u8p), ptr(lens, u64p), len(refs), C.byref(self.h)))
            else:
                check(lib().flx_index_build_on_device(int(device), ptr(pool, u8p), ptr(lens, u64p), len(refs), C.byref(self.h)))

    def save(self, path):
        check(lib().flx_index_save(self.h, path.encode()))

    # ---- the index as an HBM image + a small host part: how a job replicates it across its GPUs (floxer_amd/distributed.py)
    def meta(self):
        n = C.c_uint64(0)
        lib().flx_index_meta_export(self.h, None, C.byref(n))
        buf = np.zeros(n.value, dtype=np.uint8)
        check(lib().flx_index_meta_export(self.h, ptr(buf, u8p), C.byref(n)))
        return buf.tobytes()

    @classmethod
    def from_meta(cls, meta):
        """an index without arrays (sequence starts / lengths and symbol counts only): for context(index, image=...)"""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        buf = np.frombuffer(meta, dtype=np.uint8).copy()
        check(lib().flx_index_meta_import(ptr(buf, u8p), len(buf), C.byref(self.h)))
        return self

    def image_layout(self):
        """bytes of the five device buffers of the index's HBM image"""
        out = np.zeros(5, dtype=np.uint64)
        check(lib().flx_index_image_layout(self.h, ptr(out, u64p)))
        return [int(x) for x in out]

    def image_upload(self, device, pointers):
        arr = (C.c_void_p * 5)(*[C.c_void_p(int(p)) for p in pointers])
        check(lib().flx_index_image_upload(self.h, int(device), arr))

    def __del__(self):
        if getattr(self, "h", None):
            lib().flx_index_free(self.h)
            self.h = None

    @property
    def text_length(self):
        return lib().flx_index_text_length(self.h)

    @property
    def num_references(self):
        return lib().flx_index_num_references(self.h)

    @property
    def device_bytes(self):
        return lib().flx_index_device_bytes(self.h)

    @property
    def derived_device_bytes(self):
        """inverse suffix array + presence filter a context adds to the image when the device has room"""
        return lib().flx_index_derived_device_bytes(self.h)

    def suffix_array(self):
        out = np.zeros(self.text_length, dtype=np.uint64)
        check(lib().flx_index_copy_sa(self.h, ptr(out, u64p)))
        return out

    def suffix_array_u32(self):
        out = np.empty(self.text_length, dtype=np.uint32)
        check(lib().flx_index_copy_sa_u32(self.h, ptr(out, u32p)))
        return out

    def bwt(self, reversed_text=False):
        out = np.zeros(self.text_length, dtype=np.uint8)
        check(lib().flx_index_copy_bwt(self.h, int(reversed_text), ptr(out, u8p)))
        return out


class context:
    """One HIP device + stream + HBM-resident index."""

    def __init__(self, index, device=0, image=None):
        """image: five device buffers holding the index's HBM image (objects with data_ptr(), e.g. torch uint8 tensors; kept alive by
        this object), as uploaded by fmindex.image_upload or received from another rank; None: the context uploads its own."""
        self.index = index
        self.image = image
        self.h = C.c_void_p()
        if image is None:
            check(lib().flx_ctx_create(device, index.h, C.byref(self.h)))
        else:
            arr = (C.c_void_p * 5)(*[C.c_void_p(int(b.data_ptr())) for b in image])
            sizes = np.array([int(b.numel()) * int(b.element_size()) for b in image], dtype=np.uint64)      # what the buffers hold: checked against the layout
            check(lib().flx_ctx_create_on_image(device, index.h, arr, ptr(sizes, u64p), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            lib().flx_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def set_stream(self, hip_stream):
        check(lib().flx_ctx_set_stream(self.h, hip_stream))

    def enable_kernel_timing(self, on=True):
        check(lib().flx_ctx_enable_kernel_timing(self.h, int(on)))

    def reset_kernel_stats(self):
        check(lib().flx_ctx_reset_kernel_stats(self.h))

    def path_counters(self, reset=False):
        """seeds / anchors / requested DP work of all batches since the context was made (or the counters were reset)"""
        pc = capi.PathCounters()
        check(lib().flx_ctx_get_path_counters(self.h, C.byref(pc)))
        if reset:
            check(lib().flx_ctx_reset_path_counters(self.h))
        return {n: int(getattr(pc, n)) for n, _ in capi.PathCounters._fields_ if n != "reserved"}

    def kernel_stats(self):
        arr = (capi.KernelStat * 32)()
        n = C.c_uint32(32)
        check(lib().flx_ctx_get_kernel_stats(self.h, arr, C.byref(n)))
        return {a.name.decode(): dict(launches=a.launches, device_ms=a.device_ms, algorithmic_bytes=a.algorithmic_bytes,
                                      work_units=a.work_units) for a in arr[: n.value]}


class statistics:
    """statistics::search_and_alignment_statistics (statistics.hpp:24-172): attach to a context, align, read the TOML / terminal text"""

    def __init__(self, input_hint=None):
        self.h = C.c_void_p()
        check(lib().flx_stats_create(input_hint.encode() if input_hint else None, C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().flx_stats_free(self.h)
            self.h = None

    def attach(self, ctx):
        check(lib().flx_ctx_set_stats(ctx.h, self.h))
        return self

    @property
    def num_queries(self):
        return lib().flx_stats_num_queries(self.h)

    def format(self, toml=True):
        n = C.c_uint64(0)
        lib().flx_stats_format(self.h, int(toml), None, C.byref(n))
        buf = C.create_string_buffer(n.value)
        check(lib().flx_stats_format(self.h, int(toml), buf, C.byref(n)))
        return buf.value.decode()


# ------------------------------------------------------------------------------------------------ seam 1: searcher
def search_config(max_num_anchors_hard=500, max_num_anchors_soft=50, anchor_group_order="count_first",
                  anchor_choice_strategy="round_robin", erase_useless_anchors=True):
    return capi.SearchConfig(max_num_anchors_hard, max_num_anchors_soft, ORDER[anchor_group_order], CHOICE[anchor_choice_strategy],
                             int(erase_useless_anchors), 0)


class searcher:
    def __init__(self, ctx, config=None):
        self.ctx = ctx
        self.config = config or search_config()

    def _seeds(self, seeds):
        arr = (capi.Seed * len(seeds))()
        for i, (off, ln, err, leaf) in enumerate(seeds):
            arr[i] = capi.Seed(off, ln, err, leaf, 0)
        return arr

    def search_seeds(self, sequence, seeds):
        """seeds: [(offset, length, num_errors, pex_leaf_index)]. Returns (anchors, stats): anchors as an (n,5) uint64 array
        {seed_index, pex_leaf_index, reference_id, reference_position, num_errors} in anchor_iterator order; stats (n_seeds,4)
        {kept_useful, kept_raw, excluded_by_soft_cap, fully_excluded}."""
        seq = as_u8(sequence)
        arr = self._seeds(seeds)
        cap = max(64, len(seeds) * (self.config.max_num_anchors_soft + 1))
        out = (capi.Anchor * cap)()
        n = C.c_uint64(cap)
        stats = (capi.SeedStats * max(1, len(seeds)))()
        check(lib().flx_search_seeds(self.ctx.h, ptr(seq, u8p), len(seq), arr, len(seeds), C.byref(self.config), out, C.byref(n), stats))
        anchors = np.array([(a.seed_index, a.pex_leaf_index, a.reference_id, a.reference_position, a.num_errors) for a in out[: n.value]],
                           dtype=np.uint64).reshape(-1, 5)
        st = np.array([(s.num_kept_useful_anchors, s.num_kept_raw_anchors, s.num_excluded_raw_anchors_by_soft_cap, s.fully_excluded)
                       for s in stats[: len(seeds)]], dtype=np.uint64).reshape(-1, 4)
        return anchors, st

    def search_groups(self, sequence, seeds, max_hits=501):
        """raw search_n emission (kernel K1): (n,4) {seed_index, lb, len, errors} per seed in delegate order."""
        seq = as_u8(sequence)
        arr = self._seeds(seeds)
        cap = 1 << 16
        while True:
            out = (capi.HitGroup * cap)()
            n = C.c_uint64(cap)
            rc = lib().flx_search_groups(self.ctx.h, ptr(seq, u8p), len(seq), arr, len(seeds), max_hits, out, C.byref(n))
            if rc == -3:
                cap = n.value
                continue
            check(rc)
            break
        return np.array([(g.seed_index, g.lb, g.len, g.num_errors) for g in out[: n.value]], dtype=np.uint64).reshape(-1, 4)


# ------------------------------------------------------------------------------------------------ seam 2: align
def align_batch(ctx, query_pool, jobs, reference_pool=None):
    """jobs: [(ref_offset, ref_length, query_offset, query_length, num_allowed_errors, mode)]. reference_pool None = the context's
    reference text (offsets are then positions in the padded concatenated text). Returns a list of None | (nm, begin, cigar)."""
    q = as_u8(query_pool)
    arr = (capi.AlignJob * max(1, len(jobs)))()
    cap_words = 16
    for i, (ro, rl, qo, ql, k, mode) in enumerate(jobs):
        arr[i] = capi.AlignJob(ro, qo, rl, ql, k, mode)
        cap_words += 2 * k + 2
    res = (capi.AlignResult * max(1, len(jobs)))()
    cig = np.zeros(cap_words, dtype=np.uint32)
    words = C.c_uint64(cap_words)
    if reference_pool is None:
        rp, rl_ = None, 0
    else:
        ref = as_u8(reference_pool)
        rp, rl_ = ptr(ref, u8p), len(ref)
    check(lib().flx_align_batch(ctx.h, rp, rl_, ptr(q, u8p), len(q), arr, len(jobs), res, ptr(cig, u32p), C.byref(words)))
    out = []
    for r in res[: len(jobs)]:
        out.append((r.num_errors, r.begin, cigar_string(cig[r.cigar_offset: r.cigar_offset + r.cigar_length])) if r.exists else None)
    return out


def align(ctx, reference, query, num_allowed_errors, mode=MODE_WITH_CIGAR):
    """alignment::align for one (reference window, query) pair."""
    return align_batch(ctx, query, [(0, len(reference), 0, len(query), num_allowed_errors, mode)], reference_pool=reference)[0]


# ------------------------------------------------------------------------------------------------ seam 3: whole path
def params(error_probability=None, query_errors=None, seed_errors=2, max_anchors_hard=500, max_anchors_soft=50,
           anchor_group_order="count_first", anchor_choice_strategy="round_robin", seed_sampling_step_size=1,
           dont_erase_useless_anchors=False, bottom_up_pex_tree=False, interval_optimization=False,
           extra_verification_ratio=0.05, direct_full_verification=False, num_anchors_per_task=3000, without_cigar=False):
    """cli::command_line_input defaults (floxer_cli.hpp:41-70); one of error_probability / query_errors is required."""
    if error_probability is None and query_errors is None:
        raise FloxerError("Either a fixed number of errors in the query or an error probability must be given.")   # floxer_cli.cpp:174
    p = capi.Params()
    lib().flx_params_default(C.byref(p))
    p.query_error_probability = -1.0 if error_probability is None else float(error_probability)
    p.query_num_errors = 0 if query_errors is None else int(query_errors)
    p.pex_seed_num_errors = seed_errors
    p.search = search_config(max_anchors_hard, max_anchors_soft, anchor_group_order, anchor_choice_strategy, not dont_erase_useless_anchors)
    p.seed_sampling_step_size = seed_sampling_step_size
    p.bottom_up_pex_tree_building = int(bottom_up_pex_tree)
    p.use_interval_optimization = int(interval_optimization)
    p.extra_verification_ratio = extra_verification_ratio
    p.direct_full_verification = int(direct_full_verification)
    p.without_cigar = int(without_cigar)
    p.num_anchors_per_verification_task = num_anchors_per_task
    return p


class RunResult:
    """records of one flx_align_reads* call. `raw` is the flx_record array as the C ABI returns it; `rows` is the same as an (n,7)
    int64 matrix {read_index, flag, ref_id, pos, nm, cigar_off, cigar_len}, made on first use."""

    def __init__(self, raw, cigars, skipped):
        self.raw = raw
        self.cigars = cigars
        self.skipped = skipped
        self._rows = None

    @property
    def n_records(self):
        return len(self.raw)

    @property
    def rows(self):
        if self._rows is None:
            raw = self.raw
            self._rows = (np.stack([raw["read"].astype(np.int64), raw["flag"].astype(np.int64), raw["ref"].astype(np.int64),
                                    raw["pos"].astype(np.int64), raw["nm"].astype(np.int64), raw["coff"].astype(np.int64),
                                    raw["clen"].astype(np.int64)], axis=1) if len(raw) else np.zeros((0, 7), dtype=np.int64))
        return self._rows

    def records(self):
        return [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4]), cigar_string(self.cigars[r[5]: r[5] + r[6]])) for r in self.rows]


def _pool_and_offsets(reads):
    if isinstance(reads, tuple):
        pool, offs = as_u8(reads[0]), np.ascontiguousarray(reads[1], dtype=np.uint64)
        n = len(offs) - 1
    else:
        rs = [as_u8(r) for r in reads]
        n = len(rs)
        offs = np.zeros(n + 1, dtype=np.uint64)
        if n:
            offs[1:] = np.cumsum([len(r) for r in rs])
        pool = np.concatenate(rs) if n else np.zeros(0, np.uint8)
    if len(pool) == 0:
        pool = np.zeros(1, np.uint8)
    return pool, offs, n


class resident_reads:
    """A batch of reads uploaded to HBM once (forward + reverse complement); align it any number of times."""

    def __init__(self, ctx, reads):
        pool, offs, n = _pool_and_offsets(reads)
        self.ctx, self.n = ctx, n
        self.h = C.c_void_p()
        check(lib().flx_reads_upload(ctx.h, ptr(pool, u8p), ptr(offs, u64p), n, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            lib().flx_reads_free(self.h)
            self.h = None

    __del__ = close


def _collect_run(run, n):
    try:
        nr = lib().flx_run_num_records(run)
        nc = lib().flx_run_num_cigar_words(run)
        rec_dtype = np.dtype([("read", "<u8"), ("flag", "<u4"), ("ref", "<i4"), ("pos", "<i4"), ("nm", "<u4"), ("coff", "<u8"),
                              ("clen", "<u4"), ("res", "<u4")])
        raw = np.empty(max(1, nr), dtype=rec_dtype)
        cig = np.empty(max(1, nc), dtype=np.uint32)
        skipped = np.zeros(max(1, n), dtype=np.uint8)
        check(lib().flx_run_copy(run, raw.ctypes.data_as(C.POINTER(capi.Record)), ptr(cig, u32p), ptr(skipped, u8p)))
        raw = raw[:nr]
    finally:
        lib().flx_run_free(run)
    return RunResult(raw, cig[:nc], skipped[:n])


class aligner:
    def __init__(self, ctx, p):
        self.ctx, self.params = ctx, p

    def align_reads(self, reads):
        """reads: list of rank arrays, (pool, offsets), or resident_reads. Returns RunResult with records in --threads 1 order."""
        run = C.c_void_p()
        if isinstance(reads, resident_reads):
            check(lib().flx_align_reads_resident(self.ctx.h, C.byref(self.params), reads.h, C.byref(run)))
            return _collect_run(run, reads.n)
        pool, offs, n = _pool_and_offsets(reads)
        check(lib().flx_align_reads(self.ctx.h, C.byref(self.params), ptr(pool, u8p), ptr(offs, u64p), n, C.byref(run)))
        return _collect_run(run, n)
