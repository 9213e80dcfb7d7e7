"""ctypes binding of libfloxer_amd.so (include/floxer_amd.h). The library holds the whole path (HIP kernels + C++ host);
this module only marshals numpy arrays. There is no CPU fallback: without the built library or without a GPU every
device entry point raises FloxerError."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLX_LIBRARY") or os.path.join(_HERE, "libfloxer_amd.so")      # (FLX_LIBRARY: another build of the library, for A/B runs on one box)

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


class FloxerError(RuntimeError):
    pass


class PexNode(C.Structure):
    _fields_ = [("parent_id", C.c_uint32), ("from_", C.c_uint32), ("to", C.c_uint32), ("num_errors", C.c_uint32)]


class Seed(C.Structure):
    _fields_ = [("seq_offset", C.c_uint64), ("length", C.c_uint32), ("num_errors", C.c_uint32), ("pex_leaf_index", C.c_uint32),
                ("reserved", C.c_uint32)]


class SearchConfig(C.Structure):
    _fields_ = [("max_num_anchors_hard", C.c_uint64), ("max_num_anchors_soft", C.c_uint64), ("anchor_group_order", C.c_int32),
                ("anchor_choice_strategy", C.c_int32), ("erase_useless_anchors", C.c_int32), ("reserved", C.c_int32)]


class Anchor(C.Structure):
    _fields_ = [("seed_index", C.c_uint32), ("pex_leaf_index", C.c_uint32), ("reference_id", C.c_uint32), ("num_errors", C.c_uint32),
                ("reference_position", C.c_uint64)]


class SeedStats(C.Structure):
    _fields_ = [("num_kept_useful_anchors", C.c_uint32), ("num_kept_raw_anchors", C.c_uint32),
                ("num_excluded_raw_anchors_by_soft_cap", C.c_uint32), ("fully_excluded", C.c_uint32)]


class HitGroup(C.Structure):
    _fields_ = [("seed_index", C.c_uint32), ("lb", C.c_uint32), ("len", C.c_uint32), ("num_errors", C.c_uint32)]


class AlignJob(C.Structure):
    _fields_ = [("ref_offset", C.c_uint64), ("query_offset", C.c_uint64), ("ref_length", C.c_uint32), ("query_length", C.c_uint32),
                ("num_allowed_errors", C.c_uint32), ("mode", C.c_uint32)]


class AlignResult(C.Structure):
    _fields_ = [("exists", C.c_uint32), ("num_errors", C.c_uint32), ("begin", C.c_uint64), ("cigar_offset", C.c_uint64),
                ("cigar_length", C.c_uint32), ("reserved", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("query_error_probability", C.c_double), ("query_num_errors", C.c_uint64), ("pex_seed_num_errors", C.c_uint64),
                ("search", SearchConfig), ("seed_sampling_step_size", C.c_uint64), ("bottom_up_pex_tree_building", C.c_int32),
                ("use_interval_optimization", C.c_int32), ("extra_verification_ratio", C.c_double),
                ("direct_full_verification", C.c_int32), ("without_cigar", C.c_int32),
                ("num_anchors_per_verification_task", C.c_uint64)]


class Record(C.Structure):
    _fields_ = [("read_index", C.c_uint64), ("flag", C.c_uint32), ("reference_id", C.c_int32), ("position", C.c_int32),
                ("num_errors", C.c_uint32), ("cigar_offset", C.c_uint64), ("cigar_length", C.c_uint32), ("reserved", C.c_uint32)]


class PathCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("seeds", "seeds_with_anchors", "seeds_excluded_by_hard_cap", "seeds_selected_on_host", "anchors",
                                          "cursor_extensions", "inner_tests_requested", "root_alignments_requested",
                                          "root_alignments_found", "records", "reads", "search_reruns")] + [("reserved", C.c_uint64 * 4)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint64), ("device_ms", C.c_double), ("algorithmic_bytes", C.c_uint64),
                ("work_units", C.c_uint64)]


# every symbol include/floxer_amd.h declares (tests check that the library exports all of them)
EXPORTED = [
    "flx_last_error", "flx_version", "flx_ceil_div", "flx_floating_point_error_aware_ceil", "flx_saturate_value_to_int32_max",
    "flx_chars_to_rank_sequence", "flx_reverse_complement_rank", "flx_pex_tree_build", "flx_index_build", "flx_index_build_on_device", "flx_index_save",
    "flx_index_load", "flx_index_free", "flx_index_text_length", "flx_index_num_references", "flx_index_device_bytes", "flx_index_derived_device_bytes",
    "flx_index_copy_sa", "flx_index_copy_sa_u32", "flx_index_copy_bwt", "flx_ctx_create", "flx_ctx_destroy", "flx_ctx_set_stream", "flx_search_seeds",
    "flx_search_groups", "flx_align_batch", "flx_params_default", "flx_align_reads", "flx_reads_upload", "flx_reads_free",
    "flx_align_reads_resident", "flx_run_num_records",
    "flx_run_num_cigar_words", "flx_run_copy", "flx_run_free", "flx_ctx_enable_kernel_timing", "flx_ctx_reset_kernel_stats",
    "flx_ctx_get_kernel_stats", "flx_sam_open", "flx_sam_write", "flx_sam_close", "flx_sim_genome", "flx_sim_genome_repeats", "flx_sim_reads", "flx_ctx_get_path_counters",
    "flx_ctx_reset_path_counters", "flx_stats_create", "flx_stats_free", "flx_stats_merge", "flx_stats_num_queries", "flx_stats_format",
    "flx_ctx_set_stats", "flx_device_count", "flx_index_matches_reference", "flx_sam_set_threads", "flx_index_image_layout",
    "flx_index_image_upload", "flx_index_meta_export", "flx_index_meta_import", "flx_ctx_create_on_image",
]

_lib = None


def _share_torchs_hip_runtime():
    """One HIP runtime per process. A ROCm build of PyTorch brings its own libamdhip64.so (soname libamdhip64.so.7, like the system's)
    and looks it up by path, so a process that loads this library first and torch later ends up with two runtimes, and the second
    one cannot attach to the GPU (KFD gives a process one VM: AMDKFD_IOC_ACQUIRE_VM fails, torch reports "No HIP GPUs are
    available"). Loading torch's copy first makes the dynamic loader hand the same copy to this library (it asks for the soname)
    and to torch whenever it comes. Nothing is imported; FLX_SYSTEM_HIP=1 keeps the system's runtime (then import torch never, or
    before floxer_amd's first call)."""
    if os.environ.get("FLX_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def hip_runtime_paths():
    """the libamdhip64 copies mapped into this process (one, unless something went wrong)"""
    with open("/proc/self/maps") as f:
        return sorted({line.split()[-1] for line in f if "libamdhip64" in line})


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FloxerError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(floxer_amd has no CPU fallback)")
    _share_torchs_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.flx_last_error.restype = C.c_char_p
    L.flx_version.restype = C.c_char_p
    L.flx_ceil_div.restype = C.c_uint64
    L.flx_ceil_div.argtypes = [C.c_uint64, C.c_uint64]
    L.flx_floating_point_error_aware_ceil.restype = C.c_uint64
    L.flx_floating_point_error_aware_ceil.argtypes = [C.c_double]
    L.flx_saturate_value_to_int32_max.restype = C.c_int32
    L.flx_saturate_value_to_int32_max.argtypes = [C.c_uint64]
    L.flx_chars_to_rank_sequence.argtypes = [C.c_char_p, C.c_uint64, u8p]
    L.flx_reverse_complement_rank.argtypes = [u8p, C.c_uint64, u8p]
    L.flx_pex_tree_build.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(PexNode), C.c_uint64, u64p, u64p]
    L.flx_index_build.argtypes = [u8p, u64p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.flx_index_build_on_device.argtypes = [C.c_int, u8p, u64p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.flx_index_save.argtypes = [C.c_void_p, C.c_char_p]
    L.flx_index_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.flx_index_free.argtypes = [C.c_void_p]
    L.flx_index_text_length.restype = C.c_uint64
    L.flx_index_text_length.argtypes = [C.c_void_p]
    L.flx_index_num_references.restype = C.c_uint32
    L.flx_index_num_references.argtypes = [C.c_void_p]
    L.flx_index_device_bytes.restype = C.c_uint64
    L.flx_index_device_bytes.argtypes = [C.c_void_p]
    L.flx_index_derived_device_bytes.restype = C.c_uint64
    L.flx_index_derived_device_bytes.argtypes = [C.c_void_p]
    L.flx_index_copy_sa.argtypes = [C.c_void_p, u64p]
    L.flx_index_copy_sa_u32.argtypes = [C.c_void_p, u32p]
    L.flx_index_copy_bwt.argtypes = [C.c_void_p, C.c_int, u8p]
    L.flx_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.flx_ctx_destroy.argtypes = [C.c_void_p]
    L.flx_index_image_layout.argtypes = [C.c_void_p, u64p]
    L.flx_index_image_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.flx_index_meta_export.argtypes = [C.c_void_p, u8p, u64p]
    L.flx_index_meta_import.argtypes = [u8p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.flx_ctx_create_on_image.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p), u64p, C.POINTER(C.c_void_p)]
    L.flx_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.flx_search_seeds.argtypes = [C.c_void_p, u8p, C.c_uint64, C.POINTER(Seed), C.c_uint64, C.POINTER(SearchConfig),
                                   C.POINTER(Anchor), u64p, C.POINTER(SeedStats)]
    L.flx_search_groups.argtypes = [C.c_void_p, u8p, C.c_uint64, C.POINTER(Seed), C.c_uint64, C.c_uint64, C.POINTER(HitGroup), u64p]
    L.flx_align_batch.argtypes = [C.c_void_p, u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(AlignJob), C.c_uint64,
                                  C.POINTER(AlignResult), u32p, u64p]
    L.flx_params_default.argtypes = [C.POINTER(Params)]
    L.flx_align_reads.argtypes = [C.c_void_p, C.POINTER(Params), u8p, u64p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.flx_reads_upload.argtypes = [C.c_void_p, u8p, u64p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.flx_reads_free.argtypes = [C.c_void_p]
    L.flx_align_reads_resident.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.POINTER(C.c_void_p)]
    L.flx_run_num_records.restype = C.c_uint64
    L.flx_run_num_records.argtypes = [C.c_void_p]
    L.flx_run_num_cigar_words.restype = C.c_uint64
    L.flx_run_num_cigar_words.argtypes = [C.c_void_p]
    L.flx_run_copy.argtypes = [C.c_void_p, C.POINTER(Record), u32p, u8p]
    L.flx_run_free.argtypes = [C.c_void_p]
    L.flx_ctx_enable_kernel_timing.argtypes = [C.c_void_p, C.c_int]
    L.flx_ctx_reset_kernel_stats.argtypes = [C.c_void_p]
    L.flx_ctx_get_kernel_stats.argtypes = [C.c_void_p, C.POINTER(KernelStat), u32p]
    L.flx_sam_open.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), u64p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.flx_sam_write.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), u8p, u64p, C.POINTER(C.c_char_p), C.POINTER(Record), C.c_uint64, u32p]
    L.flx_sam_close.argtypes = [C.c_void_p]
    L.flx_stats_create.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.flx_stats_free.argtypes = [C.c_void_p]
    L.flx_stats_merge.argtypes = [C.c_void_p, C.c_void_p]
    L.flx_stats_num_queries.restype = C.c_uint64
    L.flx_stats_num_queries.argtypes = [C.c_void_p]
    L.flx_stats_format.argtypes = [C.c_void_p, C.c_int, C.c_char_p, u64p]
    L.flx_ctx_set_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.flx_device_count.restype = C.c_int
    L.flx_index_matches_reference.argtypes = [C.c_void_p, u8p, u64p, C.c_uint32]
    L.flx_sam_set_threads.argtypes = [C.c_void_p, C.c_uint32]
    L.flx_ctx_get_path_counters.argtypes = [C.c_void_p, C.POINTER(PathCounters)]
    L.flx_ctx_reset_path_counters.argtypes = [C.c_void_p]
    L.flx_sim_genome.argtypes = [C.c_uint64, C.c_uint64, u8p]
    L.flx_sim_genome_repeats.argtypes = [C.c_uint64, C.c_uint64, u8p]
    L.flx_sim_reads.argtypes = [u8p, u64p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_uint64, u8p, C.c_uint64,
                                u64p, u32p, u64p, u8p]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise FloxerError(f"floxer_amd error {rc}: {lib().flx_last_error().decode()}")


def ptr(a, t):
    return a.ctypes.data_as(t)


def as_u8(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint8))
