"""ctypes loader for libbfc_gpu.so (the C ABI of include/bfc_gpu.h).

The library is the product: if it is missing or cannot be loaded this module raises -- there is
no Python or CPU fallback for the counting path.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("BFC_GPU_LIB") or os.path.join(HERE, "libbfc_gpu.so")  # BFC_GPU_LIB: A/B a differently built library

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
f32p = C.POINTER(C.c_float)


class BfcOpt(C.Structure):
    """bfc_opt_t (bfc.h:15-33)."""
    _fields_ = [("chunk_size", C.c_int), ("n_threads", C.c_int), ("no_mt_io", C.c_int), ("q", C.c_int), ("k", C.c_int),
                ("filter_mode", C.c_int), ("refine_ec", C.c_int), ("no_qual", C.c_int), ("min_frac", C.c_float),
                ("l_pre", C.c_int), ("bf_shift", C.c_int), ("n_hashes", C.c_int), ("discard", C.c_int),
                ("max_end_ext", C.c_int), ("win_multi_ec", C.c_int), ("min_cov", C.c_int),
                ("w_ec", C.c_int), ("w_ec_high", C.c_int), ("w_absent", C.c_int), ("w_absent_high", C.c_int),
                ("max_path_diff", C.c_int), ("max_heap", C.c_int)]


class BfcBf(C.Structure):
    """bfc_bf_t (bbf.h:9-12)."""
    _fields_ = [("n_shift", C.c_int), ("n_hashes", C.c_int), ("b", C.POINTER(C.c_uint8))]


class BfcKmer(C.Structure):
    _fields_ = [("x", C.c_uint64 * 4)]


class BfcgParams(C.Structure):
    _fields_ = [("k", C.c_int), ("q", C.c_int), ("bf_shift", C.c_int), ("n_hashes", C.c_int), ("l_pre", C.c_int),
                ("filter_mode", C.c_int), ("device", C.c_int), ("max_batch_pos", C.c_uint64),
                ("region_shift", C.c_int), ("tab_cshift", C.c_int), ("debug_seen", C.c_int), ("track_order", C.c_int), ("rank", C.c_int), ("n_ranks", C.c_int), ("table_layout", C.c_int)]


# every symbol include/bfc_gpu.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "bfc_bf_init": (C.POINTER(BfcBf), [C.c_int, C.c_int]),
    "bfc_bf_destroy": (None, [C.POINTER(BfcBf)]),
    "bfc_bf_insert": (C.c_int, [C.POINTER(BfcBf), C.c_uint64]),
    "bfc_bf_get": (C.c_int, [C.POINTER(BfcBf), C.c_uint64]),
    "bfc_ch_init": (C.c_void_p, [C.c_int, C.c_int]),
    "bfc_ch_destroy": (None, [C.c_void_p]),
    "bfc_ch_insert": (C.c_int, [C.c_void_p, u64p, C.c_int, C.c_int]),
    "bfc_ch_get": (C.c_int, [C.c_void_p, u64p]),
    "bfc_ch_count": (C.c_uint64, [C.c_void_p]),
    "bfc_ch_hist": (C.c_int, [C.c_void_p, u64p, u64p]),
    "bfc_ch_dump": (C.c_int, [C.c_void_p, C.c_char_p]),
    "bfc_ch_restore": (C.c_void_p, [C.c_char_p]),
    "bfc_ch_get_k": (C.c_int, [C.c_void_p]),
    "bfc_ch_kmer_occ": (C.c_int, [C.c_void_p, C.POINTER(BfcKmer)]),
    "bfc_count": (C.c_void_p, [C.c_char_p, C.POINTER(BfcOpt)]),
    "bfc_correct": (None, [C.c_char_p, C.POINTER(BfcOpt), C.c_void_p]),
    "bfcg_params_default": (None, [C.POINTER(BfcgParams)]),
    "bfcg_create": (C.c_void_p, [C.POINTER(BfcgParams)]),
    "bfcg_destroy": (None, [C.c_void_p]),
    "bfcg_last_error": (C.c_char_p, []),
    "bfcg_build_id": (C.c_char_p, []),
    "bfcg_device_count": (C.c_int, []),
    "bfcg_reset": (C.c_int, [C.c_void_p]),
    "bfcg_count_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "bfcg_count_batch_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "bfcg_plane_words": (C.c_uint64, [C.c_uint64]),
    "bfcg_pack_planes": (None, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]),
    "bfcg_count_batch_planes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]),
    "bfcg_sync": (C.c_int, [C.c_void_p]),
    "bfcg_mg_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "bfcg_batch_limit": (C.c_uint64, [C.c_void_p]),
    "bfcg_mg_scatter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, u32p]),
    "bfcg_mg_process": (C.c_int, [C.c_void_p, C.c_void_p, u32p]),
    "bfcg_mg_allow_onepass": (None, [C.c_void_p, C.c_int]),
    "bfcg_group_unique_id": (C.c_int, [C.c_void_p]),
    "bfcg_group_create": (C.c_void_p, [C.POINTER(BfcgParams), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int]),
    "bfcg_group_destroy": (None, [C.c_void_p]),
    "bfcg_group_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "bfcg_group_slab_mode": (C.c_int, [C.c_void_p]),
    "bfcg_group_lazy_batches": (C.c_uint64, [C.c_void_p]),
    "bfcg_group_exchange_bytes": (C.c_int, [C.c_void_p, u64p]),
    "bfcg_group_ctx": (C.c_void_p, [C.c_void_p, C.c_int]),
    "bfcg_group_reset": (C.c_int, [C.c_void_p]),
    "bfcg_group_count_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), u64p]),
    "bfcg_group_count_batch_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "bfcg_group_sync": (C.c_int, [C.c_void_p]),
    "bfcg_group_stats": (C.c_int, [C.c_void_p, u64p]),
    "bfcg_group_progress": (C.c_int, [C.c_void_p, u64p, u64p, u64p, C.c_int]),
    "bfcg_group_export_table": (C.c_void_p, [C.c_void_p]),
    "bfcg_group_export_bloom": (C.POINTER(BfcBf), [C.c_void_p, C.c_int]),
    "bfcg_group_export_bloom_resident": (C.POINTER(BfcBf), [C.c_void_p, C.c_int]),
    "bfcg_trim_create": (C.c_void_p, [C.c_int, C.POINTER(BfcBf), C.c_int, C.c_uint64, C.c_uint64]),
    "bfcg_trim_destroy": (None, [C.c_void_p]),
    "bfcg_trim_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, u64p, C.c_uint64, C.c_float, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "bfcg_trim_last_ms": (C.c_float, [C.c_void_p]),
    "bfcg_trim_adopted": (C.c_int, [C.c_void_p]),
    "bfcg_trim_dev_seq": (C.c_void_p, [C.c_void_p]),
    "bfcg_stream_batches": (C.c_uint64, [C.c_void_p]),
    "bfc_ch_union": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_int]),
    "bfc_ingest_digest": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, u64p]),
    "bfc_ingest_planes_digest": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, u64p]),
    "bfc_pgz_digest": (C.c_int, [C.c_char_p, C.c_int, C.c_uint64, C.c_uint64, u64p]),
    "bfcg_kcov_create": (C.c_void_p, [C.c_void_p, C.c_int, C.c_uint64]),
    "bfcg_kcov_attach": (C.c_void_p, [C.c_void_p, C.c_uint64]),
    "bfcg_kcov_destroy": (None, [C.c_void_p]),
    "bfcg_kcov_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]),
    "bfcg_kcov_last_ms": (C.c_float, [C.c_void_p]),
    "bfcg_kcov_dev_seq": (C.c_void_p, [C.c_void_p]),
    "bfcg_kcov_dev_out": (C.c_void_p, [C.c_void_p]),
    "bfcg_dev_alloc": (C.c_void_p, [C.c_void_p, C.c_uint64]),
    "bfcg_dev_free": (None, [C.c_void_p, C.c_void_p]),
    "bfcg_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "bfcg_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "bfcg_host_alloc": (C.c_void_p, [C.c_uint64]),
    "bfcg_host_free": (None, [C.c_void_p]),
    "bfcg_stats": (C.c_int, [C.c_void_p, u64p]),
    "bfcg_progress": (C.c_int, [C.c_void_p, u64p, u64p, u64p, C.c_int]),
    "bfcg_partition_info": (C.c_int, [C.c_void_p, u64p]),
    "bfcg_s1wc_launches": (C.c_uint64, []),
    "bfcg_table_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "bfcg_last_batch_ms": (C.c_int, [C.c_void_p, f32p]),
    "bfcg_stage_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), u64p, C.c_int]),
    "bfcg_bloom_to_host": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "bfcg_export_bloom": (C.POINTER(BfcBf), [C.c_void_p, C.c_int]),
    "bfcg_export_bloom_resident": (C.POINTER(BfcBf), [C.c_void_p, C.c_int]),
    "bfcg_resident_drop": (None, [C.c_void_p]),
    "bfcg_export_table": (C.c_void_p, [C.c_void_p]),
    "bfc_ch_get_lpre": (C.c_int, [C.c_void_p]),
    "bfc_ch_export_sorted": (C.c_uint64, [C.c_void_p, u32p, u64p]),
    "bfcg_hash_positions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, u64p]),
    "bfcg_seen_flags": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
}

_lib = None


def load():
    """Load libbfc_gpu.so and bind every declared symbol. Raises OSError/AttributeError if anything is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise OSError("libbfc_gpu.so is not built (run `python -m bfc_amd.build`); there is no fallback path")
        L = C.CDLL(SO)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def build_id():
    """The library's embedded identity, e.g. 'src:0123456789abcdef git:abcdef012345'."""
    return load().bfcg_build_id().decode()


def check_build_id():
    """Raise unless the loaded library was built from the sources in this tree."""
    from . import build
    have, want = build_id(), "src:" + build.source_hash()
    if not have.startswith(want):
        raise OSError("libbfc_gpu.so is stale: built from %s, the tree is %s (run `python -m bfc_amd.build`)" % (have, want))
    return have
