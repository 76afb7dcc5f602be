"""ctypes binding of libcobs_gpu.so (the C ABI declared in include/cobs_gpu.h, cobs_gpu_batch.h, cobs_gpu_diag.h, cobs_gpu_construct.h).

The library is the product; this module only declares its entry points.  It
fails loudly if the shared object is missing -- there is no Python or CPU
fallback for any of the functions.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# COBS_GPU_LIBRARY selects another build of the same library (the tuning build with phase stamps)
LIB_PATH = os.environ.get("COBS_GPU_LIBRARY") or os.path.join(_HERE, "libcobs_gpu.so")

OK = 0
ERR_OPEN, ERR_FORMAT, ERR_QUERY_TOO_SHORT, ERR_INVALID_BASE, ERR_QUERY_TOO_LONG = 1, 2, 3, 4, 5
ERR_HIP, ERR_ARG, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_NO_DEVICE, ERR_RCCL = 6, 7, 8, 9, 10, 11
XCHG_ALLGATHER, XCHG_ALLTOALL, XCHG_REDUCE = 0, 1, 2
UNIQUE_ID_BYTES = 128

STATUS_NAMES = {
    0: "COBS_GPU_OK", 1: "COBS_GPU_ERR_OPEN", 2: "COBS_GPU_ERR_FORMAT",
    3: "COBS_GPU_ERR_QUERY_TOO_SHORT", 4: "COBS_GPU_ERR_INVALID_BASE",
    5: "COBS_GPU_ERR_QUERY_TOO_LONG", 6: "COBS_GPU_ERR_HIP", 7: "COBS_GPU_ERR_ARG",
    8: "COBS_GPU_ERR_UNSUPPORTED", 9: "COBS_GPU_ERR_CAPACITY", 10: "COBS_GPU_ERR_NO_DEVICE",
    11: "COBS_GPU_ERR_RCCL",
}


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32),
                ("shard_rank", C.c_uint32), ("shard_count", C.c_uint32),
                ("waves_per_group", C.c_uint32), ("shard_mode", C.c_uint32),
                ("hbm_budget_bytes", C.c_uint64)]


class IndexInfo(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("term_size", C.c_uint32), ("canonicalize", C.c_uint32),
                ("num_pages", C.c_uint32), ("num_hashes", C.c_uint64), ("page_size", C.c_uint64),
                ("row_size", C.c_uint64), ("counts_size", C.c_uint64), ("num_docs", C.c_uint64),
                ("doc_offset", C.c_uint64), ("hbm_bytes", C.c_uint64),
                ("first_page", C.c_uint32), ("end_page", C.c_uint32),
                ("slot_begin", C.c_uint64), ("slot_count", C.c_uint64), ("local_offset", C.c_uint64)]


class Hit(C.Structure):
    _fields_ = [("file_no", C.c_uint32), ("doc", C.c_uint32), ("score", C.c_uint32)]


class BuildParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("term_size", C.c_uint32), ("canonicalize", C.c_uint32),
                ("num_hashes", C.c_uint32), ("false_positive_rate", C.c_double),
                ("signature_size", C.c_uint64), ("page_size", C.c_uint64),
                ("device", C.c_int32), ("text_batch_bytes", C.c_uint32), ("doc_terms", C.POINTER(C.c_uint64)),
                ("set_bits_mode", C.c_uint32), ("reserved", C.c_uint32)]


class DocEntry(C.Structure):
    _fields_ = [("path", C.c_char_p), ("name", C.c_char_p), ("type", C.c_uint32), ("reserved", C.c_uint32),
                ("size", C.c_uint64), ("subdoc_index", C.c_uint64), ("term_size", C.c_uint64),
                ("term_count", C.c_uint64)]


class Xfer(C.Structure):
    _fields_ = [("peer", C.c_uint64), ("send_offset", C.c_uint64), ("send_bytes", C.c_uint64),
                ("recv_offset", C.c_uint64), ("recv_bytes", C.c_uint64)]


class Copy2D(C.Structure):
    _fields_ = [("src_rank", C.c_uint64), ("src_is_local", C.c_uint64), ("src_offset", C.c_uint64),
                ("src_pitch", C.c_uint64), ("dst_offset", C.c_uint64), ("dst_pitch", C.c_uint64),
                ("width", C.c_uint64), ("height", C.c_uint64)]


class Synth(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("term_size", C.c_uint32), ("canonicalize", C.c_uint32),
                ("num_pages", C.c_uint32), ("num_hashes", C.c_uint64), ("page_size", C.c_uint64),
                ("num_docs", C.c_uint64), ("seed", C.c_uint64),
                ("signature_sizes", C.POINTER(C.c_uint64))]


# name -> (restype, argtypes): every symbol the four headers under include/ declare
_vp, _sz, _u32, _u64, _cp, _dbl, _int = (C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64,
                                        C.c_char_p, C.c_double, C.c_int)
_pu64 = C.POINTER(C.c_uint64)
SYMBOLS = {
    "cobs_gpu_abi_version": (_u32, []),
    "cobs_gpu_last_error": (_cp, []),
    "cobs_gpu_device_count": (_int, []),
    "cobs_gpu_open": (_int, [C.POINTER(_cp), _sz, C.POINTER(Options), C.POINTER(_vp)]),
    "cobs_gpu_open_synthetic": (_int, [C.POINTER(Synth), C.POINTER(Options), C.POINTER(_vp)]),
    "cobs_gpu_plant": (_int, [_vp, _sz, _cp, _sz, C.POINTER(_u32), C.POINTER(_u32), _sz, _u64]),
    "cobs_gpu_close": (None, [_vp]),
    "cobs_gpu_set_tuning": (_int, [_vp, _cp, C.c_int64]),
    "cobs_gpu_plan_shards": (_int, [_cp, _u32, _u32, _pu64, _pu64, _pu64]),
    "cobs_gpu_page_columns": (_int, [_vp, _sz, _u32, _pu64, _pu64]),
    "cobs_gpu_num_files": (_sz, [_vp]),
    "cobs_gpu_info": (_int, [_vp, _sz, C.POINTER(IndexInfo)]),
    "cobs_gpu_signature_size": (_u64, [_vp, _sz, _u32]),
    "cobs_gpu_doc_name": (_cp, [_vp, _sz, _u64]),
    "cobs_gpu_total_counts": (_u64, [_vp]),
    "cobs_gpu_local_counts": (_u64, [_vp]),
    "cobs_gpu_read_row": (_int, [_vp, _sz, _u32, _u64, _vp, _sz]),
    "cobs_gpu_read_rows": (_int, [_vp, _sz, _u32, _u64, _u64, _vp, _sz]),
    "cobs_gpu_build_classic": (_int, [C.POINTER(_cp), C.POINTER(_cp), C.POINTER(_sz), _sz,
                                      C.POINTER(BuildParams), _cp]),
    "cobs_gpu_build_compact": (_int, [C.POINTER(_cp), C.POINTER(_cp), C.POINTER(_sz), _sz,
                                      C.POINTER(BuildParams), _cp]),
    "cobs_gpu_write_synthetic": (_int, [C.POINTER(Synth), _cp, _int]),
    "cobs_gpu_build_index": (_int, [_u32, C.POINTER(_cp), C.POINTER(_cp), C.POINTER(_sz), _sz,
                                    C.POINTER(BuildParams), C.POINTER(Options), C.POINTER(_vp)]),
    "cobs_gpu_doclist_create": (_int, [C.POINTER(_vp)]),
    "cobs_gpu_doclist_free": (None, [_vp]),
    "cobs_gpu_doclist_add": (_int, [_vp, _cp]),
    "cobs_gpu_doclist_add_recursive": (_int, [_vp, _cp, _u32]),
    "cobs_gpu_doclist_add_memory": (_int, [_vp, _cp, _cp, _sz]),
    "cobs_gpu_doclist_size": (_sz, [_vp]),
    "cobs_gpu_doclist_entry": (_int, [_vp, _sz, C.POINTER(DocEntry)]),
    "cobs_gpu_doclist_sort": (_int, [_vp, _u32]),
    "cobs_gpu_doclist_num_terms": (_int, [_vp, _sz, _u32, C.POINTER(_u64)]),
    "cobs_gpu_doclist_terms": (_int, [_vp, _sz, _u32, _vp, _sz, C.POINTER(_u64)]),
    "cobs_gpu_filetype_from_string": (_int, [_cp, C.POINTER(_u32)]),
    "cobs_gpu_build_classic_list": (_int, [_vp, C.POINTER(BuildParams), _cp]),
    "cobs_gpu_build_compact_list": (_int, [_vp, C.POINTER(BuildParams), _cp]),
    "cobs_gpu_build_index_list": (_int, [_u32, _vp, C.POINTER(BuildParams), C.POINTER(Options), C.POINTER(_vp)]),
    "cobs_gpu_build_release_buffers": (None, []),
    "cobs_gpu_combine_classic": (_int, [C.POINTER(_cp), _sz, _cp, _u64, _int]),
    "cobs_gpu_combine_compact": (_int, [C.POINTER(_cp), _sz, _cp, _u64]),
    "cobs_gpu_construct_random": (_int, [_cp, _u64, _u64, _u64, _u64, _u64, _int]),
    "cobs_gpu_search": (_int, [_vp, _cp, _sz, _dbl, _sz, C.POINTER(Hit), _sz, C.POINTER(_sz)]),
    "cobs_gpu_search_batch": (_int, [_vp, C.POINTER(_cp), C.POINTER(_sz), _sz, _dbl, _sz,
                                     C.POINTER(Hit), _sz, C.POINTER(_sz), C.POINTER(_sz)]),
    "cobs_gpu_search_batch_view": (_int, [_vp, C.POINTER(_cp), C.POINTER(_sz), _sz, _dbl, _sz,
                                          C.POINTER(C.POINTER(Hit)), C.POINTER(C.POINTER(_sz)), C.POINTER(_sz)]),
    "cobs_gpu_counts": (_int, [_vp, _cp, _sz, _vp, _sz]),
    "cobs_gpu_batch_create": (_int, [_vp, _sz, _sz, C.POINTER(_vp)]),
    "cobs_gpu_batch_destroy": (None, [_vp]),
    "cobs_gpu_batch_set_queries": (_int, [_vp, C.POINTER(_cp), C.POINTER(_sz), _sz]),
    "cobs_gpu_batch_run": (_int, [_vp, _dbl, _vp]),
    "cobs_gpu_batch_run_hits": (_int, [_vp, _dbl, _vp]),
    "cobs_gpu_batch_run_topk": (_int, [_vp, _dbl, _sz, _vp]),
    "cobs_gpu_batch_run_topk_only": (_int, [_vp, _dbl, _sz, _vp]),
    "cobs_gpu_batch_sync": (_int, [_vp, _vp, C.POINTER(_sz)]),
    "cobs_gpu_batch_counts_device": (_vp, [_vp, C.POINTER(_u32), C.POINTER(_u64)]),
    "cobs_gpu_batch_counts_host": (_int, [_vp, _sz, _vp, _sz]),
    "cobs_gpu_batch_score_histogram": (_int, [_vp, _pu64, _sz]),
    "cobs_gpu_batch_hits_host": (_int, [_vp, _sz, _sz, C.POINTER(Hit), _sz, C.POINTER(_sz)]),
    "cobs_gpu_batch_stats": (_int, [_vp, C.POINTER(_u64 * 4)]),
    "cobs_gpu_batch_kernel_ms": (_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "cobs_gpu_batch_phase_stamps": (_int, [_vp, _pu64, _sz, C.POINTER(_sz)]),
    "cobs_gpu_multi_open": (_int, [C.POINTER(_cp), _sz, C.POINTER(_int), _sz, C.POINTER(Options), C.POINTER(_vp)]),
    "cobs_gpu_multi_close": (None, [_vp]),
    "cobs_gpu_multi_size": (_sz, [_vp]),
    "cobs_gpu_multi_index": (_vp, [_vp, _sz]),
    "cobs_gpu_multi_search_batch": (_int, [_vp, C.POINTER(_cp), C.POINTER(_sz), _sz, _dbl, _sz,
                                           C.POINTER(Hit), _sz, C.POINTER(_sz), C.POINTER(_sz)]),
    "cobs_gpu_graph_replays": (_u64, [_vp]),
    "cobs_gpu_host_passes": (_u64, [_vp]),
    "cobs_gpu_stream_counters": (_int, [_vp, C.POINTER(C.c_uint64 * 2)]),
    "cobs_gpu_stream_traffic": (_int, [_vp, C.POINTER(C.c_uint64 * 4)]),
    "cobs_gpu_stream_plan": (_int, [_vp, C.POINTER(C.c_uint64 * 4)]),
    "cobs_gpu_plan_stream": (_int, [_cp, _u64, _u32, _u32, _u32, C.POINTER(C.c_uint64 * 4), C.POINTER(C.c_uint8), _sz, C.POINTER(_sz)]),
    "cobs_gpu_timers": (_int, [_vp, C.POINTER(C.c_double * 5), _int]),
    "cobs_gpu_exchange_plan": (_int, [_pu64, _pu64, _pu64, _sz, _sz, _u64, _sz, _u32, _u32, _sz,
                                      C.POINTER(Xfer), C.POINTER(Copy2D), C.POINTER(_sz), _pu64]),
    "cobs_gpu_comm_unique_id": (_int, [_vp]),
    "cobs_gpu_comm_create": (_int, [_vp, _int, _int, _int, C.POINTER(_vp)]),
    "cobs_gpu_comm_destroy": (None, [_vp]),
    "cobs_gpu_comm_rank": (_int, [_vp]),
    "cobs_gpu_comm_size": (_int, [_vp]),
    "cobs_gpu_comm_set_timeout": (None, [_vp, _u32]),
    "cobs_gpu_comm_state": (_sz, [_vp, C.c_char_p, _sz]),
    "cobs_gpu_comm_preflight": (_int, [_vp, _u32, _u64, C.POINTER(C.c_uint64 * 8)]),
    "cobs_gpu_batch_exchange_counts": (_int, [_vp, _vp, _u32, _vp]),
    "cobs_gpu_batch_global_counts_device": (_vp, [_vp, _pu64, _pu64, C.POINTER(_u32), _pu64]),
    "cobs_gpu_batch_exchange_bytes": (_u64, [_vp]),
    "cobs_gpu_batch_exchange_hits": (_int, [_vp, _vp, _vp, C.POINTER(_int)]),
    "cobs_gpu_batch_exchange_hits_owned": (_int, [_vp, _vp, _vp, C.POINTER(_int), _pu64, _pu64]),
    "cobs_gpu_batch_bucketed_hits": (_int, [_vp, _u32, _pu64, C.POINTER(_u32), _sz, C.POINTER(_sz)]),
    "cobs_gpu_hit_exchange_plan": (_int, [_pu64, _sz, _sz, C.POINTER(Xfer), _pu64]),
    "cobs_gpu_batch_exchange_topk": (_int, [_vp, _vp, _vp]),
    "cobs_gpu_sharded_search_batch": (_int, [_vp, _vp, C.POINTER(_cp), C.POINTER(_sz), _sz, _dbl, _sz,
                                             C.POINTER(Hit), _sz, C.POINTER(_sz), C.POINTER(_sz)]),
    "cobs_gpu_sharded_search_batch_split": (_int, [_vp, _vp, C.POINTER(_cp), C.POINTER(_sz), _sz, _dbl, _sz,
                                                   C.POINTER(Hit), _sz, C.POINTER(_sz), C.POINTER(_sz)]),
    "cobs_gpu_sharded_batch_create": (_int, [_vp, _vp, _u32, C.POINTER(_vp)]),
    "cobs_gpu_sharded_batch_destroy": (None, [_vp]),
    "cobs_gpu_sharded_batch_set_queries": (_int, [_vp, C.POINTER(_cp), C.POINTER(_sz), _sz]),
    "cobs_gpu_sharded_batch_step": (_int, [_vp, _dbl, _u32]),
    "cobs_gpu_sharded_batch_sync": (_int, [_vp, C.POINTER(_sz)]),
    "cobs_gpu_sharded_batch_subs": (_sz, [_vp]),
    "cobs_gpu_sharded_batch_sub": (_vp, [_vp, _sz, C.POINTER(_sz), C.POINTER(_sz)]),
    "cobs_gpu_sharded_batch_times": (_int, [_vp, C.POINTER(C.c_double * 8)]),
}

_lib = None


def load():
    """dlopen libcobs_gpu.so and bind every declared symbol (raises if missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "cobs_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C cobs_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and, when
    # it is going to be used in this process (device memory, streams,
    # torch.distributed), it must be the copy that is loaded first so that
    # libcobs_gpu binds to the same runtime instance (same device pointers).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.cobs_gpu_abi_version() != 2:
        raise ImportError("cobs_amd: libcobs_gpu.so has an unexpected ABI version")
    _lib = lib
    return lib


class CobsGpuError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status


def check(status):
    if status != OK:
        raise CobsGpuError(status, load().cobs_gpu_last_error().decode("utf-8", "replace"))
