"""ctypes binding of the C ABI in include/apss.h (libapss_hip.so).  No fallback: a missing library raises."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
SO_PATH = os.path.join(CSRC, "libapss_hip.so")

OK, E_INVALID, E_NOMEM, E_DEVICE, E_STATE, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5
FLAG_VALUE_PRUNE, FLAG_ADMISSION, FLAG_NORMALIZE, FLAG_FORCE_SCAN, FLAG_FORCE_GENERAL, FLAG_EXACT_ACCUM = 1, 2, 4, 8, 16, 32
FLAG_NO_SYMMETRY = 64

# every symbol include/apss.h declares (tests check the library exports all of them)
SYMBOLS = [
    "apss_create", "apss_destroy", "apss_last_error", "apss_set_stream", "apss_insert", "apss_query",
    "apss_insert_and_query", "apss_self_join", "apss_result_count", "apss_fetch_results", "apss_size",
    "apss_stats_get", "apss_insert_dev", "apss_query_dev", "apss_insert_and_query_dev", "apss_clear",
    "apss_results_dev", "apss_results_copy_dev", "apss_partial_scores_dev", "apss_set_head_terms", "apss_get_head_terms", "apss_set_head_fold",
    "apss_ext_ids_dev",
    # the term-sharded index of one node behind one object (csrc/apss_group.hip)
    "apss_group_create", "apss_group_destroy", "apss_group_last_error", "apss_group_set_term_cuts", "apss_group_insert",
    "apss_group_query", "apss_group_insert_and_query", "apss_group_insert_and_query_dev", "apss_group_clear",
    "apss_group_result_count", "apss_group_fetch_results", "apss_group_stats_get", "apss_group_member_stats",
]
GROUP_FORCE_EXCHANGE, GROUP_NO_RCCL = 1, 2
EXCHANGE_NONE, EXCHANGE_COPIES, EXCHANGE_RCCL = 0, 1, 2
GROUP_MAX_MEMBERS = 64
DOWNGRADE_ACC8, DOWNGRADE_HEAD = 1, 2
SYM_RAN, SYM_FLAG, SYM_NOT_WHOLE, SYM_PATH, SYM_LONG_ROWS, SYM_ONE_TILE, SYM_CHUNK = 0, 1, 2, 3, 4, 5, 6


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("dim", C.c_int32), ("theta", C.c_double),
                ("index_threshold", C.c_double), ("flags", C.c_uint32), ("device_id", C.c_int32),
                ("term_lo", C.c_int32), ("term_hi", C.c_int32), ("tile_rows", C.c_int32),
                ("head_terms", C.c_int32), ("capacity_rows", C.c_int64), ("capacity_nnz", C.c_int64)]


class Stats(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("symmetric", C.c_uint32), ("rows", C.c_int64), ("nnz", C.c_int64), ("tiles", C.c_int64), ("posting_visits", C.c_int64),
                ("candidate_pairs", C.c_int64), ("result_pairs", C.c_int64), ("probe_ms", C.c_double),
                ("build_ms", C.c_double), ("probe_launches", C.c_int64), ("hbm_bytes", C.c_int64),
                ("filter_survivors", C.c_int64), ("rescore_ms", C.c_double),
                ("head_terms", C.c_int64), ("head_pairs", C.c_int64), ("head_survivors", C.c_int64),
                ("head_ms", C.c_double), ("head_flops", C.c_double), ("thin_launches", C.c_int64),
                ("downgrades", C.c_uint32), ("head_columns", C.c_uint32), ("probe_kernel", C.c_char * 96),
                ("device_posting_visits", C.c_int64), ("symmetric_declined", C.c_uint32), ("query_chunk", C.c_int32),
                ("filter_tile_rows", C.c_int32), ("head_int8", C.c_int32),
                ("queries_per_round", C.c_int32), ("reserved0", C.c_int32)]


class GroupStats(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("n_members", C.c_int32), ("exchange", C.c_int32), ("head_terms", C.c_int32),
                ("rows", C.c_int64), ("nnz", C.c_int64), ("posting_visits", C.c_int64), ("device_posting_visits", C.c_int64),
                ("member_touched_pairs", C.c_int64), ("candidates_sum", C.c_int64), ("candidates_max", C.c_int64),
                ("union_pairs", C.c_int64), ("result_pairs", C.c_int64), ("all_gather_bytes", C.c_int64),
                ("all_reduce_bytes", C.c_int64), ("member_ms_max", C.c_double), ("probe_ms_max", C.c_double),
                ("build_ms_max", C.c_double), ("head_ms_max", C.c_double), ("exchange_ms", C.c_double),
                ("partial_ms_max", C.c_double), ("total_ms", C.c_double), ("term_cuts", C.c_int32 * (GROUP_MAX_MEMBERS + 1))]


def build_sources():
    """every file libapss_hip.so is compiled from: the translation unit, every header beside it, the ABI header"""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))) + [
        os.path.normpath(os.path.join(CSRC, "..", "..", "include", "apss.h"))]


def build(force=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  In-tree output, travels with gpurun."""
    srcs = build_sources()
    if not force and os.path.exists(SO_PATH) and os.path.getmtime(SO_PATH) >= max(os.path.getmtime(s) for s in srcs):
        return SO_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # two translation units (the handle: kernels + C ABI of one index; the group: term shards of one node + their exchange);
    # RCCL is loaded at run time (dlopen), never linked
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-o", SO_PATH,
                           os.path.join(CSRC, "apss_hip.hip"), os.path.join(CSRC, "apss_group.hip"), "-ldl", "-lpthread"])
    return SO_PATH


_lib = None


def hip_runtimes_loaded():
    """paths of the libamdhip64 images mapped into this process (there must never be two)"""
    with open("/proc/self/maps") as f:
        return sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln})


def _one_hip_runtime():
    """ONE HIP runtime per process, whatever the import order.  PyTorch-ROCm bundles its own libamdhip64 (soname
    libamdhip64.so.7, found through libtorch_hip's RPATH); libapss_hip.so asks for the same soname.  The dynamic loader
    reuses an image whose soname matches, so: if torch is installed, its bundled runtime is mapped FIRST (by path, without
    importing torch) -- libapss_hip.so then binds to it, and a later `import torch` finds its own file already mapped.
    Without torch (the C++ host mirror, the JVM) the system runtime of /opt/rocm is the only one there is."""
    import importlib.util
    if hip_runtimes_loaded():
        return  # torch (or another HIP user) came first: libapss_hip.so binds to that image by soname
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    bundled = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)


def lib():
    """Loads libapss_hip.so; raises if it has not been built (there is no CPU path to fall back to)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError("libapss_hip.so is not built (%s): run __graft_entry__.build() / make -C %s. "
                           "There is no CPU fallback." % (SO_PATH, CSRC))
    _one_hip_runtime()
    L = C.CDLL(SO_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
    pi64 = C.POINTER(C.c_int64)
    L.apss_create.restype = i32
    L.apss_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.apss_destroy.restype = None
    L.apss_destroy.argtypes = [vp]
    L.apss_last_error.restype = C.c_char_p
    L.apss_last_error.argtypes = [vp]
    L.apss_set_stream.restype = i32
    L.apss_set_stream.argtypes = [vp, vp, i32]
    for name in ("apss_insert",):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp, i64, vp, vp, vp, vp]
    for name in ("apss_query", "apss_insert_and_query"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp, i64, vp, vp, vp, vp, pi64]
    L.apss_self_join.restype = i32
    L.apss_self_join.argtypes = [vp, pi64]
    L.apss_result_count.restype = i32
    L.apss_result_count.argtypes = [vp, pi64]
    L.apss_fetch_results.restype = i32
    L.apss_fetch_results.argtypes = [vp, i64, i64, vp, vp, vp]
    L.apss_size.restype = i32
    L.apss_size.argtypes = [vp, pi64, pi64]
    L.apss_stats_get.restype = i32
    L.apss_stats_get.argtypes = [vp, C.POINTER(Stats)]
    L.apss_insert_dev.restype = i32
    L.apss_insert_dev.argtypes = [vp, i64, i64, vp, vp, vp, vp]
    for name in ("apss_query_dev", "apss_insert_and_query_dev"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp, i64, i64, vp, vp, vp, vp, pi64]
    L.apss_clear.restype = i32
    L.apss_clear.argtypes = [vp]
    L.apss_results_dev.restype = i32
    L.apss_results_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), pi64]
    L.apss_results_copy_dev.restype = i32
    L.apss_results_copy_dev.argtypes = [vp, i64, i64, vp, vp, vp]
    L.apss_partial_scores_dev.restype = i32
    L.apss_partial_scores_dev.argtypes = [vp, i64, vp, vp, vp]
    L.apss_set_head_terms.restype = i32
    L.apss_set_head_terms.argtypes = [vp, i32, vp, i32, i32]
    L.apss_set_head_fold.restype = i32
    L.apss_set_head_fold.argtypes = [vp, i32]
    L.apss_get_head_terms.restype = i32
    L.apss_get_head_terms.argtypes = [vp, i32, vp, C.POINTER(i32)]
    L.apss_ext_ids_dev.restype = i32
    L.apss_ext_ids_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.apss_group_create.restype = i32
    L.apss_group_create.argtypes = [C.POINTER(Config), i32, vp, C.c_uint32, C.POINTER(vp)]
    L.apss_group_destroy.restype = None
    L.apss_group_destroy.argtypes = [vp]
    L.apss_group_last_error.restype = C.c_char_p
    L.apss_group_last_error.argtypes = [vp]
    L.apss_group_set_term_cuts.restype = i32
    L.apss_group_set_term_cuts.argtypes = [vp, vp]
    L.apss_group_insert.restype = i32
    L.apss_group_insert.argtypes = [vp, i64, vp, vp, vp, vp]
    for name in ("apss_group_query", "apss_group_insert_and_query"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp, i64, vp, vp, vp, vp, pi64]
    L.apss_group_insert_and_query_dev.restype = i32
    L.apss_group_insert_and_query_dev.argtypes = [vp, i64, i64, vp, vp, vp, vp, pi64]
    L.apss_group_clear.restype = i32
    L.apss_group_clear.argtypes = [vp]
    L.apss_group_result_count.restype = i32
    L.apss_group_result_count.argtypes = [vp, pi64]
    L.apss_group_fetch_results.restype = i32
    L.apss_group_fetch_results.argtypes = [vp, i64, i64, vp, vp, vp]
    L.apss_group_stats_get.restype = i32
    L.apss_group_stats_get.argtypes = [vp, C.POINTER(GroupStats)]
    L.apss_group_member_stats.restype = i32
    L.apss_group_member_stats.argtypes = [vp, i32, C.POINTER(Stats)]
    _lib = L
    return L
