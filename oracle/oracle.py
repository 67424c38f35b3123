"""ctypes front-end of the CPU oracle (oracle/apss_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/apss_oracle.h.  Importable from tests/, from
__graft_entry__.smoke() and from bench.py's cpu_baseline leg, never from the product package.
Parity status: UNPINNED by the reference (no tests/golden vectors exist there, no JVM here); pinned
by the hand-derived KATs of SURVEY.md section 3.3 and a scipy float64 cross-check (tests/test_oracle.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libapss_oracle.so")

MODE_INTENDED = 0
MODE_AS_WRITTEN = 1


def build(force=False):
    """Compile the C restatement with gcc (idempotent)."""
    src = os.path.join(_HERE, "apss_oracle.c")
    hdr = os.path.join(_HERE, "apss_oracle.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-shared", "-o", _SO, src, "-lm", "-lpthread"])
    return _SO


_lib = None

_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.oracle_worker_create.restype = C.c_void_p
        L.oracle_worker_create.argtypes = [C.c_int32, C.c_double, C.c_int32]
        L.oracle_worker_destroy.argtypes = [C.c_void_p]
        L.oracle_worker_size.restype = C.c_int64
        L.oracle_worker_size.argtypes = [C.c_void_p]
        L.oracle_worker_index_data.restype = C.c_int64
        L.oracle_worker_index_data.argtypes = [C.c_void_p, C.c_int64, _i64p, _i64p, _i32p, _f64p, _i64p, _i32p,
                                               C.c_int32, C.POINTER(_i64p), C.POINTER(_i64p), C.POINTER(_f64p)]
        L.oracle_calculate_similarity.restype = C.c_double
        L.oracle_calculate_similarity.argtypes = [C.c_int32, C.c_int32, _i32p, _f64p, C.c_int32, C.c_int32, _i32p, _f64p]
        L.oracle_l2_normalize.argtypes = [C.c_int64, _i64p, _f64p]
        L.oracle_value_prune.restype = C.c_int64
        L.oracle_value_prune.argtypes = [C.c_int64, _i64p, _i32p, _f64p, C.c_double, _i64p, _i32p, _f64p]
        L.oracle_admission.argtypes = [C.c_int64, _i64p, _f64p, C.c_double, _u8p]
        L.oracle_cluster_create.restype = C.c_void_p
        L.oracle_cluster_create.argtypes = [C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.oracle_cluster_destroy.argtypes = [C.c_void_p]
        L.oracle_cluster_flush.restype = C.c_int64
        L.oracle_cluster_flush.argtypes = [C.c_void_p, C.c_int64, _i64p, _i64p, _i32p, _f64p,
                                           C.POINTER(_i64p), C.POINTER(_i64p), C.POINTER(_f64p)]
        L.oracle_cluster_sim_calls.restype = C.c_int64
        L.oracle_cluster_sim_calls.argtypes = [C.c_void_p]
        L.oracle_parse_sparse_vector.restype = C.c_int64
        L.oracle_parse_sparse_vector.argtypes = [C.c_char_p, C.POINTER(C.c_int32), _i32p, _f64p, C.c_int64]
        L.oracle_print_sparse_vector.restype = C.c_int64
        L.oracle_print_sparse_vector.argtypes = [C.c_int32, C.c_int64, _i32p, _f64p, C.c_char_p, C.c_int64]
        L.oracle_selfjoin_sample.restype = C.c_int64
        L.oracle_selfjoin_sample.argtypes = [C.c_int32, C.c_int32, C.c_double, C.c_int64, _i64p, _i32p, _f64p,
                                             C.c_int64, C.c_int64, C.c_int32, _i64p, _i64p, _f64p]
        L.oracle_selfjoin_pairs.restype = C.c_int64
        L.oracle_selfjoin_pairs.argtypes = [C.c_int32, C.c_double, C.c_int64, _i64p, _i32p, _f64p, C.c_int64,
                                            C.c_int64, _i64p, _i64p, _f64p, C.c_int64]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _csr(rowptr, indices, values):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.float64)
    if indices.size == 0:  # keep a valid pointer
        indices = np.zeros(1, np.int32)
        values = np.zeros(1, np.float64)
    return rowptr, indices, values


def _triples(n, oq, oc, os_):
    if n <= 0:
        return (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float64))
    return (np.ctypeslib.as_array(oq, (n,)).copy(), np.ctypeslib.as_array(oc, (n,)).copy(),
            np.ctypeslib.as_array(os_, (n,)).copy())


class Worker:
    """One IndexingWorkerActor (IWA:21-148)."""

    def __init__(self, dim, theta, mode=MODE_INTENDED):
        self._h = lib().oracle_worker_create(dim, theta, mode)
        if not self._h:
            raise ValueError("bad oracle worker config")
        self.mode = mode

    def close(self):
        if self._h:
            lib().oracle_worker_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __len__(self):
        return lib().oracle_worker_size(self._h)

    def index_data(self, ids, rowptr, indices, values, local=None, query_only=False, build_only=False):
        """`case IndexData(vectors)`.  local = (lptr, ldims) or None (all dims, ascending).
        Returns (q_ids, c_ids, sims) sorted by (q, c)."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        rowptr, indices, values = _csr(rowptr, indices, values)
        n = ids.size
        if local is None:
            lptr_p, ld_p = None, None
            if self.mode == MODE_AS_WRITTEN and n and np.diff(rowptr).max() > 4:
                raise ValueError("as_written is only restated for <= 4 local dims per wrapper (Scala Set1..Set4 "
                                 "keep insertion order; larger sets iterate in hash-trie order)")
        else:
            lptr = np.ascontiguousarray(local[0], dtype=np.int64)
            ld = np.ascontiguousarray(local[1], dtype=np.int32)
            if ld.size == 0:
                ld = np.zeros(1, np.int32)
            lptr_p, ld_p = _p(lptr, _i64p), _p(ld, _i32p)
        oq, oc, os_ = _i64p(), _i64p(), _f64p()
        ids_arr = ids if n else np.zeros(1, np.int64)
        m = lib().oracle_worker_index_data(self._h, n, _p(ids_arr, _i64p), _p(rowptr, _i64p), _p(indices, _i32p),
                                           _p(values, _f64p), lptr_p, ld_p, 2 if build_only else int(query_only),
                                           C.byref(oq), C.byref(oc), C.byref(os_))
        if m == -3:
            raise KeyError("NoSuchElementException: unseen dim on a frozen index (IWA:104)")
        if m < 0:
            raise ValueError("oracle_worker_index_data failed: %d" % m)
        return _triples(m, oq, oc, os_)


class Cluster:
    """The reference's shard/entry/worker fan-out around the hot path (WWA:164-183, EPA:37-49, CU:32)."""

    def __init__(self, dim, theta, mode=MODE_INTENDED, max_shard_num=1, max_entry_num=1,
                 max_index_entry_actor_num=1):
        self._h = lib().oracle_cluster_create(dim, theta, mode, max_shard_num, max_entry_num,
                                              max_index_entry_actor_num)
        if not self._h:
            raise ValueError("bad oracle cluster config")

    def close(self):
        if self._h:
            lib().oracle_cluster_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def flush(self, ids, rowptr, indices, values):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        rowptr, indices, values = _csr(rowptr, indices, values)
        oq, oc, os_ = _i64p(), _i64p(), _f64p()
        m = lib().oracle_cluster_flush(self._h, ids.size, _p(ids, _i64p), _p(rowptr, _i64p), _p(indices, _i32p),
                                       _p(values, _f64p), C.byref(oq), C.byref(oc), C.byref(os_))
        if m < 0:
            raise ValueError("oracle_cluster_flush failed: %d" % m)
        return _triples(m, oq, oc, os_)

    def sim_calls(self):
        return lib().oracle_cluster_sim_calls(self._h)


def calculate_similarity(size1, idx1, val1, size2, idx2, val2):
    i1 = np.ascontiguousarray(idx1, np.int32)
    v1 = np.ascontiguousarray(val1, np.float64)
    i2 = np.ascontiguousarray(idx2, np.int32)
    v2 = np.ascontiguousarray(val2, np.float64)
    return lib().oracle_calculate_similarity(size1, i1.size, _p(i1, _i32p), _p(v1, _f64p), size2, i2.size,
                                             _p(i2, _i32p), _p(v2, _f64p))


def l2_normalize(rowptr, values):
    rowptr = np.ascontiguousarray(rowptr, np.int64)
    out = np.array(values, dtype=np.float64, copy=True)
    if out.size:
        lib().oracle_l2_normalize(rowptr.size - 1, _p(rowptr, _i64p), _p(out, _f64p))
    return out


def value_prune(rowptr, indices, values, threshold):
    rowptr, indices, values = _csr(rowptr, indices, values)
    n = rowptr.size - 1
    orp = np.zeros(n + 1, np.int64)
    oi = np.zeros(max(indices.size, 1), np.int32)
    ov = np.zeros(max(values.size, 1), np.float64)
    nnz = lib().oracle_value_prune(n, _p(rowptr, _i64p), _p(indices, _i32p), _p(values, _f64p), threshold,
                                   _p(orp, _i64p), _p(oi, _i32p), _p(ov, _f64p))
    return orp, oi[:nnz].copy(), ov[:nnz].copy()


def admission(rowptr, values, theta):
    rowptr = np.ascontiguousarray(rowptr, np.int64)
    values = np.ascontiguousarray(values, np.float64)
    if values.size == 0:
        values = np.zeros(1, np.float64)
    keep = np.zeros(max(rowptr.size - 1, 1), np.uint8)
    lib().oracle_admission(rowptr.size - 1, _p(rowptr, _i64p), _p(values, _f64p), theta, _p(keep, _u8p))
    return keep[:rowptr.size - 1].astype(bool)


def parse_sparse_vector(text):
    b = text.encode()
    size = C.c_int32(0)
    n = lib().oracle_parse_sparse_vector(b, C.byref(size), None, None, 0)
    if n < 0:
        raise ValueError("cannot parse %r" % text)
    idx = np.zeros(max(n, 1), np.int32)
    val = np.zeros(max(n, 1), np.float64)
    lib().oracle_parse_sparse_vector(b, C.byref(size), _p(idx, _i32p), _p(val, _f64p), n)
    return size.value, idx[:n], val[:n]


def print_sparse_vector(size, indices, values):
    idx = np.ascontiguousarray(indices, np.int32)
    val = np.ascontiguousarray(values, np.float64)
    n = idx.size
    if n == 0:
        idx = np.zeros(1, np.int32)
        val = np.zeros(1, np.float64)
    need = lib().oracle_print_sparse_vector(size, n, _p(idx, _i32p), _p(val, _f64p), None, 0)
    buf = C.create_string_buffer(need + 1)
    lib().oracle_print_sparse_vector(size, n, _p(idx, _i32p), _p(val, _f64p), buf, need + 1)
    return buf.value.decode()


def ccweb_line_parser(line):
    """CCWEBVideoLoadGenerator.lineParser (core/src/main/scala/cpslab/benchmark/CCWEBVideoLoadGenerator.scala:10-21),
    restated: strip every ( ) [ ], split on ",", field 0 = video id, field 1 = vector size, the LAST `size` fields = the
    dense values, zeros dropped.  Returns (id, size, indices int32[], values float64[]); raises ValueError where the
    reference throws (NumberFormatException / ArrayIndexOutOfBoundsException)."""
    for ch in "()[]":
        line = line.replace(ch, "")
    f = line.split(",")
    while f and f[-1] == "":  # java.lang.String.split drops trailing empty strings
        f.pop()
    if len(f) < 2:
        raise ValueError("no id and size")
    size = _java_int(f[1])
    if size < 0 or size > len(f):  # takeRight gives fewer values than allIndices has entries: allValues(i) fails
        raise ValueError("fewer than `size` values")
    dense = np.array([_java_double(x) for x in f[len(f) - size:]], np.float64) if size else np.zeros(0)
    nz = np.nonzero(dense != 0)[0]
    return f[0], size, nz.astype(np.int32), dense[nz]


# ---- TF-IDF ingest of a text corpus (BASELINE config 1), restated from the reference's ETL
# etl/src/main/scala/cpslab/etl/PreprocessWithTFIDF.scala:21-52 with Spark 1.2.0 mllib's HashingTF / IDF formulas (the
# Spark sources are not in the reference tree: formula-level restatement)
def java_string_hash(s):
    """java.lang.String.hashCode over the string's chars (here: ISO-8859-1, one char per byte), int32 wrap-around"""
    h = 0
    for ch in s:
        h = (31 * h + ord(ch)) & 0xFFFFFFFF
    return h - (1 << 32) if h & 0x80000000 else h


def non_negative_mod(x, mod):
    """org.apache.spark.util.Utils.nonNegativeMod: the JVM's truncated remainder, shifted into [0, mod)"""
    r = int(np.fmod(x, mod))
    return r + mod if r < 0 else r


def document_tokens(path):
    """PreprocessWithTFIDF.scala:33-41: a file becomes ONE string -- every line + " ", then the literal "null " that the
    read loop appends -- split on " " (java.lang.String.split: trailing empty strings dropped, inner ones kept)"""
    raw = open(path, "rb").read().decode("latin-1")
    lines = raw.replace("\r\n", "\n").replace("\r", "\n").split("\n")
    if lines and lines[-1] == "":
        lines.pop()  # BufferedReader.readLine: a final line terminator does not start another line
    s = "".join(ln + " " for ln in lines) + "null "
    toks = s.split(" ")
    while toks and toks[-1] == "":
        toks.pop()
    return toks


def hashing_tf(tokens, num_features=1 << 20, cache=None):
    """mllib.feature.HashingTF.transform: index = nonNegativeMod(term.hashCode, numFeatures), value = term count"""
    tf = {}
    for t in tokens:
        i = cache.get(t) if cache is not None else None
        if i is None:
            i = non_negative_mod(java_string_hash(t), num_features)
            if cache is not None:
                cache[t] = i
        tf[i] = tf.get(i, 0.0) + 1.0
    return tf


def idf_weights(tfs):
    """mllib.feature.IDF (minDocFreq = 0): idf_t = ln((m + 1) / (df_t + 1)) over all m documents"""
    df = {}
    for tf in tfs:
        for i in tf:
            df[i] = df.get(i, 0) + 1
    m = len(tfs)
    return {i: float(np.log((m + 1.0) / (c + 1.0))) for i, c in df.items()}


def tfidf_corpus(paths, num_features=1 << 20, normalize=True):
    """the ETL over a list of files: CSR (rowptr, indices, values) of tf * idf, L2-normalised when `normalize` (the
    reference's client does that, benchmark/LoadGenerator.scala:34-37; its ETL does not)"""
    cache = {}
    tfs = [hashing_tf(document_tokens(p), num_features, cache) for p in paths]
    idf = idf_weights(tfs)
    rowptr, idx, val = [0], [], []
    for tf in tfs:
        ks = sorted(tf)
        v = np.array([tf[k] * idf[k] for k in ks], np.float64)
        if normalize:
            nrm = np.sqrt((v * v).sum())
            v = v / nrm if nrm > 0 else v
        idx += ks
        val += list(v)
        rowptr.append(len(idx))
    return np.array(rowptr, np.int64), np.array(idx, np.int32), np.array(val, np.float64)


def _java_int(s):
    if not s or not (s.lstrip("+-").isdigit() and s.isascii()) or s in "+-":
        raise ValueError("not an Int: %r" % s)
    return int(s)


def _java_double(s):
    try:
        if not s or s != s.strip():  # (Double.parseDouble trims; fields of this format never carry blanks)
            raise ValueError
        return float(s)
    except ValueError:
        raise ValueError("not a Double: %r" % s)


def selfjoin_sample(variant, dim, theta, rowptr, indices, values, q_begin, q_end, n_threads):
    """CPU baseline: returns dict(pairs, cand_pairs, visits, seconds)."""
    rowptr, indices, values = _csr(rowptr, indices, values)
    cands, visits, secs = C.c_int64(0), C.c_int64(0), C.c_double(0)
    pairs = lib().oracle_selfjoin_sample(variant, dim, theta, rowptr.size - 1, _p(rowptr, _i64p), _p(indices, _i32p),
                                         _p(values, _f64p), q_begin, q_end, n_threads, C.byref(cands),
                                         C.byref(visits), C.byref(secs))
    if pairs < 0:
        raise ValueError("oracle_selfjoin_sample failed")
    return dict(pairs=pairs, cand_pairs=cands.value, visits=visits.value, seconds=secs.value)


def selfjoin_pairs(dim, theta, rowptr, indices, values, q_begin=0, q_end=None):
    """Exact threshold join in double by dense accumulation; ids are row numbers."""
    rowptr, indices, values = _csr(rowptr, indices, values)
    n = rowptr.size - 1
    if q_end is None:
        q_end = n
    cap = 1 << 16
    while True:
        oq = np.zeros(cap, np.int64)
        oc = np.zeros(cap, np.int64)
        os_ = np.zeros(cap, np.float64)
        need = lib().oracle_selfjoin_pairs(dim, theta, n, _p(rowptr, _i64p), _p(indices, _i32p), _p(values, _f64p),
                                           q_begin, q_end, _p(oq, _i64p), _p(oc, _i64p), _p(os_, _f64p), cap)
        if need < 0:
            raise ValueError("oracle_selfjoin_pairs failed")
        if need <= cap:
            order = np.lexsort((oc[:need], oq[:need]))
            return oq[:need][order], oc[:need][order], os_[:need][order]
        cap = int(need)
