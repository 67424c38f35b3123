"""Python face of one device-resident index (one IndexingWorkerActor's state on one MI355X).

Mirrors the seam the C ABI replaces -- `case IndexData(vectors)` of
core/src/main/scala/cpslab/deploy/server/IndexingWorkerActor.scala:123-137 -- and nothing else.  All compute
is in libapss_hip.so; errors surface as ApssError carrying the library's message."""
import ctypes as C

import numpy as np

from . import _lib


class ApssError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("apss error %d: %s" % (code, msg))
        self.code = code


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a.size else C.c_void_p(0)


class ApssIndex:
    def __init__(self, dim, theta, device=0, tile_rows=0, term_range=None, flags=0, index_threshold=0.0,
                 capacity_rows=0, capacity_nnz=0, head_terms=0):
        L = _lib.lib()
        cfg = _lib.Config()
        cfg.struct_size = C.sizeof(_lib.Config)
        cfg.dim = int(dim)
        cfg.theta = float(theta)
        cfg.index_threshold = float(index_threshold)
        cfg.flags = int(flags)
        cfg.device_id = int(device)
        cfg.term_lo, cfg.term_hi = (0, 0) if term_range is None else (int(term_range[0]), int(term_range[1]))
        cfg.tile_rows = int(tile_rows)
        cfg.head_terms = int(head_terms)  # dense-head block: 0 auto, -1 never, 64 | 128 | 256 forced
        cfg.capacity_rows = int(capacity_rows)
        cfg.capacity_nnz = int(capacity_nnz)
        h = C.c_void_p()
        rc = L.apss_create(C.byref(cfg), C.byref(h))
        if rc != _lib.OK:
            raise ApssError(rc, (L.apss_last_error(None) or b"").decode())
        self._h = h
        self._L = L
        self.dim, self.theta = int(dim), float(theta)

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._L.apss_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc != _lib.OK:
            raise ApssError(rc, (self._L.apss_last_error(self._h) or b"").decode())

    # -- host-pointer path (numpy in / numpy out)
    def _csr(self, ids, rowptr, indices, values):
        ids, rowptr = _np(ids, np.int64), _np(rowptr, np.int64)
        indices, values = _np(indices, np.int32), _np(values, np.float64)
        if rowptr.size != ids.size + 1:
            raise ValueError("rowptr must have len(ids) + 1 entries")
        return ids, rowptr, indices, values

    def insert(self, ids, rowptr, indices, values):
        """buildInvertedIndex(batch), IWA:61-71"""
        ids, rowptr, indices, values = self._csr(ids, rowptr, indices, values)
        self._chk(self._L.apss_insert(self._h, ids.size, _ptr(rowptr), _ptr(indices), _ptr(values), _ptr(ids)))

    def query(self, ids, rowptr, indices, values):
        """querySimilarItems(batch) on the frozen index (IWA:74-111 with stopUpdateIndex)"""
        ids, rowptr, indices, values = self._csr(ids, rowptr, indices, values)
        n = C.c_int64(0)
        self._chk(self._L.apss_query(self._h, ids.size, _ptr(rowptr), _ptr(indices), _ptr(values), _ptr(ids), C.byref(n)))
        return self.fetch()

    def insert_and_query(self, ids, rowptr, indices, values):
        """the IndexData handler, IWA:123-137"""
        ids, rowptr, indices, values = self._csr(ids, rowptr, indices, values)
        n = C.c_int64(0)
        self._chk(self._L.apss_insert_and_query(self._h, ids.size, _ptr(rowptr), _ptr(indices), _ptr(values), _ptr(ids),
                                                C.byref(n)))
        return self.fetch()

    def self_join(self, fetch=True):
        n = C.c_int64(0)
        self._chk(self._L.apss_self_join(self._h, C.byref(n)))
        return self.fetch() if fetch else n.value

    def result_count(self):
        n = C.c_int64(0)
        self._chk(self._L.apss_result_count(self._h, C.byref(n)))
        return n.value

    def fetch(self):
        """(query ext ids, candidate ext ids, scores) of the last query-type call"""
        n = self.result_count()
        q, c, s = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.float32)
        if n:
            self._chk(self._L.apss_fetch_results(self._h, 0, n, _ptr(q), _ptr(c), _ptr(s)))
        return q, c, s

    def size(self):
        r, z = C.c_int64(0), C.c_int64(0)
        self._chk(self._L.apss_size(self._h, C.byref(r), C.byref(z)))
        return r.value, z.value

    def stats(self):
        st = _lib.Stats()
        st.struct_size = C.sizeof(_lib.Stats)
        self._chk(self._L.apss_stats_get(self._h, C.byref(st)))
        d = {k: getattr(st, k) for k, _ in _lib.Stats._fields_}
        d["probe_kernel"] = d["probe_kernel"].decode()
        return d

    def set_head_terms(self, terms, part=0, n_parts=1, fold_columns=0):
        """dense-head block named by the caller (every term shard of a join gets the same terms; shard `part` of `n_parts`
        multiplies its share of the candidate tiles); fold_columns: how many of a wide head's 256 columns are folded ones
        (64 | 128 | 192; 0: the default, 128); empty handle only"""
        t = _np(terms, np.int32)
        self._chk(self._L.apss_set_head_fold(self._h, int(fold_columns)))
        self._chk(self._L.apss_set_head_terms(self._h, t.size, _ptr(t), int(part), int(n_parts)))

    def head_terms(self):
        n = C.c_int32(0)
        out = np.zeros(32768, np.int32)
        self._chk(self._L.apss_get_head_terms(self._h, out.size, _ptr(out), C.byref(n)))
        return out[:n.value].copy()

    def clear(self):
        self._chk(self._L.apss_clear(self._h))

    # -- device-pointer path (torch tensors on this handle's GPU; torch is plumbing only)
    def set_stream(self, cuda_stream_handle):
        """run on this HIP stream (0 = the default stream, where torch works unless told otherwise)"""
        self._chk(self._L.apss_set_stream(self._h, C.c_void_p(cuda_stream_handle), 0))

    def use_own_stream(self):
        self._chk(self._L.apss_set_stream(self._h, C.c_void_p(0), 1))

    @staticmethod
    def _dev(rowptr, indices, values, ids):
        import torch
        assert rowptr.dtype == torch.int64 and indices.dtype == torch.int32
        assert values.dtype == torch.float32 and ids.dtype == torch.int64
        for t in (rowptr, indices, values, ids):
            assert t.is_cuda and t.is_contiguous()
        return (ids.numel(), indices.numel(), C.c_void_p(rowptr.data_ptr()), C.c_void_p(indices.data_ptr()),
                C.c_void_p(values.data_ptr()), C.c_void_p(ids.data_ptr()))

    def insert_dev(self, ids, rowptr, indices, values):
        n, nnz, rp, ix, vl, ex = self._dev(rowptr, indices, values, ids)
        self._chk(self._L.apss_insert_dev(self._h, n, nnz, rp, ix, vl, ex))

    def query_dev(self, ids, rowptr, indices, values):
        n, nnz, rp, ix, vl, ex = self._dev(rowptr, indices, values, ids)
        out = C.c_int64(0)
        self._chk(self._L.apss_query_dev(self._h, n, nnz, rp, ix, vl, ex, C.byref(out)))
        return out.value

    def insert_and_query_dev(self, ids, rowptr, indices, values):
        n, nnz, rp, ix, vl, ex = self._dev(rowptr, indices, values, ids)
        out = C.c_int64(0)
        self._chk(self._L.apss_insert_and_query_dev(self._h, n, nnz, rp, ix, vl, ex, C.byref(out)))
        return out.value

    def results_dev(self):
        """raw device pointers (q_row int32*, c_slot int32*, score float*, n) of the last results"""
        a, b, c, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64(0)
        self._chk(self._L.apss_results_dev(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return a.value, b.value, c.value, n.value

    def results_to(self, q_row=None, c_slot=None, score=None, offset=0):
        """copy the last results into torch device tensors (int32, int32, float32) without leaving the GPU"""
        n = max(t.numel() for t in (q_row, c_slot, score) if t is not None)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else C.c_void_p(0)  # noqa: E731
        self._chk(self._L.apss_results_copy_dev(self._h, offset, n, ptr(q_row), ptr(c_slot), ptr(score)))

    def partial_scores_dev(self, q_row, c_slot, out):
        import torch
        assert q_row.dtype == torch.int32 and c_slot.dtype == torch.int32 and out.dtype == torch.float32
        self._chk(self._L.apss_partial_scores_dev(self._h, q_row.numel(), C.c_void_p(q_row.data_ptr()),
                                                  C.c_void_p(c_slot.data_ptr()), C.c_void_p(out.data_ptr())))


class ApssGroup:
    """Python face of apss_group (include/apss.h): the term-sharded index of one node -- `devices[i]` holds member i's
    term range -- behind one object; the members' exchange (all-gather of candidate lists, all-reduce of partial scores)
    runs below the C ABI (RCCL when every member has its own GPU, device-to-device copies when members share one).
    Mirrors WriteWorkerActor.scala:164-183 + EntryProxyActor.scala:37-49 + IndexingWorkerActor.scala:122-137."""

    def __init__(self, dim, theta, devices, flags=0, index_threshold=0.0, tile_rows=0, head_terms=0, group_flags=0,
                 term_cuts=None, capacity_rows=0, capacity_nnz=0):
        L = _lib.lib()
        cfg = _lib.Config()
        cfg.struct_size = C.sizeof(_lib.Config)
        cfg.dim, cfg.theta, cfg.index_threshold = int(dim), float(theta), float(index_threshold)
        cfg.flags, cfg.tile_rows, cfg.head_terms = int(flags), int(tile_rows), int(head_terms)
        cfg.capacity_rows, cfg.capacity_nnz = int(capacity_rows), int(capacity_nnz)
        devs = _np(devices, np.int32)
        g = C.c_void_p()
        rc = L.apss_group_create(C.byref(cfg), devs.size, _ptr(devs), int(group_flags), C.byref(g))
        if rc != _lib.OK:
            raise ApssError(rc, (L.apss_group_last_error(None) or b"").decode())
        self._g, self._L = g, L
        self.dim, self.theta, self.n_members = int(dim), float(theta), int(devs.size)
        if term_cuts is not None:
            cuts = _np(term_cuts, np.int32)
            if cuts.size != devs.size + 1:
                raise ValueError("term_cuts must have n_members + 1 entries")
            self._chk(L.apss_group_set_term_cuts(self._g, _ptr(cuts)))

    def close(self):
        if getattr(self, "_g", None):
            self._L.apss_group_destroy(self._g)
            self._g = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc != _lib.OK:
            raise ApssError(rc, (self._L.apss_group_last_error(self._g) or b"").decode())

    _csr = ApssIndex._csr

    def insert(self, ids, rowptr, indices, values):
        ids, rowptr, indices, values = self._csr(ids, rowptr, indices, values)
        self._chk(self._L.apss_group_insert(self._g, ids.size, _ptr(rowptr), _ptr(indices), _ptr(values), _ptr(ids)))

    def query(self, ids, rowptr, indices, values):
        ids, rowptr, indices, values = self._csr(ids, rowptr, indices, values)
        n = C.c_int64(0)
        self._chk(self._L.apss_group_query(self._g, ids.size, _ptr(rowptr), _ptr(indices), _ptr(values), _ptr(ids), C.byref(n)))
        return self.fetch()

    def insert_and_query(self, ids, rowptr, indices, values, fetch=True):
        """the IndexData handler on every member + the exchange, IWA:123-137"""
        ids, rowptr, indices, values = self._csr(ids, rowptr, indices, values)
        n = C.c_int64(0)
        self._chk(self._L.apss_group_insert_and_query(self._g, ids.size, _ptr(rowptr), _ptr(indices), _ptr(values), _ptr(ids),
                                                      C.byref(n)))
        return self.fetch() if fetch else n.value

    def insert_and_query_dev(self, per_member):
        """per_member: for every member (ids int64, rowptr int64, indices int32, values fp32) torch tensors on ITS device"""
        n = nnz = None
        tabs = [(C.c_void_p * self.n_members)() for _ in range(4)]
        for i, (ids, rowptr, indices, values) in enumerate(per_member):
            m, z, rp, ix, vl, ex = ApssIndex._dev(rowptr, indices, values, ids)
            assert n in (None, m) and nnz in (None, z), "every member is handed the same batch"
            n, nnz = m, z
            tabs[0][i], tabs[1][i], tabs[2][i], tabs[3][i] = rp, ix, vl, ex
        out = C.c_int64(0)
        self._chk(self._L.apss_group_insert_and_query_dev(self._g, n, nnz, tabs[0], tabs[1], tabs[2], tabs[3], C.byref(out)))
        return out.value

    def result_count(self):
        n = C.c_int64(0)
        self._chk(self._L.apss_group_result_count(self._g, C.byref(n)))
        return n.value

    def fetch(self):
        n = self.result_count()
        q, c, s = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.float32)
        if n:
            self._chk(self._L.apss_group_fetch_results(self._g, 0, n, _ptr(q), _ptr(c), _ptr(s)))
        return q, c, s

    def clear(self):
        self._chk(self._L.apss_group_clear(self._g))

    def stats(self):
        st = _lib.GroupStats()
        st.struct_size = C.sizeof(_lib.GroupStats)
        self._chk(self._L.apss_group_stats_get(self._g, C.byref(st)))
        d = {k: getattr(st, k) for k, _ in _lib.GroupStats._fields_}
        d["term_cuts"] = list(st.term_cuts[: self.n_members + 1])
        return d

    def member_stats(self, member):
        st = _lib.Stats()
        st.struct_size = C.sizeof(_lib.Stats)
        self._chk(self._L.apss_group_member_stats(self._g, int(member), C.byref(st)))
        d = {k: getattr(st, k) for k, _ in _lib.Stats._fields_}
        d["probe_kernel"] = d["probe_kernel"].decode()
        return d
