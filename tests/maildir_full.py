"""Weights of the full-corpus C1 fixture (tests/golden/maildir_full_counts.npz): TF-IDF from term counts, the formulas of
the reference's ETL (etl/src/main/scala/cpslab/etl/PreprocessWithTFIDF.scala:45-52 with Spark 1.2.0 mllib IDF:
idf = ln((m + 1) / (df + 1)) over all m documents) + the client's L2 normalisation (benchmark/LoadGenerator.scala:34-37).
Shared by the fixture generator (which feeds the oracle with it) and the GPU test: both see the same doubles."""
import numpy as np


def load(path):
    z = np.load(path)
    rowptr = z["rowptr"].astype(np.int64)
    d = z["index_deltas"].astype(np.int64)  # per row: first index, then the gaps
    idx = np.empty(d.size, np.int64)
    for r in range(rowptr.size - 1):
        b, e = rowptr[r], rowptr[r + 1]
        idx[b:e] = np.cumsum(d[b:e])
    return z, rowptr, idx.astype(np.int32), z["counts"].astype(np.int64)


def weights(rowptr, idx, counts):
    m = rowptr.size - 1
    dim = int(idx.max()) + 1
    df = np.bincount(idx, minlength=dim)  # every (row, term) entry is one document containing the term
    idf = np.log((m + 1.0) / (df + 1.0))
    v = counts.astype(np.float64) * idf[idx]
    row = np.repeat(np.arange(m), np.diff(rowptr))
    nrm = np.sqrt(np.bincount(row, weights=v * v, minlength=m))
    return v / np.where(nrm > 0, nrm, 1.0)[row]
