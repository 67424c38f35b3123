"""BASELINE config 1 at FULL size: all 8,586 documents of the reference's mail corpus (data/maildir_small) as TF-IDF
vectors, with the oracle's expected output at theta = 0.7 (conf/app.conf).

The fixture holds term COUNTS, not weights: because every document of the corpus is in it, the IDF table
(ln((m + 1) / (df + 1)), Spark 1.2.0 mllib IDF) is a function of the fixture's own index arrays, and the test derives the
weights with tests/maildir_full.py::weights -- the same function this script feeds the oracle with, so the device and the
oracle see bit-identical inputs.  The derivation is checked here against the oracle's own ETL restatement
(oracle.tfidf_corpus = etl/src/main/scala/cpslab/etl/PreprocessWithTFIDF.scala:21-52 + LoadGenerator.scala:34-37).
The corpus itself stays in /root/reference; only derived numbers are committed.
Run from the repo root: python tests/golden/make_maildir_full.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "all-pairs-similarity_amd"), os.path.join(ROOT, "tests")]
from oracle import oracle  # noqa: E402
from maildir_full import weights  # noqa: E402

CORPUS = "/root/reference/data/maildir_small"
DIM = 1 << 20
THETA = 0.7

paths = sorted(os.path.join(d, f) for d, _, fs in os.walk(CORPUS) for f in fs)
cache = {}
tfs = [oracle.hashing_tf(oracle.document_tokens(p), DIM, cache) for p in paths]
rowptr, idx, cnt = [0], [], []
for tf in tfs:
    ks = sorted(tf)
    idx += ks
    cnt += [int(tf[k]) for k in ks]
    rowptr.append(len(idx))
rowptr, idx, cnt = np.array(rowptr, np.int64), np.array(idx, np.int32), np.array(cnt, np.int64)
assert cnt.max() < 2 ** 31
val = weights(rowptr, idx, cnt)
rp2, idx2, val2 = oracle.tfidf_corpus(paths, DIM, normalize=True)
assert np.array_equal(rp2, rowptr) and np.array_equal(idx2, idx) and np.abs(val2 - val).max() < 1e-12
t0 = time.time()
q, c, s = oracle.selfjoin_pairs(DIM, THETA, rowptr, idx, val)
print("oracle self-join: %.1f s" % (time.time() - t0))
out = os.path.join(os.path.dirname(__file__), "maildir_full_counts.npz")
# indices delta-coded per row (first entry of a row: the index itself): small numbers compress well
d = np.diff(idx, prepend=0).astype(np.int32)
d[rowptr[:-1][np.diff(rowptr) > 0]] = idx[rowptr[:-1][np.diff(rowptr) > 0]]
np.savez_compressed(out, dim=DIM, theta=THETA, rowptr=rowptr.astype(np.int32), index_deltas=d, counts=cnt.astype(np.uint32),
                    out_q=q.astype(np.int32), out_c=c.astype(np.int32), out_sim=s)
print("docs", len(paths), "nnz", idx.size, "max nnz", int(np.diff(rowptr).max()), "pairs >= theta", q.size,
      "file KB", os.path.getsize(out) // 1024)
