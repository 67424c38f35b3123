"""BASELINE config 1 (plumbing): TF-IDF vectors of the reference's mail corpus, restating the reference's ETL
(etl/src/main/scala/cpslab/etl/PreprocessWithTFIDF.scala:21-52) with Spark 1.2.0 mllib's HashingTF / IDF formulas
(the Spark sources are not in the reference tree; formula-level restatement):
  * a file becomes ONE string: every line + " ", and the literal "null " that the read loop appends (:36-39);
  * tokens = that string split on " " (Java String.split: trailing empty strings dropped);
  * HashingTF(2^20): index = nonNegativeMod(String.hashCode, 2^20), value = term count;
  * IDF: ln((m + 1) / (df + 1)) over ALL m documents; tf-idf = tf * idf;
  * L2 normalisation (the reference's client does it, benchmark/LoadGenerator.scala:34-37; the ETL does not).
Every 12th document (sorted by path) goes into the fixture with the oracle's expected output at theta = 0.7 (conf/app.conf).
The corpus itself stays in /root/reference; only derived numbers are committed.
Run from the repo root: python tests/golden/make_maildir_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "all-pairs-similarity_amd"), os.path.join(ROOT, "tests")]
from oracle import oracle  # noqa: E402

CORPUS = "/root/reference/data/maildir_small"
DIM = 1 << 20
THETA = 0.7
STRIDE = 12


paths = sorted(os.path.join(d, f) for d, _, fs in os.walk(CORPUS) for f in fs)
m = len(paths)
# IDF over ALL m documents, then every STRIDE-th document goes into the fixture
rp_all, idx_all, val_all = oracle.tfidf_corpus(paths, DIM, normalize=True)
sel = list(range(0, m, STRIDE))
rowptr, idx, val = [0], [], []
for r in sel:
    sl = slice(rp_all[r], rp_all[r + 1])
    idx += list(idx_all[sl])
    val += list(val_all[sl])
    rowptr.append(len(idx))
rowptr, idx = np.array(rowptr, np.int64), np.array(idx, np.int32)
val = np.array(val, np.float64).astype(np.float32).astype(np.float64)  # stored at fp32 precision (smaller file)
q, c, s = oracle.selfjoin_pairs(DIM, THETA, rowptr, idx, val)
out = os.path.join(os.path.dirname(__file__), "maildir_small_tfidf.npz")
np.savez_compressed(out, dim=DIM, theta=THETA, rowptr=rowptr, indices=idx, values=val,
                    out_q=q, out_c=c, out_sim=s, n_corpus=m, stride=STRIDE)
print("docs", m, "fixture rows", len(sel), "nnz", idx.size, "max nnz", int(np.diff(rowptr).max()), "pairs >= theta", q.size,
      "file KB", os.path.getsize(out) // 1024)
