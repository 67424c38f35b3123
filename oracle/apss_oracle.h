/*
 * apss_oracle.h -- CPU restatement of the reference's all-pairs-similarity hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the timed CPU baseline.  The product path (all-pairs-similarity_amd/) never
 * links, imports or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference (Scala 2.10 / Akka, no JVM in this image)
 * ships no tests, golden vectors or fixtures for this path and cannot be compiled or run here.
 * This restatement is pinned instead by (i) hand-derived known-answer tests of the reference's
 * semantics, (ii) an independent scipy.sparse float64 X*X^T cross-check,
 * (iii) property tests (partition invariance, streaming subset, in-batch symmetry).
 *
 * Reference files restated (paths relative to /root/reference/core/src/main/scala/cpslab):
 *   deploy/server/IndexingWorkerActor.scala:61-71   buildInvertedIndex
 *   deploy/server/IndexingWorkerActor.scala:74-111  querySimilarItems / querySimilarVectors
 *   deploy/CommonUtils.scala:98-117                 calculateSimilarity (double sparse dot)
 *   benchmark/LoadGenerator.scala:34-37             L2 normalisation
 *   deploy/server/WriteWorkerActor.scala:185-202    value prune (value > indexThreshold)
 *   deploy/server/EntryProxyActor.scala:81-93       admission filter (max-weight == 1.0)
 *   deploy/server/WriteWorkerActor.scala:164-183, EntryProxyActor.scala:37-49,
 *   deploy/CommonUtils.scala:28-40                  two-level term-modulo partitioning
 *   vector/SparseVector.scala:132-141,204-205       "(size,[i,..],[v,..])" text format
 */
#ifndef APSS_ORACLE_H
#define APSS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* semantics switch for IndexingWorkerActor.querySimilarItems */
enum {
  ORACLE_MODE_INTENDED = 0,  /* the `outputSimSet.contains(queryVectorId)` gate removed (exact threshold join) */
  ORACLE_MODE_AS_WRITTEN = 1 /* IWA:89 gate kept: a query's first-iterated local dim is never scored (quirk Q1) */
};

typedef struct oracle_worker oracle_worker;

/* one IndexingWorkerActor: vectorsStore + invertedIndex + similarityThreshold (IWA:21-25) */
oracle_worker *oracle_worker_create(int32_t dim, double theta, int32_t mode);
void oracle_worker_destroy(oracle_worker *w);
int64_t oracle_worker_size(const oracle_worker *w);

/*
 * The `case IndexData(vectors)` handler (IWA:123-137): buildInvertedIndex(batch) unless
 * `query_only` == 1 (stopUpdateIndex, IWA:125-127), then querySimilarItems(batch); `query_only` == 2 builds only (no
 * answers are computed: the warm-up phase of a latency run).
 *
 * Batch layout: n wrappers; wrapper i has the FULL vector rowptr[i]..rowptr[i+1] (indices strictly
 * increasing, < dim) and the LOCAL dims lptr[i]..lptr[i+1] (the wrapper's `indices: Set[Int]`, in
 * iteration order).  lptr == NULL means "all of the vector's dims, ascending" (single worker).
 *
 * Output: the SimilarityOutput map flattened to triples (query id, candidate id, similarity), one per
 * (qid, cid) key (later writes overwrite earlier ones like HashMap `+=`), sorted by (qid, cid).
 * Returns the number of triples, or <0 on error (-1 bad argument, -2 size mismatch a la CU:99,
 * -3 as_written with an unseen dim, the NoSuchElementException of IWA:104).
 * The triple arrays stay owned by the worker and valid until its next call.
 */
int64_t oracle_worker_index_data(oracle_worker *w, int64_t n, const int64_t *ids,
                                 const int64_t *rowptr, const int32_t *indices, const double *values,
                                 const int64_t *lptr, const int32_t *ldims, int32_t query_only,
                                 const int64_t **out_q, const int64_t **out_c, const double **out_sim);

/* CU:98-117. Returns NaN when size1 != size2 (the `require`). */
double oracle_calculate_similarity(int32_t size1, int32_t n1, const int32_t *idx1, const double *val1,
                                   int32_t size2, int32_t n2, const int32_t *idx2, const double *val2);

/* LG:34-37: values / sqrt(foldLeft(sum + v*v)). In place. */
void oracle_l2_normalize(int64_t n, const int64_t *rowptr, double *values);

/* WWA:188-194: keep (i, v) with v > threshold, order preserved (indices stay ascending); no
 * re-normalisation.  Writes compacted rows to out_*; returns the new nnz. out_rowptr has n+1 entries. */
int64_t oracle_value_prune(int64_t n, const int64_t *rowptr, const int32_t *indices, const double *values,
                           double threshold, int64_t *out_rowptr, int32_t *out_indices, double *out_values);

/* EPA:81-93 with maxWeight == 1.0 for every dim (EPA:51-57): keep[i] = (sum_j v_ij >= theta). */
void oracle_admission(int64_t n, const int64_t *rowptr, const double *values, double theta, uint8_t *keep);

/*
 * The reference's cluster fan-out around the hot path, to demonstrate partition invariance:
 * WriteWorkerActor.handleIOTrigger (one DataPacket per shard = dim % max_shard_num, vectors without a dim in
 * the shard skipped, WWA:172-175), ShardRegion entry = shard % max_entry_num (CU:32,36),
 * EntryProxyActor.spawnToIndexActor (worker i = dim % max_index_entry_actor_num, a wrapper is sent to EVERY
 * i, even with empty dims, EPA:41-46).  One oracle_worker per (entry, i); outputs are unioned and
 * de-duplicated.  `batch_ptr` splits the n vectors into n_batches consecutive flush batches
 * (batch b = rows batch_ptr[b]..batch_ptr[b+1]); every call continues the same cluster state.
 */
typedef struct oracle_cluster oracle_cluster;
oracle_cluster *oracle_cluster_create(int32_t dim, double theta, int32_t mode, int32_t max_shard_num,
                                      int32_t max_entry_num, int32_t max_index_entry_actor_num);
void oracle_cluster_destroy(oracle_cluster *c);
int64_t oracle_cluster_flush(oracle_cluster *c, int64_t n, const int64_t *ids, const int64_t *rowptr,
                             const int32_t *indices, const double *values, const int64_t **out_q,
                             const int64_t **out_c, const double **out_sim);
/* total calculateSimilarity calls made by all workers (shows the reference's duplicate scoring, Q6) */
int64_t oracle_cluster_sim_calls(const oracle_cluster *c);

/*
 * Text format of cpslab.vector.SparseVector (SV:204-205 print, SV:132-141 parse).
 * parse: returns nnz (>=0) or <0 on error; *size receives the vector size; up to cap entries written.
 * print: writes into buf (cap bytes), returns the number of bytes needed (excluding NUL).
 */
int64_t oracle_parse_sparse_vector(const char *text, int32_t *size, int32_t *indices, double *values, int64_t cap);
int64_t oracle_print_sparse_vector(int32_t size, int64_t nnz, const int32_t *indices, const double *values,
                                   char *buf, int64_t cap);

/*
 * CPU baselines for bench.py (intended semantics, single IndexingWorkerActor, self-join of a query
 * sample against all n indexed vectors).
 *   variant 0 "refcpu": CSC posting lists + per-candidate hash-map dot product in double exactly as
 *             CU:98-117 does (two maps built per call), candidates de-duplicated per query.
 *   variant 1 "optcpu": CSC posting lists + dense double accumulator per query (fairness bracket).
 *   variant 2 "refcpu-T": the reference's own parallelism -- n_threads workers, worker t owning the dims with
 *              dim % n_threads == t (EntryProxyActor.scala:41-46), each seeing every query and recomputing the full
 *              hash-map dot of every pair it reaches through one of its dims; *cand_pairs counts scorings (a pair that
 *              shares dims of k workers is scored k times, IWA:105 "TODO: need to deduplicate").
 * Queries q_begin..q_end are split over n_threads pthreads.  Returns pairs >= theta found; *cand_pairs gets
 * the number of scored candidate pairs, *visits the posting visits, *seconds the wall time of the query
 * phase only (the CSC build is excluded).
 */
int64_t oracle_selfjoin_sample(int32_t variant, int32_t dim, double theta, int64_t n, const int64_t *rowptr,
                               const int32_t *indices, const double *values, int64_t q_begin, int64_t q_end,
                               int32_t n_threads, int64_t *cand_pairs, int64_t *visits, double *seconds);

/* Exact threshold join by dense accumulation in double (optcpu kernel), returning the triples with
 * row numbers as ids.  Used by tests at sizes where the per-candidate restatement is too slow.
 * Two-call pattern: returns needed count; writes up to cap. */
int64_t oracle_selfjoin_pairs(int32_t dim, double theta, int64_t n, const int64_t *rowptr,
                              const int32_t *indices, const double *values, int64_t q_begin, int64_t q_end,
                              int64_t *out_q, int64_t *out_c, double *out_sim, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
