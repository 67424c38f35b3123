/*
 * apss.h -- C ABI of the MI355X-native all-pairs-similarity hot path (libapss_hip.so).
 *
 * Drop-in boundary.  The reference (mcgill-cpslab/all-pairs-similarity, Scala/Akka) has no FFI of its own;
 * the seam this library replaces is the `case IndexData(vectors)` handler of one IndexingWorkerActor
 *     core/src/main/scala/cpslab/deploy/server/IndexingWorkerActor.scala:123-137
 * i.e. buildInvertedIndex (IWA:61-71) + querySimilarItems (IWA:74-111) + CommonUtils.calculateSimilarity
 * (core/src/main/scala/cpslab/deploy/CommonUtils.scala:98-117), with SparkSparseVector (size, indices:
 * Array[Int], values: Array[Double]) in and the SimilarityOutput map (message/Message.scala:20-21) out.
 * A JNI shim (all-pairs-similarity_amd/jvm/) forwards 1:1 to these entry points; see INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; every function returns an int32 status (0 = ok, < 0 = error,
 * text via apss_last_error); nothing throws or calls back across the boundary; inputs are caller-owned and
 * consumed before return; results stay in the handle until the next query-type call, insert or clear (after an
 * insert or clear the result calls answer APSS_E_STATE) and are copied out with apss_fetch_results (two-call
 * pattern: ask for the count, then fetch).
 * Threading mirrors the actor model: at most one thread inside a given handle at a time, different handles
 * may be used concurrently; the library sets the device on every entry and owns one HIP stream per handle.
 *
 * Semantics: "intended" semantics of the reference -- exact threshold join
 * sim(q, c) = sum_i q_i * c_i >= theta (inclusive, IWA:93) over every candidate c sharing >= 1 indexed term
 * with q, self-exclusion by EXTERNAL id (IWA:91), each (q, c) reported once (the reference's duplicate
 * reports across term workers, IWA:105 "TODO: need to deduplicate", are collapsed).  The reference's
 * first-dim skip quirk (IWA:89 vs 106) is NOT reproduced on the device; it is restated only in oracle/.
 * Scores are accumulated in fp32 on the device (reference: double): |score - double| <= 1e-5 and set
 * membership may differ only for |score - theta| <= 1e-5.
 */
#ifndef APSS_H
#define APSS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APSS_OK 0
#define APSS_E_INVALID (-1)     /* bad argument / malformed vector: the IllegalArgumentException of `require`
                                   (CommonUtils.scala:99, vector/SparseVector.scala:75,202) */
#define APSS_E_NOMEM (-2)       /* host or device allocation failed */
#define APSS_E_DEVICE (-3)      /* HIP runtime error (no GPU, launch failure, ...) */
#define APSS_E_STATE (-4)       /* call not valid in this state (e.g. fetch before any query) */
#define APSS_E_UNSUPPORTED (-5) /* option not supported by this build */

/* apss_config.flags */
#define APSS_FLAG_VALUE_PRUNE 1u /* drop entries with value <= index_threshold before indexing, no
                                    re-normalisation (WriteWorkerActor.scala:188-194) */
#define APSS_FLAG_ADMISSION 2u   /* admit a vector only if sum_i v_i >= theta (EntryProxyActor.scala:81-93,
                                    max-weight == 1.0 per EntryProxyActor.scala:51-57) */
#define APSS_FLAG_NORMALIZE 4u   /* L2-normalise rows on ingest (benchmark/LoadGenerator.scala:34-37) */
#define APSS_FLAG_FORCE_SCAN 8u  /* always use the general accumulator-scan kernel (the theta <= 0 path) */
#define APSS_FLAG_FORCE_GENERAL 16u /* never use the per-wave speed path of the probe (test hook) */
#define APSS_FLAG_EXACT_ACCUM 32u   /* single-pass join with exact accumulators only: no coarse filter + rescoring pass */
#define APSS_FLAG_NO_SYMMETRY 64u    /* a whole-store join (apss_self_join, insert-and-query into an empty handle) probes every
                                       pair in BOTH directions like the reference (IWA:123-134) instead of running the
                                       filter over half of the tile pairs and mirroring its survivors; same results */

typedef struct apss_handle apss_handle;

typedef struct apss_config {
  int32_t struct_size;     /* sizeof(apss_config), for ABI evolution */
  int32_t dim;             /* cpslab.allpair.vectorDim: every vector's `size` (CU:99) */
  double theta;            /* cpslab.allpair.similarityThreshold (IWA:23) */
  double index_threshold;  /* cpslab.allpair.indexThreshold (WWA:35), used with APSS_FLAG_VALUE_PRUNE */
  uint32_t flags;          /* APSS_FLAG_* */
  int32_t device_id;       /* HIP device ordinal */
  int32_t term_lo;         /* term-range shard [term_lo, term_hi): only these dims are indexed and scored */
  int32_t term_hi;         /*   (0, 0) or (0, dim) = the whole term space (single GPU) */
  int32_t tile_rows;       /* candidate tile of the EXACT rendering of the index (8-B postings, u32 / fp32 accumulators in one
                              workgroup's LDS: k_probe_wave, k_probe): 0 = default (16384 rows: two workgroups share a CU's
                              160 KB of LDS), else a multiple of 64, <= 32768.  The coarse rendering the two-pass join filters
                              with has tiles of min(2 * tile_rows, 32768) rows (16-bit accumulators); with tile_rows == 0 the
                              library may pick 65536 (sparse regime, term shards with 8-bit accumulators) or 131072 rows
                              (sparser still) from the data: apss_stats.tiles says what it took */
  int32_t head_terms;      /* dense-head block: 0 = the library decides from the term distribution, -1 = never,
                              N <= 32768 = always the N most frequent terms.  Up to 256 terms: a column each.  More: ONE block of
                              256 columns, the 128 most frequent terms with a column each, the others FOLDED into the other 128
                              (a shared column holds the L2 norm of its terms: by Cauchy-Schwarz an upper bound of their partial
                              score that keeps the row's norm).  Terms in the block are scored by an MFMA contraction instead of their posting lists --
                              over INT8 rows rounded UP (a sound filter: integer products, int32 sums) or bf16 rows
                              (apss_stats.head_int8); survivors are re-scored exactly: results are the same set */
  int64_t capacity_rows;   /* hints for the initial HBM reservation (0 = grow on demand) */
  int64_t capacity_nnz;
} apss_config;

typedef struct apss_stats {
  int32_t struct_size;      /* IN: sizeof(apss_stats) as the caller compiled it -- apss_stats_get writes no more than that many
                               bytes (the struct grows at its end from release to release); OUT: the bytes written */
  uint32_t symmetric;       /* 1: the last call ran as a symmetric whole-store join (see APSS_FLAG_NO_SYMMETRY) */
  int64_t rows;             /* vectors in the store */
  int64_t nnz;              /* postings in the index */
  int64_t tiles;            /* candidate tiles */
  int64_t posting_visits;   /* (query, term, posting) FMAs of the last query-type call */
  int64_t candidate_pairs;  /* distinct (q, c != q) pairs scored by the last query-type call.  With a dense-head block:
                               max(pairs sharing a tail term, pairs sharing a head term), a LOWER bound of the distinct
                               count (a pair sharing both kinds is scored by both filters, counted once here) */
  int64_t result_pairs;     /* pairs >= theta of the last query-type call */
  double probe_ms;          /* device time of the last call's probe kernel(s), HIP events */
  double build_ms;          /* device time of the last insert's index build, HIP events */
  int64_t probe_launches;   /* probe kernel launches of the last call (re-runs after a result-buffer growth count) */
  int64_t hbm_bytes;        /* device bytes currently reserved by the handle */
  int64_t filter_survivors; /* two-pass join: pairs the coarse filter passed on to exact rescoring (0: single pass) */
  double rescore_ms;        /* two-pass join: device time of the exact rescoring kernel */
  int64_t head_terms;       /* terms held in the dense-head block at the last call (0: none) */
  int64_t head_pairs;       /* (q, c != q) pairs sharing a head term, scored by the dense contraction of the last call */
  int64_t head_survivors;   /* pairs the dense filter passed on to exact rescoring */
  double head_ms;           /* device time of the dense-head kernel, HIP events */
  double head_flops;        /* 2 * KH * (query slots x candidate rows actually multiplied) of the last call */
  int64_t thin_launches;    /* probe launches of the last call that took the thin-round filter kernel (k_probe_even: a term
                               shard's or a sparse batch's rounds of a few hundred postings) */
  uint32_t downgrades;      /* APSS_DOWNGRADE_*: permanent fallbacks (until apss_clear) this handle took because of a call it could
                               not serve on its fast layout; each costs one full index rebuild when it happens */
  uint32_t head_columns;    /* width of a row of the dense-head block at the last call: 64 | 128 | 256 (0: none) */
  char probe_kernel[96];    /* the probe kernel instantiation the last query-type call launched, as rocprofv3 prints its name up
                               to the template arguments' spelling, e.g. "k_probe_even<512, 6, 128, false, false, false>" (threads, window steps,
                               long-segment list, shard rule, signed, 8-bit accumulators) or "k_probe_even_merged<4, true>" (a term shard's rounds
                               with two query rows each: window steps, 8-bit accumulators; queries_per_round); "" before any probe */
  int64_t device_posting_visits; /* posting visits the kernels of the last call actually made: equal to posting_visits except
                                    on a symmetric whole-store join, where posting_visits / candidate_pairs keep counting what
                                    the reference's two-directional probe visits and the device visits about half of it */
  uint32_t symmetric_declined; /* APSS_SYM_*: why the last query-type call did NOT run as a symmetric join (0 when it did) */
  int32_t query_chunk;      /* query rows per workgroup of the last probe launch (the symmetric join needs a power of two that
                               divides filter_tile_rows) */
  int32_t filter_tile_rows; /* candidate rows per tile of the index rendering the last probe ran over */
  int32_t head_int8;        /* 1: the dense-head block's rows are the INT8 rendering (rounded up; v_mfma_i32_32x32x32_i8: head_flops are
                               integer operations then, against twice the bf16 peak), 0: bf16 */
  int32_t queries_per_round; /* thin-round filter of a term shard: query rows that SHARED a round of the last probe launch (1 | 2 | 4).
                                M neighbouring rows are staged as one row and a candidate's accumulator holds the sum of their M
                                filter sums -- an upper bound of each (non-negative weights), so the filter stays sound; a crossing
                                becomes M pairs, each tested against the shard rule on its exact partial score before it is
                                reported (filter_survivors: no more than without merging).  candidate_pairs then counts a
                                candidate touched by several rows of one round once: a lower bound */
  int32_t reserved0;
} apss_stats;

/* apss_stats.symmetric_declined */
#define APSS_SYM_RAN 0u          /* the call ran symmetrically */
#define APSS_SYM_FLAG 1u         /* APSS_FLAG_NO_SYMMETRY (or the debug token) */
#define APSS_SYM_NOT_WHOLE 2u    /* the query batch is not the whole indexed store (a batch onto an older store, an outside
                                    batch, rows waiting outside the tile index) */
#define APSS_SYM_PATH 3u         /* the call did not take the two-pass filter (theta <= 0, exact-accumulate, signed fallback) */
#define APSS_SYM_LONG_ROWS 4u    /* a query row of more than 512 terms is cut into parts (virtual rows) */
#define APSS_SYM_ONE_TILE 5u     /* the index has a single tile: nothing to halve */
#define APSS_SYM_CHUNK 6u        /* the query chunk does not divide a tile (query_chunk, filter_tile_rows) */

#define APSS_DOWNGRADE_ACC8 1u /* 8-bit accumulators over 65536 / 131072-row tiles given up (a long row, a large norm, an
                                  unselective byte filter): 16-bit accumulators over smaller tiles */
#define APSS_DOWNGRADE_HEAD 2u /* the dense-head block was given up (signed weights, norms out of range): its terms are back
                                  in the inverted index */

/* ---- lifetime (actor construction / stop, IWA:21-39) ---- */
int32_t apss_create(const apss_config *cfg, apss_handle **out);
void apss_destroy(apss_handle *h);
/* message of the last failing call on this handle ("" if none); h == NULL: last apss_create failure */
const char *apss_last_error(const apss_handle *h);
/* Run the handle's work on a caller-owned HIP stream (hipStream_t as void*; NULL = the device's default stream),
 * or, with use_own != 0, go back to the handle's own stream.  The handle's own stream is a blocking stream: it
 * orders itself after work already queued on the default stream (where PyTorch runs unless told otherwise). */
int32_t apss_set_stream(apss_handle *h, void *hip_stream, int32_t use_own);

/* ---- host-pointer entry points (what the JNI shim binds) ----
 * CSR batch: n rows; row i = indices[rowptr[i] .. rowptr[i+1]) strictly increasing in [0, dim), values alike
 * (doubles, as SparkSparseVector.values), ext_ids[i] = the caller's integer handle for the String id. */

/* buildInvertedIndex(batch), IWA:61-71 */
int32_t apss_insert(apss_handle *h, int64_t n, const int64_t *rowptr, const int32_t *indices,
                    const double *values, const int64_t *ext_ids);
/* querySimilarItems(batch) on a frozen index (stopUpdateIndex, IWA:125-127); unseen dims are empty lists */
int32_t apss_query(apss_handle *h, int64_t n, const int64_t *rowptr, const int32_t *indices,
                   const double *values, const int64_t *ext_ids, int64_t *n_results);
/* the IndexData handler: build, then query; the batch sees itself and everything before it (IWA:125-133) */
int32_t apss_insert_and_query(apss_handle *h, int64_t n, const int64_t *rowptr, const int32_t *indices,
                              const double *values, const int64_t *ext_ids, int64_t *n_results);
/* query every stored vector against the whole index (single-batch self-join, the benchmark driver) */
int32_t apss_self_join(apss_handle *h, int64_t *n_results);

/* results of the last query-type call: (query ext id, candidate ext id, score); order unspecified */
int32_t apss_result_count(const apss_handle *h, int64_t *n_results);
int32_t apss_fetch_results(apss_handle *h, int64_t offset, int64_t count, int64_t *out_q, int64_t *out_c,
                           float *out_score);

int32_t apss_size(const apss_handle *h, int64_t *rows, int64_t *nnz);
int32_t apss_stats_get(apss_handle *h, apss_stats *out);

/* ---- device-pointer entry points (inputs already resident in HBM: bench.py, multi-GPU host code) ----
 * d_rowptr int64[n+1] (relative to the batch: d_rowptr[0] == 0), d_indices int32[nnz], d_values float[nnz],
 * d_ext_ids int64[n]; all on the handle's device.  Work is enqueued on the handle's stream; the calls that
 * return counts synchronise that stream. */
int32_t apss_insert_dev(apss_handle *h, int64_t n, int64_t nnz, const int64_t *d_rowptr,
                        const int32_t *d_indices, const float *d_values, const int64_t *d_ext_ids);
int32_t apss_query_dev(apss_handle *h, int64_t n, int64_t nnz, const int64_t *d_rowptr,
                       const int32_t *d_indices, const float *d_values, const int64_t *d_ext_ids,
                       int64_t *n_results);
int32_t apss_insert_and_query_dev(apss_handle *h, int64_t n, int64_t nnz, const int64_t *d_rowptr,
                                  const int32_t *d_indices, const float *d_values, const int64_t *d_ext_ids,
                                  int64_t *n_results);
/* drop the index and the store, keep the reservation (lets a benchmark step re-run build + join) */
int32_t apss_clear(apss_handle *h);
/* device views of the last results: int32 query row (within the last query batch), int32 candidate slot,
 * float score; valid until the next query-type call */
int32_t apss_results_dev(apss_handle *h, const int32_t **d_q_row, const int32_t **d_c_slot,
                         const float **d_score, int64_t *n_results);
/* the same, copied device-to-device into caller-owned HBM buffers (any of the three may be NULL) on the handle's
 * stream: how multi-GPU host code keeps the candidate lists on the device between the phases of a sharded join */
int32_t apss_results_copy_dev(apss_handle *h, int64_t offset, int64_t count, int32_t *d_q_row, int32_t *d_c_slot,
                              float *d_score);

/* ---- term-range shards (multi-GPU; one handle per GPU owning dims [term_lo, term_hi)) ----
 * Phase 1, local: a query-type call on a shard handle reports CANDIDATES, pairs whose local partial score
 * p_g satisfies p_g >= theta * |q_g||c_g| / (|q||c|)  (|x_g| = L2 norm of x restricted to the shard's dims, |x| =
 * norm of the whole row as the caller handed it in, after the value prune).  Every pair with total score >= theta
 * passes this test on at least one shard, whatever the row norms and whatever the signs of the weights:
 * p_g <= |q_g||c_g| and sum_g |q_g||c_g| <= |q||c| (Cauchy-Schwarz twice), so if every shard failed the test the
 * total would be below theta.  The caller must hand every shard the WHOLE row (the library restricts it).
 * Phase 2: the host unions the candidate lists of all shards, every shard fills its exact partial score for
 * each pair with apss_partial_scores_dev, the partials are summed across GPUs (RCCL all-reduce) and
 * thresholded.  Pairs are (query row of the last query batch, candidate slot). */
int32_t apss_partial_scores_dev(apss_handle *h, int64_t n_pairs, const int32_t *d_q_row,
                                const int32_t *d_c_slot, float *d_out_partial);


/* ---- dense-head block set by the caller ----
 * The `n_terms` (<= 32768; more than 256: the first 128 with a column each, the others folded, in the order given) most frequent terms are scored by an MFMA contraction (INT8 rows rounded up, or bf16: apss_stats.head_int8) over W = [rows x n_terms] instead of
 * their posting lists (CommonUtils.scala:110-115 restricted to those dims; a FILTER: survivors are re-scored exactly).
 * On a plain handle this replaces the library's own choice (apss_config.head_terms).  On a TERM SHARD it is the only way
 * to get a block: every shard of a join must be given the SAME terms -- they are a part of their own, {H, T_1 .. T_T}, in
 * the partition the candidate rule above is proved for, so a head term lies in no shard's inverted index (it stays in the
 * store of the shard whose range holds it, for the exact partial scores of phase 2) -- and shard `part` of `n_parts`
 * multiplies the 64-row candidate tiles t with t % n_parts == part of the contraction against the whole query batch
 * (the block is cut over the GPUs by candidate row, not by term).  A query-type call on such a shard reports the union of
 * both filters' candidates (duplicates possible).  Valid on an empty handle only (before the first insert / after
 * apss_clear); the setting survives apss_clear; n_terms == 0 removes it.  Shards: non-negative weights, no
 * APSS_FLAG_ADMISSION (APSS_E_UNSUPPORTED otherwise). */
int32_t apss_set_head_terms(apss_handle *h, int32_t n_terms, const int32_t *terms, int32_t part, int32_t n_parts);
/* How many of the 256 columns of a head of more than 256 terms are FOLDED columns (64 | 128 | 192; 0 = the default, 128):
 * the 256 - columns most frequent terms keep a column each, the others share the folded ones (L2 norm per column).  Takes effect at the next
 * apss_set_head_terms; on an empty handle. */
int32_t apss_set_head_fold(apss_handle *h, int32_t columns);
/* the block's terms in block order (chosen by the library or set by the caller); *n_terms = how many there are */
int32_t apss_get_head_terms(apss_handle *h, int32_t capacity, int32_t *out_terms, int32_t *n_terms);
/* device views of the external ids: int64[rows] of the store, int64[query rows] of the last query batch (NULL before
 * any query-type call / after an insert); valid until the next insert, query-type call or clear */
int32_t apss_ext_ids_dev(apss_handle *h, const int64_t **d_store_ext, const int64_t **d_query_ext);

/* =====================================================================================================================
 * apss_group: the term-sharded index of one node -- T member shards, one per GPU -- behind ONE object.
 *
 * The reference shards its inverted index by term inside the server: WriteWorkerActor buckets every vector by
 * dim % maxShardNum and flushes one DataPacket per shard (WriteWorkerActor.scala:164-183), EntryProxyActor fans a packet out
 * to maxIndexEntryActorNum IndexingWorkerActors by dim % maxIndexEntryActorNum (EntryProxyActor.scala:37-49), and every
 * worker handles IndexData on its own (IndexingWorkerActor.scala:122-137), re-scoring the full vectors it was sent.  A group
 * is that fan-out and its workers on the GPUs of one node: member g owns a contiguous term RANGE (cut on the first batch,
 * balanced by sum df^2), stores only that slice of every vector and runs the member-local phase of the join; the group
 * combines the members' answers with the exchange the reference does not need because it replicates whole vectors:
 *
 *   1. every member, concurrently (one host thread per member): apss_insert_and_query_dev / apss_query_dev on its shard
 *      handle -> CANDIDATE pairs (the shard rule above apss_partial_scores_dev);
 *   2. all-gather of the members' candidate lists (8-B keys), sorted union on every member;
 *   3. every member: exact partial score of every pair of the union (apss_partial_scores_dev);
 *   4. all-reduce(SUM) of the partial scores, `>= theta` (IWA:93), compaction on member 0.
 *
 * Exchange (steps 2 and 4): RCCL over xGMI (ncclBroadcast-grouped all-gather, ncclAllReduce on the members' streams; librccl
 * is loaded on first use) when every member has a GPU of its own; members that SHARE a device (tests on a one-GPU box) are
 * combined by device-to-device copies and a summing kernel -- same lists, same order, same results.
 * On skewed data the members share one dense-head block (apss_set_head_terms): member 0's device runs the library's policy
 * on a sample of the first batch and every member is given the same terms; apss_config.head_terms = -1 disables it.
 *
 * One thread at a time inside a group (it is one actor's state); results stay in the group until the next query-type
 * call, insert or clear.  apss_config: dim, theta, index_threshold, flags, tile_rows, head_terms and the capacity hints
 * apply to every member; device_id, term_lo, term_hi are ignored (the group sets them).  APSS_FLAG_ADMISSION is not
 * supported with more than one member and a dense-head block (as on a single shard handle).
 * ===================================================================================================================== */
typedef struct apss_group apss_group;

#define APSS_GROUP_FORCE_EXCHANGE 1u /* run the exchange (steps 2-4) even for a group of ONE member, whose handle holds the whole
                                        term space and whose answer is already final (test hook: the RCCL path on one GPU) */
#define APSS_GROUP_NO_RCCL 2u        /* never load RCCL: combine the members with device-to-device copies (peer access or staging
                                        through the host) even when every member has its own GPU */

#define APSS_EXCHANGE_NONE 0   /* one member, its answer is final */
#define APSS_EXCHANGE_COPIES 1 /* device-to-device copies + summing kernel (members share a device, or APSS_GROUP_NO_RCCL) */
#define APSS_EXCHANGE_RCCL 2   /* RCCL collectives on the members' streams */

#define APSS_GROUP_MAX_MEMBERS 64

typedef struct apss_group_stats {
  int32_t struct_size;        /* IN: sizeof(apss_group_stats) of the caller; OUT: bytes written (as apss_stats) */
  int32_t n_members;
  int32_t exchange;           /* APSS_EXCHANGE_* the last query-type call used */
  int32_t head_terms;         /* terms in the members' shared dense-head block (0: none) */
  int64_t rows;               /* vectors in the store (every member holds its slice of each) */
  int64_t nnz;                /* postings over all members */
  int64_t posting_visits;     /* sum over the members, last query-type call (reference-equivalent: see apss_stats) */
  int64_t device_posting_visits;
  int64_t member_touched_pairs; /* sum over the members of apss_stats.candidate_pairs (a pair sharing terms in k ranges counts k times) */
  int64_t candidates_sum;     /* sum over the members of the candidate lists' lengths (step 1) */
  int64_t candidates_max;     /* ... the longest list */
  int64_t union_pairs;        /* distinct candidates (step 2) */
  int64_t result_pairs;       /* pairs >= theta */
  int64_t all_gather_bytes;   /* bytes every member RECEIVES in step 2: 8 x (candidates_sum - its own) at most */
  int64_t all_reduce_bytes;   /* 4 x union_pairs: the vector of step 4 */
  double member_ms_max;       /* wall time of step 1, slowest member (ingest + build + filter kernels + their syncs) */
  double probe_ms_max;        /* apss_stats.probe_ms, slowest member */
  double build_ms_max;
  double head_ms_max;
  double exchange_ms;         /* wall time of steps 2-4 (gather, union, partial scores, reduce, threshold) */
  double partial_ms_max;      /* ... of which apss_partial_scores_dev, slowest member */
  double total_ms;            /* wall time of the call */
  int32_t term_cuts[APSS_GROUP_MAX_MEMBERS + 1]; /* member g owns terms [term_cuts[g], term_cuts[g + 1]) (head terms excepted) */
} apss_group_stats;

/* n_members >= 1 shards on the HIP devices device_ids[0 .. n_members) (a device may appear more than once) */
int32_t apss_group_create(const apss_config *cfg, int32_t n_members, const int32_t *device_ids, uint32_t group_flags,
                          apss_group **out);
void apss_group_destroy(apss_group *g);
/* message of the last failing call ("" if none); g == NULL: last apss_group_create failure */
const char *apss_group_last_error(const apss_group *g);
/* Name the term ranges instead of letting the first batch decide: cuts[0] = 0 < cuts[1] < .. < cuts[n_members] = dim.
 * Before the group's first insert only (the members' handles are created with their ranges then). */
int32_t apss_group_set_term_cuts(apss_group *g, const int32_t *cuts);

/* host-pointer entry points (what the JNI shim binds): the CSR batch of apss_insert / apss_query / apss_insert_and_query,
 * handed WHOLE to every member (each keeps its term range) */
int32_t apss_group_insert(apss_group *g, int64_t n, const int64_t *rowptr, const int32_t *indices, const double *values,
                          const int64_t *ext_ids);
int32_t apss_group_query(apss_group *g, int64_t n, const int64_t *rowptr, const int32_t *indices, const double *values,
                         const int64_t *ext_ids, int64_t *n_results);
int32_t apss_group_insert_and_query(apss_group *g, int64_t n, const int64_t *rowptr, const int32_t *indices,
                                    const double *values, const int64_t *ext_ids, int64_t *n_results);
/* device-pointer entry point: member m reads the batch from d_rowptr[m], d_indices[m], d_values[m], d_ext_ids[m], resident
 * on ITS device (members that share a device may be given the same pointers); layouts as apss_insert_and_query_dev */
int32_t apss_group_insert_and_query_dev(apss_group *g, int64_t n, int64_t nnz, const int64_t *const *d_rowptr,
                                        const int32_t *const *d_indices, const float *const *d_values,
                                        const int64_t *const *d_ext_ids, int64_t *n_results);
/* drop index and store of every member; the reservations and the LAYOUT stay -- term cuts and dense-head terms, whether named
 * by the caller or decided from the first batch the group ever saw (they are configuration, like apss_set_head_terms on a
 * handle: a benchmark step re-runs build + join on the same layout); a new layout needs a new group */
int32_t apss_group_clear(apss_group *g);

int32_t apss_group_result_count(const apss_group *g, int64_t *n_results);
/* (query ext id, candidate ext id, score) of the last query-type call, as apss_fetch_results */
int32_t apss_group_fetch_results(apss_group *g, int64_t offset, int64_t count, int64_t *out_q, int64_t *out_c,
                                 float *out_score);
int32_t apss_group_stats_get(apss_group *g, apss_group_stats *out);
/* the shard handle's own statistics (out->struct_size set by the caller, as apss_stats_get) */
int32_t apss_group_member_stats(apss_group *g, int32_t member, apss_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* APSS_H */
