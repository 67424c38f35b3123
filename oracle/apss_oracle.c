/*
 * apss_oracle.c -- CPU restatement of the reference hot path.  See apss_oracle.h for the scope note:
 * TEST INFRASTRUCTURE ONLY, "parity unpinned" by the reference (it has no tests / golden vectors).
 *
 * Every function cites the reference lines it follows; paths are relative to
 * /root/reference/core/src/main/scala/cpslab/.
 */
#define _GNU_SOURCE
#include "apss_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ small containers */

typedef struct {
  int64_t *data;
  int64_t len, cap;
} vec_i64;
typedef struct {
  int32_t *data;
  int64_t len, cap;
} vec_i32;
typedef struct {
  double *data;
  int64_t len, cap;
} vec_f64;

#define VEC_PUSH(v, T, x)                                              \
  do {                                                                 \
    if ((v)->len == (v)->cap) {                                        \
      (v)->cap = (v)->cap ? (v)->cap * 2 : 16;                         \
      (v)->data = (T *)realloc((v)->data, (size_t)(v)->cap * sizeof(T)); \
    }                                                                  \
    (v)->data[(v)->len++] = (x);                                       \
  } while (0)

/* int64 -> int64 open-addressing map (keys are vector ids; the reference keys its maps by String id) */
typedef struct {
  int64_t *keys, *vals;
  uint8_t *used;
  int64_t cap, len;
} map_i64;

static uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

static void map_init(map_i64 *m, int64_t cap) {
  int64_t c = 16;
  while (c < cap * 2) c <<= 1;
  m->cap = c;
  m->len = 0;
  m->keys = (int64_t *)malloc((size_t)c * sizeof(int64_t));
  m->vals = (int64_t *)malloc((size_t)c * sizeof(int64_t));
  m->used = (uint8_t *)calloc((size_t)c, 1);
}
static void map_free(map_i64 *m) {
  free(m->keys);
  free(m->vals);
  free(m->used);
  memset(m, 0, sizeof(*m));
}
static int64_t *map_find(const map_i64 *m, int64_t key) {
  uint64_t h = mix64((uint64_t)key) & (uint64_t)(m->cap - 1);
  while (m->used[h]) {
    if (m->keys[h] == key) return &m->vals[h];
    h = (h + 1) & (uint64_t)(m->cap - 1);
  }
  return NULL;
}
static void map_put(map_i64 *m, int64_t key, int64_t val);
static void map_grow(map_i64 *m) {
  map_i64 n;
  map_init(&n, m->cap);
  for (int64_t i = 0; i < m->cap; ++i)
    if (m->used[i]) map_put(&n, m->keys[i], m->vals[i]);
  map_free(m);
  *m = n;
}
static void map_put(map_i64 *m, int64_t key, int64_t val) {
  if ((m->len + 1) * 2 > m->cap) map_grow(m);
  uint64_t h = mix64((uint64_t)key) & (uint64_t)(m->cap - 1);
  while (m->used[h]) {
    if (m->keys[h] == key) {
      m->vals[h] = val;
      return;
    }
    h = (h + 1) & (uint64_t)(m->cap - 1);
  }
  m->used[h] = 1;
  m->keys[h] = key;
  m->vals[h] = val;
  m->len++;
}

/* ------------------------------------------------------------------ CU:98-117 */

/*
 * CommonUtils.calculateSimilarity(SparkSparseVector, SparkSparseVector), CU:98-117.
 * The reference builds a HashMap of each vector and sums value1*value2 over vector1's entries found in
 * vector2, in Double.  The sum is taken here in ascending index order of vector1 (the Scala HashMap's
 * iteration order is not restated; the difference is at most the reordering of a <= nnz-term double sum).
 */
double oracle_calculate_similarity(int32_t size1, int32_t n1, const int32_t *idx1, const double *val1,
                                   int32_t size2, int32_t n2, const int32_t *idx2, const double *val2) {
  if (size1 != size2) return NAN; /* require(vector1.size == vector2.size), CU:99 */
  double similarity = 0.0;
  int32_t i = 0, j = 0;
  while (i < n1 && j < n2) {
    if (idx1[i] < idx2[j]) {
      ++i;
    } else if (idx1[i] > idx2[j]) {
      ++j;
    } else {
      similarity += val1[i] * val2[j];
      ++i;
      ++j;
    }
  }
  return similarity;
}

/* ------------------------------------------------------------------ one IndexingWorkerActor */

struct oracle_worker {
  int32_t dim;
  double theta;
  int32_t mode;
  /* vectorsStore: ListBuffer[SparseVectorWrapper] (IWA:22), append-only */
  vec_i64 ids;  /* sparseVector._1 */
  vec_i64 uid;  /* dense renumbering of ids, for the contains() stamps */
  vec_i64 vptr; /* len = n+1 */
  vec_i32 vidx;
  vec_f64 vval;
  /* invertedIndex: HashMap[Int, HashSet[Int]] (IWA:25): dim -> slots */
  vec_i32 *postings; /* [dim] */
  map_i64 id2uid;
  vec_i64 stamp; /* per uid: token of the output entry that contains it */
  int64_t next_token;
  int64_t sim_calls;
  /* output of the last call */
  vec_i64 out_q, out_c;
  vec_f64 out_s;
};

oracle_worker *oracle_worker_create(int32_t dim, double theta, int32_t mode) {
  if (dim <= 0) return NULL;
  oracle_worker *w = (oracle_worker *)calloc(1, sizeof(*w));
  w->dim = dim;
  w->theta = theta;
  w->mode = mode;
  w->postings = (vec_i32 *)calloc((size_t)dim, sizeof(vec_i32));
  VEC_PUSH(&w->vptr, int64_t, 0);
  map_init(&w->id2uid, 1024);
  w->next_token = 1;
  return w;
}

void oracle_worker_destroy(oracle_worker *w) {
  if (!w) return;
  for (int32_t d = 0; d < w->dim; ++d) free(w->postings[d].data);
  free(w->postings);
  free(w->ids.data);
  free(w->uid.data);
  free(w->vptr.data);
  free(w->vidx.data);
  free(w->vval.data);
  free(w->stamp.data);
  free(w->out_q.data);
  free(w->out_c.data);
  free(w->out_s.data);
  map_free(&w->id2uid);
  free(w);
}

int64_t oracle_worker_size(const oracle_worker *w) { return w->ids.len; }

static int64_t worker_uid(oracle_worker *w, int64_t id) {
  int64_t *p = map_find(&w->id2uid, id);
  if (p) return *p;
  int64_t u = w->id2uid.len;
  map_put(&w->id2uid, id, u);
  VEC_PUSH(&w->stamp, int64_t, 0);
  return u;
}

typedef struct {
  int64_t q, c;
  double s;
  int64_t seq;
} triple;

static int triple_cmp(const void *a, const void *b) {
  const triple *x = (const triple *)a, *y = (const triple *)b;
  if (x->q != y->q) return x->q < y->q ? -1 : 1;
  if (x->c != y->c) return x->c < y->c ? -1 : 1;
  return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}

/* sort by (q, c, seq), keep the LAST write per key: HashMap `+=` / `++=` overwrite semantics (IWA:94,106-107) */
static int64_t triples_finalize(triple *t, int64_t n, vec_i64 *oq, vec_i64 *oc, vec_f64 *os) {
  qsort(t, (size_t)n, sizeof(triple), triple_cmp);
  oq->len = oc->len = os->len = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (i + 1 < n && t[i + 1].q == t[i].q && t[i + 1].c == t[i].c) continue;
    VEC_PUSH(oq, int64_t, t[i].q);
    VEC_PUSH(oc, int64_t, t[i].c);
    VEC_PUSH(os, double, t[i].s);
  }
  return oq->len;
}

int64_t oracle_worker_index_data(oracle_worker *w, int64_t n, const int64_t *ids, const int64_t *rowptr,
                                 const int32_t *indices, const double *values, const int64_t *lptr,
                                 const int32_t *ldims, int32_t query_only, const int64_t **out_q,
                                 const int64_t **out_c, const double **out_sim) {
  if (!w || n < 0 || (n > 0 && (!ids || !rowptr))) return -1;
  for (int64_t i = 0; i < n; ++i) {
    int32_t prev = -1;
    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      if (indices[k] <= prev || indices[k] >= w->dim) return -1; /* SV:75 strictly increasing, < size */
      prev = indices[k];
    }
    if (lptr)
      for (int64_t k = lptr[i]; k < lptr[i + 1]; ++k)
        if (ldims[k] < 0 || ldims[k] >= w->dim) return -1;
  }

  /* ---- buildInvertedIndex (IWA:61-71): the whole batch is indexed before any query runs ---- */
  if (query_only != 1) {
    for (int64_t i = 0; i < n; ++i) {
      /* vectorsStore += candidateVector; currentIdx = vectorsStore.length - 1 */
      int32_t slot = (int32_t)w->ids.len;
      VEC_PUSH(&w->ids, int64_t, ids[i]);
      VEC_PUSH(&w->uid, int64_t, worker_uid(w, ids[i]));
      for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        VEC_PUSH(&w->vidx, int32_t, indices[k]);
        VEC_PUSH(&w->vval, double, values[k]);
      }
      VEC_PUSH(&w->vptr, int64_t, w->vidx.len);
      /* for (dim <- candidateVector.indices) invertedIndex.getOrElseUpdate(dim, HashSet) += currentIdx */
      if (lptr) {
        for (int64_t k = lptr[i]; k < lptr[i + 1]; ++k) VEC_PUSH(&w->postings[ldims[k]], int32_t, slot);
      } else {
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) VEC_PUSH(&w->postings[indices[k]], int32_t, slot);
      }
    }
  }

  if (query_only == 2) { /* build only: the warm-up of a latency run indexes the data set without reading answers */
    if (out_q) *out_q = NULL;
    if (out_c) *out_c = NULL;
    if (out_sim) *out_sim = NULL;
    return 0;
  }

  /* ---- querySimilarItems (IWA:74-111) ---- */
  vec_i64 ent_q = {0};   /* outputSimSet keys, in creation order */
  vec_i64 ent_tok = {0}; /* token of each entry (stamp value meaning "contained in this entry") */
  map_i64 ent_of;        /* qid -> entry index */
  map_init(&ent_of, n + 8);
  triple *tr = NULL;
  int64_t ntr = 0, captr = 0;
  int64_t rc = 0;
  int dup_qid = 0; /* does any id occur twice in the batch? (they share one outputSimSet entry) */
  {
    map_i64 seen;
    map_init(&seen, n + 8);
    for (int64_t i = 0; i < n; ++i) {
      if (map_find(&seen, ids[i])) dup_qid = 1;
      map_put(&seen, ids[i], 1);
    }
    map_free(&seen);
  }

  for (int64_t i = 0; i < n && rc == 0; ++i) {
    const int64_t qid = ids[i];
    const int32_t qn = (int32_t)(rowptr[i + 1] - rowptr[i]);
    const int32_t *qi = indices + rowptr[i];
    const double *qv = values + rowptr[i];
    const int64_t nl = lptr ? lptr[i + 1] - lptr[i] : qn;
    const int32_t *ld = lptr ? ldims + lptr[i] : qi;
    const int64_t q_uid = worker_uid(w, qid);
    (void)q_uid;
    for (int64_t k = 0; k < nl; ++k) { /* for (dim <- candidateVector.indices) IWA:102 */
      const int32_t d = ld[k];
      const vec_i32 *cands = &w->postings[d];
      if (cands->len == 0) {
        /* invertedIndex(dim) on a missing key throws NoSuchElementException (IWA:104); the handler
         * swallows it and the whole batch's output is lost (IWA:135-137).  Intended: empty list (Q8). */
        if (w->mode == ORACLE_MODE_AS_WRITTEN) {
          rc = -3;
          break;
        }
      }
      int64_t *entp = map_find(&ent_of, qid);
      /* querySimilarVectors(query, candidateList) IWA:80-99 */
      const int64_t first_new = ntr;
      for (int64_t p = 0; p < cands->len; ++p) {
        const int32_t slot = cands->data[p];
        const int64_t cid = w->ids.data[slot];
        /* IWA:89-91: outputSimSet.contains(q) && !outputSimSet(q).contains(c.id) && q.id != c.id */
        if (w->mode == ORACLE_MODE_AS_WRITTEN && !entp) continue; /* quirk Q1 */
        if (entp && w->stamp.data[w->uid.data[slot]] == ent_tok.data[*entp]) continue;
        if (entp && dup_qid) { /* same qid earlier in this batch: its stamps may have been overwritten since */
          int found = 0;
          for (int64_t t = 0; t < ntr && !found; ++t) found = (tr[t].q == qid && tr[t].c == cid);
          if (found) continue;
        }
        if (qid == cid) continue;
        const int64_t b = w->vptr.data[slot];
        const int32_t cn = (int32_t)(w->vptr.data[slot + 1] - b);
        /* calculateSimilarity(candidate, query), IWA:92 -> CU:90-117 (candidate is vector1) */
        const double sim = oracle_calculate_similarity(w->dim, cn, w->vidx.data + b, w->vval.data + b, w->dim,
                                                       qn, qi, qv);
        w->sim_calls++;
        if (sim >= w->theta) { /* IWA:93, inclusive */
          if (ntr == captr) {
            captr = captr ? captr * 2 : 1024;
            tr = (triple *)realloc(tr, (size_t)captr * sizeof(triple));
          }
          tr[ntr].q = qid;
          tr[ntr].c = cid;
          tr[ntr].s = sim;
          tr[ntr].seq = ntr;
          ntr++;
        }
      }
      /* outputSimSet.getOrElseUpdate(q.id, new HashMap) ++= similarVectors  (IWA:106-107) */
      if (!entp) {
        map_put(&ent_of, qid, ent_q.len);
        VEC_PUSH(&ent_q, int64_t, qid);
        VEC_PUSH(&ent_tok, int64_t, w->next_token++);
        entp = map_find(&ent_of, qid);
      }
      for (int64_t t = first_new; t < ntr; ++t) {
        int64_t *u = map_find(&w->id2uid, tr[t].c);
        w->stamp.data[*u] = ent_tok.data[*entp];
      }
    }
  }

  int64_t nout = 0;
  if (rc == 0) nout = triples_finalize(tr, ntr, &w->out_q, &w->out_c, &w->out_s);
  free(tr);
  free(ent_q.data);
  free(ent_tok.data);
  map_free(&ent_of);
  if (rc != 0) {
    w->out_q.len = w->out_c.len = w->out_s.len = 0;
    return rc;
  }
  if (out_q) *out_q = w->out_q.data;
  if (out_c) *out_c = w->out_c.data;
  if (out_sim) *out_sim = w->out_s.data;
  return nout;
}

/* ------------------------------------------------------------------ client / ingest side helpers */

/* LoadGenerator.generateVector, LG:34-37 */
void oracle_l2_normalize(int64_t n, const int64_t *rowptr, double *values) {
  for (int64_t i = 0; i < n; ++i) {
    double sum = 0.0;
    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) sum = sum + values[k] * values[k]; /* foldLeft */
    const double square_sum = sqrt(sum);
    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) values[k] = values[k] / square_sum;
  }
}

/* WriteWorkerActor.handleVectorIOMsg, WWA:188-194: filter value > threshold, Vectors.sparse re-sorts by index */
int64_t oracle_value_prune(int64_t n, const int64_t *rowptr, const int32_t *indices, const double *values,
                           double threshold, int64_t *out_rowptr, int32_t *out_indices, double *out_values) {
  int64_t o = 0;
  out_rowptr[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      if (values[k] > threshold) {
        out_indices[o] = indices[k];
        out_values[o] = values[k];
        ++o;
      }
    }
    out_rowptr[i + 1] = o;
  }
  return o;
}

/* EntryProxyActor.checkIfVectorToBeIndexed, EPA:81-93, with readMaxWeight == 1.0 for every dim (EPA:51-57):
 * calculateSimilarity(maxWeightedVector, v) = sum of v's values. */
void oracle_admission(int64_t n, const int64_t *rowptr, const double *values, double theta, uint8_t *keep) {
  for (int64_t i = 0; i < n; ++i) {
    double s = 0.0;
    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += 1.0 * values[k];
    keep[i] = (uint8_t)(s >= theta);
  }
}

/* ------------------------------------------------------------------ the cluster fan-out */

struct oracle_cluster {
  int32_t dim, mode, S, E, W;
  double theta;
  oracle_worker **workers; /* [E * W], created lazily (EPA:115-119) */
  vec_i64 out_q, out_c;
  vec_f64 out_s;
};

oracle_cluster *oracle_cluster_create(int32_t dim, double theta, int32_t mode, int32_t max_shard_num,
                                      int32_t max_entry_num, int32_t max_index_entry_actor_num) {
  if (dim <= 0 || max_shard_num <= 0 || max_entry_num <= 0 || max_index_entry_actor_num <= 0) return NULL;
  oracle_cluster *c = (oracle_cluster *)calloc(1, sizeof(*c));
  c->dim = dim;
  c->theta = theta;
  c->mode = mode;
  c->S = max_shard_num;
  c->E = max_entry_num;
  c->W = max_index_entry_actor_num;
  c->workers = (oracle_worker **)calloc((size_t)c->E * (size_t)c->W, sizeof(oracle_worker *));
  return c;
}

void oracle_cluster_destroy(oracle_cluster *c) {
  if (!c) return;
  for (int64_t i = 0; i < (int64_t)c->E * c->W; ++i) oracle_worker_destroy(c->workers[i]);
  free(c->workers);
  free(c->out_q.data);
  free(c->out_c.data);
  free(c->out_s.data);
  free(c);
}

int64_t oracle_cluster_sim_calls(const oracle_cluster *c) {
  int64_t s = 0;
  for (int64_t i = 0; i < (int64_t)c->E * c->W; ++i)
    if (c->workers[i]) s += c->workers[i]->sim_calls;
  return s;
}

int64_t oracle_cluster_flush(oracle_cluster *c, int64_t n, const int64_t *ids, const int64_t *rowptr,
                             const int32_t *indices, const double *values, const int64_t **out_q,
                             const int64_t **out_c, const double **out_sim) {
  triple *all = NULL;
  int64_t nall = 0, capall = 0;
  int64_t rc = 0;
  /* WriteWorkerActor.handleIOTrigger: one DataPacket per shard id (WWA:166-179) */
  for (int32_t s = 0; s < c->S && rc >= 0; ++s) {
    /* vectors of this packet: those with at least one dim % maxShardNum == s (WWA:172-173) */
    vec_i64 rows = {0};
    for (int64_t i = 0; i < n; ++i) {
      for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
        if (indices[k] % c->S == s) {
          VEC_PUSH(&rows, int64_t, i);
          break;
        }
    }
    if (rows.len == 0) {
      free(rows.data);
      continue; /* WWA:177 */
    }
    const int32_t entry = s % c->E; /* CU:32,36 */
    /* EntryProxyActor.spawnToIndexActor: every worker i gets every vector of the packet (EPA:41-46) */
    for (int32_t wi = 0; wi < c->W && rc >= 0; ++wi) {
      vec_i64 b_ids = {0}, b_ptr = {0}, b_lptr = {0};
      vec_i32 b_idx = {0}, b_ld = {0};
      vec_f64 b_val = {0};
      VEC_PUSH(&b_ptr, int64_t, 0);
      VEC_PUSH(&b_lptr, int64_t, 0);
      for (int64_t r = 0; r < rows.len; ++r) {
        const int64_t i = rows.data[r];
        VEC_PUSH(&b_ids, int64_t, ids[i]);
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
          VEC_PUSH(&b_idx, int32_t, indices[k]);
          VEC_PUSH(&b_val, double, values[k]);
          /* targetIndices (% maxShardNum == s) filtered by % maxIndexEntryActorNum == wi (EPA:43) */
          if (indices[k] % c->S == s && indices[k] % c->W == wi) VEC_PUSH(&b_ld, int32_t, indices[k]);
        }
        VEC_PUSH(&b_ptr, int64_t, b_idx.len);
        VEC_PUSH(&b_lptr, int64_t, b_ld.len);
      }
      oracle_worker **slot = &c->workers[(int64_t)entry * c->W + wi];
      if (!*slot) *slot = oracle_worker_create(c->dim, c->theta, c->mode);
      const int64_t *oq, *oc;
      const double *os;
      /* b_idx/b_val/b_ld may be empty (NULL data): give the worker a valid pointer anyway */
      int32_t dummy_i = 0;
      double dummy_d = 0;
      int64_t m = oracle_worker_index_data(*slot, b_ids.len, b_ids.data, b_ptr.data,
                                           b_idx.data ? b_idx.data : &dummy_i, b_val.data ? b_val.data : &dummy_d,
                                           b_lptr.data, b_ld.data ? b_ld.data : &dummy_i, 0, &oq, &oc, &os);
      if (m == -3) m = 0; /* exception swallowed, batch output lost (IWA:135-137) */
      if (m < 0) rc = m;
      for (int64_t t = 0; t < m; ++t) {
        if (nall == capall) {
          capall = capall ? capall * 2 : 1024;
          all = (triple *)realloc(all, (size_t)capall * sizeof(triple));
        }
        all[nall].q = oq[t];
        all[nall].c = oc[t];
        all[nall].s = os[t];
        all[nall].seq = nall;
        nall++;
      }
      free(b_ids.data);
      free(b_ptr.data);
      free(b_lptr.data);
      free(b_idx.data);
      free(b_ld.data);
      free(b_val.data);
    }
    free(rows.data);
  }
  int64_t nout = rc < 0 ? rc : triples_finalize(all, nall, &c->out_q, &c->out_c, &c->out_s);
  free(all);
  if (nout >= 0) {
    if (out_q) *out_q = c->out_q.data;
    if (out_c) *out_c = c->out_c.data;
    if (out_sim) *out_sim = c->out_s.data;
  }
  return nout;
}

/* ------------------------------------------------------------------ text format (SV:132-141, 204-205) */

/* Vectors.fromString: split(",\\["), strip "(", "]", "])", then toInt / toDouble. */
int64_t oracle_parse_sparse_vector(const char *text, int32_t *size, int32_t *indices, double *values, int64_t cap) {
  if (!text) return -1;
  const char *p1 = strstr(text, ",[");
  if (!p1) return -1;
  const char *p2 = strstr(p1 + 2, ",[");
  if (!p2) return -1;
  if (strstr(p2 + 2, ",[")) return -1; /* stringArray.length != 3 */
  /* size: text[0..p1) with "(" removed */
  char buf[64];
  int64_t bl = 0;
  for (const char *p = text; p < p1 && bl < 63; ++p)
    if (*p != '(') buf[bl++] = *p;
  buf[bl] = 0;
  char *end;
  long sz = strtol(buf, &end, 10);
  if (end == buf || *end != 0) return -1;
  *size = (int32_t)sz;
  /* indices: (p1+2 .. p2) with "]" removed, split(",") */
  int64_t ni = 0, nv = 0;
  const char *p = p1 + 2;
  while (p < p2) {
    while (p < p2 && (*p == ']' || *p == ',')) ++p;
    if (p >= p2) break;
    long v = strtol(p, &end, 10);
    if (end == p) return -1;
    if (ni < cap && indices) indices[ni] = (int32_t)v;
    ni++;
    p = end;
  }
  p = p2 + 2;
  while (*p) {
    while (*p == ']' || *p == ')' || *p == ',') ++p;
    if (!*p) break;
    double v = strtod(p, &end);
    if (end == p) return -1;
    if (nv < cap && values) values[nv] = v;
    nv++;
    p = end;
  }
  if (ni != nv) return -2; /* require(indices.length == values.length), SV:202 */
  return ni;
}

/* SparseVector.toString: "(%s,%s,%s)".format(size, indices.mkString("[", ",", "]"), values.mkString(...)).
 * Doubles are printed with %.17g (shortest round-trip printing a la Java's Double.toString is not restated;
 * parse(print(v)) == v holds bit-for-bit, which is what the round-trip tests check). */
int64_t oracle_print_sparse_vector(int32_t size, int64_t nnz, const int32_t *indices, const double *values,
                                   char *buf, int64_t cap) {
  int64_t o = 0;
  char tmp[64];
#define EMIT(s)                                        \
  do {                                                 \
    const char *s_ = (s);                              \
    for (; *s_; ++s_) {                                \
      if (buf && o + 1 < cap) buf[o] = *s_;            \
      ++o;                                             \
    }                                                  \
  } while (0)
  snprintf(tmp, sizeof tmp, "(%d,[", size);
  EMIT(tmp);
  for (int64_t i = 0; i < nnz; ++i) {
    snprintf(tmp, sizeof tmp, i ? ",%d" : "%d", indices[i]);
    EMIT(tmp);
  }
  EMIT("],[");
  for (int64_t i = 0; i < nnz; ++i) {
    snprintf(tmp, sizeof tmp, i ? ",%.17g" : "%.17g", values[i]);
    EMIT(tmp);
  }
  EMIT("])");
#undef EMIT
  if (buf && cap > 0) buf[o < cap ? o : cap - 1] = 0;
  return o;
}

/* ------------------------------------------------------------------ CPU baselines for bench.py */

typedef struct {
  int64_t *ptr; /* [dim+1] */
  int32_t *slot;
  double *w;
} csc_t;

static void csc_build(csc_t *c, int32_t dim, int64_t n, const int64_t *rowptr, const int32_t *indices,
                      const double *values) {
  const int64_t nnz = rowptr[n];
  c->ptr = (int64_t *)calloc((size_t)dim + 2, sizeof(int64_t));
  c->slot = (int32_t *)malloc((size_t)(nnz ? nnz : 1) * sizeof(int32_t));
  c->w = (double *)malloc((size_t)(nnz ? nnz : 1) * sizeof(double));
  for (int64_t k = 0; k < nnz; ++k) c->ptr[indices[k] + 2]++;
  for (int32_t d = 0; d < dim; ++d) c->ptr[d + 2] += c->ptr[d + 1];
  for (int64_t i = 0; i < n; ++i)
    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const int64_t o = c->ptr[indices[k] + 1]++;
      c->slot[o] = (int32_t)i;
      c->w[o] = values[k];
    }
}
static void csc_free(csc_t *c) {
  free(c->ptr);
  free(c->slot);
  free(c->w);
}

/* CU:101-116 with the two HashMap[Int, Double] actually built per call (this is where the reference's time goes) */
typedef struct {
  int32_t *k;
  double *v;
  int32_t cap;
} imap;
static void imap_reset(imap *m, int32_t n) {
  int32_t c = 8;
  while (c < 2 * n) c <<= 1;
  if (c > m->cap) {
    m->k = (int32_t *)realloc(m->k, (size_t)c * sizeof(int32_t));
    m->v = (double *)realloc(m->v, (size_t)c * sizeof(double));
  }
  m->cap = c;
  memset(m->k, 0xff, (size_t)c * sizeof(int32_t));
}
static inline void imap_put(imap *m, int32_t key, double val) {
  uint32_t h = ((uint32_t)key * 2654435761u) & (uint32_t)(m->cap - 1);
  while (m->k[h] != -1 && m->k[h] != key) h = (h + 1) & (uint32_t)(m->cap - 1);
  m->k[h] = key;
  m->v[h] = val;
}
static inline int imap_get(const imap *m, int32_t key, double *val) {
  uint32_t h = ((uint32_t)key * 2654435761u) & (uint32_t)(m->cap - 1);
  while (m->k[h] != -1) {
    if (m->k[h] == key) {
      *val = m->v[h];
      return 1;
    }
    h = (h + 1) & (uint32_t)(m->cap - 1);
  }
  return 0;
}
static double hashmap_dot(imap *m1, imap *m2, int32_t n1, const int32_t *i1, const double *v1, int32_t n2,
                          const int32_t *i2, const double *v2) {
  imap_reset(m1, n1);
  imap_reset(m2, n2);
  for (int32_t i = 0; i < n1; ++i) imap_put(m1, i1[i], v1[i]); /* CU:104-106 */
  for (int32_t i = 0; i < n2; ++i) imap_put(m2, i2[i], v2[i]); /* CU:107-109 */
  double similarity = 0.0;
  for (int32_t h = 0; h < m1->cap; ++h) { /* for ((idx, value) <- vector1Map), CU:110 */
    if (m1->k[h] == -1) continue;
    double other;
    if (imap_get(m2, m1->k[h], &other)) similarity += m1->v[h] * other;
  }
  return similarity;
}

typedef struct {
  int32_t variant, dim;
  double theta;
  int64_t n;
  const int64_t *rowptr;
  const int32_t *indices;
  const double *values;
  const csc_t *csc;
  int64_t q_begin, q_end;
  int32_t term_mod, term_rem; /* variant 2: this worker owns the dims with dim % term_mod == term_rem (0: all dims) */
  int64_t pairs, cands, visits;
  /* optional pair output (optcpu only) */
  vec_i64 *oq, *oc;
  vec_f64 *os;
} job_t;

static void *job_run(void *arg) {
  job_t *j = (job_t *)arg;
  const int64_t n = j->n;
  int64_t *stamp = (int64_t *)malloc((size_t)(n ? n : 1) * sizeof(int64_t));
  memset(stamp, 0xff, (size_t)(n ? n : 1) * sizeof(int64_t));
  double *acc = NULL;
  int32_t *touched = NULL;
  imap m1 = {0}, m2 = {0};
  if (j->variant == 1) {
    acc = (double *)calloc((size_t)(n ? n : 1), sizeof(double));
    touched = (int32_t *)malloc((size_t)(n ? n : 1) * sizeof(int32_t));
  }
  for (int64_t q = j->q_begin; q < j->q_end; ++q) {
    const int64_t qb = j->rowptr[q];
    const int32_t qn = (int32_t)(j->rowptr[q + 1] - qb);
    const int32_t *qi = j->indices + qb;
    const double *qv = j->values + qb;
    int64_t nt = 0;
    for (int32_t k = 0; k < qn; ++k) {
      const int32_t d = qi[k];
      if (j->term_mod > 0 && d % j->term_mod != j->term_rem) continue; /* another worker's dim (EPA:43) */
      const int64_t pb = j->csc->ptr[d], pe = j->csc->ptr[d + 1];
      j->visits += pe - pb;
      if (j->variant == 0 || j->variant == 2) {
        for (int64_t p = pb; p < pe; ++p) {
          const int32_t c = j->csc->slot[p];
          if (stamp[c] == q) continue; /* scored once per (q, c): the intended de-dup of IWA:90 */
          stamp[c] = q;
          if (c == q) continue; /* q.id != c.id, IWA:91 */
          const int64_t cb = j->rowptr[c];
          const double sim = hashmap_dot(&m1, &m2, (int32_t)(j->rowptr[c + 1] - cb), j->indices + cb,
                                         j->values + cb, qn, qi, qv);
          j->cands++;
          if (sim >= j->theta) j->pairs++;
        }
      } else {
        const double wq = qv[k];
        for (int64_t p = pb; p < pe; ++p) {
          const int32_t c = j->csc->slot[p];
          if (stamp[c] != q) {
            stamp[c] = q;
            touched[nt++] = c;
          }
          acc[c] += wq * j->csc->w[p];
        }
      }
    }
    if (j->variant == 1) {
      for (int64_t t = 0; t < nt; ++t) {
        const int32_t c = touched[t];
        const double sim = acc[c];
        acc[c] = 0.0;
        if (c == q) continue;
        j->cands++;
        if (sim >= j->theta) {
          j->pairs++;
          if (j->oq) {
            VEC_PUSH(j->oq, int64_t, q);
            VEC_PUSH(j->oc, int64_t, (int64_t)c);
            VEC_PUSH(j->os, double, sim);
          }
        }
      }
    }
  }
  free(stamp);
  free(acc);
  free(touched);
  free(m1.k);
  free(m1.v);
  free(m2.k);
  free(m2.v);
  return NULL;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int64_t oracle_selfjoin_sample(int32_t variant, int32_t dim, double theta, int64_t n, const int64_t *rowptr,
                               const int32_t *indices, const double *values, int64_t q_begin, int64_t q_end,
                               int32_t n_threads, int64_t *cand_pairs, int64_t *visits, double *seconds) {
  if (n_threads < 1) n_threads = 1;
  if (q_begin < 0 || q_end > n || q_begin > q_end) return -1;
  csc_t csc;
  csc_build(&csc, dim, n, rowptr, indices, values);
  job_t *jobs = (job_t *)calloc((size_t)n_threads, sizeof(job_t));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  const int64_t nq = q_end - q_begin;
  const double t0 = now_s();
  for (int32_t t = 0; t < n_threads; ++t) {
    jobs[t].variant = variant;
    jobs[t].dim = dim;
    jobs[t].theta = theta;
    jobs[t].n = n;
    jobs[t].rowptr = rowptr;
    jobs[t].indices = indices;
    jobs[t].values = values;
    jobs[t].csc = &csc;
    if (variant == 2) {
      /* refcpu-T, the reference's own parallelism: worker t of T owns the dims with dim % T == t (EntryProxyActor.scala:
       * 41-46), receives EVERY vector and recomputes the full dot product of every pair it reaches through one of its
       * dims (IWA:92) -- a pair sharing dims of k workers is scored k times.  cand_pairs then counts scorings,
       * duplicates included. */
      jobs[t].q_begin = q_begin;
      jobs[t].q_end = q_end;
      jobs[t].term_mod = n_threads;
      jobs[t].term_rem = t;
    } else {
      /* interleaved blocks of 8 queries would balance better, contiguous is what an actor's mailbox does */
      jobs[t].q_begin = q_begin + nq * t / n_threads;
      jobs[t].q_end = q_begin + nq * (t + 1) / n_threads;
    }
    pthread_create(&th[t], NULL, job_run, &jobs[t]);
  }
  int64_t pairs = 0, cands = 0, vis = 0;
  for (int32_t t = 0; t < n_threads; ++t) {
    pthread_join(th[t], NULL);
    pairs += jobs[t].pairs;
    cands += jobs[t].cands;
    vis += jobs[t].visits;
  }
  const double t1 = now_s();
  if (cand_pairs) *cand_pairs = cands;
  if (visits) *visits = vis;
  if (seconds) *seconds = t1 - t0;
  free(jobs);
  free(th);
  csc_free(&csc);
  return pairs;
}

int64_t oracle_selfjoin_pairs(int32_t dim, double theta, int64_t n, const int64_t *rowptr,
                              const int32_t *indices, const double *values, int64_t q_begin, int64_t q_end,
                              int64_t *out_q, int64_t *out_c, double *out_sim, int64_t cap) {
  if (q_begin < 0 || q_end > n || q_begin > q_end) return -1;
  csc_t csc;
  csc_build(&csc, dim, n, rowptr, indices, values);
  vec_i64 oq = {0}, oc = {0};
  vec_f64 os = {0};
  job_t j;
  memset(&j, 0, sizeof j);
  j.variant = 1;
  j.dim = dim;
  j.theta = theta;
  j.n = n;
  j.rowptr = rowptr;
  j.indices = indices;
  j.values = values;
  j.csc = &csc;
  j.q_begin = q_begin;
  j.q_end = q_end;
  j.oq = &oq;
  j.oc = &oc;
  j.os = &os;
  job_run(&j);
  const int64_t needed = oq.len;
  for (int64_t i = 0; i < needed && i < cap; ++i) {
    out_q[i] = oq.data[i];
    out_c[i] = oc.data[i];
    out_sim[i] = os.data[i];
  }
  free(oq.data);
  free(oc.data);
  free(os.data);
  csc_free(&csc);
  return needed;
}
