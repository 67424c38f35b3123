/*
 * apss_jni.c -- JNI shim: forwards cpslab.gpu.NativeApss 1:1 to the C ABI of include/apss.h.
 *
 * NOT compiled in this repository's image (no JDK / jni.h here; tests/test_jni_shim_syntax.py type-checks it against a
 * minimal declaration of the JNI functions it uses); a maintainer builds it next to the reference with
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *       all-pairs-similarity_amd/jvm/apss_jni.c -Lall-pairs-similarity_amd/csrc -lapss_hip -o libapss_jni.so
 * The shim holds no state and does no arithmetic: copy the arrays out of the Java heap, the call, the status code.
 *
 * No GetPrimitiveArrayCritical: the library calls block on H2D copies, kernels and stream syncs, and JNI forbids blocking
 * inside a critical region (the collector would stall for the whole join).  Batches are copied with Get<T>ArrayRegion
 * into C buffers first (a batch is a few MB; the PCIe copy that follows costs more), results go back with
 * Set<T>ArrayRegion.  Array lengths are checked against the CSR they describe before anything is dereferenced.
 */
#if defined(__has_include)
#if __has_include(<jni.h>)
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>

#include "apss.h"

#define H(x) ((apss_handle *)(intptr_t)(x))

JNIEXPORT jlong JNICALL Java_cpslab_gpu_NativeApss_create(JNIEnv *env, jclass cls, jint dim, jdouble theta,
                                                          jdouble indexThreshold, jint flags, jint device, jint headTerms) {
  (void)env; (void)cls;
  apss_config c = {0};
  c.struct_size = (int32_t)sizeof(c);
  c.dim = dim;
  c.theta = theta;
  c.index_threshold = indexThreshold;
  c.flags = (uint32_t)flags;
  c.device_id = device;
  c.head_terms = headTerms;
  apss_handle *h = 0;
  return apss_create(&c, &h) == APSS_OK ? (jlong)(intptr_t)h : 0;
}

JNIEXPORT void JNICALL Java_cpslab_gpu_NativeApss_destroy(JNIEnv *env, jclass cls, jlong h) {
  (void)env; (void)cls;
  apss_destroy(H(h));
}

JNIEXPORT jstring JNICALL Java_cpslab_gpu_NativeApss_lastError(JNIEnv *env, jclass cls, jlong h) {
  (void)cls;
  return (*env)->NewStringUTF(env, apss_last_error(H(h)));
}

#define G(x) ((apss_group *)(intptr_t)(x))

/* One CSR batch copied out of the Java heap, checked against the lengths it claims, handed to `call`, freed. */
typedef int32_t (*apss_submit_fn)(void *target, int mode, int64_t n, const int64_t *rp, const int32_t *ix, const double *vl,
                                  const int64_t *id, int64_t *n_res);

static jlong submit_csr(JNIEnv *env, void *target, apss_submit_fn call, jint mode, jlongArray rowptr, jintArray indices,
                        jdoubleArray values, jlongArray ids) {
  const jsize n = (*env)->GetArrayLength(env, ids);
  const jsize n_rp = (*env)->GetArrayLength(env, rowptr);
  const jsize n_ix = (*env)->GetArrayLength(env, indices);
  const jsize n_vl = (*env)->GetArrayLength(env, values);
  if (n_rp != n + 1 || n_ix != n_vl) return (jlong)APSS_E_INVALID; /* the CSR does not describe n rows */
  jlong *rp = (jlong *)malloc(sizeof(jlong) * (size_t)(n + 1));
  jlong *id = (jlong *)malloc(sizeof(jlong) * (size_t)(n > 0 ? n : 1));
  jint *ix = (jint *)malloc(sizeof(jint) * (size_t)(n_ix > 0 ? n_ix : 1));
  jdouble *vl = (jdouble *)malloc(sizeof(jdouble) * (size_t)(n_vl > 0 ? n_vl : 1));
  jlong out = (jlong)APSS_E_NOMEM;
  if (rp && id && ix && vl) {
    (*env)->GetLongArrayRegion(env, rowptr, 0, n + 1, rp);
    (*env)->GetLongArrayRegion(env, ids, 0, n, id);
    (*env)->GetIntArrayRegion(env, indices, 0, n_ix, ix);
    (*env)->GetDoubleArrayRegion(env, values, 0, n_vl, vl);
    if ((*env)->ExceptionCheck(env)) {
      out = (jlong)APSS_E_INVALID;
    } else if (rp[0] != 0 || rp[n] != (jlong)n_ix) { /* the library checks monotonicity and the indices themselves */
      out = (jlong)APSS_E_INVALID;
    } else {
      int64_t n_res = 0;
      const int32_t rc = call(target, mode, n, (const int64_t *)rp, (const int32_t *)ix, vl, (const int64_t *)id, &n_res);
      out = rc == APSS_OK ? (jlong)n_res : (jlong)rc;
    }
  }
  free(vl);
  free(ix);
  free(id);
  free(rp);
  return out;
}

typedef int32_t (*apss_fetch_fn)(void *target, int64_t count, int64_t *q, int64_t *c, float *s);

static jint fetch_triples(JNIEnv *env, void *target, apss_fetch_fn call, jlong count, jlongArray outQ, jlongArray outC,
                          jfloatArray outScore) {
  if (count < 0 || (*env)->GetArrayLength(env, outQ) < count || (*env)->GetArrayLength(env, outC) < count ||
      (*env)->GetArrayLength(env, outScore) < count)
    return APSS_E_INVALID;
  if (count == 0) return APSS_OK;
  int64_t *q = (int64_t *)malloc(sizeof(int64_t) * (size_t)count);
  int64_t *c = (int64_t *)malloc(sizeof(int64_t) * (size_t)count);
  float *s = (float *)malloc(sizeof(float) * (size_t)count);
  int32_t rc = APSS_E_NOMEM;
  if (q && c && s) {
    rc = call(target, count, q, c, s);
    if (rc == APSS_OK) {
      (*env)->SetLongArrayRegion(env, outQ, 0, (jsize)count, (const jlong *)q);
      (*env)->SetLongArrayRegion(env, outC, 0, (jsize)count, (const jlong *)c);
      (*env)->SetFloatArrayRegion(env, outScore, 0, (jsize)count, s);
    }
  }
  free(s);
  free(c);
  free(q);
  return rc;
}

static int32_t handle_submit(void *t, int mode, int64_t n, const int64_t *rp, const int32_t *ix, const double *vl, const int64_t *id,
                             int64_t *n_res) {
  if (mode == 0) return apss_insert((apss_handle *)t, n, rp, ix, vl, id);
  if (mode == 1) return apss_query((apss_handle *)t, n, rp, ix, vl, id, n_res);
  return apss_insert_and_query((apss_handle *)t, n, rp, ix, vl, id, n_res);
}

static int32_t handle_fetch(void *t, int64_t count, int64_t *q, int64_t *c, float *s) {
  return apss_fetch_results((apss_handle *)t, 0, count, q, c, s);
}

static int32_t group_submit(void *t, int mode, int64_t n, const int64_t *rp, const int32_t *ix, const double *vl, const int64_t *id,
                            int64_t *n_res) {
  if (mode == 0) return apss_group_insert((apss_group *)t, n, rp, ix, vl, id);
  if (mode == 1) return apss_group_query((apss_group *)t, n, rp, ix, vl, id, n_res);
  return apss_group_insert_and_query((apss_group *)t, n, rp, ix, vl, id, n_res);
}

static int32_t group_fetch(void *t, int64_t count, int64_t *q, int64_t *c, float *s) {
  return apss_group_fetch_results((apss_group *)t, 0, count, q, c, s);
}

/* mode 0 = insert, 1 = query (frozen index), 2 = insertAndQuery; returns the number of result triples or a negative status */
JNIEXPORT jlong JNICALL Java_cpslab_gpu_NativeApss_submit(JNIEnv *env, jclass cls, jlong h, jint mode, jlongArray rowptr,
                                                          jintArray indices, jdoubleArray values, jlongArray ids) {
  (void)cls;
  return submit_csr(env, H(h), handle_submit, mode, rowptr, indices, values, ids);
}

JNIEXPORT jint JNICALL Java_cpslab_gpu_NativeApss_fetch(JNIEnv *env, jclass cls, jlong h, jlong count, jlongArray outQ,
                                                        jlongArray outC, jfloatArray outScore) {
  (void)cls;
  return fetch_triples(env, H(h), handle_fetch, count, outQ, outC, outScore);
}

/* ---- apss_group: the term-sharded index of one node (one member per entry of `devices`) behind one object: what the
 * reference's DataPacket fan-out to maxShardNum x maxIndexEntryActorNum workers becomes on the GPUs of one host
 * (WriteWorkerActor.scala:164-183, EntryProxyActor.scala:37-49); the members' exchange runs below this boundary (RCCL). */
JNIEXPORT jlong JNICALL Java_cpslab_gpu_NativeApss_createGroup(JNIEnv *env, jclass cls, jint dim, jdouble theta,
                                                               jdouble indexThreshold, jint flags, jintArray devices,
                                                               jint headTerms, jint groupFlags) {
  (void)cls;
  const jsize n = (*env)->GetArrayLength(env, devices);
  if (n < 1 || n > APSS_GROUP_MAX_MEMBERS) return 0;
  jint dev[APSS_GROUP_MAX_MEMBERS];
  (*env)->GetIntArrayRegion(env, devices, 0, n, dev);
  if ((*env)->ExceptionCheck(env)) return 0;
  apss_config c = {0};
  c.struct_size = (int32_t)sizeof(c);
  c.dim = dim;
  c.theta = theta;
  c.index_threshold = indexThreshold;
  c.flags = (uint32_t)flags;
  c.head_terms = headTerms;
  apss_group *g = 0;
  return apss_group_create(&c, n, (const int32_t *)dev, (uint32_t)groupFlags, &g) == APSS_OK ? (jlong)(intptr_t)g : 0;
}

JNIEXPORT void JNICALL Java_cpslab_gpu_NativeApss_destroyGroup(JNIEnv *env, jclass cls, jlong g) {
  (void)env; (void)cls;
  apss_group_destroy(G(g));
}

JNIEXPORT jstring JNICALL Java_cpslab_gpu_NativeApss_groupLastError(JNIEnv *env, jclass cls, jlong g) {
  (void)cls;
  return (*env)->NewStringUTF(env, apss_group_last_error(G(g)));
}

JNIEXPORT jlong JNICALL Java_cpslab_gpu_NativeApss_groupSubmit(JNIEnv *env, jclass cls, jlong g, jint mode, jlongArray rowptr,
                                                               jintArray indices, jdoubleArray values, jlongArray ids) {
  (void)cls;
  return submit_csr(env, G(g), group_submit, mode, rowptr, indices, values, ids);
}

JNIEXPORT jint JNICALL Java_cpslab_gpu_NativeApss_groupFetch(JNIEnv *env, jclass cls, jlong g, jlong count, jlongArray outQ,
                                                             jlongArray outC, jfloatArray outScore) {
  (void)cls;
  return fetch_triples(env, G(g), group_fetch, count, outQ, outC, outScore);
}

/* {members, exchange (APSS_EXCHANGE_*), head terms, rows, candidates summed over the members, distinct candidates, result pairs,
 * bytes all-gathered per member, bytes all-reduced} of the last call: what an operator logs per batch */
JNIEXPORT jint JNICALL Java_cpslab_gpu_NativeApss_groupStats(JNIEnv *env, jclass cls, jlong g, jlongArray out) {
  (void)cls;
  if ((*env)->GetArrayLength(env, out) < 9) return APSS_E_INVALID;
  apss_group_stats st;
  st.struct_size = (int32_t)sizeof(st);
  const int32_t rc = apss_group_stats_get(G(g), &st);
  if (rc != APSS_OK) return rc;
  const jlong v[9] = {st.n_members, st.exchange, st.head_terms, st.rows, st.candidates_sum, st.union_pairs, st.result_pairs,
                      st.all_gather_bytes, st.all_reduce_bytes};
  (*env)->SetLongArrayRegion(env, out, 0, 9, v);
  return APSS_OK;
}

JNIEXPORT jint JNICALL Java_cpslab_gpu_NativeApss_setHeadTerms(JNIEnv *env, jclass cls, jlong h, jintArray terms, jint part,
                                                               jint nParts) {
  (void)cls;
  const jsize n = (*env)->GetArrayLength(env, terms);
  jint *t = (jint *)malloc(sizeof(jint) * (size_t)(n > 0 ? n : 1));
  if (!t) return APSS_E_NOMEM;
  (*env)->GetIntArrayRegion(env, terms, 0, n, t);
  const int32_t rc = (*env)->ExceptionCheck(env) ? APSS_E_INVALID : apss_set_head_terms(H(h), n, (const int32_t *)t, part, nParts);
  free(t);
  return rc;
}

JNIEXPORT jint JNICALL Java_cpslab_gpu_NativeApss_setHeadFold(JNIEnv *env, jclass cls, jlong h, jint columns) {
  (void)env; (void)cls;
  return apss_set_head_fold(H(h), columns);
}

JNIEXPORT jintArray JNICALL Java_cpslab_gpu_NativeApss_headTerms(JNIEnv *env, jclass cls, jlong h) {
  (void)cls;
  int32_t n = 0;
  if (apss_get_head_terms(H(h), 0, 0, &n) != APSS_OK) return 0;
  int32_t *t = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  if (!t) return 0;
  jintArray out = 0;
  if (apss_get_head_terms(H(h), n, t, &n) == APSS_OK) {
    out = (*env)->NewIntArray(env, n);
    if (out) (*env)->SetIntArrayRegion(env, out, 0, n, (const jint *)t);
  }
  free(t);
  return out;
}
#endif
#endif
