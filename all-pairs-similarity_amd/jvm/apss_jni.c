/*
 * apss_jni.c -- JNI shim: forwards cpslab.gpu.NativeApss 1:1 to the C ABI of include/apss.h.
 *
 * NOT compiled in this repository's image (no JDK / jni.h here); a maintainer builds it next to the reference with
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *       all-pairs-similarity_amd/jvm/apss_jni.c -Lall-pairs-similarity_amd/csrc -lapss_hip -o libapss_jni.so
 * The shim holds no state and does no arithmetic: array pinning, the call, the status code.
 */
#if defined(__has_include)
#if __has_include(<jni.h>)
#include <jni.h>
#include <stdint.h>

#include "apss.h"

#define H(x) ((apss_handle *)(intptr_t)(x))

JNIEXPORT jlong JNICALL Java_cpslab_gpu_NativeApss_create(JNIEnv *env, jclass cls, jint dim, jdouble theta,
                                                          jdouble indexThreshold, jint flags, jint device) {
  (void)env; (void)cls;
  apss_config c = {0};
  c.struct_size = (int32_t)sizeof(c);
  c.dim = dim;
  c.theta = theta;
  c.index_threshold = indexThreshold;
  c.flags = (uint32_t)flags;
  c.device_id = device;
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

/* mode 0 = insert, 1 = query (frozen index), 2 = insertAndQuery; returns the number of result triples or a negative status */
JNIEXPORT jlong JNICALL Java_cpslab_gpu_NativeApss_submit(JNIEnv *env, jclass cls, jlong h, jint mode, jlongArray rowptr,
                                                          jintArray indices, jdoubleArray values, jlongArray ids) {
  (void)cls;
  const jsize n = (*env)->GetArrayLength(env, ids);
  jlong *rp = (*env)->GetPrimitiveArrayCritical(env, rowptr, 0);
  jint *ix = (*env)->GetPrimitiveArrayCritical(env, indices, 0);
  jdouble *vl = (*env)->GetPrimitiveArrayCritical(env, values, 0);
  jlong *id = (*env)->GetPrimitiveArrayCritical(env, ids, 0);
  int64_t n_res = 0;
  int32_t rc;
  if (mode == 0) rc = apss_insert(H(h), n, (const int64_t *)rp, (const int32_t *)ix, vl, (const int64_t *)id);
  else if (mode == 1) rc = apss_query(H(h), n, (const int64_t *)rp, (const int32_t *)ix, vl, (const int64_t *)id, &n_res);
  else rc = apss_insert_and_query(H(h), n, (const int64_t *)rp, (const int32_t *)ix, vl, (const int64_t *)id, &n_res);
  (*env)->ReleasePrimitiveArrayCritical(env, ids, id, JNI_ABORT);
  (*env)->ReleasePrimitiveArrayCritical(env, values, vl, JNI_ABORT);
  (*env)->ReleasePrimitiveArrayCritical(env, indices, ix, JNI_ABORT);
  (*env)->ReleasePrimitiveArrayCritical(env, rowptr, rp, JNI_ABORT);
  return rc == APSS_OK ? (jlong)n_res : (jlong)rc;
}

JNIEXPORT jint JNICALL Java_cpslab_gpu_NativeApss_fetch(JNIEnv *env, jclass cls, jlong h, jlong count, jlongArray outQ,
                                                        jlongArray outC, jfloatArray outScore) {
  (void)cls;
  jlong *q = (*env)->GetPrimitiveArrayCritical(env, outQ, 0);
  jlong *c = (*env)->GetPrimitiveArrayCritical(env, outC, 0);
  jfloat *s = (*env)->GetPrimitiveArrayCritical(env, outScore, 0);
  const int32_t rc = apss_fetch_results(H(h), 0, count, (int64_t *)q, (int64_t *)c, s);
  (*env)->ReleasePrimitiveArrayCritical(env, outScore, s, 0);
  (*env)->ReleasePrimitiveArrayCritical(env, outC, c, 0);
  (*env)->ReleasePrimitiveArrayCritical(env, outQ, q, 0);
  return rc;
}
#endif
#endif
