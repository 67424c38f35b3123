/* Minimal stand-in for <jni.h>: ONLY what all-pairs-similarity_amd/jvm/apss_jni.c uses, so that the shim can be
 * syntax- and type-checked in an image without a JDK (tests/test_jni_shim_syntax.py: gcc -fsyntax-only).  Test
 * infrastructure; never shipped, never linked.  The declarations follow the JNI specification (jni.h of any JDK >= 8). */
#ifndef APSS_TEST_STUB_JNI_H
#define APSS_TEST_STUB_JNI_H
#include <stdint.h>
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_OK 0
typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;
typedef unsigned char jboolean;
struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass, jstring, jarray, jlongArray, jintArray, jdoubleArray, jfloatArray, jthrowable;
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
  jsize (*GetArrayLength)(JNIEnv *, jarray);
  jstring (*NewStringUTF)(JNIEnv *, const char *);
  void (*GetLongArrayRegion)(JNIEnv *, jlongArray, jsize, jsize, jlong *);
  void (*GetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, jint *);
  void (*GetDoubleArrayRegion)(JNIEnv *, jdoubleArray, jsize, jsize, jdouble *);
  void (*SetLongArrayRegion)(JNIEnv *, jlongArray, jsize, jsize, const jlong *);
  void (*SetFloatArrayRegion)(JNIEnv *, jfloatArray, jsize, jsize, const jfloat *);
  jboolean (*ExceptionCheck)(JNIEnv *);
  jintArray (*NewIntArray)(JNIEnv *, jsize);
  void (*SetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, const jint *);
};
#endif
