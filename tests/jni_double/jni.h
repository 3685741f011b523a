/*
 * jni.h -- TEST DOUBLE of the Java Native Interface header, NOT a JDK file.
 *
 * The build image has no JDK, so bindings/jni/peprml_jni.c (the JNI glue a PEPR maintainer would ship,
 * INTEGRATION.md) could never meet a compiler.  This header declares the JNI types and exactly the
 * entries of the JNIEnv function table that the glue uses, with the signatures the JNI specification
 * gives them ("Java Native Interface Specification", chapter 4: the calling form is
 * (*env)->Function(env, ...)), so that the glue compiles unchanged and can be driven by
 * tests/jni_double/jni_double.c, which implements those entries over plain C objects.
 * A real JVM lays the table out differently (230 slots, these at their specified positions): code compiled
 * against this header must never be loaded into one -- tests/test_jni_double.py builds its own library.
 */
#ifndef PEPRML_TEST_JNI_H
#define PEPRML_TEST_JNI_H
#include <stddef.h>

typedef int jint;
typedef long long jlong;
typedef double jdouble;
typedef unsigned short jchar;
typedef unsigned char jboolean;
typedef jint jsize;

struct jd_object;                          /* opaque to the glue */
typedef struct jd_object *jobject;
typedef jobject jclass, jstring, jarray, jobjectArray, jcharArray, jdoubleArray;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *, const char *);
    jsize (*GetArrayLength)(JNIEnv *, jarray);
    jchar *(*GetCharArrayElements)(JNIEnv *, jcharArray, jboolean *);
    jobject (*GetObjectArrayElement)(JNIEnv *, jobjectArray, jsize);
    const char *(*GetStringUTFChars)(JNIEnv *, jstring, jboolean *);
    jdoubleArray (*NewDoubleArray)(JNIEnv *, jsize);
    jobjectArray (*NewObjectArray)(JNIEnv *, jsize, jclass, jobject);
    jstring (*NewStringUTF)(JNIEnv *, const char *);
    void (*ReleaseCharArrayElements)(JNIEnv *, jcharArray, jchar *, jint);
    void (*ReleaseStringUTFChars)(JNIEnv *, jstring, const char *);
    void (*SetDoubleArrayRegion)(JNIEnv *, jdoubleArray, jsize, jsize, const jdouble *);
    void (*SetObjectArrayElement)(JNIEnv *, jobjectArray, jsize, jobject);
};
#endif
