/* jni.h -- COMPILE-CHECK STUB, not the JDK's header and never used to build a shim that runs.
 * The build image has no JDK, so jni/carskit_jni.cpp had never been through a compiler.  This file
 * declares only the JNI types and the JNIEnv member functions that shim uses, with the signatures of the JNI specification
 * (Java SE "JNI Functions" chapter), so that `g++ -fsyntax-only -Itests/jni_stub -Iinclude jni/carskit_jni.cpp`
 * (tests/test_java_binding_text.py) type-checks every statement of the shim against the C ABI.  It proves nothing about
 * behaviour inside a JVM; where a JDK exists the real <jni.h> is used. */
#ifndef CMI_JNI_STUB_H
#define CMI_JNI_STUB_H
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef uint8_t jboolean;
typedef jint jsize;

class _jobject {};
class _jclass : public _jobject {};
class _jthrowable : public _jobject {};
class _jstring : public _jobject {};
class _jarray : public _jobject {};
class _jintArray : public _jarray {};
class _jdoubleArray : public _jarray {};
class _jobjectArray : public _jarray {};
typedef _jobject *jobject;
typedef _jclass *jclass;
typedef _jthrowable *jthrowable;
typedef _jstring *jstring;
typedef _jarray *jarray;
typedef _jintArray *jintArray;
typedef _jdoubleArray *jdoubleArray;
typedef _jobjectArray *jobjectArray;

struct JNIEnv_ {
    jclass FindClass(const char *name);
    jint ThrowNew(jclass clazz, const char *msg);
    jsize GetArrayLength(jarray array);
    void GetIntArrayRegion(jintArray array, jsize start, jsize len, jint *buf);
    void GetDoubleArrayRegion(jdoubleArray array, jsize start, jsize len, jdouble *buf);
    void SetDoubleArrayRegion(jdoubleArray array, jsize start, jsize len, const jdouble *buf);
    jdoubleArray NewDoubleArray(jsize len);
    jobject GetObjectArrayElement(jobjectArray array, jsize index);
    void DeleteLocalRef(jobject obj);
    const char *GetStringUTFChars(jstring str, jboolean *isCopy);
    void ReleaseStringUTFChars(jstring str, const char *chars);
};
typedef JNIEnv_ JNIEnv;
#endif
