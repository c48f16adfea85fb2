/* NOT the JDK's jni.h.  A minimal declaration of the JNI types, constants and JNIEnv / JavaVM members that
 * bindings/jni/skeres_amd_jni.c uses, so that tests/test_bindings_cpu.py can compile that file FOR SYNTAX ONLY in an image
 * without a JDK (gcc -fsyntax-only).  Nothing is linked or run against it and it pins no behaviour; signatures follow the
 * JNI specification (Java SE "JNI Functions" chapter). */
#ifndef SKERES_TEST_JNI_STUB_H
#define SKERES_TEST_JNI_STUB_H
#include <stdarg.h>
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_OK 0
#define JNI_ABORT 2
#define JNI_VERSION_1_6 0x00010006

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef double jdouble;
typedef jint jsize;
struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;
typedef jarray jbyteArray;
struct _jmethodID;
typedef struct _jmethodID* jmethodID;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNIInvokeInterface_;
typedef const struct JNIInvokeInterface_* JavaVM;

struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv*, const char*);
  jint (*ThrowNew)(JNIEnv*, jclass, const char*);
  jboolean (*ExceptionCheck)(JNIEnv*);
  jobject (*NewGlobalRef)(JNIEnv*, jobject);
  void (*DeleteGlobalRef)(JNIEnv*, jobject);
  jclass (*GetObjectClass)(JNIEnv*, jobject);
  jmethodID (*GetMethodID)(JNIEnv*, jclass, const char*, const char*);
  jboolean (*CallBooleanMethod)(JNIEnv*, jobject, jmethodID, ...);
  jstring (*NewStringUTF)(JNIEnv*, const char*);
  const char* (*GetStringUTFChars)(JNIEnv*, jstring, jboolean*);
  void (*ReleaseStringUTFChars)(JNIEnv*, jstring, const char*);
  jsize (*GetArrayLength)(JNIEnv*, jarray);
  jint* (*GetIntArrayElements)(JNIEnv*, jintArray, jboolean*);
  void (*ReleaseIntArrayElements)(JNIEnv*, jintArray, jint*, jint);
  jlong* (*GetLongArrayElements)(JNIEnv*, jlongArray, jboolean*);
  void (*ReleaseLongArrayElements)(JNIEnv*, jlongArray, jlong*, jint);
  jdouble* (*GetDoubleArrayElements)(JNIEnv*, jdoubleArray, jboolean*);
  void (*ReleaseDoubleArrayElements)(JNIEnv*, jdoubleArray, jdouble*, jint);
  void (*GetDoubleArrayRegion)(JNIEnv*, jdoubleArray, jsize, jsize, jdouble*);
  void (*SetDoubleArrayRegion)(JNIEnv*, jdoubleArray, jsize, jsize, const jdouble*);
  jbyteArray (*NewByteArray)(JNIEnv*, jsize);
  void (*GetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, jbyte*);
  void (*SetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, const jbyte*);
};
struct JNIInvokeInterface_ {
  jint (*GetEnv)(JavaVM*, void**, jint);
  jint (*AttachCurrentThread)(JavaVM*, void**, void*);
  jint (*DetachCurrentThread)(JavaVM*);
};
#endif
