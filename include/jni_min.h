/*
 * include/jni_min.h — the subset of the Java Native Interface this library needs.
 *
 * No JDK (and so no <jni.h>) exists in the build image.  The JNI function table is a public,
 * versioned ABI (JNI specification, "Interface Function Table"): a JNIEnv is a pointer to a
 * pointer to a table of function pointers whose slot numbers are fixed.  Only the slots the four
 * 4mc JNI files use are named here, at their specified indices (6, 14, 23, 94,
 * 95, 100, 101, 109, 110, 167, 222, 223, 230 — the same slots the shipped reference
 * libhadoop-4mc.so calls); all others are padding.  When a real <jni.h> is available, build with
 * -DFOURMC_USE_SYSTEM_JNI to use it instead: the entry points compile unchanged.
 */
#ifndef FOURMC_JNI_MIN_H
#define FOURMC_JNI_MIN_H

#ifdef FOURMC_USE_SYSTEM_JNI
#include <jni.h>
#else
#include <stdint.h>

typedef int32_t  jint;
typedef int64_t  jlong;
typedef int8_t   jbyte;
typedef uint8_t  jboolean;
typedef void*    jobject;
typedef jobject  jclass;
typedef jobject  jstring;
typedef jobject  jarray;
typedef jarray   jbyteArray;
typedef struct fourmc_jfieldID_* jfieldID;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;

struct JNINativeInterface_ {
    void* slot_0_5[6];
    jclass   (*FindClass)(JNIEnv*, const char*);                                   /*   6 */
    void* slot_7_13[7];
    jint     (*ThrowNew)(JNIEnv*, jclass, const char*);                            /*  14 */
    void* slot_15_22[8];
    void     (*DeleteLocalRef)(JNIEnv*, jobject);                                  /*  23 */
    void* slot_24_93[70];
    jfieldID (*GetFieldID)(JNIEnv*, jclass, const char*, const char*);             /*  94 */
    jobject  (*GetObjectField)(JNIEnv*, jobject, jfieldID);                        /*  95 */
    void* slot_96_99[4];
    jint     (*GetIntField)(JNIEnv*, jobject, jfieldID);                           /* 100 */
    jlong    (*GetLongField)(JNIEnv*, jobject, jfieldID);                          /* 101 */
    void* slot_102_108[7];
    void     (*SetIntField)(JNIEnv*, jobject, jfieldID, jint);                     /* 109 */
    void     (*SetLongField)(JNIEnv*, jobject, jfieldID, jlong);                   /* 110 */
    void* slot_111_166[56];
    jstring  (*NewStringUTF)(JNIEnv*, const char*);                                /* 167 */
    void* slot_168_221[54];
    void*    (*GetPrimitiveArrayCritical)(JNIEnv*, jarray, jboolean*);             /* 222 */
    void     (*ReleasePrimitiveArrayCritical)(JNIEnv*, jarray, void*, jint);       /* 223 */
    void* slot_224_229[6];
    void*    (*GetDirectBufferAddress)(JNIEnv*, jobject);                          /* 230 */
    void* slot_231_234[4];
};

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#endif /* FOURMC_USE_SYSTEM_JNI */
#endif
