/*
 * 4mc_amd/csrc/jni_lz4.c — JNI entry points of Lz4Compressor / Lz4Decompressor on the GPU engine.
 *
 * Same exported symbols, Java field names/signatures and error behaviour as the reference
 * (native/jniCompressor.c:56-196, native/jniDecompressor.c:55-120, THROW in native/jnihelper.h:41-48):
 *   - initIDs caches the jfieldIDs of the direct buffers and their lengths;
 *   - compressBytesDirect* reads uncompressedDirectBufLen bytes of the uncompressed direct buffer,
 *     writes the compressed direct buffer, returns the codec result; r > 0 resets
 *     uncompressedDirectBufLen to 0, r <= 0 throws java/lang/InternalError("<fn> returned: <r>")
 *     and STILL returns r; a NULL buffer address returns 0 silently;
 *   - decompressBytesDirect: result >= 0 resets compressedDirectBufLen, < 0 throws.
 * One JNI call is one block (the Java side is synchronous), so each call stages its block through
 * HBM with fourmc_LZ4_* (SURVEY.md §8(b) "Batching constraint").
 */
#include <stdio.h>
#include "jni_min.h"
#include "fourmc.h"
#include "fourmc_gpu.h"

#define MSG_MAX 256

static void throw_internal(JNIEnv* env, const char* msg)
{
    jclass cls = (*env)->FindClass(env, "java/lang/InternalError");
    if (cls) { (*env)->ThrowNew(env, cls, msg); (*env)->DeleteLocalRef(env, cls); }
}

/* ---------------------------------------------------------------- Lz4Compressor */
static jfieldID c_finish, c_finished, c_ubuf, c_ulen, c_cbuf, c_bufsize;

JNIEXPORT void JNICALL
Java_com_fing_compression_fourmc_Lz4Compressor_initIDs(JNIEnv* env, jclass cls)
{
    c_finish   = (*env)->GetFieldID(env, cls, "finish", "Z");
    c_finished = (*env)->GetFieldID(env, cls, "finished", "Z");
    c_ubuf     = (*env)->GetFieldID(env, cls, "uncompressedDirectBuf", "Ljava/nio/ByteBuffer;");
    c_ulen     = (*env)->GetFieldID(env, cls, "uncompressedDirectBufLen", "I");
    c_cbuf     = (*env)->GetFieldID(env, cls, "compressedDirectBuf", "Ljava/nio/ByteBuffer;");
    c_bufsize  = (*env)->GetFieldID(env, cls, "directBufferSize", "I");
}

typedef int (*block_fn)(const char* src, char* dst, int n, int level);

static int enc_fast(const char* src, char* dst, int n, int level)
{ (void)level; return fourmc_LZ4_compress_default(src, dst, n, fourmc_LZ4_compressBound(n)); }   /* LZ4_compress, lz4.c:2661 */

static int enc_hc(const char* src, char* dst, int n, int level)
{ return fourmc_LZ4_compress_HC(src, dst, n, fourmc_LZ4_compressBound(n), level); }   /* LZ4_compressHC2, lz4hc.c:1205 */

static int enc_mc(const char* src, char* dst, int n, int level)
{ (void)level; return fourmc_LZ4_compressMC(src, dst, n); }

static jint compress_common(JNIEnv* env, jobject self, block_fn fn, int level, const char* name)
{
    jobject ubuf = (*env)->GetObjectField(env, self, c_ubuf);
    unsigned ulen = (unsigned)(*env)->GetIntField(env, self, c_ulen);
    jobject cbuf = (*env)->GetObjectField(env, self, c_cbuf);
    const char* src = (const char*)(*env)->GetDirectBufferAddress(env, ubuf);
    char* dst = (char*)(*env)->GetDirectBufferAddress(env, cbuf);
    int r;
    if (!src || !dst) return 0;
    r = fn(src, dst, (int)ulen, level);
    if (r > 0) (*env)->SetIntField(env, self, c_ulen, 0);
    else {
        char msg[MSG_MAX];
        snprintf(msg, sizeof msg, "%s returned: %d", name, r);
        throw_internal(env, msg);
    }
    return (jint)r;
}

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_Lz4Compressor_compressBytesDirect(JNIEnv* env, jobject self)
{ return compress_common(env, self, enc_fast, 0, "LZ4_compress"); }

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_Lz4Compressor_compressBytesDirectMC(JNIEnv* env, jobject self)
{ return compress_common(env, self, enc_mc, 0, "LZ4_compressMC"); }

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_Lz4Compressor_compressBytesDirectHC(JNIEnv* env, jobject self, jint level)
{ return compress_common(env, self, enc_hc, level, "LZ4_compressHC2"); }

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_Lz4Compressor_compressBound(JNIEnv* env, jclass cls, jint n)
{ (void)env; (void)cls; return fourmc_LZ4_compressBound(n); }

static jint xxhash32_common(JNIEnv* env, jbyteArray buf, jint off, jint len, jint seed)
{
    jint h;
    char* in = (char*)(*env)->GetPrimitiveArrayCritical(env, buf, 0);
    if (!in) return 0;
    h = (jint)fourmc_XXH32(in + off, (size_t)len, (unsigned)seed);   /* no JNI calls inside the critical region */
    (*env)->ReleasePrimitiveArrayCritical(env, buf, in, 0);
    return h;
}

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_Lz4Compressor_xxhash32(JNIEnv* env, jclass cls, jbyteArray buf, jint off, jint len, jint seed)
{ (void)cls; return xxhash32_common(env, buf, off, len, seed); }

/* ---------------------------------------------------------------- Lz4Decompressor */
static jfieldID d_finished, d_cbuf, d_clen, d_ubuf, d_bufsize;

JNIEXPORT void JNICALL
Java_com_fing_compression_fourmc_Lz4Decompressor_initIDs(JNIEnv* env, jclass cls)
{
    d_finished = (*env)->GetFieldID(env, cls, "finished", "Z");
    d_cbuf     = (*env)->GetFieldID(env, cls, "compressedDirectBuf", "Ljava/nio/Buffer;");
    d_clen     = (*env)->GetFieldID(env, cls, "compressedDirectBufLen", "I");
    d_ubuf     = (*env)->GetFieldID(env, cls, "uncompressedDirectBuf", "Ljava/nio/Buffer;");
    d_bufsize  = (*env)->GetFieldID(env, cls, "directBufferSize", "I");
}

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_Lz4Decompressor_decompressBytesDirect(JNIEnv* env, jobject self)
{
    jobject cbuf = (*env)->GetObjectField(env, self, d_cbuf);
    unsigned clen = (unsigned)(*env)->GetIntField(env, self, d_clen);
    jobject ubuf = (*env)->GetObjectField(env, self, d_ubuf);
    unsigned cap = (unsigned)(*env)->GetIntField(env, self, d_bufsize);
    char* dst = (char*)(*env)->GetDirectBufferAddress(env, ubuf);
    const char* src = (const char*)(*env)->GetDirectBufferAddress(env, cbuf);
    int r;
    if (!dst || !src) return 0;
    r = fourmc_LZ4_decompress_safe(src, dst, (int)clen, (int)cap);
    if (r >= 0) (*env)->SetIntField(env, self, d_clen, 0);
    else {
        char msg[MSG_MAX];
        snprintf(msg, sizeof msg, "LZ4_decompress_safe returned: %d", r);
        throw_internal(env, msg);
    }
    return r;
}

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_Lz4Decompressor_xxhash32(JNIEnv* env, jclass cls, jbyteArray buf, jint off, jint len, jint seed)
{ (void)cls; return xxhash32_common(env, buf, off, len, seed); }

/* shared with jni_zstd.c */
__attribute__((visibility("hidden"))) jint fourmc_jni_xxhash32(JNIEnv* env, jbyteArray buf, jint off, jint len, jint seed)
{ return xxhash32_common(env, buf, off, len, seed); }
__attribute__((visibility("hidden"))) void fourmc_jni_throw_internal(JNIEnv* env, const char* msg) { throw_internal(env, msg); }
