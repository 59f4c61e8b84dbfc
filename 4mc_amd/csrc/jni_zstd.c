/*
 * 4mc_amd/csrc/jni_zstd.c — JNI entry points of ZstdCompressor / ZstdDecompressor (4mz block
 * codec) and of the streaming zstd classes, so that libhadoop-4mc.so keeps the reference's full
 * export set (35 symbols).
 *
 * Block codec (native/jniZstdCompressor.c:60-201, native/jniZstdDecompressor.c:58-122): same field
 * names and error contract as the LZ4 pair; codec results follow zstd's size_t convention
 * (error <=> value > (size_t)-ZSTD_error_maxCode, native/zstd/common/error_private.h).
 * decompressBytesDirect (any level) and compressBytesDirect (zstd level 1) run on the device;
 * compressBytesDirectMC (zstd level 3) and HC(1|3|6) run on the device; other HC levels fail LOUDLY with
 * java/lang/InternalError("ZSTD_compress returned: <error code>") — there is no CPU fallback.
 *
 * Streaming classes (native/jniZstd.c, native/jniZStreamCompressor.c, native/jniZStreamDecompressor.c)
 * are a serial, cross-chunk-window stream with no independent units — out of scope of the block
 * engine; their symbols exist and report "unsupported".
 */
#include <stdio.h>
#include "jni_min.h"
#include "fourmc.h"
#include "fourmc_gpu.h"

jint fourmc_jni_xxhash32(JNIEnv* env, jbyteArray buf, jint off, jint len, jint seed);
void fourmc_jni_throw_internal(JNIEnv* env, const char* msg);

#define ZERR_GENERIC        ((size_t)-1)      /* ZSTD_error_GENERIC = 1            */
#define ZERR_MAXCODE        120               /* ZSTD_error_maxCode                */
static int z_is_error(size_t code) { return code > (size_t)-ZERR_MAXCODE; }

/* ---------------------------------------------------------------- ZstdCompressor */
static jfieldID zc_finish, zc_finished, zc_ubuf, zc_ulen, zc_cbuf, zc_bufsize;

JNIEXPORT void JNICALL
Java_com_fing_compression_fourmc_ZstdCompressor_initIDs(JNIEnv* env, jclass cls)
{
    zc_finish   = (*env)->GetFieldID(env, cls, "finish", "Z");
    zc_finished = (*env)->GetFieldID(env, cls, "finished", "Z");
    zc_ubuf     = (*env)->GetFieldID(env, cls, "uncompressedDirectBuf", "Ljava/nio/ByteBuffer;");
    zc_ulen     = (*env)->GetFieldID(env, cls, "uncompressedDirectBufLen", "I");
    zc_cbuf     = (*env)->GetFieldID(env, cls, "compressedDirectBuf", "Ljava/nio/ByteBuffer;");
    zc_bufsize  = (*env)->GetFieldID(env, cls, "directBufferSize", "I");
}

static jint zstd_compress_common(JNIEnv* env, jobject self, int level)
{
    jobject ubuf = (*env)->GetObjectField(env, self, zc_ubuf);
    jobject cbuf = (*env)->GetObjectField(env, self, zc_cbuf);
    const char* src = (const char*)(*env)->GetDirectBufferAddress(env, ubuf);
    char* dst = (char*)(*env)->GetDirectBufferAddress(env, cbuf);
    unsigned ulen = (unsigned)(*env)->GetIntField(env, self, zc_ulen);
    size_t r;
    if (!src || !dst) return 0;
    /* level 1 runs on the device; other levels come back as an error code (no CPU fallback) and throw below */
    r = fourmc_ZSTD_compress(dst, 1024u * 1024u * 1024u /* enforced in Java, jniZstdCompressor.c:93 */, src, ulen, level);
    if (!z_is_error(r)) (*env)->SetIntField(env, self, zc_ulen, 0);
    else {
        char msg[256];
        snprintf(msg, sizeof msg, "%s returned: %lu", "ZSTD_compress", (unsigned long)r);
        fourmc_jni_throw_internal(env, msg);
    }
    return (jint)r;
}

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_ZstdCompressor_compressBytesDirect(JNIEnv* env, jobject self)
{ return zstd_compress_common(env, self, 1); }

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_ZstdCompressor_compressBytesDirectMC(JNIEnv* env, jobject self)
{ return zstd_compress_common(env, self, 3); }

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_ZstdCompressor_compressBytesDirectHC(JNIEnv* env, jobject self, jint level)
{ return zstd_compress_common(env, self, level); }

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_ZstdCompressor_compressBound(JNIEnv* env, jclass cls, jint n)
{
    /* ZSTD_COMPRESSBOUND (native/zstd/zstd.h): n + n/256 + small-input margin */
    size_t s = (size_t)(unsigned)n;
    (void)env; (void)cls;
    return (jint)(s + (s >> 8) + (s < (128u << 10) ? (((128u << 10) - s) >> 11) : 0));
}

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_ZstdCompressor_xxhash32(JNIEnv* env, jclass cls, jbyteArray buf, jint off, jint len, jint seed)
{ (void)cls; return fourmc_jni_xxhash32(env, buf, off, len, seed); }

/* ---------------------------------------------------------------- ZstdDecompressor */
static jfieldID zd_finished, zd_cbuf, zd_clen, zd_ubuf, zd_bufsize;

JNIEXPORT void JNICALL
Java_com_fing_compression_fourmc_ZstdDecompressor_initIDs(JNIEnv* env, jclass cls)
{
    zd_finished = (*env)->GetFieldID(env, cls, "finished", "Z");
    zd_cbuf     = (*env)->GetFieldID(env, cls, "compressedDirectBuf", "Ljava/nio/Buffer;");
    zd_clen     = (*env)->GetFieldID(env, cls, "compressedDirectBufLen", "I");
    zd_ubuf     = (*env)->GetFieldID(env, cls, "uncompressedDirectBuf", "Ljava/nio/Buffer;");
    zd_bufsize  = (*env)->GetFieldID(env, cls, "directBufferSize", "I");
}

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_ZstdDecompressor_decompressBytesDirect(JNIEnv* env, jobject self)
{
    jobject cbuf = (*env)->GetObjectField(env, self, zd_cbuf);
    unsigned clen = (unsigned)(*env)->GetIntField(env, self, zd_clen);
    jobject ubuf = (*env)->GetObjectField(env, self, zd_ubuf);
    unsigned cap = (unsigned)(*env)->GetIntField(env, self, zd_bufsize);
    char* dst = (char*)(*env)->GetDirectBufferAddress(env, ubuf);
    const char* src = (const char*)(*env)->GetDirectBufferAddress(env, cbuf);
    int r;
    if (!dst || !src) return 0;
    r = (int)fourmc_ZSTD_decompress(dst, cap, src, clen);     /* int truncation as in jniZstdDecompressor.c:73,90 */
    if (r >= 0) (*env)->SetIntField(env, self, zd_clen, 0);
    else {
        char msg[256];
        snprintf(msg, sizeof msg, "LZ4_decompress_safe returned: %d", r);   /* text as in jniZstdDecompressor.c:96 */
        fourmc_jni_throw_internal(env, msg);
    }
    return r;
}

JNIEXPORT jint JNICALL
Java_com_fing_compression_fourmc_ZstdDecompressor_xxhash32(JNIEnv* env, jclass cls, jbyteArray buf, jint off, jint len, jint seed)
{ (void)cls; return fourmc_jni_xxhash32(env, buf, off, len, seed); }

/* ---------------------------------------------------------------- streaming zstd: unsupported */
JNIEXPORT jboolean JNICALL Java_com_fing_compression_fourmc_zstd_Zstd_isError(JNIEnv* env, jclass c, jlong code)
{ (void)env; (void)c; return z_is_error((size_t)code) != 0; }
JNIEXPORT jstring JNICALL Java_com_fing_compression_fourmc_zstd_Zstd_getErrorName(JNIEnv* env, jclass c, jlong code)
{ (void)c; return (*env)->NewStringUTF(env, z_is_error((size_t)code) ? "Unsupported in the MI355X block build (streaming zstd)" : "No error detected"); }
JNIEXPORT jint JNICALL Java_com_fing_compression_fourmc_zstd_Zstd_cStreamInSize(JNIEnv* env, jclass c)  { (void)env; (void)c; return 1 << 17; }
JNIEXPORT jint JNICALL Java_com_fing_compression_fourmc_zstd_Zstd_cStreamOutSize(JNIEnv* env, jclass c) { (void)env; (void)c; return (1 << 17) + ((1 << 17) >> 8) + 3 + 4; }
JNIEXPORT jint JNICALL Java_com_fing_compression_fourmc_zstd_Zstd_dStreamInSize(JNIEnv* env, jclass c)  { (void)env; (void)c; return (1 << 17) + 3; }
JNIEXPORT jint JNICALL Java_com_fing_compression_fourmc_zstd_Zstd_dStreamOutSize(JNIEnv* env, jclass c) { (void)env; (void)c; return 1 << 17; }

static jfieldID zs_src_pos, zs_dst_pos, zds_src_pos, zds_dst_pos;
JNIEXPORT void JNICALL Java_com_fing_compression_fourmc_zstd_ZstdStreamCompressor_initIDs(JNIEnv* env, jclass cls)
{ zs_src_pos = (*env)->GetFieldID(env, cls, "srcPos", "J"); zs_dst_pos = (*env)->GetFieldID(env, cls, "dstPos", "J"); }
/* the streaming ZstCodec (native/jniZStreamCompressor.c:96-134, jniZStreamDecompressor.c:112) is not part of the block path:
 * fail at creation, loudly, instead of handing Java a null stream handle */
static void throw_unsupported(JNIEnv* env)
{
    jclass cls = (*env)->FindClass(env, "java/lang/UnsupportedOperationException");
    if (cls) { (*env)->ThrowNew(env, cls, "streaming zstd (ZstCodec) is not served by the MI355X block build; use the 4mz block codecs"); (*env)->DeleteLocalRef(env, cls); }
}
JNIEXPORT jlong JNICALL Java_com_fing_compression_fourmc_zstd_ZstdStreamCompressor_createCStream(JNIEnv* env, jclass c)
{ (void)c; throw_unsupported(env); return 0; }
JNIEXPORT jint JNICALL Java_com_fing_compression_fourmc_zstd_ZstdStreamCompressor_freeCStream(JNIEnv* env, jclass c, jlong s)
{ (void)env; (void)c; (void)s; return 0; }
JNIEXPORT jint JNICALL Java_com_fing_compression_fourmc_zstd_ZstdStreamCompressor_initCStream(JNIEnv* env, jclass c, jlong s, jint level)
{ (void)env; (void)c; (void)s; (void)level; return (jint)ZERR_GENERIC; }
JNIEXPORT jint JNICALL Java_com_fing_compression_fourmc_zstd_ZstdStreamCompressor_compressStream(JNIEnv* env, jobject self, jlong s, jobject dst, jint dst_size, jobject src, jint src_size)
{ (void)env; (void)self; (void)s; (void)dst; (void)dst_size; (void)src; (void)src_size; return (jint)ZERR_GENERIC; }
JNIEXPORT jint JNICALL Java_com_fing_compression_fourmc_zstd_ZstdStreamCompressor_endStream(JNIEnv* env, jobject self, jlong s, jobject dst, jint dst_off, jint dst_size)
{ (void)env; (void)self; (void)s; (void)dst; (void)dst_off; (void)dst_size; return (jint)ZERR_GENERIC; }

JNIEXPORT void JNICALL Java_com_fing_compression_fourmc_zstd_ZstdStreamDecompressor_initIDs(JNIEnv* env, jclass cls)
{ zds_src_pos = (*env)->GetFieldID(env, cls, "srcPos", "J"); zds_dst_pos = (*env)->GetFieldID(env, cls, "dstPos", "J"); }
JNIEXPORT jlong JNICALL Java_com_fing_compression_fourmc_zstd_ZstdStreamDecompressor_createDStream(JNIEnv* env, jclass c)
{ (void)c; throw_unsupported(env); return 0; }
JNIEXPORT jint JNICALL Java_com_fing_compression_fourmc_zstd_ZstdStreamDecompressor_freeDStream(JNIEnv* env, jclass c, jlong s)
{ (void)env; (void)c; (void)s; return 0; }
JNIEXPORT jint JNICALL Java_com_fing_compression_fourmc_zstd_ZstdStreamDecompressor_initDStream(JNIEnv* env, jclass c, jlong s)
{ (void)env; (void)c; (void)s; return (jint)ZERR_GENERIC; }
JNIEXPORT jint JNICALL Java_com_fing_compression_fourmc_zstd_ZstdStreamDecompressor_decompressStream(JNIEnv* env, jobject self, jlong s, jobject dst, jint dst_size, jobject src, jint src_size)
{ (void)env; (void)self; (void)s; (void)dst; (void)dst_size; (void)src; (void)src_size; return (jint)ZERR_GENERIC; }
