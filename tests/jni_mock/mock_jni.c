/*
 * tests/jni_mock/mock_jni.c — drives the JNI entry points of libhadoop-4mc.so without a JVM, through
 * a mock JNIEnv that implements exactly the function-table slots the library uses (include/jni_min.h;
 * SURVEY.md §8(b)/(c) did the same against the reference's shipped .so).  TEST INFRASTRUCTURE.
 *
 * usage: mock_jni <libhadoop-4mc.so> <input file> <n bytes> <out dir>
 * Writes <out dir>/<call>.bin for every codec call and prints one "name result thrown" line per call.
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "jni_min.h"

/* a mock object = named int fields + two direct buffers */
typedef struct { const char* name; int is_buf; jint ival; void* buf; jlong lval; } field_t;
typedef struct { field_t f[8]; int nf; } obj_t;
static char g_thrown[512];

static jclass   m_FindClass(JNIEnv* e, const char* n) { (void)e; return (jclass)n; }
static jint     m_ThrowNew(JNIEnv* e, jclass c, const char* msg) { (void)e; snprintf(g_thrown, sizeof g_thrown, "%s: %s", (const char*)c, msg); return 0; }
static void     m_DeleteLocalRef(JNIEnv* e, jobject o) { (void)e; (void)o; }
static jfieldID m_GetFieldID(JNIEnv* e, jclass c, const char* name, const char* sig) { (void)e; (void)c; (void)sig; return (jfieldID)strdup(name); }
static field_t* find(jobject o, jfieldID id)
{
    obj_t* ob = (obj_t*)o;
    for (int i = 0; i < ob->nf; i++) if (!strcmp(ob->f[i].name, (const char*)id)) return &ob->f[i];
    fprintf(stderr, "mock: no field %s\n", (const char*)id); exit(2);
}
static jobject  m_GetObjectField(JNIEnv* e, jobject o, jfieldID id) { (void)e; return (jobject)find(o, id); }
static jint     m_GetIntField(JNIEnv* e, jobject o, jfieldID id) { (void)e; return find(o, id)->ival; }
static void     m_SetIntField(JNIEnv* e, jobject o, jfieldID id, jint v) { (void)e; find(o, id)->ival = v; }
static jlong    m_GetLongField(JNIEnv* e, jobject o, jfieldID id) { (void)e; return find(o, id)->lval; }
static void     m_SetLongField(JNIEnv* e, jobject o, jfieldID id, jlong v) { (void)e; find(o, id)->lval = v; }
static jstring  m_NewStringUTF(JNIEnv* e, const char* t) { (void)e; return (jstring)strdup(t); }
static void*    m_GetDirectBufferAddress(JNIEnv* e, jobject b) { (void)e; return ((field_t*)b)->buf; }
static void*    m_GetCritical(JNIEnv* e, jarray a, jboolean* c) { (void)e; if (c) *c = 0; return a; }
static void     m_ReleaseCritical(JNIEnv* e, jarray a, void* p, jint m) { (void)e; (void)a; (void)p; (void)m; }

typedef void (*init_fn)(JNIEnv*, jclass);
typedef jint (*call0_fn)(JNIEnv*, jobject);
typedef jint (*call1_fn)(JNIEnv*, jobject, jint);
typedef jint (*hash_fn)(JNIEnv*, jclass, jbyteArray, jint, jint, jint);
typedef jint (*bound_fn)(JNIEnv*, jclass, jint);

static void* sym(void* lib, const char* cls, const char* m)
{
    char name[256];
    snprintf(name, sizeof name, "Java_com_fing_compression_fourmc_%s_%s", cls, m);
    void* p = dlsym(lib, name);
    if (!p) { fprintf(stderr, "mock: missing %s\n", name); exit(2); }
    return p;
}

static void dump(const char* dir, const char* name, const void* p, long n)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s.bin", dir, name);
    FILE* f = fopen(path, "wb");
    if (n > 0) fwrite(p, 1, (size_t)n, f);
    fclose(f);
}

int main(int argc, char** argv)
{
    if (argc < 5) return 1;
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "mock: %s\n", dlerror()); return 2; }
    const int n = atoi(argv[3]);
    const char* dir = argv[4];
    const int cap = 4 * 1024 * 1024, ccap = cap + cap / 255 + 16 + 65536;
    char* raw = (char*)calloc(1, (size_t)cap + 64);
    char* comp = (char*)calloc(1, (size_t)ccap);
    char* back = (char*)calloc(1, (size_t)cap + 64);
    FILE* fi = fopen(argv[2], "rb");
    if (!fi || fread(raw, 1, (size_t)n, fi) != (size_t)n) { fprintf(stderr, "mock: cannot read input\n"); return 2; }
    fclose(fi);

    struct JNINativeInterface_ tab;
    memset(&tab, 0, sizeof tab);
    tab.FindClass = m_FindClass; tab.ThrowNew = m_ThrowNew; tab.DeleteLocalRef = m_DeleteLocalRef;
    tab.GetFieldID = m_GetFieldID; tab.GetObjectField = m_GetObjectField; tab.GetIntField = m_GetIntField;
    tab.SetIntField = m_SetIntField; tab.GetDirectBufferAddress = m_GetDirectBufferAddress;
    tab.GetPrimitiveArrayCritical = m_GetCritical; tab.ReleasePrimitiveArrayCritical = m_ReleaseCritical;
    tab.GetLongField = m_GetLongField; tab.SetLongField = m_SetLongField; tab.NewStringUTF = m_NewStringUTF;
    JNIEnv env = &tab;

    if (argc > 6 && !strcmp(argv[5], "zstream")) {
        /* the streaming ZstCodec (ZstdStreamCompressor / ZstdStreamDecompressor, native/jniZStreamCompressor.c, jniZStreamDecompressor.c):
         * the call pattern of ZstdStreamOutputStream / InputStream - input fed `chunk` bytes at a time through a direct buffer, output
         * drained whenever the compressor filled it, endStream until it returns 0; then the stream is read back the same way.
         * argv[6] = chunk; argv[7] (optional) = a .zst file to DECODE instead of compressing (cross-library check). */
        typedef jlong (*create_fn)(JNIEnv*, jclass);
        typedef jint (*free_fn)(JNIEnv*, jclass, jlong);
        typedef jint (*initc_fn)(JNIEnv*, jclass, jlong, jint);
        typedef jint (*initd_fn)(JNIEnv*, jclass, jlong);
        typedef jint (*cs_fn)(JNIEnv*, jobject, jlong, jobject, jint, jobject, jint);
        typedef jint (*end_fn)(JNIEnv*, jobject, jlong, jobject, jint, jint);
        typedef jint (*size_fn)(JNIEnv*, jclass);
        typedef jboolean (*iserr_fn)(JNIEnv*, jclass, jlong);
        typedef jstring (*name_fn)(JNIEnv*, jclass, jlong);
        const int chunk = atoi(argv[6]);
        const char* C = "zstd_ZstdStreamCompressor"; const char* D = "zstd_ZstdStreamDecompressor"; const char* Z = "zstd_Zstd";
        ((init_fn)sym(lib, C, "initIDs"))(&env, (jclass)C);
        ((init_fn)sym(lib, D, "initIDs"))(&env, (jclass)D);
        const int cin = ((size_fn)sym(lib, Z, "cStreamInSize"))(&env, (jclass)Z), cout = ((size_fn)sym(lib, Z, "cStreamOutSize"))(&env, (jclass)Z);
        const int din = ((size_fn)sym(lib, Z, "dStreamInSize"))(&env, (jclass)Z), dout = ((size_fn)sym(lib, Z, "dStreamOutSize"))(&env, (jclass)Z);
        iserr_fn is_err = (iserr_fn)sym(lib, Z, "isError");
        printf("zstream_sizes %d %d %d %d | err(-1)=%d name=%s\n", cin, cout, din, dout, (int)is_err(&env, (jclass)Z, -1),
               (const char*)((name_fn)sym(lib, Z, "getErrorName"))(&env, (jclass)Z, -1));
        char* obuf = (char*)malloc((size_t)cout + (size_t)dout + 64);
        long clen = 0;
        if (argc > 7) {                                     /* a stream written by somebody else */
            FILE* fz = fopen(argv[7], "rb");
            clen = fz ? (long)fread(comp, 1, (size_t)ccap, fz) : -1;
            if (fz) fclose(fz);
            if (clen <= 0) { fprintf(stderr, "mock: cannot read %s\n", argv[7]); return 2; }
        } else {
            g_thrown[0] = 0;
            jlong cs = ((create_fn)sym(lib, C, "createCStream"))(&env, (jclass)C);
            if (!cs || g_thrown[0]) { printf("zstream_create 0 %s\n", g_thrown[0] ? g_thrown : "-"); return 0; }
            jint r = ((initc_fn)sym(lib, C, "initCStream"))(&env, (jclass)C, cs, 3);
            if (is_err(&env, (jclass)Z, r)) { printf("zstream_init %d\n", r); return 0; }
            int bad = 0;
            for (int at = 0; at < n && !bad; at += chunk) {
                const int len = n - at < chunk ? n - at : chunk;
                field_t sb = {"src", 1, 0, raw + at, 0}, db = {"dst", 1, 0, obuf, 0};
                obj_t o = {{{"srcPos", 0, 0, 0, 0}, {"dstPos", 0, 0, 0, 0}, {"oBuffLen", 0, 0, 0, 0}}, 3};
                while (o.f[0].lval < len) {                 /* until the compressor has taken the whole chunk */
                    r = ((cs_fn)sym(lib, C, "compressStream"))(&env, &o, cs, &db, cout, &sb, len);
                    if (is_err(&env, (jclass)Z, r)) { bad = 1; break; }
                    if (clen + o.f[2].ival > ccap) { bad = 1; break; }
                    memcpy(comp + clen, obuf, (size_t)o.f[2].ival); clen += o.f[2].ival;
                }
            }
            for (int guard = 0; !bad && guard < 1000; guard++) {
                field_t db = {"dst", 1, 0, obuf, 0};
                obj_t o = {{{"srcPos", 0, 0, 0, 0}, {"dstPos", 0, 0, 0, 0}, {"oBuffLen", 0, 0, 0, 0}}, 3};
                r = ((end_fn)sym(lib, C, "endStream"))(&env, &o, cs, &db, 0, cout);
                if (is_err(&env, (jclass)Z, r)) { bad = 1; break; }
                memcpy(comp + clen, obuf, (size_t)o.f[2].ival); clen += o.f[2].ival;
                if (r == 0) break;
            }
            ((free_fn)sym(lib, C, "freeCStream"))(&env, (jclass)C, cs);
            printf("zstream_compress %ld - | bad=%d\n", clen, bad);
            dump(dir, "zstream", comp, clen);
            if (bad) return 0;
        }
        {
            g_thrown[0] = 0;
            jlong ds = ((create_fn)sym(lib, D, "createDStream"))(&env, (jclass)D);
            if (!ds || g_thrown[0]) { printf("zstream_createD 0 %s\n", g_thrown[0] ? g_thrown : "-"); return 0; }
            jint r = ((initd_fn)sym(lib, D, "initDStream"))(&env, (jclass)D, ds);
            long total = 0; int bad = is_err(&env, (jclass)Z, r) ? 1 : 0;
            for (long at = 0; at < clen && !bad; at += din) {
                const int len = (int)(clen - at < din ? clen - at : din);
                field_t sb = {"src", 1, 0, comp + at, 0}, db = {"dst", 1, 0, obuf, 0};
                obj_t o = {{{"srcPos", 0, 0, 0, 0}, {"dstPos", 0, 0, 0, 0}, {"oBuffLen", 0, 0, 0, 0}}, 3};
                for (int guard = 0; guard < 100000; guard++) {
                    o.f[1].lval = 0;                        /* the Java side hands over an empty output buffer every call */
                    r = ((cs_fn)sym(lib, D, "decompressStream"))(&env, &o, ds, &db, dout, &sb, len);
                    if (is_err(&env, (jclass)Z, r)) { bad = 1; break; }
                    if (total + o.f[2].ival > cap) { bad = 1; break; }
                    memcpy(back + total, obuf, (size_t)o.f[2].ival); total += o.f[2].ival;
                    if (o.f[0].lval >= len && o.f[2].ival < dout) break;      /* input taken, output not full: nothing more for now */
                }
            }
            ((free_fn)sym(lib, D, "freeDStream"))(&env, (jclass)D, ds);
            printf("zstream_roundtrip %ld - | bad=%d same=%d\n", total, bad, total == n && !memcmp(back, raw, (size_t)n));
        }
        return 0;
    }

    const char* codecs[2] = {"Lz4", "Zstd"};
    if (argc > 5) {
        /* BlockCompressorStream's call pattern (Lz4Codec.java:95-104 / ZstdCodec, the raw block codecs): a stream of small
         * buffers - `chunk` bytes per compressBytesDirect call on buffers of directBufferSize = chunk + overhead, far from
         * 4 MiB -, each framed with its length by the caller; here the compressed chunks are written back to back with
         * their sizes so that the test can compare every one with the oracle, then each is decompressed again. */
        const int chunk = atoi(argv[5]);
        for (int c = 0; c < 2; c++) {
            char ccls[32], dcls[32], name[96];
            snprintf(ccls, sizeof ccls, "%sCompressor", codecs[c]);
            snprintf(dcls, sizeof dcls, "%sDecompressor", codecs[c]);
            ((init_fn)sym(lib, ccls, "initIDs"))(&env, (jclass)ccls);
            ((init_fn)sym(lib, dcls, "initIDs"))(&env, (jclass)dcls);
            const int dbs = chunk + chunk / 6 + 32 + 1024;                    /* "bufferSize + compressionOverhead" of the Java side */
            snprintf(name, sizeof name, "%s/%s_stream.bin", dir, codecs[c]);
            FILE* fo = fopen(name, "wb");
            snprintf(name, sizeof name, "%s/%s_stream.sizes", dir, codecs[c]);
            FILE* fs = fopen(name, "w");
            int calls = 0, bad = 0, thrown = 0;
            for (int at = 0; at < n; at += chunk, calls++) {
                const int len = n - at < chunk ? n - at : chunk;
                obj_t co = {{{"finish", 0, 0, 0}, {"finished", 0, 0, 0}, {"uncompressedDirectBuf", 1, 0, raw + at}, {"uncompressedDirectBufLen", 0, len, 0},
                             {"compressedDirectBuf", 1, 0, comp}, {"directBufferSize", 0, dbs, 0}}, 6};
                g_thrown[0] = 0;
                jint r = ((call0_fn)sym(lib, ccls, "compressBytesDirect"))(&env, &co);
                if (g_thrown[0] || r <= 0 || co.f[3].ival != 0) { thrown++; continue; }
                fwrite(comp, 1, (size_t)r, fo); fprintf(fs, "%d\n", r);
                obj_t dob = {{{"finished", 0, 0, 0}, {"compressedDirectBuf", 1, 0, comp}, {"compressedDirectBufLen", 0, r, 0},
                              {"uncompressedDirectBuf", 1, 0, back}, {"directBufferSize", 0, dbs, 0}}, 5};
                g_thrown[0] = 0;
                jint d = ((call0_fn)sym(lib, dcls, "decompressBytesDirect"))(&env, &dob);
                if (g_thrown[0] || d != len || memcmp(back, raw + at, (size_t)len) || dob.f[2].ival != 0) bad++;
            }
            fclose(fo); fclose(fs);
            printf("%s_stream %d - | bad=%d thrown=%d\n", codecs[c], calls, bad, thrown);
        }
        return 0;
    }
    for (int c = 0; c < 2; c++) {
        char ccls[32], dcls[32], tag[64];
        snprintf(ccls, sizeof ccls, "%sCompressor", codecs[c]);
        snprintf(dcls, sizeof dcls, "%sDecompressor", codecs[c]);
        ((init_fn)sym(lib, ccls, "initIDs"))(&env, (jclass)ccls);
        ((init_fn)sym(lib, dcls, "initIDs"))(&env, (jclass)dcls);
        printf("%s_compressBound %d -\n", codecs[c], ((bound_fn)sym(lib, ccls, "compressBound"))(&env, (jclass)ccls, n));
        printf("%s_xxhash32 %d -\n", codecs[c], ((hash_fn)sym(lib, ccls, "xxhash32"))(&env, (jclass)ccls, (jbyteArray)raw, 3, n > 103 ? 100 : 0, 0));
        const char* calls[3] = {"compressBytesDirect", "compressBytesDirectMC", "compressBytesDirectHC"};
        for (int k = 0; k < 3; k++) {
            obj_t co = {{{"finish", 0, 0, 0}, {"finished", 0, 0, 0}, {"uncompressedDirectBuf", 1, 0, raw}, {"uncompressedDirectBufLen", 0, n, 0},
                         {"compressedDirectBuf", 1, 0, comp}, {"directBufferSize", 0, cap, 0}}, 6};
            g_thrown[0] = 0;
            memset(comp, 0, (size_t)ccap);
            jint r = k < 2 ? ((call0_fn)sym(lib, ccls, calls[k]))(&env, &co) : ((call1_fn)sym(lib, ccls, calls[k]))(&env, &co, c == 0 ? 4 : 1);
            snprintf(tag, sizeof tag, "%s_%s", codecs[c], calls[k]);
            printf("%s %d %s | ulen_after=%d\n", tag, r, g_thrown[0] ? g_thrown : "-", co.f[3].ival);
            if (r > 0 && r < ccap && !g_thrown[0]) {
                dump(dir, tag, comp, r);
                obj_t dob = {{{"finished", 0, 0, 0}, {"compressedDirectBuf", 1, 0, comp}, {"compressedDirectBufLen", 0, r, 0},
                              {"uncompressedDirectBuf", 1, 0, back}, {"directBufferSize", 0, cap, 0}}, 5};
                g_thrown[0] = 0;
                memset(back, 0, (size_t)cap);
                jint d = ((call0_fn)sym(lib, dcls, "decompressBytesDirect"))(&env, &dob);
                printf("%s_roundtrip %d %s | same=%d clen_after=%d\n", tag, d, g_thrown[0] ? g_thrown : "-", d == n && !memcmp(back, raw, (size_t)n), dob.f[2].ival);
            }
        }
        if (c == 1) {   /* levels the shipped Java classes never pass: 9 (every library serves it) and 15 (the reference serves it; the device
                         * library has levels 1..12 and answers with an error code + InternalError, length field untouched) */
            const int lv[2] = {9, 15};
            for (int k = 0; k < 2; k++) {
                obj_t co = {{{"finish", 0, 0, 0}, {"finished", 0, 0, 0}, {"uncompressedDirectBuf", 1, 0, raw}, {"uncompressedDirectBufLen", 0, n, 0},
                             {"compressedDirectBuf", 1, 0, comp}, {"directBufferSize", 0, cap, 0}}, 6};
                g_thrown[0] = 0;
                jint r = ((call1_fn)sym(lib, ccls, "compressBytesDirectHC"))(&env, &co, lv[k]);
                printf("Zstd_compressBytesDirectHC_level%d %d %s | ulen_after=%d\n", lv[k], r, g_thrown[0] ? g_thrown : "-", co.f[3].ival);
            }
        }
        {   /* corrupt input: the decompressor throws InternalError and returns the codec's error */
            obj_t dob = {{{"finished", 0, 0, 0}, {"compressedDirectBuf", 1, 0, raw}, {"compressedDirectBufLen", 0, n > 1000 ? 1000 : n, 0},
                          {"uncompressedDirectBuf", 1, 0, back}, {"directBufferSize", 0, cap, 0}}, 5};
            g_thrown[0] = 0;
            jint d = ((call0_fn)sym(lib, dcls, "decompressBytesDirect"))(&env, &dob);
            printf("%s_decompress_garbage %d %s\n", codecs[c], d, g_thrown[0] ? g_thrown : "-");
        }
    }
    return 0;
}
