/*
 * tools/corpus.c — deterministic synthetic corpus ("S-mix") standing in for silesia.
 *
 * silesia is not available offline; this generator produces 4 MiB blocks whose
 * classes imitate silesia's members (English-like text, XML, logs, source code, x86-like binary,
 * database records, 16-bit PCM-like samples, near-random image data, pure random) so that the
 * LZ4-fast ratio of the mix lands near silesia's (~2.1) and stored blocks are exercised.
 *
 * Block b of corpus `seed` depends only on (seed, b): any rank can generate any block range
 * (multi-GPU sharding) and the whole corpus is reproducible from two integers.
 * Neither product nor oracle: used by tests/ and bench.py only.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define BLOCK (4u << 20)

typedef struct { uint64_t s; } rng_t;
static uint64_t sm64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t rnd(rng_t* r) { return sm64(&r->s); }
static uint32_t rnd_n(rng_t* r, uint32_t n) { return (uint32_t)((rnd(r) >> 32) * (uint64_t)n >> 32); }
/* skewed index in [0,n): cube of a uniform variate ~ Zipf-ish head */
static uint32_t rnd_skew(rng_t* r, uint32_t n) {
    uint64_t u = rnd(r) >> 43;                /* 21 bits */
    uint64_t c = (u * u >> 21) * u >> 21;     /* u^3 scaled back to 21 bits */
    return (uint32_t)(c * n >> 21);
}

/* ---- vocabulary: 4096 pseudo-words built from syllables, fixed for all seeds ------------- */
#define VOCAB 4096
static char vocab[VOCAB][12];
static uint8_t vocab_len[VOCAB];
static int vocab_ready;
static void vocab_init(void) {
    static const char* syl[] = {"th","e","an","in","er","on","re","at","en","ou","it","is","or","ti",
        "as","te","et","ng","of","al","de","se","le","sa","si","ar","ve","ra","ld","ur","co","me","pa",
        "ro","li","ch","ma","st","wh","be","fo","gr","pl","tr","sh","qu","ck","ly","ed","ion"};
    rng_t r = { 0x4D43766F636162ull };
    int i, k;
    if (vocab_ready) return;
    for (i = 0; i < VOCAB; i++) {
        int ns = 1 + (i > 20) + (int)rnd_n(&r, 3), len = 0;
        for (k = 0; k < ns && len < 9; k++) {
            const char* s = syl[rnd_n(&r, sizeof syl / sizeof *syl)];
            size_t sl = strlen(s);
            memcpy(vocab[i] + len, s, sl); len += (int)sl;
        }
        vocab_len[i] = (uint8_t)len;
    }
    vocab_ready = 1;
}

typedef struct { uint8_t* p; uint8_t* end; } out_t;
static int put(out_t* o, const void* s, size_t n) {
    size_t room = (size_t)(o->end - o->p);
    if (n > room) n = room;
    memcpy(o->p, s, n); o->p += n;
    return o->p < o->end;
}
static int puts_(out_t* o, const char* s) { return put(o, s, strlen(s)); }
static int putword(out_t* o, rng_t* r) { uint32_t w = rnd_skew(r, VOCAB); return put(o, vocab[w], vocab_len[w]); }
static int putnum(out_t* o, uint64_t v) { char b[24]; int n = snprintf(b, sizeof b, "%llu", (unsigned long long)v); return put(o, b, (size_t)n); }

static void gen_text(out_t* o, rng_t* r, int dict_style) {
    while (o->p < o->end) {
        int words = 4 + (int)rnd_n(r, 14), i;
        if (dict_style) { putword(o, r); puts_(o, " (n.) [Etym: "); putword(o, r); puts_(o, "] Defn: "); }
        for (i = 0; i < words; i++) {
            uint32_t w = rnd_skew(r, dict_style ? 1024 : 2048);
            put(o, vocab[w], vocab_len[w]);
            if (i + 1 < words) puts_(o, rnd_n(r, 12) == 0 ? ", " : " ");
        }
        puts_(o, rnd_n(r, 5) == 0 ? ".\n" : ". ");
    }
}
static void gen_xml(out_t* o, rng_t* r) {
    uint64_t id = rnd(r) % 100000;
    while (o->p < o->end) {
        int i, nw = 1 + (int)rnd_n(r, 4);
        puts_(o, "  <record id=\""); putnum(o, id++); puts_(o, "\" type=\""); putword(o, r);
        puts_(o, "\">\n    <name>");
        for (i = 0; i < nw; i++) { putword(o, r); if (i + 1 < nw) puts_(o, " "); }
        puts_(o, "</name>\n    <value unit=\"ms\">"); putnum(o, rnd_n(r, 100000)); puts_(o, "."); putnum(o, rnd_n(r, 100));
        puts_(o, "</value>\n    <flags>"); putnum(o, rnd_n(r, 16)); puts_(o, "</flags>\n  </record>\n");
    }
}
static void gen_log(out_t* o, rng_t* r) {
    uint64_t t = 1700000000ull + rnd(r) % 1000000;
    static const char* verbs[] = {"GET", "GET", "GET", "POST", "PUT", "HEAD"};
    static const char* codes[] = {"200", "200", "200", "200", "304", "404", "500", "302"};
    while (o->p < o->end) {
        t += rnd_n(r, 3);
        putnum(o, t); puts_(o, " 10."); putnum(o, rnd_n(r, 4)); puts_(o, "."); putnum(o, rnd_n(r, 256)); puts_(o, ".");
        putnum(o, rnd_n(r, 256)); puts_(o, " "); puts_(o, verbs[rnd_n(r, 6)]); puts_(o, " /");
        putword(o, r); puts_(o, "/"); putword(o, r); puts_(o, rnd_n(r, 3) ? ".html" : "?id="); putnum(o, rnd_n(r, 5000));
        puts_(o, " HTTP/1.1 "); puts_(o, codes[rnd_n(r, 8)]); puts_(o, " "); putnum(o, rnd_skew(r, 200000)); puts_(o, "\n");
    }
}
static void gen_code(out_t* o, rng_t* r) {
    static const char* kw[] = {"if (", "for (", "while (", "return ", "static int ", "const char* ", "struct ", "else {", "break;", "#include <"};
    int depth = 0;
    while (o->p < o->end) {
        int i, k = (int)rnd_n(r, 10);
        for (i = 0; i < depth; i++) puts_(o, "    ");
        puts_(o, kw[k]); putword(o, r);
        if (k < 3) { puts_(o, rnd_n(r, 2) ? " == " : " < "); putword(o, r); puts_(o, ") {\n"); if (depth < 6) depth++; }
        else if (k == 7) { puts_(o, "\n"); }
        else if (k == 9) { puts_(o, ".h>\n"); }
        else { puts_(o, rnd_n(r, 2) ? " = " : "_"); putword(o, r); puts_(o, ";\n"); }
        if (depth && rnd_n(r, 4) == 0) { depth--; for (i = 0; i < depth; i++) puts_(o, "    "); puts_(o, "}\n"); }
    }
}
/* x86-like: opcode stream from a skewed table, short operands, and motif reuse */
static void gen_binary(out_t* o, rng_t* r, uint8_t* base) {
    uint8_t ops[64]; int i;
    for (i = 0; i < 64; i++) ops[i] = (uint8_t)rnd(r);
    while (o->p < o->end) {
        uint32_t sel = rnd_n(r, 100);
        size_t have = (size_t)(o->p - base);
        if (sel < 22 && have > 4096) {                      /* repeat an earlier motif (call tables, padding) */
            size_t len = 8 + rnd_skew(r, 120), back = 16 + rnd_n(r, (uint32_t)(have > 60000 ? 60000 : have - 16));
            uint8_t* s = o->p - back;
            for (; len && o->p < o->end; len--) *o->p++ = *s++;
        } else if (sel < 30) {
            size_t len = 4 + rnd_n(r, 28); uint8_t z = rnd_n(r, 3) ? 0 : 0xCC;
            for (; len && o->p < o->end; len--) *o->p++ = z;
        } else {
            uint8_t ins[8]; int n = 1 + (int)rnd_n(r, 6);
            ins[0] = ops[rnd_skew(r, 64)]; ins[1] = ops[rnd_skew(r, 64)];
            for (i = 2; i < n; i++) ins[i] = (i >= 4 && rnd_n(r, 2)) ? 0 : (uint8_t)rnd(r);
            put(o, ins, (size_t)n);
        }
    }
}
/* database: fixed 64-byte records, ascending key, low-cardinality columns */
static void gen_db(out_t* o, rng_t* r) {
    uint64_t key = rnd(r) % 1000000;
    while (o->p < o->end) {
        uint8_t rec[64]; uint32_t w = rnd_skew(r, 256);
        memset(rec, ' ', sizeof rec);
        key += 1 + rnd_n(r, 3);
        memcpy(rec, &key, 8);
        rec[8] = (uint8_t)rnd_n(r, 4); rec[9] = (uint8_t)rnd_n(r, 7);
        memcpy(rec + 12, vocab[w], vocab_len[w]);
        { uint32_t v = rnd_n(r, 1000000); memcpy(rec + 28, &v, 4); }
        { uint64_t t = 1700000000ull + key / 7; memcpy(rec + 32, &t, 8); }
        memcpy(rec + 44, vocab[rnd_skew(r, 64)], vocab_len[rnd_skew(r, 64)] & 7);
        put(o, rec, sizeof rec);
    }
}
/* SDF-like chemical table (nci): fixed-column numeric lines dominated by repeated fields */
static void gen_sdf(out_t* o, rng_t* r) {
    static const char* el[] = {"C", "C", "C", "C", "H", "H", "H", "N", "O", "O", "S", "Cl"};
    uint64_t id = rnd(r) % 100000;
    while (o->p < o->end) {
        int atoms = 8 + (int)rnd_n(r, 24), i; char b[96];
        putnum(o, id++); puts_(o, "\n  -OEChem-0102030405062D\n\n");
        snprintf(b, sizeof b, "%3d%3d  0     0  0  0  0  0  0999 V2000\n", atoms, atoms - 1); puts_(o, b);
        for (i = 0; i < atoms; i++) {
            snprintf(b, sizeof b, "%10.4f%10.4f    0.0000 %-3s 0  0  0  0  0  0  0  0  0  0  0  0\n",
                     (double)rnd_n(r, 200) / 10.0 - 5.0, (double)rnd_n(r, 160) / 10.0 - 4.0, el[rnd_n(r, 12)]);
            puts_(o, b);
        }
        for (i = 1; i < atoms; i++) { snprintf(b, sizeof b, "%3d%3d  %d  0  0  0  0\n", 1 + (int)rnd_n(r, (uint32_t)i), i + 1, 1 + (int)(rnd_n(r, 8) == 0)); puts_(o, b); }
        puts_(o, "M  END\n> <NSC>\n"); putnum(o, id); puts_(o, "\n\n$$$$\n");
    }
}
/* 16-bit little-endian random walk (audio / MR-like): high bytes repeat, low bytes noisy;
 * `background` inserts runs of zero samples (the black surround of a medical image slice) */
static void gen_pcm(out_t* o, rng_t* r, int noise_bits, int background) {
    int32_t v = 0;
    while (o->p + 1 < o->end) {
        int32_t d;
        if (background && rnd_n(r, 160) == 0) {
            size_t z = 2 * (64 + rnd_n(r, 384));
            for (; z && o->p < o->end; z--) *o->p++ = 0;
            v = 0; continue;
        }
        d = (int32_t)(rnd(r) & ((1u << noise_bits) - 1)) - (1 << (noise_bits - 1));
        v += d; if (v > 30000) v = 30000; if (v < -30000) v = -30000;
        *o->p++ = (uint8_t)v; *o->p++ = (uint8_t)((uint32_t)v >> 8);
    }
    if (o->p < o->end) *o->p++ = 0;
}
static void gen_random(out_t* o, rng_t* r) {
    while (o->p + 8 <= o->end) { uint64_t v = rnd(r); memcpy(o->p, &v, 8); o->p += 8; }
    while (o->p < o->end) *o->p++ = (uint8_t)rnd(r);
}

/* class of block b: a 12-entry cycle imitating silesia's 12 members */
static const uint8_t kCycle[12] = { 0 /*dickens*/, 4 /*mozilla*/, 6 /*mr*/, 2 /*nci: sdf*/, 4 /*ooffice*/, 5 /*osdb*/,
                                    0 /*reymont*/, 3 /*samba*/, 7 /*sao*/, 0 /*webster*/, 1 /*xml*/, 8 /*x-ray*/ };

/* Fill dst[0..len) with the bytes of corpus `seed` starting at absolute block `first_block`.
 * len need not be a multiple of 4 MiB (the last block is simply cut short). */
void corpus_fill(uint8_t* dst, size_t len, uint64_t seed, uint64_t first_block)
{
    size_t off = 0; uint64_t b = first_block;
    vocab_init();
    while (off < len) {
        size_t n = len - off < BLOCK ? len - off : BLOCK;
        uint64_t s = seed * 0x9E3779B97F4A7C15ull + b;
        rng_t r; out_t o;
        r.s = sm64(&s);
        o.p = dst + off; o.end = dst + off + n;
        switch (kCycle[b % 12]) {
            case 0: gen_text(&o, &r, (b % 12) == 9); break;
            case 1: gen_xml(&o, &r); break;
            case 2: gen_sdf(&o, &r); break;
            case 3: gen_code(&o, &r); break;
            case 4: gen_binary(&o, &r, dst + off); break;
            case 5: gen_db(&o, &r); break;
            case 6: gen_pcm(&o, &r, 6, 1); break;
            case 7: gen_pcm(&o, &r, 11, 0); break;
            default: gen_random(&o, &r); break;
        }
        off += n; b++;
    }
}

/* templated log corpus for BASELINE config 5 (64 GiB "synthetic log corpus") */
void corpus_fill_logs(uint8_t* dst, size_t len, uint64_t seed, uint64_t first_block)
{
    size_t off = 0; uint64_t b = first_block;
    vocab_init();
    while (off < len) {
        size_t n = len - off < BLOCK ? len - off : BLOCK;
        uint64_t s = seed * 0x9E3779B97F4A7C15ull + b + 0x10605;
        rng_t r; out_t o;
        r.s = sm64(&s);
        o.p = dst + off; o.end = dst + off + n;
        gen_log(&o, &r);
        off += n; b++;
    }
}
