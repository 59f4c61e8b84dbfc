/*
 * 4mc_amd/csrc/cli.c — `4mc` command line on the GPU block engine.
 *
 * Keeps the contract of the reference CLI (native/4mccli.c:132-152 usage, :190-271 switches,
 * :283-333 output-name inference, :343-358 dispatch; man page native/4mc.1): same flags
 * (-z -1..-4 -d -t -c -f -v -q -V -h/-H, -l accepted and ignored), aggregated switches, `-`
 * / stdin / stdout / null markers, automatic .4mc/.4mz naming, display levels and exit codes.
 * The work itself is done by fourmc_file.c through batched HIP launches.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "fourmc.h"

#define CLI_VERSION "v2.0.0-mi355x"
#define EXT_4MC ".4mc"
#define EXT_4MZ ".4mz"

static const char* prog;
static int level_disp = 2;   /* 0 none, 1 errors, 2 +results/warnings, 3 +progress, 4 +information */

#define SAY(...)        fprintf(stderr, __VA_ARGS__)
#define SAY_AT(l, ...)  do { if (level_disp >= (l)) SAY(__VA_ARGS__); } while (0)

static void welcome(void)
{
    SAY("*** 4mc CLI %i-bits %s by %s (%s) ***\n*** Unleashing the power of LZ4 and ZSTD by Yann Collet/Facebook ***\n",
        (int)(sizeof(void*) * 8), CLI_VERSION, "Carlo Medas", __DATE__);
}

static void usage(void)
{
    SAY("Usage :\n");
    SAY("      %s [arg] [input] [output]\n", prog);
    SAY("\n");
    SAY("input   : a filename\n");
    SAY("          with no FILE, or when FILE is - or %s, read standard input\n", FOURMC_STDINMARK);
    SAY("Arguments :\n");
    SAY(" -z     : zstd compression (default is LZ4) \n");
    SAY(" -1     : Fast compression (default) \n");
    SAY(" -2     : Medium compression \n");
    SAY(" -3     : High compression \n");
    SAY(" -4     : Ultra compression \n");
    SAY(" -d     : decompression (default for %s and %s exts)\n", EXT_4MC, EXT_4MZ);
    SAY(" -f     : overwrite output without prompting \n");
    SAY(" -V     : display Version number and exit\n");
    SAY(" -v     : verbose mode\n");
    SAY(" -q     : quiet mode\n");
    SAY(" -h     : display help and exit\n");
}

static void bad_usage(void)
{
    SAY_AT(1, "Incorrect command line arguments\n");
    if (level_disp >= 1) usage();
    exit(1);
}

static int ends_with(const char* s, const char* ext)
{
    size_t l = strlen(s), e = strlen(ext);
    return l >= e && !strcmp(s + l - e, ext);
}

int main(int argc, char** argv)
{
    int i, level = 0, decode = 0, force_stdout = 0, force_compress = 0, overwrite = 0, zstd = 0;
    char *in_name = NULL, *out_name = NULL, *owned = NULL;
    static char m_stdin[] = FOURMC_STDINMARK, m_stdout[] = FOURMC_STDOUTMARK, m_nul[] = FOURMC_NULMARK;

    prog = argv[0];
    for (i = 1; i < argc; i++) {
        char* a = argv[i];
        if (!a) continue;
        if (a[0] != '-') {
            if (!in_name) { in_name = a; continue; }
            if (!out_name) { out_name = strcmp(a, FOURMC_NULL_OUTPUT) ? a : m_nul; }
            continue;
        }
        if (a[1] == 0) {                         /* bare '-' : stdin, then stdout */
            if (!in_name) in_name = m_stdin; else out_name = m_stdout;
            continue;
        }
        for (a++; *a; a++) {
            if (*a >= '0' && *a <= '9') {
                level = 0;
                while (*a >= '0' && *a <= '9') level = level * 10 + (*a++ - '0');
                a--;
                continue;
            }
            switch (*a) {
                case 'V': welcome(); return 0;
                case 'h': case 'H': usage(); return 0;
                case 'z': zstd = 1; force_compress = 1; break;
                case 'l': break;                                  /* legacy flag: parsed, unused */
                case 'd': decode = 1; break;
                case 'c': force_stdout = 1; out_name = m_stdout; level_disp = 1; break;
                case 't': decode = 1; out_name = m_nul; break;
                case 'f': overwrite = 1; break;
                case 'v': level_disp = 4; break;
                case 'q': level_disp--; break;
                default: bad_usage();
            }
        }
    }
    if (level_disp >= 3) welcome();
    if (!in_name) in_name = m_stdin;
    if (!strcmp(in_name, FOURMC_STDINMARK) && isatty(fileno(stdin))) bad_usage();

    if (!out_name) {
        if (!isatty(fileno(stdout))) out_name = m_stdout;          /* default to a pipe if there is one */
        else {
            if (!decode && !force_compress && (ends_with(in_name, EXT_4MC) || ends_with(in_name, EXT_4MZ))) decode = 1;
            if (!decode) {
                size_t l = strlen(in_name);
                owned = (char*)calloc(1, l + 5);
                memcpy(owned, in_name, l);
                memcpy(owned + l, zstd ? EXT_4MZ : EXT_4MC, 5);
                out_name = owned;
                SAY_AT(2, "Compressed filename will be : %s \n", out_name);
            } else {
                size_t l = strlen(in_name);
                if (l > 4 && ends_with(in_name, EXT_4MC)) { /* LZ4 */ }
                else if (l > 4 && ends_with(in_name, EXT_4MZ)) zstd = 1;
                else { SAY_AT(1, "Cannot determine an output filename\n"); bad_usage(); }
                owned = (char*)calloc(1, l + 1);
                memcpy(owned, in_name, l - 4);
                out_name = owned;
                SAY_AT(2, "Decoding file %s \n", out_name);
                SAY_AT(2, zstd ? "Compression: ZSTD\n" : "Compression: LZ4\n");
            }
        }
    }
    /* no chatter in pure pipe mode */
    if (!strcmp(in_name, FOURMC_STDINMARK) && !strcmp(out_name, FOURMC_STDOUTMARK) && level_disp == 2) level_disp = 1;
    if (!strcmp(in_name, FOURMC_STDINMARK) && isatty(fileno(stdin))) bad_usage();
    if (!strcmp(out_name, FOURMC_STDOUTMARK) && isatty(fileno(stdout)) && !force_stdout) bad_usage();

    if (decode) {
        if (!zstd) fourMcDecompressFileName(level_disp, overwrite, in_name, out_name);
        else       fourMZDecompressFileName(level_disp, overwrite, in_name, out_name);
    } else {
        if (!zstd) { SAY_AT(2, "Compression: LZ4\n");  fourMCcompressFilename(level_disp, overwrite, in_name, out_name, level); }
        else       { SAY_AT(2, "Compression: ZSTD\n"); fourMZcompressFilename(level_disp, overwrite, in_name, out_name, level); }
    }
    free(owned);
    return 0;
}
