/*
 * oracle/orc_cpu_baseline.c -- host-core baseline for bench.py's `cpu_baseline` (TEST INFRASTRUCTURE).
 *
 * SURVEY.md section 8d "CPU baseline (i)": N threads, each running the block encoder / decoder
 * on round-robin frame blocks of the same synthetic stream the GPU path is timed on.
 * Engine: the system liblz4.so.1 through dlopen (LZ4_compress_default / LZ4_decompress_safe --
 * the library the reference bundles, lz4-frame-conduit.cabal:48-52) when it is present
 * ("reference"), else the restatement in orc_lz4block.c ("port").  `--port` forces the latter.
 *
 *   orc_cpu_baseline FILE BLOCK_SIZE THREADS REPS [--port]
 * prints one JSON line: best-of-REPS compress / decompress seconds and sizes.
 */
#define _GNU_SOURCE
#include "orc.h"
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int (*comp_fn)(const char*, char*, int, int);
typedef int (*decomp_fn)(const char*, char*, int, int);

static int port_comp(const char* s, char* d, int n, int cap) { return orc_lz4_compress_default((const uint8_t*)s, (uint8_t*)d, n, cap); }
static int port_decomp(const char* s, char* d, int n, int cap) { return orc_lz4_decompress_safe((const uint8_t*)s, (uint8_t*)d, n, cap, 0); }

typedef struct {
    int tid, nthreads, mode;           /* mode 0 compress, 1 decompress */
    const uint8_t* src; size_t n, bs, nblocks;
    uint8_t* comp; size_t cstride; int* csize; uint8_t* back;
    comp_fn cf; decomp_fn df; int fail;
} job;

static void* worker(void* arg)
{
    job* j = (job*)arg;
    for (size_t b = (size_t)j->tid; b < j->nblocks; b += (size_t)j->nthreads) {
        size_t off = b * j->bs, len = j->n - off < j->bs ? j->n - off : j->bs;
        if (j->mode == 0) {
            j->csize[b] = j->cf((const char*)j->src + off, (char*)j->comp + b * j->cstride, (int)len, (int)j->cstride);
            if (j->csize[b] <= 0) j->fail = 1;
        } else {
            int r = j->df((const char*)j->comp + b * j->cstride, (char*)j->back + off, j->csize[b], (int)len);
            if (r != (int)len) j->fail = 1;
        }
    }
    return NULL;
}

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

int main(int argc, char** argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s FILE BLOCK_SIZE THREADS REPS [--port]\n", argv[0]); return 2; }
    size_t bs = (size_t)atoll(argv[2]); int nt = atoi(argv[3]); int reps = atoi(argv[4]);
    int force_port = argc > 5 && !strcmp(argv[5], "--port");
    comp_fn cf = port_comp; decomp_fn df = port_decomp; const char* kind = "port";
    if (!force_port) {
        void* h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL);
        if (h) {
            comp_fn c = (comp_fn)dlsym(h, "LZ4_compress_default"); decomp_fn d = (decomp_fn)dlsym(h, "LZ4_decompress_safe");
            if (c && d) { cf = c; df = d; kind = "reference"; }
        }
    }
    FILE* f = fopen(argv[1], "rb"); if (!f) { perror("open"); return 1; }
    fseek(f, 0, SEEK_END); size_t n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* src = malloc(n); if (fread(src, 1, n, f) != n) { perror("read"); return 1; } fclose(f);
    size_t nblocks = (n + bs - 1) / bs, cstride = bs + bs / 255 + 16;
    uint8_t* comp = malloc(nblocks * cstride); uint8_t* back = malloc(n); int* csize = calloc(nblocks, sizeof(int));
    memset(comp, 0, nblocks * cstride); memset(back, 0, n);            /* touch all pages */
    pthread_t th[256]; job jobs[256]; if (nt > 256) nt = 256;
    double best[2] = {1e30, 1e30}; int fail = 0;
    for (int r = 0; r < reps + 1; r++) {                                /* rep 0 = warm-up */
        for (int mode = 0; mode < 2; mode++) {
            double t0 = now();
            for (int t = 0; t < nt; t++) {
                jobs[t] = (job){t, nt, mode, src, n, bs, nblocks, comp, cstride, csize, back, cf, df, 0};
                pthread_create(&th[t], NULL, worker, &jobs[t]);
            }
            for (int t = 0; t < nt; t++) { pthread_join(th[t], NULL); fail |= jobs[t].fail; }
            double dt = now() - t0;
            if (r > 0 && dt < best[mode]) best[mode] = dt;
        }
    }
    size_t ctotal = 0; for (size_t b = 0; b < nblocks; b++) ctotal += (size_t)csize[b];
    int ok = !fail && memcmp(src, back, n) == 0;
    printf("{\"kind\": \"%s\", \"threads\": %d, \"bytes\": %zu, \"block_size\": %zu, \"compressed\": %zu, "
           "\"t_comp\": %.6f, \"t_decomp\": %.6f, \"roundtrip_ok\": %s}\n",
           kind, nt, n, bs, ctotal, best[0], best[1], ok ? "true" : "false");
    return ok ? 0 : 1;
}
