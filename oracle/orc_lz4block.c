/*
 * oracle/orc_lz4block.c -- LZ4 block encoder/decoder restated (TEST INFRASTRUCTURE; see orc.h).
 *
 * Restates, from the published LZ4 Block format and the observable behaviour of lz4 v1.9.3
 * (lib/lz4.c, which the reference compiles in: /root/reference/lz4-frame-conduit.cabal:49),
 * the two functions the frame layer runs per block:
 *   - the level-0 "fast" encoder: greedy single-probe hash-table match finder with skip
 *     acceleration, selected because the reference hard-codes compressionLevel = 0
 *     (/root/reference/src/Codec/Compression/LZ4/Conduit.hsc:260);
 *   - the bounds-checked ("safe") decoder.
 * The encoder is written to be BIT-EXACT with liblz4 1.9.3 (pinned by the fixtures in tests/golden):
 * same hash functions, same table geometry (8192 x u16 + 4-byte hash below 64 KiB+11,
 * 4096 x u32 + 5-byte hash otherwise), same probe/insert order, same skip schedule,
 * same end-of-block rules and the same output-budget checks.
 */
#include "orc.h"
#include <string.h>

#define MINMATCH      4
#define MFLIMIT       12
#define LASTLITERALS  5
#define MAX_DISTANCE  65535u
#define LIMIT_64K     (65536 + (MFLIMIT - 1))
#define SKIP_TRIGGER  6
#define MAX_INPUT     0x7E000000

typedef enum { TT_U16 = 0, TT_U32 = 1 } table_kind;

static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

int orc_lz4_compress_bound(int n)
{
    return ((unsigned)n > (unsigned)MAX_INPUT) ? 0 : n + n / 255 + 16;
}

void orc_lz4_stream_reset(orc_lz4_stream* s) { memset(s, 0, sizeof(*s)); }

/* 4-byte Fibonacci hash to 13 bits (u16 table) / 5-byte hash to 12 bits (u32 table, 64-bit LE hosts) */
static inline uint32_t hash_at(const uint8_t* p, table_kind k)
{
    if (k == TT_U16) return (rd32(p) * 2654435761u) >> (32 - 13);
    return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> (64 - 12));
}
static inline uint32_t tab_get(const orc_lz4_stream* s, uint32_t h, table_kind k)
{
    return (k == TT_U16) ? ((const uint16_t*)s->table)[h] : s->table[h];
}
static inline void tab_put(orc_lz4_stream* s, uint32_t h, uint32_t idx, table_kind k)
{
    if (k == TT_U16) ((uint16_t*)s->table)[h] = (uint16_t)idx; else s->table[h] = idx;
}

static inline unsigned count_equal(const uint8_t* a, const uint8_t* b, const uint8_t* a_limit)
{
    const uint8_t* const a0 = a;
    while (a + 8 <= a_limit) {
        uint64_t d = rd64(a) ^ rd64(b);
        if (d) return (unsigned)(a - a0) + (unsigned)(__builtin_ctzll(d) >> 3);
        a += 8; b += 8;
    }
    while (a < a_limit && *a == *b) { a++; b++; }
    return (unsigned)(a - a0);
}

/*
 * Generic level-0 encoder.  Positions are indexes `start + (p - src)` so the table can
 * persist across linked blocks; `prefix` bytes of history lie directly in front of src.
 * `limited`: apply the output-budget checks against dst_cap (return 0 when they fail).
 */
static int encode_block(orc_lz4_stream* st, const uint8_t* src, uint8_t* dst, int n, int dst_cap,
                        int limited, table_kind k, uint32_t prefix)
{
    const uint32_t start = st->current_offset;
    const uint32_t idx_floor = start - prefix;        /* lowest index that is backed by memory */
    const long low_rel = -(long)prefix;               /* catch-up may not go below this (relative to src) */
    const long iend = n, mflimit1 = (long)n - MFLIMIT + 1, matchlimit = (long)n - LASTLITERALS;
    long ip = 0, anchor = 0, match = 0;
    size_t op = 0;                                    /* offset in dst */
    const size_t ocap = (size_t)dst_cap;
    size_t token = 0;
    uint32_t fwd_h;

    if (k == TT_U16 && n >= LIMIT_64K) return 0;
    st->dict_size += (uint32_t)n;
    st->current_offset += (uint32_t)n;

    if (n < MFLIMIT + 1) goto last_literals;

    tab_put(st, hash_at(src, k), start, k);
    ip = 1; fwd_h = hash_at(src + 1, k);

    for (;;) {
        {   /* search: probe one candidate per position, stride grows after 64 misses */
            long fwd = ip; long step = 1; unsigned tries = 1u << SKIP_TRIGGER;
            for (;;) {
                uint32_t const h = fwd_h;
                uint32_t const cur = start + (uint32_t)fwd;
                uint32_t const cand = tab_get(st, h, k);
                ip = fwd; fwd += step; step = (long)(tries++ >> SKIP_TRIGGER);
                if (fwd > mflimit1) goto last_literals;
                match = (long)((int64_t)cand - (int64_t)start);
                fwd_h = hash_at(src + fwd, k);
                tab_put(st, h, cur, k);
                if (cand < idx_floor) continue;                             /* not backed by memory */
                if (k != TT_U16 && cand + MAX_DISTANCE < cur) continue;      /* too far */
                if (rd32(src + match) == rd32(src + ip)) break;
            }
        }
        /* catch up: extend the match backwards over pending literals */
        while (ip > anchor && match > low_rel && src[ip - 1] == src[match - 1]) { ip--; match--; }

        {   /* literal run */
            unsigned const lit = (unsigned)(ip - anchor);
            token = op++;
            if (limited && op + lit + (2 + 1 + LASTLITERALS) + lit / 255 > ocap) return 0;
            if (lit >= 15) {
                unsigned len = lit - 15;
                dst[token] = 0xF0;
                for (; len >= 255; len -= 255) dst[op++] = 255;
                dst[op++] = (uint8_t)len;
            } else dst[token] = (uint8_t)(lit << 4);
            memcpy(dst + op, src + anchor, lit); op += lit;
        }
next_match:
        {   unsigned const off = (unsigned)(ip - match);
            dst[op++] = (uint8_t)off; dst[op++] = (uint8_t)(off >> 8);
        }
        {   unsigned mcode = count_equal(src + ip + MINMATCH, src + match + MINMATCH, src + matchlimit);
            ip += (long)mcode + MINMATCH;
            if (limited && op + (1 + LASTLITERALS) + (mcode + 240) / 255 > ocap) return 0;
            if (mcode >= 15) {
                dst[token] += 15; mcode -= 15;
                memset(dst + op, 0xFF, mcode / 255); op += mcode / 255;
                dst[op++] = (uint8_t)(mcode % 255);
            } else dst[token] += (uint8_t)mcode;
        }
        anchor = ip;
        if (ip >= mflimit1) break;

        tab_put(st, hash_at(src + ip - 2, k), start + (uint32_t)(ip - 2), k);

        {   /* immediate re-test at the position right after the match */
            uint32_t const h = hash_at(src + ip, k);
            uint32_t const cur = start + (uint32_t)ip;
            uint32_t const cand = tab_get(st, h, k);
            match = (long)((int64_t)cand - (int64_t)start);
            tab_put(st, h, cur, k);
            if (cand >= idx_floor && (k == TT_U16 || cand + MAX_DISTANCE >= cur)
                && rd32(src + match) == rd32(src + ip)) {
                token = op++; dst[token] = 0;
                goto next_match;
            }
        }
        fwd_h = hash_at(src + (++ip), k);
    }

last_literals:
    {   size_t const run = (size_t)(iend - anchor);
        if (limited && op + run + 1 + ((run + 255 - 15) / 255) > ocap) return 0;
        if (run >= 15) {
            size_t acc = run - 15;
            dst[op++] = 0xF0;
            for (; acc >= 255; acc -= 255) dst[op++] = 255;
            dst[op++] = (uint8_t)acc;
        } else dst[op++] = (uint8_t)(run << 4);
        memcpy(dst + op, src + anchor, run); op += run;
    }
    return (int)op;
}

int orc_lz4_compress_default(const uint8_t* src, uint8_t* dst, int n, int dst_cap)
{
    orc_lz4_stream st;
    int bound = orc_lz4_compress_bound(n);
    int limited;
    if ((unsigned)n > (unsigned)MAX_INPUT) return 0;
    limited = dst_cap < bound;
    if (n == 0) { if (limited && dst_cap <= 0) return 0; dst[0] = 0; return 1; }
    orc_lz4_stream_reset(&st);
    return encode_block(&st, src, dst, n, dst_cap, limited, (n < LIMIT_64K) ? TT_U16 : TT_U32, 0);
}

int orc_lz4_compress_continue(orc_lz4_stream* s, const uint8_t* src, uint8_t* dst, int n, int dst_cap)
{
    if ((unsigned)n > (unsigned)MAX_INPUT) return 0;
    if (n == 0) { if (dst_cap <= 0) return 0; dst[0] = 0; return 1; }
    /* index renormalisation of the real library (after 2 GiB of input) is not restated:
       reset instead, which only loses matches across that single boundary */
    if (s->current_offset + (uint32_t)n > 0x80000000u) { orc_lz4_stream_reset(s); }
    return encode_block(s, src, dst, n, dst_cap, 1, TT_U32, s->dict_size);
}

/* ------------------------------------------------------------------------------------------ */

int orc_lz4_decompress_safe(const uint8_t* src, uint8_t* dst, int csize, int dst_cap, size_t dict_size)
{
    long ip = 0, op = 0;
    const long iend = csize, oend = dst_cap;
    if (csize <= 0) return -1;
    if (dst_cap == 0) return (csize == 1 && src[0] == 0) ? 0 : -1;

    for (;;) {
        unsigned const token = src[ip++];
        size_t len = token >> 4;
        if (len == 15) {
            unsigned s;
            do {
                if (ip >= iend) return -1;
                s = src[ip++]; len += s;
            } while (s == 255);
            if (len > (size_t)0x7FFFFFFF) return -1;
        }
        /* literals */
        if ((long)len > oend - op - MFLIMIT || (long)len > iend - ip - (2 + 1 + LASTLITERALS)) {
            /* must be the last sequence: literals end exactly at the input end */
            if (ip + (long)len != iend || op + (long)len > oend) return -1;
            memmove(dst + op, src + ip, len);
            op += (long)len;
            break;
        }
        memcpy(dst + op, src + ip, len);
        ip += (long)len; op += (long)len;

        {   /* match */
            unsigned const off = (unsigned)src[ip] | ((unsigned)src[ip + 1] << 8);
            size_t mlen = token & 15;
            ip += 2;
            if (off == 0) return -1;                       /* format: 0 is invalid */
            if ((size_t)off > (size_t)op + dict_size) return -1;
            if (mlen == 15) {
                unsigned s;
                do {
                    if (ip >= iend) return -1;
                    s = src[ip++]; mlen += s;
                    if (ip >= iend - (LASTLITERALS - 1)) return -1;
                } while (s == 255);
                if (mlen > (size_t)0x7FFFFFFF) return -1;
            }
            mlen += MINMATCH;
            if ((long)mlen > oend - op - LASTLITERALS) return -1;   /* last 5 bytes must be literals */
            {   uint8_t* d = dst + op; const uint8_t* m = d - off;
                if (off >= mlen) memcpy(d, m, mlen);
                else for (size_t i = 0; i < mlen; i++) d[i] = m[i];   /* forward overlap semantics */
            }
            op += (long)mlen;
        }
    }
    return (int)op;
}

int orc_lz4_count_sequences(const uint8_t* src, int csize)
{
    long ip = 0; const long iend = csize; int nseq = 0;
    while (ip < iend) {
        unsigned const token = src[ip++];
        size_t len = token >> 4, mlen = token & 15;
        if (len == 15) { unsigned s; do { if (ip >= iend) return -1; s = src[ip++]; len += s; } while (s == 255); }
        ip += (long)len; nseq++;
        if (ip >= iend) break;
        ip += 2;
        if (mlen == 15) { unsigned s; do { if (ip >= iend) return -1; s = src[ip++]; } while (s == 255); }
    }
    return nseq;
}
