/*
 * oracle/orc_xxh32.c -- XXH32 restated (TEST INFRASTRUCTURE; see orc.h).
 *
 * Follows the published xxHash32 algorithm as used by lz4 v1.9.3's lib/xxhash.c, which
 * the reference compiles in (/root/reference/lz4-frame-conduit.cabal:52).  Used by the
 * frame layer for the header checksum byte, block checksums and the content checksum
 * (SURVEY.md section 8a row a5).
 */
#include "orc.h"
#include <string.h>

#define P1 2654435761u
#define P2 2246822519u
#define P3 3266489917u
#define P4  668265263u
#define P5  374761393u

static inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t rd32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline uint32_t round1(uint32_t acc, uint32_t w) { return rotl(acc + w * P2, 13) * P1; }

static uint32_t finish(uint32_t h, const uint8_t* p, size_t rem)
{
    while (rem >= 4) { h = rotl(h + rd32(p) * P3, 17) * P4; p += 4; rem -= 4; }
    while (rem > 0)  { h = rotl(h + (*p) * P5, 11) * P1; p++; rem--; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

uint32_t orc_xxh32(const void* data, size_t len, uint32_t seed)
{
    const uint8_t* p = (const uint8_t*)data;
    const uint8_t* end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t* lim = end - 16;
        do {
            v1 = round1(v1, rd32(p));      v2 = round1(v2, rd32(p + 4));
            v3 = round1(v3, rd32(p + 8));  v4 = round1(v4, rd32(p + 12));
            p += 16;
        } while (p <= lim);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    } else {
        h = seed + P5;
    }
    h += (uint32_t)len;
    return finish(h, p, (size_t)(end - p));
}

void orc_xxh32_reset(orc_xxh32_state* s, uint32_t seed)
{
    memset(s, 0, sizeof(*s));
    s->v[0] = seed + P1 + P2; s->v[1] = seed + P2; s->v[2] = seed; s->v[3] = seed - P1;
}

void orc_xxh32_update(orc_xxh32_state* s, const void* data, size_t len)
{
    const uint8_t* p = (const uint8_t*)data;
    const uint8_t* end = p + len;
    if (len == 0) return;
    s->total_len_32 += (uint32_t)len;
    s->large_len |= (uint32_t)((len >= 16) | (s->total_len_32 >= 16));
    if (s->memsize + len < 16) {
        memcpy(s->mem + s->memsize, p, len);
        s->memsize += (uint32_t)len;
        return;
    }
    if (s->memsize) {
        memcpy(s->mem + s->memsize, p, 16 - s->memsize);
        for (int i = 0; i < 4; i++) s->v[i] = round1(s->v[i], rd32(s->mem + 4 * i));
        p += 16 - s->memsize;
        s->memsize = 0;
    }
    while (p + 16 <= end) {
        for (int i = 0; i < 4; i++) s->v[i] = round1(s->v[i], rd32(p + 4 * i));
        p += 16;
    }
    if (p < end) {
        memcpy(s->mem, p, (size_t)(end - p));
        s->memsize = (uint32_t)(end - p);
    }
}

uint32_t orc_xxh32_digest(const orc_xxh32_state* s)
{
    uint32_t h;
    if (s->large_len) h = rotl(s->v[0], 1) + rotl(s->v[1], 7) + rotl(s->v[2], 12) + rotl(s->v[3], 18);
    else              h = s->v[2] /* == seed */ + P5;
    h += s->total_len_32;
    return finish(h, s->mem, s->memsize);
}
