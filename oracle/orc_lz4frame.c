/*
 * oracle/orc_lz4frame.c -- LZ4 frame layer restated (TEST INFRASTRUCTURE; see orc.h).
 *
 * Restates, from the LZ4 Frame format v1.6.x and the observable behaviour of lz4 v1.9.3's
 * lib/lz4frame.c (compiled into the reference: /root/reference/lz4-frame-conduit.cabal:50):
 *   - header / EndMark / checksum bytes written by LZ4F_compressBegin / LZ4F_compressEnd
 *     (called at Conduit.hsc:292 and :321), SURVEY.md section 8a row a6;
 *   - LZ4F_compressBound (Conduit.hsc:302);
 *   - the input staging of LZ4F_compressUpdate (Conduit.hsc:311): bytes are buffered until a
 *     full block exists, each full block becomes `u32le size | payload | [u32le xxh32]`,
 *     raw fallback with bit 31 when the encoder cannot fit blockSize-1 (row a1);
 *   - the per-block walk of LZ4F_decompress (Conduit.hsc:591), row a3, as a one-shot decoder.
 * Linked blocks are always encoded in "prefix mode" (history contiguous in front of the
 * block).  That is byte-identical to liblz4 whenever every LZ4F_compressUpdate call carries
 * fewer bytes than one block -- which is what the reference's compress conduit does with its
 * 16 KiB slices (Conduit.hsc:464,501) -- and decode-equivalent otherwise.
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

#define MAGIC        0x184D2204u
#define MAGIC_SKIP   0x184D2A50u
#define KB64         65536u

static inline void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline size_t err(int code) { return (size_t)-(ptrdiff_t)code; }

static const char* const k_names[] = {
    "OK_NoError", "ERROR_GENERIC", "ERROR_maxBlockSize_invalid", "ERROR_blockMode_invalid",
    "ERROR_contentChecksumFlag_invalid", "ERROR_compressionLevel_invalid", "ERROR_headerVersion_wrong",
    "ERROR_blockChecksum_invalid", "ERROR_reservedFlag_set", "ERROR_allocation_failed",
    "ERROR_srcSize_tooLarge", "ERROR_dstMaxSize_tooSmall", "ERROR_frameHeader_incomplete",
    "ERROR_frameType_unknown", "ERROR_frameSize_wrong", "ERROR_srcPtr_wrong", "ERROR_decompressionFailed",
    "ERROR_headerChecksum_invalid", "ERROR_contentChecksum_invalid", "ERROR_frameDecoding_alreadyStarted",
    "ERROR_maxCode" };

unsigned orc_is_error(size_t code) { return code > err(ORC_ERR_maxCode); }
const char* orc_error_name(size_t code)
{
    if (orc_is_error(code)) return k_names[(int)(-(ptrdiff_t)code)];
    return "Unspecified error code";
}

size_t orc_block_size(uint32_t id)
{
    if (id == 0) id = 4;
    if (id < 4 || id > 7) return 0;
    return (size_t)1 << (8 + 2 * id);      /* 4->64K 5->256K 6->1M 7->4M */
}

static size_t bound_internal(size_t src_size, const orc_prefs* p, size_t already)
{
    orc_prefs worst; memset(&worst, 0, sizeof(worst));
    worst.frameInfo.contentChecksumFlag = 1; worst.frameInfo.blockChecksumFlag = 1;
    if (!p) p = &worst;
    {
        unsigned const flush = p->autoFlush | (src_size == 0);
        size_t const bs = orc_block_size(p->frameInfo.blockSizeID);
        size_t const buffered = already < bs - 1 ? already : bs - 1;
        size_t const max_src = src_size + buffered;
        size_t const nfull = max_src / bs;
        size_t const partial = max_src & (bs - 1);
        size_t const last = flush ? partial : 0;
        size_t const nblocks = nfull + (last > 0);
        size_t const bck = 4 * (size_t)p->frameInfo.blockChecksumFlag;
        size_t const frame_end = 4 + 4 * (size_t)p->frameInfo.contentChecksumFlag;
        return (4 + bck) * nblocks + bs * nfull + last + frame_end;
    }
}

size_t orc_compress_bound(size_t src_size, const orc_prefs* p)
{
    if (p && p->autoFlush) return bound_internal(src_size, p, 0);
    return bound_internal(src_size, p, (size_t)-1);
}

size_t orc_write_header(uint8_t* dst, size_t cap, const orc_prefs* p)
{
    orc_prefs z; uint8_t* d = dst; uint32_t bsid;
    if (cap < 19) return err(ORC_ERR_dstMaxSize_tooSmall);
    if (!p) { memset(&z, 0, sizeof(z)); p = &z; }
    bsid = p->frameInfo.blockSizeID ? p->frameInfo.blockSizeID : 4;
    wr32(d, MAGIC); d += 4;
    *d++ = (uint8_t)((1u << 6) + ((p->frameInfo.blockMode & 1u) << 5) + ((p->frameInfo.blockChecksumFlag & 1u) << 4)
                     + ((unsigned)(p->frameInfo.contentSize > 0) << 3) + ((p->frameInfo.contentChecksumFlag & 1u) << 2)
                     + (p->frameInfo.dictID > 0));
    *d++ = (uint8_t)((bsid & 7u) << 4);
    if (p->frameInfo.contentSize) { wr32(d, (uint32_t)p->frameInfo.contentSize); wr32(d + 4, (uint32_t)(p->frameInfo.contentSize >> 32)); d += 8; }
    if (p->frameInfo.dictID) { wr32(d, p->frameInfo.dictID); d += 4; }
    *d = (uint8_t)(orc_xxh32(dst + 4, (size_t)(d - (dst + 4)), 0) >> 8); d++;
    return (size_t)(d - dst);
}

/* ---------------- streaming compressor ---------------- */
struct orc_cctx {
    orc_prefs prefs;
    size_t    block_size;
    int       stage;              /* 0 idle, 1 header written */
    uint8_t*  buf;                /* [64 KiB history][block_size staging] */
    size_t    hist;               /* valid history bytes, ending at buf+64K */
    size_t    staged;             /* bytes staged at buf+64K */
    uint64_t  total_in;
    orc_lz4_stream lz;
    orc_xxh32_state xxh;
};

orc_cctx* orc_cctx_create(void) { return (orc_cctx*)calloc(1, sizeof(orc_cctx)); }
void orc_cctx_free(orc_cctx* c) { if (c) { free(c->buf); free(c); } }

size_t orc_compress_begin(orc_cctx* c, uint8_t* dst, size_t cap, const orc_prefs* prefs)
{
    size_t h;
    if (cap < 19) return err(ORC_ERR_dstMaxSize_tooSmall);
    if (prefs) c->prefs = *prefs; else memset(&c->prefs, 0, sizeof(c->prefs));
    if (c->prefs.frameInfo.blockSizeID == 0) c->prefs.frameInfo.blockSizeID = 4;
    c->block_size = orc_block_size(c->prefs.frameInfo.blockSizeID);
    if (c->block_size == 0) return err(ORC_ERR_maxBlockSize_invalid);
    free(c->buf);
    c->buf = (uint8_t*)malloc(KB64 + c->block_size);
    if (!c->buf) return err(ORC_ERR_allocation_failed);
    c->hist = 0; c->staged = 0; c->total_in = 0;
    orc_lz4_stream_reset(&c->lz);
    orc_xxh32_reset(&c->xxh, 0);
    h = orc_write_header(dst, cap, &c->prefs);
    c->stage = 1;
    return h;
}

/* one frame block out of c->buf+64K[0..n): size word, payload (or raw), optional block checksum */
static size_t make_block(orc_cctx* c, uint8_t* dst, size_t n)
{
    uint8_t* const blk = c->buf + KB64;
    int csize;
    if (c->prefs.frameInfo.blockMode == 0) {         /* linked: prefix mode */
        c->lz.dict_size = (uint32_t)c->hist;
        csize = orc_lz4_compress_continue(&c->lz, blk, dst + 4, (int)n, (int)n - 1);
    } else {
        csize = orc_lz4_compress_default(blk, dst + 4, (int)n, (int)n - 1);
    }
    if (csize == 0) { csize = (int)n; wr32(dst, (uint32_t)n | 0x80000000u); memcpy(dst + 4, blk, n); }
    else wr32(dst, (uint32_t)csize);
    if (c->prefs.frameInfo.blockChecksumFlag) wr32(dst + 4 + csize, orc_xxh32(dst + 4, (size_t)csize, 0));
    if (c->prefs.frameInfo.blockMode == 0) {         /* slide history: keep last 64 KiB in front */
        size_t const tot = c->hist + n;
        size_t const keep = tot < KB64 ? tot : KB64;
        memmove(c->buf + KB64 - keep, blk + n - keep, keep);
        c->hist = keep;
    }
    return 4 + (size_t)csize + 4 * (size_t)c->prefs.frameInfo.blockChecksumFlag;
}

size_t orc_compress_update(orc_cctx* c, uint8_t* dst, size_t cap, const uint8_t* src, size_t n)
{
    uint8_t* d = dst; const uint8_t* s = src; const uint8_t* const send = src + n;
    if (c->stage != 1) return err(ORC_ERR_GENERIC);
    if (cap < bound_internal(n, &c->prefs, c->staged)) return err(ORC_ERR_dstMaxSize_tooSmall);
    while (s < send) {
        size_t want = c->block_size - c->staged;
        size_t take = (size_t)(send - s) < want ? (size_t)(send - s) : want;
        memcpy(c->buf + KB64 + c->staged, s, take);
        c->staged += take; s += take;
        if (c->staged == c->block_size) { d += make_block(c, d, c->block_size); c->staged = 0; }
    }
    if (c->prefs.autoFlush && c->staged) { d += make_block(c, d, c->staged); c->staged = 0; }
    if (c->prefs.frameInfo.contentChecksumFlag) orc_xxh32_update(&c->xxh, src, n);
    c->total_in += n;
    return (size_t)(d - dst);
}

size_t orc_compress_end(orc_cctx* c, uint8_t* dst, size_t cap)
{
    uint8_t* d = dst;
    if (c->staged) {
        if (c->stage != 1) return err(ORC_ERR_GENERIC);
        if (cap < c->staged + 8) return err(ORC_ERR_dstMaxSize_tooSmall);
        d += make_block(c, d, c->staged); c->staged = 0;
    }
    cap -= (size_t)(d - dst);
    if (cap < 4) return err(ORC_ERR_dstMaxSize_tooSmall);
    wr32(d, 0); d += 4;
    if (c->prefs.frameInfo.contentChecksumFlag) {
        if (cap < 8) return err(ORC_ERR_dstMaxSize_tooSmall);
        wr32(d, orc_xxh32_digest(&c->xxh)); d += 4;
    }
    c->stage = 0;
    if (c->prefs.frameInfo.contentSize && c->prefs.frameInfo.contentSize != c->total_in) return err(ORC_ERR_frameSize_wrong);
    return (size_t)(d - dst);
}

size_t orc_conduit_compress(const uint8_t* src, size_t n, const orc_prefs* prefs, size_t slice, uint8_t* dst, size_t cap)
{
    /* Conduit.hsc:457-533 -- begin, update per <=slice bytes, end; output concatenated */
    orc_cctx* c = orc_cctx_create(); size_t used = 0, r, off = 0;
    if (!c) return err(ORC_ERR_allocation_failed);
    r = orc_compress_begin(c, dst, cap, prefs);
    if (orc_is_error(r)) goto out;
    used = r;
    while (off < n) {
        size_t take = n - off < slice ? n - off : slice;
        r = orc_compress_update(c, dst + used, cap - used, src + off, take);
        if (orc_is_error(r)) goto out;
        used += r; off += take;
    }
    r = orc_compress_end(c, dst + used, cap - used);
    if (orc_is_error(r)) goto out;
    used += r; r = used;
out:
    orc_cctx_free(c);
    return r;
}

/* ---------------- one-shot frame decoder ---------------- */
size_t orc_decompress_frame(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* consumed, orc_frame_info* info_out)
{
    orc_frame_info fi; size_t pos, hsize, bs, op = 0; uint64_t remaining = 0;
    orc_xxh32_state xxh;
    memset(&fi, 0, sizeof(fi)); *consumed = 0;
    if (n < 7) return err(ORC_ERR_frameHeader_incomplete);
    if ((rd32(src) & 0xFFFFFFF0u) == MAGIC_SKIP) {
        size_t sz;
        if (n < 8) return err(ORC_ERR_frameHeader_incomplete);
        sz = rd32(src + 4);
        if (n < 8 + sz) return err(ORC_ERR_frameHeader_incomplete);
        fi.frameType = 1; fi.contentSize = sz;
        if (info_out) *info_out = fi;
        *consumed = 8 + sz;
        return 0;
    }
    if (rd32(src) != MAGIC) return err(ORC_ERR_frameType_unknown);
    {
        unsigned const flg = src[4];
        if ((flg >> 1) & 1) return err(ORC_ERR_reservedFlag_set);
        if (((flg >> 6) & 3) != 1) return err(ORC_ERR_headerVersion_wrong);
        hsize = 7 + ((flg >> 3) & 1 ? 8 : 0) + ((flg & 1) ? 4 : 0);
        if (n < hsize) return err(ORC_ERR_frameHeader_incomplete);
        {
            unsigned const bd = src[5];
            unsigned const bsid = (bd >> 4) & 7;
            if ((bd >> 7) & 1) return err(ORC_ERR_reservedFlag_set);
            if (bsid < 4) return err(ORC_ERR_maxBlockSize_invalid);
            if (bd & 15) return err(ORC_ERR_reservedFlag_set);
            fi.blockSizeID = bsid;
        }
        if ((uint8_t)(orc_xxh32(src + 4, hsize - 5, 0) >> 8) != src[hsize - 1]) return err(ORC_ERR_headerChecksum_invalid);
        fi.blockMode = (flg >> 5) & 1; fi.blockChecksumFlag = (flg >> 4) & 1; fi.contentChecksumFlag = (flg >> 2) & 1;
        if ((flg >> 3) & 1) { fi.contentSize = (uint64_t)rd32(src + 6) | ((uint64_t)rd32(src + 10) << 32); remaining = fi.contentSize; }
        if (flg & 1) fi.dictID = rd32(src + hsize - 5);
    }
    if (info_out) *info_out = fi;
    bs = orc_block_size(fi.blockSizeID);
    pos = hsize;
    orc_xxh32_reset(&xxh, 0);
    for (;;) {
        uint32_t bh; size_t csz; size_t produced;
        if (n - pos < 4) return err(ORC_ERR_frameHeader_incomplete);     /* truncated (streaming API would ask for more) */
        bh = rd32(src + pos); pos += 4;
        if (bh == 0) break;
        csz = bh & 0x7FFFFFFFu;
        if (csz > bs) return err(ORC_ERR_maxBlockSize_invalid);
        if (n - pos < csz + 4 * (size_t)fi.blockChecksumFlag) return err(ORC_ERR_frameHeader_incomplete);
        if (fi.blockChecksumFlag && rd32(src + pos + csz) != orc_xxh32(src + pos, csz, 0)) return err(ORC_ERR_blockChecksum_invalid);
        if (bh & 0x80000000u) {
            if (cap - op < csz) return err(ORC_ERR_dstMaxSize_tooSmall);
            memcpy(dst + op, src + pos, csz); produced = csz;
        } else {
            size_t const room = cap - op < bs ? cap - op : bs;
            size_t const dict = (fi.blockMode == 0) ? (op < KB64 ? op : KB64) : 0;
            int const r = orc_lz4_decompress_safe(src + pos, dst + op, (int)csz, (int)room, dict);
            if (r < 0) return err(room < bs ? ORC_ERR_dstMaxSize_tooSmall : ORC_ERR_GENERIC);
            produced = (size_t)r;
        }
        if (fi.contentChecksumFlag) orc_xxh32_update(&xxh, dst + op, produced);
        if (fi.contentSize) remaining -= produced;
        op += produced;
        pos += csz + 4 * (size_t)fi.blockChecksumFlag;
    }
    if (remaining) return err(ORC_ERR_frameSize_wrong);
    if (fi.contentChecksumFlag) {
        if (n - pos < 4) return err(ORC_ERR_frameHeader_incomplete);
        if (rd32(src + pos) != orc_xxh32_digest(&xxh)) return err(ORC_ERR_contentChecksum_invalid);
        pos += 4;
    }
    *consumed = pos;
    return op;
}
