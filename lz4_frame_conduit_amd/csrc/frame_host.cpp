// frame_host.cpp -- host frame layer of liblz4f_mi355x: the LZ4F-compatible streaming contexts and
// the host-pointer bulk calls.  (boundary 8b.)
//
// What stays on the host is what the reference's own frame layer does between blocks: header /
// EndMark bytes, staging input into whole blocks, the dst/src bookkeeping of LZ4F_decompress'
// state machine, and the whole-stream content checksum (XXH32 cannot be combined from parts, so it
// is one serial chain -- SURVEY.md 8f N1).  Every block encode, block decode and block checksum
// is a HIP kernel launch through engine.hip.  There is no CPU codec in this library.
//
// Behavioural contract restated from lz4 v1.9.3's lib/lz4frame.c (not in /root/reference, see
// oracle/orc.h) and pinned by tests/golden (return values, hints, error names).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <exception>
#include <functional>
#include <thread>
#include <new>
#include <vector>

#include "engine.hpp"

using namespace lz4f;

namespace lz4f {

static const char* const k_err_names[] = {
    "OK_NoError", "ERROR_GENERIC", "ERROR_maxBlockSize_invalid", "ERROR_blockMode_invalid",
    "ERROR_contentChecksumFlag_invalid", "ERROR_compressionLevel_invalid", "ERROR_headerVersion_wrong",
    "ERROR_blockChecksum_invalid", "ERROR_reservedFlag_set", "ERROR_allocation_failed", "ERROR_srcSize_tooLarge",
    "ERROR_dstMaxSize_tooSmall", "ERROR_frameHeader_incomplete", "ERROR_frameType_unknown", "ERROR_frameSize_wrong",
    "ERROR_srcPtr_wrong", "ERROR_decompressionFailed", "ERROR_headerChecksum_invalid", "ERROR_contentChecksum_invalid",
    "ERROR_frameDecoding_alreadyStarted", "ERROR_maxCode"};

const char* err_name(size_t v)
{
    if (is_err(v)) return k_err_names[(int)(-(ptrdiff_t)v)];
    return "Unspecified error code";
}

// ---------------- XXH32 on the host ----------------
static const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
static inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }
static inline void st32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static inline uint32_t lane_step(uint32_t acc, uint32_t w) { return rotl(acc + w * P2, 13) * P1; }

void Xxh32State::reset(uint32_t seed)
{
    total_len = 0; large = 0; memsize = 0;
    v[0] = seed + P1 + P2; v[1] = seed + P2; v[2] = seed; v[3] = seed - P1;
}
void Xxh32State::update(const void* data, size_t len)
{
    const uint8_t* p = (const uint8_t*)data; const uint8_t* const end = p + len;
    if (!len) return;
    total_len += (uint32_t)len;
    large |= (uint32_t)((len >= 16) | (total_len >= 16));
    if (memsize + len < 16) { memcpy(mem + memsize, p, len); memsize += (uint32_t)len; return; }
    if (memsize) {
        memcpy(mem + memsize, p, 16 - memsize);
        for (int i = 0; i < 4; i++) v[i] = lane_step(v[i], le32(mem + 4 * i));
        p += 16 - memsize; memsize = 0;
    }
    for (; p + 16 <= end; p += 16) for (int i = 0; i < 4; i++) v[i] = lane_step(v[i], le32(p + 4 * i));
    if (p < end) { memsize = (uint32_t)(end - p); memcpy(mem, p, memsize); }
}
uint32_t Xxh32State::digest() const
{
    uint32_t h = large ? rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18) : v[2] + P5;
    h += total_len;
    const uint8_t* p = mem; uint32_t rem = memsize;
    for (; rem >= 4; p += 4, rem -= 4) h = rotl(h + le32(p) * P3, 17) * P4;
    for (; rem; p++, rem--) h = rotl(h + (*p) * P5, 11) * P1;
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}
uint32_t xxh32_host(const void* data, size_t len, uint32_t seed)
{
    Xxh32State s; s.reset(seed); s.update(data, len); return s.digest();
}

// ---------------- header helpers ----------------
size_t block_size_of(unsigned id)
{
    if (id == 0) id = LZ4F_max64KB;
    if (id < 4 || id > 7) return 0;
    return (size_t)1 << (8 + 2 * id);
}

size_t write_frame_header(uint8_t* dst, const LZ4F_preferences_t& p)
{
    uint8_t* d = dst;
    st32(d, 0x184D2204u); d += 4;
    const LZ4F_frameInfo_t& f = p.frameInfo;
    *d++ = (uint8_t)((1u << 6) | (((unsigned)f.blockMode & 1u) << 5) | (((unsigned)f.blockChecksumFlag & 1u) << 4) |
                     ((unsigned)(f.contentSize > 0) << 3) | (((unsigned)f.contentChecksumFlag & 1u) << 2) | (unsigned)(f.dictID > 0));
    *d++ = (uint8_t)(((unsigned)f.blockSizeID & 7u) << 4);
    if (f.contentSize) { st32(d, (uint32_t)f.contentSize); st32(d + 4, (uint32_t)(f.contentSize >> 32)); d += 8; }
    if (f.dictID) { st32(d, f.dictID); d += 4; }
    *d = (uint8_t)(xxh32_host(dst + 4, (size_t)(d - (dst + 4))) >> 8);
    return (size_t)(d + 1 - dst);
}

size_t compress_bound_internal(size_t srcSize, const LZ4F_preferences_t* prefs, size_t alreadyBuffered)
{
    LZ4F_preferences_t worst; memset(&worst, 0, sizeof(worst));
    worst.frameInfo.contentChecksumFlag = LZ4F_contentChecksumEnabled;
    worst.frameInfo.blockChecksumFlag = LZ4F_blockChecksumEnabled;
    const LZ4F_preferences_t* p = prefs ? prefs : &worst;
    const unsigned flush = p->autoFlush | (srcSize == 0);
    const size_t bs = block_size_of(p->frameInfo.blockSizeID);
    if (!bs) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
    const size_t buffered = std::min(alreadyBuffered, bs - 1);
    const size_t maxSrc = srcSize + buffered;
    const size_t nfull = maxSrc / bs, partial = maxSrc & (bs - 1);
    const size_t last = flush ? partial : 0;
    const size_t nblocks = nfull + (last > 0);
    const size_t bck = 4 * (size_t)(p->frameInfo.blockChecksumFlag != 0);
    const size_t frameEnd = 4 + 4 * (size_t)(p->frameInfo.contentChecksumFlag != 0);
    return (4 + bck) * nblocks + bs * nfull + last + frameEnd;
}

size_t parse_frame_header(const uint8_t* src, size_t n, ParsedHeader* out)
{
    memset(out, 0, sizeof(*out));
    if (n < 7) return make_err(LZ4F_ERROR_frameHeader_incomplete);
    if (le32(src) != 0x184D2204u) return make_err(LZ4F_ERROR_frameType_unknown);
    const unsigned flg = src[4];
    if ((flg >> 1) & 1) return make_err(LZ4F_ERROR_reservedFlag_set);
    if (((flg >> 6) & 3) != 1) return make_err(LZ4F_ERROR_headerVersion_wrong);
    const size_t hs = 7 + (((flg >> 3) & 1) ? 8 : 0) + ((flg & 1) ? 4 : 0);
    if (n < hs) return make_err(LZ4F_ERROR_frameHeader_incomplete);
    const unsigned bd = src[5], bsid = (bd >> 4) & 7;
    if ((bd >> 7) & 1) return make_err(LZ4F_ERROR_reservedFlag_set);
    if (bsid < 4) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
    if (bd & 15) return make_err(LZ4F_ERROR_reservedFlag_set);
    if ((uint8_t)(xxh32_host(src + 4, hs - 5) >> 8) != src[hs - 1]) return make_err(LZ4F_ERROR_headerChecksum_invalid);
    out->info.blockMode = (LZ4F_blockMode_t)((flg >> 5) & 1);
    out->info.blockChecksumFlag = (LZ4F_blockChecksum_t)((flg >> 4) & 1);
    out->info.contentChecksumFlag = (LZ4F_contentChecksum_t)((flg >> 2) & 1);
    out->info.blockSizeID = (LZ4F_blockSizeID_t)bsid;
    out->info.frameType = LZ4F_frame;
    if ((flg >> 3) & 1) out->info.contentSize = le64(src + 6);
    if (flg & 1) out->info.dictID = le32(src + hs - 5);
    out->header_size = hs;
    out->max_block = block_size_of(bsid);
    return hs;
}

}  // namespace lz4f

// =================================================================================================
// compression context
struct LZ4F_cctx_s {
    LZ4F_preferences_t prefs;
    unsigned version;
    int stage;                       // 0 = idle, 1 = header written
    size_t block_size;
    // Page-locked staging the kernels read and write directly (engine: compress_block_pinned): pin_in = [64 KiB: the last <= 64 KiB of
    // input already encoded (linked mode), right-aligned][block_size: the input staged for the next block, always < block_size between
    // calls]; pin_out = the encoded block(s) of one call, then the result record.
    PinBuf pin_in, pin_out;
    std::vector<uint8_t> pageable;   // the same staging in ordinary memory where page-locked memory cannot be had (no usable device: block-level calls then fail loudly, staging does not)
    uint8_t* base = nullptr;
    size_t tmp_size, hist_len;
    uint64_t total_in;
    Xxh32State xxh;
    bool pinned() const { return base && base == (uint8_t*)pin_in.p; }
    uint8_t* stage_at() { return base + 65536; }
    const uint8_t* hist_at() const { return base + 65536 - hist_len; }
    bool ensure_staging()
    {
        if (base) return true;
        const size_t need = 65536 + block_size + 64;
        if (!pin_in.ensure(need)) { base = (uint8_t*)pin_in.p; return true; }
        try { pageable.assign(need, 0); } catch (...) { return false; }
        base = pageable.data();
        return true;
    }
    ~LZ4F_cctx_s() { pin_in.release(); pin_out.release(); }
};

static void push_history(LZ4F_cctx_s* c, const uint8_t* p, size_t n)
{
    if (c->prefs.frameInfo.blockMode != LZ4F_blockLinked) return;
    uint8_t* base = c->base;
    if (n >= 65536) { memmove(base, p + n - 65536, 65536); c->hist_len = 65536; return; }
    const size_t new_len = std::min<size_t>(65536, c->hist_len + n), keep_old = new_len - n;
    memmove(base + 65536 - new_len, base + 65536 - keep_old, keep_old);      // (the old bytes that stay, moved down by n)
    memmove(base + 65536 - n, p, n);
    c->hist_len = new_len;
}

// encode `n` bytes (whole blocks, the last may be short) that lie anywhere in host memory -> dst; returns bytes written or error
static size_t encode_blocks(LZ4F_cctx_s* c, uint8_t* dst, size_t cap, const uint8_t* src, size_t n)
{
    EngineLease eng;
    size_t r = eng.get();
    if (is_err(r)) return r;
    size_t written = 0;
    r = eng->compress_blocks_host(src, n, c->hist_at(), c->hist_len, (uint32_t)c->block_size,
                                  c->prefs.frameInfo.blockMode == LZ4F_blockLinked, c->prefs.frameInfo.blockChecksumFlag != 0, dst, cap, &written);
    if (is_err(r)) return r;
    push_history(c, src, n);
    return written;
}

// encode the `n` <= block_size bytes staged in pin_in (one block) -> dst.  The kernels read the staging buffer and write the block into
// pin_out through the link themselves: no copy calls, one synchronisation.
static size_t encode_staged(LZ4F_cctx_s* c, uint8_t* dst, size_t cap, size_t n)
{
    if (!c->pinned()) return encode_blocks(c, dst, cap, c->stage_at(), n);      // (staged copies; without a device this is where the call fails, loudly)
    EngineLease eng;
    size_t r = eng.get();
    if (is_err(r)) return r;
    const size_t out_cap = n + 256;
    if (c->pin_out.ensure(out_cap + 256)) return make_err(LZ4F_ERROR_allocation_failed);
    uint8_t* out = (uint8_t*)c->pin_out.p;
    size_t size = 0;
    r = eng->compress_block_pinned(c->hist_at(), c->hist_len, n, (uint32_t)c->block_size, c->prefs.frameInfo.blockMode == LZ4F_blockLinked,
                                   c->prefs.frameInfo.blockChecksumFlag != 0, out, out_cap + 192, nullptr, &size);
    if (is_err(r)) return r;
    if (size > cap) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
    memcpy(dst, out, size);
    push_history(c, c->stage_at(), n);
    return size;
}

extern "C" {

unsigned LZ4F_isError(LZ4F_errorCode_t code) { return is_err(code); }
const char* LZ4F_getErrorName(LZ4F_errorCode_t code) { return err_name(code); }
unsigned LZ4F_getVersion(void) { return LZ4F_VERSION; }

LZ4F_errorCode_t LZ4F_createCompressionContext(LZ4F_cctx** cctxPtr, unsigned version)
{
    if (!cctxPtr) return make_err(LZ4F_ERROR_GENERIC);
    LZ4F_cctx_s* c = new (std::nothrow) LZ4F_cctx_s();
    if (!c) return make_err(LZ4F_ERROR_allocation_failed);
    memset(&c->prefs, 0, sizeof(c->prefs));
    c->version = version; c->stage = 0; c->block_size = 0; c->total_in = 0; c->tmp_size = 0; c->hist_len = 0;
    *cctxPtr = c;
    return 0;
}

LZ4F_errorCode_t LZ4F_freeCompressionContext(LZ4F_cctx* cctx) { delete cctx; return 0; }

size_t LZ4F_compressBound(size_t srcSize, const LZ4F_preferences_t* prefsPtr)
{
    if (prefsPtr && prefsPtr->autoFlush) return compress_bound_internal(srcSize, prefsPtr, 0);
    return compress_bound_internal(srcSize, prefsPtr, (size_t)-1);
}

size_t LZ4F_compressBegin(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const LZ4F_preferences_t* prefsPtr)
{
    if (!c) return make_err(LZ4F_ERROR_GENERIC);
    if (dstCapacity < LZ4F_HEADER_SIZE_MAX) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
    if (prefsPtr) c->prefs = *prefsPtr; else memset(&c->prefs, 0, sizeof(c->prefs));
    if (c->prefs.compressionLevel > 2) {
        // levels >= 3 select LZ4HC upstream; unreachable from the reference (level fixed at 0, Conduit.hsc:260)
        set_last_error("compressionLevel %d: only the fast encoder exists in liblz4f_mi355x", c->prefs.compressionLevel);
        return make_err(LZ4F_ERROR_compressionLevel_invalid);
    }
    if (c->prefs.frameInfo.blockSizeID == 0) c->prefs.frameInfo.blockSizeID = LZ4F_max64KB;
    c->block_size = block_size_of(c->prefs.frameInfo.blockSizeID);
    if (!c->block_size) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
    c->tmp_size = 0; c->hist_len = 0; c->total_in = 0;
    c->base = nullptr;                                                    // (staging is made by the first LZ4F_compressUpdate: the block size may have changed)
    c->xxh.reset(0);
    const size_t h = write_frame_header((uint8_t*)dstBuffer, c->prefs);
    c->stage = 1;
    return h;
}

size_t LZ4F_compressUpdate(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize,
                           const LZ4F_compressOptions_t* /*cOptPtr: stableSrc is irrelevant, input is staged to the GPU anyway*/)
{
    if (!c || c->stage != 1) return make_err(LZ4F_ERROR_GENERIC);
    if (dstCapacity < compress_bound_internal(srcSize, &c->prefs, c->tmp_size)) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
    if (!c->ensure_staging()) return make_err(LZ4F_ERROR_allocation_failed);
    const uint8_t* src = (const uint8_t*)srcBuffer;
    uint8_t* dst = (uint8_t*)dstBuffer;
    const size_t B = c->block_size;
    size_t written = 0, pos = 0;
    const size_t avail = c->tmp_size + srcSize;
    size_t whole = (avail / B) * B;                           // bytes that form complete blocks
    if (c->prefs.autoFlush) whole = avail;                    // ... or everything
    if (whole) {
        if (c->tmp_size) {                                   // complete the staged block first (the reference's pattern: every call that emits, emits this one block)
            const size_t take = std::min(std::min(B - c->tmp_size, srcSize), whole - c->tmp_size);
            memcpy(c->stage_at() + c->tmp_size, src, take);
            const size_t n1 = c->tmp_size + take;
            c->tmp_size = 0; pos = take;
            const size_t r = encode_staged(c, dst, dstCapacity, n1);
            if (is_err(r)) return r;
            written += r; whole -= n1;
        }
        if (whole) {                                         // further whole blocks straight from the caller's buffer
            const size_t r = encode_blocks(c, dst + written, dstCapacity - written, src + pos, whole);
            if (is_err(r)) return r;
            written += r; pos += whole;
        }
    }
    if (pos < srcSize) { memcpy(c->stage_at() + c->tmp_size, src + pos, srcSize - pos); c->tmp_size += srcSize - pos; }      // necessarily < B
    if (c->prefs.frameInfo.contentChecksumFlag == LZ4F_contentChecksumEnabled) c->xxh.update(srcBuffer, srcSize);
    c->total_in += srcSize;
    return written;
}

size_t LZ4F_flush(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t*)
{
    if (!c) return make_err(LZ4F_ERROR_GENERIC);
    if (c->tmp_size == 0) return 0;
    if (c->stage != 1) return make_err(LZ4F_ERROR_GENERIC);
    if (dstCapacity < c->tmp_size + 8) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
    const size_t n = c->tmp_size;
    c->tmp_size = 0;
    return encode_staged(c, (uint8_t*)dstBuffer, dstCapacity, n);
}

size_t LZ4F_compressEnd(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t* o)
{
    if (!c) return make_err(LZ4F_ERROR_GENERIC);
    uint8_t* dst = (uint8_t*)dstBuffer;
    const size_t f = LZ4F_flush(c, dstBuffer, dstCapacity, o);
    if (is_err(f)) return f;
    dst += f; dstCapacity -= f;
    if (dstCapacity < 4) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
    st32(dst, 0); dst += 4;
    if (c->prefs.frameInfo.contentChecksumFlag == LZ4F_contentChecksumEnabled) {
        if (dstCapacity < 8) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
        st32(dst, c->xxh.digest()); dst += 4;
    }
    c->stage = 0;
    if (c->prefs.frameInfo.contentSize && c->prefs.frameInfo.contentSize != c->total_in) return make_err(LZ4F_ERROR_frameSize_wrong);
    return (size_t)(dst - (uint8_t*)dstBuffer);
}

}  // extern "C"

// =================================================================================================
// decompression context: the stage machine of LZ4F_decompress
enum DStage {
    ds_getFrameHeader = 0, ds_storeFrameHeader, ds_init, ds_getBlockHeader, ds_storeBlockHeader, ds_copyDirect, ds_getBlockChecksum,
    ds_getCBlock, ds_storeCBlock, ds_flushOut, ds_getSuffix, ds_storeSuffix, ds_getSFrameSize, ds_storeSFrameSize, ds_skipSkippable
};

struct LZ4F_dctx_s {
    LZ4F_frameInfo_t info;
    unsigned version;
    DStage stage;
    uint64_t frame_remaining;
    size_t max_block;
    std::vector<uint8_t> tmp_in;     // block payload (+ checksum) being gathered
    size_t tmp_in_size, tmp_in_target;
    std::vector<uint8_t> tmp_out;    // decoded block waiting to be flushed
    size_t tmp_out_size, tmp_out_start;
    std::vector<uint8_t> hist;       // last <= 64 KiB of output (linked frames): uploaded with each block
    uint8_t header[LZ4F_HEADER_SIZE_MAX + 1];
    Xxh32State xxh, block_xxh;
};

static void dctx_reset(LZ4F_dctx_s* d)
{
    d->stage = ds_getFrameHeader; d->frame_remaining = 0; d->max_block = 0;
    d->tmp_in_size = d->tmp_in_target = 0; d->tmp_out_size = d->tmp_out_start = 0;
    d->hist.clear();
    memset(&d->info, 0, sizeof(d->info));
}

static void note_output(LZ4F_dctx_s* d, const uint8_t* p, size_t n)
{
    if (d->info.contentChecksumFlag) d->xxh.update(p, n);
    if (d->info.contentSize) d->frame_remaining -= n;
    if (d->info.blockMode == LZ4F_blockLinked) {
        if (n >= 65536) d->hist.assign(p + n - 65536, p + n);
        else {
            d->hist.insert(d->hist.end(), p, p + n);
            if (d->hist.size() > 65536) d->hist.erase(d->hist.begin(), d->hist.begin() + (d->hist.size() - 65536));
        }
    }
}

// LZ4F_decodeHeader: src holds a complete header candidate
static size_t decode_header(LZ4F_dctx_s* d, const uint8_t* src, size_t n)
{
    if (n < 7) return make_err(LZ4F_ERROR_frameHeader_incomplete);
    memset(&d->info, 0, sizeof(d->info));
    if ((le32(src) & 0xFFFFFFF0u) == 0x184D2A50u) {
        d->info.frameType = LZ4F_skippableFrame;
        if (src == d->header) { d->tmp_in_size = n; d->tmp_in_target = 8; d->stage = ds_storeSFrameSize; return n; }
        d->stage = ds_getSFrameSize;
        return 4;
    }
    ParsedHeader ph;
    // only the first `needed` bytes are looked at; when fewer are present the stage machine gathers the rest
    if (le32(src) != 0x184D2204u) return make_err(LZ4F_ERROR_frameType_unknown);
    const unsigned flg = src[4];
    if ((flg >> 1) & 1) return make_err(LZ4F_ERROR_reservedFlag_set);
    if (((flg >> 6) & 3) != 1) return make_err(LZ4F_ERROR_headerVersion_wrong);
    const size_t hs = 7 + (((flg >> 3) & 1) ? 8 : 0) + ((flg & 1) ? 4 : 0);
    if (n < hs) {
        if (src != d->header) memcpy(d->header, src, n);
        d->tmp_in_size = n; d->tmp_in_target = hs; d->stage = ds_storeFrameHeader;
        return n;
    }
    const size_t r = parse_frame_header(src, hs, &ph);
    if (is_err(r)) return r;
    d->info = ph.info;
    d->max_block = ph.max_block;
    if (ph.info.contentSize) d->frame_remaining = ph.info.contentSize;
    d->stage = ds_init;
    return hs;
}

extern "C" {

LZ4F_errorCode_t LZ4F_createDecompressionContext(LZ4F_dctx** dctxPtr, unsigned version)
{
    if (!dctxPtr) return make_err(LZ4F_ERROR_GENERIC);
    LZ4F_dctx_s* d = new (std::nothrow) LZ4F_dctx_s();
    if (!d) { *dctxPtr = nullptr; return make_err(LZ4F_ERROR_allocation_failed); }
    d->version = version;
    dctx_reset(d);
    *dctxPtr = d;
    return 0;
}

LZ4F_errorCode_t LZ4F_freeDecompressionContext(LZ4F_dctx* dctx)
{
    LZ4F_errorCode_t r = 0;
    if (dctx) { r = (LZ4F_errorCode_t)dctx->stage; delete dctx; }   // upstream returns the stage as a hint; 0 when idle
    (void)r;
    return 0;
}

void LZ4F_resetDecompressionContext(LZ4F_dctx* dctx) { if (dctx) dctx_reset(dctx); }

size_t LZ4F_headerSize(const void* src, size_t srcSize)
{
    if (!src) return make_err(LZ4F_ERROR_srcPtr_wrong);
    if (srcSize < 5) return make_err(LZ4F_ERROR_frameHeader_incomplete);
    const uint8_t* s = (const uint8_t*)src;
    if ((le32(s) & 0xFFFFFFF0u) == 0x184D2A50u) return 8;
    if (le32(s) != 0x184D2204u) return make_err(LZ4F_ERROR_frameType_unknown);
    const unsigned flg = s[4];
    return 7 + (((flg >> 3) & 1) ? 8 : 0) + ((flg & 1) ? 4 : 0);
}

size_t LZ4F_getFrameInfo(LZ4F_dctx* d, LZ4F_frameInfo_t* frameInfoPtr, const void* srcBuffer, size_t* srcSizePtr)
{
    if (!d || !frameInfoPtr || !srcSizePtr) return make_err(LZ4F_ERROR_GENERIC);
    if (d->stage > ds_storeFrameHeader) {
        size_t o = 0, i = 0;
        *srcSizePtr = 0;
        *frameInfoPtr = d->info;
        return LZ4F_decompress(d, NULL, &o, NULL, &i, NULL);
    }
    if (d->stage == ds_storeFrameHeader) { *srcSizePtr = 0; return make_err(LZ4F_ERROR_frameDecoding_alreadyStarted); }
    const size_t hs = LZ4F_headerSize(srcBuffer, *srcSizePtr);
    if (is_err(hs)) { *srcSizePtr = 0; return hs; }
    if (*srcSizePtr < hs) { *srcSizePtr = 0; return make_err(LZ4F_ERROR_frameHeader_incomplete); }
    size_t r = decode_header(d, (const uint8_t*)srcBuffer, hs);
    if (is_err(r)) *srcSizePtr = 0;
    else { *srcSizePtr = r; r = 4; }
    *frameInfoPtr = d->info;
    return r;
}

size_t LZ4F_decompress(LZ4F_dctx* d, void* dstBuffer, size_t* dstSizePtr, const void* srcBuffer, size_t* srcSizePtr,
                       const LZ4F_decompressOptions_t* /*dOptPtr: stableDst is irrelevant, history is re-staged per block*/)
{
    if (!d || !dstSizePtr || !srcSizePtr) return make_err(LZ4F_ERROR_GENERIC);
    const uint8_t* const srcStart = (const uint8_t*)srcBuffer;
    const uint8_t* const srcEnd = srcStart ? srcStart + *srcSizePtr : srcStart;
    const uint8_t* sp = srcStart;
    uint8_t* const dstStart = (uint8_t*)dstBuffer;
    uint8_t* const dstEnd = dstStart ? dstStart + *dstSizePtr : dstStart;
    uint8_t* dp = dstStart;
    const uint8_t* selected = nullptr;
    size_t hint = 1;
    bool again = true;
    *srcSizePtr = 0; *dstSizePtr = 0;

    while (again) {
        switch (d->stage) {
        case ds_getFrameHeader:
            if ((size_t)(srcEnd - sp) >= LZ4F_HEADER_SIZE_MAX) {
                const size_t hs = decode_header(d, sp, (size_t)(srcEnd - sp));
                if (is_err(hs)) return hs;
                sp += hs;
                break;
            }
            d->tmp_in_size = 0;
            if (srcEnd - sp == 0) return LZ4F_HEADER_SIZE_MIN;
            d->tmp_in_target = LZ4F_HEADER_SIZE_MIN;
            d->stage = ds_storeFrameHeader;
            /* fall through */
        case ds_storeFrameHeader: {
            const size_t n = std::min(d->tmp_in_target - d->tmp_in_size, (size_t)(srcEnd - sp));
            memcpy(d->header + d->tmp_in_size, sp, n);
            d->tmp_in_size += n; sp += n;
            if (d->tmp_in_size < d->tmp_in_target) { hint = (d->tmp_in_target - d->tmp_in_size) + 4; again = false; break; }
            const size_t hs = decode_header(d, d->header, d->tmp_in_target);
            if (is_err(hs)) return hs;
            break;
        }
        case ds_init:
            if (d->info.contentChecksumFlag) d->xxh.reset(0);
            d->tmp_in.resize(d->max_block + 4);
            d->tmp_out.resize(d->max_block);
            d->tmp_in_size = d->tmp_in_target = 0; d->tmp_out_size = d->tmp_out_start = 0;
            d->hist.clear();
            d->stage = ds_getBlockHeader;
            /* fall through */
        case ds_getBlockHeader:
            if ((size_t)(srcEnd - sp) >= 4) { selected = sp; sp += 4; }
            else { d->tmp_in_size = 0; d->stage = ds_storeBlockHeader; }
            if (d->stage == ds_storeBlockHeader)
        case ds_storeBlockHeader: {
                const size_t n = std::min((size_t)4 - d->tmp_in_size, (size_t)(srcEnd - sp));
                memcpy(d->tmp_in.data() + d->tmp_in_size, sp, n);
                sp += n; d->tmp_in_size += n;
                if (d->tmp_in_size < 4) { hint = 4 - d->tmp_in_size; again = false; break; }
                selected = d->tmp_in.data();
            }
            {
                const uint32_t bh = le32(selected);
                const size_t csz = bh & 0x7FFFFFFFu;
                const size_t crc = d->info.blockChecksumFlag ? 4 : 0;
                if (bh == 0) { d->stage = ds_getSuffix; break; }
                if (csz > d->max_block) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
                if (bh & 0x80000000u) {
                    d->tmp_in_target = csz;
                    if (d->info.blockChecksumFlag) d->block_xxh.reset(0);
                    d->stage = ds_copyDirect;
                    break;
                }
                d->tmp_in_target = csz + crc;
                d->stage = ds_getCBlock;
                if (dp == dstEnd || sp == srcEnd) { hint = 4 + csz + crc; again = false; }
                break;
            }
        case ds_copyDirect: {
            // stored block: bytes pass straight through (no arithmetic to offload); the running block
            // checksum of a stored block is kept on the host like the content checksum
            const size_t room = std::min((size_t)(srcEnd - sp), (size_t)(dstEnd - dp));
            const size_t n = std::min(d->tmp_in_target, room);
            if (n) memcpy(dp, sp, n);
            if (d->info.blockChecksumFlag) d->block_xxh.update(sp, n);
            note_output(d, sp, n);
            sp += n; dp += n;
            if (n == d->tmp_in_target) {
                if (d->info.blockChecksumFlag) { d->tmp_in_size = 0; d->stage = ds_getBlockChecksum; }
                else d->stage = ds_getBlockHeader;
                break;
            }
            d->tmp_in_target -= n;
            hint = d->tmp_in_target + (d->info.blockChecksumFlag ? 4 : 0) + 4;
            again = false;
            break;
        }
        case ds_getBlockChecksum: {
            const uint8_t* crcSrc;
            if (srcEnd - sp >= 4 && d->tmp_in_size == 0) { crcSrc = sp; sp += 4; }
            else {
                const size_t n = std::min((size_t)4 - d->tmp_in_size, (size_t)(srcEnd - sp));
                memcpy(d->header + d->tmp_in_size, sp, n);
                d->tmp_in_size += n; sp += n;
                if (d->tmp_in_size < 4) { again = false; break; }
                crcSrc = d->header;
            }
            if (le32(crcSrc) != d->block_xxh.digest()) return make_err(LZ4F_ERROR_blockChecksum_invalid);
            d->stage = ds_getBlockHeader;
            break;
        }
        case ds_getCBlock:
            if ((size_t)(srcEnd - sp) < d->tmp_in_target) { d->tmp_in_size = 0; d->stage = ds_storeCBlock; break; }
            selected = sp; sp += d->tmp_in_target;
            if (0)
        case ds_storeCBlock: {
                const size_t n = std::min(d->tmp_in_target - d->tmp_in_size, (size_t)(srcEnd - sp));
                memcpy(d->tmp_in.data() + d->tmp_in_size, sp, n);
                d->tmp_in_size += n; sp += n;
                if (d->tmp_in_size < d->tmp_in_target) {
                    hint = (d->tmp_in_target - d->tmp_in_size) + (d->info.blockChecksumFlag ? 4 : 0) + 4;
                    again = false;
                    break;
                }
                selected = d->tmp_in.data();
            }
            {
                // a whole compressed block is available: checksum verification + decode run on the GPU
                size_t csz = d->tmp_in_target;
                const bool bck = d->info.blockChecksumFlag != 0;
                if (bck) csz -= 4;
                const bool direct = (size_t)(dstEnd - dp) >= d->max_block;
                uint8_t* out = direct ? dp : d->tmp_out.data();
                EngineLease eng;
                size_t r = eng.get();
                if (is_err(r)) return r;
                uint32_t got = 0;
                r = eng->decompress_block_host(selected, (uint32_t)csz, bck, d->hist.data(), d->hist.size(), out, (uint32_t)d->max_block,
                                               d->info.blockMode == LZ4F_blockLinked, (uint32_t)d->max_block, &got);
                if (is_err(r)) {
                    if (r == make_err(LZ4F_ERROR_GENERIC) && !direct) return make_err(LZ4F_ERROR_decompressionFailed);
                    return r;
                }
                note_output(d, out, got);
                if (direct) { dp += got; d->stage = ds_getBlockHeader; break; }
                d->tmp_out_size = got; d->tmp_out_start = 0;
                d->stage = ds_flushOut;
            }
            /* fall through */
        case ds_flushOut:
            if (dp != nullptr) {
                const size_t n = std::min(d->tmp_out_size - d->tmp_out_start, (size_t)(dstEnd - dp));
                memcpy(dp, d->tmp_out.data() + d->tmp_out_start, n);
                d->tmp_out_start += n; dp += n;
            }
            if (d->tmp_out_start == d->tmp_out_size) { d->stage = ds_getBlockHeader; break; }
            again = false; hint = 4;
            break;
        case ds_getSuffix:
            if (d->frame_remaining) return make_err(LZ4F_ERROR_frameSize_wrong);
            if (!d->info.contentChecksumFlag) { hint = 0; dctx_reset(d); again = false; break; }
            if (srcEnd - sp < 4) { d->tmp_in_size = 0; d->stage = ds_storeSuffix; }
            else { selected = sp; sp += 4; }
            if (d->stage == ds_storeSuffix)
        case ds_storeSuffix: {
                const size_t n = std::min((size_t)4 - d->tmp_in_size, (size_t)(srcEnd - sp));
                memcpy(d->tmp_in.data() + d->tmp_in_size, sp, n);
                sp += n; d->tmp_in_size += n;
                if (d->tmp_in_size < 4) { hint = 4 - d->tmp_in_size; again = false; break; }
                selected = d->tmp_in.data();
            }
            if (le32(selected) != d->xxh.digest()) return make_err(LZ4F_ERROR_contentChecksum_invalid);
            hint = 0; dctx_reset(d); again = false;
            break;
        case ds_getSFrameSize:
            if (srcEnd - sp >= 4) { selected = sp; sp += 4; }
            else { d->tmp_in_size = 4; d->tmp_in_target = 8; d->stage = ds_storeSFrameSize; }
            if (d->stage == ds_storeSFrameSize)
        case ds_storeSFrameSize: {
                const size_t n = std::min(d->tmp_in_target - d->tmp_in_size, (size_t)(srcEnd - sp));
                memcpy(d->header + d->tmp_in_size, sp, n);
                sp += n; d->tmp_in_size += n;
                if (d->tmp_in_size < d->tmp_in_target) { hint = d->tmp_in_target - d->tmp_in_size; again = false; break; }
                selected = d->header + 4;
            }
            d->info.contentSize = le32(selected);
            d->tmp_in_target = le32(selected);
            d->stage = ds_skipSkippable;
            break;
        case ds_skipSkippable: {
            const size_t n = std::min(d->tmp_in_target, (size_t)(srcEnd - sp));
            sp += n; d->tmp_in_target -= n;
            again = false; hint = d->tmp_in_target;
            if (hint) break;
            dctx_reset(d);
            break;
        }
        }
    }
    *srcSizePtr = (size_t)(sp - srcStart);
    *dstSizePtr = (size_t)(dp - dstStart);
    return hint;
}

// ---- the two C finalizers of the reference (Conduit.hsc:163-189, :539-553) ----
void haskell_lz4_freeCompressionContext(LZ4F_cctx** ctxPtr)
{
    LZ4F_cctx* ctx = *ctxPtr;
    if (ctx != NULL) {
        size_t err = LZ4F_freeCompressionContext(ctx);
        if (LZ4F_isError(err)) { fprintf(stderr, "LZ4F_freeCompressionContext failed: %s\n", LZ4F_getErrorName(err)); exit(1); }
    }
}
void haskell_lz4_freeDecompressionContext(LZ4F_dctx** ctxPtr)
{
    LZ4F_dctx* ctx = *ctxPtr;
    if (ctx != NULL) {
        size_t err = LZ4F_freeDecompressionContext(ctx);
        if (LZ4F_isError(err)) { fprintf(stderr, "LZ4F_freeDecompressionContext failed: %s\n", LZ4F_getErrorName(err)); exit(1); }
    }
}

// ---- prefixed aliases ----
unsigned lz4f_mi355x_isError(size_t c) { return LZ4F_isError(c); }
const char* lz4f_mi355x_getErrorName(size_t c) { return LZ4F_getErrorName(c); }
size_t lz4f_mi355x_createCompressionContext(LZ4F_cctx** p, unsigned v) { return LZ4F_createCompressionContext(p, v); }
size_t lz4f_mi355x_freeCompressionContext(LZ4F_cctx* c) { return LZ4F_freeCompressionContext(c); }
size_t lz4f_mi355x_compressBegin(LZ4F_cctx* c, void* d, size_t n, const LZ4F_preferences_t* p) { return LZ4F_compressBegin(c, d, n, p); }
size_t lz4f_mi355x_compressBound(size_t n, const LZ4F_preferences_t* p) { return LZ4F_compressBound(n, p); }
size_t lz4f_mi355x_compressUpdate(LZ4F_cctx* c, void* d, size_t dn, const void* s, size_t sn, const LZ4F_compressOptions_t* o) { return LZ4F_compressUpdate(c, d, dn, s, sn, o); }
size_t lz4f_mi355x_flush(LZ4F_cctx* c, void* d, size_t n, const LZ4F_compressOptions_t* o) { return LZ4F_flush(c, d, n, o); }
size_t lz4f_mi355x_compressEnd(LZ4F_cctx* c, void* d, size_t n, const LZ4F_compressOptions_t* o) { return LZ4F_compressEnd(c, d, n, o); }
size_t lz4f_mi355x_createDecompressionContext(LZ4F_dctx** p, unsigned v) { return LZ4F_createDecompressionContext(p, v); }
size_t lz4f_mi355x_freeDecompressionContext(LZ4F_dctx* c) { return LZ4F_freeDecompressionContext(c); }
size_t lz4f_mi355x_getFrameInfo(LZ4F_dctx* c, LZ4F_frameInfo_t* f, const void* s, size_t* n) { return LZ4F_getFrameInfo(c, f, s, n); }
size_t lz4f_mi355x_decompress(LZ4F_dctx* c, void* d, size_t* dn, const void* s, size_t* sn, const LZ4F_decompressOptions_t* o) { return LZ4F_decompress(c, d, dn, s, sn, o); }

// =================================================================================================
// host-pointer bulk calls
size_t lz4f_mi355x_compressFrameBound(size_t srcSize, const LZ4F_preferences_t* prefs)
{
    LZ4F_preferences_t p; memset(&p, 0, sizeof(p));
    if (prefs) p = *prefs;
    const size_t bs = block_size_of(p.frameInfo.blockSizeID);
    if (!bs) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
    const size_t nblocks = srcSize / bs + 1;
    return LZ4F_HEADER_SIZE_MAX + srcSize + nblocks * (4 + (p.frameInfo.blockChecksumFlag ? 4 : 0)) + 4 + (p.frameInfo.contentChecksumFlag ? 4 : 0);
}

size_t lz4f_mi355x_compressFrame(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const LZ4F_preferences_t* prefs)
{
    // header / all blocks through the slab pipeline (pipeline.hip: several slabs in flight, over lz4f_mi355x_use_devices() GPUs) /
    // EndMark / content checksum.  Same bytes as the streaming API gives for the same input fed in whole blocks.
    LZ4F_preferences_t p; memset(&p, 0, sizeof(p));
    if (prefs) p = *prefs;
    if (p.compressionLevel > 2) { set_last_error("compressionLevel %d: only the fast encoder exists in liblz4f_mi355x", p.compressionLevel); return make_err(LZ4F_ERROR_compressionLevel_invalid); }
    if (p.frameInfo.blockSizeID == 0) p.frameInfo.blockSizeID = LZ4F_max64KB;
    const size_t bs = block_size_of(p.frameInfo.blockSizeID);
    if (!bs) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
    if (dstCapacity < LZ4F_HEADER_SIZE_MAX) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
    if (p.frameInfo.contentSize && p.frameInfo.contentSize != srcSize) return make_err(LZ4F_ERROR_frameSize_wrong);
    uint8_t* d = (uint8_t*)dst;
    size_t used = write_frame_header(d, p);
    // the content checksum is one serial chain over the input (~6 GB/s on a core): on a thread of its own, beside the transfers
    uint32_t cck = 0;
    std::thread hasher;
    const bool want_cck = p.frameInfo.contentChecksumFlag == LZ4F_contentChecksumEnabled;
    if (want_cck) hasher = std::thread([&] { cck = xxh32_host(src, srcSize); });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } join_hasher{hasher};      // (also when the pipeline throws: bad_alloc, thread creation)
    size_t written = 0;
    size_t r;
    try {
        r = pipe_compress_blocks((const uint8_t*)src, srcSize, (uint32_t)bs, p.frameInfo.blockMode == LZ4F_blockLinked, p.frameInfo.blockChecksumFlag != 0,
                                 d + used, dstCapacity - used, &written);
    } catch (const std::exception& e) { set_last_error("compressFrame: %s", e.what()); r = make_err(LZ4F_ERROR_allocation_failed); }
    if (want_cck) hasher.join();
    if (is_err(r)) return r;
    used += written;
    if (dstCapacity - used < (size_t)(want_cck ? 8 : 4)) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
    st32(d + used, 0); used += 4;
    if (want_cck) { st32(d + used, cck); used += 4; }
    return used;
}

// ---- the block list of a finished frame in host memory (ours or any other encoder's), as the trailer the device decoder looks for ----
// (host work only: one 4-byte read per block)
static size_t walk_for_block_list(const uint8_t* s, size_t n, BlockList* bl)
{
    if (n < 7) return make_err(LZ4F_ERROR_frameHeader_incomplete);
    ParsedHeader ph;
    const size_t hs = parse_frame_header(s, n, &ph);                    // (a skippable frame is frameType_unknown here: nothing to list)
    if (is_err(hs)) return hs;
    const size_t crc = ph.info.blockChecksumFlag ? 4 : 0;
    size_t pos = ph.header_size;
    for (;;) {
        if (n - pos < 4) return make_err(LZ4F_ERROR_frameHeader_incomplete);
        const uint32_t w = le32(s + pos);
        if (w == 0) break;
        const size_t csz = w & 0x7FFFFFFFu;
        if (csz > ph.max_block) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
        if (n - pos - 4 < csz + crc) return make_err(LZ4F_ERROR_frameHeader_incomplete);
        bl->at.push_back(pos);
        pos += 4 + csz + crc;
    }
    pos += 4;
    if (ph.info.contentChecksumFlag) { if (n - pos < 4) return make_err(LZ4F_ERROR_frameHeader_incomplete); pos += 4; }
    if (pos != n) { set_last_error("block list: the frame ends at %zu, not at frameSize %zu", pos, n); return make_err(LZ4F_ERROR_frameSize_wrong); }
    return 0;
}
size_t lz4f_mi355x_blockListSize(const void* frame, size_t frameSize)
{
    try {
        BlockList bl;
        const size_t r = walk_for_block_list((const uint8_t*)frame, frameSize, &bl);
        if (is_err(r)) return r;
        return host_trailer_size(frameSize, bl.at.size());
    } catch (const std::exception& e) { set_last_error("blockListSize: %s", e.what()); return make_err(LZ4F_ERROR_allocation_failed); }
}
size_t lz4f_mi355x_appendBlockList(void* buf, size_t frameSize, size_t capacity)
{
    try {
        if (capacity < frameSize) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
        BlockList bl;
        size_t r = walk_for_block_list((const uint8_t*)buf, frameSize, &bl);
        if (is_err(r)) return r;
        r = host_trailer_size(frameSize, bl.at.size());
        if (is_err(r)) return r;
        if (r == 0) return frameSize;                                   // no block, no list
        if (capacity - frameSize < r) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
        host_write_trailer((uint8_t*)buf + frameSize, frameSize, bl.at.data(), (uint32_t)bl.at.size());
        return frameSize + r;
    } catch (const std::exception& e) { set_last_error("appendBlockList: %s", e.what()); return make_err(LZ4F_ERROR_allocation_failed); }
}

static size_t decompress_frame_common(void* dst, size_t dstCapacity, const void* src, size_t srcSize, size_t* srcConsumed,
                                      const std::function<void(const uint8_t*, size_t)>* sink)
{
    // Host walk of the size words (one 4-byte read per block), then slabs of blocks through the pipeline:
    // payloads up, table up, decode, output down - several slabs in flight.
    const uint8_t* s = (const uint8_t*)src;
    if (srcConsumed) *srcConsumed = 0;
    if (srcSize < 7) return make_err(LZ4F_ERROR_frameHeader_incomplete);
    if ((le32(s) & 0xFFFFFFF0u) == 0x184D2A50u) {
        if (srcSize < 8 || srcSize < 8 + (size_t)le32(s + 4)) return make_err(LZ4F_ERROR_frameHeader_incomplete);
        if (srcConsumed) *srcConsumed = 8 + (size_t)le32(s + 4);
        return 0;
    }
    ParsedHeader ph;
    size_t hs = parse_frame_header(s, srcSize, &ph);
    if (is_err(hs)) return hs;
    size_t decoded = 0, consumed = 0;
    size_t r = pipe_decompress_frame(s, srcSize, ph, sink ? nullptr : (uint8_t*)dst, dstCapacity, sink, &decoded, &consumed);
    if (is_err(r)) return r;
    if (srcConsumed) *srcConsumed = consumed;
    return decoded;
}

size_t lz4f_mi355x_decompressFrame(void* dst, size_t dstCapacity, const void* src, size_t srcSize, size_t* srcConsumed)
{
    try { return decompress_frame_common(dst, dstCapacity, src, srcSize, srcConsumed, nullptr); }
    catch (const std::exception& e) { set_last_error("decompressFrame: %s", e.what()); return make_err(LZ4F_ERROR_allocation_failed); }      // (nothing may unwind through the C boundary)
}

size_t lz4f_mi355x_decompressFrameTo(lz4f_mi355x_yield_fn yield, void* user, const void* src, size_t srcSize, size_t* srcConsumed)
{
    if (!yield) return make_err(LZ4F_ERROR_GENERIC);
    const std::function<void(const uint8_t*, size_t)> sink = [&](const uint8_t* p, size_t n) { yield(user, p, n); };
    // (a C++ caller's yield may throw - conduit.cpp's does: the pipeline joins its workers first, then the exception goes back to that caller;
    // a C or Haskell caller's callback cannot throw)
    return decompress_frame_common(nullptr, 0, src, srcSize, srcConsumed, &sink);
}

// ---- a frame decoded batch by batch (bounded memory): header once, runs of whole blocks as they arrive, the closing words at the end ----
struct lz4f_mi355x_fdec { ParsedHeader ph; FrameCarry carry; bool ended = false; };

size_t lz4f_mi355x_fdec_create(lz4f_mi355x_fdec** out, const void* header, size_t n, LZ4F_frameInfo_t* info)
{
    if (!out || !header) return make_err(LZ4F_ERROR_GENERIC);
    *out = nullptr;
    lz4f_mi355x_fdec* d = nullptr;
    try { d = new lz4f_mi355x_fdec(); } catch (...) { return make_err(LZ4F_ERROR_allocation_failed); }
    const size_t hs = parse_frame_header((const uint8_t*)header, n, &d->ph);
    if (is_err(hs)) { delete d; return hs; }
    if (info) *info = d->ph.info;
    *out = d;
    return d->ph.header_size;
}
size_t lz4f_mi355x_fdec_blocks(lz4f_mi355x_fdec* d, lz4f_mi355x_yield_fn yield, void* user, const void* blocks, size_t n)
{
    if (!d || !yield || d->ended) return make_err(LZ4F_ERROR_GENERIC);
    if (n == 0) return 0;
    const std::function<void(const uint8_t*, size_t)> sink = [&](const uint8_t* p, size_t k) { yield(user, p, k); };
    size_t decoded = 0, consumed = 0;
    const size_t r = pipe_decompress_frame((const uint8_t*)blocks, n, d->ph, nullptr, 0, &sink, &decoded, &consumed, &d->carry);
    if (is_err(r)) return r;
    if (consumed != n) { set_last_error("fdec_blocks: the batch is not a run of whole blocks (%zu of %zu bytes)", consumed, n); return make_err(LZ4F_ERROR_GENERIC); }
    return decoded;
}
size_t lz4f_mi355x_fdec_end(lz4f_mi355x_fdec* d, const void* tail, size_t n)
{
    if (!d || d->ended) return make_err(LZ4F_ERROR_GENERIC);
    const uint8_t* t = (const uint8_t*)tail;
    const size_t need = 4 + (d->ph.info.contentChecksumFlag ? 4 : 0);
    if (!t || n < need) return make_err(LZ4F_ERROR_frameHeader_incomplete);
    if (le32(t) != 0) return make_err(LZ4F_ERROR_GENERIC);                        // not an EndMark
    d->ended = true;
    if (d->ph.info.contentSize && d->ph.info.contentSize != d->carry.out_total) return make_err(LZ4F_ERROR_frameSize_wrong);
    if (d->ph.info.contentChecksumFlag && le32(t + 4) != d->carry.cck.digest()) return make_err(LZ4F_ERROR_contentChecksum_invalid);
    return need;
}
void lz4f_mi355x_fdec_free(lz4f_mi355x_fdec* d) { delete d; }

size_t lz4f_mi355x_use_devices(int count)
{
    const int have = logical_devices();       // (the visible ones; more only under the test switch LZ4F_MI355X_LOGICAL_DEVICES)
    if (count < 1 || count > have) { set_last_error("use_devices(%d): %d devices are visible", count, have); return make_err(LZ4F_ERROR_GENERIC); }
    set_bulk_devices(count);
    return 0;
}

}  // extern "C"
