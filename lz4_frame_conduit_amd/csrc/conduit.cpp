// conduit.cpp -- C++ mirror of the reference's stream transformers (host side above the C ABI).
//
// The reference's host is Haskell (Codec.Compression.LZ4.Conduit); GHC is not in this image, so the
// same drivers are written here in C++ with the same names, argument meaning, call sequences and
// error behaviour, on top of the same twelve LZ4F_* entry points.  A ConduitT ByteString ByteString
// becomes a pair of callbacks: `await` hands over the next input ByteString, `yield` receives each
// output ByteString.  Citations: /root/reference/src/Codec/Compression/LZ4/Conduit.hsc.
#include "conduit.hpp"

#include <string.h>

#include <algorithm>
#include <stdexcept>

#include "engine.hpp"

namespace lz4f {
namespace conduit {

// handleLz4Error (Conduit.hsc:145-160): a second, non-blocking call classifies the result
static size_t handleLz4Error(size_t ret)
{
    if (LZ4F_isError(ret)) throw std::runtime_error(std::string("lz4frame error: ") + LZ4F_getErrorName(ret));
    return ret;
}

// lz4DefaultPreferences (Conduit.hsc:248-263): 64 KiB linked blocks, no checksums, level 0, no autoFlush
LZ4F_preferences_t lz4DefaultPreferences()
{
    LZ4F_preferences_t p; memset(&p, 0, sizeof(p));
    p.frameInfo.blockSizeID = LZ4F_default; p.frameInfo.blockMode = LZ4F_blockLinked;
    p.frameInfo.contentChecksumFlag = LZ4F_noContentChecksum; p.frameInfo.frameType = LZ4F_frame;
    p.frameInfo.blockChecksumFlag = LZ4F_noBlockChecksum;
    return p;
}

// bsChunksOf (Conduit.hsc:428-433)
std::vector<Slice> bsChunksOf(size_t chunkSize, Slice bs)
{
    if (chunkSize < 1) throw std::invalid_argument("bsChunksOf: chunkSize < 1: " + std::to_string(chunkSize));
    std::vector<Slice> out;
    while (bs.size > chunkSize) { out.push_back(Slice{bs.data, chunkSize}); bs.data += chunkSize; bs.size -= chunkSize; }
    out.push_back(bs);
    return out;
}

namespace {
// withLz4CtxAndPrefsConduit (Conduit.hsc:340-354): scoped context + preferences
struct ScopedCctx {
    LZ4F_cctx* ctx = nullptr;
    LZ4F_preferences_t prefs;
    explicit ScopedCctx(const LZ4F_preferences_t* p)
    {
        prefs = p ? *p : lz4DefaultPreferences();
        handleLz4Error(LZ4F_createCompressionContext(&ctx, LZ4F_VERSION));          // Conduit.hsc:194-203
    }
    ~ScopedCctx() { LZ4F_freeCompressionContext(ctx); }                             // Conduit.hsc:206-212
};
struct ScopedDctx {
    LZ4F_dctx* ctx = nullptr;
    ScopedDctx() { handleLz4Error(LZ4F_createDecompressionContext(&ctx, LZ4F_VERSION)); }   // Conduit.hsc:559-571
    ~ScopedDctx() { LZ4F_freeDecompressionContext(ctx); }                                  // Conduit.hsc:701
};
}  // namespace

// compressYieldImmediately (Conduit.hsc:364-425)
void compressYieldImmediately(const LZ4F_preferences_t* prefsIn, const Await& await, const Yield& yield)
{
    ScopedCctx s(prefsIn);
    std::vector<uint8_t> buf(LZ4F_HEADER_SIZE_MAX);
    const size_t headerSize = handleLz4Error(LZ4F_compressBegin(s.ctx, buf.data(), LZ4F_HEADER_SIZE_MAX, &s.prefs));
    yield(Slice{buf.data(), headerSize});
    Slice bs;
    while (await(bs)) {
        const size_t size = handleLz4Error(LZ4F_compressBound(bs.size, &s.prefs));
        buf.resize(size);
        const size_t written = handleLz4Error(LZ4F_compressUpdate(s.ctx, buf.data(), size, bs.data, bs.size, NULL));
        if (written != 0) yield(Slice{buf.data(), written});
    }
    const size_t footerSize = handleLz4Error(LZ4F_compressBound(0, &s.prefs));
    buf.resize(footerSize);
    const size_t footerWritten = handleLz4Error(LZ4F_compressEnd(s.ctx, buf.data(), footerSize, NULL));
    yield(Slice{buf.data(), footerWritten});
}

// compressWithOutBufferSize (Conduit.hsc:457-533), with the preferences made a parameter
// (the reference can only pass lz4DefaultPreferences, Conduit.hsc:347)
void compressWithOutBufferSize(size_t bufferSize, const LZ4F_preferences_t* prefsIn, const Await& await, const Yield& yield)
{
    ScopedCctx s(prefsIn);
    const size_t bsInChunkSize = 16 * 1024;                                                        // :464
    const size_t compressBound = handleLz4Error(LZ4F_compressBound(LZ4F_HEADER_SIZE_MAX + bsInChunkSize, &s.prefs));   // :466
    const size_t outBufferSize = std::max(bufferSize, compressBound);                              // :467
    std::vector<uint8_t> outBuf(outBufferSize);
    auto yieldOutBuf = [&](size_t len) { yield(Slice{outBuf.data(), len}); };                        // :471-473
    const size_t headerSize = handleLz4Error(LZ4F_compressBegin(s.ctx, outBuf.data(), outBufferSize, &s.prefs));      // :475
    size_t remainingCapacity = outBufferSize - headerSize;                                         // :533

    auto compressSingleBsFitting = [&](Slice bs) {                                                  // :516-531
        if (remainingCapacity < compressBound) throw std::logic_error("precondition violated");
        const size_t offset = outBufferSize - remainingCapacity;
        const size_t written = handleLz4Error(LZ4F_compressUpdate(s.ctx, outBuf.data() + offset, remainingCapacity, bs.data, bs.size, NULL));
        if (written > remainingCapacity) throw std::logic_error("lz4fCompressUpdate wrote past buffer");
        remainingCapacity -= written;
    };
    Slice bs;
    while (await(bs)) {                                                                            // :500-503
        for (const Slice& piece : bsChunksOf(bsInChunkSize, bs)) {
            if (remainingCapacity < compressBound) {                                               // :507-512
                yieldOutBuf(outBufferSize - remainingCapacity);
                remainingCapacity = outBufferSize;
            }
            compressSingleBsFitting(piece);
        }
    }
    const size_t footerSize = handleLz4Error(LZ4F_compressBound(0, &s.prefs));                     // :490
    auto writeFooterAndYield = [&]() {                                                              // :477-481
        const size_t offset = outBufferSize - remainingCapacity;
        const size_t footerWritten = handleLz4Error(LZ4F_compressEnd(s.ctx, outBuf.data() + offset, remainingCapacity, NULL));
        yieldOutBuf(outBufferSize - remainingCapacity + footerWritten);
    };
    if (remainingCapacity >= footerSize) writeFooterAndYield();                                    // :492-498
    else { yieldOutBuf(outBufferSize - remainingCapacity); remainingCapacity = outBufferSize; writeFooterAndYield(); }
}

void compress(const Await& await, const Yield& yield) { compressWithOutBufferSize(0, nullptr, await, yield); }   // :336-337
void compressWithPreferences(const LZ4F_preferences_t& prefs, const Await& await, const Yield& yield)
{
    compressWithOutBufferSize(0, &prefs, await, yield);
}

namespace {
// conduit leftovers: CB.take n pulls exactly n bytes and pushes the rest of the chunk back
struct Upstream {
    const Await& await;
    Slice left{nullptr, 0};
    std::vector<uint8_t> keep;
    bool next(Slice& out)
    {
        if (left.size) { out = left; left = Slice{nullptr, 0}; return true; }
        return await(out);
    }
    std::vector<uint8_t> take(size_t n)
    {
        std::vector<uint8_t> got;
        Slice bs;
        while (got.size() < n && next(bs)) {
            const size_t k = std::min(n - got.size(), bs.size);
            got.insert(got.end(), bs.data, bs.data + k);
            if (k < bs.size) { left = Slice{bs.data + k, bs.size - k}; }
        }
        return got;
    }
};
}  // namespace

// decompress (Conduit.hsc:598-701)
void decompress(const Await& await, const Yield& yield)
{
    ScopedDctx s;
    Upstream up{await};
    std::vector<uint8_t> header = up.take(5);                                                      // :614
    if (header.size() != 5)
        throw std::runtime_error("lz4 decompress error: not enough bytes for header; expected 5, got " + std::to_string(header.size()));
    const uint8_t byteFLG = header[4];
    const bool contentSizeBit = (byteFLG >> 3) & 1;                                                // :618-619
    const size_t numRemainingHeaderBytes = contentSizeBit ? 2 + 8 : 2;                             // :621-623 (ignores the dictID bit, as the reference does)
    // the leftover slice may point into a caller buffer that stays valid until the next await
    {
        std::vector<uint8_t> rest = up.take(numRemainingHeaderBytes);
        header.insert(header.end(), rest.begin(), rest.end());
    }
    LZ4F_frameInfo_t frameInfo;
    size_t headerLen = header.size();
    size_t hint = handleLz4Error(LZ4F_getFrameInfo(s.ctx, &frameInfo, header.data(), &headerLen));  // :629-632

    const size_t dstBufferSizeDefault = 16 * 1024;                                                 // :634-635
    std::vector<uint8_t> dstBuffer(dstBufferSizeDefault);
    auto loopSingleBs = [&](Slice bs) {                                                            // :661-685
        for (;;) {
            const size_t outBufSize = std::max(hint, dstBufferSizeDefault);                        // :665
            if (outBufSize > dstBuffer.size()) dstBuffer.resize(outBufSize);                       // :652-659
            size_t dstSize = outBufSize, srcSize = bs.size;
            hint = handleLz4Error(LZ4F_decompress(s.ctx, dstBuffer.data(), &dstSize, bs.data, &srcSize, NULL));   // :673
            yield(Slice{dstBuffer.data(), dstSize});                                               // :679 (may be empty)
            if (srcSize < bs.size) { bs.data += srcSize; bs.size -= srcSize; continue; }           // :683
            if (srcSize == bs.size) return;
            throw std::logic_error("lz4 decompress: assertion failed: srcRead < BS.length bs");
        }
    };
    for (;;) {                                                                                     // :687-697
        Slice bs;
        if (!up.next(bs)) throw std::runtime_error("lz4 decompress error: stream ended before EndMark");
        loopSingleBs(bs);
        if (hint == 0) break;
    }
}

// ---- batched variants (new) ---------------------------------------------
// compressBatched: a frame like compressWithPreferences makes, but the input is gathered - in page-locked memory, which the
// GPU's DMA engines read directly - until `batchBytes` are there, and every batch's whole blocks go through the bulk path
// (pipeline.hip: slabs of blocks in flight, over lz4f_mi355x_use_devices() GPUs).  A linked frame's batches keep the
// 64 KiB in front of them.
namespace {
struct Pinned {
    uint8_t* p = nullptr; size_t cap = 0;
    void ensure(size_t n) { if (n <= cap) return; uint8_t* q = (uint8_t*)lz4f_mi355x_host_alloc(n); if (!q) throw std::runtime_error(std::string("lz4frame error: ") + lz4f_mi355x_last_error()); if (p) { /* caller copies what it needs first */ lz4f_mi355x_host_free(p); } p = q; cap = n; }
    ~Pinned() { lz4f_mi355x_host_free(p); }
};
}
void compressBatched(size_t batchBytes, const LZ4F_preferences_t* prefsIn, const Await& await, const Yield& yield, bool blockList)
{
    LZ4F_preferences_t prefs; memset(&prefs, 0, sizeof(prefs));
    if (prefsIn) prefs = *prefsIn;
    if (prefs.compressionLevel > 2) handleLz4Error(make_err(LZ4F_ERROR_compressionLevel_invalid));
    if (prefs.frameInfo.blockSizeID == 0) prefs.frameInfo.blockSizeID = LZ4F_max64KB;
    const size_t bs = block_size_of(prefs.frameInfo.blockSizeID);
    if (!bs) handleLz4Error(make_err(LZ4F_ERROR_maxBlockSize_invalid));
    const bool linked = prefs.frameInfo.blockMode == LZ4F_blockLinked, bck = prefs.frameInfo.blockChecksumFlag != 0;
    const bool cck = prefs.frameInfo.contentChecksumFlag == LZ4F_contentChecksumEnabled;
    uint8_t hdr[LZ4F_HEADER_SIZE_MAX];
    const size_t headerSize = write_frame_header(hdr, prefs);
    yield(Slice{hdr, headerSize});
    if (lz4f_mi355x_device_count() <= 0) { set_last_error("no usable HIP device: liblz4f_mi355x has no CPU fallback"); handleLz4Error(make_err(LZ4F_ERROR_GENERIC)); }
    if (batchBytes < bs) batchBytes = bs;
    const size_t HIST = 65536;
    Pinned in, out;
    in.ensure(HIST + batchBytes + bs + (1u << 20));              // [64 KiB of the batch before][this batch ...]
    size_t fill = 0, hist = 0;                                  // bytes gathered behind in.p + HIST; valid history in front of them
    uint64_t total = 0;
    uint64_t framePos = headerSize;                              // where the next batch's blocks land in the frame
    BlockList bl;                                                // blockList: every block's size word, for the trailer behind the frame
    Xxh32State xxh; xxh.reset(0);
    auto flushBatch = [&](bool last) {
        const size_t n = last ? fill : (fill / bs) * bs;         // whole blocks; at the end also the short one
        if (!n) return;
        const size_t cap = n + (n / bs + 2) * 8 + 64;
        out.ensure(cap);
        size_t w = 0;
        handleLz4Error(pipe_compress_blocks(in.p + HIST, n, (uint32_t)bs, linked, bck, out.p, out.cap, &w, hist));
        if (blockList && !bl.add_blocks(out.p, w, framePos, bck)) handleLz4Error(make_err(LZ4F_ERROR_GENERIC));
        framePos += w;
        if (w) yield(Slice{out.p, w});
        if (linked) { const size_t keep = std::min(HIST, hist + n); memmove(in.p + HIST - keep, in.p + HIST + n - keep, keep); hist = keep; }
        memmove(in.p + HIST, in.p + HIST + n, fill - n);
        fill -= n;
    };
    Slice bsl;
    while (await(bsl)) {
        const uint8_t* q = bsl.data; size_t left = bsl.size;
        if (cck) xxh.update(q, left);
        total += left;
        while (left) {
            const size_t room = in.cap - HIST - fill;
            const size_t take = std::min(room, left);
            memcpy(in.p + HIST + fill, q, take);
            fill += take; q += take; left -= take;
            if (fill >= batchBytes) flushBatch(false);
        }
    }
    flushBatch(true);
    uint8_t tail[8]; size_t tn = 4;
    tail[0] = tail[1] = tail[2] = tail[3] = 0;
    if (cck) { const uint32_t d = xxh.digest(); tail[4] = (uint8_t)d; tail[5] = (uint8_t)(d >> 8); tail[6] = (uint8_t)(d >> 16); tail[7] = (uint8_t)(d >> 24); tn = 8; }
    yield(Slice{tail, tn});
    if (prefs.frameInfo.contentSize && prefs.frameInfo.contentSize != total) handleLz4Error(make_err(LZ4F_ERROR_frameSize_wrong));
    if (blockList && !bl.at.empty()) {
        // a skippable frame behind the LZ4 frame (frame_dev.cuh: the trailer): any LZ4 reader skips it; lz4f_mi355x_dev_decompressFrame
        // finds the size words through it instead of walking them
        const uint64_t F = framePos + tn;
        const size_t tsz = host_trailer_size(F, bl.at.size());
        handleLz4Error(tsz);
        std::vector<uint8_t> tr(tsz);
        host_write_trailer(tr.data(), F, bl.at.data(), (uint32_t)bl.at.size());
        yield(Slice{tr.data(), tr.size()});
    }
}

// decompressBatched: decodes every frame of the stream through the bulk path, in BOUNDED memory.  The conduit walks the size words over
// the chunks as they arrive (a 4-byte read per block) and hands runs of whole blocks - `batchBytes` of them at a time - to
// lz4f_mi355x_fdec_blocks (slabs of blocks in flight on the GPU(s), output yielded slab by slab, in order); what is held is one batch of
// input, the chunk being cut and the slabs in flight, whatever the stream's length (the reference's `decompress` holds one
// max(hint, 16 KiB) buffer, Conduit.hsc:634-659; rounds 2-3 gathered the whole stream first).
// Unlike `decompress` (which mirrors the reference and stops after the first frame, Appendix C quirk 3, and cannot read dictID
// headers, quirk 2), this walks the whole stream the way LZ4F_decompress called in a loop would: skippable frames are skipped,
// concatenated frames are all decoded, any header the format allows is accepted.
void decompressBatched(const Await& await, const Yield& yield, size_t batchBytes)
{
    if (batchBytes < ((size_t)1 << 20)) batchBytes = (size_t)1 << 20;
    std::vector<uint8_t> buf;                                              // bytes received and not yet decoded: [off, buf.size())
    size_t off = 0, total_in = 0;
    bool eof = false;
    // at least `want` bytes behind `off` (false: the stream ended first)
    auto ensure = [&](size_t want) -> bool {
        while (buf.size() - off < want && !eof) {
            Slice bs;
            if (!await(bs)) { eof = true; break; }
            if (off && off == buf.size()) { buf.clear(); off = 0; }
            else if (off > (buf.size() >> 1) && off > (1u << 16)) { buf.erase(buf.begin(), buf.begin() + (ptrdiff_t)off); off = 0; }      // (keep the buffer from creeping: what is decoded goes)
            buf.insert(buf.end(), bs.data, bs.data + bs.size);
            total_in += bs.size;
        }
        return buf.size() - off >= want;
    };
    auto le32 = [&](size_t at) { const uint8_t* q = buf.data() + off + at; return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); };
    struct Ctx { const Yield* y; } ctx{&yield};
    auto to_yield = [](void* user, const void* data, size_t size) { (*((Ctx*)user)->y)(Slice{(const uint8_t*)data, size}); };
    struct Dec { lz4f_mi355x_fdec* d = nullptr; ~Dec() { lz4f_mi355x_fdec_free(d); } };
    bool any = false;
    for (;;) {
        if (!ensure(1)) break;                                               // the stream ends between frames
        if (!ensure(4)) {
            if (!any && total_in < 5) throw std::runtime_error("lz4 decompress error: not enough bytes for header; expected 5, got " + std::to_string(total_in));
            handleLz4Error(make_err(LZ4F_ERROR_frameHeader_incomplete));
        }
        any = true;
        const uint32_t magic = le32(0);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                        // skippable: magic, u32 size, payload - dropped as it arrives
            if (!ensure(8)) handleLz4Error(make_err(LZ4F_ERROR_frameHeader_incomplete));
            size_t sz = le32(4);
            off += 8;
            while (sz) {
                if (!ensure(1)) throw std::runtime_error("lz4 decompress error: stream ended before EndMark");
                const size_t take = std::min(sz, buf.size() - off);
                off += take; sz -= take;
            }
            continue;
        }
        if (!ensure(7)) {
            if (total_in < 5) throw std::runtime_error("lz4 decompress error: not enough bytes for header; expected 5, got " + std::to_string(total_in));
            handleLz4Error(make_err(LZ4F_ERROR_frameHeader_incomplete));
        }
        const size_t hs = LZ4F_headerSize(buf.data() + off, buf.size() - off);
        handleLz4Error(hs);                                                // frameType_unknown for anything that is no frame
        if (!ensure(hs)) handleLz4Error(make_err(LZ4F_ERROR_frameHeader_incomplete));
        Dec dec; LZ4F_frameInfo_t fi;
        handleLz4Error(lz4f_mi355x_fdec_create(&dec.d, buf.data() + off, hs, &fi));
        off += hs;
        const size_t crc = fi.blockChecksumFlag ? 4 : 0;
        const size_t max_block = block_size_of(fi.blockSizeID);
        // runs of whole blocks
        size_t run = 0;                                                    // bytes of whole blocks behind `off`
        for (;;) {
            if (!ensure(run + 4)) throw std::runtime_error("lz4 decompress error: stream ended before EndMark");
            const uint32_t w = le32(run);
            if (w == 0) break;
            const size_t csz = w & 0x7FFFFFFFu;
            if (csz > max_block) handleLz4Error(make_err(LZ4F_ERROR_maxBlockSize_invalid));
            const size_t step = 4 + csz + crc;
            if (!ensure(run + step)) throw std::runtime_error("lz4 decompress error: stream ended before EndMark");
            run += step;
            if (run >= batchBytes) { handleLz4Error(lz4f_mi355x_fdec_blocks(dec.d, to_yield, &ctx, buf.data() + off, run)); off += run; run = 0; }
        }
        if (run) { handleLz4Error(lz4f_mi355x_fdec_blocks(dec.d, to_yield, &ctx, buf.data() + off, run)); off += run; }
        const size_t tail = 4 + (fi.contentChecksumFlag ? 4 : 0);
        if (!ensure(tail)) throw std::runtime_error("lz4 decompress error: stream ended before EndMark");
        handleLz4Error(lz4f_mi355x_fdec_end(dec.d, buf.data() + off, tail));
        off += tail;
    }
    if (!any) throw std::runtime_error("lz4 decompress error: not enough bytes for header; expected 5, got " + std::to_string(total_in));
}

}  // namespace conduit
}  // namespace lz4f

// ---- C face (what tests and other languages bind) ----------------------------------------------
using namespace lz4f::conduit;

namespace {
template <typename F>
int guarded(char* errbuf, size_t errcap, F&& f)
{
    try { f(); if (errbuf && errcap) errbuf[0] = 0; return 0; }
    catch (const std::exception& e) {
        if (errbuf && errcap) { strncpy(errbuf, e.what(), errcap - 1); errbuf[errcap - 1] = 0; }
        return 1;
    }
}
Await wrap_await(lz4f_mi355x_await_fn a, void* user)
{
    return [a, user](Slice& out) {
        const void* p = nullptr;
        const size_t n = a(user, &p);
        if (p == nullptr) return false;
        out = Slice{(const uint8_t*)p, n};
        return true;
    };
}
Yield wrap_yield(lz4f_mi355x_yield_fn y, void* user) { return [y, user](Slice s) { y(user, s.data, s.size); }; }
}  // namespace

extern "C" {
int lz4f_mi355x_conduit_compress(size_t outBufferSize, const LZ4F_preferences_t* prefs, lz4f_mi355x_await_fn a, lz4f_mi355x_yield_fn y,
                                 void* user, char* errbuf, size_t errcap)
{
    return guarded(errbuf, errcap, [&] { compressWithOutBufferSize(outBufferSize, prefs, wrap_await(a, user), wrap_yield(y, user)); });
}
int lz4f_mi355x_conduit_compress_yield_immediately(const LZ4F_preferences_t* prefs, lz4f_mi355x_await_fn a, lz4f_mi355x_yield_fn y, void* user,
                                                   char* errbuf, size_t errcap)
{
    return guarded(errbuf, errcap, [&] { compressYieldImmediately(prefs, wrap_await(a, user), wrap_yield(y, user)); });
}
int lz4f_mi355x_conduit_decompress(lz4f_mi355x_await_fn a, lz4f_mi355x_yield_fn y, void* user, char* errbuf, size_t errcap)
{
    return guarded(errbuf, errcap, [&] { decompress(wrap_await(a, user), wrap_yield(y, user)); });
}
int lz4f_mi355x_conduit_compress_batched(size_t batchBytes, const LZ4F_preferences_t* prefs, lz4f_mi355x_await_fn a, lz4f_mi355x_yield_fn y,
                                         void* user, char* errbuf, size_t errcap)
{
    return guarded(errbuf, errcap, [&] { compressBatched(batchBytes, prefs, wrap_await(a, user), wrap_yield(y, user)); });
}
int lz4f_mi355x_conduit_compress_batched_listed(size_t batchBytes, const LZ4F_preferences_t* prefs, lz4f_mi355x_await_fn a, lz4f_mi355x_yield_fn y,
                                                void* user, char* errbuf, size_t errcap)
{
    return guarded(errbuf, errcap, [&] { compressBatched(batchBytes, prefs, wrap_await(a, user), wrap_yield(y, user), true); });
}
int lz4f_mi355x_conduit_decompress_batched(lz4f_mi355x_await_fn a, lz4f_mi355x_yield_fn y, void* user, char* errbuf, size_t errcap)
{
    return guarded(errbuf, errcap, [&] { decompressBatched(wrap_await(a, user), wrap_yield(y, user), (size_t)256 << 20); });
}
int lz4f_mi355x_conduit_decompress_batched_bounded(size_t batchBytes, lz4f_mi355x_await_fn a, lz4f_mi355x_yield_fn y, void* user, char* errbuf, size_t errcap)
{
    return guarded(errbuf, errcap, [&] { decompressBatched(wrap_await(a, user), wrap_yield(y, user), batchBytes); });
}
}
