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
// compressBatched: same frame bytes as compressWithPreferences, but input is gathered until
// `batchBytes` are available so that one LZ4F_compressUpdate call hands many blocks to the GPU.
void compressBatched(size_t batchBytes, const LZ4F_preferences_t* prefsIn, const Await& await, const Yield& yield)
{
    ScopedCctx s(prefsIn);
    std::vector<uint8_t> out(LZ4F_HEADER_SIZE_MAX), in;
    const size_t headerSize = handleLz4Error(LZ4F_compressBegin(s.ctx, out.data(), out.size(), &s.prefs));
    yield(Slice{out.data(), headerSize});
    auto flushBatch = [&]() {
        if (in.empty()) return;
        const size_t bound = handleLz4Error(LZ4F_compressBound(in.size(), &s.prefs));
        if (out.size() < bound) out.resize(bound);
        const size_t w = handleLz4Error(LZ4F_compressUpdate(s.ctx, out.data(), out.size(), in.data(), in.size(), NULL));
        if (w) yield(Slice{out.data(), w});
        in.clear();
    };
    Slice bs;
    while (await(bs)) {
        in.insert(in.end(), bs.data, bs.data + bs.size);
        if (in.size() >= batchBytes) flushBatch();
    }
    flushBatch();
    const size_t footerSize = handleLz4Error(LZ4F_compressBound(0, &s.prefs));
    if (out.size() < footerSize) out.resize(footerSize);
    const size_t fw = handleLz4Error(LZ4F_compressEnd(s.ctx, out.data(), out.size(), NULL));
    yield(Slice{out.data(), fw});
}

// decompressBatched: gathers the frame and decodes all of its blocks in one bulk call
// (lz4f_mi355x_decompressFrame: host walk of the size words, slabs of blocks per launch).
void decompressBatched(const Await& await, const Yield& yield)
{
    std::vector<uint8_t> frame;
    Slice bs;
    while (await(bs)) frame.insert(frame.end(), bs.data, bs.data + bs.size);
    if (frame.size() < 5)
        throw std::runtime_error("lz4 decompress error: not enough bytes for header; expected 5, got " + std::to_string(frame.size()));
    // Unlike `decompress` (which mirrors the reference and stops after the first frame, Appendix C quirk 3, and cannot read
    // dictID headers, quirk 2), this walks the whole stream the way LZ4F_decompress called in a loop would: skippable
    // frames are skipped, concatenated frames are all decoded, any header the format allows is accepted.
    auto le32 = [&](size_t at) { return (uint32_t)frame[at] | ((uint32_t)frame[at + 1] << 8) | ((uint32_t)frame[at + 2] << 16) | ((uint32_t)frame[at + 3] << 24); };
    size_t at = 0;
    std::vector<uint8_t> out;
    while (at < frame.size()) {
        const uint8_t* f = frame.data() + at;
        const size_t left = frame.size() - at;
        if (left < 4) handleLz4Error(make_err(LZ4F_ERROR_frameHeader_incomplete));
        const uint32_t magic = le32(at);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                        // skippable: magic, u32 size, payload
            if (left < 8) handleLz4Error(make_err(LZ4F_ERROR_frameHeader_incomplete));
            const size_t sz = le32(at + 4);
            if (left - 8 < sz) throw std::runtime_error("lz4 decompress error: stream ended before EndMark");
            at += 8 + sz;
            continue;
        }
        ParsedHeader ph;
        handleLz4Error(parse_frame_header(f, left, &ph));                  // frameType_unknown for anything else
        // capacity: blocks * maxBlockSize from a walk of the size words
        size_t cap = 0, pos = ph.header_size;
        for (;;) {
            if (left - pos < 4) throw std::runtime_error("lz4 decompress error: stream ended before EndMark");
            const uint32_t w = le32(at + pos);
            if (w == 0) break;
            const size_t step = 4 + (size_t)(w & 0x7FFFFFFFu) + (ph.info.blockChecksumFlag ? 4 : 0);
            if (left - pos < step) throw std::runtime_error("lz4 decompress error: stream ended before EndMark");
            pos += step;
            cap += ph.max_block;
        }
        out.resize(cap ? cap : 1);
        size_t used = 0;
        const size_t n = handleLz4Error(lz4f_mi355x_decompressFrame(out.data(), cap, f, left, &used));
        yield(Slice{out.data(), n});
        at += used;
    }
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
int lz4f_mi355x_conduit_decompress_batched(lz4f_mi355x_await_fn a, lz4f_mi355x_yield_fn y, void* user, char* errbuf, size_t errcap)
{
    return guarded(errbuf, errcap, [&] { decompressBatched(wrap_await(a, user), wrap_yield(y, user)); });
}
}
