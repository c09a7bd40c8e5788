// pipeline.hip -- the host-pointer bulk calls (lz4f_mi355x_compressFrame / decompressFrame / decompressFrameTo): a frame's
// blocks go to the GPU(s) in slabs of consecutive blocks, several slabs in flight.
//
// What the reference's conduits reach (Conduit.hsc:311, :591) is host memory, so this path is bounded by the PCIe link
// (Gen5 x16: ~55 GB/s each way), two orders of magnitude below the kernels.  To get near it:
//   * slabs of SLAB bytes are dealt round-robin to `slots` = 2 engines per device (an engine = a HIP stream + its device
//     workspace + pinned staging), each slot driven by its own host thread: while one slab's kernels and download run, the
//     next slab of that device is already uploading, and the link is busy in both directions;
//   * host buffers that are page-locked (lz4f_mi355x_host_alloc, what the batched conduits gather their input in) are read
//     and written by the DMA engines directly; pageable ones go through the engine's pinned staging, the copies of the
//     slots running side by side;
//   * independent blocks need nothing from each other, so with more than one device (lz4f_mi355x_use_devices) the slabs
//     - runs of consecutive blocks - are dealt over the devices as well: no collective, the host gathers sizes and puts
//     every slab's blocks behind those of the slab in front.  A linked frame's slabs take their
//     64 KiB of history from the input (compress); decoding one is a chain from slab to slab and stays on one engine.
// Output order: a slab knows where its bytes go once every slab in front has reported its size; the sizes travel long
// before the bytes do (the result record is 32 bytes), so this chain does not serialise the transfers.
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "engine.hpp"

namespace lz4f {

static std::atomic<int> g_devices{1};
int bulk_devices() { return g_devices.load(); }
void set_bulk_devices(int n) { g_devices.store(n < 1 ? 1 : n); }
// LZ4F_MI355X_LOGICAL_DEVICES=n (test switch): lz4f_mi355x_use_devices() may then ask for up to n devices although fewer are
// visible; logical device d is physical device d mod visible.  Everything above the engines - the deal of the slabs over the
// devices, the turn order, one engine set per (logical) device - runs as it would on n GPUs; the link tokens are per PHYSICAL
// device, as the link is.  It is how the multi-device code is exercised on a one-GPU box; it buys no speed.
int logical_devices()
{
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 0) { (void)hipGetLastError(); have = 0; }
    if (have <= 0) return 0;
    if (const char* v = getenv("LZ4F_MI355X_LOGICAL_DEVICES")) { const int n = atoi(v); if (n > have && n <= 64) return n; }
    return have;
}

namespace {

constexpr size_t SLAB = (size_t)64 << 20;                        // input bytes per slab (compress); a multiple of every block size

// slabs take turns to learn their output offset (and, for callers that want the bytes in order, to hand them over)
struct Turns {
    std::mutex mu; std::condition_variable cv;
    size_t next = 0, offset = 0, err = 0;
    // wait until slab k is the next one; returns false if a slab in front failed
    bool enter(size_t k) { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return next == k || err; }); return !err; }
    void leave(size_t add) { { std::lock_guard<std::mutex> g(mu); offset += add; next++; } cv.notify_all(); }
    void fail(size_t e) { { std::lock_guard<std::mutex> g(mu); if (!err) err = e; } cv.notify_all(); }
};

struct Slots {
    std::vector<EngineLease> eng;
    size_t init(int ndev, int per_dev)
    {
        const int first = selected_device();
        int have = 0;
        if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) { set_last_error("no usable HIP device: liblz4f_mi355x has no CPU fallback"); return make_err(LZ4F_ERROR_GENERIC); }
        const int logical = logical_devices();
        if (ndev > logical) ndev = logical;
        eng.resize((size_t)ndev * per_dev);
        for (int e = 0; e < per_dev; e++)
            for (int d = 0; d < ndev; d++) {                       // (slot order: device-major inside a round, so consecutive slabs go to different devices)
                size_t r = eng[(size_t)e * ndev + d].get((first + d) % have);
                if (is_err(r)) return r;
            }
        return 0;
    }
};

}  // namespace

// ---- compress: the blocks (size word, payload, checksum) of `n` input bytes -> dst -------------------------------------
size_t pipe_compress_blocks(const uint8_t* src, size_t n, uint32_t block_size, bool linked, bool bck, uint8_t* dst, size_t cap, size_t* written, size_t hist_before)
{
    *written = 0;
    if (n == 0) return 0;
    const size_t nslab = (n + SLAB - 1) / SLAB;
    const bool src_pinned = is_pinned_host(src), dst_pinned = is_pinned_host(dst);
    Slots slots;
    const int ndev = nslab >= 2 ? bulk_devices() : 1;
    size_t r = slots.init(ndev, nslab >= 2 ? 2 : 1);
    if (is_err(r)) return r;
    const size_t nslot = std::min(slots.eng.size(), nslab);
    Turns turns;
    const bool prof = getenv("LZ4F_MI355X_PROF") != nullptr;
    std::vector<std::string> errs(nslot);
    auto work = [&](size_t slot) {
        lz4f_mi355x_engine* e = slots.eng[slot].e;
        for (size_t k = slot; k < nslab; k += nslot) {
            const size_t off = k * SLAB, len = std::min(SLAB, n - off);
            const size_t hl = linked ? std::min(off + hist_before, (size_t)65536) : 0;      // (history: the input in front of the slab)
            size_t size = 0;
            const auto t0 = std::chrono::steady_clock::now();
            size_t rr = e->slab_compress(src + off, len, src + off - hl, hl, block_size, linked, bck, src_pinned, &size);
            if (is_err(rr)) { errs[slot] = last_error(); turns.fail(rr); return; }
            const auto t1 = std::chrono::steady_clock::now();
            if (!turns.enter(k)) return;
            const size_t at = turns.offset;
            if (at + size > cap) { turns.fail(make_err(LZ4F_ERROR_dstMaxSize_tooSmall)); return; }
            turns.leave(size);                                    // the slab behind may place itself; my bytes follow
            const auto t2 = std::chrono::steady_clock::now();
            rr = e->slab_fetch(dst + at, size, 0, dst_pinned);
            if (is_err(rr)) { errs[slot] = last_error(); turns.fail(rr); return; }
            if (prof) { const auto t3 = std::chrono::steady_clock::now(); auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
                        fprintf(stderr, "slab %zu slot %zu: upload+kernels %ld us, turn %ld us, download %ld us (%zu -> %zu bytes)\n", k, slot, us(t0, t1), us(t1, t2), us(t2, t3), len, size); }
        }
    };
    if (nslot == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (size_t s = 1; s < nslot; s++) th.emplace_back(work, s);
        work(0);
        for (auto& t : th) t.join();
    }
    if (turns.err) { for (auto& m : errs) if (!m.empty()) { set_last_error("%s", m.c_str()); break; } return turns.err; }
    *written = turns.offset;
    return 0;
}

// ---- decompress: a whole frame (host) -> `sink` (slab by slab, in order) ------------------------------------------------
// sink(data, size, pinned_hint): the decoded bytes of the next slab; when `flat` is given the slabs are written there
// instead (at their final offsets, straight from the device when it is page-locked) and sink is not called.
// carry != nullptr: [s, s + n) is a RUN OF WHOLE BLOCKS of a frame whose header was `ph` (no header in front; an EndMark, if the run
// holds one, ends the walk) - what the bounded batched decoder hands over batch by batch (frame_host.cpp: lz4f_mi355x_fdec_*).  The
// frame-level state lives in `carry` between the calls: the content checksum's running state, the output so far, and - linked frames -
// the last 64 KiB handed over, which the next batch's first slab decodes against.  The frame's closing checks are the caller's then.
size_t pipe_decompress_frame(const uint8_t* s, size_t n, const ParsedHeader& ph, uint8_t* flat, size_t flat_cap,
                             const std::function<void(const uint8_t*, size_t)>* sink, size_t* decoded, size_t* consumed, FrameCarry* carry)
{
    const size_t SLAB_SRC = (size_t)64 << 20, SLAB_DST = (size_t)192 << 20;
    const size_t crc = ph.info.blockChecksumFlag ? 4 : 0;
    const bool linked = ph.info.blockMode == LZ4F_blockLinked;
    auto rd32 = [&](size_t at) { return (uint32_t)s[at] | ((uint32_t)s[at + 1] << 8) | ((uint32_t)s[at + 2] << 16) | ((uint32_t)s[at + 3] << 24); };
    // the host walks the size words (a read per block of memory it holds) and cuts the block list into slabs
    struct SlabD { size_t src_at, src_len, first, count; };
    std::vector<lz4f_mi355x_block> entries;
    std::vector<SlabD> slabs;
    size_t pos = carry ? 0 : ph.header_size;
    bool endmark = false;
    for (;;) {
        if (carry && pos == n) break;                           // (a batch ends with its last whole block)
        if (n - pos < 4) return make_err(LZ4F_ERROR_frameHeader_incomplete);
        const uint32_t w = rd32(pos);
        if (w == 0) { endmark = true; break; }
        const size_t csz = w & 0x7FFFFFFFu;
        if (csz > ph.max_block) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
        if (n - pos - 4 < csz + crc) return make_err(LZ4F_ERROR_frameHeader_incomplete);
        if (slabs.empty() || (pos - slabs.back().src_at) + 4 + csz + crc > SLAB_SRC || (slabs.back().count + 1) * ph.max_block > SLAB_DST)
            slabs.push_back(SlabD{pos, 0, entries.size(), 0});
        SlabD& sl = slabs.back();
        lz4f_mi355x_block e;
        e.src_off = pos + 4 - sl.src_at; e.dst_off = sl.count * ph.max_block; e.word = w; e.dst_size = (uint32_t)ph.max_block;
        entries.push_back(e);
        pos += 4 + csz + crc;
        sl.count++; sl.src_len = pos - sl.src_at;
    }
    if (endmark) pos += 4;
    const size_t nslab = slabs.size();
    const bool src_pinned = is_pinned_host(s), dst_pinned = flat && is_pinned_host(flat);
    Xxh32State cck_here; cck_here.reset(0);
    Xxh32State& cck = carry ? carry->cck : cck_here;
    const bool want_cck = ph.info.contentChecksumFlag != 0;
    Turns turns;
    size_t out_total = 0;
    if (nslab) {
        Slots slots;
        // a linked frame is a chain from slab to slab (each needs the last 64 KiB the one before produced): one engine
        const bool parallel = !linked && nslab >= 2;
        // (four engines per device: a block's parse is a ~2.5 ms chain however few blocks a slab has - upload 0.6 + kernels 2.5 +
        // download 1.2 ms per 64 MiB slab - so it takes four slabs in flight to keep the download link busy; three gave 39 GiB/s, five 41, four 44)
        size_t r = slots.init(parallel ? bulk_devices() : 1, parallel ? (nslab >= 4 ? 4 : nslab >= 3 ? 3 : 2) : 1);
        if (is_err(r)) return r;
        const size_t nslot = std::min(slots.eng.size(), nslab);
        std::vector<std::string> errs(nslot);
        std::vector<uint8_t> hist_here;
        std::vector<uint8_t>& hist_keep = carry ? carry->hist : hist_here;      // (linked + sink: the last 64 KiB handed over)
        bool ready = false; const uint8_t* ready_ptr = nullptr; size_t ready_len = 0;      // sink: the slab whose turn it is, waiting to be handed over
        auto work = [&](size_t slot) {
            lz4f_mi355x_engine* e = slots.eng[slot].e;
            for (size_t k = slot; k < nslab; k += nslot) {
                const SlabD& sl = slabs[k];
                std::vector<lz4f_mi355x_block> ent(entries.begin() + sl.first, entries.begin() + sl.first + sl.count);
                const uint8_t* hist = nullptr; size_t hl = 0;
                if (linked && (k || (carry && !flat))) {         // (nslot == 1 here: the slab in front is complete; a batch's first slab: what the batch before left)
                    if (flat) { hl = std::min(turns.offset, (size_t)65536); hist = flat + turns.offset - hl; }
                    else { hl = hist_keep.size(); hist = hist_keep.data(); }
                }
                size_t got = 0;
                size_t rr = e->slab_decode(s + sl.src_at, sl.src_len, ent, ph, hist, hl, src_pinned, &got);
                if (is_err(rr)) { errs[slot] = last_error(); turns.fail(rr); return; }
                if (!turns.enter(k)) return;
                const size_t at = turns.offset;
                if (flat) {
                    if (at + got > flat_cap) { turns.fail(make_err(LZ4F_ERROR_dstMaxSize_tooSmall)); return; }
                    if (!want_cck && !linked) turns.leave(got);                   // the slab behind may place itself; my bytes follow
                    rr = e->slab_fetch(flat + at, got, hl, dst_pinned);
                    if (is_err(rr)) { errs[slot] = last_error(); turns.fail(rr); return; }
                    if (want_cck) cck.update(flat + at, got);                     // (one serial chain over the whole stream: in slab order)
                    if (want_cck || linked) turns.leave(got);
                } else {
                    // bytes in order through the engine's pinned buffer; the CALLER's thread hands them over (see below), this
                    // thread goes on when it has
                    if (e->h_out.ensure(got + 64)) { turns.fail(make_err(LZ4F_ERROR_allocation_failed)); return; }
                    rr = e->slab_fetch((uint8_t*)e->h_out.p, got, hl, true);
                    if (is_err(rr)) { errs[slot] = last_error(); turns.fail(rr); return; }
                    if (want_cck) cck.update(e->h_out.p, got);
                    if (linked) { const uint8_t* o = (const uint8_t*)e->h_out.p; const size_t keep = std::min(got, (size_t)65536); hist_keep.insert(hist_keep.end(), o + got - keep, o + got); if (hist_keep.size() > 65536) hist_keep.erase(hist_keep.begin(), hist_keep.begin() + (hist_keep.size() - 65536)); }
                    { std::unique_lock<std::mutex> g(turns.mu); ready_ptr = (const uint8_t*)e->h_out.p; ready_len = got; ready = true; turns.cv.notify_all(); turns.cv.wait(g, [&] { return !ready || turns.err; }); }
                    turns.leave(got);
                }
            }
        };
        if (sink) {
            // every slot on a thread of its own; this thread hands the slabs to the caller's sink, in order
            std::exception_ptr sink_failed;
            std::vector<std::thread> th;
            for (size_t sl = 0; sl < nslot; sl++) th.emplace_back(work, sl);
            for (size_t k = 0; k < nslab; k++) {
                std::unique_lock<std::mutex> g(turns.mu);
                turns.cv.wait(g, [&] { return ready || turns.err; });
                if (turns.err) break;
                const uint8_t* ptr = ready_ptr; const size_t len = ready_len;
                g.unlock();
                // (the sink is the caller's code - a conduit's yield - and may throw: the workers are joinable threads, so the failure is
                // recorded, they are let go and joined, and only then does the exception travel on)
                try { (*sink)(ptr, len); }
                catch (...) { sink_failed = std::current_exception(); turns.fail(make_err(LZ4F_ERROR_GENERIC)); g.lock(); ready = false; turns.cv.notify_all(); break; }
                g.lock();
                ready = false;
                turns.cv.notify_all();
            }
            for (auto& t : th) t.join();
            if (sink_failed) std::rethrow_exception(sink_failed);
        } else if (nslot == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (size_t sl = 1; sl < nslot; sl++) th.emplace_back(work, sl);
            work(0);
            for (auto& t : th) t.join();
        }
        if (turns.err) { for (auto& m : errs) if (!m.empty()) { set_last_error("%s", m.c_str()); break; } return turns.err; }
        out_total = turns.offset;
    }
    if (carry) { carry->out_total += out_total; *decoded = out_total; *consumed = pos; return 0; }
    if (ph.info.contentSize && ph.info.contentSize != out_total) return make_err(LZ4F_ERROR_frameSize_wrong);
    if (want_cck) {
        if (n - pos < 4) return make_err(LZ4F_ERROR_frameHeader_incomplete);
        if (rd32(pos) != cck.digest()) return make_err(LZ4F_ERROR_contentChecksum_invalid);
        pos += 4;
    }
    *decoded = out_total; *consumed = pos;
    return 0;
}

}  // namespace lz4f
