// engine.hpp -- per-device engine: HIP stream, grow-only HBM workspace, kernel launch plumbing.
// Host side of the product above the C ABI (C++, because the reference's host is compiled code).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>

#include "../../include/lz4f_mi355x.h"

namespace lz4f {

// ---- errors (LZ4F convention: (size_t)-code) ----
inline size_t make_err(int code) { return (size_t)-(ptrdiff_t)code; }
inline bool   is_err(size_t v) { return v > make_err(LZ4F_ERROR_maxCode); }
const char*   err_name(size_t v);
void          set_last_error(const char* fmt, ...);
const char*   last_error();

// ---- host XXH32 (header checksum byte; whole-stream content checksum, which is serial) ----
struct Xxh32State {
    uint32_t total_len, large, v[4], memsize;
    uint8_t  mem[16];
    void reset(uint32_t seed = 0);
    void update(const void* data, size_t len);
    uint32_t digest() const;
};
uint32_t xxh32_host(const void* data, size_t len, uint32_t seed = 0);

// ---- frame header helpers (row a6; host side) ----
size_t block_size_of(unsigned blockSizeID);                 // 0 when invalid
size_t write_frame_header(uint8_t* dst, const LZ4F_preferences_t& p);      // p.frameInfo.blockSizeID already resolved
size_t compress_bound_internal(size_t srcSize, const LZ4F_preferences_t* prefs, size_t alreadyBuffered);
struct ParsedHeader { LZ4F_frameInfo_t info; size_t header_size; size_t max_block; };
// LZ4F_decodeHeader semantics for a complete header at src (>= header size bytes available)
size_t parse_frame_header(const uint8_t* src, size_t n, ParsedHeader* out);

struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    int ensure(size_t n);          // grow-only hipMalloc; 0 on success
    void release();
};
struct PinBuf {
    void* p = nullptr; size_t cap = 0;
    int ensure(size_t n);          // hipHostMalloc
    void release();
};

}  // namespace lz4f

struct lz4f_mi355x_engine {
    int   device = 0;
    void* stream = nullptr;        // hipStream_t
    bool  own_stream = false;
    lz4f::DevBuf info, recs, table, blk_bytes, res, bad;   // workspace of the block kernels
    lz4f::DevBuf ixtmp;                                    // in-band index: made here, copied into the frame's trailer
    lz4f::DevBuf walkbuf;                                  // frames without a block table: the parallel walk's candidates
    lz4f::DevBuf density;                                  // k_density_probe: which decoder a frame of big blocks without an index goes to
    lz4f::DevBuf e1_scratch;                               // pass E1: per workgroup, the slice lists of the tile it is searching
    lz4f::DevBuf spx;                                      // big independent blocks without an index: the true points of every block's check lines (decode_spx.cuh)
    lz4f::DevBuf selfix, selfcnt;                          // linked frames without an index: the one made here, and its per-block counts
    lz4f::DevBuf pdbuf;                                    // dense frames by pointer doubling: a word per output byte
    lz4f::DevBuf postab;                                   // dense frames: output position / 64 -> sequence (k_build_postab)
    lz4f::DevBuf desc, seqcnt;                             // two-kernel decode: sequence descriptors, per-block counts
    lz4f::DevBuf d_in, d_out;                              // staging for the host-pointer paths
    lz4f::PinBuf h_in, h_out, h_small;
    // A-B and development switches: read from the environment ONCE, when the engine is made (engines of the host-pointer calls
    // live in a pool: lz4f_mi355x_release_engines() makes the next ones read it again)
    struct Switches {
        bool no_index, no_selfindex, no_resolve, no_trace, no_doubling, trace_always, no_groups, no_window, serial_walk, no_trailer, no_content_check, no_density_probe, no_spx, no_overlap, prof, e1_sync, no_selffeed;
        unsigned dense_mode;           // dense payloads in big independent blocks (LZ4F_MI355X_DENSE_MODE): 0 by block count, 1 workgroup per block (decode_relay.cuh), 2 wave per block
        unsigned group_kib;            // (development) bytes of consecutive small blocks of a linked frame a workgroup of the indexed copy kernel takes
        unsigned feed_round;           // (test switch) sequences per parse round of the self-feeding copy kernel's first wave
        int chain_gate; char decode_mode; unsigned e1_run, e1_solo, seed, dblk_lds, recs_per_tile; unsigned long long wait_ticks;
        void read();
    } sw;
    void*  recs_ctl_clean = nullptr;                       // == recs.p while the record pool's control words are known to be zero (or about to be: the last call's scan)
    size_t ix_seq_cap = 0;                                 // indexed decode: descriptor workspace, in sequences (grow-only)
    bool  timing = false;
    void* ev[24] = {nullptr};      // hipEvent_t pairs (begin,end) per timing slot
    bool  ev_used[12] = {false};
    void  tick(int slot, bool end, void* on_stream = nullptr);
    // a second stream of the engine's own: work that only reads what the main stream's kernels read (block-checksum verification beside
    // the decode) is forked onto it and joined before the verdict (fork / join: events between the two streams, nothing on the host)
    void* aux_stream = nullptr; void* ev_fork = nullptr; void* ev_join = nullptr;
    bool  aux_ready();
    bool  aux_pending = false;     // work forked onto aux_stream that the main stream has not waited for yet

    // ---- device-pointer paths (asynchronous on `stream`) ----
    struct CompressJob {
        const uint8_t* d_src; uint64_t src_size; uint64_t first_off;
        uint32_t block_size; bool linked; bool block_checksum; bool endmark;
        bool content_checksum;                                   // (with endmark) XXH32 of the whole input behind the EndMark: k_xxh32_content
        uint8_t header[20]; uint32_t header_size;
    };
    // returns 0 or an LZ4F error; d_res/d_table may be null (internal buffers are used)
    size_t launch_compress(const CompressJob& j, uint8_t* d_dst, uint64_t dst_cap, lz4f_mi355x_result* d_res, lz4f_mi355x_block* d_table,
                           void* d_index = nullptr, size_t index_cap = 0);
    struct DecompressJob {
        const uint8_t* d_frame; uint64_t frame_cap; uint8_t* d_dst; uint64_t dst_cap; uint64_t hist0;
        void* d_index; size_t index_size;   // sequence index written by launch_compress (optional; independent blocks only)
        uint32_t block_size; bool linked; bool block_checksum;
        bool content_checksum;                                   // the frame's FLG asks for one: verified behind the decode (k_xxh32_content)
        const lz4f_mi355x_block* d_table; uint32_t n_blocks;     // when d_table != null the walk is skipped
        bool table_in_place;                                     // the engine's own table already holds n_blocks entries
        lz4f_mi355x_block* table_direct;                         // or: a table in device memory that is the engine's to overwrite (its own staging copy): used where it lies
        uint32_t max_blocks;                                     // grid bound when walking
        const uint64_t* hint_list; uint32_t hint_n;              // from the frame's trailer: where the size words should be (checked on the device)
        uint32_t ix_seqs, ix_entries;                            // what the index says it holds (header / trailer footer: sizes the workspace; checked on the device)
    };
    size_t launch_decompress(const DecompressJob& j, lz4f_mi355x_result* d_res);
    size_t sync();

    // ---- host-pointer helpers used by the streaming contexts and the bulk host calls ----
    // Encode `n` bytes at src (host) as frame blocks of block_size (last may be short); `hist` bytes of
    // history (host) precede them when linked.  Appends [size word][payload][checksum] per block to out.
    size_t compress_blocks_host(const uint8_t* src, size_t n, const uint8_t* hist, size_t hist_len,
                                uint32_t block_size, bool linked, bool block_checksum, uint8_t* dst, size_t dst_cap, size_t* written);
    // The same in two halves, for the pipelined bulk calls (pipeline.hip): upload + kernels + result (the blocks stay in d_out),
    // then the download.  `*_pinned`: the host buffer is page-locked, no staging copy.
    size_t slab_compress(const uint8_t* src, size_t n, const uint8_t* hist, size_t hist_len, uint32_t block_size, bool linked, bool block_checksum,
                         bool src_pinned, size_t* size);
    size_t slab_fetch(uint8_t* dst, size_t size, size_t d_off, bool dst_pinned);
    // One block (or a short run of them) out of page-locked host memory and back into it, no copy calls: the kernels read the input
    // (hist_len bytes of history, then n bytes) and write the block(s) and the result record through the link themselves.  What the
    // LZ4F_* streaming functions do per completed block (frame_host.cpp): upload, download and one of the two synchronisations of the
    // staged path are gone.  Both buffers from lz4f::PinBuf (hipHostMalloc).
    size_t compress_block_pinned(const uint8_t* pin_src, size_t hist_len, size_t n, uint32_t block_size, bool linked, bool block_checksum,
                                 uint8_t* pin_dst, size_t dst_cap, void* pin_res, size_t* size);
    size_t slab_decode(const uint8_t* frame_part, size_t part_len, const std::vector<lz4f_mi355x_block>& entries, const lz4f::ParsedHeader& ph,
                       const uint8_t* hist, size_t hist_len, bool src_pinned, size_t* got, uint8_t* fetch_to = nullptr, size_t fetch_room = 0);
    // Decode one compressed block payload (host; followed by its 4-byte checksum when bck) with `hist_len`
    // bytes of history (host, linked frames).  The checksum is verified and the block decoded on the GPU;
    // the decoded bytes land in dst (host).
    size_t decompress_block_host(const uint8_t* payload, uint32_t csize, bool bck, const uint8_t* hist, size_t hist_len,
                                 uint8_t* dst, uint32_t dst_cap, bool linked, uint32_t block_size, uint32_t* decoded);
    // Whole frame (host) -> dst (host): host walk of the size words, then one device call per slab of blocks.
    size_t decompress_frame_host(const uint8_t* frame, size_t n, const lz4f::ParsedHeader& ph, uint8_t* dst, size_t cap,
                                 size_t* decoded, size_t* consumed);

    ~lz4f_mi355x_engine();

private:
    size_t run_decode_slab(const uint8_t* frame_part, size_t part_len, const std::vector<lz4f_mi355x_block>& entries,
                           const lz4f::ParsedHeader& ph, const uint8_t* hist, size_t hist_len, uint8_t* dst, size_t dst_room, size_t* got);
};

namespace lz4f {
// an engine for one host-pointer call: borrowed from the process-wide pool (made on first use), given back by the lease.
// device < 0: the calling thread's selected device
size_t acquire_engine(lz4f_mi355x_engine** out, int device = -1);
void   release_engine(lz4f_mi355x_engine* e);
void   release_idle_engines();
struct EngineLease {
    lz4f_mi355x_engine* e = nullptr;
    size_t get(int device = -1) { return acquire_engine(&e, device); }
    ~EngineLease() { release_engine(e); }
    lz4f_mi355x_engine* operator->() const { return e; }
};
size_t new_engine(lz4f_mi355x_engine** out, int device, void* stream, bool borrow);
int    selected_device();
bool   is_pinned_host(const void* p);
// pipelined bulk paths (pipeline.hip)
int    bulk_devices();
void   set_bulk_devices(int n);
int    logical_devices();             // visible devices, or LZ4F_MI355X_LOGICAL_DEVICES when that is larger (test switch, pipeline.hip)
// hist_before: valid input bytes in front of src (a linked frame's blocks reach 64 KiB back)
size_t pipe_compress_blocks(const uint8_t* src, size_t n, uint32_t block_size, bool linked, bool bck, uint8_t* dst, size_t cap, size_t* written,
                            size_t hist_before = 0);
// what a frame decoded batch by batch carries from one batch of blocks to the next (pipeline.hip)
struct FrameCarry { Xxh32State cck; uint64_t out_total = 0; std::vector<uint8_t> hist; FrameCarry() { cck.reset(0); } };
size_t pipe_decompress_frame(const uint8_t* frame, size_t n, const ParsedHeader& ph, uint8_t* flat, size_t flat_cap,
                             const std::function<void(const uint8_t*, size_t)>* sink, size_t* decoded, size_t* consumed, FrameCarry* carry = nullptr);
uint32_t pick_chunk_size(uint32_t block_size);
// The block-list trailer (frame_dev.cuh: the trailer) made on the host, for frames the host paths assemble or are handed:
// `at` = where every block's size word is in the frame.  host_trailer_size: the bytes it takes behind a frame of frame_size bytes
// (0 without blocks, an error code beyond what a skippable frame can say); host_write_trailer writes them at `dst` (= frame + frame_size).
struct BlockList {
    std::vector<uint64_t> at;
    // the size words of whole blocks lying in blocks[0..n), which sit at frame_off in the frame; false when they do not tile n bytes
    bool add_blocks(const uint8_t* blocks, size_t n, uint64_t frame_off, bool block_checksum);
};
size_t host_trailer_size(uint64_t frame_size, uint64_t n_blocks);
void   host_write_trailer(uint8_t* dst, uint64_t frame_size, const uint64_t* at, uint32_t n_blocks);
}
