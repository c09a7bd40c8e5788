// decode_indexed.cuh -- decode of big independent blocks and of linked frames through a SEQUENCE INDEX: the compressor's,
// or (linked frames that come without one) an index the decoder makes itself
// (DESIGN.md section 8 item 1).
//
// The scalar parser of decode_fused.cuh costs ~200 scalar instructions per sequence and a CU has one scalar unit:
// however the waves are arranged, the GPU parses the headline workload in >= 2 ms.  A lane-per-entry walk of the same
// chains needs entry points.  The compressor has them: pass E2 walks the records with both positions at hand and notes,
// every 16 sequences, where the token sits in the block's payload and which output position the sequence starts at.
// That table (16 bytes per 16 sequences) is the index; frames stay plain LZ4 frames, and a frame without an index (or
// with one that does not fit) is decoded by the generic kernels.
//   k_index_blocks / k_build_index / pass E2 (compress side, encode.cuh)  the index
//   k_check_index     the index header against this call's geometry, buffer and workspace (on the device)
//   k_parse_indexed   one LANE per entry: walks its sequences in the payload and writes 16-byte descriptors
//                     {literal source, literal length, output position, match length, offset} to HBM; checks that
//                     it ends exactly where the next entry starts (else the index is declared unusable)
//                     A match whose source lies inside a recent literal run (or inside such a match) is marked
//                     DIRECT: its bytes are in the payload, so it is copied like a literal run, in any order.
//   k_resolve_direct  one lane per sequence: matches whose source lies in a literal run further back (binary search over
//                     the block's descriptors, following plain matches a few hops) become direct as well
//   k_copy_indexed    one workgroup per block: the copier waves of decode_fused.cuh, fed with those descriptors by
//                     wave 0 instead of by a parser wave; only the matches that are not direct form a chain.
//                     Linked frames: a workgroup per group of consecutive small blocks; chain matches that read another
//                     workgroup's output are set aside and replayed when that one is far enough.
//   k_selfindex_walk / k_selfindex_scan   linked frames without an index: a lane per block walks the payload (parsing
//                     needs no history), a scan places the blocks, the entries are written - then everything above
//   k_dense_gate / k_build_postab / k_pd_init / k_pd_round / k_pd_verdict   DENSE frames (text, repetitive records: most
//                     sequences on the match chain) are not walked as chains at all: every output byte takes one hop back
//                     to where it comes from, then pointer doubling over a word per output byte (see below)
//   k_trace_copy      the same origins hop by hop, for when that scratch cannot be had
#pragma once
#include "common.cuh"
#include "decode.cuh"
#include "decode_fused.cuh"
#include "decode_linked.cuh"
#include "encode.cuh"

namespace lz4f {

// 8 payload bytes at `pos`, never touching memory at or beyond in + readable
__device__ __forceinline__ uint64_t pt_load8(const uint8_t* __restrict__ in, uint32_t pos, uint64_t readable)
{
    if ((uint64_t)pos + 8 <= readable) { typedef uint64_t u64u __attribute__((aligned(1))); return *(const u64u*)(in + pos); }
    uint64_t v = 0;
    for (uint32_t i = 0; i < 8; i++) if ((uint64_t)pos + i < readable) v |= (uint64_t)in[pos + i] << (8 * i);
    return v;
}

// 16 payload bytes at `pos` (same rule)
__device__ __forceinline__ void pt_load16(const uint8_t* __restrict__ in, uint32_t pos, uint64_t readable, uint64_t& lo, uint64_t& hi)
{
    if ((uint64_t)pos + 16 <= readable) { typedef uint64_t u64u __attribute__((aligned(1))); lo = *(const u64u*)(in + pos); hi = *(const u64u*)(in + pos + 8); return; }
    lo = pt_load8(in, pos, readable); hi = pt_load8(in, pos + 8, readable);
}

// The index header is checked on the device (no host round trip on this path): usable at all, made for this geometry, its
// tables inside the buffer the caller named, its sequences inside the descriptor workspace the engine has.  flags[0] != 0
// sends the call to the generic decoder; flags[8] / flags[9] carry the entry and sequence counts to the kernels behind.
__global__ void k_check_index(const void* __restrict__ ix, uint64_t ix_size, uint32_t n_max, uint32_t chunks_per_block, uint32_t chunk_size,
                              uint64_t seq_cap, uint32_t* __restrict__ flags, const ResultRec* __restrict__ res = nullptr)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // the index is laid out for the frame's block count: the walk's (res), of which the host knows an upper bound (its own
    // index for a linked frame) or a claim (the trailer's: a frame with more blocks than that is not this index's frame)
    const uint32_t n_blocks = res ? res->n_blocks : n_max;
    const IxHeader hd = *(const IxHeader*)ix;
    const bool ok = (!res || res->status == ST_OK) && n_blocks != 0 && n_blocks <= n_max && hd.magic == IX_MAGIC && hd.n_blocks == n_blocks && hd.chunks_per_block == chunks_per_block && hd.stride == IX_STRIDE &&
                    hd.total_entries <= (uint64_t)n_blocks * chunks_per_block * ix_max_entries_per_chunk(chunk_size) &&
                    hd.total_seqs <= (uint64_t)hd.total_entries * (IX_STRIDE + 1) && hd.total_seqs <= seq_cap &&
                    ix_entries_at(n_blocks, chunks_per_block) + (uint64_t)hd.total_entries * sizeof(IxEntry) <= ix_size;
    flags[7] = n_blocks;
    flags[8] = ok ? hd.total_entries : 0u;
    flags[9] = ok ? hd.total_seqs : 0u;
    if (!ok) atomicOr(flags, 1u);
}

// `my_nseq` sequences of the payload in[0, csize) from `pos` on (output position `op`): their descriptors -> out[0 .. my_nseq).
// Input-side rules only.  Returns true when something is wrong; `pos` is left where the walk ended (the caller checks that it
// is where the next run starts).  `is_tail`: the run holds the block's last sequence.  `out_front`: output bytes in front of the
// block that a match of a linked frame may reach.
__device__ __forceinline__ bool parse_run(const uint8_t* __restrict__ in, uint32_t csize, uint64_t readable, uint32_t& pos, uint32_t op, uint32_t my_nseq,
                                          bool is_tail, SeqDesc* __restrict__ out, uint32_t linked, uint64_t out_front,
                                          uint32_t* opr = nullptr, uint32_t seq0 = 0, uint32_t omask = 0, uint32_t* op_end = nullptr)
{   // op_end: where the output stands behind the run's last sequence.  opr: (the feeder that parses for itself, fz_feeder_parse) every sequence's output position also goes into a ring in LDS
    bool bad = false;
    // the last four output ranges whose bytes are known to sit in the payload (literal runs, and matches that copied from such a
    // range): a match that lies inside one is DIRECT -- a plain copy out of the payload that needs no earlier output
    uint32_t r0o = 0, r0n = 0, r0p = 0, r1o = 0, r1n = 0, r1p = 0, r2o = 0, r2n = 0, r2p = 0, r3o = 0, r3n = 0, r3p = 0;
    auto remember = [&](uint32_t o, uint32_t n, uint32_t pp) { r3o = r2o; r3n = r2n; r3p = r2p; r2o = r1o; r2n = r1n; r2p = r1p; r1o = r0o; r1n = r0n; r1p = r0p; r0o = o; r0n = n; r0p = pp; };
    // One dependent load per sequence: the 16 bytes at the match offset also hold the match-length bytes, the NEXT token and
    // its literal-length bytes (a lane walks its sequences at memory latency, so the loads on that chain are what counts).
    // (The descriptor of sequence i is stored BEHIND the load of sequence i + 1: loads and stores share one counter and come back in issue
    // order, so a store in front of the load would put a write's round trip on the chain as well - in the feeder that parses for itself
    // that was 9.6 k cycles per sequence instead of one memory latency.)
    uint64_t w, w_hi;
    SeqDesc pend{0u, 0u, 0u, 0u};
    pt_load16(in, pos, readable, w, w_hi);
    for (uint32_t i = 0; i < my_nseq && !bad; i++) {
        if (pos >= csize) { bad = true; break; }
        const uint32_t token = (uint32_t)w & 0xFF;
        uint32_t lit = token >> 4, p = pos + 1;
        if (lit == 15) {
            const uint64_t x = w >> 8;
            const uint32_t f = (uint32_t)__builtin_ctzll(~x), k = f >> 3;
            if (k < 7) { lit += 255u * k + (uint32_t)((x >> (f & 56u)) & 0xFF); p += k + 1; }
            else { for (;;) { if (p >= csize || lit > (1u << 24)) { bad = true; break; } const uint32_t v = in[p++]; lit += v; if (v != 255) break; } if (bad) break; }
        }
        if (p > csize || lit >= (1u << 24)) { bad = true; break; }
        const uint32_t in_left = csize - p;
        uint32_t mlen = 0, off = 0;
        if (lit + 8 > in_left) {                                               // the block's last sequence
            if (lit != in_left || !is_tail || i + 1 != my_nseq) { bad = true; break; }
            pos = csize;
        } else {
            const uint32_t q = p + lit;
            uint64_t w2, w2_hi;
            pt_load16(in, q, readable, w2, w2_hi);
            __builtin_amdgcn_sched_barrier(0);
            if (i) out[i - 1] = pend;
            __builtin_amdgcn_sched_barrier(0);
            off = (uint32_t)w2 & 0xFFFF;
            if (off == 0) { bad = true; break; }
            mlen = token & 15; uint32_t pn = q + 2;
            bool reload = false;
            if (mlen == 15) {
                const uint64_t x = w2 >> 16;
                const uint32_t f = (uint32_t)__builtin_ctzll(~x), k = f >> 3;
                if (k < 6) { mlen += 255u * k + (uint32_t)((x >> (f & 56u)) & 0xFF); pn += k + 1; }
                else { reload = true; for (;;) { if (pn >= csize || mlen > (1u << 24)) { bad = true; break; } const uint32_t v = in[pn++]; mlen += v; if (v != 255) break; } if (bad) break; }
                if (pn + 4 >= csize) { bad = true; break; }
            }
            mlen += 4;
            pos = pn;
            if (reload) pt_load16(in, pos, readable, w, w_hi);
            else {                                                             // the next token is 2..8 bytes into what is already here
                const uint32_t sh = (pn - q) * 8u;
                w = sh >= 64 ? w2_hi : ((w2 >> sh) | (w2_hi << (64u - sh)));
            }
        }
        uint32_t f24 = off, mw = mlen;
        if (lit) remember(op, lit, p);
        if (mlen) {
            const uint32_t dm = op + lit;
            // a source in front of the block: only in a linked frame, and only as far as there is output before this block
            if (off > dm && (!linked || (uint64_t)(off - dm) > out_front)) { bad = true; break; }      // (hist0: output of an earlier call in front of this frame part)
            const uint32_t s0 = dm - off;
            uint32_t msrc = 0xFFFFFFFFu;
            if (mlen <= off && off <= dm) {
                if (s0 >= r0o && s0 + mlen <= r0o + r0n) msrc = r0p + (s0 - r0o);
                else if (s0 >= r1o && s0 + mlen <= r1o + r1n) msrc = r1p + (s0 - r1o);
                else if (s0 >= r2o && s0 + mlen <= r2o + r2n) msrc = r2p + (s0 - r2o);
                else if (s0 >= r3o && s0 + mlen <= r3o + r3n) msrc = r3p + (s0 - r3o);
            }
            if (msrc < IX_SRC_BIAS) { f24 = msrc + IX_SRC_BIAS; mw |= 0x80000000u; remember(dm, mlen, msrc); }
        }
        if (i && !mlen) out[i - 1] = pend;                                     // (the block's last sequence: no load was issued)
        pend = SeqDesc{p | ((f24 & 0xFFu) << 24), lit | (((f24 >> 8) & 0xFFu) << 24), op, mw | (((f24 >> 16) & 0x7Fu) << 24)};
        if (i + 1 == my_nseq) out[i] = pend;
        if (opr) opr[(seq0 + i) & omask] = op;
        op += lit + mlen;
    }
    if (op_end) *op_end = op;
    return bad;
}

// k_parse_indexed, one lane per index entry.  (The index structures and k_build_index live in encode.cuh: pass E2 writes the entries.)
// Input-side rules only (the feeder wave checks the ones that need output positions).  `flags[0]` is set when anything
// disagrees with the index: the caller then falls back to the generic decoder.  Every entry must end exactly where the next
// one starts and the first one of a block at payload byte 0, so the descriptors are a complete parse of the payload itself:
// a wrong index can make the call fall back, never change the bytes that come out.
__device__ __forceinline__ void parse_entry(const uint8_t* __restrict__ frame, uint64_t frame_cap, const BlockOut* __restrict__ table,
                                            const void* __restrict__ ix, uint32_t n_blocks, uint32_t n_entries,
                                            SeqDesc* __restrict__ desc, uint64_t desc_cap, uint32_t* __restrict__ flags, uint32_t gid, uint32_t linked, uint64_t hist0)
{
    const IxBlock* blocks = ix_blocks(ix);
    if (gid < n_blocks) {
        // the block table must hand out the descriptors and the entries without gaps or overlaps (every descriptor is then
        // written by exactly one lane), and a compressed block without entries cannot be decoded from the index
        const IxBlock bk = blocks[gid];
        const bool stored = (table[gid].word >> 31) != 0;
        bool wrong = stored ? (bk.nentries != 0 || bk.nseq != 0) : (bk.nentries == 0 || bk.nseq == 0);
        if (gid == 0) wrong |= bk.seq_base != 0 || bk.entry_base != 0;
        const uint64_t seq_end = (uint64_t)bk.seq_base + bk.nseq, ent_end = (uint64_t)bk.entry_base + bk.nentries;
        if (gid + 1 < n_blocks) wrong |= blocks[gid + 1].seq_base != seq_end || blocks[gid + 1].entry_base != ent_end;
        else wrong |= seq_end != desc_cap || ent_end != n_entries;
        if (wrong) atomicOr(flags, 1u);
    }
    if (gid >= n_entries) return;
    const IxEntry* entries = ix_entries(ix, n_blocks);
    const IxEntry me = entries[gid];
    const uint32_t b = me.nseq_blk >> 8, my_nseq = me.nseq_blk & 0xFFu;
    if (b >= n_blocks) { atomicOr(flags, 1u); return; }
    const IxBlock blk = blocks[b];
    const BlockOut e = table[b];
    const uint32_t csize = e.word & 0x7FFFFFFFu;
    bool bad = (e.word >> 31) != 0 || e.src_off + csize > frame_cap || me.in_off >= csize || my_nseq == 0 || gid < blk.entry_base ||
               gid >= blk.entry_base + blk.nentries || (uint64_t)me.seq_off + my_nseq > blk.nseq || (uint64_t)blk.seq_base + blk.nseq > desc_cap;
    if (bad) { atomicOr(flags, 1u); return; }
    const bool is_head = gid == blk.entry_base, is_tail = gid + 1 == blk.entry_base + blk.nentries;
    const uint32_t stop = is_tail ? csize : entries[gid + 1].in_off;           // where the next entry of this block starts (or the payload ends)
    if (is_head && (me.in_off != 0 || me.out_pos != 0 || me.seq_off != 0)) bad = true;      // the entries must cover the payload from its first byte
    if (!is_tail && entries[gid + 1].seq_off != me.seq_off + my_nseq) bad = true;
    if (is_tail && me.seq_off + my_nseq != blk.nseq) bad = true;
    const uint8_t* in = frame + e.src_off;
    const uint64_t readable = frame_cap - e.src_off;
    SeqDesc* out = desc + blk.seq_base + me.seq_off;
    uint32_t pos = me.in_off;
    if (!bad) bad = parse_run(in, csize, readable, pos, me.out_pos, my_nseq, is_tail, out, linked, e.dst_off + hist0);
    if (!bad && pos != stop) bad = true;                                       // must end exactly where the next entry starts
    if (bad) atomicOr(flags, 1u);
}

__global__ __launch_bounds__(256) void k_parse_indexed(const uint8_t* __restrict__ frame, uint64_t frame_cap, const BlockOut* __restrict__ table,
                                                       const void* __restrict__ ix, uint32_t n_max,
                                                       SeqDesc* __restrict__ desc, uint32_t* __restrict__ flags, uint32_t linked, uint64_t hist0)
{
    if (*flags) return;
    const uint32_t n_blocks = flags[7] < n_max ? flags[7] : n_max;      // (k_check_index: the frame's block count)
    const uint32_t n_entries = flags[8];
    const uint64_t desc_cap = flags[9];
    const uint32_t n_lanes = n_entries > n_blocks ? n_entries : n_blocks;
    for (uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x; gid < n_lanes; gid += gridDim.x * blockDim.x)
        parse_entry(frame, frame_cap, table, ix, n_blocks, n_entries, desc, desc_cap, flags, gid, linked, hist0);
}


// ------------------------------------------------------------------------------------------------
// SELF-INDEXING: a linked frame that comes without an index (a foreign one - the reference's default output) is one serial
// chain for the window kernel (3.5 GiB/s).  But PARSING a block needs no history, so the index can be made here: a lane per
// block walks its payload twice - once to count sequences and output bytes, then (after a scan over the blocks has given
// every block its place) to write an entry every IX_STRIDE sequences - and the frame goes through the same kernels as
// one with the compressor's index (k_parse_indexed checks every entry against the payload as usual).  A lane walks at memory
// latency (~1.5 us per sequence), all blocks at once: 0.2 ms for 64 KiB blocks of 1 KiB sequences, ~12 ms for 4 MiB blocks.
// MODE 0: count (cnt[b], osz[b]); MODE 1: write the entries.  Anything odd sets flags[0]: the generic kernels then decode
// the frame and give the verdict.
template <int MODE>
__global__ __launch_bounds__(256) void k_selfindex_walk(const uint8_t* __restrict__ frame, uint64_t frame_cap, const BlockOut* __restrict__ table,
                                                        const ResultRec* __restrict__ res, uint32_t n_max, uint32_t* __restrict__ cnt,
                                                        uint32_t* __restrict__ osz, void* __restrict__ ix, uint32_t* __restrict__ flags,
                                                        const uint32_t* __restrict__ only_if = nullptr)
{
    if (res->status != ST_OK || (MODE == 1 && *flags)) return;
    if (only_if && *only_if == 0) return;                                // (k_density_probe: dense payloads go to k_selfindex_walk_wave)
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const BlockOut e = table[b];
    const uint32_t csize = e.word & 0x7FFFFFFFu;
    if (e.word >> 31) { if (MODE == 0) { cnt[b] = 0; osz[b] = csize; } return; }
    if (csize == 0 || e.src_off + csize > frame_cap) { atomicOr(flags, 1u); if (MODE == 0) { cnt[b] = 0; osz[b] = 0; } return; }
    const uint8_t* in = frame + e.src_off;
    const uint64_t readable = frame_cap - e.src_off;
    IxEntry* ent = nullptr;
    if (MODE == 1) { const IxBlock bk = ix_blocks((const void*)ix)[b]; ent = (IxEntry*)((uint8_t*)ix + ix_entries_at(n, ((const IxHeader*)ix)->chunks_per_block)) + bk.entry_base; }
    const uint32_t total = MODE == 1 ? cnt[b] : 0u;
    uint32_t pos = 0, op = 0, k = 0;
    bool bad = false;
    uint64_t w, w_hi;
    pt_load16(in, pos, readable, w, w_hi);
    for (;;) {
        if (pos >= csize) { bad = true; break; }
        if (MODE == 1 && (k % IX_STRIDE) == 0) {
            const uint32_t ns = total - k < IX_STRIDE ? total - k : IX_STRIDE;
            ent[k / IX_STRIDE] = IxEntry{pos, op, k, ns | (b << 8)};
        }
        const uint32_t token = (uint32_t)w & 0xFF;
        uint32_t lit = token >> 4, p = pos + 1;
        if (lit == 15) {
            const uint64_t x = w >> 8;
            const uint32_t f = (uint32_t)__builtin_ctzll(~x), kk = f >> 3;
            if (kk < 7) { lit += 255u * kk + (uint32_t)((x >> (f & 56u)) & 0xFF); p += kk + 1; }
            else { for (;;) { if (p >= csize || lit > (1u << 24)) { bad = true; break; } const uint32_t v = in[p++]; lit += v; if (v != 255) break; } if (bad) break; }
        }
        if (p > csize || lit >= (1u << 24)) { bad = true; break; }
        const uint32_t in_left = csize - p;
        k++;
        if (lit + 8 > in_left) { if (lit != in_left) bad = true; op += lit; break; }       // the block's last sequence
        const uint32_t q = p + lit;
        uint64_t w2, w2_hi;
        pt_load16(in, q, readable, w2, w2_hi);
        uint32_t mlen = token & 15, pn = q + 2;
        bool reload = false;
        if (mlen == 15) {
            const uint64_t x = w2 >> 16;
            const uint32_t f = (uint32_t)__builtin_ctzll(~x), kk = f >> 3;
            if (kk < 6) { mlen += 255u * kk + (uint32_t)((x >> (f & 56u)) & 0xFF); pn += kk + 1; }
            else { reload = true; for (;;) { if (pn >= csize || mlen > (1u << 24)) { bad = true; break; } const uint32_t v = in[pn++]; mlen += v; if (v != 255) break; } if (bad) break; }
        }
        op += lit + mlen + 4;
        pos = pn;
        if (op > (1u << 23)) { bad = true; break; }
        if (reload) pt_load16(in, pos, readable, w, w_hi);
        else { const uint32_t sh = (pn - q) * 8u; w = sh >= 64 ? w2_hi : ((w2 >> sh) | (w2_hi << (64u - sh))); }
    }
    if (bad || (MODE == 1 && k != total)) { atomicOr(flags, 1u); if (MODE == 0) { cnt[b] = 0; osz[b] = 0; } return; }
    if (MODE == 0) { cnt[b] = k; osz[b] = op; }
}

// The same walk with a WAVE per block and the lanes finding the tokens (decode.cuh: wave_decode_block_win without the copies): a
// window of 64 payload bytes per step, every lane reading its byte as a token, the scalar unit hopping from token to token, a
// prefix sum placing them - and the lanes whose sequence number is a multiple of IX_STRIDE writing their entry themselves.
// Tokens with more than one literal-length byte or with match-length bytes, and the block's last ~100 bytes, go one at a time,
// wave-uniformly.  Text in 64 KiB blocks: ~5000 sequences per block at ~1.3 us each for a lane (two passes: 13 ms per 256 MiB)
// against ~14 per window here.  Data with few sequences per block is the other way round - 65536 blocks of 64 sequences are 1024
// waves of lanes all in flight at once (0.3 ms) but eight rounds of waves here (2.4 ms) - so k_density_probe picks.  Same MODEs,
// same verdicts.
template <int MODE, int WAVES_PER_WG>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_selfindex_walk_wave(const uint8_t* __restrict__ frame, uint64_t frame_cap, const BlockOut* __restrict__ table,
                                                                           const ResultRec* __restrict__ res, uint32_t n_max, uint32_t* __restrict__ cnt,
                                                                           uint32_t* __restrict__ osz, void* __restrict__ ix, uint32_t* __restrict__ flags,
                                                                           const uint32_t* __restrict__ only_if = nullptr)
{
    if (res->status != ST_OK || (MODE == 1 && *flags)) return;
    if (only_if && *only_if == 0) return;                                // (k_density_probe: sparse payloads go to k_selfindex_walk, a lane per block)
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t b = uni(blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6)), lane = lane_id();
    if (b >= n) return;
    const BlockOut e = table[b];
    const uint32_t csize = e.word & 0x7FFFFFFFu;
    if (e.word >> 31) { if (MODE == 0) { cnt[b] = 0; osz[b] = csize; } return; }
    if (csize == 0 || e.src_off + csize > frame_cap) { atomicOr(flags, 1u); if (MODE == 0) { cnt[b] = 0; osz[b] = 0; } return; }
    const uint8_t* in = frame + e.src_off;
    const uint64_t readable = frame_cap - e.src_off;
    IxEntry* ent = nullptr;
    if (MODE == 1) { const IxBlock bk = ix_blocks((const void*)ix)[b]; ent = (IxEntry*)((uint8_t*)ix + ix_entries_at(n, ((const IxHeader*)ix)->chunks_per_block)) + bk.entry_base; }
    const uint32_t total = MODE == 1 ? cnt[b] : 0u;
    uint32_t pos = 0, op = 0, k = 0;
    bool bad = false;
    typedef uint32_t u32_ua1 __attribute__((aligned(1)));
    for (;;) {
        if (pos >= csize) { bad = true; break; }
        // ---- the lanes' path (see wave_decode_block_win): tokens with at most one literal-length byte, ending inside the window ----
        if (csize - pos >= 96u && op < (1u << 23)) {
            const uint32_t d = *(const u32_ua1*)(in + pos + lane);
            const uint32_t t = d & 0xFFu, litn = t >> 4, ml = t & 15u, e1 = (d >> 8) & 0xFFu;
            const uint32_t hdr = litn == 15u ? 2u : 1u, lit = litn == 15u ? 15u + e1 : litn;
            const bool easy = ml != 15u && !(litn == 15u && e1 == 255u) && lane + hdr + lit + 2u <= 64u;
            const uint32_t nx = easy ? lane + hdr + lit + 2u : 255u;
            uint64_t mask = 0;
            uint32_t s = 0, sp = 0, nn;
            do {
                nn = (uint32_t)__builtin_amdgcn_readlane((int)nx, (int)s);
                asm("s_bitset1_b64 %0, %1" : "+s"(mask) : "s"(s));
                sp = s; s = nn;
            } while (nn < 64u);
            if (nn > 64u) { mask &= ~(1ull << sp); s = sp; }
            if (mask) {
                const bool is_tok = (mask >> lane) & 1ull;
                const uint32_t tout = is_tok ? lit + ml + 4u : 0u;
                const uint32_t incl = dpp_incl_scan_add(tout), ex = incl - tout;
                const uint32_t rank = (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
                if (MODE == 1) {
                    const uint32_t kk = k + rank;
                    if (is_tok && (kk % IX_STRIDE) == 0 && kk < total) ent[kk / IX_STRIDE] = IxEntry{pos + lane, op + ex, kk, (total - kk < IX_STRIDE ? total - kk : IX_STRIDE) | (b << 8)};
                }
                k += (uint32_t)__builtin_popcountll(mask);
                op += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                pos += s;
                if (op > (1u << 23)) { bad = true; break; }
                continue;
            }
        }
        // ---- one sequence, wave-uniformly (every lane the same loads) ----
        if (MODE == 1 && (k % IX_STRIDE) == 0 && k < total) {
            const uint32_t ns = total - k < IX_STRIDE ? total - k : IX_STRIDE;
            ent[k / IX_STRIDE] = IxEntry{pos, op, k, ns | (b << 8)};
        }
        uint64_t w, w_hi;
        pt_load16(in, pos, readable, w, w_hi);
        const uint32_t token = (uint32_t)w & 0xFF;
        uint32_t lit = token >> 4, p = pos + 1;
        if (lit == 15) {
            const uint64_t x = w >> 8;
            const uint32_t f = (uint32_t)__builtin_ctzll(~x), kk = f >> 3;
            if (kk < 7) { lit += 255u * kk + (uint32_t)((x >> (f & 56u)) & 0xFF); p += kk + 1; }
            else { for (;;) { if (p >= csize || lit > (1u << 24)) { bad = true; break; } const uint32_t v = in[p++]; lit += v; if (v != 255) break; } if (bad) break; }
        }
        if (p > csize || lit >= (1u << 24)) { bad = true; break; }
        const uint32_t in_left = csize - p;
        k++;
        if (lit + 8 > in_left) { if (lit != in_left) bad = true; op += lit; break; }       // the block's last sequence
        const uint32_t q = p + lit;
        uint64_t w2, w2_hi;
        pt_load16(in, q, readable, w2, w2_hi);
        uint32_t mlen = token & 15, pn = q + 2;
        if (mlen == 15) {
            const uint64_t x = w2 >> 16;
            const uint32_t f = (uint32_t)__builtin_ctzll(~x), kk = f >> 3;
            if (kk < 6) { mlen += 255u * kk + (uint32_t)((x >> (f & 56u)) & 0xFF); pn += kk + 1; }
            else { for (;;) { if (pn >= csize || mlen > (1u << 24)) { bad = true; break; } const uint32_t v = in[pn++]; mlen += v; if (v != 255) break; } if (bad) break; }
        }
        op += lit + mlen + 4;
        pos = pn;
        if (op > (1u << 23)) { bad = true; break; }
    }
    if (bad || (MODE == 1 && k != total)) { atomicOr(flags, 1u); if (MODE == 0) { cnt[b] = 0; osz[b] = 0; } return; }
    if (MODE == 0) { cnt[b] = k; osz[b] = op; }
}

// One workgroup: exclusive scans over the blocks (sequences, entries, output bytes) -> the index's block table and header,
// and every block's place in the output.  flags[8] / flags[9] get the totals (the host reads them to size the workspaces).
__global__ __launch_bounds__(1024) void k_selfindex_scan(BlockOut* __restrict__ table, const ResultRec* __restrict__ res, uint32_t n_max,
                                                         const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ osz, void* __restrict__ ix,
                                                         uint32_t chunks_per_block, uint64_t dst_cap, uint32_t block_size, uint32_t* __restrict__ flags,
                                                         uint32_t strict = 0)
{
    __shared__ uint64_t s_a[1024], s_b[1024], s_c[1024];
    __shared__ uint64_t c_a, c_b, c_c;
    if (res->status != ST_OK) return;
    if (strict && *flags) return;                                        // (independent blocks: the table stays what the generic decoder needs)
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t t = threadIdx.x;
    IxBlock* blocks = ix_blocks(ix);
    // A thread takes K consecutive blocks.  First it only JUDGES while it sums them up: a block that decodes to more than block_size, a short block inside
    // the frame, an output that does not fit dst_cap, counts beyond 32 bits - any of these raises flags[0] and nothing is written, so the table the generic
    // decoders fall back on keeps the walk's capacity-clamped entries (a payload's claimed sizes never reach them).  Only a clean verdict lets it go over
    // its blocks again and write the index's block table and every block's place in the output.  (Round 4: one scan over the 1024 threads' sums between the
    // two; until then a scan per 1024 blocks and per pass - 133 us for the 16384 blocks of a GiB in 64 KiB blocks, the longest step of a foreign linked
    // frame's self-index.)
    const uint32_t K = (n + 1023u) / 1024u;
    const uint32_t b0 = t * K < n ? t * K : n, b1 = b0 + K < n ? b0 + K : n;
    uint64_t la = 0, lb = 0, lc = 0;
    bool odd = false;
    for (uint32_t b = b0; b < b1; b++) {
        const uint64_t va = cnt[b], vc = osz[b];
        odd |= vc > block_size || (b + 1 < n && vc != block_size);          // (short blocks inside a frame: the generic kernels)
        la += va; lb += (va + IX_STRIDE - 1) / IX_STRIDE; lc += vc;
    }
    if (odd) atomicOr(flags, 1u);
    s_a[t] = la; s_b[t] = lb; s_c[t] = lc;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint64_t aa = t >= off ? s_a[t - off] : 0, ab = t >= off ? s_b[t - off] : 0, ac = t >= off ? s_c[t - off] : 0;
        __syncthreads();
        s_a[t] += aa; s_b[t] += ab; s_c[t] += ac;
        __syncthreads();
    }
    if (t == 1023) {
        c_a = s_a[1023]; c_b = s_b[1023]; c_c = s_c[1023];
        if (c_c > dst_cap || c_a >= (1ull << 32) || c_b >= (1ull << 32)) atomicOr(flags, 1u);
    }
    __threadfence_block();
    __syncthreads();
    if (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;      // (uniform: every thread reads the word behind the barrier)
    {
        uint64_t ea = s_a[t] - la, eb = s_b[t] - lb, ec = s_c[t] - lc;      // what the blocks in front of mine add up to
        for (uint32_t b = b0; b < b1; b++) {
            const uint64_t va = cnt[b], vb = (va + IX_STRIDE - 1) / IX_STRIDE, vc = osz[b];
            blocks[b] = IxBlock{(uint32_t)ea, (uint32_t)va, (uint32_t)eb, (uint32_t)vb};
            table[b].dst_off = ec;
            table[b].dst_size = (uint32_t)vc;
            ea += va; eb += vb; ec += vc;
        }
    }
    if (t == 0) {
        *(IxHeader*)ix = IxHeader{IX_MAGIC, n, chunks_per_block, (uint32_t)c_a, (uint32_t)c_b, IX_STRIDE, 1u, 0u};
        flags[8] = (uint32_t)c_b; flags[9] = (uint32_t)c_a;
    }
}

// ------------------------------------------------------------------------------------------------
// One lane per sequence: where in the PAYLOAD do the bytes of my match come from?  The lane looks up the sequence that
// produced its first source byte (binary search over the block's descriptors, which are sorted by output position).  Inside
// that sequence's literal run: found.  Inside its match (not overlapping itself): the same question one offset further back,
// a few hops at most.  Anything else (a source that straddles two runs, run-length matches) stays a match for the chain.
// The descriptors are only read here; the answer goes to dsrc[i] (IX_NOT_DIRECT = none) and the feeder wave merges it in.
constexpr uint32_t IXR_HOPS = 6;
constexpr uint32_t IXL_PUB = 8;                                // set-aside destinations a block publishes for the block behind (more: that one waits for all of it)
constexpr uint32_t IX_NOT_DIRECT = 0xFFFFFFFFu;
// In a linked frame the source may start in the block before (offsets reach 64 KiB back): the search then runs over that
// block's descriptors - or, if that block is stored, its payload IS its output.  The answer is the payload position
// relative to the asking block's payload (negative for the block before) + IX_SRC_BIAS.
constexpr uint32_t IXT_FLAG = 24;                               // flags[IXT_FLAG] != 0: a dense frame, decoded by the tracers below (2: decided before k_resolve_direct, which was skipped)
constexpr uint32_t IXT_SAMPLE_ALL = 256;                         // frames of up to this many blocks count every block's first workgroup
constexpr uint32_t IXT_CHAIN = 32;                               // flags[32..63]: matches left on the chain (striped; summed into flags[10] by the gate)
constexpr uint32_t IXT_DECIDED = 25;                            // flags[IXT_DECIDED]: what the gate decided (kept for the host: the next call's hint)
// a tracer gives up (too deep, or something inconsistent).  If the resolved sources exist (gate value 1) the copier workgroups
// launched behind take the frame - they check everything themselves; otherwise the generic kernels do
__device__ __forceinline__ void ixt_fail(uint32_t* flags)
{
    if (__hip_atomic_load(flags + IXT_FLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2u) atomicOr(flags, 2u);
    else __hip_atomic_store(flags + IXT_FLAG, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void k_resolve_direct(void* __restrict__ ix, const BlockOut* __restrict__ table, const ResultRec* __restrict__ res,
                                                        uint32_t n_max, const SeqDesc* __restrict__ desc, uint32_t* __restrict__ dsrc,
                                                        uint32_t* __restrict__ flags, uint32_t count_it, uint32_t linked)
{
    if (res->status != ST_OK || *flags || flags[IXT_FLAG]) return;      // (dense by the early gate: nobody will ask for the resolved sources)
    const uint64_t desc_cap = flags[9];
    const uint32_t b = blockIdx.x;                             // (gridDim.y workgroups per block)
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    if (b >= n) return;
    const IxBlock* blocks = ix_blocks(ix);
    const IxBlock blk = blocks[b];
    if ((uint64_t)blk.seq_base + blk.nseq > desc_cap) return;
    const SeqDesc* bd0 = desc + blk.seq_base;
    const uint64_t my_pay = table[b].src_off;
    uint32_t chain_mine = 0;
    for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < blk.nseq; i += gridDim.y * 256) {
        const SeqDesc d = bd0[i];
        const uint32_t ml = d.w & 0xFFFFFFu;
        uint32_t found = IX_NOT_DIRECT;
        if (ml != 0 && !(d.w >> 31)) {
            const uint32_t off = (d.x >> 24) | ((d.y >> 24) << 8), dm = d.z + (d.y & 0xFFFFFFu);
            if (ml <= off) {                                    // (overlapping matches replicate their own output: never direct)
                int64_t s0 = (int64_t)dm - off;                 // source start, relative to the output of block cb
                uint32_t cb = b, hi = i;
                const SeqDesc* bd = bd0;
                int64_t pay = 0;                                // payload of block cb relative to mine
                for (uint32_t hop = 0; hop < IXR_HOPS; hop++) {
                    if (s0 < 0) {                               // in the block before (linked frames)
                        if (!linked || cb == 0) break;
                        cb--;
                        const BlockOut pe = table[cb];
                        s0 += pe.dst_size;
                        if (s0 < 0) { if (count_it & 1u) atomicAdd(flags + 12, 1u); break; }                      // further back than one block: leave it to the chain
                        pay = (int64_t)pe.src_off - (int64_t)my_pay;
                        if (pe.word >> 31) {                    // stored: output position == payload position
                            if ((uint64_t)s0 + ml <= pe.dst_size) { const int64_t v = pay + s0 + IX_SRC_BIAS; if (v >= 0 && v < (1 << 23)) found = (uint32_t)v; }
                            break;
                        }
                        const IxBlock pb = blocks[cb];
                        if (pb.nseq == 0 || (uint64_t)pb.seq_base + pb.nseq > desc_cap) break;
                        bd = desc + pb.seq_base; hi = pb.nseq - 1;
                    }
                    uint32_t lo = 0;                            // last sequence that starts at or before s0
                    const uint32_t key = (uint32_t)s0;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi + 1) >> 1;
                        if (bd[mid].z <= key) lo = mid; else hi = mid - 1;
                    }
                    const SeqDesc dj = bd[lo];
                    const uint32_t opj = dj.z, litj = dj.y & 0xFFFFFFu, mlj = dj.w & 0xFFFFFFu, dmj = opj + litj;
                    const uint32_t fj = (dj.x >> 24) | ((dj.y >> 24) << 8);
                    if (key < opj) break;
                    if ((uint64_t)key + ml <= dmj) {                                                               // in its literal run
                        const int64_t v = pay + (dj.x & 0xFFFFFFu) + (key - opj) + IX_SRC_BIAS;
                        if (v >= 0 && v < (1 << 23)) found = (uint32_t)v;
                        break;
                    }
                    if (key < dmj || (uint64_t)key + ml > (uint64_t)dmj + mlj) { if (count_it & 1u) atomicAdd(flags + 13, 1u); break; }                              // straddles
                    if (dj.w >> 31) {                                                                              // in a direct match (marked by the parse)
                        const int64_t v = pay + (int64_t)(fj | (((dj.w >> 24) & 0x7Fu) << 16)) + (key - dmj);      // (already biased)
                        if (v >= 0 && v < (1 << 23)) found = (uint32_t)v;
                        break;
                    }
                    if (fj == 0 || mlj > fj) { if (count_it & 1u) atomicAdd(flags + 14, 1u); break; }                                                                // in a run-length match
                    if ((count_it & 1u) && hop + 1 == IXR_HOPS) atomicAdd(flags + 15, 1u);
                    s0 -= fj; hi = lo;                                                                             // in a plain match: follow it (possibly into the block before)
                }
            }
            if (count_it & 1u) atomicAdd(flags + (found != IX_NOT_DIRECT ? 5 : 6), 1u);
            if ((count_it & 1u) && linked && found == IX_NOT_DIRECT && off > dm) atomicAdd(flags + 11, 1u);
        } else if ((count_it & 1u) && ml) atomicAdd(flags + 4, 1u);
        chain_mine += (ml != 0 && !(d.w >> 31) && found == IX_NOT_DIRECT) ? 1u : 0u;      // how long is the chain that stays?
        dsrc[blk.seq_base + i] = found;
    }
    {                                                           // (for whoever chooses between the copiers and the tracers: one atomic per wave, over 32 striped words)
        for (int o = 32; o; o >>= 1) chain_mine += __shfl_xor(chain_mine, o);
        // (a SAMPLE: the first of the gridDim.y workgroups of a block, of every eighth block when there are many - the gate scales it up;
        // an atomic from every wave of the grid cost the headline 0.14 ms)
        const bool sampled = blockIdx.y == 0 && (n <= IXT_SAMPLE_ALL || (b & 7u) == 0);
        if (sampled && (threadIdx.x & 63u) == 0 && chain_mine) atomicAdd(flags + IXT_CHAIN + ((b * 4 + (threadIdx.x >> 6)) & 31u), chain_mine);
    }
}

// ------------------------------------------------------------------------------------------------
// DENSE frames (most sequences end in a match that is not direct: text).  The copier workgroups walk such a block as the one
// chain it is, ~0.35 us per sequence.  But every output byte originates in a literal: follow a match byte back through the
// descriptors - source position -> the sequence that produced it -> its literal run (done) or its match (one more hop) - and
// the whole frame is gathers, no chain.  A thread per 16 output bytes traces the first byte, gets the origin and how far the
// bytes behind it share it (the shortest remainder of a run along the way), copies that many, traces again.
//   k_dense_gate      decides (on the device) from k_resolve_direct's count of matches left on the chain
//   k_build_postab    per block: output position / 64 -> sequence (so that a hop is a read and a few steps, not a 12-step binary
//                     search), and the check that the descriptors tile the block's output
//   k_trace_copy      the gathers
constexpr uint32_t IXT_MAX_HOPS = 4096;                         // then the frame goes to the generic kernels

// early != 0: before k_resolve_direct, from the sequence density alone (under 20 output bytes per sequence: text) - such a
// frame goes to the doubling kernels, which do not need the resolved sources, so k_resolve_direct is skipped (flag value 2)
__global__ void k_dense_gate(uint32_t* __restrict__ flags, uint32_t on, const BlockOut* __restrict__ table, const ResultRec* __restrict__ res, uint32_t n_max, uint32_t early, uint32_t resolve_gy)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (early) {
        uint32_t v = 0;
        if (on && !*flags && res->status == ST_OK) {
            const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
            const uint64_t total = n ? table[n - 1].dst_off + table[n - 1].dst_size : 0;
            if ((uint64_t)flags[9] * 20 > total) v = 2u;
        }
        flags[IXT_FLAG] = v; flags[IXT_DECIDED] = v;
        return;
    }
    if (flags[IXT_FLAG]) return;
    uint64_t chain = 0;
    for (uint32_t q = 0; q < 32; q++) chain += flags[IXT_CHAIN + q];
    const uint32_t nb = res->n_blocks < n_max ? res->n_blocks : n_max;
    chain *= (uint64_t)resolve_gy * (nb <= IXT_SAMPLE_ALL ? 1u : 8u);             // (k_resolve_direct counted a sample)
    flags[10] = chain > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)chain;
    const bool dense = !*flags && chain * 2 > flags[9];
    flags[IXT_FLAG] = (on && !*flags && (on > 1 || dense)) ? 1u : 0u;
    flags[IXT_DECIDED] = dense || flags[IXT_FLAG];             // (also when the tracers were not on offer this time: the host's hint for the next call)
}

__global__ __launch_bounds__(256) void k_build_postab(void* __restrict__ ix, const BlockOut* __restrict__ table, const ResultRec* __restrict__ res, uint32_t n_max,
                                                      const SeqDesc* __restrict__ desc, uint32_t* __restrict__ postab, uint32_t* __restrict__ flags)
{
    if (res->status != ST_OK || *flags || !flags[IXT_FLAG]) return;
    const uint32_t b = blockIdx.x;
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    if (b >= n) return;
    const BlockOut e = table[b];
    if (e.word >> 31) return;
    const IxBlock blk = ix_blocks((const void*)ix)[b];
    if ((uint64_t)blk.seq_base + blk.nseq > flags[9] || (e.dst_off & 63u)) { ixt_fail(flags); return; }
    const SeqDesc* bd = desc + blk.seq_base;
    uint32_t* T = postab + (e.dst_off >> 6);
    for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < blk.nseq; i += gridDim.y * 256) {
        const SeqDesc d = bd[i];
        const uint32_t op = d.z, end = op + (d.y & 0xFFFFFFu) + (d.w & 0xFFFFFFu);
        // the descriptors must tile [0, size): first at 0, each where the one before ends, the last at the block's size
        const bool okay = (i == 0 ? op == 0 : true) && (i + 1 < blk.nseq ? bd[i + 1].z == end : end == e.dst_size) && end >= op && end <= e.dst_size;
        if (!okay) { ixt_fail(flags); continue; }
        for (uint32_t k = (op + 63u) >> 6; (k << 6) < end; k++) T[k] = i;                 // I hold output positions 64k in [op, end)
    }
}

// Depth: a phrase of a text is a copy of its last occurrence, which is a copy of the one before ... - traced to the literal that
// is thousands of hops.  So a trace stays inside its REGION (64 KiB of output): a source in an earlier region is read from the
// output itself, once that region is complete (a count of bytes per region, added to by each workgroup behind a release; the
// workgroups of earlier regions have lower indexes, so they are running or done - the same forward-progress rule as
// k_copy_indexed's).  Regions of one block therefore finish in order, ~5 us each; independent blocks have their regions
// interleaved over the grid (region k of every block, then region k+1 ...) so that all blocks advance together.
constexpr uint32_t IXT_STACK = 6;                                // set-aside ranges per thread
constexpr uint32_t IXT_TB = 16;                                  // output bytes per thread
constexpr uint32_t IXT_WG_BYTES = 256 * IXT_TB;
constexpr uint32_t IXT_REGION_LOG = 12;                          // a region = a workgroup's bytes
constexpr uint32_t IXT_REGION = 1u << IXT_REGION_LOG;
static_assert(IXT_REGION == IXT_WG_BYTES, "one workgroup per region");

__global__ __launch_bounds__(256) void k_trace_copy(const uint8_t* __restrict__ frame, uint64_t frame_cap, uint8_t* dst, const BlockOut* __restrict__ table,
                                                    const ResultRec* __restrict__ res, uint32_t n_max, void* __restrict__ ix, const SeqDesc* __restrict__ desc,
                                                    const uint32_t* __restrict__ dsrc, const uint32_t* __restrict__ postab, uint32_t* flags,
                                                    uint32_t linked, uint32_t block_size, uint64_t hist0, uint32_t* region_cnt, uint32_t count_it, uint32_t n_wg)
{   // n_wg: 4 KiB pieces of output, a workgroup's work each; the grid strides over them (the kernel is launched for every linked frame and returns at once
    // unless the frame is dense: as a grid of a workgroup per piece that was 57 us of empty dispatches per GiB, a tenth of a sparse frame's decode)
    __shared__ uint32_t go;
    __shared__ uint64_t stk_g[IXT_STACK][256];                          // per thread: the ranges set aside where a trace split (see below)
    __shared__ uint32_t stk_s[IXT_STACK][256];
    if (threadIdx.x == 0) go = (res->status == ST_OK && !__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) && flags[IXT_FLAG]) ? 1u : 0u;
    __syncthreads();
    if (!go) return;
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    for (uint32_t wgi = blockIdx.x; wgi < n_wg; wgi += gridDim.x) {
    // which 4 KiB of the output is this workgroup's
    uint64_t wg_pos;
    if (linked || block_size <= IXT_REGION) wg_pos = (uint64_t)wgi * IXT_WG_BYTES;
    else {
        const uint32_t per_k = n_max * (IXT_REGION / IXT_WG_BYTES);
        const uint32_t k = wgi / per_k, rem = wgi % per_k;
        wg_pos = (uint64_t)(rem / (IXT_REGION / IXT_WG_BYTES)) * block_size + (uint64_t)k * IXT_REGION + (uint64_t)(rem % (IXT_REGION / IXT_WG_BYTES)) * IXT_WG_BYTES;
    }
    const bool have_dsrc = flags[IXT_FLAG] != 2u;                      // (2: k_resolve_direct was skipped)
    const uint64_t gpos = wg_pos + threadIdx.x * IXT_TB;
    const uint32_t my_region = (uint32_t)(wg_pos >> IXT_REGION_LOG);
    const uint32_t b = (uint32_t)(wg_pos / block_size);
    bool bad = false;
    uint32_t wg_bytes = 0;
    if (b < n) {
        const BlockOut e0 = table[b];
        if (e0.dst_off != (uint64_t)b * block_size) bad = true;                             // (all blocks but the last are full: checked when the index was made)
        else {
            const uint64_t wg_rel = wg_pos - e0.dst_off;
            wg_bytes = wg_rel >= e0.dst_size ? 0u : (e0.dst_size - wg_rel < IXT_WG_BYTES ? (uint32_t)(e0.dst_size - wg_rel) : IXT_WG_BYTES);
        }
        const uint32_t i0 = (uint32_t)(gpos - e0.dst_off);
        const IxBlock* blocks = ix_blocks((const void*)ix);
        uint8_t* out = dst + gpos;
        const uint32_t want = (bad || i0 >= e0.dst_size) ? 0u : (e0.dst_size - i0 < IXT_TB ? e0.dst_size - i0 : IXT_TB);
        uint32_t cb_have = 0xFFFFFFFFu;                                 // (the block whose table entries are in registers)
        BlockOut e = e0;
        IxBlock blk = {0, 0, 0, 0};
        uint32_t ready_region = 0xFFFFFFFFu;                            // the last region this thread has seen complete
        // One loop, one hop per turn: where does output byte gpos + got come from, and how many bytes behind it come from right
        // behind that?  A lane whose trace ends copies the piece and starts on its next one in the same turn (with a loop
        // per piece the wave would wait for its deepest trace once per piece).
        uint32_t got = 0, hops = 0, sq_next = 0xFFFFFFFFu, n_turns = 0, n_pieces = 0, n_fromout = 0;
        uint64_t g = gpos;                                              // (position in the frame's output)
        uint32_t span = want;
        uint32_t stk_base = 0, stk_n = 0;
        const uint32_t tid = threadIdx.x;
        // A range travels together until part of it turns out to come from elsewhere (the next sequence, the end of a fold ...):
        // the rest is set aside AT THAT DEPTH and taken up when the front part is done - the hops down to there are shared.
        // (A few entries per thread; when they run out the oldest is dropped and those bytes start from the top again.)
        auto shorten = [&](uint32_t left) {
            if (left >= span) return;
            const uint32_t at = (stk_base + stk_n) % IXT_STACK;
            stk_g[at][tid] = g + left; stk_s[at][tid] = span - left;
            if (stk_n == IXT_STACK) stk_base = (stk_base + 1) % IXT_STACK; else stk_n++;
            span = left;
        };
        while (got < want && !bad) {
            const uint8_t* origin = nullptr;
            int kind = 0;                                               // 1: in the payload, 2: in the output (earlier region or call)
            bool moved = false;
            n_turns++;
            do {
                const uint32_t cb = (uint32_t)(g / block_size);
                if (cb != cb_have) { e = table[cb]; blk = blocks[cb]; cb_have = cb; }
                const uint32_t j = (uint32_t)(g - e.dst_off);
                if (e.dst_off != (uint64_t)cb * block_size || j >= e.dst_size) break;
                if (e.word >> 31) { shorten(e.dst_size - j); origin = frame + e.src_off + j; kind = 1; break; }
                const SeqDesc* bd = desc + blk.seq_base;
                // the sequence that holds j: the table says where to start; four descriptors at once (one latency), the last
                // of them that begins at or before j is it - or, if j is behind its end, the search goes on next turn
                uint32_t sq = sq_next != 0xFFFFFFFFu ? sq_next : postab[(e.dst_off >> 6) + (j >> 6)];
                sq_next = 0xFFFFFFFFu;
                if (sq >= blk.nseq) break;
                const uint32_t last = blk.nseq - 1;
                const SeqDesc c0 = bd[sq], c1 = bd[sq + 1 < last ? sq + 1 : last], c2 = bd[sq + 2 < last ? sq + 2 : last], c3 = bd[sq + 3 < last ? sq + 3 : last];
                SeqDesc d = c0;
                const uint32_t sq0 = sq;
                if (c1.z <= j) { d = c1; sq = sq0 + 1 < last ? sq0 + 1 : last; }
                if (c2.z <= j) { d = c2; sq = sq0 + 2 < last ? sq0 + 2 : last; }
                if (c3.z <= j) { d = c3; sq = sq0 + 3 < last ? sq0 + 3 : last; }
                const uint32_t lit = d.y & 0xFFFFFFu, ml = d.w & 0xFFFFFFu, op = d.z, dm = op + lit;
                const uint32_t f24 = (d.x >> 24) | ((d.y >> 24) << 8) | (((d.w >> 24) & 0x7Fu) << 16);
                if (j < op) break;
                if (j >= dm + ml) { if (sq >= last) break; sq_next = sq + 1; moved = true; break; }
                if (j < dm) {                                          // a literal: the origin
                    shorten(dm - j);
                    origin = frame + e.src_off + (d.x & 0xFFFFFFu) + (j - op); kind = 1;
                    break;
                }
                const uint32_t r = j - dm;
                const uint32_t ds = (d.w >> 31) ? f24 : (have_dsrc ? dsrc[blk.seq_base + sq] : IX_NOT_DIRECT);
                if (ds < (1u << 23)) {                                 // a direct match: its bytes are in the payload
                    shorten(ml - r);
                    origin = frame + e.src_off + ((int64_t)ds - (int64_t)IX_SRC_BIAS) + r; kind = 1;
                    break;
                }
                const uint32_t off = f24 & 0xFFFFu;
                if (off == 0) break;
                uint32_t rr = r, left = ml - r;
                if (off <= r) rr = r % off;                            // the match replicates its own output: fold
                if (off < ml && off - rr < left) left = off - rr;      // (a folded run repeats every `off` bytes)
                shorten(left);
                const uint64_t back = (uint64_t)off + (r - rr);
                if (back > g) {                                        // before this call's output: the history (linked frames)
                    if (!linked || back - g > hist0) break;
                    if (back - g < span) shorten((uint32_t)(back - g));
                    origin = dst - (back - g); kind = 2;
                    break;
                }
                if (!linked && g - back < e.dst_off) break;            // (independent blocks: nothing in front)
                g -= back;
                moved = true;
                const uint32_t rg = (uint32_t)(g >> IXT_REGION_LOG);
                if (rg != my_region) {                                 // in an earlier region: if that one is complete, its bytes; else on through it
                    if (rg > my_region) { moved = false; break; }
                    if (rg == ready_region || __hip_atomic_load(region_cnt + rg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= IXT_REGION) {
                        if (rg != ready_region) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); ready_region = rg; }
                        const uint32_t to_end = IXT_REGION - (uint32_t)(g & (IXT_REGION - 1));
                        shorten(to_end);
                        origin = dst + g; kind = 2;
                    }
                }
            } while (false);
            if (origin) {
                if (span == 0 || (kind == 1 && (origin < frame || origin + span > frame + frame_cap))) { bad = true; break; }
                for (uint32_t q = 0; q < span; q++) out[got + q] = origin[q];
                got += span; n_pieces++; n_fromout += kind == 2;
                hops = 0;
                if (stk_n) { stk_n--; const uint32_t at = (stk_base + stk_n) % IXT_STACK; g = stk_g[at][tid]; span = stk_s[at][tid]; }
                else { g = gpos + got; span = want - got; }
            } else if (!moved || ++hops >= IXT_MAX_HOPS) bad = true;
        }
        if (count_it) { atomicAdd((unsigned long long*)(flags + 26), (unsigned long long)n_turns); atomicAdd(flags + 28, n_pieces); atomicAdd(flags + 29, n_fromout); atomicMax(flags + 30, n_turns); }
    }
    if (bad) ixt_fail(flags);
    // this workgroup's bytes are in memory: count them into the region
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && wg_bytes) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(region_cnt + my_region, wg_bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The same origins by POINTER DOUBLING (4 bytes of scratch per output byte: 16 GiB for a 4 GiB frame, of the 288 there are;
// if that cannot be had, k_trace_copy does it).
// k_trace_copy walks ~70 hops per piece because every occurrence of a phrase copies the one before.  Here every output
// byte takes ONE hop (k_pd_init): bytes that come from the payload (literals, direct matches, stored blocks) or from an
// earlier call's output are written at once, the others leave the output position they copy in ptr[].  Then rounds of
// ptr[i] = ptr[ptr[i]] (k_pd_round, one launch each): a byte whose source was finished in an EARLIER round copies it (the
// launch boundary is what makes that byte visible), otherwise it adopts its source's source - the depth halves per round.
constexpr uint32_t IXP_DONE = 0xFFFFFFE0u;                       // ptr[i] >= IXP_DONE: finished in round ptr[i] - IXP_DONE (0 = by k_pd_init)
constexpr uint32_t IXP_ROUNDS = 12;
constexpr uint32_t IXP_JUMPS = 8;                                // steps along the trail per round
constexpr uint64_t IXP_MAX_SPAN = 0xFFF00000ull;                 // (positions are 32-bit words below IXP_DONE)
constexpr uint32_t IXP_STRIPES = 64;                             // the count of open bytes per round is kept in 64 words (a quarter of a million waves adding to one word take 3 ms)

__global__ __launch_bounds__(256) void k_pd_init(const uint8_t* __restrict__ frame, uint64_t frame_cap, uint8_t* dst, const BlockOut* __restrict__ table,
                                                 const ResultRec* __restrict__ res, uint32_t n_max, void* __restrict__ ix, const SeqDesc* __restrict__ desc,
                                                 const uint32_t* __restrict__ dsrc, const uint32_t* __restrict__ postab, uint32_t* flags,
                                                 uint32_t linked, uint32_t block_size, uint64_t hist0, uint32_t* __restrict__ ptr, uint32_t* __restrict__ remaining)
{
    if (res->status != ST_OK || *flags || !flags[IXT_FLAG]) return;
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const bool have_dsrc = flags[IXT_FLAG] != 2u;                      // (2: k_resolve_direct was skipped)
    const uint64_t gpos = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * IXT_TB;
    const uint32_t b = (uint32_t)(gpos / block_size);
    if (b >= n) return;
    const BlockOut e = table[b];
    if (e.dst_off != (uint64_t)b * block_size) { ixt_fail(flags); return; }            // (all blocks but the last are full: checked when the index was made)
    const uint32_t i0 = (uint32_t)(gpos - e.dst_off);
    if (i0 >= e.dst_size) return;
    const uint32_t want = e.dst_size - i0 < IXT_TB ? e.dst_size - i0 : IXT_TB;
    const IxBlock blk = ix_blocks((const void*)ix)[b];
    const SeqDesc* bd = desc + blk.seq_base;
    uint8_t* out = dst + gpos;
    uint32_t* pout = ptr + gpos;
    uint32_t got = 0, open = 0;
    bool bad = false;
    if (e.word >> 31) {                                                // a stored block: its payload is its output
        for (uint32_t q = 0; q < want; q++) { out[q] = frame[e.src_off + i0 + q]; pout[q] = IXP_DONE; }
        return;
    }
    if (blk.nseq == 0) { ixt_fail(flags); return; }
    const uint32_t last = blk.nseq - 1;
    uint32_t sq = postab[(e.dst_off >> 6) + (i0 >> 6)];
    while (got < want && !bad) {
        const uint32_t j = i0 + got;
        if (sq > last) { bad = true; break; }
        SeqDesc d = bd[sq];
        while (sq < last && j >= d.z + (d.y & 0xFFFFFFu) + (d.w & 0xFFFFFFu)) d = bd[++sq];        // (sequences are ~13 bytes on such data: a step or two)
        const uint32_t lit = d.y & 0xFFFFFFu, ml = d.w & 0xFFFFFFu, op = d.z, dm = op + lit;
        const uint32_t f24 = (d.x >> 24) | ((d.y >> 24) << 8) | (((d.w >> 24) & 0x7Fu) << 16);
        if (j < op || j >= dm + ml) { bad = true; break; }
        uint32_t span = want - got;
        const uint8_t* origin = nullptr;
        if (j < dm) {                                                  // a literal run
            if (dm - j < span) span = dm - j;
            origin = frame + e.src_off + (d.x & 0xFFFFFFu) + (j - op);
        } else {
            const uint32_t r = j - dm;
            const uint32_t ds = (d.w >> 31) ? f24 : (have_dsrc ? dsrc[blk.seq_base + sq] : IX_NOT_DIRECT);
            if (ml - r < span) span = ml - r;
            if (ds < (1u << 23)) origin = frame + e.src_off + ((int64_t)ds - (int64_t)IX_SRC_BIAS) + r;      // a direct match
            else {
                const uint32_t off = f24 & 0xFFFFu;
                if (off == 0) { bad = true; break; }
                uint32_t rr = r;
                if (off <= r) rr = r % off;                            // the match replicates its own output: fold
                if (off < ml && off - rr < span) span = off - rr;
                const uint64_t back = (uint64_t)off + (r - rr);
                const uint64_t g = gpos + got;
                if (back > g) {                                        // before this call's output: the history (linked frames)
                    if (!linked || back - g > hist0) { bad = true; break; }
                    if (back - g < span) span = (uint32_t)(back - g);
                    for (uint32_t q = 0; q < span; q++) { out[got + q] = (dst - (back - g))[q]; pout[got + q] = IXP_DONE; }
                } else {
                    if (!linked && g - back < e.dst_off) { bad = true; break; }
                    for (uint32_t q = 0; q < span; q++) pout[got + q] = (uint32_t)(g - back) + q;
                    open += span;
                }
                got += span;
                continue;
            }
        }
        if (origin < frame || origin + span > frame + frame_cap) { bad = true; break; }
        for (uint32_t q = 0; q < span; q++) { out[got + q] = origin[q]; pout[got + q] = IXP_DONE; }
        got += span;
    }
    if (bad) ixt_fail(flags);
    for (int o = 32; o; o >>= 1) open += __shfl_xor(open, o);
    if ((threadIdx.x & 63u) == 0 && open) atomicAdd(remaining + ((blockIdx.x * 4 + (threadIdx.x >> 6)) % IXP_STRIPES), open);
}

// one round; 4 output bytes per thread.  remaining[r][stripe] count the bytes still open after round r (r = 0: after k_pd_init)
__global__ __launch_bounds__(256) void k_pd_round(uint8_t* dst, uint32_t* ptr, const BlockOut* __restrict__ table, const ResultRec* __restrict__ res, uint32_t n_max,
                                                  uint32_t round, uint32_t* __restrict__ remaining, uint32_t* flags, uint8_t* __restrict__ grp = nullptr, uint32_t n_grp = 0)
{
    if (res->status != ST_OK || *flags || !flags[IXT_FLAG]) return;
    if (__ballot(remaining[(round - 1) * IXP_STRIPES + (threadIdx.x & 63u)] != 0) == 0) return;      // nothing was open after the round before
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    if (n == 0) return;
    const uint64_t total = table[n - 1].dst_off + table[n - 1].dst_size;
    // A wave's 256 bytes that had nothing open after the round before have nothing open now: one byte per wave and round says so
    // (grp[round parity][wave]), and the late rounds - a few per cent of the bytes still open, in few places - cost what they work on
    // instead of a pass over every pointer word (3 ms per round and 256 MiB whatever was open).
    const uint32_t gw = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (grp && round >= 2u && gw < n_grp && uni((uint32_t)grp[(size_t)((round - 1u) & 1u) * n_grp + gw]) == 0u) {
        if ((threadIdx.x & 63u) == 0) grp[(size_t)(round & 1u) * n_grp + gw] = 0;
        return;
    }
    // a wave takes 256 consecutive bytes, 64 at a time: neighbouring lanes hold neighbouring bytes, whose sources are mostly
    // neighbours too (one sector); the four passes are staged so that their scattered loads are in flight together
    const uint64_t base = ((uint64_t)blockIdx.x * 256 + (threadIdx.x & ~63u)) * 4 + (threadIdx.x & 63u);
    uint32_t open = 0;
    uint32_t cur[4];                                                  // where byte k's trail stands
    uint32_t st[4];                                                   // 0: already finished / not mine, 1: open, 2: cur is a finished byte to copy
    uint8_t by[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint64_t i = base + k * 64;
        cur[k] = i < total ? ptr[i] : IXP_DONE;
        st[k] = cur[k] < IXP_DONE ? 1u : 0u;
        if (st[k] && cur[k] >= i) { ixt_fail(flags); st[k] = 0; }  // (sources lie in front: anything else is a corrupted descriptor)
    }
    // IXP_JUMPS steps along the trail per round (each word read may already be this round's: any value it has held is valid)
#pragma unroll
    for (uint32_t jump = 0; jump < IXP_JUMPS; jump++) {
        uint32_t v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) v[k] = st[k] == 1u ? ptr[cur[k]] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            if (st[k] != 1u) continue;
            if (v[k] >= IXP_DONE) { st[k] = (v[k] - IXP_DONE < round) ? 2u : 3u; }      // finished earlier: copy it; in this very round: not visible yet, wait here
            else if (v[k] >= cur[k]) { ixt_fail(flags); st[k] = 0; }
            else cur[k] = v[k];
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) by[k] = st[k] == 2u ? dst[cur[k]] : (uint8_t)0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (st[k] == 0u) continue;
        const uint64_t i = base + k * 64;
        if (st[k] == 2u) { dst[i] = by[k]; ptr[i] = IXP_DONE + round; }
        else { ptr[i] = cur[k]; open++; }
    }
    for (int o = 32; o; o >>= 1) open += __shfl_xor(open, o);
    if ((threadIdx.x & 63u) == 0 && open) atomicAdd(remaining + round * IXP_STRIPES + ((blockIdx.x * 4 + (threadIdx.x >> 6)) % IXP_STRIPES), open);
    if (grp && (threadIdx.x & 63u) == 0 && gw < n_grp) grp[(size_t)(round & 1u) * n_grp + gw] = open ? 1 : 0;
}

__global__ void k_pd_verdict(uint32_t* flags, const uint32_t* __restrict__ remaining)
{
    if (blockIdx.x == 0 && threadIdx.x < IXP_STRIPES && flags[IXT_FLAG] && !*flags && remaining[IXP_ROUNDS * IXP_STRIPES + threadIdx.x] != 0) ixt_fail(flags);      // deeper than 2^16: the generic kernels
}

// ------------------------------------------------------------------------------------------------
// One workgroup per block: the fused decoder's copier waves (decode_fused.cuh), fed from the descriptor array instead of
// by a parser wave.
template <class C>
__global__ __launch_bounds__(64 * C::WAVES, FZ_FED_OCC) void k_copy_indexed(const uint8_t* __restrict__ frame, uint8_t* dst, BlockOut* __restrict__ table,
                                                                   const ResultRec* __restrict__ res, uint32_t n_max, void* __restrict__ ix,
                                                                   const SeqDesc* __restrict__ desc, const uint32_t* __restrict__ dsrc, uint32_t* __restrict__ flags,
                                                                   unsigned long long* prof, uint32_t linked, uint32_t* __restrict__ done, uint32_t group, uint64_t hist0,
                                                                   uint64_t wait_ticks)
{   // group: consecutive blocks per workgroup (1 except for small blocks of a linked frame, see below)
    __shared__ FzShared<C> sh;
    if (res->status != ST_OK || *flags || flags[IXT_FLAG]) return;           // index unusable: the generic kernel launched behind does the work; dense: traced
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t g = blockIdx.x, tid = threadIdx.x;
    const uint32_t b0 = g * group;
    if (b0 >= n) return;
    const uint32_t b1 = (b0 + group < n) ? b0 + group : n;
    // (A linked frame with most of its matches on the chain runs its groups one after the other - each waits for the one in front.
    // That is still 1.4-2x faster than the window kernel on such data (text 64 MiB: 2.5 vs 3.4 s; `datagen.structured` 48 MiB:
    // 0.15 vs 0.30 s), so there is no gate any more; LZ4F_MI355X_CHAIN_GATE=n leaves frames with more than 1/n of their
    // sequences on the chain to decode_linked.cuh.)
    if (linked > 1u && (uint64_t)flags[10] * (linked >> 1) > flags[9]) { if (g == 0 && tid == 0) atomicOr(flags, 4u); return; }
    if (!linked) {                                                           // (group == 1)
        const uint32_t b = b0;
        const BlockOut e = table[b];
        const uint32_t csz = e.word & 0x7FFFFFFFu;
        int32_t got;
        if (e.word >> 31) {                                                  // stored block: all waves copy a slice
            got = -2;
            if (csz <= e.dst_size) {
                const uint32_t per = (((csz + C::WAVES - 1) / C::WAVES) + 15) & ~15u;
                const uint32_t a = (tid >> 6) * per;
                if (a < csz) wave_copy_disjoint(dst + e.dst_off + a, frame + e.src_off + a, (csz - a < per) ? csz - a : per);
                got = (int32_t)csz;
            }
        } else {
            const IxBlock blk = ix_blocks(ix)[b];
            got = fz_decode_block<C, true>(sh, frame + e.src_off, csz, dst + e.dst_off, e.dst_size, 0, frame, prof, desc + blk.seq_base, blk.nseq,
                                           dsrc ? dsrc + blk.seq_base : nullptr, 0ull, nullptr, 0u, true);
        }
        // A block that did not come out (descriptors that do not tile the output, a bad offset, no room) is left as it was and
        // the generic kernel launched behind decodes the frame again: its verdict is the one the caller gets.
        if (tid == 0) { if (got < 0) atomicOr(flags, 2u); else table[b].dst_size = (uint32_t)got; }
        return;
    }
    // ---- linked frame ----
    // A workgroup decodes `group` consecutive blocks one after the other; matches may reach into the output in front of a block,
    // direct matches into the payload in front of it.  In front of the group's first block that output is another workgroup's:
    // matches that read it (and whatever reads what they will write) are SET ASIDE (fz_copier) - the list lives through the whole
    // group, rebased from block to block - and replayed at the end, when the group in front says so:
    //   done[g]: 0 running, 3 main pass done (its set-aside destinations are published; group == 1 only), 1 all done, 2 failed
    // A set-aside match needs bytes of the group in front.  They are final once that one's main pass is over - unless they are among
    // the bytes it set aside itself; only then does this one wait for all of it, so big blocks do not form a chain.  Why groups:
    // every word left for the neighbour needs an agent-scope release, and with the neighbour possibly on another XCD that fence
    // writes this XCD's L2 back (~65-130 ns each at the level of the whole GPU: 65536 single blocks of 64 KiB cost 7 ms of them).
    bool failed = false;
    uint32_t last_size = 0, own_front = 0;
    BlockOut e{};
    if (tid == 0) { sh.wait_lo = (uint32_t)wait_ticks; sh.wait_hi = (uint32_t)(wait_ticks >> 32); }      // (read by fz_copier, behind fz_decode_block's barrier)
    for (uint32_t b = b0; b < b1 && !failed; b++) {
        e = table[b];
        const uint32_t csz = e.word & 0x7FFFFFFFu;
        if (b > b0) {                                                        // the list of set-aside matches moves on with the block base
            __syncthreads();
            if (tid < FZ_PEND && tid < sh.pend_n) sh.pend_dst[tid] -= last_size;
            __syncthreads();
        }
        int32_t got;
        if (e.word >> 31) {
            got = -2;
            if (csz <= e.dst_size) {
                const uint32_t per = (((csz + C::WAVES - 1) / C::WAVES) + 15) & ~15u;
                const uint32_t a = (tid >> 6) * per;
                if (a < csz) wave_copy_disjoint(dst + e.dst_off + a, frame + e.src_off + a, (csz - a < per) ? csz - a : per);
                got = (int32_t)csz;
                if (b == b0) { __syncthreads(); if (tid == 0) { sh.pend_n = 0; sh.prev_ready = 0; sh.status = 0; } __syncthreads(); }   // (a stored block sets nothing aside)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            const IxBlock blk = ix_blocks(ix)[b];
            got = fz_decode_block<C, true>(sh, frame + e.src_off, csz, dst + e.dst_off, e.dst_size, e.dst_off + hist0, frame, prof, desc + blk.seq_base, blk.nseq,
                                           dsrc ? dsrc + blk.seq_base : nullptr, e.src_off, g > 0 ? done + (g - 1) : nullptr, own_front, b == b0);
        }
        if (got < 0) failed = true;
        else { if (tid == 0) table[b].dst_size = (uint32_t)got; last_size = (uint32_t)got; own_front = own_front + last_size < (1u << 30) ? own_front + last_size : (1u << 30); }
    }
    __syncthreads();
    const uint32_t np = failed ? 0u : uni(sh.pend_n);
    uint32_t* pcnt = done + n_max;
    uint32_t* prange = done + 2 * (size_t)n_max + (size_t)g * 2 * IXL_PUB;
    if (np) {
        // (state 3 is one more release fence: worth it for big blocks, where every block sets something aside and the replays would
        // otherwise form a chain through the whole frame; groups of small blocks just wait for all of the group in front)
        if (group == 1 && e.dst_size >= (1u << 20)) {
            if (np <= IXL_PUB && tid < np) { prange[2 * tid] = sh.pend_dst[tid]; prange[2 * tid + 1] = sh.pend_dst[tid] + sh.pend_len[tid]; }
            __syncthreads();
            if (tid == 0) { pcnt[g] = np <= IXL_PUB ? np : 0xFFFFFFFFu; __threadfence(); __hip_atomic_store(done + g, 3u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
        }
        if ((tid >> 6) == 0) {                                               // wave 0 replays
            const uint32_t* pd = done + (g - 1);                             // (g > 0: a group without one in front sets nothing aside)
            uint32_t v = 0;
            auto poll = [&](bool all) {
                // (relaxed polls, one acquire at the end: an acquire per poll would empty this CU's vector cache every time)
                for (const uint64_t t0 = __builtin_amdgcn_s_memrealtime(); !fz_wait_expired(t0, wait_ticks);) {
                    v = __hip_atomic_load(pd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v == 1u || v == 2u || (v == 3u && !all)) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); return; }
                    __builtin_amdgcn_s_sleep(8);
                }
                v = 2u;
            };
            poll(false);
            if (prof && lane_id() == 0) { atomicAdd(flags + 22, 1u); if (v == 1u) atomicAdd(flags + 23, 1u); }
            if (v == 3u) {                                                   // (group == 1) do my sources touch what it still has to write?
                const uint32_t pc = pcnt[g - 1];
                bool hit = pc == 0xFFFFFFFFu;
                if (!hit) {
                    const int32_t psz = (int32_t)table[b0 - 1].dst_size;
                    const uint32_t* pr = done + 2 * (size_t)n_max + (size_t)(g - 1) * 2 * IXL_PUB;
                    for (uint32_t q = 0; q < np && !hit; q++) {
                        const int32_t s0 = (int32_t)sh.pend_dst[q] - (int32_t)sh.pend_off[q];
                        if (s0 >= 0) continue;
                        const int32_t lo = psz + s0, hi = psz + ((s0 + (int32_t)sh.pend_len[q] < 0) ? s0 + (int32_t)sh.pend_len[q] : 0);
                        for (uint32_t k = 0; k < pc; k++) hit |= lo < (int32_t)pr[2 * k + 1] && hi > (int32_t)pr[2 * k];
                    }
                }
                if (prof && lane_id() == 0) { atomicAdd(flags + 20, 1u); if (hit) atomicAdd(flags + 21, 1u); }
                if (hit) poll(true);
            }
            if (v == 2u) sh.status = -1;
            else {
                uint8_t* out = dst + e.dst_off;                              // (positions on the list are relative to the group's last block)
                for (uint32_t q = 0; q < np; q++) wave_copy_match(out + (int32_t)sh.pend_dst[q], sh.pend_off[q], sh.pend_len[q]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        failed = failed || (int32_t)uni((uint32_t)sh.status) < 0;
    }
    if (tid == 0) {
        if (failed) atomicOr(flags, 2u);
        // (this release is what a linked frame of small blocks costs, hence the groups)
        __threadfence();
        __hip_atomic_store(done + g, failed ? 2u : 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}


// ------------------------------------------------------------------------------------------------
// Round 4: ONE kernel for independent blocks that come with an index (the compressor's out of the trailer, or - decode_spx.cuh - the
// stretch points a foreign frame was cut into).  k_parse_indexed and k_resolve_direct were kernels of their own in front of the
// copy (0.23 + 0.12 ms of the bench decode's 1.78, and three launch boundaries); here the workgroup's FIRST WAVE does both for its
// block while the copier waves already move bytes:
//   * parse: the lanes take the block's next runs (index entries of <= 16 sequences, or stretches), each walks its run with
//     parse_run() - the same code, the same rules - and writes the descriptors to the workspace in HBM; every sequence's output
//     position also goes into a ring in LDS (the stage area the parser wave of the generic decoder uses: 4096 words, 1024 in the
//     4-wave shape).  A round takes as many runs as hold at most half a ring of sequences.
//   * resolve, per ring slot of 64 descriptors: a lane whose match is not direct yet looks for the sequence that produced its first
//     source byte - a binary search over the ring (LDS), then ONE read of that descriptor (written by this wave a moment ago: L2) -
//     and follows plain matches a few hops, like k_resolve_direct.  What it finds is written back into the descriptor, so later
//     hops end at the first direct match.  A source older than the ring remembers (4096 sequences: more than 64 KiB of anything
//     that is not dense) stays on the chain.
//   * then the slot is checked (tiling, room, offsets) and published exactly as fz_feeder does.
// A block's feeder needs ~0.3 ms of its ~1.4; the copiers are never the ones waiting after the first round.  Anything odd sets
// flags bit 1 and the generic decoder launched behind decodes the frame again, as with k_copy_indexed.
template <class C> struct FzOpr { static constexpr uint32_t N = (2u * (C::STAGE + FZ_OVER) >= 16384u) ? 4096u : (2u * (C::STAGE + FZ_OVER) >= 8192u) ? 2048u : 1024u; };
static_assert(FzOpr<FzCfgS8>::N * 4 <= 2 * (FzCfgS8::STAGE + FZ_OVER) && FzOpr<FzCfgS4>::N * 4 <= 2 * (FzCfgS4::STAGE + FZ_OVER), "the ring of output positions lives in the stage area");

struct FzRun { uint32_t in_off, out_pos, seq_off, nseq, stop; bool tail, bad; };
// the runs of one block out of an index: its entries
struct FzRunsIx {
    const IxEntry* ent; uint32_t n, b, nseq_blk, csize;
    __device__ __forceinline__ uint32_t count() const { return n; }
    __device__ __forceinline__ FzRun get(uint32_t u) const
    {
        const IxEntry me = ent[u];
        FzRun r{me.in_off, me.out_pos, me.seq_off, me.nseq_blk & 0xFFu, 0u, u + 1 == n, false};
        r.bad = (me.nseq_blk >> 8) != b || r.nseq == 0 || me.in_off >= csize || (uint64_t)me.seq_off + r.nseq > nseq_blk;
        if (u == 0) r.bad |= (me.in_off | me.out_pos | me.seq_off) != 0;       // the entries must cover the payload from its first byte
        if (r.tail) { r.stop = csize; r.bad |= me.seq_off + r.nseq != nseq_blk; }
        else { const IxEntry nx = ent[u + 1]; r.stop = nx.in_off; r.bad |= nx.seq_off != me.seq_off + r.nseq; }
        return r;
    }
};

template <class C, class RUNS>
__device__ __forceinline__ void fz_feeder_parse(FzShared<C>& sh, const RUNS& runs, const uint8_t* __restrict__ in, uint32_t csize, uint64_t readable,
                                                SeqDesc* desc, uint32_t nseq, uint32_t cap, unsigned long long* prof, uint32_t round_max)
{
    constexpr uint32_t OPR = FzOpr<C>::N, OMASK = OPR - 1;
    const uint32_t ROUND_MAX = (round_max >= 17u && round_max < OPR / 2) ? round_max : OPR / 2;      // (a test switch, LZ4F_MI355X_FEED_ROUND: rounds so small that every stretch of a foreign frame goes in pieces - the `part` path below)
    unsigned long long t_parse = 0, t_res = 0, t_ring = 0, n_round = 0, t_load = 0;
    const unsigned long long t_begin = clock64();
    uint32_t* opr = (uint32_t*)&sh.stage[0][0];
    const uint32_t lane = lane_id();
    uint32_t per = (nseq + (C::WAVES - 1) - 1) / (C::WAVES - 1);
    per = per > 64 ? 64u : (per < 8 ? 8u : per);
    const uint32_t nslots = (nseq + per - 1) / per, nruns = runs.count();
    uint32_t expect = 0, status = nseq ? 0u : 1u, published = 0;
    uint32_t W = 0, u0 = 0;                                                  // sequences parsed so far (descriptors in HBM, positions in the ring); next run
    uint4 nxt = {0u, 0u, 0u, 0u}; bool nxt_ok = false;
    uint32_t part_left = 0, part_pos = 0, part_out = 0;                      // a run that goes in pieces: sequences left of it, where its next piece starts
    for (uint32_t slot = 0; slot < nslots && !status; slot++) {
        const uint32_t first = slot * per;
        const uint32_t count = (nseq - first < per) ? nseq - first : per;
        // ---- parse until this slot's sequences are there ----
        const unsigned long long z0 = clock64();
        while (W < first + count) {
            n_round++;
            if (u0 >= nruns) { status = 1; break; }
            const bool have = u0 + lane < nruns;
            FzRun r{0u, 0u, 0u, 0u, 0u, false, false};
            if (have) r = runs.get(u0 + lane);
            // (a run of more sequences than a round may hold - a stretch of a foreign frame's payload with short sequences, or one the
            // stitching thread walked itself - goes in pieces: lane 0 takes the next ROUND_MAX of it, alone, and the rest waits in `part`)
            if (part_left && lane == 0) { r.in_off = part_pos; r.out_pos = part_out; r.seq_off = W; r.nseq = part_left; }
            const uint32_t whole0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.nseq);
            const bool cut = whole0 > ROUND_MAX;                             // (wave-uniform)
            if (cut && lane == 0) r.nseq = ROUND_MAX;
            uint32_t tot;
            const uint32_t before = wave_excl_scan(have ? r.nseq : 0u, tot);
            const uint64_t refused = __ballot(have && lane != 0 && (cut || before + r.nseq > ROUND_MAX)), hv = __ballot(have);
            const uint32_t R = refused ? (uint32_t)__builtin_ctzll(refused) : (uint32_t)__builtin_popcountll(hv);      // (a prefix of the lanes)
            const bool mine = lane < R;
            bool bad = mine && (r.bad || r.nseq > ROUND_MAX || (lane == 0 && r.seq_off != W));
            uint32_t pos = r.in_off, op_end = 0;
            if (mine && !bad) {
                bad = parse_run(in, csize, readable, pos, r.out_pos, r.nseq, r.tail && !(cut && lane == 0), desc + r.seq_off, 0u, 0ull, opr, r.seq_off, OMASK, &op_end);
                if (!(cut && lane == 0)) bad |= pos != r.stop;               // must end exactly where the next run starts
            }
            if (__ballot(bad)) { status = 1; break; }
            W = (uint32_t)__builtin_amdgcn_readlane((int)(r.seq_off + r.nseq), (int)(R - 1));
            if (cut) {                                                       // (R == 1: the run stays the next one until all of it is parsed)
                part_left = whole0 - ROUND_MAX;
                part_pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos); part_out = (uint32_t)__builtin_amdgcn_readfirstlane((int)op_end);
            } else { part_left = 0; u0 += R; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (status) break;
        const unsigned long long z1 = clock64(); t_parse += z1 - z0;
        uint4 d = nxt;
        if (!nxt_ok && lane < count) d = ((const uint4*)desc)[first + lane];
        nxt_ok = false;
        if (slot + 1 < nslots) {                                             // the next slot's descriptors travel while this one is resolved, if they are parsed already
            const uint32_t nf = first + per, nc = (nseq - nf < per) ? nseq - nf : per;
            if (W >= nf + nc) { nxt = uint4{0u, 0u, 0u, 0u}; if (lane < nc) nxt = ((const uint4*)desc)[nf + lane]; nxt_ok = true; }
        }
        const unsigned long long z2 = clock64(); t_load += z2 - z1;
        // ---- resolve: where in the payload do my match's bytes come from? ----
        {
            const uint32_t ml = d.w & 0xFFFFFFu;
            uint32_t found = IX_NOT_DIRECT;
            if (lane < count && ml != 0 && !(d.w >> 31)) {
                const uint32_t off = (d.x >> 24) | ((d.y >> 24) << 8), dm = d.z + (d.y & 0xFFFFFFu);
                if (ml <= off && off <= dm) {                                // (overlapping matches replicate their own output: never direct)
                    uint32_t key = dm - off, hi = first + lane;
                    const uint32_t ring_lo = W > OPR ? W - OPR : 0u;
                    for (uint32_t hop = 0; hop < IXR_HOPS; hop++) {
                        if (hi < ring_lo) break;
                        const uint32_t olo = opr[ring_lo & OMASK], ohi = opr[hi & OMASK];
                        if (olo > key) break;                                // older than the ring remembers
                        // the last sequence that starts at or before key.  A guess from the average sequence length first, then outwards
                        // from it in doubling steps, then bisection: 3-4 LDS reads where sequences are of a size (12 for plain bisection)
                        uint32_t lo = ring_lo;                               // (opr[lo] <= key, opr[hi + 1] > key throughout)
                        if (ohi <= key) lo = hi;
                        else {
                            uint32_t g = lo + (uint32_t)((float)(key - olo) * (float)(hi - lo) / (float)(ohi - olo));
                            g = g >= hi ? hi - 1 : g;
                            uint32_t step = 1;
                            if (opr[g & OMASK] <= key) {
                                lo = g; hi = hi - 1;
                                while (lo + step <= hi) {
                                    if (opr[(lo + step) & OMASK] <= key) { lo += step; step <<= 1; }
                                    else { hi = lo + step - 1; break; }
                                }
                            } else {
                                hi = g - 1;                                  // (g > lo: opr[lo] <= key < opr[g])
                                while (hi + 1 >= lo + step) {
                                    const uint32_t t = hi + 1 - step;
                                    if (opr[t & OMASK] > key) { hi = t - 1; step <<= 1; }
                                    else { lo = t; break; }
                                }
                            }
                        }
                        while (lo < hi) {
                            const uint32_t mid = (lo + hi + 1) >> 1;
                            if (opr[mid & OMASK] <= key) lo = mid; else hi = mid - 1;
                        }
                        const SeqDesc dj = desc[lo];
                        const uint32_t opj = dj.z, litj = dj.y & 0xFFFFFFu, mlj = dj.w & 0xFFFFFFu, dmj = opj + litj;
                        const uint32_t fj = (dj.x >> 24) | ((dj.y >> 24) << 8);
                        if (key < opj) break;
                        if ((uint64_t)key + ml <= dmj) {                     // in its literal run
                            const uint32_t v = (dj.x & 0xFFFFFFu) + (key - opj) + IX_SRC_BIAS;
                            if (v < (1u << 23)) found = v;
                            break;
                        }
                        if (key < dmj || (uint64_t)key + ml > (uint64_t)dmj + mlj) break;      // straddles
                        if (dj.w >> 31) {                                    // in a direct match
                            const uint32_t v = (fj | (((dj.w >> 24) & 0x7Fu) << 16)) + (key - dmj);      // (already biased)
                            if (v < (1u << 23)) found = v;
                            break;
                        }
                        if (fj == 0 || mlj > fj || fj > key) break;          // in a run-length match
                        key -= fj; hi = lo;                                  // in a plain match: follow it
                    }
                }
            }
            if (found != IX_NOT_DIRECT) {
                d = uint4{(d.x & 0xFFFFFFu) | ((found & 0xFFu) << 24), (d.y & 0xFFFFFFu) | (((found >> 8) & 0xFFu) << 24), d.z,
                          (d.w & 0xFFFFFFu) | 0x80000000u | ((found >> 16) << 24)};
                ((uint4*)desc)[first + lane] = d;                            // (later hops end here)
            }
        }
        // ---- check and publish (as fz_feeder) ----
        const unsigned long long z3 = clock64(); t_res += z3 - z2;
        const uint32_t p = d.x & 0xFFFFFFu, lit = d.y & 0xFFFFFFu, op = d.z, ml = d.w & 0xFFFFFFu;
        const bool direct = (d.w >> 31) != 0;
        const uint32_t f24 = (d.x >> 24) | ((d.y >> 24) << 8) | (((d.w >> 24) & 0x7Fu) << 16);
        const uint64_t dm = (uint64_t)op + lit, end = dm + ml;
        const uint32_t prev_end = __shfl_up((uint32_t)end, 1);
        const bool last = first + lane + 1 == nseq;
        bool bad = op != (lane == 0 ? expect : prev_end) || (uint64_t)p + lit > csize || end > cap;
        if (last) bad |= ml != 0 || direct;
        else {
            bad |= ml < 4 || end + 5 > cap;
            const int64_t rel = (int64_t)f24 - (int64_t)FZ_SRC_BIAS;
            bad |= direct ? (rel < 0 || rel + (int64_t)ml > (int64_t)csize) : (f24 == 0 || f24 > 65535u || f24 > dm);
        }
        if (__ballot(lane < count && bad)) { status = 1; break; }
        expect = __shfl((uint32_t)end, (int)count - 1);
        const unsigned long long z4 = clock64();
        while (slot >= lds_peek(&sh.match_done) + C::RING && (int32_t)lds_peek((const uint32_t*)&sh.status) >= 0) __builtin_amdgcn_s_sleep(8);
        t_ring += clock64() - z4;
        if ((int32_t)lds_peek((const uint32_t*)&sh.status) < 0) { status = 1; break; }
        sh.ring[slot % C::RING][lane] = d;
        sh.slot_cnt[slot % C::RING] = count;
        if (slot + 1 == nslots) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_poke(&sh.produced, slot + 1);
        published = slot + 1;
    }
    if (prof && lane == 0) { const unsigned long long tt = clock64() - t_begin; atomicMax(prof + 66, tt); atomicAdd(prof + 67, tt); atomicAdd(prof + 68, t_ring); atomicAdd(prof + 73, t_parse); atomicAdd(prof + 74, t_res); }
    if (prof && blockIdx.x == 0 && lane == 0) { prof[0] = clock64() - t_begin; prof[1] = t_ring; prof[2] = nseq; prof[3] = t_parse; prof[4] = t_load; prof[5] = t_res; prof[6] = n_round; }
    if (!status && (W != nseq || u0 != nruns)) status = 1;                  // every run used, every sequence parsed
    sh.total_slots = status ? published : nslots;
    sh.last_count = status ? 64u : nseq - (nslots - 1) * per;
    sh.out_size = expect;
    sh.status = status ? -1 : 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_poke(&sh.finished, 1u);
    if (!status) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); lds_poke(&sh.produced, nslots); }
}

// the workgroup of fz_decode_block<C, true>, its first wave feeding itself
template <class C, class RUNS>
__device__ __forceinline__ int32_t fz_decode_block_self(FzShared<C>& sh, const RUNS& runs, const uint8_t* __restrict__ in, uint32_t csize, uint64_t readable,
                                                        uint8_t* out, uint32_t cap, const uint8_t* safe, unsigned long long* prof, SeqDesc* desc, uint32_t nseq, uint32_t round_max)
{
    const uint32_t wave = uni(threadIdx.x >> 6);
    __syncthreads();
    if (threadIdx.x == 0) { sh.produced = 0; sh.total_slots = 0; sh.last_count = 64; sh.finished = 0; sh.match_done = 0; sh.status = 0; sh.out_size = 0; sh.pend_n = 0; sh.prev_ready = 0; sh.own_front = 0; }
    if (threadIdx.x < C::RING) sh.slot_done[threadIdx.x] = 0;
    __syncthreads();
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        fz_feeder_parse<C, RUNS>(sh, runs, in, csize, readable, desc, nseq, cap, prof, round_max);
        __builtin_amdgcn_s_setprio(0);
    }
    else fz_copier<C, true>(sh, in, out, wave - 1, safe, prof, nullptr);
    __syncthreads();
    const int32_t st = (int32_t)uni((uint32_t)sh.status);
    const uint32_t osz = uni(sh.out_size);
    return st < 0 ? -1 : (int32_t)osz;
}

// RSRC: what gives a block its runs (FzSrcIx below; FzSrcSpx in decode_spx.cuh)
struct FzSrcIx {
    const void* ix; uint32_t n_blocks;
    __device__ __forceinline__ bool make(uint32_t b, const IxBlock& blk, uint32_t csize, uint32_t n_entries /* k_check_index: inside the buffer */, FzRunsIx& r) const
    {
        if ((uint64_t)blk.entry_base + blk.nentries > n_entries || blk.nentries == 0) return false;
        r = FzRunsIx{ix_entries(ix, n_blocks) + blk.entry_base, blk.nentries, b, blk.nseq, csize};
        return true;
    }
    typedef FzRunsIx Runs;
};

template <class C, class RSRC>
__global__ __launch_bounds__(64 * C::WAVES, FZ_FED_OCC) void k_copy_selffed(const uint8_t* __restrict__ frame, uint64_t frame_cap, uint8_t* dst, BlockOut* __restrict__ table,
                                                                   const ResultRec* __restrict__ res, uint32_t n_max, const void* __restrict__ ix,
                                                                   SeqDesc* desc, uint32_t* __restrict__ flags, unsigned long long* prof, RSRC rsrc, uint32_t round_max)
{
    __shared__ FzShared<C> sh;
    if (res->status != ST_OK || *flags || flags[IXT_FLAG]) return;           // index unusable (k_check_index): the generic kernel launched behind does the work
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const unsigned long long wg_t0 = prof ? __builtin_amdgcn_s_memrealtime() : 0ull, wg_c0 = prof ? clock64() : 0ull;
    if (b >= n) return;
    const uint64_t desc_cap = flags[9];
    const BlockOut e = table[b];
    const uint32_t csz = e.word & 0x7FFFFFFFu;
    const IxBlock* blocks = ix_blocks(ix);
    const IxBlock blk = blocks[b];
    // the index's block table must hand out the descriptors and the entries without gaps or overlaps (every descriptor is then
    // written by exactly one workgroup), and a compressed block without sequences cannot be decoded from the index
    const bool stored = (e.word >> 31) != 0;
    bool wrong = stored ? (blk.nentries != 0 || blk.nseq != 0) : blk.nseq == 0;
    if (b == 0) wrong |= blk.seq_base != 0 || blk.entry_base != 0;
    const uint64_t seq_end = (uint64_t)blk.seq_base + blk.nseq, ent_end = (uint64_t)blk.entry_base + blk.nentries;
    if (b + 1 < n) { const IxBlock nb = blocks[b + 1]; wrong |= nb.seq_base != seq_end || nb.entry_base != ent_end; }
    else wrong |= seq_end != desc_cap;
    wrong |= e.src_off + csz > frame_cap;
    int32_t got = -1;
    if (!wrong) {
        if (stored) {                                                        // stored block: all waves copy a slice
            got = -2;
            if (csz <= e.dst_size) {
                const uint32_t per = (((csz + C::WAVES - 1) / C::WAVES) + 15) & ~15u;
                const uint32_t a = (tid >> 6) * per;
                if (a < csz) wave_copy_disjoint(dst + e.dst_off + a, frame + e.src_off + a, (csz - a < per) ? csz - a : per);
                got = (int32_t)csz;
            }
        } else {
            typename RSRC::Runs runs;
            if (rsrc.make(b, blk, csz, flags[8], runs))
                got = fz_decode_block_self<C>(sh, runs, frame + e.src_off, csz, frame_cap - e.src_off, dst + e.dst_off, e.dst_size, frame, prof, desc + blk.seq_base, blk.nseq, round_max);
        }
    }
    if (tid == 0) { if (got < 0) atomicOr(flags, 2u); else table[b].dst_size = (uint32_t)got; }
    if (prof && tid == 0) {                                                  // developer aid: how the workgroups of the grid spread in time
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), cc = clock64() - wg_c0;
        atomicMax(prof + 64, cc); atomicAdd(prof + 65, cc); atomicMin(prof + 70, wg_t0); atomicMax(prof + 71, wg_t0); atomicMax(prof + 72, t1); atomicMin(prof + 75, t1);
    }
}

}  // namespace lz4f
