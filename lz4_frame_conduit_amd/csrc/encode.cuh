// encode.cuh -- LZ4 block encode on gfx950.
//
// Replaces the inner block loop of LZ4F_compressUpdate (called at
// /root/reference/src/Codec/Compression/LZ4/Conduit.hsc:311): per frame block, greedy 4-byte
// hash-table match finding and token / literal / offset / match-length emission.
//
// MI355X mapping (not a port of the CPU loop, which probes one position at a time):
//   pass E1  find_matches  one workgroup per run of 64 KiB chunks.  The last 128 KiB of input (the chunk and its 64 KiB window)
//                          and ONE hash table (8192 x {u16 position, u8 tag}) live in LDS; the chunk is cut into 128-byte slices
//                          which the 16 waves take in order from a counter, each parsing its slices greedily like the CPU encoder
//                          (64 positions per probe step, the stride grows after misses), verifying and extending in LDS;
//                          the slices' sequence lists are merged into the chunk's record list {literals, match length, offset}.
//   pass S   layout        sizes of all chunks/blocks (raw fallback when a block would not shrink),
//                          exclusive scan -> final byte offset of every chunk in the frame; writes
//                          size words, frame header, EndMark.
//   pass E2  emit          one wave per chunk: tokens, length bytes, offsets and the literal copies
//                          (16 B per lane) straight to their final position.
// The output is a valid LZ4 block (end-of-block rules of Appendix A.2 are enforced); it is not
// byte-identical to liblz4's -- parity is round-trip identity + ratio tolerance (tests/).
#pragma once
#include <type_traits>
#include "common.cuh"

namespace lz4f {

constexpr uint32_t MFLIMIT = 12, LASTLIT = 5, MINMATCH = 4;
constexpr uint32_t ENC_POOL_SHORT = 0x200u;      // result.flags (LZ4F_MI355X_ENC_POOL_SHORT): tiles went out as literals because the record pool was used up

struct ChunkInfo {           // 32 bytes, one per chunk
    uint32_t nrec;           // sequence records found
    uint32_t first_lit;      // literal run of the first record (before carry-in)
    uint32_t tail_lit;       // literals after the last match (whole chunk when nrec == 0)
    uint32_t body_size;      // encoded bytes of all records as found (no carry-in, no final literals)
    uint32_t carry_in;       // literals inherited from the preceding chunk(s)        [pass S]
    uint32_t flags;          // bit0: block stored raw, bit1: last chunk of its block  [pass S]
    uint64_t out_off;        // absolute offset in the frame of this chunk's first byte [pass S]
};

struct EncGeom {
    uint64_t src_size;           // bytes readable at src (history + blocks)
    uint64_t first_off;          // block 0 starts here; bytes in front of it are history (used when linked)
    uint32_t write_endmark;      // 0: emit blocks only (streaming API), 1: header + blocks + EndMark, 2: + room for the content checksum
    uint32_t block_size;
    uint32_t chunk_size;         // divides block_size
    uint32_t chunks_per_block;
    uint32_t n_blocks;
    uint32_t n_chunks;
    uint32_t linked;             // matches may reach 64 KiB back across block starts
    uint32_t block_checksum;
    uint32_t header_size;
    uint8_t  header[20];
    uint32_t max_rec_per_chunk;  // the most records a chunk can have (one per 4 bytes)
    uint64_t rec_pool;           // records the pool holds (rec_pool_of): the tiles' lists are allocated from it as they are merged
    uint32_t seed_stride;        // pass E1: a run's 64 KiB of history go into the table at every seed_stride-th position
    uint32_t tiles_per_wg;       // pass E1: consecutive chunks a workgroup takes
    uint32_t e1_solo;            // (development) pass E1: bit 0 - only the workgroup's first wave parses (the sequential parse, for comparing ratios); bit 1 - nothing is parsed (the cost of everything else)
};

// The record workspace: [control: bump pointer, tiles that found the pool empty | u32 per chunk: where its list starts | the pool].
// A tile's list is allocated when its merge knows how long it is - the workspace is sized for what inputs have (the engine: a record per
// 5.3 input bytes by default), not for one per 4 bytes everywhere; a tile that finds the pool empty is emitted as literals (valid, bigger)
// and counted.
__device__ __forceinline__ unsigned long long* rec_ctl(const uint64_t* recs) { return (unsigned long long*)recs; }
__device__ __forceinline__ uint32_t* rec_offs(const uint64_t* recs) { return (uint32_t*)(recs + 8); }
__host__ __device__ __forceinline__ uint64_t rec_pool_at(uint32_t n_chunks) { return 8 + (((uint64_t)n_chunks + 1 + 15) / 16) * 8; }      // in records (8 bytes)
__device__ __forceinline__ uint64_t* rec_pool_of(const uint64_t* recs, const EncGeom& g) { return (uint64_t*)recs + rec_pool_at(g.n_chunks); }

__device__ __forceinline__ uint32_t len_ext_bytes(uint32_t v) { return v >= 15 ? (v - 15) / 255 + 1 : 0; }
__device__ __forceinline__ uint32_t seq_size(uint32_t lit, uint32_t mlen)
{
    return 1 + len_ext_bytes(lit) + lit + 2 + len_ext_bytes(mlen - MINMATCH);
}
__device__ __forceinline__ uint64_t pack_rec(uint32_t lit, uint32_t mlen, uint32_t off)
{
    return (uint64_t)lit | ((uint64_t)mlen << 24) | ((uint64_t)off << 48);
}

struct uint4_ua { uint32_t x, y, z, w; } __attribute__((packed, aligned(1)));
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return *(const u32_ua*)p; }
typedef uint64_t u64_ua __attribute__((aligned(1)));
// 8 bytes at p, never reading at or beyond `end`
__device__ __forceinline__ uint64_t ld64_guard(const uint8_t* p, const uint8_t* end)
{
    if (p + 8 <= end) return *(const u64_ua*)p;
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) if (p + i < end) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

// ------------------------------- pass E1, one workgroup per run of tiles ------------------------
// A workgroup searches one 64 KiB tile (= chunk) at a time and goes through a run of consecutive tiles.  What its waves
// share lives in LDS: the last 128 KiB of input (the tile and the 64 KiB window in front of it) in a ring, and ONE hash
// table (8192 x u16, the table shape liblz4 uses for small inputs) that is kept up to date across tiles - so the input is
// read from HBM once (plus 64 KiB of history per run), a candidate is verified and a match extended by LDS reads, and
// nothing is seeded per chunk.
//   * The tile is cut into 256-byte slices which the waves take from a counter in LDS, in order: at any time the waves
//     work on neighbouring slices (a front of E1_WAVES x 256 bytes), so everything further back than the front is in the
//     table when a position is probed.  A slice is parsed greedily like the CPU encoder does (64 positions per step, the
//     stride grows after a step without a hit), from its first byte - or from where a match published by another wave
//     ends, and not at all when such a match covers it.  Matches end at the tile's end, not the slice's.
//   * A table entry is a position's low 16 bits.  Every entry therefore names exactly one position 1..65536 bytes back,
//     and the ring holds that position's real bytes whoever wrote the entry (a wave further ahead included): a candidate
//     is a match iff its bytes say so.
//   * Slices are parsed as if nothing in front reached into them.  At the end of the tile a prefix maximum over the slices'
//     last match ends says how much of each slice the slices in front have covered; sequences inside that are dropped, the
//     one that straddles it is shortened, and the kept sequences of all slices are written out in order as the tile's
//     record list.  What passes S and E2 see is a 64 KiB chunk, as before.
constexpr uint32_t E1_TILE = 65536, E1_RING = 2 * E1_TILE, E1_RMASK = E1_RING - 1;
#ifndef E1_SLICE_BYTES
#define E1_SLICE_BYTES 128
#endif
constexpr uint32_t E1_SLICE = E1_SLICE_BYTES, E1_NSLICE = E1_TILE / E1_SLICE, E1_REC_PER_SLICE = E1_SLICE / MINMATCH;
constexpr uint32_t E1_HASH_LOG = 13;                 // 8192 entries: a position's low 16 bits + 8 more hash bits in a byte array of their own
#ifndef E1_WAVES
#define E1_WAVES 16
#endif
#ifdef E1_DEBUG          // development: every loop bounded, counts of the bounds hit in the scratch buffer's last words
#define E1DBG(...) __VA_ARGS__
#else
#define E1DBG(...)
#endif
#define E1_DBG_AT (((uint64_t)gridDim.x) * (2 * E1_NSLICE * E1_REC_PER_SLICE))
struct alignas(16) E1Shared {
    uint16_t table[1u << E1_HASH_LOG];       // (table and tags first: LDS offsets below 64 KiB fit a DS instruction's offset field)
    uint8_t  tags[1u << E1_HASH_LOG];
    uint32_t s_end[2][E1_NSLICE];            // per slice: end of its last match (0: none); by tile parity
    uint8_t  s_n[2][E1_NSLICE];              // per slice: sequences found
    uint32_t next, cov;                      // next slice to hand out; furthest match end published so far
    uint32_t first;                          // the tile's first slice is done
    uint32_t cur[16];                        // per wave: the slice it has (a lower bound while it is between slices; ~0 when it is through with the tile)
    uint32_t pieces[2];                      // 4 KiB pieces of the NEXT tile already requested (by tile parity)
    uint32_t mode, hits;                     // 0: density of the data not known yet, 1: sparse (long sequences), 2: dense; sequences found while it is not known
    uint32_t tail_d[2];                      // by tile parity: the distance of a match that was cut at the tile's end (0: none) - see E1_OPENER
    uint32_t idle[64];                       // where lanes 1..63 point their share of a wave-wide atomic
    alignas(16) uint8_t ring[E1_RING + 16];  // + the first 16 bytes again, so that a read across the end needs no wrap
};
static_assert(sizeof(E1Shared) <= 163840, "one workgroup per CU");
static_assert(64 * E1_WAVES >= E1_NSLICE, "a thread per slice when the lists are merged");
#ifndef E1_GRAB
#define E1_GRAB 8
#endif
#ifndef E1_GRAB_D
#define E1_GRAB_D 1
#endif
#ifndef E1_DENSE_PASS
#define E1_DENSE_PASS 1
#endif
#ifndef E1_V2
#define E1_V2 1
#endif
#ifndef E1_V3
#define E1_V3 1
#endif
#ifndef E1_V4
#define E1_V4 1
#endif
// Round 4.  Sparse data (long sequences: the mode word says so) is searched from stride 4 on instead of 1: a grab of the bench input is
// 512 random bytes and a 512-byte copy, and strides 1, 2 / 3, 4 / 5, 6 took 2.8 probe iterations to get across the random half where
// 4, 5 / 6, 7 take 1.2 - the match is found a few bytes late and the backward extension recovers its start, as it does for liblz4's
// own skipping.  Measured on the bench input (tools/ab_run.py): stride 1: 3.66-3.69 ms, ratio 1.9613; 2: 3.57; 3: 3.22; 4: 3.14, 1.9507;
// 5: 3.16; 6: 3.06 (with E1_IP2 = 0); 8: 2.90 but ratio 1.878, below liblz4's 1.944.  Dense data (text) never gets here.  What it costs
// elsewhere (tools/ratio_sparse.py, size against liblz4's): rows of 128..2048 bytes +1.0..1.5 %, random-length copies +1 %, structured +1.5..3 %.
#ifndef E1_STEP0
#define E1_STEP0 4
#endif
// liblz4 also indexes ip - 2 behind a match; here that is one more LDS round trip per match for nothing measurable (bench input: ratio
// 1.9615 without against 1.9613 with, text 1.8584 both; 3.56 against 3.66 ms)
#ifndef E1_IP2
#define E1_IP2 0
#endif
// Round 4.  The tile's first slice goes to the first wave alone (see the loop); the other fifteen wait for it only where the tile before
// ended inside a match (E1_OPENER 2) - the case the wait is there for - instead of always (1); 0: never.
#ifndef E1_OPENER
#define E1_OPENER 2
#endif
#ifndef E1_ALIGN
#define E1_ALIGN 1
#endif
// the stride the search goes on with behind a match where sequences are long (liblz4 starts again at 1)
#ifndef E1_STEP_HIT
#define E1_STEP_HIT 1
#endif
constexpr uint32_t E1_GRAB_SPARSE = E1_GRAB, E1_GRAB_DENSE = E1_GRAB_D, E1_PROBE_SLICES = 16, E1_DENSE_HITS = 24;

// 4 / 8 bytes at any byte position of the ring.  (A byte-unaligned ds_read_b32 / _b64 is legal on gfx950 but keeps the LDS busy
// for ~20 cycles per wave instruction - measured: SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS - so the reads are dword-aligned and the
// bytes are picked with v_alignbyte.)
struct __attribute__((aligned(4))) e1_w2 { uint32_t a, b; };
struct __attribute__((aligned(4))) e1_w3 { uint32_t a, b, c; };
__device__ __forceinline__ uint32_t e1_ld32(const uint8_t* ring, uint32_t pos)
{
    const uint32_t i = pos & E1_RMASK;
    const e1_w2 w = *(const e1_w2*)(ring + (i & ~3u));
    return __builtin_amdgcn_alignbyte(w.b, w.a, i & 3u);
}
__device__ __forceinline__ uint64_t e1_ld64(const uint8_t* ring, uint32_t pos)
{
    const uint32_t i = pos & E1_RMASK;
    const e1_w3 w = *(const e1_w3*)(ring + (i & ~3u));
    return (uint64_t)__builtin_amdgcn_alignbyte(w.b, w.a, i & 3u) | ((uint64_t)__builtin_amdgcn_alignbyte(w.c, w.b, i & 3u) << 32);
}
// five dwords from the dword at or below pos: 16 bytes at any byte position, picked with v_alignbyte by the caller
struct __attribute__((aligned(4))) e1_w5 { uint32_t a, b, c, d, e; };
__device__ __forceinline__ e1_w5 e1_ld5(const uint8_t* ring, uint32_t pos)
{
    return *(const e1_w5*)(ring + ((pos & E1_RMASK) & ~3u));
}
// how many of the 16 bytes at two positions (given as five dwords each + the positions' byte phases) agree, from the first on
__device__ __forceinline__ uint32_t e1_same16(const e1_w5& s1, const e1_w5& s2, uint32_t k1, uint32_t k2)
{
    const uint32_t x0 = __builtin_amdgcn_alignbyte(s1.b, s1.a, k1) ^ __builtin_amdgcn_alignbyte(s2.b, s2.a, k2);
    const uint32_t x1 = __builtin_amdgcn_alignbyte(s1.c, s1.b, k1) ^ __builtin_amdgcn_alignbyte(s2.c, s2.b, k2);
    const uint32_t x2 = __builtin_amdgcn_alignbyte(s1.d, s1.c, k1) ^ __builtin_amdgcn_alignbyte(s2.d, s2.c, k2);
    const uint32_t x3 = __builtin_amdgcn_alignbyte(s1.e, s1.d, k1) ^ __builtin_amdgcn_alignbyte(s2.e, s2.d, k2);
    const uint64_t lo = (uint64_t)x0 | ((uint64_t)x1 << 32), hi = (uint64_t)x2 | ((uint64_t)x3 << 32);
    return lo ? (uint32_t)(__builtin_ctzll(lo) >> 3) : hi ? 8u + (uint32_t)(__builtin_ctzll(hi) >> 3) : 16u;
}
__device__ __forceinline__ uint32_t e1_mix(uint32_t v) { return v * 2654435761u; }
__device__ __forceinline__ uint32_t e1_slot(uint32_t hv) { return hv >> (32 - E1_HASH_LOG); }
__device__ __forceinline__ uint32_t e1_tag(uint32_t hv) { return (hv >> (24 - E1_HASH_LOG)) & 0xFFu; }
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t& total)
{
    const uint32_t lane = lane_id();
    uint32_t incl = v;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) { const uint32_t t = __shfl_up(incl, sft); if ((int)lane >= sft) incl += t; }
    total = __builtin_amdgcn_readlane(incl, 63);
    return incl - v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) { const uint32_t t = __shfl_xor(v, s); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s);
    return v;
}
// minimum over lanes 0..15 (one DPP row), the same value to every lane
__device__ __forceinline__ uint32_t wave_min16(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, false); v = t < v ? t : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
}
__device__ __forceinline__ uint32_t wave_excl_scan_max(uint32_t v)
{
    const uint32_t lane = lane_id();
    uint32_t incl = v;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) { const uint32_t t = __shfl_up(incl, sft); if ((int)lane >= sft) incl = t > incl ? t : incl; }
    const uint32_t ex = __shfl_up(incl, 1);
    return lane == 0 ? 0u : ex;
}
constexpr uint32_t E1_PIECE_LOG = 12, E1_NPIECE = E1_TILE >> E1_PIECE_LOG;
// 4 KiB piece pc of the tile at position nts: global -> ring, four LDS-DMA wave instructions.  (From inline asm: hipcc would
// make every later LDS read wait for an LDS-DMA it knows about.  The issuing wave's s_waitcnt vmcnt covers it.)
__device__ __forceinline__ void e1_dma_piece(E1Shared& sh, const uint8_t* __restrict__ src, uint64_t org, uint32_t nts, uint32_t pc)
{
    const uint32_t lane = lane_id();
    const uint32_t ring0 = (uint32_t)(uintptr_t)(lptr_t)sh.ring;
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
        const uint32_t at = nts + (pc << E1_PIECE_LOG) + q * 1024u;
        const uint8_t* gsrc = src + (org + at + lane * 16u);
        const uint32_t ldst = uni(ring0 + (at & E1_RMASK));
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
    }
}

// bytes [a, b) of the input (positions; abs = org + pos) -> ring.  16-byte units where a whole unit lies inside, single bytes at the ends.
__device__ __forceinline__ void e1_fill(E1Shared& sh, const uint8_t* __restrict__ src, uint64_t org, uint32_t a, uint32_t b, uint32_t tid)
{
    constexpr uint32_t NT = 64 * E1_WAVES;
    auto put = [&](uint32_t p) { const uint8_t v = src[org + p]; sh.ring[p & E1_RMASK] = v; if ((p & E1_RMASK) < 16) sh.ring[E1_RING + (p & E1_RMASK)] = v; };
    const uint32_t a16 = (a + 15u) & ~15u, b16 = b & ~15u;
    if (a16 < b16) {
        for (uint32_t u = a16 + tid * 16; u < b16; u += NT * 16) {
            const b16_ua t = *(const b16_ua*)(src + (org + u));
            const uint32_t i = u & E1_RMASK;
            *(uint4*)(sh.ring + i) = uint4{t.a, t.b, t.c, t.d};
            if (i == 0) *(uint4*)(sh.ring + E1_RING) = uint4{t.a, t.b, t.c, t.d};
        }
        if (tid < a16 - a) put(a + tid);
        if (tid < b - b16) put(b16 + tid);
    } else {
        for (uint32_t p = a + tid; p < b; p += NT) put(p);
    }
}

// one record of a slice's list as the walk over it sees it: kept (possibly shortened) or dropped
struct E1Walk {
    uint32_t pos, cov, bend;                 // end of the record before; what the slices in front cover; block end
    __device__ __forceinline__ bool step(uint64_t x, uint32_t& start, uint32_t& mlen)
    {
        const uint32_t lit = (uint32_t)(x & 0xFFFFFFu); mlen = (uint32_t)((x >> 24) & 0xFFFFFFu);
        start = pos + lit;
        const uint32_t end = start + mlen;
        pos = end;
        if (end <= cov) return false;
        if (start < cov) {                                               // straddles: what is left of it, if that is still a match
            const uint32_t ml2 = end - cov;
            if (ml2 < MINMATCH || cov + MFLIMIT > bend) return false;
            start = cov; mlen = ml2;
        }
        return true;
    }
};

__global__ __launch_bounds__(64 * E1_WAVES) void k_find_matches(const uint8_t* __restrict__ src, EncGeom g, ChunkInfo* __restrict__ info,
                                                                uint64_t* __restrict__ recs, uint64_t* __restrict__ scratch)
{
    __shared__ E1Shared sh;
    constexpr uint32_t NT = 64 * E1_WAVES;
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = uni(tid >> 6);
    uint64_t* spec = scratch + (uint64_t)blockIdx.x * (2 * E1_NSLICE * E1_REC_PER_SLICE);      // this workgroup's slice lists (as found), for two tiles
    const uint32_t c0 = blockIdx.x * g.tiles_per_wg;
    const uint32_t c1 = (c0 + g.tiles_per_wg < g.n_chunks) ? c0 + g.tiles_per_wg : g.n_chunks;
    // The chunks tile the stream: chunk c starts at first_off + c * 64 KiB.  Positions: the run's first tile starts at 65536,
    // tile k at 65536 * (k + 1); up to 64 KiB of history sit in front of the first.  abs = org + pos.
    const uint64_t s0 = g.first_off + (uint64_t)c0 * E1_TILE;
    const uint64_t org = s0 - E1_TILE;                                   // (may wrap below zero: only ever used with pos >= the history's start)
    uint32_t hist = 0, len0 = 0;
    if (s0 < g.src_size) {
        const uint64_t bstart0 = g.first_off + (uint64_t)(c0 / g.chunks_per_block) * g.block_size;
        const uint64_t bend0 = (bstart0 + g.block_size < g.src_size) ? bstart0 + g.block_size : g.src_size;
        const uint64_t low_abs = g.linked ? 0 : bstart0;
        hist = (uint32_t)((s0 - low_abs < E1_TILE) ? s0 - low_abs : E1_TILE);
        len0 = (uint32_t)((bend0 - s0 < E1_TILE) ? bend0 - s0 : E1_TILE);
    }
    const uint32_t hist0 = E1_TILE - hist;                               // first valid position
    for (uint32_t i = tid; i < (1u << E1_HASH_LOG) / 2; i += NT) ((uint32_t*)sh.table)[i] = 0;
    for (uint32_t i = tid; i < (1u << E1_HASH_LOG) / 4; i += NT) ((uint32_t*)sh.tags)[i] = 0;
    e1_fill(sh, src, org, hist0, E1_TILE + len0, tid);                   // history + first tile
    __syncthreads();
    // the history's positions into the table, every seed_stride-th, in position order (later ones win)
    // (deterministic parse: by one wave, so that which of two positions with one slot stays is a matter of order, not of timing)
    const uint32_t nt_seed = (g.e1_solo & 1u) ? WAVE : NT;
    if (tid < nt_seed) for (uint32_t p = hist0 + tid * g.seed_stride; p < E1_TILE; p += nt_seed * g.seed_stride)
        if (p + 4 <= E1_TILE + len0) { const uint32_t hv = e1_mix(e1_ld32(sh.ring, p)); sh.table[e1_slot(hv)] = (uint16_t)p; sh.tags[e1_slot(hv)] = (uint8_t)e1_tag(hv); }

    // ---- a tile's slice lists -> its record list.  One wave does it (the last one), on its own, while the others are already
    // ---- searching the next tile: the lists and the per-slice words are double-buffered by tile parity.
    auto merge = [&](uint32_t c, uint32_t par, uint32_t ts, uint32_t te, uint32_t bend, uint32_t nslice) {
        constexpr uint32_t SP = E1_NSLICE / WAVE;                        // consecutive slices per lane
        const uint32_t* s_end = sh.s_end[par];
        const uint8_t* s_n = sh.s_n[par];
        const uint64_t* spc = spec + (size_t)par * (E1_NSLICE * E1_REC_PER_SLICE);
        const uint32_t s_lo = lane * SP;
        // what the slices in front of a slice have covered = the maximum of their last match ends
        uint32_t lmax = 0;
#pragma unroll
        for (uint32_t j = 0; j < SP; j++) { const uint32_t e = (s_lo + j < nslice) ? s_end[s_lo + j] : 0u; lmax = e > lmax ? e : lmax; }
        uint32_t cov0 = wave_excl_scan_max(lmax);
        if (cov0 < ts) cov0 = ts;
        // (all of a lane's lists are requested at once - their counts, and the first two records of each, which is all that
        // most lists have: one memory round trip for the wave instead of one per list)
        uint32_t nn[SP]; ulonglong2 x01[SP];
#pragma unroll
        for (uint32_t j = 0; j < SP; j++) nn[j] = (s_lo + j < nslice) ? s_n[s_lo + j] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < SP; j++) x01[j] = *(const ulonglong2*)(spc + (nn[j] ? s_lo + j : 0u) * E1_REC_PER_SLICE);
        // first walk: how many records are kept, where the lane's first kept one starts, the encoded size of its others
        uint32_t kept = 0, f_start = 0, f_mlen = 0, last_end = 0, body = 0;
        {
            uint32_t cov = cov0;
#pragma unroll
            for (uint32_t j = 0; j < SP; j++) {
                const uint32_t sj = s_lo + j, n = nn[j];
                const uint64_t* lst = spc + sj * E1_REC_PER_SLICE;
                E1Walk w{ts + sj * E1_SLICE, cov, bend};
                // (two records per load, the next pair on its way while this one is looked at: a dense tile has ~10 records per list,
                // and one at a time that was 130 dependent memory round trips per lane and merge - as long as the parse of the tile)
                ulonglong2 cur = x01[j];
                for (uint32_t r = 0; r < n; r += 2) {
                    ulonglong2 nx2 = cur;
                    if (r + 2 < n) nx2 = *(const ulonglong2*)(lst + r + 2);
#pragma unroll
                    for (uint32_t h = 0; h < 2; h++) {
                        if (r + h >= n) break;
                        uint32_t st, ml;
                        if (!w.step(h ? cur.y : cur.x, st, ml)) continue;
                        if (kept == 0) { f_start = st; f_mlen = ml; }
                        else body += seq_size(st - last_end, ml);
                        last_end = st + ml; kept++;
                    }
                    cur = nx2;
                }
                const uint32_t e = (sj < nslice) ? s_end[sj] : 0u; cov = e > cov ? e : cov;
            }
        }
        uint32_t ktot;
        uint32_t prev_end = wave_excl_scan_max(last_end);                // end of the last kept match in front of this lane's slices
        const uint32_t off = wave_excl_scan(kept, ktot);                 // my place in the list
        uint32_t kmax = wave_max(last_end);
        if (prev_end < ts) prev_end = ts;
        if (kmax < ts) kmax = ts;
        const uint32_t f_lit = kept ? f_start - prev_end : 0u;
        if (kept) body += seq_size(f_lit, f_mlen);
        const uint32_t btot = wave_sum(body);
        const uint64_t has = __ballot(kept != 0);
        const uint32_t tile_first = has ? (uint32_t)__builtin_amdgcn_readlane(f_lit, (int)__builtin_ctzll(has)) : 0u;
        // the list's place in the pool
        unsigned long long base = 0;
        if (lane == 0 && ktot) base = atomicAdd(rec_ctl(recs), (unsigned long long)ktot);
        base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)base);
        const bool room = base + ktot <= g.rec_pool;                     // (wave-uniform)
        if (!room && lane == 0) atomicAdd(rec_ctl(recs) + 1, 1ull);
        // second walk: the kept records, literal runs counted from the kept match in front
        if (room) {
            if (lane == 0) rec_offs(recs)[c] = (uint32_t)base;
            uint64_t* out = rec_pool_of(recs, g) + base + off;
            uint32_t cov = cov0, jn = 0, le = prev_end;
#pragma unroll
            for (uint32_t j = 0; j < SP; j++) {
                const uint32_t sj = s_lo + j, n = nn[j];
                const uint64_t* lst = spc + sj * E1_REC_PER_SLICE;
                E1Walk w{ts + sj * E1_SLICE, cov, bend};
                ulonglong2 cur = x01[j];
                for (uint32_t r = 0; r < n; r += 2) {
                    ulonglong2 nx2 = cur;
                    if (r + 2 < n) nx2 = *(const ulonglong2*)(lst + r + 2);
#pragma unroll
                    for (uint32_t h = 0; h < 2; h++) {
                        if (r + h >= n) break;
                        const uint64_t x = h ? cur.y : cur.x;
                        uint32_t st, ml;
                        if (!w.step(x, st, ml)) continue;
                        out[jn++] = pack_rec(st - le, ml, (uint32_t)(x >> 48));
                        le = st + ml;
                    }
                    cur = nx2;
                }
                const uint32_t e = (sj < nslice) ? s_end[sj] : 0u; cov = e > cov ? e : cov;
            }
        }
        ChunkInfo* ci = info + c;
        if (lane == 0) {
            if (room) { ci->nrec = ktot; ci->first_lit = tile_first; ci->tail_lit = te - kmax; ci->body_size = btot; }
            else      { ci->nrec = 0; ci->first_lit = 0; ci->tail_lit = te - ts; ci->body_size = 0; }      // no room for the list: the tile goes out as literals
            sh.mode = (uint64_t)ktot * 64 > (uint64_t)(te - ts) ? 2u : 1u;      // how the next tiles hand out their slices
        }
    };

    if (tid == 0) { sh.mode = 0; sh.cov = E1_TILE; sh.hits = 0; sh.first = 0; sh.next = E1_ALIGN ? 0 : 1; sh.pieces[0] = 0; sh.pieces[1] = 0; sh.tail_d[0] = 0; sh.tail_d[1] = 0; }
    if (tid < 16) sh.cur[tid] = 0;
    uint32_t mode_tile = 0;                                              // (deterministic parse) sh.mode as it stood when this tile began
    uint32_t hold = 1;                                                   // this tile's first slice is waited for: the tile before was left inside a match that goes on
    bool pend = false;                                                   // a tile whose lists are not merged yet
    uint32_t p_c = 0, p_par = 0, p_ts = 0, p_te = 0, p_bend = 0, p_ns = 0;
    __syncthreads();                                                     // the first tile is in the ring, the table is seeded, counters are set
    for (uint32_t c = c0; c < c1; c++) {
        const uint32_t k = c - c0;
        const uint32_t ts = E1_TILE * (k + 1), par = k & 1u;
        const uint32_t blk = c / g.chunks_per_block;
        const uint64_t bstart = g.first_off + (uint64_t)blk * g.block_size;
        const uint64_t bend_abs = (bstart + g.block_size < g.src_size) ? bstart + g.block_size : g.src_size;
        const uint64_t cs_abs = s0 + (uint64_t)k * E1_TILE;
        if (cs_abs >= bend_abs) {                                        // chunk beyond a short last block (uniform: the run ends here)
            ChunkInfo* ci = info + c;
            if (tid == 0) { ci->nrec = 0; ci->first_lit = 0; ci->tail_lit = 0; ci->body_size = 0; }
            continue;
        }
        const uint32_t tlen = (uint32_t)((bend_abs - cs_abs < E1_TILE) ? bend_abs - cs_abs : E1_TILE);
        const uint32_t te = ts + tlen;
        const uint32_t bend = ts + (uint32_t)(bend_abs - cs_abs);        // block end (<= 4 MiB ahead)
        const uint32_t blen = (uint32_t)(bend_abs - bstart);
        const uint32_t nslice = (tlen + E1_SLICE - 1) / E1_SLICE;
        uint32_t low = hist0;                                            // matches may not start before this
        if (!g.linked) { const uint64_t back = cs_abs - bstart; if (back < (uint64_t)(ts - hist0)) low = ts - (uint32_t)back; }
        uint64_t* spec_t = spec + (size_t)par * (E1_NSLICE * E1_REC_PER_SLICE);

        uint32_t nlen = 0;                                               // the next tile of this run
        const uint32_t nts = ts + E1_TILE;
        if (c + 1 < c1) {
            const uint32_t nblk = (c + 1) / g.chunks_per_block;
            const uint64_t nbstart = g.first_off + (uint64_t)nblk * g.block_size;
            const uint64_t nbend = (nbstart + g.block_size < g.src_size) ? nbstart + g.block_size : g.src_size;
            const uint64_t ncs = cs_abs + E1_TILE;
            if (ncs < nbend) nlen = (uint32_t)((nbend - ncs < E1_TILE) ? nbend - ncs : E1_TILE);
        }
        const bool nlen_full = nlen == E1_TILE;

        E1DBG(const unsigned long long q0 = clock64();)
        if (pend && wave == E1_WAVES - 1) {                              // (the others are searching already)
            *(lane == 0 ? &sh.cur[wave] : &sh.idle[lane]) = 0xFFFFFFFFu;     // (holds no slice meanwhile; the one it takes afterwards lies beyond all that are held)
            merge(p_c, p_par, p_ts, p_te, p_bend, p_ns);
        }

        E1DBG(const unsigned long long q1 = clock64();)
        // ---- slices, in order, to whichever wave is free ----
        // a match may start at p iff p + 4 <= te and p + MFLIMIT <= bend; it may end at min(te, bend - LASTLIT).
        // Blocks shorter than MFLIMIT + 1 bytes are literals only (Appendix A.2).
        const uint32_t end_lim = (te < bend - LASTLIT) ? te : bend - LASTLIT;
        E1DBG(uint32_t dq = 0; unsigned long long a_deq = 0, a_probe = 0, a_hit = 0, n_step = 0, n_hit = 0, n_deq = 0;)
        bool fresh = true;
        for (;;) {
            if ((g.e1_solo & 1u) && tid >= WAVE) break;
            E1DBG(const unsigned long long z0 = clock64(); n_deq++;)
            E1DBG(if (++dq > 100000) { if (lane == 0) atomicAdd((unsigned long long*)&scratch[E1_DBG_AT + 0], 1ull); break; })
            // How many 128-byte slices to take: four where sequences are long (fewer restarts of the search, and of the stride), one
            // where they are short - the waves then work within 2 KiB of each other, and what a position's nearest earlier
            // occurrence usually is (text) has been indexed when it is probed.  A tile's first 2 KiB decide for a run's first tile,
            // the later tiles go by the ones before.
            // (deterministic parse: the mode word is written by the merge wave whenever it gets there, so it is looked at once per tile,
            // between the tile's two closing barriers, where the merge of the tile before has long finished)
            uint32_t mode = (g.e1_solo & 1u) ? mode_tile : uni(__hip_atomic_load(&sh.mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            // The tile's first slice goes to the first wave alone, and the others start when it is through (a poll with a budget:
            // going ahead without it costs ratio, nothing else).  Where the tile begins inside a long or periodic match - one hot
            // table slot per distinct four bytes, always holding a position of whoever is furthest ahead - that slice finds it,
            // publishes its end, and nobody searches what it covers.
            // (E1_ALIGN, round 4: only a tile that is waited for has an opener.  Otherwise the first wave takes a helping like everybody - the
            // helpings then start at multiples of their size instead of one slice behind: on the bench input, rows of 512 bytes, a helping
            // began 128 bytes into a random row and ended 128 bytes into the next one, which cost a second probe step per helping, at stride 1)
            const bool opener = fresh && wave == 0 && (!E1_ALIGN || hold);
            const uint32_t grab = opener ? 1u : mode == 1 ? E1_GRAB_SPARSE : mode == 2 ? E1_GRAB_DENSE : 1u;
            uint32_t si = 0;
            if (E1_ALIGN && opener) si = uni(atomicAdd(lane == 0 ? &sh.next : &sh.idle[lane], grab));      // (the others wait for sh.first: this is slice 0)
            if (!opener) {
                if (fresh && E1_OPENER && ((E1_OPENER == 1 && !E1_ALIGN) || hold)) for (uint32_t spin = 0; spin < 4096 && uni(__hip_atomic_load(&sh.first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0; spin++) __builtin_amdgcn_s_sleep(2);
                si = uni(atomicAdd(lane == 0 ? &sh.next : &sh.idle[lane], grab));
            }
            fresh = false;
            if (si >= nslice) { *(lane == 0 ? &sh.cur[wave] : &sh.idle[lane]) = 0xFFFFFFFFu; break; }
            *(lane == 0 ? &sh.cur[wave] : &sh.idle[lane]) = si;
            // The part of the window that every wave has left behind is where the next tile goes: once in a while, ask what the
            // hindmost wave is at and request the 4 KiB pieces that have become free since (the atomic hands each piece to one wave).
            if (nlen_full && (si & 31u) < grab) {
                uint32_t mn = lane < E1_WAVES ? __hip_atomic_load(&sh.cur[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0xFFFFFFFFu;
                mn = wave_min16(mn);
                uint32_t ok_p = mn == 0xFFFFFFFFu ? 0u : (mn * E1_SLICE) >> E1_PIECE_LOG;       // pieces below this lie in front of everybody's window
                if (ok_p > E1_NPIECE) ok_p = E1_NPIECE;
                const uint32_t had = uni(atomicMax(lane == 0 ? &sh.pieces[par] : &sh.idle[lane], ok_p));
                for (uint32_t pc = had; pc < ok_p; pc++) e1_dma_piece(sh, src, org, ts + E1_TILE, pc);
            }
            // (decided when slice 2 x 16 is handed out: the sixteen waves take the first sixteen together, and their counts arrive when
            // they are through - deciding at slice 16 saw no hits yet and called every run's first tile sparse)
            if (mode == 0 && si >= 2 * E1_PROBE_SLICES) {
                mode = uni(__hip_atomic_load(&sh.hits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) > E1_DENSE_HITS ? 2u : 1u;
                *(lane == 0 ? &sh.mode : &sh.idle[lane]) = mode;
                mode_tile = mode;
            }
            const uint32_t cs = ts + si * E1_SLICE;
            const uint32_t ce = (cs + grab * E1_SLICE < te) ? cs + grab * E1_SLICE : te;
            uint64_t* myrec = spec_t + si * E1_REC_PER_SLICE;
            E1DBG(a_deq += clock64() - z0;)
            uint32_t nrec = 0, anchor = cs;
            uint32_t ip = uni(__hip_atomic_load(&sh.cov, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));      // (a match published by another wave may reach into this slice, or over it)
            if (ip < cs) ip = cs;
            if (!(g.e1_solo & 2u) && blen >= MFLIMIT + 1 && ip + MINMATCH <= te && ip + MFLIMIT <= bend && ip < ce) {
                uint32_t last_start = ce - 1;
                if (te - MINMATCH < last_start) last_start = te - MINMATCH;
                if (bend - MFLIMIT < last_start) last_start = bend - MFLIMIT;
                const uint32_t floor_b = ip;                               // backward extension stops here
                uint32_t step = (E1_STEP0 > 1 && mode == 1 && !opener) ? (uint32_t)E1_STEP0 : 1u, two = opener ? 0u : 1u;
                E1DBG(uint32_t it = 0;)
                while (ip <= last_start) {
                    E1DBG(if (++it > 100000) { if (lane == 0) { atomicAdd((unsigned long long*)&scratch[E1_DBG_AT + 1], 1ull); scratch[E1_DBG_AT + 8] = ((uint64_t)ip << 32) | last_start; scratch[E1_DBG_AT + 9] = ((uint64_t)step << 32) | si; } break; })
                    E1DBG(const unsigned long long z1 = clock64(); n_step++;)
                    // Two steps at once - A, and B = the step after A if A finds nothing - so that their LDS round trips overlap.
                    // (B reads the table before A's positions go in: like two positions of one step, they do not see each other.  The tile's
                    // first step goes alone: where the data repeats with a short period, what it indexes is what the second step finds.)
                    const uint32_t ipB = ip + WAVE * step, stepB = step + 1;
                    const uint32_t pA = ip + lane * step, pB = ipB + lane * stepB;
                    const bool actA = pA <= last_start, actB = two && pB <= last_start;
#if E1_V3
                    // (both steps' words asked for before either is looked at, then both steps' table entries: two LDS round trips on the
                    // path of a step that finds nothing instead of three - the compiler, left alone, asks for B's words behind A's wait)
                    const uint32_t iA = pA & E1_RMASK, iB = pB & E1_RMASK;
                    const e1_w2 wA = *(const e1_w2*)(sh.ring + (iA & ~3u)), wB = *(const e1_w2*)(sh.ring + (iB & ~3u));
                    __builtin_amdgcn_sched_barrier(0);
                    const uint32_t vA = __builtin_amdgcn_alignbyte(wA.b, wA.a, iA & 3u), vB = __builtin_amdgcn_alignbyte(wB.b, wB.a, iB & 3u);
                    const uint32_t hvA = e1_mix(vA), hA = e1_slot(hvA), tgA = e1_tag(hvA);
                    const uint32_t hvB = e1_mix(vB), hB = e1_slot(hvB), tgB = e1_tag(hvB);
                    const uint32_t eA = sh.table[hA], etA = sh.tags[hA], eB = sh.table[hB], etB = sh.tags[hB];
                    __builtin_amdgcn_sched_barrier(0);
#else
                    const uint32_t vA = e1_ld32(sh.ring, pA), vB = e1_ld32(sh.ring, pB);      // (idle lanes read on in the ring: harmless)
                    const uint32_t hvA = e1_mix(vA), hA = e1_slot(hvA), tgA = e1_tag(hvA);
                    const uint32_t hvB = e1_mix(vB), hB = e1_slot(hvB), tgB = e1_tag(hvB);
                    const uint32_t eA = sh.table[hA], etA = sh.tags[hA], eB = sh.table[hB], etB = sh.tags[hB];
#endif
                    const uint32_t distA = (pA - eA) & 0xFFFFu, distB = (pB - eB) & 0xFFFFu;
                    // 21 hash bits agree: worth a look at the bytes (in incompressible input one step in five gets that far).
                    // Written so that each question is ONE vector compare whose result IS the ballot: tag difference, distance range and
                    // "is this lane probing at all" folded into x < limit.  (As a chain of && hipcc makes an exec-mask block per term -
                    // save, branch, restore on the scalar unit, which the 16 waves of the workgroup share and this loop runs out of.)
                    const uint32_t roomA = pA - low, roomB = pB - low;                                   // (a distance is < 2^16 anyway: the cap keeps the marker bits above any limit)
#if E1_V3
                    // (the limit is capped at 65535, LZ4's largest offset: a distance of 0 - an empty slot, or the position itself 64 KiB on - reads
                    // as 0xFFFF here and fails the compare by itself, no marker bit of its own)
                    const uint32_t xA = (uint32_t)(uint16_t)((uint16_t)distA - (uint16_t)1u) | ((etA ^ tgA) << 20);
                    const uint32_t xB = (uint32_t)(uint16_t)((uint16_t)distB - (uint16_t)1u) | ((etB ^ tgB) << 20);
#else
                    const uint32_t xA = ((distA - 1u) & 0xFFFFu) | ((etA ^ tgA) << 20) | (distA == 0u ? 1u << 19 : 0u);
                    const uint32_t xB = ((distB - 1u) & 0xFFFFu) | ((etB ^ tgB) << 20) | (distB == 0u ? 1u << 19 : 0u);
#endif
                    // The table is shared with waves further ahead, and a run of one byte (or of a short period) is a single hot slot that
                    // always holds a position of whoever is furthest ahead.  Such runs are found without it: the lane below probes the
                    // position `step` bytes back, and if its four bytes are mine, that is a match.
                    const uint32_t nbA = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vA, 0x111 /* row_shr:1 */, 0xf, 0xf, E1_V3 ? true : false);
                    const uint32_t nbB = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vB, 0x111 /* row_shr:1 */, 0xf, 0xf, E1_V3 ? true : false);
#if E1_V2
                    // (round 3: a run candidate is a candidate like the table's - distance `step`, checked against the bytes like them - instead of
                    // a ballot of its own, and "is this lane probing at all" is a scalar mask on the ballots instead of a select per lane:
                    // ~14 instructions off a step of ~125, none of them a branch)
                    const uint64_t actmA = __ballot(pA <= last_start), actmB = __ballot(pB <= last_start) & (two ? ~0ull : 0ull);
                    constexpr uint32_t LIMCAP = E1_V3 ? 65535u : 65536u;
                    const uint32_t limA = roomA < LIMCAP ? roomA : LIMCAP, limB = roomB < LIMCAP ? roomB : LIMCAP;
                    const bool rleA = nbA == vA, rleB = nbB == vB;              // (a row's first lane sees 0 for its neighbour: a false alarm there does not survive the byte compare)
                    const uint32_t dA = rleA ? step : distA, dB = rleB ? stepB : distB;
                    const uint32_t yA = rleA ? step - 1u : xA, yB = rleB ? stepB - 1u : xB;
                    uint64_t m = __ballot(yA < limA) & actmA;
                    if (m) m &= __ballot(e1_ld32(sh.ring, pA - dA) == vA);                       // (every lane looks: harmless, the ring is a power of two)
                    bool inB = false;
                    if (!m) {
                        // (lanes that have nothing to insert store to a word of their own: a select on the address instead of an exec-mask block)
                        *(actA ? &sh.table[hA] : (uint16_t*)&sh.idle[lane]) = (uint16_t)pA; *(actA ? &sh.tags[hA] : (uint8_t*)&sh.idle[lane]) = (uint8_t)tgA;
                        m = __ballot(yB < limB) & actmB;
                        if (m) m &= __ballot(e1_ld32(sh.ring, pB - dB) == vB);
                        inB = true;
                    }
#else
                    const uint32_t limA = actA ? (roomA < 65536u ? roomA : 65536u) : 0u, limB = actB ? (roomB < 65536u ? roomB : 65536u) : 0u;
                    const uint32_t row0 = (lane & 15u) == 0u ? 1u : 0u;
                    uint64_t m = __ballot(xA < limA);
                    if (m) m &= __ballot(e1_ld32(sh.ring, pA - distA) == vA);                    // (every lane looks: harmless, the ring is a power of two)
                    uint64_t mr = __ballot(((nbA ^ vA) | row0 | (actA ? 0u : 1u)) == 0u) & ~m;
                    m |= mr;
                    bool inB = false;
                    if (!m) {
                        // (lanes that have nothing to insert store to a word of their own: a select on the address instead of an exec-mask block)
                        *(actA ? &sh.table[hA] : (uint16_t*)&sh.idle[lane]) = (uint16_t)pA; *(actA ? &sh.tags[hA] : (uint8_t*)&sh.idle[lane]) = (uint8_t)tgA;
                        m = __ballot(xB < limB);
                        if (m) m &= __ballot(e1_ld32(sh.ring, pB - distB) == vB);
                        mr = __ballot(((nbB ^ vB) | row0 | (actB ? 0u : 1u)) == 0u) & ~m;
                        m |= mr;
                        inB = true;
                    }
                    const uint32_t dA = ((mr >> lane) & 1ull) ? step : distA, dB = ((mr >> lane) & 1ull) ? stepB : distB;
#endif
                    // the greedy parse indexes the positions up to the match it takes, not the ones it jumps over: those come up
                    // again in the next step and would find themselves in the table instead of their candidates
                    const uint32_t L = m ? (uint32_t)__builtin_ctzll(m) : WAVE;
                    {
                        const bool ins = (inB ? actB : actA) && lane <= L;
                        const uint32_t hI = inB ? hB : hA, pI = inB ? pB : pA, tI = inB ? tgB : tgA;
                        *(ins ? &sh.table[hI] : (uint16_t*)&sh.idle[lane]) = (uint16_t)pI; *(ins ? &sh.tags[hI] : (uint8_t*)&sh.idle[lane]) = (uint8_t)tI;
                    }
                    E1DBG(const unsigned long long z2 = clock64(); a_probe += z2 - z1;)
                    if (!m) { ip = two ? ipB + WAVE * stepB : ipB; step += 1 + two; two = 1; continue; }
                    E1DBG(n_hit++;)
                    // ---- dense data (text: a match every ~13 bytes): all the matches of this step at once ----
                    // One match per step means ~6 steps per 64 positions, each a dozen LDS round trips and ~150 scalar instructions.
                    // Here every lane with a verified candidate measures its own match (up to 20 bytes forwards, 4 backwards), the scalar
                    // unit only hops from a match's end to the next lane with one - the greedy parse, as liblz4 would walk it - and the
                    // chosen lanes write their records themselves.  A match of 20 bytes or more ends the pass and is taken by the code
                    // below, which extends it as far as it goes.
                    if (E1_DENSE_PASS && !inB && mode == 2 && step == 1) {
                        const uint32_t di = dA;                                               // (step == 1 here: a run candidate's distance is 1)
                        const uint64_t x0 = e1_ld64(sh.ring, pA + 4) ^ e1_ld64(sh.ring, pA + 4 - di);
                        const uint64_t x1 = e1_ld64(sh.ring, pA + 12) ^ e1_ld64(sh.ring, pA + 12 - di);
                        const uint32_t xb = e1_ld32(sh.ring, pA - 4) ^ e1_ld32(sh.ring, pA - 4 - di);
                        const uint32_t g0 = x0 ? (uint32_t)(__builtin_ctzll(x0) >> 3) : 8u, g1 = x1 ? (uint32_t)(__builtin_ctzll(x1) >> 3) : 8u;
                        const uint32_t fraw = 4u + g0 + (g0 == 8u ? g1 : 0u), flim = end_lim - pA;         // (a probing lane has pA + 4 <= end_lim)
                        const uint32_t fl = fraw < flim ? fraw : flim;
                        const uint32_t lng = (fraw >= 20u && flim > 20u) ? 1u : 0u;
                        const uint32_t nbk = xb ? (uint32_t)(__builtin_clz(xb) >> 3) : 4u;                   // bytes in front that agree
                        const uint32_t flx = fl | (lng << 31);                                 // (one v_readlane per hop)
                        if (!((uint32_t)__builtin_amdgcn_readlane((int)flx, (int)L) >> 31)) {
                            uint64_t taken = 0, rem = m;
                            uint32_t endrel = 0, stop_at = WAVE;
                            for (;;) {
                                const uint32_t sx = (uint32_t)__builtin_ctzll(rem);
                                const uint32_t fx = (uint32_t)__builtin_amdgcn_readlane((int)flx, (int)sx);
                                if (fx >> 31) { stop_at = sx; break; }
                                taken |= 1ull << sx;
                                endrel = sx + fx;
                                if (endrel >= WAVE) break;
                                rem &= ~((1ull << endrel) - 1ull);
                                if (!rem) break;
                            }
                            const bool is_t = (taken >> lane) & 1ull;
                            const uint32_t pe = dpp_excl_scan_max(is_t ? lane + fl : 0u);         // where the chosen match in front of me ends (0: none in this step)
                            const uint32_t before = pe ? ip + pe : anchor;                       // ... as a position: my literals start there
                            const uint32_t bfloor = pe ? ip + pe : (anchor > floor_b ? anchor : floor_b);
                            uint32_t nb = nbk;
                            if (pA - bfloor < nb) nb = pA - bfloor;
                            if (pA - di - low < nb) nb = pA - di - low;
                            const uint32_t rank = (uint32_t)__builtin_popcountll(taken & ((1ull << lane) - 1ull));
                            if (is_t) myrec[nrec + rank] = pack_rec(pA - nb - before, fl + nb, di);
                            nrec += (uint32_t)__builtin_popcountll(taken);
                            E1DBG(if (blockIdx.x == 0 && lane == 0) { atomicAdd((unsigned long long*)&scratch[E1_DBG_AT + 13], 1ull); atomicAdd((unsigned long long*)&scratch[E1_DBG_AT + 14], (unsigned long long)__builtin_popcountll(taken)); atomicAdd((unsigned long long*)&scratch[E1_DBG_AT + 15], (unsigned long long)(stop_at < WAVE ? stop_at : (endrel > WAVE ? endrel : WAVE))); })
                            anchor = ip + endrel;
                            const uint32_t nrel = stop_at < WAVE ? stop_at : (endrel > WAVE ? endrel : WAVE);       // where the search goes on
                            // the positions between the matches go into the table (the ones up to the first match are in already)
                            const bool ins2 = actA && lane > L && lane < nrel && lane >= pe;
                            *(ins2 ? &sh.table[hA] : (uint16_t*)&sh.idle[lane]) = (uint16_t)pA; *(ins2 ? &sh.tags[hA] : (uint8_t*)&sh.idle[lane]) = (uint8_t)tgA;
                            ip += nrel;
                            atomicMax(lane == 0 ? &sh.cov : &sh.idle[lane], anchor);
                            step = 1; two = 1;
                            E1DBG(a_hit += clock64() - z2;)
                            continue;
                        }
                    }
                    E1DBG(if (blockIdx.x == 0 && lane == 0) { atomicAdd((unsigned long long*)&scratch[E1_DBG_AT + 4 + (inB ? 0 : mode != 2 ? 1 : step != 1 ? 2 : 3)], 1ull); })
                    const uint32_t dsel = inB ? dB : dA;
                    uint32_t mp = (uint32_t)__builtin_amdgcn_readlane(inB ? pB : pA, L);
                    const uint32_t d = (uint32_t)__builtin_amdgcn_readlane(dsel, L);
                    // backwards: up to the last match end (or where this slice began) and the lowest position a match may read
                    uint32_t room = mp - (anchor > floor_b ? anchor : floor_b); if (mp - d - low < room) room = mp - d - low;
                    uint32_t nb = 0;
                    E1DBG(uint32_t itb = 0;)
#if E1_V4
                    // (round 3: no exec-mask blocks in the two loops - every lane reads, which is harmless in a ring, and the limits are applied to
                    // the counts; the first backward round and the first forward round are asked for together; forwards 16 bytes per lane, so
                    // that a 512-byte match ends in the round it starts in)
                    uint32_t fw = 0;
                    {
                        const uint32_t kb = lane + 1;
                        const uint32_t b1 = sh.ring[(mp - kb) & E1_RMASK], b2 = sh.ring[(mp - d - kb) & E1_RMASK];
                        const uint32_t a0 = mp + lane * 16;
                        const e1_w5 s1 = e1_ld5(sh.ring, a0), s2 = e1_ld5(sh.ring, a0 - d);
                        __builtin_amdgcn_sched_barrier(0);
                        const uint64_t ne = __ballot(b1 != b2 || kb > room);
                        nb = ne ? (uint32_t)__builtin_ctzll(ne) : WAVE;
                        uint32_t g0 = e1_same16(s1, s2, a0 & 3u, (a0 - d) & 3u);
                        const int32_t r = (int32_t)(end_lim - a0); g0 = r <= 0 ? 0u : (g0 > (uint32_t)r ? (uint32_t)r : g0);
                        const uint64_t stop = __ballot(g0 < 16);
                        if (stop) { const uint32_t f = (uint32_t)__builtin_ctzll(stop); fw = f * 16 + (uint32_t)__builtin_amdgcn_readlane(g0, f); }
                        else fw = WAVE * 16 | 0x80000000u;                       // (goes on)
                    }
                    for (bool more = nb == WAVE; more;) {                        // (rare: more than 64 bytes backwards)
                        E1DBG(if (++itb > 100000) { if (lane == 0) { atomicAdd((unsigned long long*)&scratch[E1_DBG_AT + 2], 1ull); scratch[E1_DBG_AT + 10] = ((uint64_t)room << 32) | nb; } break; })
                        const uint32_t kb = nb + lane + 1;
                        const uint32_t b1 = sh.ring[(mp - kb) & E1_RMASK], b2 = sh.ring[(mp - d - kb) & E1_RMASK];
                        const uint64_t ne = __ballot(b1 != b2 || kb > room);
                        const uint32_t n1 = ne ? (uint32_t)__builtin_ctzll(ne) : WAVE;
                        nb += n1;
                        more = n1 == WAVE;
                    }
                    E1DBG(uint32_t itf = 0;)
                    while (fw >> 31) {
                        fw &= 0x7FFFFFFFu;
                        E1DBG(if (++itf > 100000) { if (lane == 0) { atomicAdd((unsigned long long*)&scratch[E1_DBG_AT + 3], 1ull); scratch[E1_DBG_AT + 11] = ((uint64_t)mp << 32) | fw; scratch[E1_DBG_AT + 12] = ((uint64_t)end_lim << 32) | d; } break; })
                        const uint32_t a0 = mp + fw + lane * 16;
                        const e1_w5 s1 = e1_ld5(sh.ring, a0), s2 = e1_ld5(sh.ring, a0 - d);
                        uint32_t g0 = e1_same16(s1, s2, a0 & 3u, (a0 - d) & 3u);
                        const int32_t r = (int32_t)(end_lim - a0); g0 = r <= 0 ? 0u : (g0 > (uint32_t)r ? (uint32_t)r : g0);
                        const uint64_t stop = __ballot(g0 < 16);
                        if (stop) { const uint32_t f = (uint32_t)__builtin_ctzll(stop); fw += f * 16 + (uint32_t)__builtin_amdgcn_readlane(g0, f); }
                        else fw = (fw + WAVE * 16) | 0x80000000u;
                    }
#else
                    for (;;) {
                        E1DBG(if (++itb > 100000) { if (lane == 0) { atomicAdd((unsigned long long*)&scratch[E1_DBG_AT + 2], 1ull); scratch[E1_DBG_AT + 10] = ((uint64_t)room << 32) | nb; } break; })
                        const uint32_t kb = nb + lane + 1;
                        const bool eq = kb <= room && sh.ring[(mp - kb) & E1_RMASK] == sh.ring[(mp - d - kb) & E1_RMASK];
                        const uint64_t ne = __ballot(!eq);
                        const uint32_t n1 = ne ? (uint32_t)__builtin_ctzll(ne) : WAVE;
                        nb += n1;
                        if (n1 < WAVE) break;
                    }
                    // forwards: 512 bytes per round
                    uint32_t fw = 0;
                    E1DBG(uint32_t itf = 0;)
                    for (;;) {
                        E1DBG(if (++itf > 100000) { if (lane == 0) { atomicAdd((unsigned long long*)&scratch[E1_DBG_AT + 3], 1ull); scratch[E1_DBG_AT + 11] = ((uint64_t)mp << 32) | fw; scratch[E1_DBG_AT + 12] = ((uint64_t)end_lim << 32) | d; } break; })
                        const uint32_t a0 = mp + fw + lane * 8;
                        uint32_t g0 = 0;
                        if (a0 < end_lim) {
                            const uint64_t x = e1_ld64(sh.ring, a0) ^ e1_ld64(sh.ring, a0 - d);
                            g0 = x ? (uint32_t)(__builtin_ctzll(x) >> 3) : 8u;
                            const uint32_t r = end_lim - a0; if (g0 > r) g0 = r;
                        }
                        const uint64_t stop = __ballot(g0 < 8);
                        if (stop) { const uint32_t f = (uint32_t)__builtin_ctzll(stop); fw += f * 8 + (uint32_t)__builtin_amdgcn_readlane(g0, f); break; }
                        fw += WAVE * 8;
                    }
#endif
                    mp -= nb;
                    const uint32_t mlen = nb + fw;
                    myrec[nrec] = pack_rec(mp - anchor, mlen, d);              // (every lane stores the same 8 bytes)
                    nrec++;
                    anchor = ip = mp + mlen;
                    step = (E1_STEP_HIT > 1 && mode == 1) ? (uint32_t)E1_STEP_HIT : 1u; two = 1;
                    atomicMax(lane == 0 ? &sh.cov : &sh.idle[lane], ip);
                    if (E1_OPENER == 2) *((lane == 0 && ip == end_lim) ? &sh.tail_d[par] : &sh.idle[lane]) = d;      // (cut at the tile's end: the next tile's first slice may find the rest)
                    // like the CPU encoder, also index ip - 2
                    E1DBG(a_hit += clock64() - z2;)
                    if (E1_IP2 && ip >= low + 2 && ip + 2 <= te) { const uint32_t q2 = ip - 2, hv2 = e1_mix(e1_ld32(sh.ring, q2)); *(lane == 0 ? &sh.table[e1_slot(hv2)] : (uint16_t*)&sh.idle[lane]) = (uint16_t)q2; *(lane == 0 ? &sh.tags[e1_slot(hv2)] : (uint8_t*)&sh.idle[lane]) = (uint8_t)e1_tag(hv2); }
                }
            }
            // (lane k for slice si + k; the other lanes store to words of their own: no branch, no 64 stores to one address)
            const bool mine_s = lane < grab && si + lane < nslice;
            *(mine_s ? &sh.s_end[par][si + lane] : &sh.idle[lane]) = (lane == 0 && nrec) ? anchor : 0u;
            *(mine_s ? &sh.s_n[par][si + lane] : (uint8_t*)&sh.idle[lane]) = (uint8_t)(lane == 0 ? nrec : 0u);
            if (mode == 0) atomicAdd(lane == 0 ? &sh.hits : &sh.idle[lane], nrec);
            if (opener) *(lane == 0 ? &sh.first : &sh.idle[lane]) = 1u;
        }
        E1DBG(const unsigned long long q2 = clock64();)
        __builtin_amdgcn_s_waitcnt(0);                                   // my slice lists are written
        __syncthreads();                                                 // every slice is done: nobody reads the ring's older half any more
        mode_tile = uni(__hip_atomic_load(&sh.mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        E1DBG(const unsigned long long q3 = clock64();)
        pend = true; p_c = c; p_par = par; p_ts = ts; p_te = te; p_bend = bend; p_ns = nslice;

        // ---- what is left of the next tile into the ring; counters for it ----
        if (tid == 0) { sh.tail_d[par ^ 1u] = 0; sh.cov = nts; sh.hits = 0; sh.first = 0; sh.next = E1_ALIGN ? 0 : 1; sh.pieces[par ^ 1u] = 0; }      // (E1_ALIGN 0: slice 0 is the first wave's)
        if (tid < 16) sh.cur[tid] = 0;
        if (nlen) {
            if (nlen_full) {
                const uint32_t had = sh.pieces[par];
                for (uint32_t pc = had + wave; pc < E1_NPIECE; pc += E1_WAVES) e1_dma_piece(sh, src, org, nts, pc);
                // the ring's first bytes again behind its end
                if ((nts & E1_RMASK) == 0 && tid >= 64 && tid < 68) ((uint32_t*)(sh.ring + E1_RING))[tid - 64] = *(const u32_ua*)(src + (org + nts + (tid - 64) * 4u));
            } else e1_fill(sh, src, org, nts, nts + nlen, tid);          // a short last block's last tile
        }
        E1DBG(const unsigned long long q4 = clock64();)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // my part of the next tile has landed
        E1DBG(const unsigned long long q5 = clock64();)
        __syncthreads();
        {   // does the match that was cut at this tile's end go on in the next one?  (its bytes are in the ring now)
            const uint32_t td = uni(__hip_atomic_load(&sh.tail_d[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            hold = (g.e1_solo & 1u) || (td != 0 && nlen >= 4 && uni(e1_ld32(sh.ring, nts)) == uni(e1_ld32(sh.ring, nts - td))) ? 1u : 0u;
        }
        E1DBG(if (blockIdx.x == 0 && lane == 0) { unsigned long long* d = (unsigned long long*)&scratch[E1_DBG_AT + 16 + wave * 8]; d[0] += q1 - q0; d[1] += q2 - q1; d[2] += q3 - q2; d[3] += q4 - q3; d[4] += q5 - q4; d[5] += clock64() - q5; d[6] += 1; unsigned long long* f = (unsigned long long*)&scratch[E1_DBG_AT + 160 + wave * 6]; f[0] += a_deq; f[1] += a_probe; f[2] += a_hit; f[3] += n_deq; f[4] += n_step; f[5] += n_hit; })
    }
    if (pend && wave == E1_WAVES - 1) merge(p_c, p_par, p_ts, p_te, p_bend, p_ns);
}

// ------------------------------- pass S --------------------------------------------------------
// Three launches: per block on a wave (chunks in the lanes, carries and offsets by prefix sums), the scan over the blocks on
// one workgroup, the per-chunk fix-up on a thread per chunk.

// (the three steps as device functions: a call of a few blocks - the streaming API's one block per call - runs them in ONE launch,
// k_layout_small below; each of these launches costs ~5 us of an otherwise idle GPU there)
__device__ __forceinline__ void layout_block(const EncGeom& g, ChunkInfo* __restrict__ info, BlockOut* __restrict__ table, uint32_t* __restrict__ blk_bytes, uint32_t b)
{
    const uint32_t lane = lane_id();
    const uint64_t bstart = g.first_off + (uint64_t)b * g.block_size;
    const uint32_t blen = (uint32_t)((bstart + g.block_size < g.src_size) ? g.block_size : g.src_size - bstart);
    ChunkInfo* ci = info + (uint64_t)b * g.chunks_per_block;
    const uint32_t nch = (blen + g.chunk_size - 1) / g.chunk_size;
    uint32_t carry_run = 0, total = 0;                      // literals pending from the groups before; payload bytes so far
    for (uint32_t c0 = 0; c0 < nch; c0 += WAVE) {
        const uint32_t c = c0 + lane;
        const bool act = c < nch;
        const uint4 x = act ? *(const uint4*)&ci[c] : uint4{0u, 0u, 0u, 0u};       // {nrec, first_lit, tail_lit, body_size}
        const bool has = act && x.x != 0;
        const uint64_t m = __ballot(has);
        uint32_t sum_tail, sum_size;
        const uint32_t pt = wave_excl_scan(act ? x.z : 0u, sum_tail);
        const uint64_t below = m & ((1ull << lane) - 1ull);
        const uint32_t p = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;   // the chunk with records before me in this group
        const uint32_t pt_p = __shfl(pt, (int)p);
        const uint32_t carry = below ? pt - pt_p : carry_run + pt;                // its tail and every tail since, or all tails since the group began
        const uint32_t size = has ? x.w + carry + len_ext_bytes(x.y + carry) - len_ext_bytes(x.y) : 0u;
        const uint32_t off = wave_excl_scan(size, sum_size);
        if (act) {
            *(uint2*)&ci[c].carry_in = uint2{carry, (c + 1 == nch) ? 2u : 0u};       // carry_in, flags
            ci[c].out_off = total + off;                    // relative for now
        }
        if (m) { const uint32_t last = 63u - (uint32_t)__builtin_clzll(m); carry_run = sum_tail - (uint32_t)__shfl(pt, (int)last); }
        else carry_run += sum_tail;
        total += sum_size;
    }
    total += 1 + len_ext_bytes(carry_run) + carry_run;      // final literal-only sequence
    const bool raw = total >= blen;                         // LZ4F stores raw when it does not fit blockSize-1
    if (raw) for (uint32_t c = lane; c < nch; c += WAVE) ci[c].flags = ((c + 1 == nch) ? 2u : 0u) | 1u;
    if (lane == 0) {
        table[b].word = raw ? (blen | 0x80000000u) : total;
        table[b].dst_off = bstart - g.first_off;
        table[b].dst_size = blen;
        blk_bytes[b] = 4 + (raw ? blen : total) + 4 * g.block_checksum;
    }
}
template <int WAVES_PER_WG>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_layout_blocks(EncGeom g, ChunkInfo* __restrict__ info, BlockOut* __restrict__ table,
                                                                     uint32_t* __restrict__ blk_bytes)
{
    const uint32_t b = uni(blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6));
    if (b >= g.n_blocks) return;
    layout_block(g, info, table, blk_bytes, b);
}

// the scan over the blocks, header, EndMark, result record (one workgroup of 1024)
__device__ __forceinline__ void layout_scan(const EncGeom& g, BlockOut* __restrict__ table, const uint32_t* __restrict__ blk_bytes,
                                            uint8_t* __restrict__ dst, uint64_t dst_cap, ResultRec* __restrict__ res, const uint64_t* __restrict__ recs,
                                            uint64_t* s_part, uint64_t& s_carry)
{
    const uint32_t t = threadIdx.x;
    if (t == 0) s_carry = g.header_size;
    __syncthreads();
    for (uint32_t base = 0; base < g.n_blocks; base += 1024) {
        const uint32_t b = base + t;
        const uint64_t v = (b < g.n_blocks) ? blk_bytes[b] : 0;
        s_part[t] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            uint64_t add = (t >= off) ? s_part[t - off] : 0;
            __syncthreads();
            s_part[t] += add;
            __syncthreads();
        }
        if (b < g.n_blocks) table[b].src_off = s_carry + s_part[t] - v + 4;       // payload follows the size word
        __syncthreads();
        if (t == 1023) s_carry += s_part[1023];
        __syncthreads();
    }
    const uint64_t frame_size = s_carry + (g.write_endmark == 2 ? 8 : g.write_endmark ? 4 : 0);
    const bool fits = frame_size <= dst_cap;
    if (fits) {
        if (t < g.header_size) dst[t] = g.header[t];
        if (t < 4 && g.write_endmark) dst[s_carry + t] = 0;
    }
    if (t == 0 && res) {
        res->size = fits ? frame_size : 0; res->consumed = g.src_size - g.first_off;
        res->status = fits ? ST_OK : ST_DSTSMALL; res->n_blocks = g.n_blocks; res->first_bad_block = 0xFFFFFFFFu; res->flags = g.header[4] | ((g.n_chunks && rec_ctl(recs)[1]) ? ENC_POOL_SHORT : 0u);
    }
    // (the pool's bump pointer and its count of tiles turned away: read, and left at zero for the next call's pass E1 - the host zeroes them
    // only when the workspace is new)
    if (t == 0 && g.n_chunks) { rec_ctl(recs)[0] = 0ull; rec_ctl(recs)[1] = 0ull; }
}
__global__ __launch_bounds__(1024) void k_layout_scan(EncGeom g, BlockOut* __restrict__ table, const uint32_t* __restrict__ blk_bytes,
                                                      uint8_t* __restrict__ dst, uint64_t dst_cap, ResultRec* __restrict__ res, const uint64_t* __restrict__ recs)
{
    __shared__ uint64_t s_part[1024];
    __shared__ uint64_t s_carry;
    layout_scan(g, table, blk_bytes, dst, dst_cap, res, recs, s_part, s_carry);
}

// a thread per chunk: absolute offsets; the first chunk of a block writes the block's size word
__device__ __forceinline__ void layout_chunk(const EncGeom& g, ChunkInfo* __restrict__ info, const BlockOut* __restrict__ table,
                                             uint8_t* __restrict__ dst, const ResultRec* res, uint32_t chunk)
{
    if (res->status != ST_OK) { info[chunk].flags |= 4u; return; }               // does not fit: pass E2 does nothing
    const uint32_t b = chunk / g.chunks_per_block, c = chunk % g.chunks_per_block;
    const BlockOut e = table[b];
    const uint32_t nch = (e.dst_size + g.chunk_size - 1) / g.chunk_size;
    if (c == 0) { const uint32_t w = e.word; uint8_t* q = dst + e.src_off - 4; q[0] = (uint8_t)w; q[1] = (uint8_t)(w >> 8); q[2] = (uint8_t)(w >> 16); q[3] = (uint8_t)(w >> 24); }
    if (c >= nch) return;
    info[chunk].out_off = (e.word >> 31) ? e.src_off + (uint64_t)c * g.chunk_size : e.src_off + info[chunk].out_off;
}
__global__ __launch_bounds__(256) void k_layout_chunks(EncGeom g, ChunkInfo* __restrict__ info, const BlockOut* __restrict__ table,
                                                       uint8_t* __restrict__ dst, const ResultRec* __restrict__ res)
{
    const uint32_t chunk = blockIdx.x * 256 + threadIdx.x;
    if (chunk >= g.n_chunks) return;
    layout_chunk(g, info, table, dst, res, chunk);
}
// the three in one launch, for calls of a few blocks (<= LAYOUT_SMALL_BLOCKS blocks): one workgroup, a wave per block, then the scan, then a
// thread per chunk - what one step leaves in memory for the next is the workgroup's own, behind a barrier
constexpr uint32_t LAYOUT_SMALL_BLOCKS = 64, LAYOUT_SMALL_CHUNKS = 4096;
__global__ __launch_bounds__(1024) void k_layout_small(EncGeom g, ChunkInfo* info, BlockOut* table, uint32_t* blk_bytes, uint8_t* __restrict__ dst, uint64_t dst_cap,
                                                       ResultRec* res, const uint64_t* __restrict__ recs)
{
    __shared__ uint64_t s_part[1024];
    __shared__ uint64_t s_carry;
    for (uint32_t b = uni(threadIdx.x >> 6); b < g.n_blocks; b += 16) layout_block(g, info, table, blk_bytes, b);
    __threadfence_block(); __syncthreads();
    layout_scan(g, table, blk_bytes, dst, dst_cap, res, recs, s_part, s_carry);
    __threadfence_block(); __syncthreads();
    for (uint32_t chunk = threadIdx.x; chunk < g.n_chunks; chunk += 1024) layout_chunk(g, info, table, dst, res, chunk);
}

// ------------------------------- sequence index (optional) -------------------------------------
// A side table for this library's own decoder (decode_indexed.cuh): an entry point into the block's payload every IX_STRIDE
// sequences -- where the token sits, which output position the sequence starts at.  The frame does not change.
//   k_build_index (after pass S)  per chunk: where its entries and sequences start within the block; per block: totals,
//                                 exclusive scans over the blocks; the header
//   pass E2                       writes the entries while it walks the records (it has both positions at hand)
constexpr uint32_t IX_MAGIC = 0x3258494Cu;                     // "LIX2"
constexpr uint32_t IX_SRC_BIAS = 1u << 22;                     // direct matches: payload position relative to the block's payload, + this (the source may sit in the block before)
constexpr uint32_t IX_STRIDE = 16;                             // sequences per entry (a lane of the decoder walks one entry at memory latency)
struct IxHeader { uint32_t magic, n_blocks, chunks_per_block, total_seqs, total_entries, stride, pad0, pad1; };
struct IxBlock  { uint32_t seq_base, nseq, entry_base, nentries; };    // nseq == 0: stored block / nothing to index
struct IxChunk  { uint32_t ent_off, seq_off; };                // first entry / first sequence of the chunk within its block; ent_off bit 31: the block's final sequence follows this chunk's records
struct IxEntry  { uint32_t in_off, out_pos, seq_off, nseq_blk; };      // payload offset, output position, first sequence (all within the block); sequences | block << 8
__host__ __device__ inline uint32_t ix_max_entries_per_chunk(uint32_t chunk_size) { return (chunk_size / 4 + 1 + IX_STRIDE - 1) / IX_STRIDE + 1; }
// What lz4f_mi355x_dev_index_size recommends: room for one sequence per 64 bytes of input on average.  Denser streams make
// the compressor mark the index unusable (they are decoded by the generic kernels, which is the better choice for them anyway).
__host__ __device__ inline size_t ix_typical_entries(uint64_t src_size, uint32_t n_chunks) { return (size_t)(src_size / (64u * IX_STRIDE)) + n_chunks + 64; }
__host__ __device__ inline size_t ix_entries_at(uint32_t n_blocks, uint32_t chunks_per_block)
{
    return sizeof(IxHeader) + (size_t)n_blocks * sizeof(IxBlock) + (size_t)n_blocks * chunks_per_block * sizeof(IxChunk);
}
__device__ __forceinline__ IxBlock* ix_blocks(void* ix) { return (IxBlock*)((uint8_t*)ix + sizeof(IxHeader)); }
__device__ __forceinline__ const IxBlock* ix_blocks(const void* ix) { return (const IxBlock*)((const uint8_t*)ix + sizeof(IxHeader)); }
__device__ __forceinline__ IxChunk* ix_chunks(void* ix, uint32_t n_blocks) { return (IxChunk*)((uint8_t*)ix + sizeof(IxHeader) + (size_t)n_blocks * sizeof(IxBlock)); }
// (the decoder does not know the compressor's chunking: the entries start behind a table whose size the header gives)
__device__ __forceinline__ IxEntry* ix_entries_w(void* ix, uint32_t n_blocks, uint32_t chunks_per_block) { return (IxEntry*)((uint8_t*)ix + ix_entries_at(n_blocks, chunks_per_block)); }
__device__ __forceinline__ const IxEntry* ix_entries(const void* ix, uint32_t n_blocks)
{
    return (const IxEntry*)((const uint8_t*)ix + ix_entries_at(n_blocks, ((const IxHeader*)ix)->chunks_per_block));
}

// per block on a wave (chunks in the lanes): the part of k_build_index that walks the chunks
template <int WAVES_PER_WG>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_index_blocks(EncGeom g, const ChunkInfo* __restrict__ info, const BlockOut* __restrict__ table,
                                                                    const ResultRec* __restrict__ res, void* __restrict__ ix, uint64_t ix_capacity)
{
    const uint32_t lane = lane_id();
    const uint32_t b = uni(blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6));
    if (b >= g.n_blocks || ix_capacity < ix_entries_at(g.n_blocks, g.chunks_per_block)) return;
    IxBlock* blocks = ix_blocks(ix);
    IxChunk* ck = ix_chunks(ix, g.n_blocks) + (uint64_t)b * g.chunks_per_block;
    const bool usable = res->status == ST_OK;              // (linked frames too: entries are per block, parsing needs no history)
    const BlockOut e = table[b];
    const ChunkInfo* ci = info + (uint64_t)b * g.chunks_per_block;
    const uint32_t nch = (e.dst_size + g.chunk_size - 1) / g.chunk_size;
    uint32_t nseq = 0, nent = 0, last = 0xFFFFFFFFu;
    for (uint32_t c0 = 0; c0 < g.chunks_per_block; c0 += WAVE) {
        const uint32_t c = c0 + lane;
        const uint32_t nr = (c < nch && usable && !(e.word >> 31)) ? ci[c].nrec : 0u;
        uint32_t ts, te;
        const uint32_t ps = wave_excl_scan(nr, ts), pe = wave_excl_scan((nr + IX_STRIDE - 1) / IX_STRIDE, te);
        if (c < g.chunks_per_block) ck[c] = IxChunk{nent + pe, nseq + ps};
        const uint64_t m = __ballot(nr != 0);
        if (m) last = c0 + 63u - (uint32_t)__builtin_clzll(m);
        nseq += ts; nent += te;
    }
    if (lane == 0) {
        if (last != 0xFFFFFFFFu) { ck[last].ent_off |= 0x80000000u; nseq += 1; }     // the block's final literal-only sequence
        blocks[b].nseq = nseq; blocks[b].nentries = nent;
    }
}

__global__ __launch_bounds__(1024) void k_build_index(EncGeom g, const ChunkInfo* __restrict__ info, const BlockOut* __restrict__ table,
                                                      const ResultRec* __restrict__ res, void* __restrict__ ix, uint64_t ix_capacity,
                                                      uint32_t blocks_done = 0)
{
    __shared__ uint32_t s_a[1024], s_b[1024];
    __shared__ uint32_t s_carry_a, s_carry_b;
    const uint32_t t = threadIdx.x;
    IxHeader* hd = (IxHeader*)ix;
    IxBlock* blocks = ix_blocks(ix);
    IxChunk* chunks = ix_chunks(ix, g.n_blocks);
    const size_t fixed = ix_entries_at(g.n_blocks, g.chunks_per_block);
    if (ix_capacity < fixed) { if (t == 0 && ix_capacity >= sizeof(IxHeader)) hd->magic = 0; return; }
    const bool usable = res->status == ST_OK;              // (linked frames too: entries are per block, parsing needs no history)
    // 1) per block: its chunks in order (unless k_index_blocks did that already)
    for (uint32_t b = t; b < g.n_blocks && !blocks_done; b += 1024) {
        const BlockOut e = table[b];
        const ChunkInfo* ci = info + (uint64_t)b * g.chunks_per_block;
        IxChunk* ck = chunks + (uint64_t)b * g.chunks_per_block;
        const uint32_t nch = (e.dst_size + g.chunk_size - 1) / g.chunk_size;
        uint32_t nseq = 0, nent = 0, last = 0xFFFFFFFFu;
        for (uint32_t c0 = 0; c0 < g.chunks_per_block; c0 += 8) {
            uint32_t nr[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) nr[k] = (c0 + k < nch) ? ci[c0 + k].nrec : 0;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t c = c0 + k;
                if (c >= g.chunks_per_block) break;
                ck[c] = IxChunk{nent, nseq};
                if (usable && !(e.word >> 31) && nr[k]) { nseq += nr[k]; nent += (nr[k] + IX_STRIDE - 1) / IX_STRIDE; last = c; }
            }
        }
        if (last != 0xFFFFFFFFu) { ck[last].ent_off |= 0x80000000u; nseq += 1; }     // the block's final literal-only sequence
        blocks[b].nseq = nseq; blocks[b].nentries = nent;
    }
    __syncthreads();
    // 2) exclusive scans of both counts over the blocks, in tiles of 1024
    if (t == 0) { s_carry_a = 0; s_carry_b = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < g.n_blocks; base += 1024) {
        const uint32_t b = base + t;
        const uint32_t va = (b < g.n_blocks) ? blocks[b].nseq : 0, vb = (b < g.n_blocks) ? blocks[b].nentries : 0;
        s_a[t] = va; s_b[t] = vb;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const uint32_t aa = (t >= off) ? s_a[t - off] : 0, ab = (t >= off) ? s_b[t - off] : 0;
            __syncthreads();
            s_a[t] += aa; s_b[t] += ab;
            __syncthreads();
        }
        if (b < g.n_blocks) { blocks[b].seq_base = s_carry_a + s_a[t] - va; blocks[b].entry_base = s_carry_b + s_b[t] - vb; }
        __syncthreads();
        if (t == 1023) { s_carry_a += s_a[1023]; s_carry_b += s_b[1023]; }
        __syncthreads();
    }
    if (t == 0) {
        const bool fits = (uint64_t)s_carry_b * sizeof(IxEntry) <= ix_capacity - fixed;     // else: too many sequences for this index, the decoder does without
        *hd = IxHeader{usable && fits ? IX_MAGIC : 0u, g.n_blocks, g.chunks_per_block, s_carry_a, s_carry_b, IX_STRIDE, g.linked ? 1u : 0u, 0u};
    }
}

// ------------------------------- pass E2 -------------------------------------------------------
__device__ __forceinline__ void emit_len_ext(uint8_t* p, uint32_t v /* value minus 15 */)
{
    const uint32_t n255 = v / 255, lane = lane_id();
    for (uint32_t i = lane; i < n255; i += WAVE) p[i] = 255;
    if (lane == 0) p[n255] = (uint8_t)(v - n255 * 255);
}

// ------------------------------- pass E2, 64 records at a time ----------------------------------
// Walking a chunk's records one by one - token, length bytes, literal copy, offset - is one copy in flight and ~90 scalar
// instructions per record.  Here 64 records sit in the lanes: two prefix sums give every
// record its place in the payload and its literals' place in the input, each lane writes its own token / length bytes /
// offset, and the literal runs of all 64 are copied by the lane-level gather of decode_fused.cuh (16-byte units found by
// binary search over a prefix table in LDS, two rounds in flight), runs under 16 bytes with a lane per byte.
#ifndef E2_V2
#define E2_V2 1               // pass E2, long literal runs: the records' arithmetic in the lanes, 64 records at a time (0: the round 2 loop, all of it scalar)
#endif
template <int WAVES_PER_WG, bool split = false>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_emit_gather(const uint8_t* __restrict__ src, EncGeom g,
                                                                   const ChunkInfo* __restrict__ info, const uint64_t* __restrict__ recs,
                                                                   uint8_t* __restrict__ dst, const BlockOut* __restrict__ table, void* __restrict__ ix)
{   // split (a call of few chunks - the streaming API's one block): the workgroup's waves share ONE chunk instead of taking one each.  Every wave
    // walks all of the chunk's batches of 64 records for their sizes (two prefix sums), and emits its share: every nsub-th record of a batch of
    // long literal runs, every nsub-th batch of short ones.  A 64 KiB block's emit is then a quarter as long as one wave's walk through it.
    __shared__ uint4 s_gt[WAVES_PER_WG][2][64];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t chunk = uni(split ? blockIdx.x : blockIdx.x * WAVES_PER_WG + wave);
    const uint32_t sub = split ? wave : 0u, nsub = split ? (uint32_t)WAVES_PER_WG : 1u;
    if (chunk >= g.n_chunks) return;
    const uint32_t blk = chunk / g.chunks_per_block, cib = chunk % g.chunks_per_block;
    const uint64_t bstart = g.first_off + (uint64_t)blk * g.block_size;
    const uint64_t bend_abs = (bstart + g.block_size < g.src_size) ? bstart + g.block_size : g.src_size;
    const uint64_t cs_abs = bstart + (uint64_t)cib * g.chunk_size;
    if (cs_abs >= bend_abs) return;
    const uint64_t ce_abs = (cs_abs + g.chunk_size < bend_abs) ? cs_abs + g.chunk_size : bend_abs;
    const ChunkInfo ci = info[chunk];
    if (ci.flags & 4u) return;
    if (ci.flags & 1u) {                                   // stored block: this chunk's slice of it
        uint8_t* o = dst + ci.out_off;
        const uint64_t n = ce_abs - cs_abs;
        if (sub) return;
        for (uint64_t off = 0; off < n; off += 1u << 20) {
            const uint32_t m = (uint32_t)((n - off < (1u << 20)) ? n - off : (1u << 20));
            wave_copy_disjoint(o + off, src + cs_abs + off, m);
        }
        return;
    }
    const uint64_t* rec = rec_pool_of(recs, g) + (ci.nrec ? rec_offs(recs)[chunk] : 0u);
    uint64_t lp_off = cs_abs - ci.carry_in;                 // input offset of the pending literal run
    uint64_t o_off = ci.out_off;                            // frame offset of the next token
    IxEntry* ent = nullptr; uint32_t ent_seq0 = 0, ent_last = 0; uint64_t pay0 = 0;
    if (ix && ((const IxHeader*)ix)->magic == IX_MAGIC && ci.nrec) {
        const IxChunk ck = ix_chunks(ix, g.n_blocks)[chunk];
        ent = ix_entries_w(ix, g.n_blocks, g.chunks_per_block) + ix_blocks(ix)[blk].entry_base + (ck.ent_off & 0x7FFFFFFFu);
        ent_seq0 = ck.seq_off; ent_last = ck.ent_off >> 31;
        pay0 = table[blk].src_off;
    }
    uint4* T0 = s_gt[wave][0];
    uint4* T1 = s_gt[wave][1];
    auto scan = [&](uint32_t v, uint32_t& total) -> uint32_t {              // exclusive prefix sum over the wave
        uint32_t incl = v;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) { const uint32_t t = __shfl_up(incl, sft); if ((int)lane >= sft) incl += t; }
        total = __builtin_amdgcn_readlane(incl, 63);
        return incl - v;
    };
    auto find = [&](uint32_t u) -> uint32_t {                               // last run whose first unit is <= u
        uint32_t j = 0;
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1) { const uint32_t c = j + step; if (T0[c].x <= u) j = c; }
        return j;
    };
    auto put_ext = [&](uint8_t* q, uint32_t v /* value minus 15 */) {       // one lane: the 255,255,...,rest bytes of a length
        const uint32_t n255 = v / 255;
        for (uint32_t i = 0; i < n255; i++) q[i] = 255;
        q[n255] = (uint8_t)(v - n255 * 255);
    };
    // Long literal runs (a record per >= 192 input bytes on average): the record-at-a-time walk with its wave-wide copies moves
    // them faster (synth50: 1.09 vs 1.22 ms); the batch is for the short ones (text: 9.6 -> 1.1 ms per GiB).
    if ((uint64_t)ci.nrec * 192 <= ce_abs - cs_abs) {
        uint8_t* o = dst + o_off;
        const uint8_t* lp = src + lp_off;
#if E2_V2
        // The records' arithmetic - token, how many length bytes, their last one - for 64 records at once in the lanes; the trip per record
        // gets four numbers by readlane and stores exactly what the loop below stores.  (Walked with everything wave-uniform, every record is
        // computed on the scalar unit, divisions by 255 included: 220 scalar instructions per record, and a CU has one such unit - 88 %
        // busy on the bench input, this kernel's limit there.)
        for (uint32_t r0 = 0; r0 < ci.nrec; r0 += WAVE) {
            const uint32_t rl = r0 + lane;
            const uint64_t xr = rl < ci.nrec ? rec[rl] : 0ull;
            uint32_t vlit = (uint32_t)(xr & 0xFFFFFFu);
            const uint32_t vmlen = (uint32_t)((xr >> 24) & 0xFFFFFFu), voff = (uint32_t)(xr >> 48);
            if (rl == 0) vlit += ci.carry_in;
            const uint32_t vmcode = rl < ci.nrec ? vmlen - MINMATCH : 0u;
            const uint32_t vhb = 1 + len_ext_bytes(vlit), vtb = 2 + len_ext_bytes(vmcode);
            const uint32_t vtoken = ((vlit < 15 ? vlit : 15) << 4) | (vmcode < 15 ? vmcode : 15);
            const uint32_t vlrest = vlit >= 15 ? (vlit - 15) % 255 : 0u, vmrest = vmcode >= 15 ? (vmcode - 15) % 255 : 0u;
            const bool vbig = vhb > WAVE || vtb > WAVE;                    // (lengths of 16 KiB and more: the loop over their 255s)
            const uint32_t vp1 = (vbig ? 0u : vhb | (vtb << 8)) | (vtoken << 16) | (vlrest << 24), vp2 = voff | (vmrest << 16), vadv = vlit + vmlen;
            const uint32_t nb = ci.nrec - r0 < WAVE ? ci.nrec - r0 : WAVE;
            if constexpr (split) {
                // (the waves share the batch: where a record's literals come from and where its bytes go is two prefix sums over the batch, and
                // every wave emits every nsub-th record)
                const bool vact = rl < ci.nrec;
                uint32_t tot_sz, tot_adv;
                const uint32_t vso = scan(vact ? vhb + vlit + vtb : 0u, tot_sz), vsa = scan(vact ? vadv : 0u, tot_adv);
                for (uint32_t k = sub; k < nb; k += nsub) {
                    const uint32_t r = r0 + k;
                    const uint32_t lit = (uint32_t)__builtin_amdgcn_readlane((int)vlit, (int)k), so = (uint32_t)__builtin_amdgcn_readlane((int)vso, (int)k);
                    const uint32_t sa = (uint32_t)__builtin_amdgcn_readlane((int)vsa, (int)k);
                    const uint32_t p1 = (uint32_t)__builtin_amdgcn_readlane((int)vp1, (int)k), p2 = (uint32_t)__builtin_amdgcn_readlane((int)vp2, (int)k);
                    const uint32_t hb = p1 & 0xFFu, tb = (p1 >> 8) & 0xFFu, token = (p1 >> 16) & 0xFFu, off = p2 & 0xFFFFu;
                    uint8_t* q = o + so;
                    const uint8_t* ls = lp + sa;
                    if (ent && (r % IX_STRIDE) == 0 && lane == 0) {
                        uint32_t ns = ci.nrec - r < IX_STRIDE ? ci.nrec - r : IX_STRIDE;
                        if (ent_last && r + IX_STRIDE >= ci.nrec) ns += 1;
                        ent[r / IX_STRIDE] = IxEntry{(uint32_t)((uint64_t)(q - dst) - pay0), (uint32_t)((uint64_t)(ls - src) - bstart), ent_seq0 + r, ns | (blk << 8)};
                    }
                    if (hb) {
                        const uint32_t hi = lane < hb - 1 ? lane : hb - 1, ti = lane < tb - 1 ? lane : tb - 1;
                        q[hi] = (uint8_t)(hi == 0 ? token : hi < hb - 1 ? 255u : p1 >> 24);
                        q += hb;
                        wave_copy_disjoint(q, ls, lit);
                        q += lit;
                        q[ti] = (uint8_t)(ti == 0 ? off : ti == 1 ? off >> 8 : ti < tb - 1 ? 255u : p2 >> 16);
                    } else {
                        const uint32_t mcode = (uint32_t)__builtin_amdgcn_readlane((int)vmcode, (int)k);
                        if (lane == 0) *q = (uint8_t)token;
                        q += 1;
                        if (lit >= 15) { emit_len_ext(q, lit - 15); q += len_ext_bytes(lit); }
                        wave_copy_disjoint(q, ls, lit);
                        q += lit;
                        if (lane == 0) { q[0] = (uint8_t)off; q[1] = (uint8_t)(off >> 8); }
                        q += 2;
                        if (mcode >= 15) emit_len_ext(q, mcode - 15);
                    }
                }
                o += tot_sz; lp += tot_adv;
                continue;
            }
            for (uint32_t k = 0; k < nb; k++) {
                const uint32_t r = r0 + k;
                if (ent && (r % IX_STRIDE) == 0 && lane == 0) {
                    uint32_t ns = ci.nrec - r < IX_STRIDE ? ci.nrec - r : IX_STRIDE;
                    if (ent_last && r + IX_STRIDE >= ci.nrec) ns += 1;
                    ent[r / IX_STRIDE] = IxEntry{(uint32_t)((uint64_t)(o - dst) - pay0), (uint32_t)((uint64_t)(lp - src) - bstart), ent_seq0 + r, ns | (blk << 8)};
                }
                const uint32_t lit = (uint32_t)__builtin_amdgcn_readlane((int)vlit, (int)k), adv = (uint32_t)__builtin_amdgcn_readlane((int)vadv, (int)k);
                const uint32_t p1 = (uint32_t)__builtin_amdgcn_readlane((int)vp1, (int)k), p2 = (uint32_t)__builtin_amdgcn_readlane((int)vp2, (int)k);
                const uint32_t hb = p1 & 0xFFu, tb = (p1 >> 8) & 0xFFu, token = (p1 >> 16) & 0xFFu, off = p2 & 0xFFFFu;
                if (hb) {
                    const uint32_t hi = lane < hb - 1 ? lane : hb - 1, ti = lane < tb - 1 ? lane : tb - 1;
                    o[hi] = (uint8_t)(hi == 0 ? token : hi < hb - 1 ? 255u : p1 >> 24);
                    o += hb;
                    wave_copy_disjoint(o, lp, lit);
                    o += lit;
                    o[ti] = (uint8_t)(ti == 0 ? off : ti == 1 ? off >> 8 : ti < tb - 1 ? 255u : p2 >> 16);
                    o += tb;
                } else {
                    const uint32_t mcode = adv - lit - MINMATCH;
                    if (lane == 0) *o = (uint8_t)token;
                    o += 1;
                    if (lit >= 15) { emit_len_ext(o, lit - 15); o += len_ext_bytes(lit); }
                    wave_copy_disjoint(o, lp, lit);
                    o += lit;
                    if (lane == 0) { o[0] = (uint8_t)off; o[1] = (uint8_t)(off >> 8); }
                    o += 2;
                    if (mcode >= 15) { emit_len_ext(o, mcode - 15); o += len_ext_bytes(mcode); }
                }
                lp += adv;
            }
        }
#else
        for (uint32_t r = 0; r < ci.nrec; r++) {
            if (ent && (r % IX_STRIDE) == 0 && lane == 0) {
                uint32_t ns = ci.nrec - r < IX_STRIDE ? ci.nrec - r : IX_STRIDE;
                if (ent_last && r + IX_STRIDE >= ci.nrec) ns += 1;
                ent[r / IX_STRIDE] = IxEntry{(uint32_t)((uint64_t)(o - dst) - pay0), (uint32_t)((uint64_t)(lp - src) - bstart), ent_seq0 + r, ns | (blk << 8)};
            }
            const uint64_t x = rec[r];
            uint32_t lit = (uint32_t)(x & 0xFFFFFFu);
            const uint32_t mlen = (uint32_t)((x >> 24) & 0xFFFFFFu), off = (uint32_t)(x >> 48);
            if (r == 0) lit += ci.carry_in;
            const uint32_t mcode = mlen - MINMATCH;
            const uint32_t token = ((lit < 15 ? lit : 15) << 4) | (mcode < 15 ? mcode : 15);
            // The control bytes - token + literal-length bytes in front of the literals, offset + match-length bytes behind them -
            // go out as two stores of ALL lanes: lane i writes byte i, the lanes beyond the last byte write that last byte again
            // (same address, same value).  As `if (lane == 0)` blocks and a loop per length they were four exec-mask blocks and
            // ~40 scalar instructions per record, and this kernel's limit on such data is the CU's scalar unit (73 % busy).
            const uint32_t hb = 1 + len_ext_bytes(lit), tb = 2 + len_ext_bytes(mcode);
            if (hb <= WAVE && tb <= WAVE) {
                const uint32_t hi = lane < hb - 1 ? lane : hb - 1, ti = lane < tb - 1 ? lane : tb - 1;
                const uint32_t lrest = lit >= 15 ? (lit - 15) % 255 : 0u, mrest = mcode >= 15 ? (mcode - 15) % 255 : 0u;
                o[hi] = (uint8_t)(hi == 0 ? token : hi < hb - 1 ? 255u : lrest);
                o += hb;
                wave_copy_disjoint(o, lp, lit);
                o += lit;
                o[ti] = (uint8_t)(ti == 0 ? off : ti == 1 ? off >> 8 : ti < tb - 1 ? 255u : mrest);
                o += tb;
            } else {
                if (lane == 0) *o = (uint8_t)token;
                o += 1;
                if (lit >= 15) { emit_len_ext(o, lit - 15); o += len_ext_bytes(lit); }
                wave_copy_disjoint(o, lp, lit);
                o += lit;
                if (lane == 0) { o[0] = (uint8_t)off; o[1] = (uint8_t)(off >> 8); }
                o += 2;
                if (mcode >= 15) { emit_len_ext(o, mcode - 15); o += len_ext_bytes(mcode); }
            }
            lp += lit + mlen;
        }
#endif
        o_off = (uint64_t)(o - dst); lp_off = (uint64_t)(lp - src);
    } else
    for (uint32_t r0 = 0; r0 < ci.nrec; r0 += WAVE) {
        const uint32_t r = r0 + lane;
        const bool act = r < ci.nrec;
        const uint64_t x = act ? rec[r] : 0ull;
        uint32_t lit = (uint32_t)(x & 0xFFFFFFu);
        const uint32_t mlen = (uint32_t)((x >> 24) & 0xFFFFFFu), off = (uint32_t)(x >> 48);
        if (r == 0) lit += ci.carry_in;
        const uint32_t mcode = act ? mlen - MINMATCH : 0u;
        const uint32_t le = len_ext_bytes(lit), me = len_ext_bytes(mcode);
        uint32_t tot_sz, tot_adv;
        const uint32_t so = scan(act ? 1 + le + lit + 2 + me : 0u, tot_sz);
        const uint32_t sa = scan(act ? lit + mlen : 0u, tot_adv);
        if (split && (r0 / WAVE) % nsub != sub) { lp_off += tot_adv; o_off += tot_sz; continue; }      // (another wave's batch - only its sizes are mine to know)
        uint8_t* ob = dst + o_off;
        const uint8_t* sb = src + lp_off;
        // ---- control bytes: every lane its own record (long length runs are rare: a lane loops over them) ----
        if (act) {
            uint8_t* q = ob + so;
            q[0] = (uint8_t)(((lit < 15 ? lit : 15) << 4) | (mcode < 15 ? mcode : 15));
            if (lit >= 15) put_ext(q + 1, lit - 15);
            uint8_t* qo = q + 1 + le + lit;
            qo[0] = (uint8_t)off; qo[1] = (uint8_t)(off >> 8);
            if (mcode >= 15) put_ext(qo + 2, mcode - 15);
            if (ent && (r % IX_STRIDE) == 0) {
                uint32_t ns = ci.nrec - r < IX_STRIDE ? ci.nrec - r : IX_STRIDE;
                if (ent_last && r + IX_STRIDE >= ci.nrec) ns += 1;                 // the block's final sequence rides on its last entry
                ent[r / IX_STRIDE] = IxEntry{(uint32_t)(o_off + so - pay0), (uint32_t)(lp_off + sa - bstart), ent_seq0 + r, ns | (blk << 8)};
            }
        }
        // ---- literal runs of >= 16 bytes: 16-byte units ----
        uint32_t total;
        {
            const uint32_t uL = (act && lit >= 16) ? (lit + 15) >> 4 : 0u;
            const uint32_t P = scan(uL, total);
            if (total) {
                T0[lane] = uint4{P, uL, lit, 0u};
                T1[lane] = uint4{sa, so + 1 + le, 0u, 0u};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            }
        }
        if (total) {
            struct P16 { uint32_t a, b, c, d; };
            auto ld = [&](P16& pc, uint32_t u, uint32_t& dof, bool& on) {
                const uint32_t j = find(u);
                const uint4 e0 = T0[j], e1 = T1[j];
                uint32_t o16 = (u - e0.x) * 16u;
                o16 = (o16 + 16u > e0.z) ? e0.z - 16u : o16;
                on = u < total;
                dof = e1.y + o16;
                const uint8_t* a = on ? sb + (e1.x + o16) : src;
                const b16_ua t = *(const b16_ua*)a;
                pc.a = t.a; pc.b = t.b; pc.c = t.c; pc.d = t.d;
            };
            auto st = [&](const P16& pc, uint32_t dof, bool on) { if (on) *(b16_ua*)(ob + dof) = b16_ua{pc.a, pc.b, pc.c, pc.d}; };
            P16 a0, b0;
            uint32_t da0, db0;
            bool xa0, xb0;
            uint32_t base = 0;
            ld(a0, base + lane, da0, xa0); base += 64;
            for (;;) {
                const bool more_b = base < total;
                ld(b0, base + lane, db0, xb0); base += 64;
                st(a0, da0, xa0);
                if (!more_b) break;
                const bool more_a = base < total;
                ld(a0, base + lane, da0, xa0); base += 64;
                st(b0, db0, xb0);
                if (!more_a) break;
            }
        }
        // ---- literal runs of 1..15 bytes: a lane per byte ----
        uint32_t total_b;
        {
            const uint32_t bL = (act && lit < 16) ? lit : 0u;
            const uint32_t P = scan(bL, total_b);
            if (total_b) {
                T0[lane] = uint4{P, bL, 0u, 0u};
                T1[lane] = uint4{sa, so + 1 + le, 0u, 0u};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            }
        }
        for (uint32_t base = 0; base < total_b; base += 128) {
            uint8_t v[2]; uint32_t dof[2]; bool on[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const uint32_t u = base + 64 * i + lane;
                const uint32_t j = find(u);
                const uint4 e0 = T0[j], e1 = T1[j];
                const uint32_t k = u - e0.x;
                on[i] = u < total_b;
                dof[i] = e1.y + k;
                const uint8_t* a = on[i] ? sb + (e1.x + k) : src;
                v[i] = *a;
            }
#pragma unroll
            for (int i = 0; i < 2; i++) if (on[i]) ob[dof[i]] = v[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();     // tables are rewritten by the next batch
        lp_off += tot_adv; o_off += tot_sz;
    }
    if ((ci.flags & 2u) && sub == 0) {                     // last chunk: final literal-only sequence
        uint8_t* o = dst + o_off;
        const uint32_t lit = ci.nrec ? ci.tail_lit : ci.tail_lit + ci.carry_in;
        if (lane == 0) *o = (uint8_t)((lit < 15 ? lit : 15) << 4);
        o += 1;
        if (lit >= 15) { emit_len_ext(o, lit - 15); o += len_ext_bytes(lit); }
        wave_copy_disjoint(o, src + lp_off, lit);
    }
}

}  // namespace lz4f
