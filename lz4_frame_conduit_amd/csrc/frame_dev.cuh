// frame_dev.cuh -- device-side frame plumbing around the block kernels:
//   k_xxh32_blocks   XXH32 of each block payload (row a5; block checksums, Frame format "BlockChecksum")
//   k_walk_frame     header validation + the serial walk over the block size words (row a3's
//                    getBlockHeader stage) -> block table
//   k_decode_blocks  one wave per table entry (decode.cuh) / one wave for a whole linked frame
//   k_finish_decode  totals, first error, and the (rare) compaction when a non-final block is short
#pragma once
#include "common.cuh"
#include "decode.cuh"
#include "decode_relay.cuh"
#include "decode_fused.cuh"
#include "decode_linked.cuh"
#include "encode.cuh"
#include "encode_solo.cuh"
#include "decode_indexed.cuh"
#include "decode_spx.cuh"

namespace lz4f {

// ------------------------------- XXH32 ----------------------------------------------------------
// XXH32 has four serial accumulator chains (acc = rotl(acc + w*P2, 13) * P1) per 16-byte stripe and no
// way to combine partial states, so one payload is one dependent chain.  Mapping: one wave per
// block; the 64 lanes fetch 1 KiB per step (16 B per lane, coalesced) and do the w*P2 products in
// parallel; the four chains then run on the scalar unit, fed by v_readlane.
__device__ __forceinline__ uint32_t xxh32_finish(uint32_t h, const uint8_t* p, uint32_t rem)
{
    while (rem >= 4) { h = rotl32(h + ld32(p) * XP3, 17) * XP4; p += 4; rem -= 4; }
    while (rem > 0) { h = rotl32(h + (*p) * XP5, 11) * XP1; p++; rem--; }
    h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
    return h;
}

__device__ __forceinline__ uint32_t wave_xxh32(const uint8_t* __restrict__ p, uint32_t len)
{
    const uint32_t lane = lane_id();
    uint32_t h;
    uint32_t done = 0;
    if (len >= 16) {
        uint32_t v1 = XP1 + XP2, v2 = XP2, v3 = 0, v4 = 0u - XP1;
        const uint32_t nstripes = len >> 4;
        for (uint32_t s0 = 0; s0 < nstripes; s0 += WAVE) {
            const uint32_t n = (nstripes - s0 < WAVE) ? nstripes - s0 : WAVE;     // uniform
            uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
            if (lane < n) {
                const b16_ua w = *(const b16_ua*)(p + (size_t)(s0 + lane) * 16);
                m0 = w.a * XP2; m1 = w.b * XP2; m2 = w.c * XP2; m3 = w.d * XP2;
            }
            if (n == WAVE) {
#pragma unroll
                for (int i = 0; i < WAVE; i++) {
                    v1 = rotl32(v1 + __builtin_amdgcn_readlane(m0, i), 13) * XP1;
                    v2 = rotl32(v2 + __builtin_amdgcn_readlane(m1, i), 13) * XP1;
                    v3 = rotl32(v3 + __builtin_amdgcn_readlane(m2, i), 13) * XP1;
                    v4 = rotl32(v4 + __builtin_amdgcn_readlane(m3, i), 13) * XP1;
                }
            } else {
                for (uint32_t i = 0; i < n; i++) {
                    v1 = rotl32(v1 + __builtin_amdgcn_readlane(m0, i), 13) * XP1;
                    v2 = rotl32(v2 + __builtin_amdgcn_readlane(m1, i), 13) * XP1;
                    v3 = rotl32(v3 + __builtin_amdgcn_readlane(m2, i), 13) * XP1;
                    v4 = rotl32(v4 + __builtin_amdgcn_readlane(m3, i), 13) * XP1;
                }
            }
        }
        done = nstripes << 4;
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = XP5;
    }
    h += len;
    return xxh32_finish(h, p + done, len - done);
}

// off/len given explicitly (lz4f_mi355x_dev_xxh32)
template <int WAVES_PER_WG>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_xxh32_ranges(const uint8_t* __restrict__ base, const uint64_t* __restrict__ off,
                                                                    const uint32_t* __restrict__ len, uint32_t n, uint32_t* __restrict__ out)
{
    const uint32_t b = uni(blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6));
    if (b >= n) return;
    const uint32_t h = wave_xxh32(base + off[b], len[b]);
    if (lane_id() == 0) out[b] = h;
}

// mode 0: write the checksum word after each payload of the block table (compress)
// mode 1: compare with the stored word; on mismatch flag the block (decompress)
template <int WAVES_PER_WG>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_xxh32_blocks(uint8_t* frame, BlockOut* __restrict__ table, const ResultRec* __restrict__ res,
                                                                    uint32_t n_max, uint32_t mode, uint32_t* __restrict__ bad)
{
    const uint32_t b = uni(blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6));
    const uint32_t n = res ? res->n_blocks : n_max;
    if (b >= n || b >= n_max) return;
    if (res && res->status != ST_OK) return;
    const uint64_t off = table[b].src_off;
    const uint32_t len = table[b].word & 0x7FFFFFFFu;
    const uint32_t h = wave_xxh32(frame + off, len);
    if (lane_id() == 0) {
        uint8_t* c = frame + off + len;
        if (mode == 0) { c[0] = (uint8_t)h; c[1] = (uint8_t)(h >> 8); c[2] = (uint8_t)(h >> 16); c[3] = (uint8_t)(h >> 24); }
        else {
            const uint32_t stored = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
            if (stored != h) atomicMin(bad, b);
        }
    }
}


// One payload, one wave, FOUR LANES: XXH32's four accumulators are independent chains until the end, and a wave issues one
// instruction every ~4.7 cycles whatever it is (tools/probe/chain_rates.hip) - so the chains run as lanes 0..3 of the SAME
// VALU instructions (v_add, v_alignbit, v_mul_lo: three per 16-byte stripe for all four) instead of one after the other on the
// scalar unit (twenty).  Operands: the wave fetches 1 KiB per step (16 B per lane, coalesced), multiplies by P2 in all lanes,
// parks the products in LDS stripe by stripe (four steps in flight), and lane c reads word c of every stripe back with immediate offsets
// (ds_read_b32, independent of the chain).  The probe's loop does 15.7 cycles per stripe against 82 for the scalar form; the
// kernel ~23: ~1.4 GB/s per payload.  Few big payloads (4 MiB blocks: 256 per GiB; the content checksum: ONE) are bound by exactly this; many small ones
// fill the machine with a wave each (wave_xxh32).  `len` is 64 bits: the content checksum runs over the whole stream (XXH32
// adds the length modulo 2^32).  `park`: 2 KiB of LDS owned by the calling wave (two steps: the next one's products are
// written while this one's are read).  The result is valid in every lane.
constexpr uint32_t XXH_PARK = 2048;
__device__ __forceinline__ uint32_t lane4_xxh32(const uint8_t* __restrict__ p, uint64_t len, uint32_t* park)
{
    const uint32_t lane = lane_id();
    const uint64_t nstripes = len >> 4;
    uint32_t h = XP5;
    if (len >= 16) {
        const uint32_t c = lane & 3u;
        uint32_t v = c == 0 ? XP1 + XP2 : c == 1 ? XP2 : c == 2 ? 0u : 0u - XP1;
        auto fetch = [&](uint64_t s0) -> b16_ua {                                // (no branch: a stripe beyond the end re-reads the last one and is never used)
            const uint64_t s = s0 + lane < nstripes ? s0 + lane : nstripes - 1;
            return *(const b16_ua*)(p + s * 16);
        };
        auto store = [&](const b16_ua& w, uint32_t half) {
            *(uint4*)((uint8_t*)park + half * 1024u + lane * 16u) = uint4{w.a * XP2, w.b * XP2, w.c * XP2, w.d * XP2};
        };
        // four steps of 1 KiB in flight: HBM latency (~2 us) is several chains of 64 (~0.4 us each).  The four registers are
        // named, not rotated: a rotation is a move, and a move of a register that is still being loaded is a wait for it.
        // (issued in this order - the barriers keep hipcc from shuffling them - so that "the oldest load" is the same one on
        // the way into the loop and around it)
        b16_ua r0 = fetch(0);            asm volatile("" ::: "memory");
        b16_ua r1 = fetch(WAVE);         asm volatile("" ::: "memory");
        b16_ua r2 = fetch(2 * WAVE);     asm volatile("" ::: "memory");
        b16_ua r3 = fetch(3 * WAVE);     asm volatile("" ::: "memory");
        uint32_t half = 0;
        auto step = [&](const b16_ua& r, uint64_t s0) {
            store(r, half);                                                        // (the other half is what the previous step read: in-order LDS)
            const uint32_t n = (nstripes - s0 < WAVE) ? (uint32_t)(nstripes - s0) : WAVE;     // uniform
            const uint32_t* q = (const uint32_t*)((const uint8_t*)park + half * 1024u) + c;
            // (the empty asm keeps hipcc from fusing this stripe's multiply with the next stripe's add into v_mad_u64_u32 - a 64-bit
            // multiply-add that takes four passes: 34 cycles per stripe instead of 16)
            if (n == WAVE) {
#pragma unroll
                for (int i = 0; i < WAVE; i++) { v = rotl32(v + q[i * 4], 13) * XP1; asm("" : "+v"(v)); }
            } else {
                for (uint32_t i = 0; i < n; i++) { v = rotl32(v + q[i * 4], 13) * XP1; asm("" : "+v"(v)); }
            }
            half ^= 1u;
        };
        for (uint64_t s0 = 0; s0 < nstripes; s0 += 4 * WAVE) {
            // (every path issues the same four loads in the same order: the compiler then waits for the oldest one only)
            step(r0, s0); r0 = fetch(s0 + 4 * WAVE);
            if (s0 + WAVE < nstripes) step(r1, s0 + WAVE);
            r1 = fetch(s0 + 5 * WAVE);
            if (s0 + 2 * WAVE < nstripes) step(r2, s0 + 2 * WAVE);
            r2 = fetch(s0 + 6 * WAVE);
            if (s0 + 3 * WAVE < nstripes) step(r3, s0 + 3 * WAVE);
            r3 = fetch(s0 + 7 * WAVE);
        }
        h = rotl32((uint32_t)__builtin_amdgcn_readlane(v, 0), 1) + rotl32((uint32_t)__builtin_amdgcn_readlane(v, 1), 7) +
            rotl32((uint32_t)__builtin_amdgcn_readlane(v, 2), 12) + rotl32((uint32_t)__builtin_amdgcn_readlane(v, 3), 18);
    }
    h += (uint32_t)len;
    const uint64_t done = nstripes << 4;
    return xxh32_finish(h, p + done, (uint32_t)(len - done));
}

// (asked for as dynamic LDS and never touched: at most four of these one-wave workgroups per CU, so each chain has a SIMD's issue
// slots to itself - the dispatcher otherwise stacks several on one SIMD while others idle, and the chain is issue-bound)
constexpr uint32_t XXH_SPREAD_LDS = 36u << 10;
constexpr uint32_t XXH_LANE4_BELOW = 16384;     // fewer blocks than this: the four-lane chain (more: a wave each on the scalar unit fills the machine)
// k_xxh32_blocks with the four-lane chain (lane4_xxh32): the host takes it when there are too few blocks to fill the machine
// with the scalar form
template <int WAVES_PER_WG>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_xxh32_blocks4(uint8_t* frame, BlockOut* __restrict__ table, const ResultRec* __restrict__ res,
                                                                     uint32_t n_max, uint32_t mode, uint32_t* __restrict__ bad)
{
    __shared__ __attribute__((aligned(16))) uint32_t park[WAVES_PER_WG][XXH_PARK / 4];
    const uint32_t wv = uni(threadIdx.x >> 6);
    const uint32_t b = uni(blockIdx.x * WAVES_PER_WG + wv);
    const uint32_t n = res ? res->n_blocks : n_max;
    if (b >= n || b >= n_max) return;
    if (res && res->status != ST_OK) return;
    const uint64_t off = table[b].src_off;
    const uint32_t len = table[b].word & 0x7FFFFFFFu;
    const uint32_t h = lane4_xxh32(frame + off, len, park[wv]);
    if (lane_id() == 0) {
        uint8_t* c = frame + off + len;
        if (mode == 0) { c[0] = (uint8_t)h; c[1] = (uint8_t)(h >> 8); c[2] = (uint8_t)(h >> 16); c[3] = (uint8_t)(h >> 24); }
        else {
            const uint32_t stored = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
            if (stored != h) atomicMin(bad, b);
        }
    }
}

// The content checksum (Frame format "Content checksum"; LZ4F_compressEnd / LZ4F_decompress at the frame's end): XXH32 of the whole
// stream, one chain by construction.  One wave (lane4_xxh32: ~1.4 GB/s, measured rate in DESIGN.md) - still 0.75 s per GiB
// beside a codec that takes milliseconds, so a frame asks for it by its FLG bit and pays for it; the default prefs of the
// reference's conduit (Conduit.hsc:205) leave it off.
//   mode 0 (compress): `data` is the input; the word goes to the frame's last 4 bytes (res->size counts them already).
//   mode 1 (decompress): `data` is the decoded output, res->size its length; the stored word is the 4 bytes in front of
//                        res->consumed; a mismatch sets ERROR_contentChecksum_invalid (18).
__global__ __launch_bounds__(64) void k_xxh32_content(const uint8_t* __restrict__ data, uint64_t data_len, uint8_t* frame, ResultRec* __restrict__ res, uint32_t mode)
{
    __shared__ __attribute__((aligned(16))) uint32_t park[XXH_PARK / 4];
    if (res->status != ST_OK) return;
    if (mode == 1 && !((res->flags >> 2) & 1)) return;                 // (the frame has none)
    const uint64_t len = mode == 0 ? data_len : res->size;
    const uint32_t h = lane4_xxh32(data, len, park);
    if (threadIdx.x == 0) {
        uint8_t* c = frame + (mode == 0 ? res->size : res->consumed) - 4;
        if (mode == 0) { c[0] = (uint8_t)h; c[1] = (uint8_t)(h >> 8); c[2] = (uint8_t)(h >> 16); c[3] = (uint8_t)(h >> 24); }
        else {
            const uint32_t stored = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
            if (stored != h) res->status = 18;                         // contentChecksum_invalid
        }
    }
}

// ------------------------------- frame walk -----------------------------------------------------
__device__ __forceinline__ uint32_t rd32_any(const uint8_t* p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ uint32_t xxh32_small(const uint8_t* p, uint32_t len)   // len < 16: header descriptor
{
    uint32_t h = XP5 + len;
    while (len >= 4) { h = rotl32(h + rd32_any(p) * XP3, 17) * XP4; p += 4; len -= 4; }
    while (len > 0) { h = rotl32(h + (*p) * XP5, 11) * XP1; p++; len--; }
    h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
    return h;
}

// Single thread: the walk is a pointer chase (each size word's position depends on all earlier
// ones).  Same validation order as the oracle's orc_decompress_frame.
__global__ __launch_bounds__(64) void k_walk_frame(const uint8_t* __restrict__ frame, uint64_t frame_cap, uint64_t dst_cap,
                             BlockOut* __restrict__ table, uint32_t table_cap, ResultRec* __restrict__ res, const uint32_t* __restrict__ walked = nullptr)
{
    // One wave, every lane walking the same chain (all stores are of identical values to identical addresses): what the other 63
    // lanes are there for is the read-ahead below.
    if (threadIdx.x >= WAVE || blockIdx.x != 0) return;
    if (walked && *walked) return;                                      // the parallel walk (below) has written table and result
    const uint32_t lane = lane_id();
    ResultRec r; r.size = 0; r.consumed = 0; r.status = ST_OK; r.n_blocks = 0; r.first_bad_block = 0xFFFFFFFFu; r.flags = 0;
    // (the read-ahead's landing place: LDS, by LDS-DMA - a load without a destination register, so nothing the compiler allocates can be hit by
    // one that is still in flight; round 3 named v250..v253 in a clobber list, which reserves nothing between statements)
    __shared__ uint32_t ra_sink[WAVE];
    const uint32_t sink0 = uni((uint32_t)(uintptr_t)(lptr_t)ra_sink);
    auto fail = [&](uint32_t st) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); r.status = st; *res = r; };      // (a read-ahead may be in flight)
    if (frame_cap < 7) return fail(12);                              // frameHeader_incomplete
    const uint32_t magic = rd32_any(frame);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                      // skippable frame: no output
        if (frame_cap < 8) return fail(12);
        const uint64_t sz = rd32_any(frame + 4);
        if (frame_cap < 8 + sz) return fail(12);
        r.consumed = 8 + sz; r.flags = 0x100; *res = r; return;
    }
    if (magic != 0x184D2204u) return fail(13);                       // frameType_unknown
    const uint32_t flg = frame[4];
    if ((flg >> 1) & 1) return fail(8);                              // reservedFlag_set
    if (((flg >> 6) & 3) != 1) return fail(6);                       // headerVersion_wrong
    const uint32_t hsize = 7 + (((flg >> 3) & 1) ? 8 : 0) + ((flg & 1) ? 4 : 0);
    if (frame_cap < hsize) return fail(12);
    const uint32_t bd = frame[5];
    const uint32_t bsid = (bd >> 4) & 7;
    if ((bd >> 7) & 1) return fail(8);
    if (bsid < 4) return fail(ST_MAXBLOCK);
    if (bd & 15) return fail(8);
    if (((xxh32_small(frame + 4, hsize - 5) >> 8) & 0xFF) != frame[hsize - 1]) return fail(17);   // headerChecksum_invalid
    r.flags = flg;
    const uint32_t bs = 1u << (8 + 2 * bsid);
    const uint32_t bck = (flg >> 4) & 1;
    uint64_t content = 0;
    if ((flg >> 3) & 1) content = (uint64_t)rd32_any(frame + 6) | ((uint64_t)rd32_any(frame + 10) << 32);
    uint64_t pos = hsize, out = 0;
    uint32_t n = 0;
    // Every hop is a dependent read that misses every cache: 0.66 us per block (4 GiB in 4 MiB blocks: 0.68 ms).  Where the size word
    // AFTER the next one will be is not known - but blocks of one stream are often of similar size, so the 64 lanes touch the 32 KiB
    // around where it would be if the next block were as big as this one (four lines each; round 2: one line each, 8 KiB, which
    // caught about half of the hops of the bench's frame).  A right guess makes that hop an L2 hit, a wrong one costs nothing: the
    // words read ahead are only looked at to keep the loads alive.
    // (The read-ahead must not be waited for: memory operations return in order, so the next size word is asked for FIRST and the
    // read-ahead behind it, and the wait is for all but the youngest - which the compiler cannot be told, hence the asm.  The
    // read-ahead is a global_load_lds_dword: it lands in 256 bytes of LDS nobody reads, no register is written.  Two or four lines
    // per lane measured slower than one - 0.478 / 0.600 against 0.451 ms for the bench frame - and are gone.)
    if (frame_cap - pos < 4) return fail(12);
    uint32_t w = *(const u32_ua*)(frame + pos);
    for (;;) {
        pos += 4;
        if (w == 0) break;
        const uint32_t csz = w & 0x7FFFFFFFu;
        if (csz > bs) return fail(ST_MAXBLOCK);
        if (frame_cap - pos < (uint64_t)csz + 4 * bck) return fail(12);
        if (n >= table_cap) return fail(ST_DSTSMALL);
        if (out >= dst_cap) return fail(ST_DSTSMALL);
        table[n].src_off = pos; table[n].dst_off = out; table[n].word = w;
        table[n].dst_size = (uint32_t)((dst_cap - out < bs) ? dst_cap - out : bs);      // capacity; decode overwrites
        out += bs;                                                  // provisional: full blocks (fixed up after decode)
        pos += (uint64_t)csz + 4 * bck;
        n++;
        if (frame_cap - pos < 4) return fail(12);
        const uint64_t guess = pos + 4 + csz + 4 * bck + (uint64_t)lane * 128u;          // lane 32 = the guess itself
        const bool inside = guess >= 4096u + 4u && guess - 4096u + 4u <= frame_cap;
        const uint8_t* pw = frame + pos;
        const uint8_t* pa = inside ? frame + ((guess - 4096u) & ~(uint64_t)3) : pw;
        uint32_t keep;
        asm volatile("global_load_dword %0, %2, off\n\ts_mov_b32 %1, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %3, off\n\ts_mov_b32 m0, %1\n\ts_waitcnt vmcnt(1)"
                     : "=&v"(w), "=&s"(keep) : "v"(pw), "v"(pa), "s"(sink0) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((flg >> 2) & 1) { if (frame_cap - pos < 4) return fail(12); pos += 4; }   // content checksum: verified behind the decode (k_xxh32_content)
    r.n_blocks = n; r.consumed = pos; r.size = content;             // size = declared content size until decode fills it
    *res = r;
}

// ------------------------------- frame walk, in parallel ----------------------------------------
// The size words are a linked list: 16 384 dependent HBM reads (0.66 us each) for 1 GiB in 64 KiB blocks.  A block start has
// a look, though: a word <= maxBlockSize (top bits clear) that leads to another such word, or to the EndMark.  So:
//   k_walk_cand     every byte position of the frame is looked at once (a streaming read); a position whose word could be a
//                   size word and whose chain survives WK_HOPS hops is a candidate; a workgroup keeps those of its 64 KiB
//   k_walk_order    scan over the workgroups' counts -> the candidates in position order (one workgroup)
//   k_walk_link     candidate i is a block start iff candidate i+1 is exactly where its size word points (or that is the
//                   EndMark): if the list, from the first byte behind the header on, is such a chain, it IS the walk - the
//                   block table is written from it; anything else (a false candidate in between, too many candidates)
//                   leaves the frame to k_walk_frame, which then runs behind.
// A wrong guess can therefore cost time, never change the table.
constexpr uint32_t WK_CHUNK = 65536, WK_SLOTS = 30, WK_HOPS = 3;
struct WalkState {                     // device scratch, zeroed per call
    uint32_t done;                     // 1: the block table and the result record are final (k_walk_frame returns at once)
    uint32_t overflow;                 // a workgroup had more candidates than slots / more candidates than table entries
    uint32_t total;                    // candidates
    uint32_t first_end;                // lowest candidate whose size word points at the EndMark
    uint32_t first_break;              // lowest candidate whose successor is not where its size word points
    uint32_t hsize, bs, bck, flg;      // from the header
    uint32_t head_ok;
    uint64_t content;
};
struct WalkChunk { uint32_t n; uint32_t off[WK_SLOTS]; uint32_t pad; };      // 128 bytes per 64 KiB of frame

// header checks of k_walk_frame, shared (returns 0 and fills the fields, or the LZ4F error code)
__device__ __forceinline__ uint32_t walk_header(const uint8_t* __restrict__ frame, uint64_t frame_cap, uint32_t& hsize, uint32_t& bs, uint32_t& bck,
                                                uint32_t& flg, uint64_t& content)
{
    if (frame_cap < 7) return 12;
    const uint32_t magic = rd32_any(frame);
    if (magic != 0x184D2204u) return 13;                              // (skippable frames: k_walk_frame)
    flg = frame[4];
    if ((flg >> 1) & 1) return 8;
    if (((flg >> 6) & 3) != 1) return 6;
    hsize = 7 + (((flg >> 3) & 1) ? 8 : 0) + ((flg & 1) ? 4 : 0);
    if (frame_cap < hsize) return 12;
    const uint32_t bd = frame[5];
    const uint32_t bsid = (bd >> 4) & 7;
    if ((bd >> 7) & 1) return 8;
    if (bsid < 4) return ST_MAXBLOCK;
    if (bd & 15) return 8;
    if (((xxh32_small(frame + 4, hsize - 5) >> 8) & 0xFF) != frame[hsize - 1]) return 17;
    bs = 1u << (8 + 2 * bsid);
    bck = (flg >> 4) & 1;
    content = 0;
    if ((flg >> 3) & 1) content = (uint64_t)rd32_any(frame + 6) | ((uint64_t)rd32_any(frame + 10) << 32);
    return 0;
}

__global__ void k_walk_head(const uint8_t* __restrict__ frame, uint64_t frame_cap, WalkState* __restrict__ ws, uint32_t preset_total = 0)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t hsize = 0, bs = 0, bck = 0, flg = 0; uint64_t content = 0;
    const uint32_t st = walk_header(frame, frame_cap, hsize, bs, bck, flg, content);
    ws->done = 0; ws->overflow = 0; ws->total = preset_total; ws->first_end = 0xFFFFFFFFu; ws->first_break = 0xFFFFFFFFu;
    ws->hsize = hsize; ws->bs = bs; ws->bck = bck; ws->flg = flg; ws->content = content;
    ws->head_ok = st == 0 ? 1u : 0u;                                  // (a bad header: k_walk_frame gives the verdict)
}

// does the word at `pos` look like a size word whose block fits the frame?  -> position of the next word
__device__ __forceinline__ bool walk_step(const uint8_t* __restrict__ frame, uint64_t frame_cap, uint32_t bs, uint32_t bck, uint64_t pos, uint64_t& next, bool& end)
{
    if (pos > frame_cap || frame_cap - pos < 4) return false;       // (positions out of a trailer are anybody's numbers)
    const uint32_t w = *(const u32_ua*)(frame + pos);
    end = w == 0;
    if (end) { next = pos + 4; return true; }
    const uint32_t csz = w & 0x7FFFFFFFu;
    if (csz > bs) return false;
    next = pos + 4 + csz + 4 * bck;
    return next + 4 <= frame_cap;                                     // (there is at least an EndMark behind every block)
}

__global__ __launch_bounds__(256) void k_walk_cand(const uint8_t* __restrict__ frame, uint64_t frame_cap, const WalkState* __restrict__ ws,
                                                   WalkChunk* __restrict__ chunks)
{
    __shared__ uint32_t s_n;
    __shared__ uint32_t s_off[WK_SLOTS];
    if (!ws->head_ok) return;
    const uint32_t bs = ws->bs, bck = ws->bck, hsize = ws->hsize;
    const uint64_t c0 = (uint64_t)blockIdx.x * WK_CHUNK;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    // a size word's upper bits: bit 31 may be set (stored block), the count is <= bs
    const uint32_t hi_mask = ~((bs << 1) - 1u) & 0x7FFFFFFFu;       // bits that must be clear (bs itself is allowed: checked by walk_step)
    // 20 bytes: a thread's 16 positions' words (guarded at the frame's end); four pieces are asked for before the first is
    // looked at - one 16-byte load in flight per thread is a sixth of what the memory system needs to stream
    auto load20 = [&](uint64_t p0, uint32_t (&w)[5]) {
        w[0] = w[1] = w[2] = w[3] = w[4] = 0;
        if (p0 + 4 > frame_cap) return;
        if (p0 + 20 <= frame_cap) { const b16_ua t = *(const b16_ua*)(frame + p0); w[0] = t.a; w[1] = t.b; w[2] = t.c; w[3] = t.d; w[4] = *(const u32_ua*)(frame + p0 + 16); }
        else { for (uint32_t k = 0; k < 20 && p0 + k < frame_cap; k++) w[k >> 2] |= (uint32_t)frame[p0 + k] << (8 * (k & 3)); }
    };
    auto scan16 = [&](uint32_t o, const uint32_t (&w)[5]) {
        const uint64_t p0 = c0 + o;
        if (p0 + 4 > frame_cap) return;
        // a size word's top byte is 0x00 or 0x80 for every block size there is: if none of the 16 positions' top bytes (bytes 3..18
        // of the 20) is, there is nothing to look at - which is the case nearly everywhere in text, and what makes this pass a
        // streaming read there.  (SWAR zero-byte test on x & 0x7F7F7F7F; a borrow can only add false alarms.)
        auto zb = [](uint32_t x) { const uint32_t y = x & 0x7F7F7F7Fu; return (y - 0x01010101u) & 0x80808080u; };
        if (((zb(w[0]) & 0x80000000u) | zb(w[1]) | zb(w[2]) | zb(w[3]) | (zb(w[4]) & 0x00808080u)) == 0u) return;
        // which of the 16 words could be size words (a bit each), then only those are followed - in a loop, not 16 inlined copies:
        // this kernel was 40 KB of code, more than the instruction cache, and streamed at a third of what a plain read does
        uint32_t plausible = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t sh = (i & 3) * 8;
            const uint32_t v = sh ? (w[i >> 2] >> sh) | (w[(i >> 2) + 1] << (32 - sh)) : w[i >> 2];
            plausible |= ((v & hi_mask) == 0 && v != 0) ? 1u << i : 0u;   // (the EndMark is an end, not a block)
        }
#pragma unroll 1
        while (plausible) {
            const uint32_t i = (uint32_t)__builtin_ctz(plausible);
            plausible &= plausible - 1;
            const uint64_t p = p0 + i;
            if (p < hsize || p + 4 > frame_cap) continue;
            // a chain of WK_HOPS plausible words (or fewer, up to the EndMark) from here
            uint64_t q = p, nx = 0; bool end = false, good = true;
#pragma unroll 1
            for (uint32_t h = 0; h < WK_HOPS && good && !end; h++) { good = walk_step(frame, frame_cap, bs, bck, q, nx, end); if (h == 0 && end) good = false; q = nx; }
            if (!good) continue;
            const uint32_t at = atomicAdd(&s_n, 1u);
            if (at < WK_SLOTS) s_off[at] = o + i;
        }
    };
    constexpr uint32_t STRIDE = 256 * 16;                              // (WK_CHUNK = 16 strides: four rounds of four)
    for (uint32_t o = threadIdx.x * 16; o < WK_CHUNK; o += 4 * STRIDE) {
        uint32_t wa[5], wb[5], wc[5], wd[5];
        load20(c0 + o, wa); load20(c0 + o + STRIDE, wb); load20(c0 + o + 2 * STRIDE, wc); load20(c0 + o + 3 * STRIDE, wd);
        scan16(o, wa); scan16(o + STRIDE, wb); scan16(o + 2 * STRIDE, wc); scan16(o + 3 * STRIDE, wd);
    }
    __syncthreads();
    if (threadIdx.x == 0) chunks[blockIdx.x].n = s_n;
    if (threadIdx.x < WK_SLOTS && threadIdx.x < s_n) chunks[blockIdx.x].off[threadIdx.x] = s_off[threadIdx.x];
}

// one workgroup: exclusive scan over the chunks' counts, then every chunk's candidates, sorted, into the list
__global__ __launch_bounds__(1024) void k_walk_order(WalkChunk* __restrict__ chunks, uint32_t n_chunks, WalkState* __restrict__ ws, uint64_t* __restrict__ list, uint32_t list_cap)
{
    __shared__ uint32_t s_part[1024];
    __shared__ uint32_t s_carry;
    if (!ws->head_ok) return;
    const uint32_t t = threadIdx.x;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_chunks; base += 1024) {
        const uint32_t c = base + t;
        uint32_t n = c < n_chunks ? chunks[c].n : 0u;
        if (n > WK_SLOTS) { ws->overflow = 1; n = 0; }
        s_part[t] = n;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const uint32_t a = t >= off ? s_part[t - off] : 0u;
            __syncthreads();
            s_part[t] += a;
            __syncthreads();
        }
        const uint32_t at = s_carry + s_part[t] - n;
        if (n) {
            if (at + n > list_cap) ws->overflow = 1;
            else {
                uint32_t v[WK_SLOTS];
                for (uint32_t i = 0; i < n; i++) v[i] = chunks[c].off[i];
                for (uint32_t i = 1; i < n; i++) { const uint32_t x = v[i]; uint32_t j = i; while (j && v[j - 1] > x) { v[j] = v[j - 1]; j--; } v[j] = x; }
                for (uint32_t i = 0; i < n; i++) list[at + i] = (uint64_t)c * WK_CHUNK + v[i];
            }
        }
        __syncthreads();
        if (t == 1023) s_carry += s_part[1023];
        __syncthreads();
    }
    if (t == 0) ws->total = s_carry;
}

// ---- big blocks: the walk from many seeds at once ----
// A frame of 4 MiB blocks is a chain of ~1000 dependent reads that miss every cache (0.45 ms for the bench's frame), and looking at
// every byte position (k_walk_cand) would read the whole frame.  But a window of maxBlockSize + 8 bytes anywhere in the frame holds
// a size word, and a position that passes the plausibility test WK_SEED_HOPS times in a row is one (2^-9 per hop for random bytes).
//   k_walk_seeds    workgroup s scans the frame from s / n of its length on, 512 KiB at a time, until it has the first such position
//   k_walk_chains   a lane per seed follows the size words from its seed to the next lane's seed: all lanes hop at once, so the
//                   chain of 1000 reads becomes chains of 1000 / n; the positions visited, in order, are the candidate list
// and k_walk_link / k_walk_verdict decide as for small blocks: the list is THE walk iff every entry is exactly where the one in
// front points, from the first byte behind the header to the EndMark.  A seed that is no block start breaks the chain where the
// true one arrives next to it, and the frame is left to k_walk_frame behind: a wrong guess costs time, never changes the table.
constexpr uint32_t WK_SEEDS = 64, WK_SEED_HOPS = 4, WK_SEED_PIECE = 131072, WK_SEED_LIST = 2048, WK_LANE_CAP = 192;
constexpr uint64_t WK_NOSEED = ~0ull;
__host__ __device__ inline uint32_t wk_seed_pieces(uint32_t bs) { return (bs + 64u + WK_SEED_PIECE - 1) / WK_SEED_PIECE; }
// grid: WK_SEEDS x wk_seed_pieces(bs) workgroups - every 128 KiB piece of every window by a workgroup of its own; seeds[s] (preset to
// WK_NOSEED) gets the lowest find.  Two phases per workgroup: the piece is streamed and the plausible positions (one in 512 of random
// bytes) are put on a list in LDS; then a thread per listed position follows its chain.  (Following a chain where it is found puts
// two to four dependent reads into every round of the stream: 0.27-0.43 ms for what is 0.05 ms of reading.)
__global__ __launch_bounds__(256) void k_walk_seeds(const uint8_t* __restrict__ frame, uint64_t frame_cap, const WalkState* __restrict__ ws, unsigned long long* __restrict__ seeds)
{
    __shared__ uint32_t s_best, s_n;
    __shared__ uint32_t s_list[WK_SEED_LIST];
    if (!ws->head_ok) return;
    const uint32_t bs = ws->bs, bck = ws->bck, hsize = ws->hsize, t = threadIdx.x;
    const uint32_t npc = wk_seed_pieces(bs), s = blockIdx.x / npc, pc = blockIdx.x % npc;
    if (s == 0) { if (t == 0 && pc == 0) seeds[0] = hsize; return; }     // (the first block's size word follows the header)
    const uint64_t span = frame_cap - hsize;
    const uint64_t a = hsize + (span / WK_SEEDS) * s;
    uint64_t wend = a + bs + 64u; if (wend > frame_cap) wend = frame_cap;
    const uint64_t pp = a + (uint64_t)pc * WK_SEED_PIECE;
    if (pp >= wend) return;
    const uint32_t hi_mask = ~((bs << 1) - 1u) & 0x7FFFFFFFu;
    if (t == 0) { s_best = 0xFFFFFFFFu; s_n = 0; }
    __syncthreads();
    for (uint64_t p0 = pp; p0 < pp + WK_SEED_PIECE && p0 < wend; p0 += 16384u) {
        // a thread looks at 64 positions of the round: four loads of 20 bytes asked for together
        uint32_t w[4][5];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint64_t q = p0 + (uint64_t)k * 4096u + t * 16u;
            w[k][0] = w[k][1] = w[k][2] = w[k][3] = w[k][4] = 0;
            if (q + 20 <= wend) { const b16_ua x = *(const b16_ua*)(frame + q); w[k][0] = x.a; w[k][1] = x.b; w[k][2] = x.c; w[k][3] = x.d; w[k][4] = *(const u32_ua*)(frame + q + 16); }
            else for (uint32_t b = 0; b < 20 && q + b < wend; b++) w[k][b >> 2] |= (uint32_t)frame[q + b] << (8 * (b & 3));
        }
#pragma unroll 1
        for (uint32_t k = 0; k < 4; k++) {
            const uint64_t q = p0 + (uint64_t)k * 4096u + t * 16u;
            // (a size word's top byte is 0x00 or 0x80: where none of the 16 positions' top bytes is, there is nothing to look at - see k_walk_cand)
            auto zb = [](uint32_t x) { const uint32_t y = x & 0x7F7F7F7Fu; return (y - 0x01010101u) & 0x80808080u; };
            if (((zb(w[k][0]) & 0x80000000u) | zb(w[k][1]) | zb(w[k][2]) | zb(w[k][3]) | (zb(w[k][4]) & 0x00808080u)) == 0u) continue;
            uint32_t plausible = 0;
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) {
                const uint32_t sh = (i & 3) * 8;
                const uint32_t v = sh ? (w[k][i >> 2] >> sh) | (w[k][(i >> 2) + 1] << (32 - sh)) : w[k][i >> 2];
                plausible |= ((v & hi_mask) == 0 && v != 0) ? 1u << i : 0u;
            }
#pragma unroll 1
            while (plausible) {
                const uint32_t i = (uint32_t)__builtin_ctz(plausible);
                plausible &= plausible - 1;
                if (q + i + 4 > wend) continue;
                const uint32_t at = atomicAdd(&s_n, 1u);
                if (at < WK_SEED_LIST) s_list[at] = (uint32_t)(q + i - pp);      // (a list that overflows - payload full of small words - loses finds: fewer seeds, never wrong ones)
            }
        }
    }
    __syncthreads();
    const uint32_t n = s_n < WK_SEED_LIST ? s_n : WK_SEED_LIST;
    for (uint32_t j = t; j < n; j += 256) {
        const uint32_t o = s_list[j];
        uint64_t c = pp + o, nx = 0; bool end = false, good = true;
#pragma unroll 1
        for (uint32_t h = 0; h < WK_SEED_HOPS && good && !end; h++) { good = walk_step(frame, frame_cap, bs, bck, c, nx, end); if (h == 0 && end) good = false; c = nx; }
        if (good) atomicMin(&s_best, o);
    }
    __syncthreads();
    if (t == 0 && s_best != 0xFFFFFFFFu) atomicMin(&seeds[s], (unsigned long long)(pp + s_best));
}
// one workgroup, a thread per seed: follow the size words from my seed to the next thread's (the positions wait in LDS), then - after a
// scan over the threads' counts - into the list
__global__ __launch_bounds__(WK_SEEDS) void k_walk_chains(const uint8_t* __restrict__ frame, uint64_t frame_cap, WalkState* __restrict__ ws,
                                                          const unsigned long long* __restrict__ seeds, uint64_t* __restrict__ list, uint32_t list_cap)
{
    __shared__ uint64_t s_seed[WK_SEEDS + 1];
    __shared__ uint32_t s_cnt[WK_SEEDS];
    __shared__ uint64_t s_pos[WK_LANE_CAP][WK_SEEDS];                    // [i][t]: thread t's i-th position (neighbouring threads in neighbouring banks)
    if (!ws->head_ok) return;
    const uint32_t bs = ws->bs, bck = ws->bck, t = threadIdx.x;
    s_seed[t] = seeds[t];
    if (t == 0) s_seed[WK_SEEDS] = WK_NOSEED;
    __syncthreads();
    uint64_t start = s_seed[t];
    for (uint32_t k = 0; k < t; k++) if (s_seed[k] != WK_NOSEED && s_seed[k] >= start) start = WK_NOSEED;      // (a seed another thread in front has, or has passed: windows closer together than the blocks are long)
    uint64_t stop = WK_NOSEED;
    for (uint32_t k = t + 1; k < WK_SEEDS; k++) if (s_seed[k] != WK_NOSEED && s_seed[k] > start) { stop = s_seed[k]; break; }
    uint64_t p = start; uint32_t c = 0; bool over = false;
    while (p != WK_NOSEED && p < stop) {
        uint64_t nx = 0; bool end = false;
        if (!walk_step(frame, frame_cap, bs, bck, p, nx, end) || end) break;
        if (c >= WK_LANE_CAP) { over = true; break; }
        s_pos[c][t] = p;
        c++; p = nx;
    }
    s_cnt[t] = over ? 0xFFFFFFFFu : c;
    __syncthreads();
    uint32_t base = 0, tot = 0; bool any_over = false;
    for (uint32_t k = 0; k < WK_SEEDS; k++) { if (s_cnt[k] == 0xFFFFFFFFu) { any_over = true; break; } if (k == t) base = tot; tot += s_cnt[k]; }
    if (any_over || tot > list_cap) { if (t == 0) ws->overflow = 1; return; }      // (uniform: every thread sees the same counts; k_walk_frame walks)
    for (uint32_t i = 0; i < c; i++) list[base + i] = s_pos[i][t];
    if (t == 0) ws->total = tot;
}

// A false candidate (a payload position whose bytes happen to chain) would break the list.  A true block start - but for the
// first - is where another candidate's size word points, a false one practically never is: mark every candidate's
// successor, then keep the first candidate and the marked ones.
__global__ __launch_bounds__(256) void k_walk_mark(const uint8_t* __restrict__ frame, uint64_t frame_cap, const WalkState* __restrict__ ws,
                                                   const uint64_t* __restrict__ list, uint32_t* __restrict__ mark)
{
    if (!ws->head_ok || ws->overflow) return;
    const uint32_t total = ws->total, bs = ws->bs, bck = ws->bck;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        uint64_t nx = 0; bool end = false;
        if (!walk_step(frame, frame_cap, bs, bck, list[i], nx, end) || end) continue;
        uint32_t lo = i + 1, hi = total;                                // first entry >= nx (the list is sorted; the successor lies behind me)
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (list[mid] < nx) lo = mid + 1; else hi = mid; }
        if (lo < total && list[lo] == nx) mark[lo] = 1;
    }
}
// one workgroup: the kept candidates, in order, into list2 (ws->total becomes their number)
__global__ __launch_bounds__(1024) void k_walk_filter(WalkState* __restrict__ ws, const uint64_t* __restrict__ list, const uint32_t* __restrict__ mark, uint64_t* __restrict__ list2)
{
    __shared__ uint32_t s_part[1024];
    __shared__ uint32_t s_carry;
    if (!ws->head_ok || ws->overflow) return;
    const uint32_t t = threadIdx.x, total = ws->total;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < total; base += 1024) {
        const uint32_t i = base + t;
        const uint32_t k = (i < total && (i == 0 || mark[i])) ? 1u : 0u;
        s_part[t] = k;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const uint32_t a = t >= off ? s_part[t - off] : 0u;
            __syncthreads();
            s_part[t] += a;
            __syncthreads();
        }
        if (k) list2[s_carry + s_part[t] - 1] = list[i];
        __syncthreads();
        if (t == 1023) s_carry += s_part[1023];
        __syncthreads();
    }
    if (t == 0) ws->total = s_carry;
}

__global__ __launch_bounds__(256) void k_walk_link(const uint8_t* __restrict__ frame, uint64_t frame_cap, uint64_t dst_cap, WalkState* __restrict__ ws,
                                                   const uint64_t* __restrict__ list, BlockOut* __restrict__ table, uint32_t table_cap)
{
    if (!ws->head_ok || ws->overflow) return;
    const uint32_t total = ws->total;
    const uint32_t bs = ws->bs, bck = ws->bck;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const uint64_t p = list[i];
        uint64_t nx = 0; bool end = false;
        bool ok = walk_step(frame, frame_cap, bs, bck, p, nx, end) && !end;
        bool last = false;
        if (ok) {
            uint64_t n2 = 0; bool e2 = false;
            const bool ok2 = walk_step(frame, frame_cap, bs, bck, nx, n2, e2);
            last = ok2 && e2;                                         // my size word points at the EndMark
            ok = last || (i + 1 < total && list[i + 1] == nx);
        }
        if (!ok) atomicMin(&ws->first_break, i);
        else if (last) atomicMin(&ws->first_end, i);
        if (ok && i < table_cap) {
            const uint64_t out = (uint64_t)i * bs;
            const uint32_t w = *(const u32_ua*)(frame + p);
            table[i].src_off = p + 4; table[i].dst_off = out; table[i].word = w;
            table[i].dst_size = out >= dst_cap ? 0u : (uint32_t)((dst_cap - out < bs) ? dst_cap - out : bs);
        }
    }
}

// the verdict: is the list, from the first byte behind the header to its first EndMark, one unbroken chain?
__global__ void k_walk_verdict(const uint8_t* __restrict__ frame, uint64_t frame_cap, uint64_t dst_cap, WalkState* __restrict__ ws,
                               const uint64_t* __restrict__ list, uint32_t table_cap, ResultRec* __restrict__ res)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!ws->head_ok || ws->overflow) return;
    const uint32_t hsize = ws->hsize, bs = ws->bs, bck = ws->bck, flg = ws->flg;
    uint32_t n = 0; uint64_t pos = hsize;
    if (ws->total == 0 || list[0] != hsize) {
        // no block at all?  then the EndMark follows the header
        if (frame_cap - pos < 4 || *(const u32_ua*)(frame + pos) != 0) return;
        pos += 4;
    } else {
        const uint32_t e = ws->first_end;
        if (e == 0xFFFFFFFFu || ws->first_break <= e || e >= table_cap) return;
        if ((uint64_t)e * bs >= dst_cap) return;                      // (no room: k_walk_frame says so)
        n = e + 1;
        const uint64_t p = list[e];
        pos = p + 4 + (*(const u32_ua*)(frame + p) & 0x7FFFFFFFu) + 4 * bck + 4;      // behind the EndMark
    }
    if ((flg >> 2) & 1) { if (frame_cap - pos < 4) return; pos += 4; }               // content checksum
    ResultRec r; r.size = ws->content; r.consumed = pos; r.status = ST_OK; r.n_blocks = n; r.first_bad_block = 0xFFFFFFFFu; r.flags = flg;
    *res = r;
    ws->done = 1;
}

// ------------------------------- the frame's trailer (in-band block list + sequence index) ------
// What this library's own decoder can use - where every block's size word sits, and the compressor's sequence index (encode.cuh) -
// travels IN the byte stream, as a skippable frame (magic 0x184D2A5E) right behind the LZ4 frame: liblz4, the `lz4` tool and
// the reference's decompress conduit skip or never reach it, this decoder finds it from its last 32 bytes.
//   frame | 5E 2A 4D 18 | u32 size | pad to 16 | u64 word_pos[n_blocks (+1 to even)] | index (IxHeader ... entries) or nothing |
//         | footer (32 bytes): u32 sequences, u32 entries of the index (for sizing the decoder's workspace), 8 zero bytes,
//         |                    u32 'LZIX', u32 n_blocks, u64 trailer bytes (from the magic on)
// Nothing in it is trusted: the positions are accepted only if they are the chain the size words themselves form (k_walk_link /
// k_walk_verdict, as for the candidates of the parallel walk), the index only as far as k_parse_indexed can follow it in the payload.
constexpr uint32_t TR_MAGIC = 0x184D2A5Eu, TR_FOOT = 0x58495A4Cu;
struct TrailerFoot { uint32_t total_seqs, total_entries, pad0, pad1, magic, n_blocks; uint64_t total; };      // 32 bytes; the last 16 identify it
struct TrailerPlan { uint64_t at, list_at, ix_at, ix_bytes, total; uint32_t n_list, ok; };

__global__ void k_trailer_plan(uint8_t* __restrict__ dst, uint64_t dst_cap, ResultRec* __restrict__ res, uint32_t n_blocks, const void* __restrict__ ix,
                               uint64_t ix_fixed, TrailerPlan* __restrict__ plan)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    TrailerPlan p; memset(&p, 0, sizeof(p));
    if (res->status == ST_OK && n_blocks) {
        const uint64_t F = res->size;
        const IxHeader* hd = (const IxHeader*)ix;
        const bool with_ix = ix && hd->magic == IX_MAGIC;                        // (round 4: linked frames too - until then their decoder indexed them itself, 0.3 ms per GiB of 64 KiB blocks)
        p.at = F;
        p.list_at = (F + 8 + 15) & ~(uint64_t)15;
        p.n_list = (n_blocks + 1) & ~1u;
        p.ix_at = p.list_at + (uint64_t)p.n_list * 8;
        p.ix_bytes = with_ix ? (ix_fixed + (uint64_t)hd->total_entries * sizeof(IxEntry) + 15) & ~(uint64_t)15 : 0;
        p.total = p.ix_at + p.ix_bytes + sizeof(TrailerFoot) - F;
        p.ok = (F + p.total <= dst_cap && p.total - 8 < 0xFFFFFFFFull) ? 1u : 0u;
        if (p.ok) {
            uint8_t* t = dst + F;
            const uint32_t sz = (uint32_t)(p.total - 8);
            t[0] = 0x5E; t[1] = 0x2A; t[2] = 0x4D; t[3] = 0x18; t[4] = (uint8_t)sz; t[5] = (uint8_t)(sz >> 8); t[6] = (uint8_t)(sz >> 16); t[7] = (uint8_t)(sz >> 24);
            for (uint64_t q = F + 8; q < p.list_at; q++) dst[q] = 0;
            TrailerFoot f{with_ix ? hd->total_seqs : 0u, with_ix ? hd->total_entries : 0u, 0u, 0u, TR_FOOT, n_blocks, p.total};
            memcpy(dst + p.ix_at + p.ix_bytes, &f, sizeof(f));
            res->size = F + p.total;
        } else res->status = ST_DSTSMALL;        // the caller asked for the trailer (LZ4F_MI355X_INBAND) and it does not fit: say so instead of leaving it out
    }
    *plan = p;
}
__global__ __launch_bounds__(256) void k_trailer_copy(uint8_t* __restrict__ dst, const TrailerPlan* __restrict__ plan, const BlockOut* __restrict__ table,
                                                      uint32_t n_blocks, const void* __restrict__ ix)
{
    const TrailerPlan p = *plan;
    if (!p.ok) return;
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x, step = (uint64_t)gridDim.x * 256;
    uint64_t* list = (uint64_t*)(dst + p.list_at);                   // (16-byte aligned within the frame buffer; the caller's buffer is)
    for (uint64_t i = t; i < p.n_list; i += step) list[i] = i < n_blocks ? table[i].src_off - 4 : 0;
    const uint4* s = (const uint4*)ix; uint4* d = (uint4*)(dst + p.ix_at);
    for (uint64_t i = t; i < p.ix_bytes / 16; i += step) d[i] = s[i];
}

// ------------------------------- block decode ---------------------------------------------------
__global__ void k_init_result(ResultRec* res, uint32_t n_blocks, uint32_t flags)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        res->size = 0; res->consumed = 0; res->status = ST_OK; res->n_blocks = n_blocks; res->first_bad_block = 0xFFFFFFFFu; res->flags = flags;
    }
}
// A block table that came from the caller (lz4f_mi355x_dev_decompressBlocks / ...Indexed): every entry inside the frame and the output
// buffer, and the entries in order without overlap on either side - the blocks are decoded concurrently, so two entries that share output
// bytes would be a race, and entries that share frame bytes are not a walk of any frame.  A table that fails gets ERROR_GENERIC and
// first_bad_block; nothing is decoded (every decode kernel returns at once on a result that is not OK).
__device__ __forceinline__ void check_table_entries(const BlockOut* __restrict__ table, uint32_t n, uint64_t frame_cap, uint64_t dst_cap, uint32_t block_size,
                                                    uint32_t bck, uint32_t linked, ResultRec* res)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const BlockOut e = table[i];
        const uint64_t csz = e.word & 0x7FFFFFFFu;
        bool bad = csz > block_size || e.src_off > frame_cap || frame_cap - e.src_off < csz + 4ull * bck;
        bad = bad || e.dst_off > dst_cap || e.dst_size > block_size || (!linked && dst_cap - e.dst_off < e.dst_size);
        if (!bad && i + 1 < n) {
            const BlockOut x = table[i + 1];
            bad = x.src_off < e.src_off + csz + 4ull * bck + 4ull;                           // (the next payload lies behind this one, its checksum and the next size word)
            if (!linked) bad = bad || x.dst_off < e.dst_off + e.dst_size;                    // (linked frames: positions are the decoder's, packed as it goes)
        }
        if (bad) { atomicMin(&res->first_bad_block, i); atomicMax(&res->status, (uint32_t)ST_GENERIC); }
    }
}
__global__ __launch_bounds__(256) void k_check_table(const BlockOut* __restrict__ table, uint32_t n, uint64_t frame_cap, uint64_t dst_cap, uint32_t block_size,
                                                     uint32_t bck, uint32_t linked, ResultRec* __restrict__ res)
{
    check_table_entries(table, n, frame_cap, dst_cap, block_size, bck, linked, res);
}
// The same for a table of a few blocks (the streaming API: one block per call) in ONE launch with what else a decode call begins with: the
// result record's initial state and the verdict words of the finishing kernels (`w`: [0] block-checksum verdict, [1] first failed block,
// [2] "something has to move", [4..5] sum of the sizes).  One workgroup: the record is set before anybody checks against it.
__global__ __launch_bounds__(256) void k_begin_table_small(const BlockOut* __restrict__ table, uint32_t n, uint64_t frame_cap, uint64_t dst_cap, uint32_t block_size,
                                                           uint32_t bck, uint32_t linked, ResultRec* res, uint32_t* __restrict__ w)
{
    if (threadIdx.x == 0) { res->size = 0; res->consumed = 0; res->status = ST_OK; res->n_blocks = n; res->first_bad_block = 0xFFFFFFFFu; res->flags = 0u; }
    if (threadIdx.x < 8) w[threadIdx.x] = threadIdx.x < 2 ? 0xFFFFFFFFu : 0u;
    __threadfence_block(); __syncthreads();
    check_table_entries(table, n, frame_cap, dst_cap, block_size, bck, linked, res);
}
__global__ void k_set_block(BlockOut* t, BlockOut e) { if (threadIdx.x == 0 && blockIdx.x == 0) *t = e; }

// Which decoder for a frame of big independent blocks that came without a usable index?  The fused workgroups parse on the scalar
// unit - right for long sequences (a block of synth50 is 4 k of them), hopeless for text (320 k sequences per 4 MiB block: 190 ms
// per GiB) - where the wave-per-block decoder, whose lanes find the tokens, takes a fifth of that although it leaves most of the
// machine idle.  So 64 lanes each read the first 512 payload bytes of a block (spread over the frame) and count sequences - not
// the first one, which is the literal run of a block that has nothing to refer to yet (at 192 bytes it alone made text look sparse):
// under 24 payload bytes per sequence -> flags[0] = 1 (dense: k_decode_blocks), else flags[1] = 1 (k_decode_blocks_fused).
__global__ __launch_bounds__(64) void k_density_probe(const uint8_t* __restrict__ frame, uint64_t frame_cap, const BlockOut* __restrict__ table,
                                                      const ResultRec* __restrict__ res, uint32_t n_max, uint32_t* __restrict__ flags)
{
    const uint32_t lane = lane_id();
    uint32_t seqs = 0, bytes = 0;
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    if (res->status == ST_OK && n) {
        const uint32_t b = (uint32_t)(((uint64_t)lane * n) >> 6);
        const BlockOut e = table[b];
        const uint32_t csz = e.word & 0x7FFFFFFFu;
        if (!(e.word >> 31) && csz >= 64 && e.src_off + csz <= frame_cap && (lane == 0 || b != (uint32_t)(((uint64_t)(lane - 1) * n) >> 6))) {
            const uint8_t* in = frame + e.src_off;
            // (few blocks: each lane's look is longer - 32 KiB over the frame either way, 8 KiB per block at most.  A block's first bytes have nothing to refer
            // to: the first 512 bytes of three blocks of text are mostly literals, 25-30 payload bytes per sequence, and the frame went to the scalar parser)
            const uint32_t want = n >= 64u ? 512u : n >= 4u ? 512u * (64u / n) : 8192u;
            const uint32_t lim = csz < want ? csz - 16u : want - 16u;
            uint32_t pos = 0, first_end = 0;
            while (pos < lim && seqs < 2048u) {                          // (lengths only; a malformed payload just gives a number)
                const uint32_t t = in[pos++];
                uint32_t lit = t >> 4;
                if (lit == 15) { uint32_t x; do { x = pos < lim ? in[pos] : 0u; pos++; lit += x; } while (x == 255u && pos < lim); }
                pos += lit + 2;
                if ((t & 15u) == 15u) { uint32_t x; do { x = pos < lim ? in[pos] : 0u; pos++; } while (x == 255u && pos < lim); }
                seqs++;
                if (seqs == 1) first_end = pos;
            }
            if (seqs > 1) { seqs -= 1; bytes = (pos < want + 512u ? pos : want + 512u) - (first_end < pos ? first_end : pos); } else { seqs = 0; bytes = 0; }
        }
    }
    uint32_t st = seqs, bt = bytes;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) { st += __shfl_xor(st, sft); bt += __shfl_xor(bt, sft); }
    if (lane == 0) { const bool dense = st != 0 && bt < 24u * st; flags[0] = dense ? 1u : 0u; flags[1] = dense ? 0u : 1u; flags[2] = st ? bt / st : 0u; }      // [2]: payload bytes per sequence in the sample (0: no sequence seen)
}

// `hist0`: valid bytes directly in front of dst (streaming API: the previous blocks' last 64 KiB).
template <int WAVES_PER_WG>
__global__ __launch_bounds__(64 * WAVES_PER_WG, 8) void k_decode_blocks(const uint8_t* __restrict__ frame, uint8_t* dst, uint64_t dst_cap,
                                                                     BlockOut* __restrict__ table, const ResultRec* __restrict__ res,
                                                                     uint32_t n_max, uint32_t linked, uint32_t block_size, uint64_t hist0,
                                                                     uint64_t frame_cap, const uint32_t* __restrict__ only_if = nullptr)
{
    __shared__ uint32_t expand[WAVES_PER_WG][64];
    const uint32_t w = uni(blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6));
    if (res->status != ST_OK) return;
    if (only_if && *only_if == 0) return;                                // (k_density_probe sent the frame to the other decoder)
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t lane = lane_id();
    if (!linked) {
        if (w >= n) return;
        const BlockOut e = table[w];
        const uint32_t csz = e.word & 0x7FFFFFFFu;
        int32_t got;
        if (e.word >> 31) {
            if (csz > e.dst_size) got = -2;
            else { wave_copy_disjoint(dst + e.dst_off, frame + e.src_off, csz); got = (int32_t)csz; }
        } else {
            got = wave_decode_block_win<true>(frame + e.src_off, csz, frame_cap - e.src_off, dst + e.dst_off, e.dst_size, expand[uni(threadIdx.x >> 6)]);
        }
        if (lane == 0) table[w].dst_size = (uint32_t)got;            // negative = failed
        return;
    }
    // linked frame: one wave, blocks in order, output packed (each block sees the bytes before it)
    if (w != 0) return;
    uint64_t out = 0;
    for (uint32_t b = 0; b < n; b++) {
        const BlockOut e = table[b];
        const uint32_t csz = e.word & 0x7FFFFFFFu;
        const uint32_t room = (uint32_t)((dst_cap - out < block_size) ? dst_cap - out : block_size);
        int32_t got;
        if (e.word >> 31) {
            if (csz > room) got = -2;
            else { wave_copy_disjoint(dst + out, frame + e.src_off, csz); got = (int32_t)csz; }
        } else {
            got = wave_decode_block(frame + e.src_off, csz, dst + out, room, out + hist0);
        }
        if (lane == 0) { table[b].dst_off = out; table[b].dst_size = (uint32_t)got; }
        if (got < 0) {
            for (uint32_t k = b + 1 + lane; k < n; k += WAVE) table[k].dst_size = 0;   // not decoded
            break;
        }
        out += (uint32_t)got;
    }
}

// totals + status; compacts the output if a non-final block of an independent frame decoded short
// What k_finish_decode has to know about the block table, found by a thread per block instead of one wave walking it 64
// entries per dependent load (16384 blocks: 0.22 ms): the first failed block, whether every block already sits where its
// predecessors end (independent frames: all blocks but the last decoded to full size, at b * blockSize), the sum of the sizes.
//   w[0] block-checksum verdict (k_xxh32_blocks), w[1] first failed block (preset 0xFFFFFFFF), w[2] != 0: something has to move,
//   w[4..5] sum of the decoded sizes (preset 0)
__global__ __launch_bounds__(256) void k_finish_check(const BlockOut* __restrict__ table, const ResultRec* __restrict__ res, uint32_t n_max,
                                                      uint32_t linked, uint32_t block_size, uint32_t* __restrict__ w)
{
    if (res->status != ST_OK) return;
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t b = blockIdx.x * 256u + threadIdx.x, lane = lane_id();
    const bool in = b < n;
    const uint32_t sz = in ? table[b].dst_size : 0u;
    const uint64_t at = in ? table[b].dst_off : 0ull;
    const bool bad = in && (int32_t)sz < 0;
    if (bad) atomicMin(&w[1], b);
    const bool moves = in && !linked && (at != (uint64_t)b * block_size || (b + 1 < n && sz != block_size));
    const uint64_t mv = __ballot(moves);
    uint32_t sum = bad ? 0u : sz;                                      // (<= 64 * 4 MiB per wave: fits)
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) sum += __shfl_xor(sum, sft);
    if (lane == 0) {
        if (mv) atomicOr(&w[2], 1u);
        if (sum) atomicAdd((unsigned long long*)(w + 4), (unsigned long long)sum);
    }
}

__global__ __launch_bounds__(64) void k_finish_decode(uint8_t* dst, BlockOut* __restrict__ table, ResultRec* res, uint32_t n_max,
                                                      uint32_t linked, uint32_t block_size, const uint32_t* __restrict__ bad_ck, uint32_t with_ck,
                                                      uint32_t plan = 0, const uint32_t* __restrict__ ix_flags = nullptr, uint32_t check_here = 0,
                                                      uint64_t dst_cap = ~0ull)
{   // check_here (n_max <= 64): what k_finish_check leaves in bad_ck[1], [2], [4..5] is worked out by this wave itself - a launch less for calls of a few blocks
    // which way the call went (lz4f_mi355x.h: LZ4F_MI355X_PATH_*): what the host launched, and whether the indexed kernels gave up
    if (lane_id() == 0) res->flags = (res->flags & 0xFFFu) | (plan << 12) | ((ix_flags && *ix_flags) ? (LZ4F_MI355X_PATH_INDEX_DROPPED << 12) : 0u);
    if (res->status != ST_OK) return;
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    uint32_t n_scan = n;
    const uint32_t lane = lane_id();
    uint32_t w1 = 0xFFFFFFFFu, w2 = 0u; uint64_t w45 = 0;
    if (check_here) {                                                  // (k_finish_check's rules, a block per lane)
        const bool in = lane < n;
        const uint32_t sz = in ? table[lane].dst_size : 0u;
        const uint64_t at = in ? table[lane].dst_off : 0ull;
        const bool bad = in && (int32_t)sz < 0;
        const uint64_t bm = __ballot(bad);
        if (bm) w1 = (uint32_t)__builtin_ctzll(bm);
        w2 = __ballot(in && !linked && (at != (uint64_t)lane * block_size || (lane + 1 < n && sz != block_size))) ? 1u : 0u;
        uint64_t sum = bad ? 0u : sz;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) sum += __shfl_xor(sum, sft);
        w45 = sum;
    } else { w1 = bad_ck[1]; w2 = bad_ck[2]; w45 = *(const uint64_t*)(bad_ck + 4); }
    // first failed block (k_finish_check)
    const uint32_t first_bad = w1 < n ? w1 : 0xFFFFFFFFu;
    const uint32_t bad_kind = first_bad != 0xFFFFFFFFu ? table[first_bad].dst_size : 0u;
    const uint32_t ck = with_ck ? bad_ck[0] : 0xFFFFFFFFu;
    if (ck != 0xFFFFFFFFu && ck <= first_bad) {
        if (lane == 0) { res->status = ST_BLOCKCK; res->first_bad_block = ck; }
        return;
    }
    if (first_bad != 0xFFFFFFFFu) {
        // (a block that did not decode into LESS room than a whole block is "the output does not fit", as the oracle's frame decoder and
        // the host paths call it: oracle/orc_lz4frame.c "room < bs ? dstMaxSize_tooSmall : GENERIC")
        const uint64_t at_bad = table[first_bad].dst_off;
        const bool short_room = at_bad <= dst_cap && dst_cap - at_bad < block_size;
        if (lane == 0) { res->status = (bad_kind == (uint32_t)-2 || short_room) ? ST_DSTSMALL : ST_GENERIC; res->first_bad_block = first_bad; }
        return;
    }
    // usual case, checked 64 blocks per step: every block already sits where its predecessors end (all blocks but
    // the last decoded to full size), so nothing has to move and the total is a sum
    uint64_t out = w45;
    bool in_place = w2 == 0;
    if (!in_place) { out = 0; in_place = true; }                       // (k_finish_check's rule is stricter than needed: look properly)
    else n_scan = 0;
    for (uint32_t b0 = 0; b0 < n_scan; b0 += WAVE) {
        const uint32_t b = b0 + lane;
        const uint32_t sz = b < n ? table[b].dst_size : 0u;
        const uint64_t at = b < n ? table[b].dst_off : 0ull;
        uint64_t incl = sz;
        for (uint32_t d = 1; d < WAVE; d <<= 1) { const uint64_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        if (!linked && __ballot(b < n && at != out + incl - sz)) { in_place = false; break; }
        out += __shfl(incl, WAVE - 1);
    }
    if (!in_place) {
        out = 0;
        for (uint32_t b = 0; b < n; b++) {          // uniform, serial: a rare path (a non-final block of an independent frame decoded short)
            const uint64_t at = table[b].dst_off; const uint32_t sz = table[b].dst_size;
            if (at != out) {
                for (uint32_t o = 0; o < sz; o += 1024) {            // forward move to a lower address
                    const uint32_t m = (sz - o < 1024) ? sz - o : 1024;
                    uint8_t v = 0; b16_ua v16;
                    const uint32_t i = lane * 16;
                    const bool full = i + 16 <= m;
                    if (full) v16 = *(const b16_ua*)(dst + at + o + i);
                    if (full) *(b16_ua*)(dst + out + o + i) = v16;
                    for (uint32_t j = (m & ~15u) + lane; j < m; j += WAVE) { v = dst[at + o + j]; dst[out + o + j] = v; }
                }
                if (lane == 0) table[b].dst_off = out;
            }
            out += sz;
        }
    }
    if (lane == 0) {
        const uint64_t declared = res->size;
        res->size = out;
        if ((res->flags >> 3) & 1) { if (declared != out) res->status = 14; }   // frameSize_wrong
    }
}

}  // namespace lz4f
